"""Developer probe (GPU box): latency of ONE registration at a time (the reference's deployment, fast_lio_sam_qn.cpp:213-219) over the bench's mixed pairs (8 scenes x re-pose
variants 0 / 4 / 7) and the 80 %-overlap pairs: the default lone path (device look, persistent kernel) against the same context as a batch member (k_tick chain, unseeded
phase extended while the pose still moves by metres).  argv[1]: knobs for the lone path, e.g. '{"single_from_tick": 3}'."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np, torch
torch.cuda.init()
from qn_amd import engine, synth
N = 100000
knobs = eval(sys.argv[1]) if len(sys.argv) > 1 else {}
scenes = []
for j in range(8):
    s, t, _ = synth.make_pair(j, N); scenes.append((torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()))
sets = {}
for v in (0, 4, 7):
    out = []
    for s, t in scenes:
        if v:
            a = 0.01 * v; ca, sa = float(np.cos(a)), float(np.sin(a))
            R = torch.tensor([[ca, -sa, 0.0], [sa, ca, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32, device=t.device)
            t = (t @ R.T + torch.tensor([0.05 * v, -0.03 * v, 0.0], dtype=torch.float32, device=t.device)).contiguous()
        out.append((s, t))
    sets["variant%d" % v] = out
ov = []
for j in range(8):
    s, t, _ = synth.make_pair(9000 + j, N, shift=24.0); ov.append((torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()))
sets["overlap80"] = ov
torch.cuda.synchronize()
for mode in ("lone", "batch_member"):
    ctx = engine.Context(N + 1024)
    if mode == "batch_member":
        ctx.debug_set("batch_member", 1); ctx.debug_set("pair_pipeline", 0)
    else:
        for k_, v_ in knobs.items(): ctx.debug_set(k_, v_)
    g = engine.NanoGICP(ctx); g.setCorrespondenceRandomness(20); g.setMaximumIterations(20); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer("gn"); g.setForceIterations(20); g.bind()
    def register(s, t):
        g.setInputSourceDevice(s.data_ptr(), N, 12); g.calculateSourceCovariances(); g.setInputTargetDevice(t.data_ptr(), N, 12); g.calculateTargetCovariances(); return g.align()
    for name, ps in sets.items():
        for s, t in ps[:2]: register(s, t)
        lat = []
        for s, t in ps:
            w = []
            for _ in range(3):
                t0 = time.perf_counter(); register(s, t); w.append(1e3 * (time.perf_counter() - t0))
            lat.append(min(w))
        print("LONE %-12s %-10s per pair ms: %s  median %.3f max %.3f" % (mode, name, " ".join("%.2f" % x for x in lat), float(np.median(lat)), max(lat)), flush=True)
    ctx.close()
