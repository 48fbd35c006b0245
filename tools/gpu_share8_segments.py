"""Developer probe (GPU box): host timeline (knob batch_trace) of ONE 8-pair qn_multi_align_best call for an easy block (re-pose variant 0) and the hardest one (variant 7)
of the bench's 64 mixed pairs: which segments the call is made of and what each costs (prep + enqueue vs synchronisation)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np, torch, ctypes as C
torch.cuda.init()
from qn_amd import engine, synth
N = 100000
scenes = []
for j in range(8):
    s, t, _ = synth.make_pair(j, N); scenes.append((torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()))
def block(v):
    out = []
    for i in range(8):
        s, t = scenes[i]
        if v:
            a = 0.01 * v; ca, sa = float(np.cos(a)), float(np.sin(a))
            R = torch.tensor([[ca, -sa, 0.0], [sa, ca, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32, device=t.device)
            t = (t @ R.T + torch.tensor([0.05 * v, -0.03 * v, 0.0], dtype=torch.float32, device=t.device)).contiguous()
        out.append((s, t))
    return out
knobs = eval(sys.argv[1]) if len(sys.argv) > 1 else {}
for v in (0, 4, 7):
    pairs = block(v); torch.cuda.synchronize()
    d = [(s.data_ptr(), N, t.data_ptr(), N, 12, 1) for s, t in pairs]
    mg = engine.MultiGpu(1, N + 1024, in_flight=3)
    p = engine.GicpParams(); engine.lib().qn_gicp_default_params(C.byref(p))
    p.k_correspondences, p.max_iterations, p.max_corr_dist, p.optimizer, p.force_iterations = 20, 20, 52.5, 1, 20
    mg.set_params(p); mg.debug_set("batch_lanes", 8); mg.debug_set("batch_share_source", 0)
    for k_, v_ in knobs.items(): mg.debug_set(k_, v_)
    for _ in range(3): mg.align_best(d)
    w = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); mg.align_best(d); w.append(1e3 * (time.perf_counter() - t0))
    print("SEG variant %d: 8-pair call median %.3f ms (%s)" % (v, float(np.median(w)), [round(x, 2) for x in w]), flush=True)
    mg.debug_set("batch_trace", 1)
    sys.stderr.write("== variant %d\n" % v); sys.stderr.flush()
    mg.align_best(d)
    mg.debug_set("batch_trace", 0)
    mg.close()
