"""Developer probe (GPU box): the pose step of every Gauss-Newton iteration (max |t|, max |R - I| from the iteration trace; moved = dt + dR x the source's reach from the
origin, the quantity of the hand-over rule - look_decide, qn_gicp_kernels.cuh) for the bench's mixed pairs (8 scenes x re-pose variants) and for 80 %-overlap pairs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np, torch, ctypes as C
torch.cuda.init()
from qn_amd import engine, synth
N = 100000
def run(pairs, tag):
    ctx = engine.Context(N + 1024); ctx.debug_set("batch_lanes", 8); ctx.debug_set("batch_share_source", 0)
    p = engine.GicpParams(); engine.lib().qn_gicp_default_params(C.byref(p))
    p.k_correspondences, p.max_iterations, p.max_corr_dist, p.optimizer, p.force_iterations = 20, 20, 52.5, 1, 20
    ctx.check(engine.lib().qn_gicp_set_params(ctx.h, C.byref(p)))
    d = [(s.data_ptr(), N, t.data_ptr(), N, 12, 1) for s, t in pairs]
    engine.gicp_align_batch(ctx, d)
    for l, (s, t) in enumerate(pairs):
        reach = float(torch.linalg.norm(torch.maximum(s.max(0).values.abs(), s.min(0).values.abs())))
        tr = engine.lane_trace(ctx, l)
        mv = [x[4] + x[3] * reach for x in tr]
        print("STEP %s lane %d reach %.0f moved/iter: %s" % (tag, l, reach, " ".join("%.3f" % m for m in mv[:10])), flush=True)
    g = ctx.grid_info(1); print("STEP %s target cell %.3f" % (tag, g["cell"]))
    ctx.close()
scenes = []
for j in range(8):
    s, t, _ = synth.make_pair(j, N); scenes.append((torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()))
for v in (0, 4, 7):
    out = []
    for i in range(8):
        s, t = scenes[i]
        if v:
            a = 0.01 * v; ca, sa = float(np.cos(a)), float(np.sin(a))
            R = torch.tensor([[ca, -sa, 0.0], [sa, ca, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32, device=t.device)
            t = (t @ R.T + torch.tensor([0.05 * v, -0.03 * v, 0.0], dtype=torch.float32, device=t.device)).contiguous()
        out.append((s, t))
    run(out, "variant%d" % v)
ov = []
for j in range(8):
    s, t, _ = synth.make_pair(j, N, shift=24.0); ov.append((torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()))
run(ov, "overlap80")
