"""Developer probe (GPU box): 64 candidate targets against ONE query cloud (BASELINE configs[3] as a loop-closure tick sees it) through 3 contexts x 8 lanes, source sharing on."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np, torch
torch.cuda.init()
from qn_amd import engine, synth
N = 100000
s, t, _ = synth.make_pair(0, N)
ds, dt0 = torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()
tg = []
for v in range(64):
    a = 0.004 * (v + 1); ca, sa = float(np.cos(a)), float(np.sin(a))
    R = torch.tensor([[ca, -sa, 0.0], [sa, ca, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32, device="cuda")
    tg.append((dt0 @ R.T + torch.tensor([0.02 * v, -0.01 * v, 0.0], dtype=torch.float32, device="cuda")).contiguous())
torch.cuda.synchronize()
ctxs = [engine.Context(N + 1024) for _ in range(3)]
for cx in ctxs:
    cx.debug_set("batch_lanes", 8)
    g = engine.NanoGICP(cx); g.setCorrespondenceRandomness(20); g.setMaximumIterations(20); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer("gn"); g.setForceIterations(20); g.bind()
d = [(ds.data_ptr(), N, x.data_ptr(), N, 12, 1) for x in tg]
engine.icp_alignment_batch(ctxs, d); torch.cuda.synchronize()
for rep in range(4):
    t0 = time.perf_counter(); r, v, st = engine.icp_alignment_batch(ctxs, d); torch.cuda.synchronize(); w = time.perf_counter() - t0
    assert all(x == 0 for x in st)
    print("shared query: %.1f pairs/s (fitness[0] %.6f, fitness[63] %.6f)" % (64 / w, r[0].fitness, r[63].fitness), flush=True)
