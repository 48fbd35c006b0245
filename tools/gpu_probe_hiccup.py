"""Developer probe (GPU box): wall time of short batches (20 pairs, the driver's --steps 20) repeated many times - looks for multi-millisecond hiccups."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import torch
torch.cuda.init()
from qn_amd import engine, synth
N = 100000
pairs = []
for j in range(8):
    s, t, _ = synth.make_pair(j, N)
    pairs.append((torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()))
torch.cuda.synchronize()
for nctx, lanes in [(3, 8), (1, 20), (2, 10), (4, 1)]:
    ctxs = [engine.Context(N + 1024) for _ in range(nctx)]
    for cx in ctxs:
        cx.debug_set("batch_lanes", lanes)
        g = engine.NanoGICP(cx); g.setCorrespondenceRandomness(20); g.setMaximumIterations(20); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer("gn"); g.setForceIterations(20); g.bind()
    d = [(pairs[j % 8][0].data_ptr(), N, pairs[j % 8][1].data_ptr(), N, 12, 1) for j in range(20)]
    for _ in range(4):
        engine.icp_alignment_batch(ctxs, d)
    torch.cuda.synchronize()
    ts = []
    for _ in range(80):
        t0 = time.perf_counter(); engine.icp_alignment_batch(ctxs, d); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    s = sorted(ts)
    print("%dx%d  median %.2f ms  p90 %.2f  max %.2f  outliers(>1.5x median): %s" % (nctx, lanes, s[40], s[72], s[-1], [round(x, 1) for x in ts if x > 1.5 * s[40]]), flush=True)
    for cx in ctxs:
        cx.close()
