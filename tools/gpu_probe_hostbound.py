"""Developer probe: is the batch throughput host-launch-bound?  Same batch at several cloud sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np, torch
from qn_amd import engine, synth
KNOB = float(sys.argv[1]) if len(sys.argv) > 1 else 4
for N in (100000, 25000, 6000):
    src, tgt, T = synth.make_pair(0, N, extent=120.0 if N >= 30000 else 40.0)
    s = torch.from_numpy(src).cuda(); t = torch.from_numpy(tgt).cuda(); torch.cuda.synchronize()
    for C_ in (1, 4):
        ctxs = [engine.Context(N + 1024) for _ in range(C_)]
        for cx in ctxs:
            g = engine.NanoGICP(cx); g.setCorrespondenceRandomness(20); g.setMaximumIterations(20); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer("gn"); g.setForceIterations(20)
        descs = [(s.data_ptr(), N, t.data_ptr(), N, 12, 1)] * 24
        engine.icp_alignment_batch(ctxs, descs[:8])
        t0 = time.perf_counter(); engine.icp_alignment_batch(ctxs, descs); dt = time.perf_counter() - t0
        print("N=%6d in_flight=%d: %.3f ms/registration" % (N, C_, dt / 24 * 1e3))
        for cx in ctxs: cx.close()
