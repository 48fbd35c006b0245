"""Developer probe: throughput with several contexts (streams) driven by host threads."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np, torch
from qn_amd import engine, synth
N = 100000
src, tgt, T = synth.make_pair(0, N)
s = torch.from_numpy(src).cuda(); t = torch.from_numpy(tgt).cuda(); torch.cuda.synchronize()
def mk():
    ctx = engine.Context(N + 1024); g = engine.NanoGICP(ctx)
    g.setCorrespondenceRandomness(20); g.setMaximumIterations(20); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer("gn"); g.setForceIterations(20)
    return ctx, g
def reg(g):
    g.setInputSourceDevice(s.data_ptr(), N, 12); g.calculateSourceCovariances(); g.setInputTargetDevice(t.data_ptr(), N, 12); g.calculateTargetCovariances(); return g.align()
for C_ in (1, 2, 3, 4, 6, 8):
    objs = [mk() for _ in range(C_)]
    for _, g in objs: reg(g)
    K = 24
    def work(g, n):
        for _ in range(n): reg(g)
    th = [threading.Thread(target=work, args=(g, K // C_)) for _, g in objs]
    t0 = time.perf_counter(); [x.start() for x in th]; [x.join() for x in th]; dt = time.perf_counter() - t0
    print("contexts %d: %.1f registrations/s (%.3f ms amortised)" % (C_, (K // C_) * C_ / dt, dt / ((K // C_) * C_) * 1e3))
    for ctx, _ in objs: ctx.close()
