"""developer probe: does a registration from pageable HOST buffers leave the process in a slower state for later launches?"""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np
import torch; torch.cuda.init()
from qn_amd import engine, synth
N = 100000
src, tgt, _ = synth.make_pair(0, N)
ds, dt = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda(); torch.cuda.synchronize()
ctx = engine.Context(N + 1024)
g = engine.NanoGICP(ctx); g.setCorrespondenceRandomness(20); g.setMaximumIterations(20); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer("gn"); g.setForceIterations(20)
def reg():
    g.setInputSourceDevice(ds.data_ptr(), N, 12); g.calculateSourceCovariances(); g.setInputTargetDevice(dt.data_ptr(), N, 12); g.calculateTargetCovariances(); return g.align()
def t_align(tag):
    reg(); reg()
    al = []
    for _ in range(40):
        t0 = time.perf_counter(); g.align(); al.append(1e3 * (time.perf_counter() - t0))
    rg = []
    for _ in range(20):
        t0 = time.perf_counter(); reg(); rg.append(1e3 * (time.perf_counter() - t0))
    print("%-46s align median %.4f  registration median %.4f" % (tag, np.median(al), np.median(rg)))
t_align("fresh process")
g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances(); g.align()
t_align("after one registration from host buffers")
pin_s = torch.from_numpy(src).pin_memory()
t_align("after pinning a host tensor")
ctx.prof_reset(); ctx.prof_enable(True); reg(); ctx.synchronize(); ctx.prof_enable(False)
t_align("after a profiled registration (hipEvents)")
c2 = engine.Context(N + 1024)
t_align("after creating a second context")
c2.close()
t_align("after closing it")
