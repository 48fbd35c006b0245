"""Developer probe: cost of the k-NN + covariance stage under different knobs."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np
from qn_amd import engine, synth
N = 100000
src, tgt, T = synth.make_pair(0, N)
ctx = engine.Context(N + 1024)
g = engine.NanoGICP(ctx)
g.setCorrespondenceRandomness(20)
g.setInputSource(src)
def counters():
    out = (C.c_uint32 * 16)(); ctx.check(ctx._l.qn_debug_get_counters(ctx.h, out)); return list(out)
def run(label, **kn):
    for k, v in kn.items(): ctx.debug_set(k, v)
    if "cell" in kn: g.setInputSource(src)
    g.calculateSourceCovariances(); ctx.synchronize()
    ctx.debug_set("dbg_counters", 1); g.calculateSourceCovariances(); ctx.synchronize(); c = counters(); ctx.debug_set("dbg_counters", 0)
    t0 = time.perf_counter()
    for _ in range(20): g.calculateSourceCovariances()
    ctx.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
    print("%-44s %.3f ms  clusters %6d cand %9d retries %6d list %6d" % (label, dt, c[0], c[1], c[3], c[4]))
run("old sorted-list", knn_hist=0)
for m in (1.5, 2.0, 2.5, 3.0):
    for rd in (1, 2):
        run("hist margin %.1f rounds %d" % (m, rd), knn_hist=1, margin_knn=m, knn_rounds=rd)
