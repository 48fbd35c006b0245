"""Developer probe (GPU box): search statistics + per-family timings for the bench workload."""
import ctypes as C, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np
from qn_amd import engine, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
src, tgt, T = synth.make_pair(0, N)
ctx = engine.Context(N + 1024)
for kv in sys.argv[2:]:
    k, v = kv.split("="); ctx.debug_set(k, float(v))
g = engine.NanoGICP(ctx)
g.setCorrespondenceRandomness(20); g.setMaximumIterations(20); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer("gn"); g.setForceIterations(20)

def counters():
    out = (C.c_uint32 * 16)(); ctx.check(ctx._l.qn_debug_get_counters(ctx.h, out)); return list(out)

ctx.debug_set("dbg_counters", 1)
g.setInputSource(src); print("grid", ctx.grid_info(0))
g.calculateSourceCovariances(); ctx.synchronize()
c = counters(); nw = (N + 63) // 64
print("kNN src: clusters/wave %.2f cand/cluster %.0f flushes/cluster %.2f retries %d fallback %d" % (c[0] / nw, c[1] / max(c[0], 1), c[2] / max(c[0], 1), c[3], c[4]))
ctx.debug_set("dbg_counters", 1)
g.setInputTarget(tgt); g.calculateTargetCovariances(); ctx.synchronize()
ctx.debug_set("dbg_counters", 1)
r = g.align(); ctx.synchronize()
c = counters(); it = 21
print("NN (20 it + fitness): clusters/wave %.2f cand/cluster %.0f retries/it %d fallback/it %d" % (c[0] / nw / it, c[1] / max(c[0], 1), c[3] / it, c[5] / it))
ctx.debug_set("dbg_counters", 0)
# timings
def reg():
    g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances(); return g.align()
for _ in range(3): reg()
t = time.perf_counter()
for _ in range(10): reg()
print("ms/registration (host clouds, incl. H2D): %.3f" % ((time.perf_counter() - t) * 100))
t = time.perf_counter()
for _ in range(10): g.align()
print("ms/align: %.3f" % ((time.perf_counter() - t) * 100))
ctx.prof_reset(); ctx.prof_enable(True)
for _ in range(3): reg()
ctx.synchronize(); ctx.prof_enable(False)
print({k: (round(v[0] / 3, 4), v[1] // 3) for k, v in ctx.prof_stats().items() if v[1]})
