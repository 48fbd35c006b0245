"""Developer probe: per-iteration cost of the NN passes (forced GN iteration sweep)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np
from qn_amd import engine, synth
N = 100000
src, tgt, T = synth.make_pair(0, N)
ctx = engine.Context(N + 1024)
g = engine.NanoGICP(ctx)
g.setCorrespondenceRandomness(20); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer("gn")
g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
def counters():
    out = (C.c_uint32 * 16)(); ctx.check(ctx._l.qn_debug_get_counters(ctx.h, out)); return list(out)
prev = None
for n in [1, 2, 3, 4, 5, 6, 8, 12, 20]:
    g.setMaximumIterations(n); g.setForceIterations(n)
    g.align()
    ctx.debug_set("dbg_counters", 1)
    g.align(); ctx.synchronize(); c = counters(); ctx.debug_set("dbg_counters", 0)
    ctx.prof_reset(); ctx.prof_enable(True); g.align(); ctx.synchronize(); ctx.prof_enable(False)
    st = ctx.prof_stats()
    print("iters %2d: list entries(total incl fitness) %7d clusters %6d cand %8d retries %6d skipped %8d big %6d | ms: search %.3f list %.3f acc %.3f solve %.3f fit %.3f" % (
        n, c[5], c[0], c[1], c[3], c[6], c[7], st['nn_search'][0], st['nn_fallback'][0], st['accumulate'][0], st['solve'][0], st['fitness'][0]))
