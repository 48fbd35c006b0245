"""developer probe: list sizes of the first four unseeded passes (fixed hand-over at tick 4), per pair"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np
from qn_amd import engine, synth
N = 100000
for pid in [int(a) for a in sys.argv[1:]] or [7, 3]:
    src, tgt, T = synth.make_pair(pid, N)
    ctx = engine.Context(N + 1024)
    ctx.debug_set("single_from_tick", 0); ctx.debug_set("track_from_tick", 5); ctx.debug_set("fused_from_tick", 5)
    g = engine.NanoGICP(ctx)
    g.setCorrespondenceRandomness(20); g.setMaximumIterations(20); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer("gn")
    g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
    def counters():
        out = (C.c_uint32 * 16)(); ctx.check(ctx._l.qn_debug_get_counters(ctx.h, out)); return list(out)
    prev = (0, 0)
    for it in (1, 2, 3, 4, 5):
        g.setForceIterations(it)
        ctx.debug_set("dbg_counters", 1); r = g.align(); ctx.synchronize(); c = counters(); ctx.debug_set("dbg_counters", 0)
        print("pair %d forced %d: this tick 16-per-wave %7d  one-per-wave %7d  (counters %s)" % (pid, it, c[5] - prev[0], c[7] - prev[1], c[:10]))
        prev = (c[5], c[7])
        f = lambda u: float(np.array([u], dtype=np.uint32).view(np.float32)[0])
        print("      slowest one-per-wave entry so far: %.1f us  sorted position %d  start radius %.3f  neighbour at %.3f m  query (%.2f, %.2f)  [cell %.3f]" % (c[10] / 100.0, c[11], f(c[12]), f(c[13]) if c[13] != 0xffffffff else -1, f(c[14]), f(c[15]), ctx.grid_info(1)["cell"]))
    ctx.close()
