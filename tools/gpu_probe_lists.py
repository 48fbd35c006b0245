"""Developer probe (GPU box): how many queries the unseeded 1-NN passes hand to their list passes, per tick (debug counters 5 = leftovers served
16 per wave, 7 = far leftovers served one per wave; both accumulate over the launches of an align, so ticks are separated by forcing 1, 2, 3 iterations).
usage: python tools/gpu_probe_lists.py [pair_id=0] [shift]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np
from qn_amd import engine, synth
N = 100000
pid = int(sys.argv[1]) if len(sys.argv) > 1 else 0
shift = float(sys.argv[2]) if len(sys.argv) > 2 else None
src, tgt, T = synth.make_pair(pid, N, shift=shift)
ctx = engine.Context(N + 1024)
g = engine.NanoGICP(ctx)
g.setCorrespondenceRandomness(20); g.setMaximumIterations(20); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer("gn")
g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
def counters():
    out = (C.c_uint32 * 16)(); ctx.check(ctx._l.qn_debug_get_counters(ctx.h, out)); return list(out)
prev = (0, 0)
for it in (1, 2, 3):
    g.setForceIterations(it)
    ctx.debug_set("dbg_counters", 1); r = g.align(); ctx.synchronize(); c = counters(); ctx.debug_set("dbg_counters", 0)
    # the closing pass (fitness sweep) of a forced run is tracked, not a list pass of k_nn_search<0>; counters 5 / 7 also see the fitness search's lists (MODE 1) when it runs unseeded
    print("forced %d iteration(s): list entries so far  16-per-wave %7d  one-per-wave %7d   (this tick: %7d / %7d of %d queries)" % (it, c[5], c[7], c[5] - prev[0], c[7] - prev[1], N))
    if c[13]:
        print("      one-per-wave searches so far: %d entries, %.2f rounds, %.0f candidates, %.0f enumerated segments per entry" % (c[13], c[10] / c[13], c[11] / c[13], c[12] / c[13]))
    prev = (c[5], c[7])
ctx.close()
