"""Developer probe (GPU box): what the far-query refreshes (k_far / in-tick misses: wave_ball_collect) of one forced-GN align cost - calls, candidates streamed and
segments enumerated per call (debug counters 14 / 15 / 9), next to the one-per-wave searches of the unseeded passes (13 / 10 / 11 / 12).
usage: python tools/gpu_probe_far.py [pair_id=0] [shift=24]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np
from qn_amd import engine, synth
N = 100000
pid = int(sys.argv[1]) if len(sys.argv) > 1 else 0
shift = float(sys.argv[2]) if len(sys.argv) > 2 else 24.0
src, tgt, T = synth.make_pair(pid, N, shift=shift)
ctx = engine.Context(N + 1024)
g = engine.NanoGICP(ctx)
g.setCorrespondenceRandomness(20); g.setMaximumIterations(20); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer("gn"); g.setForceIterations(20)
g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
out = (C.c_uint32 * 16)()
ctx.debug_set("dbg_counters", 1); r = g.align(); ctx.synchronize(); ctx.check(ctx._l.qn_debug_get_counters(ctx.h, out)); ctx.debug_set("dbg_counters", 0)
c = list(out)
print("pair %d shift %.0f: ball collections %d, %.0f candidates and %.0f segments per call; one-per-wave searches %d entries, %.2f rounds, %.0f candidates, %.0f segments per entry"
      % (pid, shift, c[14], c[15] / max(c[14], 1), c[9] / max(c[14], 1), c[13], c[10] / max(c[13], 1), c[11] / max(c[13], 1), c[12] / max(c[13], 1)))
ctx.close()
