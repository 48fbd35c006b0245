"""Developer probe (GPU box): wall-clock time of the one-per-wave entries of the unseeded list passes (knob list_probe: per wave, slowest entry / entries / busy time).
Ticks are separated by forcing 1, 2, 3 iterations (the probe holds the LAST unseeded list pass of an align).  usage: python tools/gpu_probe_list_times.py [pair_id=0] [shift]"""
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
ctx.debug_set("single_from_tick", 0)          # fixed hand-over after three unseeded iterations: no conditional tick
ctx.debug_set("list_probe", 1)
g = engine.NanoGICP(ctx)
g.setCorrespondenceRandomness(20); g.setMaximumIterations(20); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer("gn")
g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
buf = np.zeros(4 * 16384, dtype=np.uint64)
for it in (1, 2, 3):
    g.setForceIterations(it)
    for _ in range(3): r = g.align()
    ctx.check(ctx._l.qn_debug_get_list_probe(ctx.h, buf.ctypes.data_as(C.c_void_p)))
    mx = (buf[0::4] >> np.uint64(32)).astype(np.float64) / 100.0; n = (buf[0::4] & np.uint64(0xffffffff)).astype(np.int64); busy = buf[1::4].astype(np.float64) / 100.0
    act = n > 0
    per = busy[act] / n[act]
    print("tick %d: %d entries on %d waves (max %d per wave); per entry: mean %.1f us  median %.1f  p90 %.1f  p99 %.1f  max %.1f;  busiest wave %.1f us, mean busy %.1f us"
          % (it - 1, n.sum(), act.sum(), n.max(), per.mean(), np.median(per), np.percentile(per, 90), np.percentile(per, 99), mx.max(), busy.max(), busy[act].mean()))
    hist = np.histogram(mx[act], bins=[0, 5, 10, 20, 40, 80, 160, 1e9])[0]
    print("        slowest entry per wave, histogram (us) <5 <10 <20 <40 <80 <160 more:", hist.tolist())
    a = buf[2::4]; b = buf[3::4]
    rounds = (a >> np.uint64(48)).astype(np.int64); segs = ((a >> np.uint64(24)) & np.uint64(0xffffff)).astype(np.int64); cand = (a & np.uint64(0xffffff)).astype(np.int64)
    rf = (b >> np.uint64(32)).astype(np.uint32).view(np.float32); dn = (b & np.uint64(0xffffffff)).astype(np.uint32).view(np.float32)
    for lo, hi in ((0, 10), (10, 20), (20, 40), (40, 80), (80, 1e9)):
        m = act & (mx >= lo) & (mx < hi)
        if m.any():
            print("        slowest entries of %5.0f-%-5.0f us (%5d): rounds %.1f  enumerated segments %.0f  candidates %.0f  first radius %.2f m  neighbour at %.2f m"
                  % (lo, min(hi, 999), m.sum(), rounds[m].mean(), segs[m].mean(), cand[m].mean(), rf[m].mean(), dn[m].mean()))
ctx.close()
