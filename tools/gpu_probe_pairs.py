"""developer probe: per-pair single-stream latency of the BASELINE configs[1] workload (which pairs are slow, and which path they take)"""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np
import torch; torch.cuda.init()
from qn_amd import engine, synth
N = 100000
ids = [int(a) for a in sys.argv[1:]] or list(range(8))
shift = float(os.environ["SHIFT"]) if "SHIFT" in os.environ else None
ctx = engine.Context(N + 1024)
for k, v in json.loads(os.environ.get("QN_DEBUG_KNOBS", "{}")).items(): ctx.debug_set(k, float(v))
g = engine.NanoGICP(ctx); g.setCorrespondenceRandomness(20); g.setMaximumIterations(20); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer("gn"); g.setForceIterations(20)
for pid in ids:
    src, tgt, T = synth.make_pair(pid, N, shift=shift)
    ds, dt = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda(); torch.cuda.synchronize()
    def reg():
        g.setInputSourceDevice(ds.data_ptr(), N, 12); g.calculateSourceCovariances(); g.setInputTargetDevice(dt.data_ptr(), N, 12); g.calculateTargetCovariances(); return g.align()
    reg(); reg()
    rg = []
    for _ in range(12):
        t0 = time.perf_counter(); reg(); rg.append(1e3 * (time.perf_counter() - t0))
    p0 = ctx.debug_get("persist_launches"); al = []
    for _ in range(20):
        t0 = time.perf_counter(); r = g.align(); al.append(1e3 * (time.perf_counter() - t0))
    used = ctx.debug_get("persist_launches") - p0
    yaw = np.degrees(np.arctan2(T[1, 0], T[0, 0]))
    print("pair %4d  T_gt yaw %6.2f deg t (%.2f %.2f)  registration median %.4f min %.4f  align median %.4f  persistent aligns %d/20  score %.4f" % (
        pid, yaw, T[0, 3], T[1, 3], np.median(rg), np.min(rg), np.median(al), used, r.fitness))
    tr = g.trace()
    print("      steps (max|dt| m, max|dR| -> ~displacement at 85 m):", " ".join("%d:%.3g/%.2g=%.3g" % (i, tr[i, 4], tr[i, 3], tr[i, 4] + 85 * tr[i, 3]) for i in range(min(8, len(tr)))))
    print("      extra unseeded ticks chosen:", int(ctx.debug_get("extra_unseeded")))
    if os.environ.get("FAMILIES"):
        ctx.prof_reset(); ctx.prof_enable(True); reg(); reg(); ctx.synchronize(); ctx.prof_enable(False)
        print("      families (ms per registration, chain):", {k: round(v[0] / 2, 4) for k, v in ctx.prof_stats().items() if v[1] > 0}, " launches:", {k: v[1] // 2 for k, v in ctx.prof_stats().items() if v[1] > 0 and k in ("nn_search", "gn_tick_fused", "far_refresh")})
