"""Developer probe: robustness of the search tails to isolated points (real scans have them, the synthetic scene does not):
a fraction of the cloud is replaced by points scattered uniformly in the bounding volume."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np
from qn_amd import engine, synth
N = int(os.environ.get("N", "100000"))
src0, tgt0, T = synth.make_pair(0, N)
ctx = engine.Context(N + 1024)
rng = np.random.default_rng(3)
for frac in (0.0, 0.01, 0.05, 0.2):
    def spoil(c):
        c = c.copy(); m = int(frac * len(c))
        if m:
            lo, hi = c.min(0), c.max(0); hi[2] = lo[2] + 25.0
            c[rng.choice(len(c), m, replace=False)] = rng.uniform(lo, hi, size=(m, 3)).astype(np.float32)
        return c
    src, tgt = spoil(src0), spoil(tgt0)
    for _ in range(2): r = engine.icp_alignment(ctx, src, tgt)
    t = time.perf_counter()
    for _ in range(5): r = engine.icp_alignment(ctx, src, tgt)
    dt = (time.perf_counter() - t) / 5 * 1e3
    ctx.prof_reset(); ctx.prof_enable(True); engine.icp_alignment(ctx, src, tgt); ctx.synchronize(); ctx.prof_enable(False)
    st = {k: round(v[0], 3) for k, v in ctx.prof_stats().items() if v[1]}
    print("outliers %4.0f%%: icpAlignment %.3f ms iters=%d valid=%s  %s" % (100 * frac, dt, r["iterations"], r["valid"], st))
    g = engine.NanoGICP(ctx); g.setCorrespondenceRandomness(20); g.setMaximumIterations(20); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer("gn"); g.setForceIterations(20)
    g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
    for _ in range(2): g.align()
    t = time.perf_counter()
    for _ in range(5): g.align()
    print("               GN x20 align only: %.3f ms" % ((time.perf_counter() - t) / 5 * 1e3))
