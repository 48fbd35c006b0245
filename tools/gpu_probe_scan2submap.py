import os, sys, time
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np
from qn_amd import engine, synth
src, tgt, T = synth.make_pair(9, 100000)
rng = np.random.default_rng(1)
ctx = engine.Context(101024)
for ns in (5000, 20000, 100000):
    s = src[rng.choice(len(src), ns, replace=False)] if ns < len(src) else src
    for _ in range(2): r = engine.icp_alignment(ctx, s, tgt)
    t = time.perf_counter()
    for _ in range(10): r = engine.icp_alignment(ctx, s, tgt)
    dt = (time.perf_counter() - t) / 10 * 1e3
    ctx.prof_reset(); ctx.prof_enable(True); engine.icp_alignment(ctx, s, tgt); ctx.synchronize(); ctx.prof_enable(False)
    st = {k: round(v[0], 3) for k, v in ctx.prof_stats().items() if v[1]}
    print("scan %6d -> submap %6d: icpAlignment %.3f ms iters=%d valid=%s %s" % (ns, len(tgt), dt, r["iterations"], r["valid"], st))
