"""Developer probe: the reference's own operating point: ~30k-point keyframe clouds, k = 15, LM, <= 32 iterations,
real stopping rule, clouds handed over as host buffers (qn_icp_alignment, as the nano_gicp shim does)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np
from qn_amd import engine, synth
for N in (10000, 30000, 100000):
    src, tgt, T = synth.make_pair(5, N)
    ctx = engine.Context(N + 1024)
    import json
    for kk, vv in json.loads(os.environ.get("QN_DEBUG_KNOBS", "{}")).items(): ctx.debug_set(kk, float(vv))
    for _ in range(3): r = engine.icp_alignment(ctx, src, tgt)
    t = time.perf_counter()
    for _ in range(20): r = engine.icp_alignment(ctx, src, tgt)
    dt = (time.perf_counter() - t) / 20 * 1e3
    ctx.prof_reset(); ctx.prof_enable(True); engine.icp_alignment(ctx, src, tgt); ctx.synchronize(); ctx.prof_enable(False)
    st = {k: round(v[0], 3) for k, v in ctx.prof_stats().items() if v[1]}
    print("N=%6d  icpAlignment (host buffers, k=15, LM): %.3f ms  iters=%d valid=%s score=%.4f  %s" % (N, dt, r["iterations"], r["valid"], r["score"], st))
    ctx.close()
