"""Developer probe (GPU box): Quatro coarse stage timings (BASELINE config 3)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np
from qn_amd import engine, synth
for N in (30000, 100000):
    src, tgt, T = synth.make_pair(400, N, mode="quatro")
    ctx = engine.Context(N + 1024)
    q = engine.Quatro(ctx)
    r = q.align(src, tgt, debug=True)
    print("N=%d valid=%s mutual=%d corres=%d clique=%d rot_it=%d coarse err %s" % (N, r["valid"], len(r["mutual"]), len(r["corres"]), len(r["clique"]), r["rot_iterations"], synth.pose_error(r["T"], T)))
    t = time.perf_counter()
    for _ in range(3): q.align(src, tgt)
    print("  quatro.align: %.2f ms" % ((time.perf_counter() - t) / 3 * 1e3))
    ctx.prof_reset(); ctx.prof_enable(True); q.align(src, tgt); ctx.synchronize(); ctx.prof_enable(False)
    print("  ", {k: round(v[0], 3) for k, v in ctx.prof_stats().items() if v[1]})
    t = time.perf_counter(); c = engine.coarse_to_fine_alignment(ctx, src, tgt); dt = time.perf_counter() - t
    print("  coarse_to_fine: %.2f ms valid=%s score=%.4f err vs GT %s" % (dt * 1e3, c["valid"], c["score"], synth.pose_error(c["T"], T)))
    if N == 30000 and "--cpu" in sys.argv:
        from oracle import oracle as orc
        t = time.perf_counter(); o = orc.quatro_align(src, tgt); print("  CPU oracle quatro_align: %.1f ms (threads %d)" % ((time.perf_counter() - t) * 1e3, orc.num_threads()))
    ctx.close()
