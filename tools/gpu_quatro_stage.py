"""Quatro coarse stage: per-stage times, matrix-core feature matching on / off (same answer), survivors and fallbacks."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np, torch
torch.cuda.init()
from qn_amd import engine, synth
for npts in (30000, 100000):
    qs, qt, _ = synth.make_pair(400 + npts // 1000, npts, mode="quatro")
    ctx = engine.Context(npts + 1024)
    for kk, vv in json.loads(os.environ.get("QN_DEBUG_KNOBS", "{}")).items():
        ctx.debug_set(kk, float(vv))
    res = {}
    for mm in (1, 0):
        ctx.debug_set("feat_mfma", mm)
        q = engine.Quatro(ctx)
        q.align(qs, qt)
        lat = []
        for _ in range(5):
            t0 = time.perf_counter(); T, valid = q.align(qs, qt); lat.append(1e3 * (time.perf_counter() - t0))
        ctx.prof_reset(); ctx.prof_enable(True); q.align(qs, qt); ctx.synchronize(); ctx.prof_enable(False)
        st = ctx.prof_stats()
        stage = {k: round(st[k][0], 3) for k in ("grid_build", "fpfh_normals", "fpfh_spfh", "fpfh_fpfh", "feat_match", "match_tail") if st[k][1] > 0}
        res[mm] = np.array(T)
        print(npts, "mfma" if mm else "valu", "align ms median %.3f" % np.median(lat), stage, "survivors", ctx.debug_get("feat_survivors"), "fallbacks", ctx.debug_get("feat_fallbacks"), "valid", valid,
              "wall ms: features %.3f match %.3f solve %.3f" % (ctx.debug_get("quatro_wall_features_ms"), ctx.debug_get("quatro_wall_match_ms"), ctx.debug_get("quatro_wall_solve_ms")))
    print("  same T:", np.array_equal(res[0], res[1]))
    ctx.close()
