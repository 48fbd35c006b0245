"""Developer A/B (GPU box): quatro::align latency and stage times under knob sets, 30k and 100k; reports whether T and the correspondence count equal the first knob set's.
usage: python tools/gpu_quatro_ab.py '[{}, {"normals_fg": 8}, ...]' [sizes like 30000,100000]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np, torch
torch.cuda.init()
from qn_amd import engine, synth
sets = json.loads(sys.argv[1]) if len(sys.argv) > 1 else [{}]
sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "30000,100000").split(",")]
for npts in sizes:
    pairs = [synth.make_pair(pid, npts, mode="quatro")[:2] for pid in ((402, 403, 419) if npts <= 30000 else (400 + npts // 1000,))]
    ref = None
    for knobs in sets:
        ctx = engine.Context(npts + 1024)
        for k, v in knobs.items():
            ctx.debug_set(k, float(v))
        q = engine.Quatro(ctx)
        outs = []; lat = []
        for qs, qt in pairs:
            q.align(qs, qt); q.align(qs, qt)
            for _ in range(7):
                t0 = time.perf_counter(); T, valid = q.align(qs, qt); lat.append(1e3 * (time.perf_counter() - t0))
            r = q.align(qs, qt, debug=True)
            outs.append((r["T"].tobytes(), len(r["corres"]), r["valid"]))
        qs, qt = pairs[0]
        ctx.prof_reset(); ctx.prof_enable(True); q.align(qs, qt); ctx.synchronize(); ctx.prof_enable(False)
        st = ctx.prof_stats()
        stage = {k: round(st[k][0], 3) for k in ("grid_build", "fpfh_normals", "fpfh_spfh", "fpfh_fpfh", "feat_match", "match_tail") if st[k][1] > 0}
        if ref is None: ref = outs
        print("QUATRO %d knobs %s: align median %.3f ms (p10 %.3f) stages %s wall f/m/s %.3f/%.3f/%.3f same_as_first %s corres %s" % (
            npts, json.dumps(knobs), float(np.median(lat)), float(np.percentile(lat, 10)), stage, ctx.debug_get("quatro_wall_features_ms"), ctx.debug_get("quatro_wall_match_ms"),
            ctx.debug_get("quatro_wall_solve_ms"), outs == ref, [o[1] for o in outs]), flush=True)
        ctx.close()
