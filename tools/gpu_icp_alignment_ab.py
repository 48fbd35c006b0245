"""Developer A/B (GPU box): LoopClosure::icpAlignment in one call (qn_icp_alignment, host buffers and device buffers) at the reference's operating point under knob sets.
usage: python tools/gpu_icp_alignment_ab.py '[{}, {"tgt_early": 0}]' [sizes]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np, torch
torch.cuda.init()
from qn_amd import engine, synth
sets = json.loads(sys.argv[1]) if len(sys.argv) > 1 else [{}]
sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "30000,100000").split(",")]
for npts in sizes:
    pairs = [synth.make_pair(700 + j, npts)[:2] for j in range(4)]
    ref = None
    for rep in range(2):
        for knobs in sets:
            ctx = engine.Context(npts + 1024)
            for k, v in knobs.items():
                ctx.debug_set(k, float(v))
            for s, t in pairs: engine.icp_alignment(ctx, s, t)
            lat = []; outs = []
            for j in range(40):
                s, t = pairs[j % 4]
                t0 = time.perf_counter(); r = engine.icp_alignment(ctx, s, t); lat.append(1e3 * (time.perf_counter() - t0))
                if j < 4: outs.append((r["iterations"], r["score"], r["T"].tobytes()))
            if ref is None: ref = outs
            print("ICP %d knobs %s: host buffers median %.4f ms p10 %.4f  same_records %s" % (npts, json.dumps(knobs), float(np.median(lat)), float(np.percentile(lat, 10)), outs == ref), flush=True)
            ctx.close()
