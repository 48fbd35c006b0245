"""Developer sweep (GPU box): registrations/s of the headline workload (BASELINE configs[1]) for (contexts = streams) x (lanes = pairs per launch).
usage: python tools/gpu_batch_sweep.py [steps] [configs like 4x1,2x8,...] [shift]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import torch
torch.cuda.init()
from qn_amd import engine, synth
engine.DEBUG_KNOBS_FROM_ENV = True
N = 100000
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 160
cfgs = [tuple(int(x) for x in c.split("x")) for c in (sys.argv[2] if len(sys.argv) > 2 else "4x1,1x8,2x4,2x8,1x16,4x4").split(",")]
shift = float(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] != "-" else None
pairs = []
for j in range(8):
    s, t, _ = synth.make_pair(j if shift is None else 9000 + j, N, shift=shift)
    pairs.append((torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()))
torch.cuda.synchronize()
out = {}
for nctx, lanes in cfgs:
    ctxs = [engine.Context(N + 1024) for _ in range(nctx)]
    for cx in ctxs:
        cx.debug_set("batch_lanes", lanes); cx.debug_set("batch_share_source", 0)      # every registration rebuilds its source (8 distinct pairs cycling over 8 lanes would otherwise meet their own source again)
        g = engine.NanoGICP(cx); g.setCorrespondenceRandomness(20); g.setMaximumIterations(20); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer("gn"); g.setForceIterations(20); g.bind()
    def batch(n):
        d = [(pairs[j % 8][0].data_ptr(), N, pairs[j % 8][1].data_ptr(), N, 12, 1) for j in range(n)]
        return engine.icp_alignment_batch(ctxs, d, score_thr=1.5)
    batch(2 * nctx * lanes); torch.cuda.synchronize()
    best = 0.0; runs = []
    for rep in range(3):
        t0 = time.perf_counter(); r, v, st = batch(steps); torch.cuda.synchronize(); w = time.perf_counter() - t0
        assert all(x == 0 for x in st), st
        runs.append(round(steps / w, 1))
    lp = [cx.debug_get("batch_launches") / max(1.0, cx.debug_get("batch_pairs")) for cx in ctxs]
    out["%dx%d" % (nctx, lanes)] = {"reg_per_s": runs, "launches_per_registration": round(lp[0], 2), "fitness0": r[0].fitness}
    print("%dx%d" % (nctx, lanes), out["%dx%d" % (nctx, lanes)], flush=True)
    for cx in ctxs:
        cx.close()
print(json.dumps(out))
