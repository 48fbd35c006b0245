"""Developer probe (GPU box): the driver's short run - a fresh process, warm-up, then 20-pair regions - with the batched path's host timeline (knob batch_trace) on stderr."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import torch
torch.cuda.init()
from qn_amd import engine, synth
N = 100000
pairs = []
for j in range(8):
    s, t, _ = synth.make_pair(j, N)
    pairs.append((torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()))
torch.cuda.synchronize()
nctx, lanes = 3, 8
ctxs = [engine.Context(N + 1024) for _ in range(nctx)]
for cx in ctxs:
    cx.debug_set("batch_lanes", lanes); cx.debug_set("batch_share_source", 0)
    g = engine.NanoGICP(cx); g.setCorrespondenceRandomness(20); g.setMaximumIterations(20); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer("gn"); g.setForceIterations(20); g.bind()
def batch(n):
    d = [(pairs[j % 8][0].data_ptr(), N, pairs[j % 8][1].data_ptr(), N, 12, 1) for j in range(n)]
    return engine.icp_alignment_batch(ctxs, d)
batch(96); torch.cuda.synchronize()
for cx in ctxs:
    cx.debug_set("batch_trace", 1)
for rep in range(6):
    t0 = time.perf_counter(); batch(20); torch.cuda.synchronize(); w = 1e3 * (time.perf_counter() - t0)
    print("region %d: %.2f ms" % (rep, w), file=sys.stderr, flush=True)
