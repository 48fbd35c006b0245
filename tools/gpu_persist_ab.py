"""developer A/B: align() latency with and without the persistent kernel (single stream), BASELINE configs[1] workload"""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np
import torch; torch.cuda.init()
from qn_amd import engine, synth
N = 100000
pairs = [synth.make_pair(j, N) for j in range(4)]
dev = [(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()) for s, t, _ in pairs]
torch.cuda.synchronize()
for knobs in ({"persist": 0}, {"persist": 1}, {"persist": 0}, {"persist": 1}):
    ctx = engine.Context(N + 1024)
    for k, v in knobs.items(): ctx.debug_set(k, v)
    for k, v in json.loads(os.environ.get("QN_DEBUG_KNOBS", "{}")).items(): ctx.debug_set(k, float(v))
    g = engine.NanoGICP(ctx); g.setCorrespondenceRandomness(20); g.setMaximumIterations(20); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer("gn"); g.setForceIterations(20)
    def reg(j):
        s, t = dev[j % 4]
        g.setInputSourceDevice(s.data_ptr(), N, 12); g.calculateSourceCovariances(); g.setInputTargetDevice(t.data_ptr(), N, 12); g.calculateTargetCovariances()
        return g.align()
    reg(0); reg(1)
    lat = []
    for j in range(24):
        t0 = time.perf_counter(); reg(j); lat.append(1e3 * (time.perf_counter() - t0))
    al = []
    for j in range(40):
        t0 = time.perf_counter(); r = g.align(); al.append(1e3 * (time.perf_counter() - t0))
    print("knobs %-16s registration median %.4f p10 %.4f ms   align median %.4f p10 %.4f ms   persist launches %d  T[0,3] %.12f" % (
        json.dumps(knobs), np.median(lat), np.percentile(lat, 10), np.median(al), np.percentile(al, 10), ctx.debug_get("persist_launches"), r.T64[3]))
    ctx.close()
