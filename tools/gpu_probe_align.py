"""Developer probe: align-only time (20 forced GN iterations) under different scheduling knobs."""
import os, sys, time, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np
from qn_amd import engine, synth
N = 100000
src, tgt, T = synth.make_pair(int(os.environ.get("PAIR", "0")), N)
ctx = engine.Context(N + 1024)
g = engine.NanoGICP(ctx)
g.setCorrespondenceRandomness(20); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer("gn"); g.setMaximumIterations(20); g.setForceIterations(20)
g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
def run(label, **kn):
    for k, v in kn.items(): ctx.debug_set(k, v)
    for _ in range(2): g.align()
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(10): g.align()
    ctx.synchronize(); dt = (time.perf_counter() - t0) / 10 * 1e3
    ctx.prof_reset(); ctx.prof_enable(True); g.align(); ctx.synchronize(); ctx.prof_enable(False)
    st = ctx.prof_stats()
    print("%-50s %.3f ms | search %.3f list %.3f acc %.3f solve %.3f" % (label, dt, st['nn_search'][0], st['nn_fallback'][0], st['accumulate'][0], st['solve'][0]))
import json
cfgs = json.loads(sys.argv[1]) if len(sys.argv) > 1 else [{}]
for c in cfgs: run(json.dumps(c), **c)
