"""Developer probe: run 3 single-stream registrations (the last one is the one to look at in a rocprofv3 kernel trace)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np, torch
from qn_amd import engine, synth
N = 100000
src, tgt, T = synth.make_pair(0, N)
s = torch.from_numpy(src).cuda(); t = torch.from_numpy(tgt).cuda(); torch.cuda.synchronize()
ctx = engine.Context(N + 1024)
g = engine.NanoGICP(ctx); g.setCorrespondenceRandomness(20); g.setMaximumIterations(20); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer("gn"); g.setForceIterations(20)
for _ in range(3):
    engine.icp_alignment_batch([ctx], [(s.data_ptr(), N, t.data_ptr(), N, 12, 1)])
