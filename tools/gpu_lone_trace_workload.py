import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np, torch
torch.cuda.init()
from qn_amd import engine, synth
N = 100000
pairs = []
for j in range(3):
    s, t, _ = synth.make_pair(j, N); pairs.append((torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()))
torch.cuda.synchronize()
cx = engine.Context(N + 1024)
g = engine.NanoGICP(cx); g.setCorrespondenceRandomness(20); g.setMaximumIterations(20); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer("gn"); g.setForceIterations(20); g.bind()
def reg(j):
    s, t = pairs[j % 3]
    g.setInputSourceDevice(s.data_ptr(), N, 12); g.calculateSourceCovariances(); g.setInputTargetDevice(t.data_ptr(), N, 12); g.calculateTargetCovariances(); return g.align()
for j in range(6): reg(j)
lat = []
for j in range(12):
    t0 = time.perf_counter(); reg(j); lat.append(1e3 * (time.perf_counter() - t0))
print("lone registration ms", np.median(lat))
