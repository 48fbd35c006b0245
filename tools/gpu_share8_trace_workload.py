import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np, torch, ctypes as C
torch.cuda.init()
from qn_amd import engine, synth
N = 100000
pairs = []
for j in range(8):
    s, t, _ = synth.make_pair(j, N); pairs.append((torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()))
torch.cuda.synchronize()
mg = engine.MultiGpu(1, N + 1024, in_flight=3)
p = engine.GicpParams(); engine.lib().qn_gicp_default_params(C.byref(p))
p.k_correspondences, p.max_iterations, p.max_corr_dist, p.optimizer, p.force_iterations = 20, 20, 52.5, 1, 20
mg.set_params(p); mg.debug_set("batch_lanes", 8); mg.debug_set("batch_share_source", 0)
d = [(s.data_ptr(), N, t.data_ptr(), N, 12, 1) for s, t in pairs]
for _ in range(4): mg.align_best(d)
torch.cuda.synchronize(); t0 = time.perf_counter(); mg.align_best(d); print("call ms", 1e3 * (time.perf_counter() - t0))
