"""Workload for a rocprofv3 kernel trace of ONE 8-pair qn_multi_align_best call (a rank's whole work at N = 8): argv[1] = re-pose variant of the bench's mixed pairs (0 = as generated)."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np, torch, ctypes as C
torch.cuda.init()
from qn_amd import engine, synth
N = 100000; ov = len(sys.argv) > 1 and sys.argv[1] == "ov"; v = 0 if ov else (int(sys.argv[1]) if len(sys.argv) > 1 else 0)      # "ov": the 80 %-overlap pairs
pairs = []
for j in range(8):
    s, t, _ = synth.make_pair(j, N, shift=24.0) if ov else synth.make_pair(j, N); s, t = torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()
    if v:
        a = 0.01 * v; ca, sa = float(np.cos(a)), float(np.sin(a))
        R = torch.tensor([[ca, -sa, 0.0], [sa, ca, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32, device=t.device)
        t = (t @ R.T + torch.tensor([0.05 * v, -0.03 * v, 0.0], dtype=torch.float32, device=t.device)).contiguous()
    pairs.append((s, t))
torch.cuda.synchronize()
mg = engine.MultiGpu(1, N + 1024, in_flight=3)
p = engine.GicpParams(); engine.lib().qn_gicp_default_params(C.byref(p))
p.k_correspondences, p.max_iterations, p.max_corr_dist, p.optimizer, p.force_iterations = 20, 20, 52.5, 1, 20
mg.set_params(p); mg.debug_set("batch_lanes", 8); mg.debug_set("batch_share_source", 0)
for k_, v_ in (eval(sys.argv[2]) if len(sys.argv) > 2 else {}).items(): mg.debug_set(k_, v_)
d = [(s.data_ptr(), N, t.data_ptr(), N, 12, 1) for s, t in pairs]
for _ in range(4): mg.align_best(d)
torch.cuda.synchronize(); time.sleep(0.002)
t0 = time.perf_counter(); mg.align_best(d); print("call ms", 1e3 * (time.perf_counter() - t0))
