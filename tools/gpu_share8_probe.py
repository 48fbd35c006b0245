"""Developer probe (GPU box): what ONE rank does at N = 8 (BASELINE configs[3]: 64 pairs over 8 GPUs = 8 pairs per rank), measured on one GPU with the bench's own 64 pairs
(8 scenes x 8 re-pose variants of rising difficulty): wall of the 64-pair qn_multi_align_best call and of the eight 8-pair calls (blocks of 8 consecutive pairs), for the
minimum share of the call's last round (knob batch_min_share) and the contexts per GPU.  Sum of the eight block walls / 64-pair wall = what the short calls cost."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np, torch, ctypes as C
torch.cuda.init()
from qn_amd import engine, synth
N = 100000
scenes = []
for j in range(8):
    s, t, _ = synth.make_pair(j, N); scenes.append((torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()))
pairs = []
for i in range(64):
    s, t = scenes[i % 8]; v = i // 8
    if v:
        a = 0.01 * v; ca, sa = float(np.cos(a)), float(np.sin(a))
        R = torch.tensor([[ca, -sa, 0.0], [sa, ca, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32, device=t.device)
        t = (t @ R.T + torch.tensor([0.05 * v, -0.03 * v, 0.0], dtype=torch.float32, device=t.device)).contiguous()
    pairs.append((s, t))
torch.cuda.synchronize()
d = [(s.data_ptr(), N, t.data_ptr(), N, 12, 1) for s, t in pairs]
for inflight, share in ((3, 1), (3, 4), (3, 8), (2, 4), (4, 4)):
    mg = engine.MultiGpu(1, N + 1024, in_flight=inflight)
    p = engine.GicpParams(); engine.lib().qn_gicp_default_params(C.byref(p))
    p.k_correspondences, p.max_iterations, p.max_corr_dist, p.optimizer, p.force_iterations = 20, 20, 52.5, 1, 20
    mg.set_params(p); mg.debug_set("batch_lanes", 8); mg.debug_set("batch_share_source", 0); mg.debug_set("batch_min_share", share)
    for _ in range(2): mg.align_best(d[:16])
    w64 = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); mg.align_best(d); w64.append(1e3 * (time.perf_counter() - t0))
    blocks = []
    for b in range(8):
        w = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); mg.align_best(d[8 * b:8 * b + 8]); w.append(1e3 * (time.perf_counter() - t0))
        blocks.append(float(np.median(w)))
    print("SHARE8 in_flight %d min_share %d: 64-pair call %.2f ms | 8-pair blocks %s ms, mean %.3f | projected at 8 GPUs %.2fx" % (
        inflight, share, float(np.median(w64)), [round(x, 2) for x in blocks], float(np.mean(blocks)), float(np.median(w64)) / (float(np.mean(blocks)) + 0.03)), flush=True)
    mg.close()
