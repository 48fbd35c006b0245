"""Developer sweep (GPU box): pairs/s of qn_coarse_to_fine_align_batch on 64 true-loop 30k pairs for (contexts x lanes); plus the one-pair path's latency.
usage: python tools/gpu_c2f_sweep.py [cfgs like 4x8,6x4,...]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np
import torch
torch.cuda.init()
from qn_amd import engine, synth
engine.DEBUG_KNOBS_FROM_ENV = True
NQ = 30000
cfgs = [tuple(int(x) for x in c.split("x")) for c in (sys.argv[1] if len(sys.argv) > 1 else "4x8,3x8,6x4,8x4,6x8,4x4,2x8").split(",")]
scenes = [synth.make_pair(j, NQ, mode="quatro") for j in (402, 403, 404, 409, 410, 412, 419, 420)]
dev = [(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()) for s, t, _ in scenes]
qd = []
for i in range(64):
    s_, t_ = dev[i % 8]; v = i // 8
    if v:
        a = 0.004 * v; ca, sa = float(np.cos(a)), float(np.sin(a))
        R = torch.tensor([[ca, -sa, 0.0], [sa, ca, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32, device=t_.device)
        t_ = (t_ @ R.T + torch.tensor([0.02 * v, -0.01 * v, 0.0], dtype=torch.float32, device=t_.device)).contiguous()
    qd.append((s_, t_))
torch.cuda.synchronize()
descs = [(s_.data_ptr(), NQ, t_.data_ptr(), NQ, 12, 1) for s_, t_ in qd]


def mk(lanes):
    cx = engine.Context(NQ + 1024); cx.debug_set("batch_lanes", lanes)
    g = engine.NanoGICP(cx); g.setCorrespondenceRandomness(15); g.setMaximumIterations(32); g.setMaxCorrespondenceDistance(52.5); g.setTransformationEpsilon(0.01); g.bind()
    engine.Quatro(cx)
    return cx


for nctx, lanes in cfgs:
    ctxs = [mk(lanes) for _ in range(nctx)]
    engine.coarse_to_fine_align_batch(ctxs, descs[:2 * nctx * lanes]); torch.cuda.synchronize()
    runs = []
    for rep in range(4):
        t0 = time.perf_counter(); r = engine.coarse_to_fine_align_batch(ctxs, descs); torch.cuda.synchronize(); runs.append(round(64 / (time.perf_counter() - t0), 1))
    print("C2F %dx%d: pairs/s %s  valid %d/64" % (nctx, lanes, runs, sum(x["valid"] for x in r)), flush=True)
    for cx in ctxs:
        cx.close()
cx = mk(1)
for (s_, t_) in qd[:8]:
    engine.coarse_to_fine_alignment_device(cx, s_.data_ptr(), NQ, t_.data_ptr(), NQ, 12)
lat = []
for (s_, t_) in qd[:24]:
    t0 = time.perf_counter(); r = engine.coarse_to_fine_alignment_device(cx, s_.data_ptr(), NQ, t_.data_ptr(), NQ, 12); lat.append(1e3 * (time.perf_counter() - t0))
print("C2F one pair at a time: median %.3f ms p10 %.3f p90 %.3f" % (float(np.median(lat)), float(np.percentile(lat, 10)), float(np.percentile(lat, 90))))
q = engine.Quatro(cx); lat = []
for (s_, t_) in qd[:24]:
    t0 = time.perf_counter(); q.align_device(s_.data_ptr(), NQ, t_.data_ptr(), NQ, 12); lat.append(1e3 * (time.perf_counter() - t0))
print("quatro::align one pair at a time: median %.3f ms; wall split features %.3f match %.3f solve %.3f" % (float(np.median(lat)), cx.debug_get("quatro_wall_features_ms"), cx.debug_get("quatro_wall_match_ms"), cx.debug_get("quatro_wall_solve_ms")))
