"""Developer sweep (GPU box), ONE process, pairs generated once: throughput of the headline workload (BASELINE configs[1]) for (contexts x lanes) under several
knob sets, then the lone-registration latency (one registration at a time on one stream, and align() alone) under several knob sets.
usage: python tools/gpu_knob_sweep.py '<json: {"cfgs": ["3x8", ...], "knobs": [{}, {"tick_rpb": 3}, ...], "lone_knobs": [{}, {"nn_lane": 1}], "steps": 200, "shift": null, "params": "rop" (optional: LM at the reference's operating point instead of 20 forced GN)}>'"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np
import torch
torch.cuda.init()
from qn_amd import engine, synth
spec = json.loads(sys.argv[1]) if len(sys.argv) > 1 else {}
N = 100000
steps = int(spec.get("steps", 200)); shift = spec.get("shift")
pairs = []
for j in range(8):
    s, t, _ = synth.make_pair(j if shift is None else 9000 + j, N, shift=shift)
    pairs.append((torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()))
torch.cuda.synchronize()


def bind(cx):
    g = engine.NanoGICP(cx)
    if spec.get("params") == "rop":      # the reference's operating point: k = 15, LM, <= 32 iterations, the real stopping rule
        g.setCorrespondenceRandomness(15); g.setMaximumIterations(32); g.setMaxCorrespondenceDistance(52.5); g.setTransformationEpsilon(0.01); g.bind()
    else:
        g.setCorrespondenceRandomness(20); g.setMaximumIterations(20); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer("gn"); g.setForceIterations(20); g.bind()
    return g


ref_fit = None
for cfg in spec.get("cfgs", ["3x8"]):
    nctx, lanes = (int(x) for x in cfg.split("x"))
    for knobs in spec.get("knobs", [{}]):
        ctxs = [engine.Context(N + 1024) for _ in range(nctx)]
        for cx in ctxs:
            cx.debug_set("batch_lanes", lanes); cx.debug_set("batch_share_source", 0)
            for k, v in knobs.items():
                cx.debug_set(k, float(v))
            bind(cx)
        def batch(n):
            d = [(pairs[j % 8][0].data_ptr(), N, pairs[j % 8][1].data_ptr(), N, 12, 1) for j in range(n)]
            return engine.icp_alignment_batch(ctxs, d, score_thr=1.5)
        batch(4 * nctx * lanes); torch.cuda.synchronize()
        tw = time.perf_counter()
        while time.perf_counter() - tw < 0.25:
            batch(2 * nctx * lanes)
        torch.cuda.synchronize()
        runs = []
        for rep in range(5):
            t0 = time.perf_counter(); r, v, st = batch(steps); torch.cuda.synchronize(); w = time.perf_counter() - t0
            assert all(x == 0 for x in st), st
            runs.append(round(steps / w, 1))
        fit = [r[i].fitness for i in range(8)]
        if ref_fit is None: ref_fit = fit
        print("THROUGHPUT %s knobs %s: median %.1f  runs %s  records_equal_first %s" % (cfg, json.dumps(knobs), float(np.median(runs)), runs, fit == ref_fit), flush=True)
        for cx in ctxs:
            cx.close()

for knobs in spec.get("lone_knobs", []):
    cx = engine.Context(N + 1024)
    for k, v in knobs.items():
        cx.debug_set(k, float(v))
    g = bind(cx)
    def register(j):
        s, t = pairs[j % 8]
        g.setInputSourceDevice(s.data_ptr(), N, 12); g.calculateSourceCovariances()
        g.setInputTargetDevice(t.data_ptr(), N, 12); g.calculateTargetCovariances()
        return g.align()
    for j in range(4): register(j)
    lat = []
    for j in range(40):
        t0 = time.perf_counter(); r = register(j); lat.append(1e3 * (time.perf_counter() - t0))
    register(0); cx.synchronize(); al = []
    for _ in range(40):
        t0 = time.perf_counter(); r0 = g.align(); al.append(1e3 * (time.perf_counter() - t0))
    print("LONE knobs %s: registration median %.4f p10 %.4f p90 %.4f ms | align median %.4f ms | fitness0 %.17g" % (
        json.dumps(knobs), float(np.median(lat)), float(np.percentile(lat, 10)), float(np.percentile(lat, 90)), float(np.median(al)), r0.fitness), flush=True)
    cx.close()
