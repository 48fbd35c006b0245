"""Developer probe (GPU box): pairs/s of the batched coarse-to-fine path on the generator's OWN scene mix (scenes 400-407: Quatro's coarse pose is wrong on five of the eight,
Nano-GICP then runs its 32 LM iterations on misaligned clouds) and on the true-loop scenes, under knob sets.  argv: one JSON-ish dict per knob set."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np, torch
torch.cuda.init()
from qn_amd import engine, synth
NQ, nb = 30000, 64
sets = {"mixed": [synth.make_pair(400 + j, NQ, mode="quatro") for j in range(8)], "true_loops": [synth.make_pair(j, NQ, mode="quatro") for j in (402, 403, 404, 409, 410, 412, 419, 420)]}
dev = {k: [(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()) for s, t, _ in v] for k, v in sets.items()}
torch.cuda.synchronize()
for ks in (sys.argv[1:] or ["{}"]):
    knobs = eval(ks)
    ctxs = [engine.Context(NQ + 1024) for _ in range(4)]
    for cx in ctxs:
        cx.debug_set("batch_lanes", 8)
        for k_, v_ in knobs.items(): cx.debug_set(k_, v_)
        g = engine.NanoGICP(cx); g.setCorrespondenceRandomness(15); g.setMaximumIterations(32); g.setMaxCorrespondenceDistance(52.5); g.setTransformationEpsilon(0.01); g.bind()
        engine.Quatro(cx)
    for name, d in dev.items():
        descs = [(d[i % 8][0].data_ptr(), NQ, d[i % 8][1].data_ptr(), NQ, 12, 1) for i in range(nb)]
        engine.coarse_to_fine_align_batch(ctxs, descs[:32])
        w = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r = engine.coarse_to_fine_align_batch(ctxs, descs); torch.cuda.synchronize(); w.append(time.perf_counter() - t0)
        print("C2F %-28s %-10s %7.1f pairs/s  valid %d  iterations %s" % (ks, name, nb / float(np.median(w)), sum(x["valid"] for x in r), [x["iterations"] for x in r[:8]]), flush=True)
    for cx in ctxs: cx.close()
