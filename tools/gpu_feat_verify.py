"""Matrix-core feature matching against the VALU search, every query of both directions (debug knob feat_verify), several sampling steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np, torch
torch.cuda.init()
from qn_amd import engine, synth
for npts, pid in ((3000, 7), (30000, 331), (30000, 430), (100000, 500)):
    qs, qt, _ = synth.make_pair(pid, npts, mode="quatro")
    ctx = engine.Context(npts + 1024)
    fb = 0.0
    for step in (1, 4, 8, 64, 1000):
        ctx.debug_set("feat_sample", step); ctx.debug_set("feat_verify", 1)
        q = engine.Quatro(ctx); q.align(qs, qt)
        over = ctx.debug_get("feat_fallbacks") > fb; fb = ctx.debug_get("feat_fallbacks")
        # (when the survivor list overflows the screened result is incomplete BY CONSTRUCTION: it is discarded and the search repeated with the VALU kernel;
        #  the mismatch count of such a line describes the discarded result, tests/test_gpu_feat_mm.py::test_survivor_overflow_falls_back_to_the_valu_search)
        print("[overflow -> VALU search used]" if over else "[screened result used]     ",npts, pid, "step", step, "verified", ctx.debug_get("feat_verified"), "mismatches", ctx.debug_get("feat_mismatches"), "first", ctx.debug_get("feat_first_mismatch"),
              "survivors", ctx.debug_get("feat_survivors"), "fallbacks", ctx.debug_get("feat_fallbacks"))
    ctx.close()
