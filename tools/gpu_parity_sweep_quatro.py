"""Stress tool (GPU box): Quatro coarse stage and coarse-to-fine, GPU vs oracle, over seeded pairs (yaw up to 180 deg)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np
from qn_amd import engine, synth
from oracle import oracle as orc          # checker only
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ctx = engine.Context(40000); bad = 0; t0 = time.time()
for case in range(ncases):
    n = [4000, 7000, 12000, 20000][case % 4]
    src, tgt, T = synth.make_pair(900 + case, n, extent=45.0 if n <= 12000 else 70.0, mode="quatro")
    q = engine.Quatro(ctx)
    Tq, valid = q.align(src, tgt)
    o = orc.quatro_align(src, tgt)
    c = engine.coarse_to_fine_alignment(ctx, src, tgt)
    oc = orc.coarse_to_fine_alignment(src, tgt)
    dq = synth.pose_error(np.asarray(Tq), o["T"]); dc = synth.pose_error(c["T"], oc["T"])
    ok = (bool(valid) == bool(o["valid"]) and dq[0] <= 1e-4 and dq[1] <= 1e-4 and c["valid"] == oc["valid"]
          and (not oc["valid"] or (dc[0] <= 1e-4 and dc[1] <= 1e-4 and abs(c["score"] - oc["score"]) <= 1e-6 * max(oc["score"], 1e-12))))
    if not ok:
        bad += 1
        print("MISMATCH case %d n=%d: quatro valid %s/%s dT %.2e %.2e | c2f valid %s/%s dT %.2e %.2e score %.6g/%.6g" % (
            case, n, valid, o["valid"], dq[0], dq[1], c["valid"], oc["valid"], dc[0], dc[1], c["score"], oc["score"]))
print("%d cases, %d mismatches, %.1f s" % (ncases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
