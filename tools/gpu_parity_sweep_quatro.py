"""Stress tool (GPU box): Quatro coarse stage and coarse-to-fine, GPU vs oracle, over seeded pairs (yaw up to 180 deg).
usage: python tools/gpu_parity_sweep_quatro.py [n_cases [seed]] [--lanes B [--contexts C]]
  --lanes B: the coarse-to-fine registrations go through qn_coarse_to_fine_align_batch, B pairs per run of a context (ragged runs, pairs of different sizes, the occasional
             pair Quatro cannot register) - every record against the oracle AND against the one-pair entry point (bit for bit)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np
from qn_amd import engine, synth
from oracle import oracle as orc          # checker only
ap = argparse.ArgumentParser()
ap.add_argument("ncases", nargs="?", type=int, default=16); ap.add_argument("seed", nargs="?", type=int, default=0)
ap.add_argument("--lanes", type=int, default=0); ap.add_argument("--contexts", type=int, default=2)
a = ap.parse_args()
ncases = a.ncases; bad = 0; t0 = time.time()


def make(case):
    n = [4000, 7000, 12000, 20000][case % 4]
    return synth.make_pair(900 + 100 * a.seed + case, n, extent=45.0 if n <= 12000 else 70.0, mode="quatro")[:2]


def check_c2f(case, n, c, oc):
    dc = synth.pose_error(c["T"], oc["T"]) if oc["valid"] else (0.0, 0.0)
    ok = c["valid"] == oc["valid"] and (not oc["valid"] or (dc[0] <= 1e-4 and dc[1] <= 1e-4 and abs(c["score"] - oc["score"]) <= 1e-6 * max(oc["score"], 1e-12)))
    if not ok:
        print("MISMATCH case %d n=%d: c2f valid %s/%s dT %.2e %.2e score %.6g/%.6g" % (case, n, c["valid"], oc["valid"], dc[0], dc[1], c["score"], oc.get("score", float("nan"))))
    return ok


if a.lanes:
    import ctypes as C
    ctxs = []
    for _ in range(max(1, a.contexts)):
        cx = engine.Context(21024); cx.debug_set("batch_lanes", a.lanes)
        p = engine.GicpParams(); engine.lib().qn_gicp_default_params(C.byref(p))
        p.k_correspondences = 15; p.max_iterations = 32; p.max_corr_dist = 52.5; p.transformation_epsilon = 0.01
        cx.check(engine.lib().qn_gicp_set_params(cx.h, C.byref(p))); engine.Quatro(cx); ctxs.append(cx)
    one = engine.Context(21024)
    clouds = [make(c) for c in range(ncases)]
    got = engine.coarse_to_fine_align_batch(ctxs, [(s, len(s), t, len(t), 12, 0) for s, t in clouds])
    nvalid = 0
    for case, ((s, t), g) in enumerate(zip(clouds, got)):
        r = engine.coarse_to_fine_alignment(one, s, t)
        same = g["status"] == 0 and g["valid"] == r["valid"] and g["score"] == r["score"] and np.array_equal(g["T"], r["T"]) and np.array_equal(g["T_quatro"], r["T_quatro"])
        if not same:
            bad += 1; print("MISMATCH case %d n=%d: the batch record differs from the one-pair entry point (status %d)" % (case, len(s), g["status"]))
        oc = orc.coarse_to_fine_alignment(s, t)
        nvalid += int(oc["valid"])
        if not check_c2f(case, len(s), g, oc): bad += 1
    print("%d coarse-to-fine cases through %d contexts x %d lanes, %d valid per the oracle, %d mismatches, %.1f s" % (ncases, len(ctxs), a.lanes, nvalid, bad, time.time() - t0))
    sys.exit(1 if bad else 0)

ctx = engine.Context(40000)
for case in range(ncases):
    src, tgt = make(case); n = len(src)
    q = engine.Quatro(ctx)
    Tq, valid = q.align(src, tgt)
    o = orc.quatro_align(src, tgt)
    c = engine.coarse_to_fine_alignment(ctx, src, tgt)
    oc = orc.coarse_to_fine_alignment(src, tgt)
    dq = synth.pose_error(np.asarray(Tq), o["T"])
    ok = bool(valid) == bool(o["valid"]) and dq[0] <= 1e-4 and dq[1] <= 1e-4
    if not ok:
        print("MISMATCH case %d n=%d: quatro valid %s/%s dT %.2e %.2e" % (case, n, valid, o["valid"], dq[0], dq[1]))
    if not (ok and check_c2f(case, n, c, oc)): bad += 1
print("%d cases, %d mismatches, %.1f s" % (ncases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
