"""Stress tool (GPU box): randomized GPU-vs-oracle parity over many seeded pairs, sizes, k, optimizers and stopping rules.
Prints one line per mismatch and a summary; exit code 1 on any mismatch.  usage: python tools/gpu_parity_sweep.py [n_cases [seed]]   (seed != 0: other pairs and, for a third of the cases, 80 % overlap)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np
from qn_amd import engine, synth
from oracle import oracle as orc          # checker only
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(2024 + seed)
ctx = engine.Context(70000)
bad = 0; t0 = time.time(); worst_t = worst_r = 0.0
for case in range(ncases):
    n = int(rng.choice([300, 1500, 4000, 9000, 20000, 60000]))
    k = int(rng.choice([10, 15, 20, 24, 27, 32]))
    opt = str(rng.choice(["lm", "gn"]))
    max_iter = int(rng.choice([8, 32]))
    eps = float(rng.choice([0.01, 5e-4]))
    ext = float(rng.choice([25.0, 45.0, 70.0]))
    pid = 5000 + 1000 * seed + case
    shift = None if seed == 0 or rng.random() > 0.33 else 0.2 * (120.0 if n > 20000 else max(ext, 45.0) if n > 4000 else ext)
    if n > 20000: ext = 120.0
    elif n > 4000: ext = max(ext, 45.0)
    try:
        src, tgt, T = synth.make_pair(pid, n, extent=ext, shift=shift)
    except RuntimeError:
        src, tgt, T = synth.make_pair(pid, n)
    if rng.random() < 0.25: tgt = tgt[: int(0.7 * n)]                      # ragged sizes
    if rng.random() < 0.35:                                                 # isolated points scattered in the bounding volume (far-query k-NN / 1-NN paths)
        frac = float(rng.choice([0.005, 0.02, 0.1]))
        for c in (src, tgt):
            m = max(1, int(frac * len(c))); lo, hi = c.min(0), c.max(0); hi[2] = lo[2] + 20.0
            c[rng.choice(len(c), m, replace=False)] = rng.uniform(lo, hi, size=(m, 3)).astype(np.float32)
    g = engine.NanoGICP(ctx)
    g.setCorrespondenceRandomness(k); g.setMaximumIterations(max_iter); g.setMaxCorrespondenceDistance(52.5)
    g.setTransformationEpsilon(eps); g.setOptimizer(opt)
    g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
    g.align(); r = g.result_dict()
    o = orc.GicpOracle(k=k, max_iter=max_iter, max_corr_dist=52.5, trans_eps=eps, optimizer=opt)
    o.set_source(src); o.compute_covariances(0); o.set_target(tgt); o.compute_covariances(1)
    ro = o.align()
    dt, dr = synth.pose_error(r["T"], ro["T"])
    worst_t, worst_r = max(worst_t, dt), max(worst_r, dr)
    ok = (r["iterations"] == ro["iterations"] and r["converged"] == ro["converged"] and dt <= 1e-4 and dr <= 1e-4
          and abs(r["fitness"] - ro["fitness"]) <= 1e-6 * max(ro["fitness"], 1e-12))
    if not ok:
        bad += 1
        print("MISMATCH case %d pair %d n=%d k=%d %s max_iter=%d eps=%g: iters %d/%d conv %s/%s dT %.2e m %.2e rad fitness %.8g/%.8g" % (
            case, pid, n, k, opt, max_iter, eps, r["iterations"], ro["iterations"], r["converged"], ro["converged"], dt, dr, r["fitness"], ro["fitness"]))
print("%d cases, %d mismatches, worst |dT| %.2e m %.2e rad, %.1f s" % (ncases, bad, worst_t, worst_r, time.time() - t0))
sys.exit(1 if bad else 0)
