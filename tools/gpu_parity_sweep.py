"""Stress tool (GPU box): randomized GPU-vs-oracle parity over many seeded pairs, sizes, k, optimizers and stopping rules.
Prints one line per mismatch and a summary; exit code 1 on any mismatch.
usage: python tools/gpu_parity_sweep.py [n_cases [seed]] [--lanes B] [--hard]
  --hard: half of the pairs get their target re-posed by a further 0.03-0.09 rad of yaw and up to 0.5 m (the bench's "re-pose variants"): their pose still moves by metres at the
          third to sixth iteration, so the unseeded phase is extended per lane / per look (unseeded_goes_on, look_again) - lanes of ONE launch then run in different regimes
  seed != 0: other pairs and, for a third of the cases, 80 % overlap
  --lanes B: the cases go through qn_gicp_align_batch B at a time (the pair as a grid dimension: NnLaneK, TickK with two rows per block, the grouped list pass) -
             the path bench.py's headline runs; the parameters (k, optimizer, max_iter, eps) are drawn per batch (a batch shares its context's parameters),
             everything else (size, extent, overlap, ragged target, scattered isolated points) per pair."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np
from qn_amd import engine, synth
from oracle import oracle as orc          # checker only
ap = argparse.ArgumentParser()
ap.add_argument("ncases", nargs="?", type=int, default=40); ap.add_argument("seed", nargs="?", type=int, default=0); ap.add_argument("--lanes", type=int, default=0); ap.add_argument("--hard", action="store_true")
args = ap.parse_args()
ncases, seed, lanes = args.ncases, args.seed, args.lanes
rng = np.random.default_rng(2024 + seed + (7919 * lanes if lanes else 0) + (104729 if args.hard else 0))
ctx = engine.Context(70000)
if lanes:
    ctx.debug_set("batch_lanes", lanes)
bad = 0; t0 = time.time(); worst_t = worst_r = 0.0


def draw_params():
    return dict(k=int(rng.choice([10, 15, 20, 24, 27, 32])), opt=str(rng.choice(["lm", "gn"])), max_iter=int(rng.choice([8, 32])), eps=float(rng.choice([0.01, 5e-4])))


def draw_pair(case):
    n = int(rng.choice([300, 1500, 4000, 9000, 20000, 60000]))
    ext = float(rng.choice([25.0, 45.0, 70.0]))
    pid = 5000 + 1000 * seed + case + (500000 if lanes else 0)
    shift = None if (seed == 0 and not lanes) or rng.random() > 0.33 else 0.2 * (120.0 if n > 20000 else max(ext, 45.0) if n > 4000 else ext)
    if n > 20000: ext = 120.0
    elif n > 4000: ext = max(ext, 45.0)
    try:
        src, tgt, T = synth.make_pair(pid, n, extent=ext, shift=shift)
    except RuntimeError:
        src, tgt, T = synth.make_pair(pid, n)
    if args.hard and rng.random() < 0.5:                                    # a target that is a few degrees further off: large pose steps for several iterations
        a = float(rng.uniform(0.03, 0.09)) * (1 if rng.random() < 0.5 else -1); ca, sa = np.cos(a), np.sin(a)
        R = np.array([[ca, -sa, 0.0], [sa, ca, 0.0], [0.0, 0.0, 1.0]])
        tgt = (tgt.astype(np.float64) @ R.T + np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), 0.0])).astype(np.float32)
    if rng.random() < 0.25: tgt = tgt[: int(0.7 * n)]                      # ragged sizes
    if rng.random() < 0.35:                                                 # isolated points scattered in the bounding volume (far-query k-NN / 1-NN paths)
        frac = float(rng.choice([0.005, 0.02, 0.1]))
        for c in (src, tgt):
            m = max(1, int(frac * len(c))); lo, hi = c.min(0), c.max(0); hi[2] = lo[2] + 20.0
            c[rng.choice(len(c), m, replace=False)] = rng.uniform(lo, hi, size=(m, 3)).astype(np.float32)
    return pid, n, src, tgt


def check(case, pid, n, P, r, ro):
    global bad, worst_t, worst_r
    dt, dr = synth.pose_error(r["T"], ro["T"])
    worst_t, worst_r = max(worst_t, dt), max(worst_r, dr)
    ok = (r["iterations"] == ro["iterations"] and r["converged"] == ro["converged"] and dt <= 1e-4 and dr <= 1e-4
          and abs(r["fitness"] - ro["fitness"]) <= 1e-6 * max(ro["fitness"], 1e-12) and r.get("status", 0) == 0)
    if not ok:
        bad += 1
        print("MISMATCH case %d pair %d n=%d k=%d %s max_iter=%d eps=%g: status %d iters %d/%d conv %s/%s dT %.2e m %.2e rad fitness %.8g/%.8g" % (
            case, pid, n, P["k"], P["opt"], P["max_iter"], P["eps"], r.get("status", 0), r["iterations"], ro["iterations"], r["converged"], ro["converged"], dt, dr, r["fitness"], ro["fitness"]))


def oracle_run(P, src, tgt):
    o = orc.GicpOracle(k=P["k"], max_iter=P["max_iter"], max_corr_dist=52.5, trans_eps=P["eps"], optimizer=P["opt"])
    o.set_source(src); o.compute_covariances(0); o.set_target(tgt); o.compute_covariances(1)
    return o.align()


case = 0
while case < ncases:
    P = draw_params()
    g = engine.NanoGICP(ctx)
    g.setCorrespondenceRandomness(P["k"]); g.setMaximumIterations(P["max_iter"]); g.setMaxCorrespondenceDistance(52.5)
    g.setTransformationEpsilon(P["eps"]); g.setOptimizer(P["opt"])
    if not lanes:
        pid, n, src, tgt = draw_pair(case)
        g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
        g.align(); r = g.result_dict()
        check(case, pid, n, P, r, oracle_run(P, src, tgt))
        case += 1
        continue
    g.bind()
    m = min(lanes, ncases - case)
    if rng.random() < 0.3 and m > 1: m = int(rng.integers(1, m + 1))          # ragged runs (fewer pairs than lanes)
    batch = [draw_pair(case + i) for i in range(m)]
    before = ctx.debug_get("batch_pairs")
    res, val, st = engine.gicp_align_batch(ctx, [(s, len(s), t, len(t), 12, 0) for _, _, s, t in batch], score_thr=1.5)
    assert ctx.debug_get("batch_pairs") - before == m, "the batch did not go through the lanes"
    for i, ((pid, n, src, tgt), rr) in enumerate(zip(batch, res)):
        r = dict(T=np.array(rr.T64).reshape(4, 4), iterations=rr.iterations, converged=bool(rr.converged), fitness=rr.fitness, status=st[i])
        ro = oracle_run(P, src, tgt)
        check(case + i, pid, n, P, r, ro)
        if bool(val[i]) != bool(ro["converged"] and ro["fitness"] < 1.5):
            bad += 1; print("MISMATCH case %d pair %d: valid %d vs oracle" % (case + i, pid, val[i]))
    case += m
print("%d cases%s%s, %d mismatches, worst |dT| %.2e m %.2e rad, %.1f s" % (ncases, " through %d lanes" % lanes if lanes else "", " (hard: half re-posed)" if args.hard else "", bad, worst_t, worst_r, time.time() - t0))
sys.exit(1 if bad else 0)
