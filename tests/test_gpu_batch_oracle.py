"""The BATCHED path (qn_gicp_align_batch: the pair as a grid dimension of every launch - NnLaneK, TickK with two rows per block, the grouped list pass,
source borrowing) DIRECTLY against the CPU oracle, at the sizes the bench's headline is quoted on.  tests/test_gpu_batch.py compares the lanes with the
classic path (HIP vs HIP); here every record of a batch is held against oracle/gicp_oracle.cpp, the restatement of what LoopClosure::icpAlignment computes
(fast_lio_sam_qn/src/loop_closure.cpp:110-136: set x2, cov x2, align, score, `converged && score < thr`).

(a) BASELINE configs[1] literally, 8 lanes: 100k x 100k, k = 20, 20 forced Gauss-Newton iterations, 4 aligned pairs + 4 80 %-overlap pairs (target window
    shifted 24 m: those lanes enter the far-query refresh regime) - T, iterations, the y0 trajectory of every iteration and the score of EVERY record;
(b) the reference's operating point (k = 15, LM, real stopping rule), 8 lanes at 30k;
(c) qn_multi_align_best with the candidates of one query sharing their source: T of every record against the oracle (not only against the lone path);
(d) a ragged batch (5 pairs through 4 lanes, lanes of different sizes) at a k whose selection kernel differs (k = 27 > 24)."""
import numpy as np
import pytest
from qn_amd import synth

pytestmark = pytest.mark.gpu
TOL_T, TOL_R = 1e-4, 1e-4          # north-star tolerance: <= 1e-4 m, <= 1e-4 rad


def params(engine, *, k=15, max_iter=32, optimizer="lm", force=0, eps=0.01):
    import ctypes as C
    p = engine.GicpParams(); engine.lib().qn_gicp_default_params(C.byref(p))
    p.k_correspondences = k; p.max_iterations = max_iter; p.max_corr_dist = 52.5; p.transformation_epsilon = eps
    p.optimizer = 1 if optimizer == "gn" else 0; p.force_iterations = force
    return p


def run_lanes(engine, cap, p, clouds, lanes):
    import ctypes as C
    ctx = engine.Context(cap)
    ctx.debug_set("batch_lanes", lanes)
    ctx.check(engine.lib().qn_gicp_set_params(ctx.h, C.byref(p)))
    res, val, st = engine.gicp_align_batch(ctx, [(s, len(s), t, len(t), 12, 0) for s, t in clouds], score_thr=1.5)
    n_runs = (len(clouds) + lanes - 1) // lanes
    # the traces of the LAST run's lanes (lane l of a run = its l-th pair)
    first = (n_runs - 1) * lanes
    traces = {first + l: engine.lane_trace(ctx, l) for l in range(len(clouds) - first)}
    assert ctx.debug_get("batch_pairs") == len(clouds) and ctx.debug_get("batch_launches") > 0, "the pairs did not go through the lanes"
    ctx.close()
    return res, val, st, traces


def oracle_run(oracle, s, t, *, k, max_iter, optimizer="lm", force=0, eps=0.01):
    o = oracle.GicpOracle(k=k, max_iter=max_iter, max_corr_dist=52.5, trans_eps=eps, optimizer=optimizer, force_iterations=force)
    o.set_source(s); o.compute_covariances(0); o.set_target(t); o.compute_covariances(1)
    return o.align()


def pose_error_f32(Ta, Tb):
    """pose error between two f32-ROUNDED matrices (the 96-byte records carry getFinalTransformation(), a Matrix4f): the rotation angle from the skew part
    (first order in the difference) - synth.pose_error's arccos turns one f32 ulp on the diagonal (6e-8) into 2.4e-4 rad"""
    D = np.linalg.inv(Ta) @ Tb
    W = 0.5 * (D[:3, :3] - D[:3, :3].T)
    return float(np.linalg.norm(D[:3, 3])), float(np.arcsin(min(1.0, np.linalg.norm([W[2, 1], W[0, 2], W[1, 0]]))))


def test_batched_bench_workload_every_record_vs_oracle(oracle):
    """(a): what `value`, `value_overlap80` and `batch64` of bench.py run."""
    from qn_amd import engine
    N, K, ITERS = 100000, 20, 20
    clouds = [synth.make_pair(60 + i, N, shift=(24.0 if i >= 4 else None))[:2] for i in range(8)]
    p = params(engine, k=K, max_iter=ITERS, optimizer="gn", force=ITERS)
    res, val, st, traces = run_lanes(engine, N + 1024, p, clouds, lanes=8)
    for i, ((s, t), r) in enumerate(zip(clouds, res)):
        ro = oracle_run(oracle, s, t, k=K, max_iter=ITERS, optimizer="gn", force=ITERS)
        assert st[i] == 0 and r.iterations == ITERS == ro["iterations"], (i, st[i], r.iterations)
        T = np.array(r.T64).reshape(4, 4)
        assert np.abs(T - ro["T"]).max() <= 1e-9, (i, np.abs(T - ro["T"]).max())
        tr, tro = traces[i], ro["trace"]
        assert tr.shape == tro.shape and tr.shape[0] == ITERS, (i, tr.shape, tro.shape)
        assert np.allclose(tr[:, 0], tro[:, 0], rtol=1e-8), (i, np.abs(tr[:, 0] / tro[:, 0] - 1).max())          # y0 (cost) of every iteration
        dt, dr = synth.pose_error(T, ro["T"])
        assert dt <= TOL_T and dr <= TOL_R, (i, dt, dr)
        assert abs(r.fitness - ro["fitness"]) <= 1e-6 * ro["fitness"], (i, r.fitness, ro["fitness"])
        assert np.abs(np.array(r.T, dtype=np.float32).reshape(4, 4) - ro["Tf"]).max() <= 1e-6                          # getFinalTransformation(): the f32 matrix
        assert bool(val[i]) == bool(ro["converged"] and ro["fitness"] < 1.5)                                         # loop_closure.cpp:129


def test_batched_reference_operating_point_vs_oracle(oracle):
    """(b): k = 15, LM, the real stopping rule (loop_closure.cpp:9-16 + config.yaml), 30k points: lanes stop at different iterations and sit in different far-query regimes"""
    from qn_amd import engine
    N = 30000
    clouds = [synth.make_pair(160 + i, N, shift=(24.0 if i % 2 else None))[:2] for i in range(8)]
    p = params(engine)
    res, val, st, traces = run_lanes(engine, N + 1024, p, clouds, lanes=8)
    iters = set()
    for i, ((s, t), r) in enumerate(zip(clouds, res)):
        o = oracle.icp_alignment(s, t)
        ro = o["raw"]
        assert st[i] == 0 and bool(val[i]) == o["valid"] and bool(r.converged) == o["converged"] and r.iterations == o["iterations"], (i, st[i], val[i], r.iterations, o["iterations"])
        dt, dr = synth.pose_error(np.array(r.T64).reshape(4, 4), ro["T"])
        assert dt <= TOL_T and dr <= TOL_R, (i, dt, dr)
        tr, tro = traces[i], ro["trace"]
        assert tr.shape == tro.shape and np.array_equal(tr[:, 5:], tro[:, 5:]), i                                    # inner tries and accepted flags of every outer iteration
        assert np.allclose(tr[:, 0], tro[:, 0], rtol=1e-8), i
        assert abs(r.fitness - o["score"]) <= 1e-6 * max(o["score"], 1e-12), (i, r.fitness, o["score"])
        iters.add(r.iterations)
    assert len(iters) > 1, "the lanes were meant to stop at different iterations"


def test_multi_align_best_shared_source_every_record_vs_oracle(oracle):
    """(c): the candidates of ONE query (same source buffer; lanes borrow one preparation of it) through qn_multi_align_best: every record's transform vs the oracle"""
    import torch
    import ctypes as C
    from qn_amd import engine
    src, tgt0, _ = synth.make_pair(260, 20000)
    tgts = []
    for v in range(9):
        a = 0.003 * v; c, s = np.cos(a), np.sin(a)
        R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
        tgts.append((tgt0.astype(np.float64) @ R.T + np.array([0.04 * v, -0.03 * v, 0.0])).astype(np.float32))
    mg = engine.MultiGpu(torch.cuda.device_count(), 21024, in_flight=2)
    mg.set_params(params(engine)); mg.debug_set("batch_lanes", 4)
    recs, best = mg.align_best([(src, len(src), t, len(t), 12, 0) for t in tgts])
    scores = []
    for v, (t, r) in enumerate(zip(tgts, recs)):
        o = oracle.icp_alignment(src, t)
        assert r.status == 0 and bool(r.valid) == o["valid"] and bool(r.converged) == o["converged"] and r.iterations == o["iterations"], (v, r.status, r.iterations, o["iterations"])
        dt, dr = pose_error_f32(np.array(r.T, dtype=np.float32).reshape(4, 4).astype(np.float64), o["T"])
        assert dt <= TOL_T and dr <= TOL_R, (v, dt, dr)
        assert abs(r.fitness - o["score"]) <= 1e-6 * o["score"]
        scores.append((o["score"], v) if o["valid"] else (np.inf, v))
    w = min(scores)
    assert (best is None) == (w[0] == np.inf)
    if best is not None:
        assert abs(best.fitness - w[0]) <= 1e-6 * w[0]
    mg.close()


def test_ragged_batch_large_k_vs_oracle(oracle):
    """(d): 5 pairs of different sizes through 4 lanes (a full run and a ragged one), k = 27 (the 48-candidate selection kernels), GN with the real stopping rule;
    isolated points scattered through the volume put work on the far-query paths of every lane"""
    from qn_amd import engine
    rng = np.random.default_rng(5)
    clouds = []
    for i in range(5):
        s, t, _ = synth.make_pair(360 + i, 9000 + 2500 * i, extent=45.0, shift=(9.0 if i % 2 else None))
        for c in (s, t):
            m = max(1, len(c) // 50); lo, hi = c.min(0), c.max(0); hi[2] = lo[2] + 20.0
            c[rng.choice(len(c), m, replace=False)] = rng.uniform(lo, hi, size=(m, 3)).astype(np.float32)
        clouds.append((s, t))
    p = params(engine, k=27, max_iter=24, optimizer="gn", eps=5e-4)
    res, val, st, _ = run_lanes(engine, 22000, p, clouds, lanes=4)
    for i, ((s, t), r) in enumerate(zip(clouds, res)):
        ro = oracle_run(oracle, s, t, k=27, max_iter=24, optimizer="gn", eps=5e-4)
        assert st[i] == 0 and r.iterations == ro["iterations"] and bool(r.converged) == ro["converged"], (i, r.iterations, ro["iterations"])
        dt, dr = synth.pose_error(np.array(r.T64).reshape(4, 4), ro["T"])
        assert dt <= TOL_T and dr <= TOL_R, (i, dt, dr)
        assert abs(r.fitness - ro["fitness"]) <= 1e-6 * max(ro["fitness"], 1e-12)
