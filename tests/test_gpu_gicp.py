"""GPU-vs-oracle parity for the Nano-GICP path, through the C-ABI.
Integer/index work is compared bit-exact; floating point within the tolerance written in each test;
the end-to-end bar is BASELINE.json's: <= 1e-4 m and <= 1e-4 rad on identical inputs."""
import numpy as np
import pytest
from qn_amd import synth

pytestmark = pytest.mark.gpu

TOL_T, TOL_R = 1e-4, 1e-4


@pytest.fixture(scope="module")
def eng():
    from qn_amd import engine
    ctx = engine.Context(120000)
    yield engine, ctx
    ctx.close()


def _pair_small():
    return synth.make_pair(41, 6000, extent=45.0)


def test_knn_bit_exact(eng, oracle):
    engine, ctx = eng
    src, _, _ = _pair_small()
    src = src.copy(); src[200:215] = src[100]          # exact duplicates: ties must resolve to the lowest index
    g = engine.NanoGICP(ctx); g.setInputSource(src)
    o = oracle.GicpOracle(); o.set_source(src)
    for k in (1, 15, 16, 17, 20, 24, 25, 32):      # both histogram list capacities (k <= 24 / k > 24) and every BestK<KMAX> tail
        idx, d2 = g.knn(0, k)
        oi, od = o.knn(0, src, k)
        assert np.array_equal(idx, oi), k
        assert np.array_equal(d2, od), k


def test_knn_sorted_list_path_bit_exact(oracle):
    """The general k-NN path (wave_search + BestK<KMAX>, knn_hist = 0: the tail the histogram path falls back on) alone."""
    from qn_amd import engine
    ctx = engine.Context(8192)
    src, _, _ = _pair_small()
    ctx.debug_set("knn_hist", 0)
    g = engine.NanoGICP(ctx); g.setInputSource(src)
    o = oracle.GicpOracle(); o.set_source(src)
    for k in (15, 20, 24, 32):
        idx, d2 = g.knn(0, k)
        oi, od = o.knn(0, src, k)
        assert np.array_equal(idx, oi) and np.array_equal(d2, od), k
    ctx.close()


def test_knn_bit_exact_with_isolated_points_and_ties(eng, oracle):
    """Far queries (isolated points scattered in the bounding volume: the one-query-per-wave path with its second-level
    histogram and tile-wise box enumeration) and a regular lattice (every distance tied many times: list overflows fall back
    to the sorted-list kernel, ties must still resolve to the lowest index)."""
    engine, ctx = eng
    src, _, _ = synth.make_pair(43, 20000)
    rng = np.random.default_rng(9)
    lo, hi = src.min(0), src.max(0); hi[2] = lo[2] + 25.0
    src = src.copy(); sel = rng.choice(len(src), 600, replace=False)
    src[sel] = rng.uniform(lo, hi, size=(600, 3)).astype(np.float32)
    gx, gy = np.meshgrid(np.arange(60, dtype=np.float32) * 0.5, np.arange(60, dtype=np.float32) * 0.5)
    lattice = np.stack([gx.ravel() + 200.0, gy.ravel(), np.zeros(3600, np.float32)], 1).astype(np.float32)
    for cloud, ks in ((src, (15, 20, 32)), (lattice, (20,))):
        g = engine.NanoGICP(ctx); g.setInputSource(cloud)
        o = oracle.GicpOracle(); o.set_source(cloud)
        for k in ks:
            idx, d2 = g.knn(0, k); oi, od = o.knn(0, cloud, k)
            assert np.array_equal(idx, oi) and np.array_equal(d2, od), k


def test_knn_bit_exact_sparse_and_tiny(eng, oracle):
    """Sparse cloud (most queries leave the LDS-staged stencil -> exact ball fallback) and n < k."""
    engine, ctx = eng
    rng = np.random.default_rng(5)
    pts = rng.uniform(-40, 40, size=(900, 3)).astype(np.float32)
    g = engine.NanoGICP(ctx); g.setInputSource(pts)
    o = oracle.GicpOracle(); o.set_source(pts)
    idx, d2 = g.knn(0, 20); oi, od = o.knn(0, pts, 20)
    assert np.array_equal(idx, oi) and np.array_equal(d2, od)
    tiny = pts[:7]
    g.setInputSource(tiny); o.set_source(tiny)
    idx, d2 = g.knn(0, 15); oi, od = o.knn(0, tiny, 15)
    assert np.array_equal(idx, oi) and np.array_equal(d2, od)


@pytest.mark.parametrize("k", [15, 20])
def test_covariances(eng, oracle, k):
    engine, ctx = eng
    src, tgt, _ = _pair_small()
    g = engine.NanoGICP(ctx); g.setCorrespondenceRandomness(k)
    o = oracle.GicpOracle(k=k)
    for which, cloud in ((0, src), (1, tgt)):
        (g.setInputSource if which == 0 else g.setInputTarget)(cloud)
        assert (g.calculateSourceCovariances if which == 0 else g.calculateTargetCovariances)()
        (o.set_source if which == 0 else o.set_target)(cloud); o.compute_covariances(which)
        C, Co = g.covariances(which), o.covariances(which)
        assert np.abs(C - Co).max() < 1e-9          # same k-NN sets, same f64 Jacobi: only libm sqrt/div ulps


def test_linearize_and_error(eng, oracle):
    engine, ctx = eng
    src, tgt, T = _pair_small()
    g = engine.NanoGICP(ctx); g.setCorrespondenceRandomness(15); g.setMaxCorrespondenceDistance(52.5)
    g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
    o = oracle.GicpOracle(k=15, max_corr_dist=52.5)
    o.set_source(src); o.compute_covariances(0); o.set_target(tgt); o.compute_covariances(1)
    for X in (np.eye(4), T):
        H, b, e, corr, sqd = g.linearize(X)
        Ho, bo, eo, co, so = o.linearize(X)
        assert np.array_equal(corr, co)                     # 1-NN indices bit-exact
        assert np.array_equal(sqd, so)                      # f32 squared distances bit-exact
        assert np.abs(H - Ho).max() <= 1e-10 * np.abs(Ho).max()
        assert np.abs(b - bo).max() <= 1e-10 * np.abs(bo).max()
        assert abs(e - eo) <= 1e-10 * eo
        X2 = X.copy(); X2[:3, 3] += [0.03, -0.02, 0.01]
        assert abs(g.compute_error(X2) - o.compute_error(X2)) <= 1e-10 * eo


def test_max_corr_dist_gating(eng, oracle):
    engine, ctx = eng
    src, tgt, _ = _pair_small()
    g = engine.NanoGICP(ctx); g.setCorrespondenceRandomness(15); g.setMaxCorrespondenceDistance(0.5)
    g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
    o = oracle.GicpOracle(k=15, max_corr_dist=0.5)
    o.set_source(src); o.compute_covariances(0); o.set_target(tgt); o.compute_covariances(1)
    H, b, e, corr, sqd = g.linearize(np.eye(4)); Ho, bo, eo, co, so = o.linearize(np.eye(4))
    assert (co < 0).any() and np.array_equal(corr, co) and np.array_equal(sqd, so)


def _align_both(engine, ctx, oracle, src, tgt, **kw):
    k = kw.get("k", 15); opt = kw.get("optimizer", "lm")
    g = engine.NanoGICP(ctx)
    g.setCorrespondenceRandomness(k); g.setMaximumIterations(kw.get("max_iter", 32))
    g.setMaxCorrespondenceDistance(kw.get("max_corr_dist", 52.5)); g.setTransformationEpsilon(kw.get("trans_eps", 0.01))
    g.setOptimizer(opt); g.setForceIterations(kw.get("force", 0))
    g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
    g.align()
    r = g.result_dict()
    o = oracle.GicpOracle(k=k, max_iter=kw.get("max_iter", 32), max_corr_dist=kw.get("max_corr_dist", 52.5),
                          trans_eps=kw.get("trans_eps", 0.01), optimizer=opt, force_iterations=kw.get("force", 0))
    o.set_source(src); o.compute_covariances(0); o.set_target(tgt); o.compute_covariances(1)
    ro = o.align()
    return g, r, o, ro


def _check_parity(r, ro):
    assert r["iterations"] == ro["iterations"] and r["converged"] == ro["converged"]
    dt, dr = synth.pose_error(r["T"], ro["T"])
    assert dt <= TOL_T and dr <= TOL_R, (dt, dr)
    assert np.abs(r["Tf"] - ro["Tf"]).max() <= 1e-5
    assert abs(r["fitness"] - ro["fitness"]) <= 1e-6 * max(ro["fitness"], 1e-12)
    tr, tro = r["trace"], ro["trace"]
    assert tr.shape == tro.shape
    assert np.array_equal(tr[:, 5:], tro[:, 5:])                      # inner tries / accepted flags identical
    assert np.allclose(tr[:, 0], tro[:, 0], rtol=1e-8)                # y0 trajectory
    assert np.allclose(tr[:, 1], tro[:, 1], rtol=1e-6)                # lambda trajectory


@pytest.mark.parametrize("pair_id", range(50, 58))
def test_align_parity_reference_config(eng, oracle, pair_id):
    """8 seeded pairs at the reference's operating point (k=15, 32 iters, LM, eps 0.01)."""
    engine, ctx = eng
    src, tgt, T = synth.make_pair(pair_id, 5000, extent=45.0)
    g, r, o, ro = _align_both(engine, ctx, oracle, src, tgt)
    _check_parity(r, ro)
    assert np.abs(g.alignedCloud() - o.transformed_source(ro["Tf"])).max() <= 1e-4


@pytest.mark.parametrize("opt,eps", [("lm", 1e-5), ("gn", 1e-5), ("gn", 0.01)])
def test_align_parity_tight_and_gn(eng, oracle, opt, eps):
    engine, ctx = eng
    src, tgt, T = synth.make_pair(60, 8000, extent=50.0)
    g, r, o, ro = _align_both(engine, ctx, oracle, src, tgt, optimizer=opt, trans_eps=eps, max_iter=64, k=20)
    _check_parity(r, ro)


def test_align_forced_iterations_gn(eng, oracle):
    """BASELINE config 2 shape: k=20, 20 forced GN iterations."""
    engine, ctx = eng
    src, tgt, T = synth.make_pair(61, 8000, extent=50.0)
    g, r, o, ro = _align_both(engine, ctx, oracle, src, tgt, optimizer="gn", force=20, k=20)
    assert r["iterations"] == 20 == ro["iterations"]
    dt, dr = synth.pose_error(r["T"], ro["T"])
    assert dt <= TOL_T and dr <= TOL_R


def test_align_parity_30k(eng, oracle):
    """BASELINE config 1 size: two 30k-point clouds, reference operating point."""
    engine, ctx = eng
    src, tgt, T = synth.make_pair(1, 30000)
    g, r, o, ro = _align_both(engine, ctx, oracle, src, tgt)
    _check_parity(r, ro)
    dt, dr = synth.pose_error(r["T"], T)
    assert dt < 0.05 and dr < 0.005


def test_align_parity_100k(eng, oracle):
    """BASELINE config 2 size (100k x 100k, k=20) with the real stopping rule."""
    engine, ctx = eng
    src, tgt, T = synth.make_pair(0, 100000)
    g, r, o, ro = _align_both(engine, ctx, oracle, src, tgt, k=20)
    _check_parity(r, ro)


def test_icp_alignment_accept_logic(eng, oracle):
    engine, ctx = eng
    src, tgt, T = synth.make_pair(62, 6000, extent=45.0)
    r = engine.icp_alignment(ctx, src, tgt)
    ro = oracle.icp_alignment(src, tgt)
    assert r["valid"] == ro["valid"] and r["converged"] == ro["converged"]
    assert abs(r["score"] - ro["score"]) <= 1e-6 * ro["score"]
    dt, dr = synth.pose_error(r["T"], ro["T"])
    assert dt <= TOL_T and dr <= TOL_R
    # a badly overlapping pair must be rejected on score exactly as the oracle rejects it
    src2, tgt2, _ = synth.make_pair(63, 6000, extent=45.0, shift=18.0)
    r2, ro2 = engine.icp_alignment(ctx, src2, tgt2), oracle.icp_alignment(src2, tgt2)
    assert r2["valid"] == ro2["valid"]


def test_identity_pair(eng, oracle):
    engine, ctx = eng
    src, _, _ = synth.make_pair(64, 3000, extent=35.0)
    r = engine.icp_alignment(ctx, src, src)
    assert r["converged"] and r["score"] == 0.0 and r["iterations"] == 1
    assert np.abs(r["T"] - np.eye(4)).max() < 1e-6


def test_strided_pointxyzi_input(eng, oracle):
    """PointXYZI layout (32 B stride, intensity and padding ignored) gives the same answer as packed xyz."""
    engine, ctx = eng
    src, tgt, _ = synth.make_pair(65, 4000, extent=40.0)
    def xyzi(a):
        out = np.full((len(a), 8), 7.0, dtype=np.float32); out[:, :3] = a; out[:, 3] = 1.0; return out
    r1 = engine.icp_alignment(ctx, src, tgt)
    r2 = engine.icp_alignment(ctx, xyzi(src), xyzi(tgt))
    assert np.array_equal(r1["T"], r2["T"]) and r1["score"] == r2["score"]


def test_boundary_status_codes(eng):
    engine, ctx = eng
    g = engine.NanoGICP(ctx)
    empty = np.zeros((0, 3), dtype=np.float32)
    src, tgt, _ = synth.make_pair(66, 2000, extent=30.0)
    r = engine.icp_alignment(ctx, empty, tgt)
    assert not r["valid"] and not r["converged"]
    g.setInputSource(src)
    assert g.align() is None and not g.hasConverged()            # covariances/target missing -> not ready, never a crash
    bad = src.copy(); bad[5, 1] = np.nan
    # non-finite coordinates: the grid's numbers are computed on the device (no host round trip in setInputSource), so the refusal
    # arrives with the first call that synchronises - and the context is usable again afterwards
    def both(a, b):
        g.setInputSource(a); g.calculateSourceCovariances(); g.setInputTarget(b); g.calculateTargetCovariances()
    both(bad, tgt)
    with pytest.raises(engine.EngineError) as ei:
        g.align()
    assert ei.value.status == engine.QN_ERR_INVALID_ARG and "non-finite" in str(ei.value)
    assert g.align() is None                                      # the refused cloud is gone: not ready
    both(src, bad)
    with pytest.raises(engine.EngineError) as ei:
        g.align()
    assert ei.value.status == engine.QN_ERR_INVALID_ARG and "target" in str(ei.value)
    both(src, tgt)
    assert g.align() is not None and g.hasConverged()
    with pytest.raises(engine.EngineError) as ei:
        g.setInputSource(np.zeros((ctx.max_points + 1, 3), dtype=np.float32))
    assert ei.value.status == engine.QN_ERR_CAPACITY


def test_deterministic_rerun(eng):
    engine, ctx = eng
    src, tgt, _ = synth.make_pair(67, 6000, extent=45.0)
    a = engine.icp_alignment(ctx, src, tgt); b = engine.icp_alignment(ctx, src, tgt)
    assert np.array_equal(a["T"], b["T"]) and a["score"] == b["score"]


def test_ragged_sizes_and_far_from_origin(eng, oracle):
    """ns != nt, and map-frame coordinates kilometres from the origin (f32 cell arithmetic under stress)."""
    engine, ctx = eng
    src, _, _ = synth.make_pair(80, 3000, extent=40.0)
    _, tgt, _ = synth.make_pair(80, 9000, extent=40.0)
    for off in (np.zeros(3), np.array([5231.5, -8120.25, 312.0])):
        s = (src.astype(np.float64) + off).astype(np.float32); t = (tgt.astype(np.float64) + off).astype(np.float32)
        g, r, o, ro = _align_both(engine, ctx, oracle, s, t)
        _check_parity(r, ro)
        H, b, e, corr, sqd = g.linearize(np.eye(4)); Ho, bo, eo, co, so = o.linearize(np.eye(4))
        assert np.array_equal(corr, co) and np.array_equal(sqd, so)


def test_tiny_clouds_and_k_larger_than_cloud(eng, oracle):
    engine, ctx = eng
    rng = np.random.default_rng(9)
    src = rng.uniform(-2, 2, size=(11, 3)).astype(np.float32)
    tgt = (src + rng.normal(0, 0.01, src.shape) + [0.05, 0.02, 0.0]).astype(np.float32)
    g = engine.NanoGICP(ctx); g.setCorrespondenceRandomness(15)
    g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
    o = oracle.GicpOracle(k=15); o.set_source(src); o.compute_covariances(0); o.set_target(tgt); o.compute_covariances(1)
    assert np.abs(g.covariances(0) - o.covariances(0)).max() < 1e-9        # k > n: all n points are the neighbourhood
    H, b, e, corr, sqd = g.linearize(np.eye(4)); Ho, bo, eo, co, so = o.linearize(np.eye(4))
    assert np.array_equal(corr, co) and np.array_equal(sqd, so) and np.allclose(H, Ho, rtol=1e-9)
    assert g.align() is not None                                           # never a crash, whatever it converges to


def test_lm_path_with_rejections(eng, oracle):
    """A bad initial guess makes LM reject steps (inner tries > 1): the device controller must follow the same branches."""
    engine, ctx = eng
    src, tgt, T = synth.make_pair(81, 5000, extent=45.0)
    guess = np.eye(4, dtype=np.float32); guess[:3, 3] = [3.0, -2.5, 0.4]
    g = engine.NanoGICP(ctx); g.setCorrespondenceRandomness(15); g.setMaximumIterations(32); g.setMaxCorrespondenceDistance(52.5); g.setTransformationEpsilon(0.01)
    g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
    g.align(guess); r = g.result_dict()
    o = oracle.GicpOracle(k=15, max_iter=32, max_corr_dist=52.5, trans_eps=0.01)
    o.set_source(src); o.compute_covariances(0); o.set_target(tgt); o.compute_covariances(1)
    ro = o.align(guess.astype(np.float64))
    _check_parity(r, ro)


def test_randomized_configurations(eng, oracle):
    """Seeded sweep over sizes, k (all list capacities / BestK tails), optimizer, iteration cap, epsilon and ragged target
    sizes: same iteration count, same convergence flag, transform within the north-star tolerance, same score.
    (tools/gpu_parity_sweep.py is the long version: 60 cases up to 60k points.)"""
    engine, ctx = eng
    rng = np.random.default_rng(77)
    for case in range(10):
        n = int(rng.choice([300, 1500, 4000, 9000])); k = int(rng.choice([10, 15, 20, 24, 27, 32]))
        opt = str(rng.choice(["lm", "gn"])); max_iter = int(rng.choice([8, 32])); eps = float(rng.choice([0.01, 5e-4]))
        src, tgt, _ = synth.make_pair(7000 + case, n, extent=45.0 if n > 4000 else float(rng.choice([25.0, 45.0])))
        if case % 3 == 0:
            tgt = tgt[: int(0.7 * n)]
        g = engine.NanoGICP(ctx)
        g.setCorrespondenceRandomness(k); g.setMaximumIterations(max_iter); g.setMaxCorrespondenceDistance(52.5)
        g.setTransformationEpsilon(eps); g.setOptimizer(opt)
        g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
        g.align(); r = g.result_dict()
        o = oracle.GicpOracle(k=k, max_iter=max_iter, max_corr_dist=52.5, trans_eps=eps, optimizer=opt)
        o.set_source(src); o.compute_covariances(0); o.set_target(tgt); o.compute_covariances(1)
        ro = o.align()
        dt, dr = synth.pose_error(r["T"], ro["T"])
        assert r["iterations"] == ro["iterations"] and r["converged"] == ro["converged"], (case, n, k, opt)
        assert dt <= TOL_T and dr <= TOL_R, (case, dt, dr)
        assert abs(r["fitness"] - ro["fitness"]) <= 1e-6 * max(ro["fitness"], 1e-12)


@pytest.mark.parametrize("group", [2048, 16384])
def test_grouped_far_queries_equal_one_per_wave(eng, group):
    """Far-list entries served in groups of neighbours (wave_search_far16, the batch regime's list pass) find the neighbours the one-entry-per-wave search
    finds: same correspondences and squared distances for a misaligned pair, a partially overlapping pair and a scene 8 km from the origin."""
    engine, ctx = eng
    cases = [synth.make_pair(90, 30000)[:2], synth.make_pair(91, 30000, shift=24.0)[:2]]
    s8, t8, _ = synth.make_pair(92, 20000)
    off = np.array([8000.0, -7500.0, 120.0])
    cases.append(((s8.astype(np.float64) + off).astype(np.float32), (t8.astype(np.float64) + off).astype(np.float32)))
    for src, tgt in cases:
        out = []
        for fg in (0, group):
            ctx.debug_set("far_group", fg)
            g = engine.NanoGICP(ctx)
            g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
            H, b, e, corr, sqd = g.linearize(np.eye(4))
            out.append((np.array(corr).copy(), np.array(sqd).copy(), np.array(H).copy()))
        ctx.debug_set("far_group", -1)
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
