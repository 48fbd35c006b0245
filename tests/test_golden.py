"""Known-answer fixtures under tests/golden/ (made by tests/golden/make_golden.py; provenance in its header: produced by
the independent numpy/scipy restatement or in closed form - the reference has no golden vectors, parity stays unpinned).
CPU: the C++ oracle reproduces them.  GPU (-m gpu): the HIP path, through the C-ABI, reproduces them."""
import os
import numpy as np
import pytest
from qn_amd import synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GICP_CASES = ["gicp_lm_k15.npz", "gicp_gn_k20.npz"]


def _load(name):
    d = np.load(os.path.join(G, name))
    kw = dict(k=int(d["k"]), max_iter=int(d["max_iter"]), max_corr_dist=float(d["max_corr_dist"]), trans_eps=float(d["trans_eps"]),
              optimizer=str(d["optimizer"]))
    return d, kw


def _check_align(d, T, iterations, converged, fitness):
    dt, dr = synth.pose_error(np.asarray(T, dtype=np.float64), d["T"])
    assert iterations == int(d["iterations"]) and bool(converged) == bool(d["converged"])
    assert dt <= 1e-4 and dr <= 1e-4, (dt, dr)                                  # the north-star tolerance: 1e-4 m / 1e-4 rad
    assert abs(fitness - float(d["fitness"])) <= 1e-5 * float(d["fitness"])


def _check_cov(C, Cg):
    bad = np.abs(C[:len(Cg)] - Cg).reshape(len(Cg), -1).max(1) > 1e-8
    assert bad.mean() <= 0.05           # exact ties at the k-th neighbour may pick a different point (cKDTree vs lowest-index rule)


def _check_lin(d, H, b, e):
    assert np.allclose(H, d["H0"], rtol=1e-6, atol=1e-6 * np.abs(d["H0"]).max())
    assert np.allclose(b, d["b0"], rtol=1e-6, atol=1e-6 * np.abs(d["b0"]).max())
    assert abs(e - float(d["e0"])) <= 1e-6 * float(d["e0"])


# ------------------------------------------------------------------ CPU: the C++ oracle against the fixtures
@pytest.mark.parametrize("name", GICP_CASES)
def test_oracle_reproduces_gicp_golden(oracle, name):
    d, kw = _load(name)
    g = oracle.GicpOracle(**kw)
    g.set_source(d["src"]); g.compute_covariances(0); g.set_target(d["tgt"]); g.compute_covariances(1)
    _check_cov(g.covariances(0), d["cov_src"])
    H, b, e, _, _ = g.linearize(np.eye(4))
    _check_lin(d, H, b, e)
    r = g.align()
    _check_align(d, r["T"], r["iterations"], r["converged"], r["fitness"])


def test_oracle_plane_covariance_closed_form(oracle):
    d = np.load(os.path.join(G, "cov_plane.npz"))
    g = oracle.GicpOracle(k=int(d["k"])); g.set_source(d["pts"]); g.compute_covariances(0)
    assert np.abs(g.covariances(0) - d["C"]).max() < 1e-5     # f32 coordinates: the fitted normal is exact to ~1e-6


def test_oracle_so3_exp_rodrigues(oracle):
    d = np.load(os.path.join(G, "so3.npz"))
    for w, R in zip(d["omega"], d["R"]):
        assert np.abs(oracle.so3_exp(w) - R).max() < 1e-12


def test_oracle_voxel_grid_golden(oracle):
    d = np.load(os.path.join(G, "voxel.npz"))
    assert np.array_equal(oracle.voxel_grid(d["cloud"], float(d["leaf"])), d["out"])


# ------------------------------------------------------------------ GPU: the HIP path (C-ABI) against the same fixtures
@pytest.mark.gpu
@pytest.mark.parametrize("name", GICP_CASES)
def test_gpu_reproduces_gicp_golden(name):
    from qn_amd import engine
    d, kw = _load(name)
    ctx = engine.Context(4096)
    g = engine.NanoGICP(ctx)
    g.setCorrespondenceRandomness(kw["k"]); g.setMaximumIterations(kw["max_iter"]); g.setMaxCorrespondenceDistance(kw["max_corr_dist"])
    g.setTransformationEpsilon(kw["trans_eps"]); g.setOptimizer(kw["optimizer"])
    g.setInputSource(d["src"]); g.calculateSourceCovariances(); g.setInputTarget(d["tgt"]); g.calculateTargetCovariances()
    _check_cov(g.covariances(0), d["cov_src"])
    H, b, e, _, _ = g.linearize(np.eye(4))
    _check_lin(d, H, b, e)
    g.align()
    r = g.result_dict()
    _check_align(d, r["T"], r["iterations"], r["converged"], r["fitness"])
    ctx.close()


@pytest.mark.gpu
def test_gpu_plane_covariance_closed_form():
    from qn_amd import engine
    d = np.load(os.path.join(G, "cov_plane.npz"))
    ctx = engine.Context(1024); g = engine.NanoGICP(ctx)
    g.setCorrespondenceRandomness(int(d["k"])); g.setInputSource(d["pts"]); g.calculateSourceCovariances()
    assert np.abs(g.covariances(0) - d["C"]).max() < 1e-5
    ctx.close()


@pytest.mark.gpu
def test_gpu_voxel_grid_golden():
    from qn_amd import engine
    d = np.load(os.path.join(G, "voxel.npz"))
    store = engine.KeyframeStore()
    kid = store.add(d["cloud"])
    ptr, n = store.assemble([kid], [np.eye(4)], float(d["leaf"]), 0)
    assert np.array_equal(store.download(0, n), d["out"])
    store.close()


# ------------------------------------------------------------------ Quatro fixtures (oracle/py_quatro.py, the second restatement)
QUATRO_CASES = ["quatro_a.npz", "quatro_b.npz"]


def _rows_off(a, b, tol=1e-4):
    """(#rows whose finiteness differs, #rows with a bin off by more than tol) between two descriptor sets"""
    fa, fb = np.isfinite(a).all(1), np.isfinite(b).all(1)
    both = fa & fb
    off = np.zeros(len(a), bool)
    off[both] = np.abs(a[both] - b[both]).max(1) > tol
    return int((fa != fb).sum()), int(off.sum())


def _qparams(d, oracle=None, advanced=False):
    v = d["params"]
    kw = dict(fpfh_normal_radius=float(v[0]), fpfh_radius=float(v[1]), noise_bound=float(v[2]), rot_gnc_factor=float(v[3]), rot_cost_diff_thr=float(v[4]),
              rot_max_iter=int(v[5]), distance_threshold=float(v[6]), max_num_corres=int(v[7]), rng_seed=int(v[8]), use_optimized_matching=not advanced)
    return kw, float(v[9])


@pytest.mark.parametrize("name", QUATRO_CASES)
def test_oracle_reproduces_quatro_golden(oracle, name):
    """C++ oracle (hash-grid radius search, Jacobi, polynomial atan2, f32 brute-force matcher, running-sum TLS) against the
    numpy/scipy restatement's fixture (cKDTree, eigh, libm arctan2, KD-tree matcher, direct TLS)."""
    d = np.load(os.path.join(G, name))
    kw, tscale = _qparams(d)
    nrm, sp, fp = oracle.quatro_fpfh(d["src"], kw["fpfh_normal_radius"], kw["fpfh_radius"])
    assert np.nanmax(np.abs(nrm - d["normals_s"])) < 1e-6
    assert _rows_off(sp, d["spfh_s"]) == (0, 0), "SPFH: bins moved by the polynomial atan2 / f32 op order"
    nf, off = _rows_off(fp, d["fpfh_s"])
    assert nf == 0 and off == 0, (nf, off)                                   # <= 1e-4 per bin, every point
    _, _, fpt = oracle.quatro_fpfh(d["tgt"], kw["fpfh_normal_radius"], kw["fpfh_radius"])
    assert _rows_off(fpt, d["fpfh_t"]) == (0, 0)
    # matcher on the FIXTURE's descriptors: same cross-checked matches, same correspondences (optimized and advanced)
    for adv, mk, ck, qk, tk in ((False, "mutual", "corres", "clique", "T"), (True, "mutual_adv", "corres_adv", "clique_adv", "T_adv")):
        kw2, _ = _qparams(d, advanced=adv)
        p = oracle.QuatroParams(tuple_scale=tscale, **kw2)
        mutual, corres = oracle.quatro_match(d["src"], d["tgt"], d["fpfh_s"], d["fpfh_t"], p)
        assert np.array_equal(mutual, d[mk]) and np.array_equal(corres, d[ck])
        r = oracle.quatro_solve(d["src"], d["tgt"], corres, p)
        assert r["clique"].tolist() == d[qk].tolist() and r["valid"] == bool(d["valid_adv" if adv else "valid"])
        assert np.abs(r["T"] - d[tk]).max() < 1e-9
        assert r["rot_iterations"] == int(d["rot_iterations_adv" if adv else "rot_iterations"])
    # end to end
    r = oracle.quatro_align(d["src"], d["tgt"], oracle.QuatroParams(tuple_scale=tscale, **kw))
    assert np.array_equal(r["corres"], d["corres"]) and np.abs(r["T"] - d["T"]).max() < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("name", QUATRO_CASES)
def test_gpu_reproduces_quatro_golden(name):
    """The HIP path, through the C-ABI, against the same fixture: qn_fpfh, qn_match_optimized on the fixture's descriptors,
    qn_quatro_solve, and quatro<>::align end to end for both matchers."""
    from qn_amd import engine
    d = np.load(os.path.join(G, name))
    kw, tscale = _qparams(d)
    ctx = engine.Context(8192)
    q = engine.Quatro(ctx, **kw)
    n = len(d["src"])
    fp = engine.fpfh(ctx, d["src"])
    nf, off = _rows_off(fp, d["fpfh_s"])
    print("%s: FPFH rows off by > 1e-4: %d of %d (finiteness differs: %d)" % (name, off, n, nf))
    assert nf == 0 and off <= max(2, n // 500)          # a last-bit f32 normal (different f64 summation order) can move a pair across a bin edge
    corres = engine.match_optimized(ctx, d["src"], d["tgt"], d["fpfh_s"], d["fpfh_t"], kw["distance_threshold"], kw["max_num_corres"], tscale)
    assert np.array_equal(corres, d["corres"])
    s = engine.quatro_solve(d["src"], d["tgt"], d["corres"], q.p)
    assert s["clique"].tolist() == d["clique"].tolist() and np.abs(s["T"] - d["T"]).max() < 1e-9
    r = q.align(d["src"], d["tgt"], debug=True)
    if off == 0:
        assert np.array_equal(r["mutual"], d["mutual"]) and np.array_equal(r["corres"], d["corres"])
        assert r["clique"].tolist() == d["clique"].tolist() and np.abs(r["T"] - d["T"]).max() < 1e-9 and r["rot_iterations"] == int(d["rot_iterations"])
    dt, dr = synth.pose_error(r["T"], d["T"])
    assert r["valid"] == bool(d["valid"]) and (off > 0 or (dt <= 1e-4 and dr <= 1e-4))
    kwa, _ = _qparams(d, advanced=True)
    qa = engine.Quatro(ctx, **kwa)
    ra = qa.align(d["src"], d["tgt"], debug=True)
    if off == 0:
        assert np.array_equal(ra["corres"], d["corres_adv"]) and np.abs(ra["T"] - d["T_adv"]).max() < 1e-9
    # device-pointer entry point: same answer with both clouds resident in HBM
    import torch
    s_d = torch.from_numpy(np.ascontiguousarray(d["src"])).cuda(); t_d = torch.from_numpy(np.ascontiguousarray(d["tgt"])).cuda()
    q = engine.Quatro(ctx, **kw)                                   # parameters live in the context: back to optimizedMatching
    Td, vd = q.align_device(s_d.data_ptr(), len(d["src"]), t_d.data_ptr(), len(d["tgt"]), 12)
    assert vd == r["valid"] and np.array_equal(Td, r["T"])
    ctx.close()


@pytest.mark.gpu
def test_gpu_quatro_params_are_not_silently_ignored(oracle):
    """estimate_scale (estimat_scale_, include/loop_closure.h:44, passed at loop_closure.cpp:24) runs TEASER++'s TLS scale solver: on a target that IS the source scaled by
    1.03 and moved (FPFH at fixed radii is not scale invariant: a few per cent is what the descriptors still match across), quatro<>::align recovers scale, yaw and
    translation - same correspondence set, same scale, same pose as the oracle; use_optimized_matching = 0 selects advancedMatching (covered above); nonsense parameters are refused."""
    from qn_amd import engine, synth
    src, _, _ = synth.make_pair(340, 6000, extent=42.0, mode="quatro")
    th = 0.6; R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    rng = np.random.default_rng(5)
    tgt = (1.03 * (src.astype(np.float64) @ R.T) + np.array([4.0, -3.0, 0.2]) + rng.normal(0, 0.01, src.shape)).astype(np.float32)
    ctx = engine.Context(8192)
    kw = dict(estimate_scale=True)
    q = engine.Quatro(ctx, **kw)
    r = q.align(src, tgt, debug=True)
    o = oracle.quatro_align(src, tgt, oracle.QuatroParams(**kw))
    assert r["valid"] == o["valid"] and np.array_equal(r["corres"], o["corres"])
    os_ = oracle.quatro_solve_scaled(src, tgt, o["corres"], oracle.QuatroParams(**kw))
    assert abs(q.scale() - os_["scale"]) <= 1e-12 and np.abs(r["T"] - o["T"]).max() <= 1e-9
    assert r["valid"] and abs(q.scale() - 1.03) < 0.01, (r["valid"], q.scale())
    dt, dr = synth.pose_error(r["T"], np.block([[R, np.array([[4.0], [-3.0], [0.2]])], [np.zeros((1, 3)), np.ones((1, 1))]]))
    assert dt < 0.5 and dr < 0.02
    with pytest.raises(engine.EngineError) as ei:
        engine.Quatro(ctx, noise_bound=-1.0)
    assert ei.value.status == engine.QN_ERR_INVALID_ARG
    engine.Quatro(ctx, use_optimized_matching=False)
    ctx.close()
