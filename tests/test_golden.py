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
