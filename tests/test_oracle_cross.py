"""C++ oracle vs the independent numpy/scipy oracle."""
import numpy as np
import pytest
from qn_amd import synth
from oracle import py_oracle


@pytest.fixture(scope="module")
def pair():
    return synth.make_pair(31, 2500, extent=35.0)


def test_covariances_agree(oracle, pair):
    src, _, _ = pair
    g = oracle.GicpOracle(k=15); g.set_source(src); g.compute_covariances(0)
    C = g.covariances(0)
    Cp, _ = py_oracle.covariances(src, 15)
    bad = np.abs(C - Cp).reshape(len(src), -1).max(1) > 1e-9
    assert bad.mean() < 2e-3        # ties at the k-th neighbour / degenerate neighbourhoods only


def test_linearize_agrees(oracle, pair):
    src, tgt, T = pair
    g = oracle.GicpOracle(k=15, max_corr_dist=52.5)
    g.set_source(src); g.compute_covariances(0); g.set_target(tgt); g.compute_covariances(1)
    p = py_oracle.PyGicp(src, tgt, k=15, max_corr_dist=52.5)
    # plug the C++ covariances into the python oracle so only the linearisation is compared
    p.cs, p.ct = g.covariances(0), g.covariances(1)
    for X in [np.eye(4), T]:
        H, b, e, corr, _ = g.linearize(X)
        Hp, bp, ep = p.linearize(X)
        assert (corr != p.j).mean() < 1e-3
        assert np.allclose(H, Hp, rtol=1e-6) and np.allclose(b, bp, rtol=1e-6, atol=1e-6 * np.abs(b).max())
        assert abs(e - ep) < 1e-6 * e
        X2 = X.copy(); X2[0, 3] += 0.05
        assert abs(g.compute_error(X2) - p.compute_error(X2)) < 1e-6 * e


@pytest.mark.parametrize("opt", ["lm", "gn"])
def test_align_agrees(oracle, pair, opt):
    src, tgt, T = pair
    g = oracle.GicpOracle(k=15, max_iter=32, max_corr_dist=52.5, trans_eps=0.01, optimizer=opt)
    g.set_source(src); g.compute_covariances(0); g.set_target(tgt); g.compute_covariances(1)
    r = g.align()
    p = py_oracle.PyGicp(src, tgt, k=15, max_iter=32, max_corr_dist=52.5, trans_eps=0.01, optimizer=opt)
    rp = p.align()
    dt, dr = synth.pose_error(r["T"], rp["T"])
    assert r["iterations"] == rp["iterations"] and r["converged"] == rp["converged"]
    assert dt < 1e-5 and dr < 1e-6, (dt, dr)
    assert abs(r["fitness"] - rp["fitness"]) < 1e-6 * max(r["fitness"], 1e-9)
