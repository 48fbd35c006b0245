"""Known-answer tests pinning the CPU oracle.
The reference ships no tests or golden vectors, so these KATs + the
cross-oracle checks in test_oracle_cross.py are what anchor the restatement."""
import numpy as np
import pytest
from qn_amd import synth


def rodrigues(om):
    th = np.linalg.norm(om)
    if th < 1e-12:
        return np.eye(3)
    k = om / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def test_so3_exp_matches_rodrigues(oracle):
    rng = np.random.default_rng(0)
    for s in [1e-7, 1e-4, 1e-2, 0.3, 2.0]:
        om = rng.normal(size=3); om *= s / np.linalg.norm(om)
        assert np.abs(oracle.so3_exp(om) - rodrigues(om)).max() < 1e-12


def test_sym_eig3_vs_numpy(oracle):
    rng = np.random.default_rng(1)
    for _ in range(50):
        A = rng.normal(size=(3, 3)); A = A @ A.T
        w, V = oracle.sym_eig3(A)
        wn = np.sort(np.linalg.eigvalsh(A))[::-1]
        assert np.allclose(w, wn, rtol=1e-12, atol=1e-13)
        assert np.abs(V @ np.diag(w) @ V.T - A).max() < 1e-12
        assert np.abs(V.T @ V - np.eye(3)).max() < 1e-13


def test_ldlt6_vs_numpy(oracle):
    rng = np.random.default_rng(2)
    for _ in range(20):
        A = rng.normal(size=(6, 6)); A = A @ A.T + 1e-3 * np.eye(6)
        b = rng.normal(size=6)
        assert np.allclose(oracle.ldlt_solve6(A, b), np.linalg.solve(A, b), rtol=1e-9, atol=1e-11)


def test_knn_exact_vs_bruteforce(oracle):
    rng = np.random.default_rng(3)
    pts = rng.uniform(-5, 5, size=(700, 3)).astype(np.float32)
    pts[100:120] = pts[50]            # duplicates -> distance ties, lowest index must win
    g = oracle.GicpOracle()
    g.set_source(pts)
    idx, d2 = g.knn(0, pts, 12)
    d = pts[:, None, :] - pts[None, :, :]
    D = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]     # same f32 op order
    assert D.dtype == np.float32
    order = np.lexsort((np.broadcast_to(np.arange(len(pts)), D.shape), D), axis=1)[:, :12]
    assert np.array_equal(idx, order)
    assert np.array_equal(d2, np.take_along_axis(D, order, 1))


def test_plane_covariance_closed_form(oracle):
    """Points on an exact plane => C = I - 0.999 n n^T."""
    rng = np.random.default_rng(4)
    n = np.array([0.3, -0.5, 0.81]); n /= np.linalg.norm(n)
    u = np.cross(n, [1, 0, 0]); u /= np.linalg.norm(u); v = np.cross(n, u)
    ab = rng.uniform(-3, 3, size=(400, 2))
    pts = (ab[:, :1] * u + ab[:, 1:] * v).astype(np.float64)
    g = oracle.GicpOracle(k=10)
    g.set_source(pts.astype(np.float32)); g.compute_covariances(0)
    C = g.covariances(0)
    assert np.abs(C - (np.eye(3) - 0.999 * np.outer(n, n))).max() < 5e-6   # f32 rounding of the inputs


def test_three_point_linearize_by_hand(oracle):
    """Hand-computable H, b, cost with identity Mahalanobis-like covariances."""
    src = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [1, 0, 1]], dtype=np.float32)
    t = np.array([0.01, -0.02, 0.03])
    tgt = (src + t).astype(np.float32)
    g = oracle.GicpOracle(k=6)
    g.set_source(src); g.compute_covariances(0); g.set_target(tgt); g.compute_covariances(1)
    H, b, e, corr, sqd = g.linearize(np.eye(4))
    assert np.array_equal(corr, np.arange(6))
    Cs, Ct = g.covariances(0), g.covariances(1)
    He = np.zeros((6, 6)); be = np.zeros(6); ee = 0
    for i in range(6):
        M = np.linalg.inv(Ct[i] + Cs[i])
        p = src[i].astype(np.float64); err = tgt[i].astype(np.float64) - p
        J = np.hstack([np.array([[0, -p[2], p[1]], [p[2], 0, -p[0]], [-p[1], p[0], 0]]), -np.eye(3)])
        He += J.T @ M @ J; be += J.T @ M @ err; ee += err @ M @ err
    assert np.allclose(H, He, rtol=1e-12, atol=1e-14) and np.allclose(b, be, rtol=1e-12, atol=1e-14)
    assert abs(e - ee) < 1e-15 + 1e-12 * ee


def test_identity_pair_score_zero(oracle):
    src, _, _ = synth.make_pair(5, 1500, extent=30.0)
    r = oracle.icp_alignment(src, src, k=15)
    assert r["converged"] and r["score"] == 0.0
    assert np.abs(r["T"] - np.eye(4)).max() < 1e-6
    assert r["iterations"] == 1


@pytest.mark.parametrize("opt", ["lm", "gn"])
def test_recover_known_transform_noise_free(oracle, opt):
    """Source = rigidly transformed target => T recovered (tight thresholds)."""
    rng = np.random.default_rng(7)
    src, _, _ = synth.make_pair(11, 3000, extent=30.0)
    T = synth.random_gt(rng, "gicp")
    tgt = (src.astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    g = oracle.GicpOracle(k=15, max_iter=64, trans_eps=1e-7, rot_eps=1e-7, optimizer=opt)
    g.set_source(src); g.compute_covariances(0); g.set_target(tgt); g.compute_covariances(1)
    r = g.align()
    dt, dr = synth.pose_error(r["T"], T)
    assert dt < 2e-5 and dr < 2e-6, (dt, dr, r["iterations"])
    assert r["fitness"] < 1e-9


def test_noisy_pair_within_noise(oracle):
    src, tgt, T = synth.make_pair(21, 4000, extent=40.0)
    r = oracle.icp_alignment(src, tgt)
    dt, dr = synth.pose_error(r["T"], T)
    assert r["valid"] and dt < 0.05 and dr < 0.005, (dt, dr, r["score"])
