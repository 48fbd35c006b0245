"""Second, independent restatement of Nano-GICP in numpy/scipy ("a second,
independent Python oracle cross-checks the C++ oracle stage by stage").

ORACLE - TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
Uses scipy.spatial.cKDTree (f64 distances on the f32 coordinates) and numpy.linalg
(svd / inv / solve) - i.e. different code AND different numerics from oracle/*.cpp, so
agreement between the two pins each against the shared specification.
"""
import numpy as np
from scipy.spatial import cKDTree


def covariances(pts, k):
    """A.1.3: k-NN incl. self, cov = X X^T / k, PLANE regularisation via SVD, values (1,1,1e-3)."""
    p = pts.astype(np.float64)
    _, idx = cKDTree(p).query(p, k=k)
    nb = p[idx]                                    # n,k,3
    nb = nb - nb.mean(1, keepdims=True)
    cov = np.einsum("nki,nkj->nij", nb, nb) / k
    # cov is symmetric PSD, so its SVD is its eigendecomposition with U == V.  For an exactly
    # singular cov (noise-free planar patch) LAPACK's svd may return u3 = -v3; the specification
    # both oracles follow is U == V (C = V diag(1,1,1e-3) V^T), hence eigh, eigenvalues descending.
    _, V = np.linalg.eigh(cov)
    V = V[:, :, ::-1]
    return np.einsum("nik,k,njk->nij", V, np.array([1.0, 1.0, 1e-3]), V), idx


def so3_exp(om):
    th2 = float(om @ om)
    if th2 < 1e-10:
        imag = 0.5 - th2 / 48.0 + th2 * th2 / 3840.0
        real = 1.0 - th2 / 8.0 + th2 * th2 / 384.0
    else:
        th = np.sqrt(th2)
        imag = np.sin(th / 2) / th
        real = np.cos(th / 2)
    w, (x, y, z) = real, imag * om
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def skew(v):
    z = np.zeros(len(v))
    return np.stack([np.stack([z, -v[:, 2], v[:, 1]], 1), np.stack([v[:, 2], z, -v[:, 0]], 1),
                     np.stack([-v[:, 1], v[:, 0], z], 1)], 1)


class PyGicp:
    def __init__(self, src, tgt, k=20, max_iter=64, max_corr_dist=np.inf, trans_eps=5e-4, rot_eps=2e-3,
                 optimizer="lm", lm_max_iter=10, lm_init_lambda_factor=1e-9):
        self.src = src.astype(np.float32); self.tgt = tgt.astype(np.float32)
        self.k, self.max_iter, self.thr = k, max_iter, max_corr_dist
        self.trans_eps, self.rot_eps, self.optimizer = trans_eps, rot_eps, optimizer
        self.lm_max_iter, self.lm_f = lm_max_iter, lm_init_lambda_factor
        self.cs, _ = covariances(self.src, k)
        self.ct, _ = covariances(self.tgt, k)
        self.tree = cKDTree(self.tgt.astype(np.float64))

    def update_correspondences(self, T):
        Tf = T.astype(np.float32)
        q = ((Tf[:3, 0] * self.src[:, :1] + Tf[:3, 1] * self.src[:, 1:2]) + Tf[:3, 2] * self.src[:, 2:3]) + Tf[:3, 3]
        d, j = self.tree.query(q.astype(np.float64), k=1)
        self.valid = d * d < self.thr * self.thr
        self.j = j
        R = T[:3, :3]
        RCR = self.ct[j] + R @ self.cs @ R.T
        self.M = np.linalg.inv(RCR)

    def _err(self, T):
        tA = self.src.astype(np.float64) @ T[:3, :3].T + T[:3, 3]
        e = self.tgt[self.j].astype(np.float64) - tA
        return e, tA

    def linearize(self, T):
        self.update_correspondences(T)
        e, tA = self._err(T)
        v = self.valid
        J = np.concatenate([skew(tA), -np.broadcast_to(np.eye(3), (len(tA), 3, 3))], 2)   # n,3,6
        MJ = self.M @ J
        H = np.einsum("nri,nrj->ij", J[v], MJ[v])
        b = np.einsum("nri,nr->i", J[v], np.einsum("nij,nj->ni", self.M, e)[v])
        y = np.einsum("ni,nij,nj->", e[v], self.M[v], e[v])
        return H, b, float(y)

    def compute_error(self, T):
        e, _ = self._err(T)
        v = self.valid
        return float(np.einsum("ni,nij,nj->", e[v], self.M[v], e[v]))

    def is_converged(self, d):
        return max(np.abs(d[:3, :3] - np.eye(3)).max() / self.rot_eps, np.abs(d[:3, 3]).max() / self.trans_eps) < 1

    def align(self, guess=None):
        x0 = np.eye(4) if guess is None else guess.astype(np.float64).copy()
        lam = -1.0; converged = False; iters = 0
        for i in range(self.max_iter):
            if converged:
                break
            iters = i + 1
            H, b, y0 = self.linearize(x0)
            ok = False
            if self.optimizer == "gn":
                d = np.linalg.solve(H, -b)
                delta = np.eye(4); delta[:3, :3] = so3_exp(d[:3]); delta[:3, 3] = d[3:]
                x0 = delta @ x0; ok = True
            else:
                if lam < 0:
                    lam = self.lm_f * np.abs(np.diag(H)).max()
                nu = 2.0
                for _ in range(self.lm_max_iter):
                    d = np.linalg.solve(H + lam * np.eye(6), -b)
                    delta = np.eye(4); delta[:3, :3] = so3_exp(d[:3]); delta[:3, 3] = d[3:]
                    xi = delta @ x0
                    yi = self.compute_error(xi)
                    rho = (y0 - yi) / (d @ (lam * d - b))
                    if rho < 0:
                        if self.is_converged(delta):
                            ok = True; break
                        lam *= nu; nu *= 2; continue
                    x0 = xi; lam *= max(1.0 / 3.0, 1 - (2 * rho - 1) ** 3); ok = True
                    break
            if not ok:
                break
            converged = self.is_converged(delta)
        Tf = x0.astype(np.float32)
        q = Tf[:3, 0] * self.src[:, :1] + (Tf[:3, 1] * self.src[:, 1:2] + (Tf[:3, 2] * self.src[:, 2:3] + Tf[:3, 3]))
        d, _ = self.tree.query(q.astype(np.float64), k=1)
        return dict(T=x0, Tf=Tf, iterations=iters, converged=converged, fitness=float(np.mean((d * d).astype(np.float32).astype(np.float64))))
