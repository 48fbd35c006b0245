"""ctypes wrapper around oracle/liboracle.so (the C++ CPU restatement).

ORACLE - TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/oracle_math.hpp).
Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
FLOAT_MAX = 3.4028234663852886e38


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".hpp"))]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.orc_gicp_create.restype = C.c_void_p
        _lib.orc_gicp_linearize.restype = C.c_double
        _lib.orc_gicp_compute_error.restype = C.c_double
        _lib.orc_gicp_fitness.restype = C.c_double
        if hasattr(_lib, "orc_quatro_create"):
            _lib.orc_quatro_create.restype = C.c_void_p
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class GicpOracle:
    """Mirrors the nano_gicp::NanoGICP call surface used at loop_closure.cpp:9-16,120-133."""

    def __init__(self, k=20, max_iter=64, max_corr_dist=FLOAT_MAX, trans_eps=5e-4, rot_eps=2e-3,
                 optimizer="lm", lm_max_iter=10, lm_init_lambda_factor=1e-9, force_iterations=0, num_threads=0):
        self._l = lib()
        self._h = C.c_void_p(self._l.orc_gicp_create())
        self.k = k
        self.max_iter = max_iter
        ip = np.array([k, max_iter, 0 if optimizer == "lm" else 1, lm_max_iter, force_iterations, num_threads], dtype=np.int32)
        dp = np.array([max_corr_dist, trans_eps, rot_eps, lm_init_lambda_factor], dtype=np.float64)
        self._l.orc_gicp_set_params(self._h, _p(ip), _p(dp))
        self.n = [0, 0]

    def __del__(self):
        try:
            self._l.orc_gicp_destroy(self._h)
        except Exception:
            pass

    def set_source(self, xyz):
        xyz = _f32(xyz); self.n[0] = len(xyz)
        self._l.orc_gicp_set_source(self._h, _p(xyz), C.c_int(len(xyz)))

    def set_target(self, xyz):
        xyz = _f32(xyz); self.n[1] = len(xyz)
        self._l.orc_gicp_set_target(self._h, _p(xyz), C.c_int(len(xyz)))

    def compute_covariances(self, which):
        self._l.orc_gicp_cov(self._h, C.c_int(which))

    def covariances(self, which):
        out = np.zeros((self.n[which], 3, 3))
        self._l.orc_gicp_get_cov(self._h, C.c_int(which), _p(out))
        return out

    def knn(self, which, q, k):
        q = _f32(q)
        idx = np.zeros((len(q), k), dtype=np.int32); d2 = np.zeros((len(q), k), dtype=np.float32)
        self._l.orc_gicp_knn(self._h, C.c_int(which), _p(q), C.c_int(len(q)), C.c_int(k), _p(idx), _p(d2))
        return idx, d2

    def linearize(self, T):
        T = _f64(T); H = np.zeros((6, 6)); b = np.zeros(6)
        corr = np.zeros(self.n[0], dtype=np.int32); sqd = np.zeros(self.n[0], dtype=np.float32)
        e = self._l.orc_gicp_linearize(self._h, _p(T), _p(H), _p(b), _p(corr), _p(sqd))
        return H, b, float(e), corr, sqd

    def compute_error(self, T):
        return float(self._l.orc_gicp_compute_error(self._h, _p(_f64(T))))

    def align(self, guess=None):
        guess = np.eye(4) if guess is None else _f64(guess)
        od = np.zeros(53); of = np.zeros(16, dtype=np.float32); oi = np.zeros(3, dtype=np.int32)
        cap = max(self.max_iter, 512)
        tr = np.zeros((cap, 7))
        self._l.orc_gicp_align(self._h, _p(guess), _p(od), _p(of), _p(oi), _p(tr), C.c_int(cap))
        return dict(T=od[:16].reshape(4, 4).copy(), H=od[16:52].reshape(6, 6).copy(), fitness=float(od[52]),
                    Tf=of.reshape(4, 4).copy(), iterations=int(oi[0]), converged=bool(oi[1]),
                    trace=tr[:oi[2]].copy())

    def fitness(self, Tf, max_range=1.7976931348623157e308):
        return float(self._l.orc_gicp_fitness(self._h, _p(_f32(Tf)), C.c_double(max_range)))

    def transformed_source(self, Tf):
        out = np.zeros((self.n[0], 3), dtype=np.float32)
        self._l.orc_gicp_transformed_source(self._h, _p(_f32(Tf)), _p(out))
        return out


def icp_alignment(src, dst, *, k=15, max_iter=32, max_corr_dist=52.5, trans_eps=0.01, score_thr=1.5, **kw):
    """LoopClosure::icpAlignment (loop_closure.cpp:110-136) at the reference's effective config
    (SURVEY Appendix C).  Returns dict(valid, converged, score, T (f64 cast of the f32 result))."""
    g = GicpOracle(k=k, max_iter=max_iter, max_corr_dist=max_corr_dist, trans_eps=trans_eps, **kw)
    g.set_source(src); g.compute_covariances(0)
    g.set_target(dst); g.compute_covariances(1)
    r = g.align()
    valid = r["converged"] and r["fitness"] < score_thr          # loop_closure.cpp:129
    return dict(valid=bool(valid), converged=r["converged"], score=r["fitness"],
                T=r["Tf"].astype(np.float64), iterations=r["iterations"], raw=r)


def so3_exp(om):
    R = np.zeros((3, 3)); lib().orc_so3_exp(_p(_f64(om)), _p(R)); return R


def sym_eig3(A):
    w = np.zeros(3); V = np.zeros((3, 3)); lib().orc_sym_eig3(_p(_f64(A)), _p(w), _p(V)); return w, V


def ldlt_solve6(A, rhs):
    x = np.zeros(6); lib().orc_ldlt_solve6(_p(_f64(A)), _p(_f64(rhs)), _p(x)); return x


def num_threads():
    return int(lib().orc_num_threads())
