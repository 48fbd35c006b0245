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


# ---------------------------------------------------------------- Quatro
class QuatroParams:
    """The 10 ctor arguments of quatro<PointType> (loop_closure.cpp:18-27) at the reference's effective values
    (SURVEY Appendix C), plus the tuple-test seed."""

    def __init__(self, fpfh_normal_radius=0.9, fpfh_radius=1.5, noise_bound=0.3, rot_gnc_factor=1.4, rot_cost_diff_thr=1e-4,
                 rot_max_iter=50, estimate_scale=False, use_optimized_matching=True, distance_threshold=35.0,
                 max_num_corres=200, rng_seed=1, tuple_scale=0.95):
        self.dp = np.array([fpfh_normal_radius, fpfh_radius, noise_bound, rot_gnc_factor, rot_cost_diff_thr,
                            distance_threshold, tuple_scale], dtype=np.float64)
        self.ip = np.array([rot_max_iter, int(estimate_scale), int(use_optimized_matching), max_num_corres, rng_seed], dtype=np.int32)
        self.fpfh_normal_radius, self.fpfh_radius = fpfh_normal_radius, fpfh_radius


def quatro_fpfh(xyz, rn=0.9, rf=1.5):
    xyz = _f32(xyz); n = len(xyz)
    nrm = np.zeros((n, 3), np.float32); sp = np.zeros((n, 33), np.float32); fp = np.zeros((n, 33), np.float32)
    lib().orc_quatro_fpfh(_p(xyz), C.c_int(n), C.c_double(rn), C.c_double(rf), _p(nrm), _p(sp), _p(fp))
    return nrm, sp, fp


def quatro_feature_nn(q, c):
    q = _f32(q); c = _f32(c); nn = np.zeros(len(q), np.int32)
    lib().orc_quatro_feature_nn(_p(q), C.c_int(len(q)), _p(c), C.c_int(len(c)), _p(nn))
    return nn


def quatro_match(src, dst, fs, ft, p=None):
    p = p or QuatroParams(); src = _f32(src); dst = _f32(dst); fs = _f32(fs); ft = _f32(ft)
    cap = min(len(src), len(dst)) + 8
    mutual = np.zeros((cap, 2), np.int32); corres = np.zeros((cap, 2), np.int32); n_out = np.zeros(2, np.int32)
    lib().orc_quatro_match(_p(src), C.c_int(len(src)), _p(dst), C.c_int(len(dst)), _p(fs), _p(ft), _p(p.dp), _p(p.ip), _p(mutual), _p(corres), _p(n_out))
    return mutual[:n_out[0]].copy(), corres[:n_out[1]].copy()


def quatro_solve(src, dst, corres, p=None):
    p = p or QuatroParams(); src = _f32(src); dst = _f32(dst); corres = np.ascontiguousarray(corres, dtype=np.int32)
    T = np.zeros((4, 4)); oi = np.zeros(3, np.int32); clique = np.zeros(max(len(corres), 1), np.int32)
    lib().orc_quatro_solve(_p(src), _p(dst), _p(corres), C.c_int(len(corres)), _p(p.dp), _p(p.ip), _p(T), _p(oi), _p(clique))
    return dict(T=T, valid=bool(oi[0]), clique=clique[:oi[1]].copy(), rot_iterations=int(oi[2]))


def quatro_solve_scaled(src, dst, corres, p):
    """solve with estimate_scale: also returns TEASER++'s scale estimate"""
    src = _f32(src); dst = _f32(dst); corres = np.ascontiguousarray(corres, dtype=np.int32)
    T = np.zeros((4, 4)); oi = np.zeros(3, np.int32); clique = np.zeros(max(len(corres), 1), np.int32); od = np.zeros(1)
    lib().orc_quatro_solve_scaled(_p(src), _p(dst), _p(corres), C.c_int(len(corres)), _p(p.dp), _p(p.ip), _p(T), _p(oi), _p(clique), _p(od))
    return dict(T=T, valid=bool(oi[0]), clique=clique[:oi[1]].copy(), rot_iterations=int(oi[2]), scale=float(od[0]))


def quatro_align(src, dst, p=None):
    p = p or QuatroParams(); src = _f32(src); dst = _f32(dst)
    T = np.zeros((4, 4)); oi = np.zeros(4, np.int32); corres = np.zeros((4096, 2), np.int32)
    lib().orc_quatro_align(_p(src), C.c_int(len(src)), _p(dst), C.c_int(len(dst)), _p(p.dp), _p(p.ip), _p(T), _p(oi), _p(corres), C.c_int(4096))
    return dict(T=T, valid=bool(oi[0]), clique_size=int(oi[1]), rot_iterations=int(oi[2]), corres=corres[:oi[3]].copy())


def max_clique(adj):
    adj = np.ascontiguousarray(adj, dtype=np.uint8); n = len(adj); out = np.zeros(n, np.int32)
    m = lib().orc_max_clique(_p(adj), C.c_int(n), _p(out))
    return out[:m].copy()


def coarse_to_fine_alignment(src, dst, qp=None, **gicp_kw):
    """LoopClosure::coarseToFineAlignment (loop_closure.cpp:138-159): Quatro, transformPcd (f64 matrix on f32
    points, utilities.hpp:164-175), icpAlignment, compose T_gicp * T_quatro."""
    q = quatro_align(src, dst, qp)
    if not q["valid"]:
        return dict(valid=False, converged=False, score=1.7976931348623157e308, T=np.eye(4), quatro=q)
    Tq = q["T"]
    s = np.ascontiguousarray(src, dtype=np.float32).astype(np.float64)
    coarse = (((Tq[:3, 0] * s[:, :1] + Tq[:3, 1] * s[:, 1:2]) + Tq[:3, 2] * s[:, 2:3]) + Tq[:3, 3]).astype(np.float32)
    r = icp_alignment(coarse, dst, **gicp_kw)
    r["T"] = r["T"] @ Tq
    r["quatro"] = q; r["coarse"] = coarse
    return r


# ---------------------------------------------------------------- cloud assembly feeder
def transform_pcd(xyz, T):
    xyz = _f32(xyz); out = np.zeros_like(xyz)
    lib().orc_transform_pcd(_p(xyz), C.c_int(len(xyz)), _p(_f64(T)), _p(out)); return out


def voxel_grid(xyz, leaf):
    xyz = _f32(xyz); out = np.zeros_like(xyz)
    m = lib().orc_voxel_grid(_p(xyz), C.c_int(len(xyz)), C.c_float(leaf), _p(out))
    return out[:m].copy()


def assemble_submap(keyframes, poses, idxs, leaf):
    """setSrcAndDstCloud's inner loop (loop_closure.cpp:70-107): transformPcd of each keyframe, concatenate, voxelize."""
    parts = [transform_pcd(keyframes[i], poses[i]) for i in idxs]
    return voxel_grid(np.concatenate(parts, 0), leaf)


def loop_candidates(pos, stamp, query, radius, tdiff, max_k=64):
    pos = _f64(pos); stamp = _f64(stamp); out = np.zeros(max_k, np.int32)
    m = lib().orc_loop_candidates(_p(pos), _p(stamp), C.c_int(len(pos)), C.c_int(query), C.c_double(radius), C.c_double(tdiff), C.c_int(max_k), _p(out))
    return out[:m].copy()
