"""ctypes binding of include/qn_engine.h.  Mirrors the reference's operator surface:
`NanoGICP` has the member functions LoopClosure calls on nano_gicp::NanoGICP
(fast_lio_sam_qn/src/loop_closure.cpp:9-16, 120-133), argument meaning and failure
behaviour included (no exceptions for a failed registration: hasConverged() == False)."""
import ctypes as C
import os
import numpy as np
from . import build as _build

QN_OK, QN_ERR_INVALID_ARG, QN_ERR_EMPTY_CLOUD, QN_ERR_CAPACITY, QN_ERR_NOT_READY, QN_ERR_HIP, QN_ERR_NO_DEVICE = range(7)
QN_SOURCE, QN_TARGET = 0, 1
FLOAT_MAX = 3.4028234663852886e38
KERNEL_FAMILIES = ["grid_build", "knn_cov", "nn_search", "nn_fallback", "accumulate", "solve", "fitness", "transform",
                   "fpfh_normals", "fpfh_spfh", "fpfh_fpfh", "feat_match", "gn_tick_fused", "knn_select", "match_tail", "far_refresh", "align_persist"]


class GicpParams(C.Structure):
    _fields_ = [("k_correspondences", C.c_int32), ("max_iterations", C.c_int32), ("max_corr_dist", C.c_double),
                ("transformation_epsilon", C.c_double), ("rotation_epsilon", C.c_double), ("optimizer", C.c_int32),
                ("lm_max_iterations", C.c_int32), ("lm_init_lambda_factor", C.c_double), ("force_iterations", C.c_int32),
                ("ransac_iterations", C.c_int32), ("ransac_outlier_threshold", C.c_double),
                ("euclidean_fitness_epsilon", C.c_double)]


class GicpResult(C.Structure):
    _fields_ = [("T", C.c_float * 16), ("T64", C.c_double * 16), ("H", C.c_double * 36), ("fitness", C.c_double),
                ("iterations", C.c_int32), ("converged", C.c_int32), ("lm_failed", C.c_int32), ("reserved", C.c_int32)]


class IterTrace(C.Structure):
    _fields_ = [("y0", C.c_double), ("lambda_", C.c_double), ("rho", C.c_double), ("max_dR", C.c_double),
                ("max_dt", C.c_double), ("inner", C.c_int32), ("accepted", C.c_int32)]


class KernelStat(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("launches", C.c_int64)]


class EngineError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("qn_engine status %d (%s)" % (status, msg))
        self.status = status


_lib = None
DEBUG_KNOBS_FROM_ENV = False     # harness opt-in for the QN_DEBUG_KNOBS environment variable (Context.__init__)


def lib():
    """Loads the in-tree libqn_engine.so.  Fails loudly when it is missing or unloadable."""
    global _lib
    if _lib is None:
        path = _build.LIB
        if not os.path.exists(path):
            raise ImportError("libqn_engine.so is not built (run __graft_entry__.build()); there is no CPU fallback")
        _lib = C.CDLL(path)
        _lib.qn_status_str.restype = C.c_char_p
        _lib.qn_last_error.restype = C.c_char_p
        _lib.qn_last_error.argtypes = [C.c_void_p]
        _lib.qn_ctx_stream.restype = C.c_void_p
        _lib.qn_ctx_stream.argtypes = [C.c_void_p]
        _lib.qn_ctx_destroy.argtypes = [C.c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    def __init__(self, max_points, device=0):
        self._l = lib()
        h = C.c_void_p()
        st = self._l.qn_ctx_create(C.c_int(device), C.c_uint32(max_points), C.byref(h))
        if st != QN_OK:
            raise EngineError(st, self._l.qn_status_str(st).decode())
        self.h = h
        self.max_points = max_points
        # developer tuning (qn_debug_set) for every context of the process - a whole test file or bench run under a knob.  Opt-in only:
        # a harness sets engine.DEBUG_KNOBS_FROM_ENV = True (tests/conftest.py, bench.py, tools/); a production host never reads the variable.
        knobs = os.environ.get("QN_DEBUG_KNOBS") if DEBUG_KNOBS_FROM_ENV else None
        if knobs:
            import json, sys
            try:
                for k, v in json.loads(knobs).items():
                    self.debug_set(k, float(v))
            except Exception:
                self.close()
                raise
            print("qn_amd: QN_DEBUG_KNOBS applied to a context: %s" % knobs, file=sys.stderr)

    def close(self):
        if getattr(self, "h", None):
            self._l.qn_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, st):
        if st != QN_OK:
            raise EngineError(st, self._l.qn_status_str(st).decode() + ": " + self._l.qn_last_error(self.h).decode())

    @property
    def stream(self):
        return self._l.qn_ctx_stream(self.h)

    def synchronize(self):
        self.check(self._l.qn_ctx_synchronize(self.h))

    def debug_set(self, key, value):
        self.check(self._l.qn_debug_set(self.h, key.encode(), C.c_double(value)))

    def debug_get(self, key):
        v = C.c_double()
        self.check(self._l.qn_debug_get(self.h, key.encode(), C.byref(v)))
        return v.value

    def grid_info(self, which):
        out = np.zeros(8)
        self.check(self._l.qn_debug_get_grid(self.h, C.c_int(which), _p(out)))
        return dict(origin=out[:3], cell=out[3], dims=out[4:7].astype(int), eps=out[7])

    # profiling hooks
    def prof_enable(self, on=True):
        self.check(self._l.qn_prof_enable(self.h, C.c_int(1 if on else 0)))

    def prof_reset(self):
        self.check(self._l.qn_prof_reset(self.h))

    def prof_stats(self):
        out = {}
        for i, name in enumerate(KERNEL_FAMILIES):
            ks = KernelStat()
            self.check(self._l.qn_prof_get(self.h, C.c_int(i), C.byref(ks)))
            out[name] = (ks.total_ms, ks.launches)
        return out


def _cloud_arg(xyz):
    """(pointer-holder, n, stride_bytes) for an (n,3) or (n,4)/(n,8) float32 array."""
    a = np.ascontiguousarray(xyz, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] < 3:
        raise ValueError("cloud must be (n, >=3) float32")
    return a, a.shape[0], a.shape[1] * 4


class NanoGICP:
    """nano_gicp::NanoGICP<PointType, PointType> as LoopClosure uses it."""

    def __init__(self, ctx):
        self.ctx = ctx
        self._l = ctx._l
        self.p = GicpParams()
        self._l.qn_gicp_default_params(C.byref(self.p))
        self._res = None
        self._n = [0, 0]
        self._push()

    def _push(self):
        self.ctx.check(self._l.qn_gicp_set_params(self.ctx.h, C.byref(self.p)))

    def bind(self):
        """Make this object's parameters the context's again (several NanoGICP objects may share one context; the context holds ONE set)."""
        self._push()

    # --- the 8 setters of loop_closure.cpp:9-16
    def setNumThreads(self, n):                      # CPU thread count: meaningless on the GPU, accepted
        self.num_threads = n

    def setCorrespondenceRandomness(self, k):
        self.p.k_correspondences = k; self._push()

    def setMaximumIterations(self, n):
        self.p.max_iterations = n; self._push()

    def setRANSACIterations(self, n):
        self.p.ransac_iterations = n; self._push()

    def setMaxCorrespondenceDistance(self, d):
        self.p.max_corr_dist = d; self._push()

    def setTransformationEpsilon(self, e):
        self.p.transformation_epsilon = e; self._push()

    def setEuclideanFitnessEpsilon(self, e):
        self.p.euclidean_fitness_epsilon = e; self._push()

    def setRANSACOutlierRejectionThreshold(self, t):
        self.p.ransac_outlier_threshold = t; self._push()

    # extras the reference leaves at defaults
    def setRotationEpsilon(self, e):
        self.p.rotation_epsilon = e; self._push()

    def setOptimizer(self, name):
        self.p.optimizer = 0 if name == "lm" else 1; self._push()

    def setForceIterations(self, n):
        self.p.force_iterations = n; self._push()

    # --- clouds
    def setInputSource(self, xyz):
        a, n, stride = _cloud_arg(xyz); self._n[0] = n
        st = self._l.qn_gicp_set_source(self.ctx.h, _p(a), C.c_uint32(n), C.c_uint32(stride))
        if st != QN_ERR_EMPTY_CLOUD:
            self.ctx.check(st)

    def setInputTarget(self, xyz):
        a, n, stride = _cloud_arg(xyz); self._n[1] = n
        st = self._l.qn_gicp_set_target(self.ctx.h, _p(a), C.c_uint32(n), C.c_uint32(stride))
        if st != QN_ERR_EMPTY_CLOUD:
            self.ctx.check(st)

    def setInputSourceDevice(self, ptr, n, stride):
        self._n[0] = n
        self.ctx.check(self._l.qn_gicp_set_source_device(self.ctx.h, C.c_void_p(ptr), C.c_uint32(n), C.c_uint32(stride)))

    def setInputTargetDevice(self, ptr, n, stride):
        self._n[1] = n
        self.ctx.check(self._l.qn_gicp_set_target_device(self.ctx.h, C.c_void_p(ptr), C.c_uint32(n), C.c_uint32(stride)))

    def calculateSourceCovariances(self):
        return self._l.qn_gicp_compute_covariances(self.ctx.h, C.c_int(QN_SOURCE)) == QN_OK

    def calculateTargetCovariances(self):
        return self._l.qn_gicp_compute_covariances(self.ctx.h, C.c_int(QN_TARGET)) == QN_OK

    def align(self, guess=None):
        """Returns the transformed source cloud like align(output) fills `output`; on an unusable
        input (empty cloud) the engine, like the reference, reports hasConverged() == False."""
        res = GicpResult()
        g = None if guess is None else np.ascontiguousarray(guess, dtype=np.float32)
        st = self._l.qn_gicp_align(self.ctx.h, None if g is None else _p(g), C.byref(res))
        if st in (QN_ERR_EMPTY_CLOUD, QN_ERR_NOT_READY):
            self._res = None
            return None
        self.ctx.check(st)
        self._res = res
        return res

    def alignedCloud(self):
        out = np.zeros((self._n[0], 4), dtype=np.float32)
        self.ctx.check(self._l.qn_gicp_transformed_source(self.ctx.h, _p(out), C.c_uint32(16)))
        return out[:, :3]

    def getFitnessScore(self, max_range=1.7976931348623157e308):
        if self._res is None:
            return 1.7976931348623157e308
        if max_range >= 1.7976931348623157e308:
            return self._res.fitness
        s = C.c_double()
        self.ctx.check(self._l.qn_gicp_fitness(self.ctx.h, C.c_double(max_range), C.byref(s)))
        return s.value

    def hasConverged(self):
        return bool(self._res.converged) if self._res is not None else False

    def getFinalTransformation(self):
        return np.array(self._res.T, dtype=np.float32).reshape(4, 4) if self._res is not None else np.eye(4, dtype=np.float32)

    # --- parity read-backs
    def result_dict(self):
        r = self._res
        return dict(T=np.array(r.T64).reshape(4, 4), Tf=np.array(r.T, dtype=np.float32).reshape(4, 4), H=np.array(r.H).reshape(6, 6),
                    fitness=r.fitness, iterations=r.iterations, converged=bool(r.converged), lm_failed=bool(r.lm_failed),
                    trace=self.trace())

    def trace(self):
        buf = (IterTrace * 1024)(); n = C.c_uint32()
        self.ctx.check(self._l.qn_gicp_get_trace(self.ctx.h, buf, C.c_uint32(1024), C.byref(n)))
        return np.array([[t.y0, t.lambda_, t.rho, t.max_dR, t.max_dt, t.inner, t.accepted] for t in buf[:n.value]]).reshape(-1, 7)

    def covariances(self, which):
        out = np.zeros((self._n[which], 3, 3))
        self.ctx.check(self._l.qn_gicp_get_covariances(self.ctx.h, C.c_int(which), _p(out)))
        return out

    def knn(self, which, k):
        idx = np.zeros((self._n[which], k), dtype=np.int32); d2 = np.zeros((self._n[which], k), dtype=np.float32)
        self.ctx.check(self._l.qn_gicp_knn(self.ctx.h, C.c_int(which), C.c_int(k), _p(idx), _p(d2)))
        return idx, d2

    def linearize(self, T):
        T = np.ascontiguousarray(T, dtype=np.float64)
        H = np.zeros((6, 6)); b = np.zeros(6); e = C.c_double()
        corr = np.zeros(self._n[0], dtype=np.int32); sqd = np.zeros(self._n[0], dtype=np.float32)
        self.ctx.check(self._l.qn_gicp_linearize(self.ctx.h, _p(T), _p(H), _p(b), C.byref(e), _p(corr), _p(sqd)))
        return H, b, e.value, corr, sqd

    def compute_error(self, T):
        T = np.ascontiguousarray(T, dtype=np.float64); e = C.c_double()
        self.ctx.check(self._l.qn_gicp_compute_error(self.ctx.h, _p(T), C.byref(e)))
        return e.value


class _ParamsScope:
    """The helpers below register at the reference's effective config on a context the caller handed in: the context's own NanoGICP
    parameters are read first (qn_gicp_get_params) and put back afterwards, so a caller's configured object keeps working."""

    def __init__(self, ctx):
        self.ctx = ctx; self.saved = GicpParams()
        ctx.check(ctx._l.qn_gicp_get_params(ctx.h, C.byref(self.saved)))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.ctx.check(self.ctx._l.qn_gicp_set_params(self.ctx.h, C.byref(self.saved)))
        return False


def _reference_gicp(ctx, k, max_iter, max_corr_dist, trans_eps):
    p = GicpParams(); ctx._l.qn_gicp_default_params(C.byref(p))
    p.k_correspondences, p.max_iterations, p.max_corr_dist, p.transformation_epsilon = k, max_iter, max_corr_dist, trans_eps
    ctx.check(ctx._l.qn_gicp_set_params(ctx.h, C.byref(p)))


def icp_alignment(ctx, src, dst, *, k=15, max_iter=32, max_corr_dist=52.5, trans_eps=0.01, score_thr=1.5):
    """LoopClosure::icpAlignment (loop_closure.cpp:110-136) at the reference's effective config."""
    a, ns, stride = _cloud_arg(src); b, nt, _ = _cloud_arg(dst)
    res = GicpResult(); valid = C.c_int()
    with _ParamsScope(ctx):
        _reference_gicp(ctx, k, max_iter, max_corr_dist, trans_eps)
        st = ctx._l.qn_icp_alignment(ctx.h, _p(a), C.c_uint32(ns), _p(b), C.c_uint32(nt), C.c_uint32(stride),
                                     C.c_double(score_thr), C.byref(res), C.byref(valid))
    if st == QN_ERR_EMPTY_CLOUD:
        return dict(valid=False, converged=False, score=1.7976931348623157e308, T=np.eye(4), iterations=0)
    ctx.check(st)
    return dict(valid=bool(valid.value), converged=bool(res.converged), score=res.fitness,
                T=np.array(res.T, dtype=np.float32).reshape(4, 4).astype(np.float64), iterations=res.iterations)


def fpfh(ctx, xyz):
    """qn_fpfh: n x 33 FPFH descriptors (NaN rows where PCL yields none), radii from the context's Quatro parameters."""
    a, n, stride = _cloud_arg(xyz)
    out = np.zeros((n, 33), dtype=np.float32)
    ctx.check(ctx._l.qn_fpfh(ctx.h, _p(a), C.c_uint32(n), C.c_uint32(stride), _p(out)))
    return out


def match_optimized(ctx, src, dst, fs, ft, thr_dist=35.0, num_max_corres=200, tuple_scale=0.95):
    """qn_match_optimized: Matcher::optimizedMatching on two clouds and their descriptors -> (m, 2) int32 pairs (src idx, dst idx)."""
    a, ns, stride = _cloud_arg(src); b, nt, _ = _cloud_arg(dst)
    fs = np.ascontiguousarray(fs, dtype=np.float32); ft = np.ascontiguousarray(ft, dtype=np.float32)
    pairs = np.zeros((num_max_corres, 2), dtype=np.int32); n = C.c_uint32()
    ctx.check(ctx._l.qn_match_optimized(ctx.h, _p(a), C.c_uint32(ns), _p(b), C.c_uint32(nt), C.c_uint32(stride), _p(fs), _p(ft),
                                        C.c_float(thr_dist), C.c_int(num_max_corres), C.c_float(tuple_scale), _p(pairs), C.c_uint32(num_max_corres), C.byref(n)))
    return pairs[:min(n.value, num_max_corres)].copy()


# ---------------------------------------------------------------------------------------- Quatro
class QuatroParams(C.Structure):
    _fields_ = [("fpfh_normal_radius", C.c_double), ("fpfh_radius", C.c_double), ("noise_bound", C.c_double),
                ("rot_gnc_factor", C.c_double), ("rot_cost_diff_thr", C.c_double), ("rot_max_iter", C.c_int32),
                ("estimate_scale", C.c_int32), ("use_optimized_matching", C.c_int32), ("distance_threshold", C.c_double),
                ("max_num_corres", C.c_int32), ("rng_seed", C.c_uint32), ("tuple_scale", C.c_double)]


def quatro_default_params():
    p = QuatroParams(); lib().qn_quatro_default_params(C.byref(p)); return p


class Quatro:
    """quatro<PointType> as LoopClosure uses it: the 10-argument constructor in the reference's order
    (fast_lio_sam_qn/src/loop_closure.cpp:18-27) and align(src, dst) -> (4x4 f64, is_converged) (:144)."""

    def __init__(self, ctx, fpfh_normal_radius=0.9, fpfh_radius=1.5, noise_bound=0.3, rot_gnc_factor=1.4, rot_cost_diff_thr=1e-4,
                 rot_max_iter=50, estimate_scale=False, use_optimized_matching=True, distance_threshold=35.0, max_num_corres=200,
                 rng_seed=1):
        self.ctx = ctx; self._l = ctx._l
        p = quatro_default_params()
        p.fpfh_normal_radius, p.fpfh_radius, p.noise_bound = fpfh_normal_radius, fpfh_radius, noise_bound
        p.rot_gnc_factor, p.rot_cost_diff_thr, p.rot_max_iter = rot_gnc_factor, rot_cost_diff_thr, rot_max_iter
        p.estimate_scale, p.use_optimized_matching = int(estimate_scale), int(use_optimized_matching)
        p.distance_threshold, p.max_num_corres, p.rng_seed = distance_threshold, max_num_corres, rng_seed
        self.p = p
        ctx.check(self._l.qn_quatro_set_params(ctx.h, C.byref(p)))
        self._n = [0, 0]

    def align(self, src, dst, debug=False):
        a, ns, stride = _cloud_arg(src); b, nt, _ = _cloud_arg(dst); self._n = [ns, nt]
        T = np.zeros((4, 4)); valid = C.c_int()
        if not debug:
            st = self._l.qn_quatro_align(self.ctx.h, _p(a), C.c_uint32(ns), _p(b), C.c_uint32(nt), C.c_uint32(stride), _p(T), C.byref(valid))
            if st != QN_ERR_EMPTY_CLOUD:
                self.ctx.check(st)
            return T, bool(valid.value)
        cap = min(ns, nt) + 8
        mutual = np.zeros((cap, 2), np.int32); corres = np.zeros((cap, 2), np.int32); clique = np.zeros(cap, np.int32)
        nm, nc, nq, it = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_int32()
        self.ctx.check(self._l.qn_quatro_align_debug(self.ctx.h, _p(a), C.c_uint32(ns), _p(b), C.c_uint32(nt), C.c_uint32(stride), _p(T), C.byref(valid),
                                                     _p(mutual), C.byref(nm), _p(corres), C.byref(nc), C.c_uint32(cap), _p(clique), C.byref(nq), C.byref(it)))
        return dict(T=T, valid=bool(valid.value), mutual=mutual[:nm.value].copy(), corres=corres[:nc.value].copy(),
                    clique=clique[:nq.value].copy(), rot_iterations=it.value)

    def align_device(self, src_ptr, ns, dst_ptr, nt, stride):
        """qn_quatro_align_device: both clouds are HIP device pointers (n points, `stride` bytes apart)."""
        self._n = [ns, nt]
        T = np.zeros((4, 4)); valid = C.c_int()
        self.ctx.check(self._l.qn_quatro_align_device(self.ctx.h, C.c_void_p(src_ptr), C.c_uint32(ns), C.c_void_p(dst_ptr), C.c_uint32(nt), C.c_uint32(stride), _p(T), C.byref(valid)))
        return T, bool(valid.value)

    def scale(self):
        """TEASER++'s scale estimate of the latest align (1 unless estimate_scale)"""
        v = C.c_double()
        self.ctx.check(self._l.qn_quatro_get_scale(self.ctx.h, C.byref(v)))
        return v.value

    def features(self, which):
        n = self._n[which]
        nrm = np.zeros((n, 3), np.float32); sp = np.zeros((n, 33), np.float32); fp = np.zeros((n, 33), np.float32)
        self.ctx.check(self._l.qn_quatro_get_features(self.ctx.h, C.c_int(which), _p(nrm), _p(sp), _p(fp)))
        return nrm, sp, fp


def quatro_solve(src, dst, corres, params=None):
    """Host-side Matcher tail + TEASER++/Quatro solve on given correspondences (no GPU involved)."""
    p = params or quatro_default_params()
    a, _, stride = _cloud_arg(src); b, _, _ = _cloud_arg(dst)
    corres = np.ascontiguousarray(corres, dtype=np.int32)
    T = np.zeros((4, 4)); valid = C.c_int(); clique = np.zeros(max(len(corres), 1), np.int32); nq = C.c_uint32()
    scale = C.c_double(1.0)
    st = lib().qn_quatro_solve_scaled(_p(a), _p(b), C.c_uint32(stride), _p(corres), C.c_uint32(len(corres)), C.byref(p), _p(T), C.byref(valid), _p(clique), C.byref(nq), C.byref(scale))
    if st != QN_OK:
        raise EngineError(st, lib().qn_status_str(st).decode())
    return dict(T=T, valid=bool(valid.value), clique=clique[:nq.value].copy(), scale=scale.value)


def coarse_to_fine_alignment(ctx, src, dst, *, quatro=None, k=15, max_iter=32, max_corr_dist=52.5, trans_eps=0.01, score_thr=1.5):
    """LoopClosure::coarseToFineAlignment (loop_closure.cpp:138-159) at the reference's effective config."""
    quatro = quatro or Quatro(ctx)
    a, ns, stride = _cloud_arg(src); b, nt, _ = _cloud_arg(dst)
    res = GicpResult(); valid = C.c_int(); T = np.zeros((4, 4)); Tq = np.zeros((4, 4))
    with _ParamsScope(ctx):
        _reference_gicp(ctx, k, max_iter, max_corr_dist, trans_eps)
        st = ctx._l.qn_coarse_to_fine_alignment(ctx.h, _p(a), C.c_uint32(ns), _p(b), C.c_uint32(nt), C.c_uint32(stride), C.c_double(score_thr),
                                                C.byref(res), _p(T), _p(Tq), C.byref(valid))
    if st == QN_ERR_EMPTY_CLOUD:
        return dict(valid=False, converged=False, score=1.7976931348623157e308, T=np.eye(4), T_quatro=np.eye(4))
    ctx.check(st)
    return dict(valid=bool(valid.value), converged=bool(res.converged), score=res.fitness, T=T, T_quatro=Tq, iterations=res.iterations,
                T_gicp=np.array(res.T, dtype=np.float32).reshape(4, 4).astype(np.float64))


def coarse_to_fine_alignment_device(ctx, src_ptr, ns, dst_ptr, nt, stride, *, quatro=None, k=15, max_iter=32, max_corr_dist=52.5, trans_eps=0.01, score_thr=1.5):
    """qn_coarse_to_fine_alignment_device: the same with both clouds resident on the GPU (e.g. KeyframeStore.assemble outputs)."""
    quatro = quatro or Quatro(ctx)
    res = GicpResult(); valid = C.c_int(); T = np.zeros((4, 4)); Tq = np.zeros((4, 4))
    with _ParamsScope(ctx):
        _reference_gicp(ctx, k, max_iter, max_corr_dist, trans_eps)
        ctx.check(ctx._l.qn_coarse_to_fine_alignment_device(ctx.h, C.c_void_p(src_ptr), C.c_uint32(ns), C.c_void_p(dst_ptr), C.c_uint32(nt), C.c_uint32(stride),
                                                           C.c_double(score_thr), C.byref(res), _p(T), _p(Tq), C.byref(valid)))
    return dict(valid=bool(valid.value), converged=bool(res.converged), score=res.fitness, T=T, T_quatro=Tq, iterations=res.iterations,
                T_gicp=np.array(res.T, dtype=np.float32).reshape(4, 4).astype(np.float64))


# ---------------------------------------------------------------------------------------- batch
class PairDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("ns", C.c_uint32), ("dst", C.c_void_p), ("nt", C.c_uint32), ("stride_bytes", C.c_uint32), ("on_device", C.c_int32)]


def icp_alignment_batch(contexts, pairs, score_thr=1.5):
    """pairs: list of (src_ptr_or_array, ns, dst_ptr_or_array, nt, stride_bytes, on_device).  Arrays are host float32
    clouds; ints are device pointers.  Every context must already carry its NanoGICP parameters.
    Returns (results[GicpResult], valid[int], status[int])."""
    n = len(pairs)
    descs = (PairDesc * n)(); keep = []
    for i, (s, ns, d, nt, stride, dev) in enumerate(pairs):
        if not dev:
            s = np.ascontiguousarray(s, dtype=np.float32); d = np.ascontiguousarray(d, dtype=np.float32); keep += [s, d]
            descs[i] = PairDesc(s.ctypes.data, ns, d.ctypes.data, nt, stride, 0)
        else:
            descs[i] = PairDesc(s, ns, d, nt, stride, 1)
    results = (GicpResult * n)(); valid = (C.c_int * n)(); status = (C.c_int * n)()
    hs = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
    st = lib().qn_icp_alignment_batch(hs, C.c_uint32(len(contexts)), descs, C.c_uint32(n), C.c_double(score_thr), results, valid, status)
    if st != QN_OK:
        raise EngineError(st, lib().qn_status_str(st).decode())
    return results, list(valid), list(status)


def gicp_align_batch(ctx, pairs, score_thr=1.5):
    """qn_gicp_align_batch: the same batch on ONE context, the pair as a grid dimension of every kernel launch (`batch_lanes` pairs in lockstep).
    pairs / return value as icp_alignment_batch."""
    n = len(pairs)
    descs = (PairDesc * n)(); keep = []
    for i, (s, ns, d, nt, stride, dev) in enumerate(pairs):
        if not dev:
            s = np.ascontiguousarray(s, dtype=np.float32); d = np.ascontiguousarray(d, dtype=np.float32); keep += [s, d]
            descs[i] = PairDesc(s.ctypes.data, ns, d.ctypes.data, nt, stride, 0)
        else:
            descs[i] = PairDesc(s, ns, d, nt, stride, 1)
    results = (GicpResult * n)(); valid = (C.c_int * n)(); status = (C.c_int * n)()
    st = lib().qn_gicp_align_batch(ctx.h, descs, C.c_uint32(n), C.c_double(score_thr), results, valid, status)
    if st != QN_OK:
        raise EngineError(st, lib().qn_status_str(st).decode() + ": " + lib().qn_last_error(ctx.h).decode())
    return results, list(valid), list(status)


def coarse_to_fine_align_batch(contexts, pairs, score_thr=1.5):
    """qn_coarse_to_fine_align_batch: n independent coarseToFineAlignment calls (loop_closure.cpp:138-159) over the contexts' lanes.  pairs as icp_alignment_batch; every
    context must carry its NanoGICP and Quatro parameters.  -> list of dict(valid, converged, score, iterations, T (= T_gicp * T_quatro), T_quatro, T_gicp (f32 record as f64), status)"""
    n = len(pairs)
    descs = (PairDesc * max(n, 1))(); keep = []
    for i, (s, ns, d, nt, stride, dev) in enumerate(pairs):
        if not dev:
            s = np.ascontiguousarray(s, dtype=np.float32); d = np.ascontiguousarray(d, dtype=np.float32); keep += [s, d]
            descs[i] = PairDesc(s.ctypes.data, ns, d.ctypes.data, nt, stride, 0)
        else:
            descs[i] = PairDesc(s, ns, d, nt, stride, 1)
    results = (GicpResult * max(n, 1))(); valid = (C.c_int * max(n, 1))(); status = (C.c_int * max(n, 1))()
    Tt = np.zeros((max(n, 1), 4, 4)); Tq = np.zeros((max(n, 1), 4, 4))
    hs = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
    st = lib().qn_coarse_to_fine_align_batch(hs, C.c_uint32(len(contexts)), descs, C.c_uint32(n), C.c_double(score_thr), results, _p(Tt), _p(Tq), valid, status)
    if st != QN_OK:
        raise EngineError(st, lib().qn_status_str(st).decode())
    return [dict(valid=bool(valid[i]), converged=bool(results[i].converged), score=results[i].fitness, iterations=results[i].iterations, T=Tt[i].copy(), T_quatro=Tq[i].copy(),
                 T_gicp=np.array(results[i].T, dtype=np.float32).reshape(4, 4).astype(np.float64), status=int(status[i])) for i in range(n)]


def lane_trace(ctx, lane):
    """qn_gicp_get_lane_trace: the iteration trace (y0, lambda, rho, max_dR, max_dt, inner, accepted) of lane `lane` of the latest qn_gicp_align_batch run"""
    buf = (IterTrace * 1024)(); n = C.c_uint32()
    ctx.check(lib().qn_gicp_get_lane_trace(ctx.h, C.c_uint32(lane), buf, C.c_uint32(1024), C.byref(n)))
    return np.array([[t.y0, t.lambda_, t.rho, t.max_dR, t.max_dt, t.inner, t.accepted] for t in buf[:n.value]]).reshape(-1, 7)


class PairRecord(C.Structure):
    _fields_ = [("pair_id", C.c_int32), ("status", C.c_int32), ("valid", C.c_int32), ("converged", C.c_int32), ("iterations", C.c_int32),
                ("reserved", C.c_int32), ("fitness", C.c_double), ("T", C.c_float * 16)]


class MultiGpu:
    """qn_multi_*: candidate pairs sharded pair i -> GPU i mod N inside one process, one RCCL all-gather of the result records."""

    def __init__(self, n_gpus, max_points, in_flight=4, device_ids=None):
        self._l = lib(); h = C.c_void_p()
        self._l.qn_multi_last_error.restype = C.c_char_p; self._l.qn_multi_last_error.argtypes = [C.c_void_p]
        self._l.qn_multi_destroy.argtypes = [C.c_void_p]
        ids = None if device_ids is None else (C.c_int * n_gpus)(*device_ids)
        st = self._l.qn_multi_init(C.c_int(n_gpus), ids, C.c_uint32(max_points), C.c_int(in_flight), C.byref(h))
        if st != QN_OK:
            raise EngineError(st, self._l.qn_status_str(st).decode() + ": " + self._l.qn_multi_last_error(None).decode())
        self.h = h; self.n_gpus = n_gpus

    def close(self):
        if getattr(self, "h", None):
            self._l.qn_multi_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != QN_OK:
            raise EngineError(st, self._l.qn_status_str(st).decode() + ": " + self._l.qn_multi_last_error(self.h).decode())

    def set_params(self, p):
        self._check(self._l.qn_multi_set_params(self.h, C.byref(p)))

    def debug_set(self, key, value):
        self._check(self._l.qn_multi_debug_set(self.h, key.encode(), C.c_double(value)))

    def set_quatro_params(self, qp):
        """enable_quatro_ (loop_closure.h:54): a QuatroParams = every pair becomes a coarseToFineAlignment; None = Nano-GICP only"""
        self._check(self._l.qn_multi_set_quatro_params(self.h, C.byref(qp) if qp is not None else None))

    def timing(self):
        """(per-GPU ms [n_gpus], gather ms) of the latest align_best"""
        per = (C.c_double * self.n_gpus)(); gms = C.c_double()
        self._check(self._l.qn_multi_get_timing(self.h, per, C.byref(gms)))
        return list(per), gms.value

    def gpu_count(self):
        return int(self._l.qn_multi_gpu_count(self.h))

    def rccl_ranks(self):
        """ranks of the communicator as RCCL reports them (ncclCommCount), -1 on failure"""
        return int(self._l.qn_multi_rccl_ranks(self.h))

    def verify_gather(self):
        """after align_best: every GPU's receive buffer holds the table GPU 0 received"""
        self._check(self._l.qn_multi_verify_gather(self.h))

    def align_best(self, pairs, score_thr=1.5):
        """pairs as for icp_alignment_batch.  -> (records[n], best or None)"""
        n = len(pairs)
        descs = (PairDesc * max(n, 1))(); keep = []
        for i, (s, ns, d, nt, stride, dev) in enumerate(pairs):
            if not dev:
                s = np.ascontiguousarray(s, dtype=np.float32); d = np.ascontiguousarray(d, dtype=np.float32); keep += [s, d]
                descs[i] = PairDesc(s.ctypes.data, ns, d.ctypes.data, nt, stride, 0)
            else:
                descs[i] = PairDesc(s, ns, d, nt, stride, 1)
        recs = (PairRecord * max(n, 1))(); best = PairRecord(); found = C.c_int()
        self._check(self._l.qn_multi_align_best(self.h, descs, C.c_uint32(n), C.c_double(score_thr), recs, C.byref(best), C.byref(found)))
        return list(recs[:n]), (best if found.value else None)


# ---------------------------------------------------------------------------------------- keyframe store / cloud assembly
class KeyframeStore:
    """Device-resident keyframe clouds + LoopClosure::setSrcAndDstCloud on the GPU (loop_closure.cpp:58-108)."""

    def __init__(self, device=0):
        self._l = lib(); h = C.c_void_p()
        st = self._l.qn_kf_store_create(C.c_int(device), C.byref(h))
        if st != QN_OK:
            raise EngineError(st, self._l.qn_status_str(st).decode())
        self.h = h
        self._l.qn_kf_last_error.restype = C.c_char_p; self._l.qn_kf_last_error.argtypes = [C.c_void_p]
        self._l.qn_kf_store_destroy.argtypes = [C.c_void_p]

    def close(self):
        if getattr(self, "h", None):
            self._l.qn_kf_store_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != QN_OK:
            raise EngineError(st, self._l.qn_status_str(st).decode() + ": " + self._l.qn_kf_last_error(self.h).decode())

    def add(self, xyz):
        a, n, stride = _cloud_arg(xyz); kid = C.c_int32()
        self._check(self._l.qn_kf_add(self.h, _p(a), C.c_uint32(n), C.c_uint32(stride), C.byref(kid)))
        return kid.value

    def assemble(self, ids, poses, leaf, slot):
        """-> (device pointer of float4 points, count)"""
        ids = np.ascontiguousarray(ids, dtype=np.int32); poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(len(ids), 16)
        ptr = C.c_void_p(); n = C.c_uint32()
        self._check(self._l.qn_kf_assemble(self.h, _p(ids), _p(poses), C.c_uint32(len(ids)), C.c_double(leaf), C.c_int(slot), C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def download(self, slot, n):
        out = np.zeros((n, 3), np.float32)
        self._check(self._l.qn_kf_download(self.h, C.c_int(slot), _p(out)))
        return out


def loop_candidates(pos, stamps, query, radius, tdiff, max_k=64):
    pos = np.ascontiguousarray(pos, dtype=np.float64); stamps = np.ascontiguousarray(stamps, dtype=np.float64)
    out = np.zeros(max_k, np.int32); n = C.c_uint32()
    st = lib().qn_loop_candidates(_p(pos), _p(stamps), C.c_uint32(len(pos)), C.c_uint32(query), C.c_double(radius), C.c_double(tdiff), C.c_uint32(max_k), _p(out), C.byref(n))
    if st != QN_OK:
        raise EngineError(st, lib().qn_status_str(st).decode())
    return out[:n.value].copy()
