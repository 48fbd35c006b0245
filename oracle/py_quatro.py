"""Second, independent restatement of the Quatro coarse stage in numpy/scipy.

ORACLE - TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (third_party/Quatro is an empty submodule; nothing of the
reference's arithmetic exists to pin against - this file pins the C++ oracle and the HIP path to a second reading of
the same published algorithms: PCL NormalEstimation / FPFHEstimation (Rusu 2009), the FGR / TEASER++ matcher with
Quatro's optimizedMatching and TEASER++'s advancedMatching, TEASER++'s solver with Quatro's yaw-only GNC-TLS rotation;
call site fast_lio_sam_qn/src/loop_closure.cpp:144, parameters :18-27).

Deliberately DIFFERENT code and numerics from oracle/quatro_oracle.cpp:
  * radius search: scipy cKDTree.query_ball_point on f64 coordinates (d <= r) instead of a hash grid with f32 d2 < r2;
  * normals: numpy.linalg.eigh on the mean-centred covariance instead of cyclic Jacobi;
  * pair features: numpy f32 arithmetic with numpy.arctan2 (libm) instead of the fixed-order scalar code and the Cephes
    polynomial qn_atan2f - histogram-bin flips between the two are exactly the divergences the polynomial (error
    <= 5e-7 rad, tests/test_quatro_cpu.py) or a last-bit f32 normal can cause; the tests COUNT them;
  * feature matching: cKDTree in 33-D (f64 distances) instead of the f32 sequential brute force;
  * rotation: weighted 2-D Procrustes by numpy.linalg.svd instead of the closed-form atan2;
  * translation: direct evaluation of every consensus set instead of TEASER++'s running sums;
  * maximum clique: integer-bitset branch and bound written for this file.
Shared by specification (both restatements and the GPU path): the seeded LCG that replaces rand() in the tuple test,
the lexicographically smallest maximum clique, and the operation order of the f32 distance gate.
"""
import numpy as np
from scipy.spatial import cKDTree


class Params:
    def __init__(self, fpfh_normal_radius=0.9, fpfh_radius=1.5, noise_bound=0.3, rot_gnc_factor=1.4, rot_cost_diff_thr=1e-4,
                 rot_max_iter=50, use_optimized_matching=True, distance_threshold=35.0, max_num_corres=200, rng_seed=1,
                 tuple_scale=0.95, estimate_scale=False):
        self.__dict__.update(locals()); del self.__dict__["self"]


# ------------------------------------------------------------------ FPFH
def normals(xyz, radius):
    p = xyz.astype(np.float64)
    nb = cKDTree(p).query_ball_point(p, radius)
    out = np.full((len(p), 3), np.nan, np.float32)
    for i, idx in enumerate(nb):
        if len(idx) < 3:
            continue
        q = p[idx]
        c = np.cov(q.T, bias=True)
        _, V = np.linalg.eigh(c)
        n = V[:, 0]                                   # eigenvector of the smallest eigenvalue
        if -(n @ p[i]) < 0:                           # flipNormalTowardsViewpoint, viewpoint = (0, 0, 0)
            n = -n
        out[i] = n.astype(np.float32)
    return out


def _pair_features(p1, n1, P2, N2):
    """pcl::computePairFeatures of one source point against an array of neighbours, f32."""
    f = np.float32
    dp = (P2 - p1).astype(f)
    f4 = np.sqrt((dp * dp).sum(1, dtype=f))
    ok = f4 > 0
    f4s = np.where(ok, f4, f(1))
    a1 = (dp * n1).sum(1, dtype=f) / f4s
    a2 = (dp * N2).sum(1, dtype=f) / f4s
    swap = np.abs(a1) < np.abs(a2)
    A = np.where(swap[:, None], N2, n1[None, :]).astype(f)
    B = np.where(swap[:, None], n1[None, :], N2).astype(f)
    d = np.where(swap[:, None], -dp, dp).astype(f)
    f3 = np.where(swap, -a2, a1).astype(f)
    v = np.cross(d, A).astype(f)
    vn = np.sqrt((v * v).sum(1, dtype=f))
    ok &= vn > 0
    v = (v / np.where(vn > 0, vn, f(1))[:, None]).astype(f)
    w = np.cross(A, v).astype(f)
    f2 = (v * B).sum(1, dtype=f)
    f1 = np.arctan2((w * B).sum(1, dtype=f), (A * B).sum(1, dtype=f)).astype(f)
    return f1, f2, f3, ok


def fpfh(xyz, normal_radius, fpfh_radius):
    """-> normals (n,3) f32, spfh (n,33) f32, fpfh (n,33) f32 (NaN rows where PCL has no descriptor)."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    nrm = normals(xyz, normal_radius)
    p = xyz.astype(np.float64)
    nb = cKDTree(p).query_ball_point(p, fpfh_radius)
    n = len(p)
    fin = np.isfinite(nrm).all(1)
    spfh = np.zeros((n, 33), np.float32)
    d_pi = np.float32(1.0) / (np.float32(2.0) * np.float32(np.pi))
    for i in range(n):
        if not fin[i]:
            continue
        idx = np.array(sorted(nb[i]))
        use = idx[(idx != i) & fin[idx]]
        if len(use) == 0:
            continue
        f1, f2, f3, ok = _pair_features(xyz[i], nrm[i], xyz[use], nrm[use])
        h1 = np.clip(np.floor(11.0 * ((f1.astype(np.float64) + np.pi) * float(d_pi))), 0, 10).astype(int)[ok]
        h2 = np.clip(np.floor(11.0 * ((f2.astype(np.float64) + 1.0) * 0.5)), 0, 10).astype(int)[ok]
        h3 = np.clip(np.floor(11.0 * ((f3.astype(np.float64) + 1.0) * 0.5)), 0, 10).astype(int)[ok]
        incr = np.float32(100.0) / np.float32(len(idx) - 1)
        cnt = np.zeros(33, np.int64)
        np.add.at(cnt, h1, 1); np.add.at(cnt, 11 + h2, 1); np.add.at(cnt, 22 + h3, 1)
        spfh[i] = cnt.astype(np.float32) * incr
    out = np.full((n, 33), np.nan, np.float32)
    for i in range(n):
        if not fin[i]:
            continue
        idx = np.array(nb[i])
        d2 = ((xyz[idx] - xyz[i]).astype(np.float32) ** 2).sum(1, dtype=np.float32)
        keep = d2 > 0
        if not keep.any():
            continue
        w = (np.float32(1.0) / d2[keep]).astype(np.float32)
        acc = (spfh[idx[keep]] * w[:, None]).astype(np.float32).astype(np.float64).sum(0)
        s = acc.reshape(3, 11).sum(1)
        if s[0] == 0.0:
            continue
        scale = np.where(s != 0, 100.0 / np.where(s != 0, s, 1.0), 0.0)
        out[i] = (acc.reshape(3, 11) * scale[:, None]).reshape(33).astype(np.float32)
    return nrm, spfh, out


# ------------------------------------------------------------------ matching
def _lcg(state):
    state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
    return state, state >> 8


def feature_nn(q, c):
    """exact NN of every finite query row among the finite candidate rows, -1 otherwise (f64 distances; equal distances -
    identical descriptors are common on planar ground - resolve to the lowest candidate index, as the specification says)."""
    fq = np.isfinite(q).all(1); fc = np.isfinite(c).all(1)
    nn = np.full(len(q), -1, np.int64)
    if fq.any() and fc.any():
        ci = np.flatnonzero(fc)
        C64 = c[fc].astype(np.float64); Q64 = q[fq].astype(np.float64)
        tree = cKDTree(C64)
        d, j = tree.query(Q64, k=1)
        ties = tree.query_ball_point(Q64, d * (1 + 1e-9) + 1e-12)
        for t, (cand, dj) in enumerate(zip(ties, d)):
            cand = np.array(sorted(cand))
            dd = ((C64[cand] - Q64[t]) ** 2).sum(1)
            j[t] = cand[np.flatnonzero(dd == dd.min())[0]]
        nn[fq] = ci[j]
    return nn


def _mutual(Fi, Fj):
    """Matcher initial matching + cross-check: j -> nearest i, reverse search only for the hit i (lazy), keep mutual pairs."""
    j_to_i = feature_nn(Fj, Fi)
    hit = np.unique(j_to_i[j_to_i >= 0])
    i_to_j = np.full(len(Fi), -1, np.int64)
    if len(hit):
        i_to_j[hit] = feature_nn(Fi[hit], Fj)
    return [(int(j_to_i[j]), j) for j in range(len(Fj)) if j_to_i[j] >= 0 and i_to_j[j_to_i[j]] == j]


def _tuple_test(Pi, Pj, cand, p, cap):
    f = np.float32
    ncorr = len(cand)
    if ncorr < 3:
        return []
    scale = f(p.tuple_scale); rng = int(p.rng_seed) & 0xFFFFFFFF
    dist = lambda a, b: np.sqrt(f(f(f((a[0] - b[0]) * (a[0] - b[0])) + f((a[1] - b[1]) * (a[1] - b[1]))) + f((a[2] - b[2]) * (a[2] - b[2]))))
    tup = []
    for _ in range(ncorr * 100):
        r = []
        for _k in range(3):
            rng, v = _lcg(rng); r.append(v % ncorr)
        a = [Pi[cand[k][0]] for k in r]; b = [Pj[cand[k][1]] for k in r]
        li = [dist(a[0], a[1]), dist(a[1], a[2]), dist(a[2], a[0])]
        lj = [dist(b[0], b[1]), dist(b[1], b[2]), dist(b[2], b[0])]
        if all((f(li[k] * scale) < lj[k]) and (lj[k] < f(li[k] / scale)) for k in range(3)):
            tup += [cand[k] for k in r]
        if cap is not None and len(tup) > cap:
            break
    return tup


def matching(src, dst, fs, ft, p):
    """Matcher::calculateCorrespondences -> optimizedMatching (use_optimized_matching) or advancedMatching.
    Returns (mutual pairs before the tuple test, final correspondences), both sorted unique (src idx, dst idx)."""
    src = np.ascontiguousarray(src, np.float32); dst = np.ascontiguousarray(dst, np.float32)
    swapped = len(dst) > len(src)                       # fi = the larger cloud
    Pi, Pj, Fi, Fj = (dst, src, ft, fs) if swapped else (src, dst, fs, ft)
    mi = Pi.astype(np.float64).mean(0).astype(np.float32); mj = Pj.astype(np.float64).mean(0).astype(np.float32)
    Ni = (Pi - mi).astype(np.float32); Nj = (Pj - mj).astype(np.float32)      # normalizePoints, absolute scale
    cand = _mutual(Fi, Fj)
    if p.use_optimized_matching:                        # distance gate of optimizedMatching
        f = np.float32
        def far(i, j):
            d = Ni[i] - Nj[j]
            return np.sqrt(f(f(d[0] * d[0]) + f(d[1] * d[1])) + f(d[2] * d[2])) > f(p.distance_threshold)
        cand = [c for c in cand if not far(*c)]
    un = lambda L: sorted(set((j, i) if swapped else (i, j) for i, j in L))
    tup = _tuple_test(Ni, Nj, cand, p, p.max_num_corres if p.use_optimized_matching else None)
    return np.array(un(cand), np.int64).reshape(-1, 2), np.array(un(tup), np.int64).reshape(-1, 2)


# ------------------------------------------------------------------ solver
def max_clique_lex(adj):
    """lexicographically smallest maximum clique; adj = (n, n) 0/1 symmetric, zero diagonal."""
    n = len(adj)
    N = [sum(1 << j for j in range(n) if adj[i][j] and i != j) for i in range(n)]

    def omega(P, stop):
        best = 0

        def expand(P, size):
            nonlocal best
            if best >= stop:
                return
            if P == 0:
                best = max(best, size); return
            order, colour, U, c = [], [], P, 0
            while U:
                c += 1; Q = U
                while Q:
                    v = (Q & -Q).bit_length() - 1
                    Q &= ~(1 << v); U &= ~(1 << v); Q &= ~N[v]
                    order.append(v); colour.append(c)
            for t in range(len(order) - 1, -1, -1):
                if size + colour[t] <= best:
                    return
                v = order[t]
                expand(P & N[v], size + 1)
                if best >= stop:
                    return
                P &= ~(1 << v)
        expand(P, 0)
        return best

    allv = (1 << n) - 1
    w = omega(allv, 1 << 30)
    chosen, cand = [], allv
    for v in range(n):
        if len(chosen) >= w:
            break
        if not (cand >> v) & 1:
            continue
        rest = cand & N[v] & ~((1 << (v + 1)) - 1)
        need = w - len(chosen) - 1
        if need == 0 or omega(rest, need) >= need:
            chosen.append(v); cand = rest
    return chosen


def _tls_1d(X, alpha):
    """TEASER++ scalar TLS: over the consensus sets swept by the sorted interval ends, minimise
    sum_in (x - mean_in)^2 + alpha * #out; first minimum in sweep order."""
    N = len(X)
    # ties: an interval that starts at a value enters before one that ends there leaves
    ev = sorted([(X[i] - alpha, 0, i) for i in range(N)] + [(X[i] + alpha, 1, i) for i in range(N)], key=lambda e: (e[0], e[1]))
    inside = set(); best = None
    for v, kind, i in ev:
        if kind == 0:
            inside.add(i)
        else:
            inside.discard(i)
        if not inside:
            continue
        xs = np.array([X[k] for k in sorted(inside)])
        xh = xs.mean()
        cost = ((xs - xh) ** 2).sum() + alpha * (N - len(xs))
        if best is None or cost < best[0] - 0.0:
            best = (cost, xh)
    return best[1]


def _tls_ranges(X, R):
    """TEASER++ scalar TLS with one range per measurement (the scale stage), as prefix sums over the sorted interval ends: for every sweep position the consensus set's
    weighted mean (weights 1 / range^2) and cost = sum_in (x - xh)^2 + sum_out range; the first minimum in sweep order wins.  -> (estimate, inlier mask)"""
    X = np.asarray(X, np.float64); R = np.asarray(R, np.float64); N = len(X)
    val = np.concatenate([X - R, X + R]); kind = np.concatenate([np.zeros(N, np.int64), np.ones(N, np.int64)]); idx = np.concatenate([np.arange(N), np.arange(N)])
    order = np.lexsort((np.where(kind == 0, -idx, idx), kind, val))      # by value; at equal values the C++ sort's "tag descending": starts (tag i + 1) before ends (tag -i - 1), starts by
    # descending index, ends by ascending index
    sgn = np.where(kind[order] == 0, 1.0, -1.0); ii = idx[order]
    w = 1.0 / (R[ii] ** 2)
    card = np.cumsum(sgn); sw = np.cumsum(sgn * w); sxw = np.cumsum(sgn * w * X[ii]); sx = np.cumsum(sgn * X[ii]); sxx = np.cumsum(sgn * X[ii] ** 2)
    out_pen = R.sum() - np.cumsum(sgn * R[ii])
    with np.errstate(divide="ignore", invalid="ignore"):
        xh = sxw / sw
        cost = (card * xh * xh + sxx - 2 * sx * xh) + out_pen
    cost = np.where(np.isnan(cost), np.inf, cost)
    k = int(np.argmin(cost))                                   # first minimum
    est = float(xh[k])
    return est, np.abs(X - est) <= R


def solve(src, dst, corres, p):
    out = dict(T=np.eye(4), valid=False, clique=[], rot_iterations=0, scale=1.0)
    M = len(corres)
    if M == 0:
        return out
    S = src[corres[:, 0]].astype(np.float64); D = dst[corres[:, 1]].astype(np.float64)
    beta = 2.0 * p.noise_bound
    da = np.linalg.norm(S[None, :, :] - S[:, None, :], axis=2); db = np.linalg.norm(D[None, :, :] - D[:, None, :], axis=2)
    scale = 1.0
    if getattr(p, "estimate_scale", False):                    # TLSScaleSolver over all TIMs; its inliers are the edges
        iu, ju = np.triu_indices(M, 1)
        ok = da[iu, ju] > 0
        iu, ju = iu[ok], ju[ok]
        if len(iu) == 0:
            return out
        scale, inl = _tls_ranges(db[iu, ju] / da[iu, ju], beta / da[iu, ju])
        if not scale > 0:
            return out
        adj = np.zeros((M, M), np.uint8); adj[iu[inl], ju[inl]] = 1; adj[ju[inl], iu[inl]] = 1
        out["scale"] = scale
    else:
        adj = (np.abs(db - da) <= beta).astype(np.uint8); np.fill_diagonal(adj, 0)
    C = max_clique_lex(adj)
    out["clique"] = C
    m = len(C)
    if m <= 1:
        return out
    nxt = C[1:] + C[:1]
    A = S[nxt] - S[C]; B = (D[nxt] - D[C]) / scale            # chain TIMs (dst TIMs with the scale removed)
    nb2 = (2.0 * p.noise_bound / scale) ** 2                  # TEASER++ rescales the rotation solver's bound by 2 / scale
    if nb2 < 1e-16:
        nb2 = 1e-2
    w = np.ones(m); mu = 1.0; prev = np.inf; R2 = np.eye(2)
    for it in range(p.rot_max_iter):
        out["rot_iterations"] = it + 1
        H = (A[:, :2] * w[:, None]).T @ B[:, :2]              # sum w a b^T
        U, _, Vt = np.linalg.svd(H)
        d = np.sign(np.linalg.det(Vt.T @ U.T)) or 1.0
        R2 = Vt.T @ np.diag([1.0, d]) @ U.T
        R = np.eye(3); R[:2, :2] = R2
        res = ((B - A @ R.T) ** 2).sum(1)
        if it == 0:
            mu = 1.0 / (2.0 * res.max() / nb2 - 1.0)
            if mu <= 0:
                break
        th1, th2 = (mu + 1) / mu * nb2, mu / (mu + 1) * nb2
        cost = float((w * res).sum())
        with np.errstate(divide="ignore", invalid="ignore"):
            wn = np.sqrt(nb2 * mu * (mu + 1) / res) - mu
        w = np.where(res >= th1, 0.0, np.where(res <= th2, 1.0, wn))
        diff = abs(cost - prev); mu *= p.rot_gnc_factor; prev = cost
        if diff < p.rot_cost_diff_thr:
            break
    R = np.eye(3); R[:2, :2] = R2
    X = D[C] - scale * (S[C] @ R.T)
    t = np.array([_tls_1d(list(X[:, k]), p.noise_bound) for k in range(3)])
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    out["T"] = T; out["valid"] = True
    return out


def align(src, dst, p=None):
    """quatro<PointType>::align (loop_closure.cpp:144) with every intermediate product."""
    p = p or Params()
    src = np.ascontiguousarray(src, np.float32); dst = np.ascontiguousarray(dst, np.float32)
    n1, s1, f1 = fpfh(src, p.fpfh_normal_radius, p.fpfh_radius)
    n2, s2, f2 = fpfh(dst, p.fpfh_normal_radius, p.fpfh_radius)
    mutual, corres = matching(src, dst, f1, f2, p)
    r = solve(src, dst, corres, p)
    r.update(normals=(n1, n2), spfh=(s1, s2), fpfh=(f1, f2), mutual=mutual, corres=corres)
    return r
