#!/usr/bin/env python
"""Replay harness (SURVEY.md 8f rank 4 / BASELINE configs[4]): the loop-closure side of FastLioSamQn on a synthetic
keyframe stream, with the registration engine on the GPU and the pose graph on the host.

What is reproduced from the reference (fast_lio_sam_qn/src/fast_lio_sam_qn.cpp):
  * keyframes every `keyframe_thr` metres of odometry (FQ:124, config.yaml:7), each with its sensor-frame cloud (PosePcd, PP:7-19);
  * prior + odometry BetweenFactors with the reference's variances (FQ:112-116, 132-143);
  * loopTimerFunc (FQ:203-252): candidate = closest keyframe within `loop_detection_radius` and older than
    `loop_detection_timediff_threshold` (LC:34-56) -> setSrcAndDstCloud (LC:58-108) -> coarseToFineAlignment /
    icpAlignment (LC:110-159) -> if valid: BetweenFactor(latest, closest, (T_reg * pose_latest).between(pose_closest)), variance = score
    on all 6 dof (FQ:220-238) -> re-optimise, rewrite all corrected poses (FQ:180-188).
What is NOT the reference: GTSAM/iSAM2 is not installed here, so the pose graph is a small batch SE(3) Gauss-Newton in numpy
(same factors, same noise models) - the optimiser stays on the host either way, as the north-star prescribes.
The registration engine is the product under test: keyframe clouds resident in HBM (qn_kf_store), cloud assembly + voxel grid
on the device, Nano-GICP and (--quatro) Quatro + Nano-GICP on the device through the `_device` entry points: a loop attempt moves no
point cloud across PCIe.  `backend="oracle"` runs the same loop with the CPU oracle's assembly and registrations instead (test
infrastructure: tests/test_replay.py compares the two); `save_dir` writes the corrected trajectory the way saveFlagCallback does
(FQ:344-373): poses_kitti.txt (3x4 row-major, default stream precision) and poses_tum.txt ("#timestamp x y z qx qy qz qw", 8 decimals).
The reference's extra iSAM2::update() calls after a loop (FQ:160-165) are iSAM2 relinearisation sweeps; the batch Gauss-Newton stand-in
iterates to convergence instead.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import ctypes as C
import numpy as np


# ------------------------------------------------------------------ SE(3) helpers (host pose graph)
def hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])


def exp_se3(xi):
    w, v = xi[:3], xi[3:]
    th = np.linalg.norm(w)
    W = hat(w)
    if th < 1e-9:
        R = np.eye(3) + W; V = np.eye(3) + 0.5 * W
    else:
        R = np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th ** 2 * W @ W
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * W + (th - np.sin(th)) / th ** 3 * W @ W
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = V @ v
    return T


def log_se3(T):
    R, t = T[:3, :3], T[:3, 3]
    c = np.clip((np.trace(R) - 1) / 2, -1, 1); th = np.arccos(c)
    if th < 1e-9:
        w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / 2
        Vi = np.eye(3) - 0.5 * hat(w)
    else:
        w = th / (2 * np.sin(th)) * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
        W = hat(w)
        Vi = np.eye(3) - 0.5 * W + (1 / th ** 2 - (1 + np.cos(th)) / (2 * th * np.sin(th))) * W @ W
    return np.r_[w, Vi @ t]


def inv(T):
    Ti = np.eye(4); Ti[:3, :3] = T[:3, :3].T; Ti[:3, 3] = -T[:3, :3].T @ T[:3, 3]
    return Ti


class PoseGraph:
    """prior + between factors, Gauss-Newton with right perturbations X <- X exp(dx); numeric Jacobians (graphs here are small)."""

    def __init__(self):
        self.poses = []; self.factors = []           # (i, j, Z, sqrt_info[6]) ; j = -1 for a prior on i

    def add_pose(self, T):
        self.poses.append(T.copy()); return len(self.poses) - 1

    def add_prior(self, i, Z, var):
        self.factors.append((i, -1, Z.copy(), 1.0 / np.sqrt(var)))

    def add_between(self, i, j, Z, var):
        self.factors.append((i, j, Z.copy(), 1.0 / np.sqrt(var)))

    def _res(self, f, Xi, Xj):
        i, j, Z, s = f
        E = inv(Z) @ (Xi if j < 0 else inv(Xi) @ Xj)
        return s * log_se3(E)

    def optimize(self, iters=25):
        n = len(self.poses)
        for _ in range(iters):
            H = np.zeros((6 * n, 6 * n)); g = np.zeros(6 * n)
            for f in self.factors:
                i, j = f[0], f[1]
                Xi = self.poses[i]; Xj = self.poses[j] if j >= 0 else None
                r0 = self._res(f, Xi, Xj)
                eps = 1e-6; Ji = np.zeros((6, 6)); Jj = np.zeros((6, 6))
                for k in range(6):
                    d = np.zeros(6); d[k] = eps
                    Ji[:, k] = (self._res(f, Xi @ exp_se3(d), Xj) - r0) / eps
                    if j >= 0:
                        Jj[:, k] = (self._res(f, Xi, Xj @ exp_se3(d)) - r0) / eps
                si = slice(6 * i, 6 * i + 6)
                H[si, si] += Ji.T @ Ji; g[si] += Ji.T @ r0
                if j >= 0:
                    sj = slice(6 * j, 6 * j + 6)
                    H[sj, sj] += Jj.T @ Jj; g[sj] += Jj.T @ r0; H[si, sj] += Ji.T @ Jj; H[sj, si] += Jj.T @ Ji
            dx = np.linalg.solve(H + 1e-9 * np.eye(6 * n), -g)
            for i in range(n):
                self.poses[i] = self.poses[i] @ exp_se3(dx[6 * i:6 * i + 6])
            if np.abs(dx).max() < 1e-6:
                break


# ------------------------------------------------------------------ synthetic keyframe stream
def make_stream(n_kf, seed, scan_range=28.0, drift_yaw=0.004, drift_xy=0.03, pts_per_scan=9000, yaw_bias=0.006):
    from qn_amd import synth
    rng = np.random.default_rng(seed)
    scene = synth.Scene(rng, 120.0)
    world = scene.sample(rng, 700000, (-60, 60, -60, 60))
    s = np.linspace(0, 2 * np.pi, n_kf, endpoint=False)           # figure-8: passes the centre twice -> loops
    xy = np.c_[30 * np.sin(s), 22 * np.sin(2 * s)]
    head = np.arctan2(np.gradient(xy[:, 1]), np.gradient(xy[:, 0]))
    gt = []
    for k in range(n_kf):
        T = np.eye(4); T[:3, :3] = synth._rot_zyx(head[k], 0, 0); T[:3, 3] = [xy[k, 0], xy[k, 1], 1.8]
        gt.append(T)
    scans = []
    for k in range(n_kf):
        d = np.linalg.norm(world[:, :2] - xy[k], axis=1)
        sel = np.flatnonzero(d < scan_range)
        sel = rng.choice(sel, min(len(sel), pts_per_scan * 3), replace=False)
        p = world[sel] + rng.normal(0, 0.02, (len(sel), 3))
        local = (p - gt[k][:3, 3]) @ gt[k][:3, :3]                  # sensor frame (PosePcd::pcd_, PP:39)
        local = synth.voxel_centroids(local, 0.2)
        if len(local) > pts_per_scan:
            local = local[np.sort(rng.choice(len(local), pts_per_scan, replace=False))]
        scans.append(local.astype(np.float32))
    odom = [gt[0].copy()]
    for k in range(1, n_kf):
        rel = inv(gt[k - 1]) @ gt[k]
        noise = exp_se3(np.r_[0, 0, rng.normal(0, drift_yaw) + yaw_bias, rng.normal(0, drift_xy, 2), 0])   # biased yaw drift
        odom.append(odom[-1] @ rel @ noise)
    return scans, gt, odom, np.arange(n_kf) * 1.0


def rot_to_quat(R):
    """(qx, qy, qz, qw) of a rotation matrix (what tf::Matrix3x3::getRotation yields in poseEigToPoseStamped, utilities.hpp)"""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2; q = [(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2; q = [0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s, (R[2, 1] - R[1, 2]) / s]
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2; q = [(R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s, (R[0, 2] - R[2, 0]) / s]
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2; q = [(R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s, (R[1, 0] - R[0, 1]) / s]
    return np.array(q)


def write_kitti_tum(save_dir, poses, stamps):
    """saveFlagCallback's pose files (fast_lio_sam_qn.cpp:344-373): KITTI = the 3x4 [R|t] row by row with the stream's default
    formatting (6 significant digits), TUM = header line + `stamp x y z qx qy qz qw`, fixed, 8 decimals."""
    os.makedirs(save_dir, exist_ok=True)
    with open(os.path.join(save_dir, "poses_kitti.txt"), "w") as fk, open(os.path.join(save_dir, "poses_tum.txt"), "w") as ft:
        ft.write("#timestamp x y z qx qy qz qw\n")
        for T, st in zip(poses, stamps):
            fk.write(" ".join("%g" % T[r, c] for r in range(3) for c in range(4)) + "\n")
            q = rot_to_quat(T[:3, :3])
            ft.write("%.8f %.8f %.8f %.8f %.8f %.8f %.8f %.8f\n" % (st, T[0, 3], T[1, 3], T[2, 3], q[0], q[1], q[2], q[3]))


def ate(poses, gt):
    return float(np.sqrt(np.mean([np.sum((a[:3, 3] - b[:3, 3]) ** 2) for a, b in zip(poses, gt)])))


def run(n_kf=70, seed=7, use_quatro=False, radius=12.0, tdiff=15.0, voxel=0.3, submap_range=5, score_thr=1.5, verbose=True, backend="gpu", save_dir=None):
    scans, gt, odom, stamps = make_stream(n_kf, seed)
    if backend == "gpu":
        from qn_amd import engine
        store = engine.KeyframeStore()
        ctx = engine.Context(400000)
        g = engine.NanoGICP(ctx)                                    # LoopClosure ctor, loop_closure.cpp:9-16, SURVEY App. C values
        g.setCorrespondenceRandomness(15); g.setMaximumIterations(32); g.setMaxCorrespondenceDistance(1.5 * radius); g.setTransformationEpsilon(0.01)
        quatro = engine.Quatro(ctx) if use_quatro else None
        loop_candidates = engine.loop_candidates
    else:
        from oracle import oracle as orc                           # the checker's side of the comparison (tests only)
        loop_candidates = orc.loop_candidates
    pg = PoseGraph(); ids = []; corrected = []
    prior_var = np.array([1e-4, 1e-4, 1e-4, 1e-2, 1e-2, 1e-2]); odom_var = prior_var.copy()   # FQ:112-114, 132-133 (rot, then trans)
    loops = []; t_reg = []
    for k in range(n_kf):
        if backend == "gpu":
            ids.append(store.add(scans[k]))
        pose = odom[k] if k == 0 else corrected[-1] @ (inv(odom[k - 1]) @ odom[k])           # realtime pose = last corrected * delta odom (FQ:93-103)
        pg.add_pose(pose); corrected.append(pose)
        if k == 0:
            pg.add_prior(0, pose, prior_var)
        else:
            pg.add_between(k - 1, k, inv(odom[k - 1]) @ odom[k], odom_var)
        # ---- loopTimerFunc
        pos = np.array([c[:3, 3] for c in corrected])
        cand = loop_candidates(pos, stamps[:k + 1], k, radius, tdiff, max_k=1)
        if len(cand) == 0:
            continue
        c = int(cand[0])
        t0 = time.perf_counter()
        if use_quatro:
            sub = [c]                                                                        # LC:89-92: scan to scan
        else:
            lo, hi = max(0, c - submap_range), min(k - 1, c + submap_range)                  # LC:98-104: scan to submap
            sub = list(range(lo, hi + 1))
        if backend == "gpu":
            ps, ns = store.assemble([ids[k]], [corrected[k]], voxel, 0)
            pd, nd = store.assemble([ids[i] for i in sub], [corrected[i] for i in sub], voxel, 1)
            if use_quatro:                                                                   # both clouds stay on the device (qn_coarse_to_fine_alignment_device)
                r = engine.coarse_to_fine_alignment_device(ctx, ps, ns, pd, nd, 16, quatro=quatro, max_corr_dist=1.5 * radius, score_thr=score_thr)
                valid, score, Treg = r["valid"], r["score"], r["T"]
            else:
                res = engine.GicpResult(); v = C.c_int()
                ctx.check(ctx._l.qn_icp_alignment_device(ctx.h, C.c_void_p(ps), C.c_uint32(ns), C.c_void_p(pd), C.c_uint32(nd), C.c_uint32(16),
                                                         C.c_double(score_thr), C.byref(res), C.byref(v)))
                valid, score, Treg = bool(v.value), res.fitness, np.array(res.T, dtype=np.float64).reshape(4, 4)
        else:
            src = orc.assemble_submap(scans, corrected, [k], voxel); dst = orc.assemble_submap(scans, corrected, sub, voxel)
            if use_quatro:
                r = orc.coarse_to_fine_alignment(src, dst, max_corr_dist=1.5 * radius, score_thr=score_thr)
            else:
                r = orc.icp_alignment(src, dst, max_corr_dist=1.5 * radius, score_thr=score_thr)
            valid, score, Treg = r["valid"], r["score"], r["T"]
        t_reg.append(time.perf_counter() - t0)
        if not valid:
            continue
        pose_from = Treg @ corrected[k]; pose_to = corrected[c]                              # FQ:224-225
        pg.add_between(k, c, inv(pose_from) @ pose_to, np.full(6, max(score, 1e-6)))        # FQ:226-233
        loops.append((k, c, score))
        pg.optimize()
        corrected = [p.copy() for p in pg.poses]                                             # FQ:180-188
    out = dict(n_keyframes=n_kf, loops=len(loops), attempts=len(t_reg), ate_odometry=ate(odom, gt), ate_corrected=ate(corrected, gt),
               ms_per_attempt=1e3 * float(np.mean(t_reg)) if t_reg else None, quatro=use_quatro, loop_list=loops, poses=corrected)
    if save_dir:
        write_kitti_tum(save_dir, corrected, stamps)
    if verbose:
        print({k: v for k, v in out.items() if k not in ("poses", "loop_list")})
    if backend == "gpu":
        ctx.close(); store.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--keyframes", type=int, default=70)
    ap.add_argument("--quatro", action="store_true")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--save-dir", default=None, help="write poses_kitti.txt / poses_tum.txt (FQ:344-373) here")
    a = ap.parse_args()
    run(a.keyframes, a.seed, a.quatro, save_dir=a.save_dir)
