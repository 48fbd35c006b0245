"""Generate tests/golden/*.npz - small known-answer fixtures for the registration path.

PROVENANCE (read this): the reference's own Nano-GICP / Quatro sources are un-vendored submodules (empty directories under
/root/reference/third_party, .gitmodules:7-12) and the reference ships no tests or golden vectors (SURVEY.md section 4,
8c), so these fixtures can NOT come from the reference: parity stays "unpinned".  What they pin instead:
  * gicp_*.npz   - produced by oracle/py_oracle.py, the INDEPENDENT numpy/scipy restatement (cKDTree, numpy.linalg), never
                   by the C++ oracle or the HIP path that are checked against them;
  * cov_plane.npz, so3.npz - closed-form answers (C = I - 0.999 n n^T on an exact plane; Rodrigues), no code under test involved;
  * voxel.npz    - pcl::VoxelGrid semantics on a hand-checkable cloud, produced by a 20-line pure-numpy restatement below;
  * quatro_*.npz - the Quatro coarse stage (FPFH, optimizedMatching / advancedMatching, TEASER++ solve with the yaw-only GNC)
                   produced by oracle/py_quatro.py, the INDEPENDENT numpy/scipy restatement (cKDTree radius / 33-D searches,
                   numpy.linalg.eigh / svd, libm arctan2) - different code and numerics from oracle/quatro_oracle.cpp and from
                   the HIP kernels that are both checked against these files.
Inputs are stored in the files (not regenerated), so the fixtures stay valid if the synthetic generator changes.

    python tests/golden/make_golden.py          # rewrites the .npz files next to this script
"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
from qn_amd import synth
from oracle import py_oracle, py_quatro


def gicp_case(name, pair_id, n, extent, **kw):
    src, tgt, T_gt = synth.make_pair(pair_id, n, extent=extent)
    p = py_oracle.PyGicp(src, tgt, **kw)
    r = p.align()
    cov_s, knn_s = py_oracle.covariances(src, kw["k"])
    H, b, e = p.linearize(np.eye(4))
    np.savez_compressed(os.path.join(HERE, name), src=src, tgt=tgt, T_gt=T_gt, T=r["T"], iterations=r["iterations"],
                        converged=int(r["converged"]), fitness=r["fitness"], cov_src=cov_s[:64], H0=H, b0=b, e0=e,
                        k=kw["k"], max_iter=kw["max_iter"], max_corr_dist=kw["max_corr_dist"], trans_eps=kw["trans_eps"],
                        optimizer=kw["optimizer"])
    print(name, "iters", r["iterations"], "conv", r["converged"], "fitness %.6f" % r["fitness"])


def quatro_case(name, pair_id, n, extent, n_tgt=None):
    src, tgt, T_gt = synth.make_pair(pair_id, n, n_tgt, extent=extent, mode="quatro")
    p = py_quatro.Params()
    r = py_quatro.align(src, tgt, p)
    pa = py_quatro.Params(use_optimized_matching=False)
    mut_a, cor_a = py_quatro.matching(src, tgt, r["fpfh"][0], r["fpfh"][1], pa)
    ra = py_quatro.solve(src, tgt, cor_a, pa)
    np.savez_compressed(os.path.join(HERE, name), src=src, tgt=tgt, T_gt=T_gt,
                        normals_s=r["normals"][0], normals_t=r["normals"][1], spfh_s=r["spfh"][0], fpfh_s=r["fpfh"][0], fpfh_t=r["fpfh"][1],
                        mutual=r["mutual"].astype(np.int32), corres=r["corres"].astype(np.int32), clique=np.array(r["clique"], np.int32),
                        T=r["T"], valid=int(r["valid"]), rot_iterations=r["rot_iterations"],
                        mutual_adv=mut_a.astype(np.int32), corres_adv=cor_a.astype(np.int32), clique_adv=np.array(ra["clique"], np.int32),
                        T_adv=ra["T"], valid_adv=int(ra["valid"]), rot_iterations_adv=ra["rot_iterations"],
                        params=np.array([p.fpfh_normal_radius, p.fpfh_radius, p.noise_bound, p.rot_gnc_factor, p.rot_cost_diff_thr, p.rot_max_iter,
                                         p.distance_threshold, p.max_num_corres, p.rng_seed, p.tuple_scale]))
    print(name, "mutual", len(r["mutual"]), "corres", len(r["corres"]), "clique", len(r["clique"]), "valid", r["valid"], "err", synth.pose_error(r["T"], T_gt),
          "| advanced corres", len(cor_a), "clique", len(ra["clique"]), "err", synth.pose_error(ra["T"], T_gt))


def voxel_grid_numpy(xyz, leaf):
    """pcl::VoxelGrid (utilities.hpp:38-51 calls it with leaf = voxel_res): leaf index = floor(p / leaf) - floor(min / leaf),
    linear index ix + iy*dx + iz*dx*dy, output = centroid per occupied leaf in ascending linear index, f32 accumulation in
    point order (PCL sums Eigen::Vector4f centroids)."""
    inv = np.float32(1.0) / np.float32(leaf)
    mn = np.floor(xyz.min(0) * inv).astype(np.int64); mx = np.floor(xyz.max(0) * inv).astype(np.int64)
    d = mx - mn + 1
    ijk = np.floor(xyz * inv).astype(np.int64) - mn
    lin = ijk[:, 0] + ijk[:, 1] * d[0] + ijk[:, 2] * d[0] * d[1]
    order = np.argsort(lin, kind="stable")
    out = []
    s = 0
    while s < len(order):
        e = s
        acc = np.zeros(3, np.float32)
        while e < len(order) and lin[order[e]] == lin[order[s]]:
            acc = (acc + xyz[order[e]]).astype(np.float32); e += 1
        out.append(acc / np.float32(e - s)); s = e
    return np.array(out, np.float32)


def main():
    gicp_case("gicp_lm_k15.npz", 101, 1500, 30.0, k=15, max_iter=32, max_corr_dist=52.5, trans_eps=0.01, optimizer="lm")   # reference operating point
    gicp_case("gicp_gn_k20.npz", 102, 1200, 30.0, k=20, max_iter=32, max_corr_dist=52.5, trans_eps=0.01, optimizer="gn")
    quatro_case("quatro_a.npz", 401, 3000, 30.0)                      # equal sizes
    quatro_case("quatro_b.npz", 402, 2600, 34.0, n_tgt=3100)          # target larger than source: the matcher's swapped branch
    # exact plane: every covariance is I - 0.999 n n^T
    rng = np.random.default_rng(5)
    n = np.array([0.3, -0.2, 0.933]); n /= np.linalg.norm(n)
    u = np.cross(n, [1, 0, 0]); u /= np.linalg.norm(u); v = np.cross(n, u)
    ab = rng.uniform(-5, 5, size=(400, 2))
    pts = (ab[:, :1] * u + ab[:, 1:] * v + 2.0 * n).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "cov_plane.npz"), pts=pts, normal=n, C=np.eye(3) - 0.999 * np.outer(n, n), k=12)
    # Rodrigues
    om = rng.uniform(-1, 1, size=(16, 3)) * rng.uniform(1e-6, 3.0, size=(16, 1))
    Rs = []
    for w in om:
        th = np.linalg.norm(w); K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / th
        Rs.append(np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K)
    np.savez_compressed(os.path.join(HERE, "so3.npz"), omega=om, R=np.array(Rs))
    # voxel grid
    cloud = rng.uniform(-3, 3, size=(3000, 3)).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "voxel.npz"), cloud=cloud, leaf=np.float32(0.3), out=voxel_grid_numpy(cloud, 0.3))


if __name__ == "__main__":
    main()
