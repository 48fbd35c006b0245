"""Deterministic synthetic scan-pair generator (SURVEY.md §8d, row "G").

No KITTI data exists in the container, so every config of BASELINE.json is run on a
seeded "street scene": ground plane, building walls, poles and boxes, sampled on
surfaces, voxel-snapped to centroids at 0.3 m (what the reference's `voxelizePcd`
does before registration: fast_lio_sam_qn/include/utilities.hpp:38-51, called at
src/loop_closure.cpp:107) and resampled to exactly N points.

  source = scene sampled in window W_s                      (world frame)
  target = T_gt * (scene sampled independently in W_t + N(0, sigma))

so a registration of source onto target should recover T_gt.  seed = 20241220 + pair_id.
Pure numpy (PCG64) - identical output here and on the GPU box (same image).
"""
import numpy as np

BASE_SEED = 20241220


def _rot_zyx(yaw, pitch, roll):
    cy, sy = np.cos(yaw), np.sin(yaw)
    cp, sp = np.cos(pitch), np.sin(pitch)
    cr, sr = np.cos(roll), np.sin(roll)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return Rz @ Ry @ Rx


class Scene:
    """Piecewise-planar/cylindrical world; `sample(rng, m, window)` draws m surface points."""

    def __init__(self, rng, extent=120.0):
        self.extent = float(extent)
        h = extent / 2.0
        s = extent / 120.0  # object sizes scale with the scene so small test scenes stay busy
        nwalls, npoles, nboxes = 10, 40, 30
        # walls: (x0, y0, dx, dy, height) vertical rectangles, axis aligned
        self.walls = []
        for i in range(nwalls):
            L = rng.uniform(25, 45) * s
            H = rng.uniform(8, 16) * min(1.0, max(s, 0.35))
            cx, cy = rng.uniform(-h * 0.85, h * 0.85, 2)
            if i % 2 == 0:
                self.walls.append((cx - L / 2, cy, L, 0.0, H))
            else:
                self.walls.append((cx, cy - L / 2, 0.0, L, H))
        self.poles = [(rng.uniform(-h * 0.9, h * 0.9), rng.uniform(-h * 0.9, h * 0.9),
                       0.15, 6.0 * min(1.0, max(s, 0.35))) for _ in range(npoles)]
        self.boxes = []
        for _ in range(nboxes):
            cx, cy = rng.uniform(-h * 0.9, h * 0.9, 2)
            sx, sy = (2.0, 4.0) if rng.random() < 0.5 else (4.0, 2.0)
            self.boxes.append((cx, cy, sx * max(s, 0.5), sy * max(s, 0.5), 1.5))

    def sample(self, rng, m, window):
        """window = (xmin, xmax, ymin, ymax): points outside are rejected (re-drawn)."""
        out = []
        need = m
        while need > 0:
            p = self._draw(rng, int(need * 1.6) + 64)
            keep = (p[:, 0] >= window[0]) & (p[:, 0] <= window[1]) & \
                   (p[:, 1] >= window[2]) & (p[:, 1] <= window[3])
            p = p[keep][:need]
            out.append(p)
            need -= len(p)
        return np.concatenate(out, 0)

    def _draw(self, rng, m):
        h = self.extent / 2.0
        kind = rng.random(m)
        pts = np.empty((m, 3))
        # ground 40 %
        g = kind < 0.40
        ng = int(g.sum())
        pts[g] = np.stack([rng.uniform(-h, h, ng), rng.uniform(-h, h, ng), np.zeros(ng)], 1)
        # walls 35 %
        w = (kind >= 0.40) & (kind < 0.75)
        nw = int(w.sum())
        wi = rng.integers(0, len(self.walls), nw)
        W = np.array(self.walls)[wi]
        u = rng.random(nw)
        pts[w] = np.stack([W[:, 0] + u * W[:, 2], W[:, 1] + u * W[:, 3], rng.random(nw) * W[:, 4]], 1)
        # poles 5 %
        p = (kind >= 0.75) & (kind < 0.80)
        npz = int(p.sum())
        pi = rng.integers(0, len(self.poles), npz)
        P = np.array(self.poles)[pi]
        a = rng.uniform(0, 2 * np.pi, npz)
        pts[p] = np.stack([P[:, 0] + P[:, 2] * np.cos(a), P[:, 1] + P[:, 2] * np.sin(a),
                           rng.random(npz) * P[:, 3]], 1)
        # boxes 20 % (4 sides + top, area weighted roughly)
        b = kind >= 0.80
        nb = int(b.sum())
        bi = rng.integers(0, len(self.boxes), nb)
        B = np.array(self.boxes)[bi]
        face = rng.integers(0, 5, nb)
        u, v = rng.random(nb), rng.random(nb)
        x = np.where(face == 0, B[:, 0] - B[:, 2] / 2,
            np.where(face == 1, B[:, 0] + B[:, 2] / 2, B[:, 0] + (u - 0.5) * B[:, 2]))
        y = np.where(face == 2, B[:, 1] - B[:, 3] / 2,
            np.where(face == 3, B[:, 1] + B[:, 3] / 2,
            np.where(face == 4, B[:, 1] + (v - 0.5) * B[:, 3], B[:, 1] + (u - 0.5) * B[:, 3])))
        # for faces 0/1 x is fixed and y varies with u
        y = np.where((face == 0) | (face == 1), B[:, 1] + (u - 0.5) * B[:, 3], y)
        z = np.where(face == 4, B[:, 4], v * B[:, 4])
        pts[b] = np.stack([x, y, z], 1)
        return pts


def voxel_centroids(p, leaf):
    """pcl::VoxelGrid semantics: one centroid per occupied leaf, output ordered by leaf index."""
    q = np.floor(p / leaf).astype(np.int64)
    q -= q.min(0)
    dims = q.max(0) + 1
    key = q[:, 0] + dims[0] * (q[:, 1] + dims[1] * q[:, 2])
    order = np.argsort(key, kind="stable")
    key_s = key[order]
    first = np.flatnonzero(np.r_[True, key_s[1:] != key_s[:-1]])
    cnt = np.diff(np.r_[first, len(key_s)])
    sums = np.add.reduceat(p[order], first, axis=0)
    return sums / cnt[:, None]


def random_gt(rng, mode="gicp"):
    """T_gt distribution of SURVEY §8d: small for GICP-only pairs, large yaw/xy for Quatro pairs."""
    d = np.pi / 180.0
    if mode == "gicp":
        yaw = rng.uniform(-10, 10) * d
        pitch, roll = rng.uniform(-1, 1, 2) * d
        t = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.5, 1.5), rng.uniform(-0.2, 0.2)])
    elif mode == "quatro":
        yaw = rng.uniform(-180, 180) * d
        pitch, roll = rng.uniform(-2, 2, 2) * d
        t = np.array([rng.uniform(-10, 10), rng.uniform(-10, 10), rng.uniform(-0.2, 0.2)])
    elif mode == "identity":
        yaw = pitch = roll = 0.0
        t = np.zeros(3)
    else:
        raise ValueError(mode)
    T = np.eye(4)
    T[:3, :3] = _rot_zyx(yaw, pitch, roll)
    T[:3, 3] = t
    return T


def make_pair(pair_id, n_src, n_tgt=None, *, extent=120.0, leaf=0.3, sigma=0.02,
              shift=None, mode="gicp"):
    """Returns (src[n_src,3] f32, tgt[n_tgt,3] f32, T_gt[4,4] f64).

    `shift` [m]: the target's scene window is the source's shifted along x (default extent/24,
    i.e. 5 m on the 120 m scene: ~96 % overlap, which keeps the PCL fitness score - the mean
    squared NN distance over ALL source points - under the reference's 1.5 m^2 accept
    threshold, loop_closure.cpp:129 + config.yaml:21, for a correct registration)."""
    n_tgt = n_src if n_tgt is None else n_tgt
    rng = np.random.Generator(np.random.PCG64(BASE_SEED + int(pair_id)))
    scene = Scene(rng, extent)
    T = random_gt(rng, mode)
    h = extent / 2.0
    shift = extent / 24.0 if shift is None else float(shift)
    win_s = (-h, h - shift, -h, h)
    win_t = (-h + shift, h, -h, h)

    def cloud(n, win, transform, noise):
        over = 3.0
        for _ in range(8):
            raw = scene.sample(rng, int(n * over) + 256, win)
            if noise > 0:
                raw = raw + rng.normal(0.0, noise, raw.shape)
            if transform is not None:
                raw = raw @ transform[:3, :3].T + transform[:3, 3]
            c = voxel_centroids(raw, leaf)
            if len(c) >= n:
                sel = rng.permutation(len(c))[:n]
                sel.sort()
                return c[sel].astype(np.float32)
            over *= 1.8
        raise RuntimeError("scene too small for %d points at leaf %.2f" % (n, leaf))

    src = cloud(n_src, win_s, None, 0.0)
    tgt = cloud(n_tgt, win_t, T, sigma)
    return src, tgt, T


def lever_arm_pair(seed, n=3000, offset=300.0, rot_sigma=0.1, scene=True, noise=0.05):
    """A small pair 300 m from the origin with a wrong initial rotation: the lever arm makes the cost strongly non-quadratic in the
    rotation, so Levenberg-Marquardt REJECTS trial steps (inner tries 2..9) - pairs near the origin never do.  scene = True: a street-scene
    pair (surface points: well-conditioned plane normals); False: uniform points in a cube (near-isotropic neighbourhoods: the plane normal
    of A.1.3 is ill-conditioned there, see SURVEY App. B-4 - for probing, not for strict parity).  -> (src, tgt, guess[4,4] f64)"""
    rng = np.random.Generator(np.random.PCG64(77000 + int(seed)))
    if scene:
        src, tgt, _ = make_pair(600 + int(seed), n, extent=30.0)
        off = np.array([offset, 0.0, 0.0])
        src = (src.astype(np.float64) + off).astype(np.float32); tgt = (tgt.astype(np.float64) + off).astype(np.float32)
    w = rng.normal(0, rot_sigma, 3); th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K
    if not scene:
        src = (rng.uniform(-3, 3, (n, 3)) + [offset, 0, 0]).astype(np.float32)
        tgt = (src + rng.normal(0, noise, (n, 3))).astype(np.float32)
    guess = np.eye(4); guess[:3, :3] = R; guess[:3, 3] = rng.uniform(-1, 1, 3)
    return src, tgt, guess


def pose_error(T_a, T_b):
    """(translation error [m], rotation error [rad]) between two 4x4 transforms."""
    D = np.linalg.inv(T_a) @ T_b
    c = (np.trace(D[:3, :3]) - 1.0) / 2.0
    return float(np.linalg.norm(D[:3, 3])), float(np.arccos(np.clip(c, -1.0, 1.0)))
