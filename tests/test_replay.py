"""Replay harness: synthetic keyframe stream -> candidate search -> device-side cloud assembly ->
registration on the GPU -> loop factors into a host pose graph.  Loop closures must be found and must pull the drifted
trajectory back towards ground truth."""
import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_pose_graph_stand_in_closes_a_loop():
    import numpy as np
    import replay
    scans, gt, odom, st = replay.make_stream(24, 3, pts_per_scan=50)
    pg = replay.PoseGraph(); var = np.array([1e-4] * 3 + [1e-2] * 3)
    for k, T in enumerate(odom):
        pg.add_pose(T)
        if k == 0:
            pg.add_prior(0, T, var)
        else:
            pg.add_between(k - 1, k, replay.inv(odom[k - 1]) @ odom[k], var)
    pg.add_between(23, 0, replay.inv(gt[23]) @ gt[0], np.full(6, 1e-4))
    pg.optimize()
    assert replay.ate(pg.poses, gt) < 0.5 * replay.ate(odom, gt)


@pytest.mark.gpu
@pytest.mark.parametrize("use_quatro", [False, True])
def test_replay_finds_loops_and_reduces_drift(use_quatro):
    import replay
    out = replay.run(n_kf=60, seed=7, use_quatro=use_quatro, verbose=False)
    assert out["attempts"] >= 5 and out["loops"] >= 2, out
    assert out["ate_corrected"] < 0.7 * out["ate_odometry"], out


def test_kitti_tum_writers(tmp_path):
    """saveFlagCallback's file formats (fast_lio_sam_qn.cpp:344-373)"""
    import numpy as np
    import replay
    T = np.eye(4); T[:3, :3] = [[0, -1, 0], [1, 0, 0], [0, 0, 1]]; T[:3, 3] = [1.5, -2.25, 0.125]
    replay.write_kitti_tum(str(tmp_path), [np.eye(4), T], [0.0, 1.5])
    k = open(tmp_path / "poses_kitti.txt").read().splitlines()
    assert k[0] == "1 0 0 0 0 1 0 0 0 0 1 0" and k[1] == "0 -1 0 1.5 1 0 0 -2.25 0 0 1 0.125"
    t = open(tmp_path / "poses_tum.txt").read().splitlines()
    assert t[0] == "#timestamp x y z qx qy qz qw"
    v = [float(x) for x in t[2].split()]
    assert v[:4] == [1.5, 1.5, -2.25, 0.125] and np.allclose(v[4:], [0, 0, np.sqrt(0.5), np.sqrt(0.5)], atol=1e-8)
    assert all(len(x.split(".")[1]) == 8 for x in t[2].split())


@pytest.mark.gpu
@pytest.mark.parametrize("use_quatro", [False, True])
def test_replay_matches_the_oracle_run(use_quatro, tmp_path):
    """the same keyframe stream through the engine and through the CPU oracle: same loop attempts accepted, same corrected trajectory"""
    import numpy as np
    import replay
    a = replay.run(n_kf=40, seed=11, use_quatro=use_quatro, verbose=False, save_dir=str(tmp_path))
    b = replay.run(n_kf=40, seed=11, use_quatro=use_quatro, verbose=False, backend="oracle")
    assert [(k, c) for k, c, _ in a["loop_list"]] == [(k, c) for k, c, _ in b["loop_list"]] and a["attempts"] == b["attempts"]
    for (_, _, sa), (_, _, sb) in zip(a["loop_list"], b["loop_list"]):
        assert abs(sa - sb) <= 1e-5 * max(sb, 1e-9)
    d = max(np.linalg.norm(p[:3, 3] - q[:3, 3]) for p, q in zip(a["poses"], b["poses"]))
    assert d < 1e-3, d
    assert (tmp_path / "poses_kitti.txt").exists() and len(open(tmp_path / "poses_tum.txt").read().splitlines()) == 41
