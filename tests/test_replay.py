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
