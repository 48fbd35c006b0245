"""The feeder rows: candidate search on the host (CPU test), and keyframe-resident cloud
assembly - transformPcd + concatenation + pcl::VoxelGrid - on the GPU, bit-exact against the oracle, then the whole
loop attempt (assemble on the device -> icpAlignment on the device) against the oracle pipeline."""
import numpy as np
import pytest
from qn_amd import synth


def _trajectory(n=40, seed=0):
    rng = np.random.default_rng(seed)
    t = np.linspace(0, 2 * np.pi, n)
    pos = np.c_[25 * np.sin(t), 18 * np.sin(2 * t), 0.1 * rng.normal(size=n)]      # figure-8: revisits the centre
    stamps = np.arange(n) * 2.5
    return pos, stamps


def test_loop_candidates_match_reference_rule(oracle):
    from qn_amd import engine
    pos, stamps = _trajectory()
    for q in (len(pos) - 1, 25, 12):
        got = engine.loop_candidates(pos[:q + 1], stamps[:q + 1], q, 35.0, 30.0)
        exp = oracle.loop_candidates(pos[:q + 1], stamps[:q + 1], q, 35.0, 30.0)
        assert np.array_equal(got, exp)
        # restatement of fetchClosestKeyframeIdx (loop_closure.cpp:34-56): the single closest admissible keyframe
        best, bd = -1, 35.0 * 3.0
        for i in range(q):
            d = np.linalg.norm(pos[i] - pos[q])
            if 35.0 > d and 30.0 < stamps[q] - stamps[i] and d < bd:
                bd, best = d, i
        assert (got[0] if len(got) else -1) == best


def _keyframes(nkf=7, npts=6000, seed=3):
    rng = np.random.default_rng(seed)
    world, _, _ = synth.make_pair(seed, 30000, extent=50.0, leaf=0.1)
    kfs, poses = [], []
    for k in range(nkf):
        T = synth._rot_zyx(0.05 * k, 0.01 * rng.normal(), 0.01 * rng.normal()); P = np.eye(4); P[:3, :3] = T; P[:3, 3] = [1.5 * k - 4, 0.3 * k, 0.02 * k]
        sel = rng.choice(len(world), npts, replace=False)
        w = world[sel].astype(np.float64) + rng.normal(0, 0.01, (npts, 3))
        kfs.append(((w - P[:3, 3]) @ P[:3, :3]).astype(np.float32))                   # sensor frame: P^-1 * world
        poses.append(P)
    return kfs, poses


@pytest.mark.gpu
def test_assemble_and_voxelize_bit_exact(oracle):
    from qn_amd import engine
    kfs, poses = _keyframes()
    store = engine.KeyframeStore()
    ids = [store.add(k) for k in kfs]
    for sel in ([3], [0, 1, 2, 3, 4, 5, 6], [2, 3, 4]):
        ptr, n = store.assemble([ids[i] for i in sel], [poses[i] for i in sel], 0.3, 1)
        got = store.download(1, n)
        exp = oracle.assemble_submap(kfs, poses, sel, 0.3)
        assert n == len(exp)
        assert np.array_equal(got, exp)                      # same leaves, same order, same f32 centroids
    store.close()


@pytest.mark.gpu
def test_assemble_drops_non_finite_points_like_pcl_voxelgrid(oracle):
    """pcl::VoxelGrid (include/utilities.hpp:38-51) on a non-dense cloud skips non-finite points - in the bounds and in the leaves - and carries on;
    so does qn_kf_assemble: a keyframe with NaN / inf points gives exactly the submap of the same keyframe without them."""
    from qn_amd import engine
    kfs, poses = _keyframes()
    rng = np.random.default_rng(3)
    dirty = []
    for k in kfs:
        d = k.copy(); bad = rng.choice(len(d), 37, replace=False)
        d[bad[:20], rng.integers(0, 3, 20)] = np.nan; d[bad[20:], 0] = np.inf
        dirty.append((d, np.setdiff1d(np.arange(len(d)), bad)))
    store = engine.KeyframeStore()
    ids_dirty = [store.add(d) for d, _ in dirty]
    sel = [1, 2, 3]
    ptr, n = store.assemble([ids_dirty[i] for i in sel], [poses[i] for i in sel], 0.3, 0)
    got = store.download(0, n)
    clean = [d[keep] for d, keep in dirty]
    exp = oracle.assemble_submap(clean, poses, sel, 0.3)
    assert n == len(exp) and np.array_equal(got, exp)
    only_nan = np.full((5, 3), np.nan, np.float32)
    kid = store.add(only_nan)
    with pytest.raises(engine.EngineError) as ei:
        store.assemble([kid], [np.eye(4)], 0.3, 0)
    assert ei.value.status == engine.QN_ERR_EMPTY_CLOUD
    store.close()


@pytest.mark.gpu
def test_loop_attempt_on_device_matches_oracle_pipeline(oracle):
    """setSrcAndDstCloud (scan-to-submap branch, loop_closure.cpp:94-105) + icpAlignment with nothing leaving the GPU."""
    from qn_amd import engine
    import ctypes as C
    kfs, poses = _keyframes()
    drift = np.eye(4); drift[:3, :3] = synth._rot_zyx(0.03, 0, 0); drift[:3, 3] = [0.4, -0.3, 0.05]
    src_pose = drift @ poses[6]                                  # the query keyframe's drifted pose
    store = engine.KeyframeStore(); ids = [store.add(k) for k in kfs]
    ps, ns = store.assemble([ids[6]], [src_pose], 0.3, 0)
    pd, nd = store.assemble(ids[0:5], poses[0:5], 0.3, 1)
    ctx = engine.Context(max(ns, nd) + 1024)
    g = engine.NanoGICP(ctx); g.setCorrespondenceRandomness(15); g.setMaximumIterations(32); g.setMaxCorrespondenceDistance(52.5); g.setTransformationEpsilon(0.01)
    res = engine.GicpResult(); valid = C.c_int()
    ctx.check(ctx._l.qn_icp_alignment_device(ctx.h, C.c_void_p(ps), C.c_uint32(ns), C.c_void_p(pd), C.c_uint32(nd), C.c_uint32(16), C.c_double(1.5), C.byref(res), C.byref(valid)))
    src_o = oracle.assemble_submap(kfs, {6: src_pose}, [6], 0.3)
    dst_o = oracle.assemble_submap(kfs, poses, [0, 1, 2, 3, 4], 0.3)
    ro = oracle.icp_alignment(src_o, dst_o)
    assert bool(valid.value) == ro["valid"] and bool(res.converged) == ro["converged"] and res.iterations == ro["iterations"]
    dt, dr = synth.pose_error(np.array(res.T, dtype=np.float64).reshape(4, 4), ro["T"])
    assert dt <= 1e-4 and dr <= 1e-4
    assert abs(res.fitness - ro["score"]) <= 1e-6 * ro["score"]
    ctx.close(); store.close()
