"""Adversarial inputs for the tracked / bound-pruned nearest-neighbour passes (k_nn_track, qn_gicp_kernels.cuh): every tracked pass of
an align() is compared, ALL queries, with a fresh unseeded search of the same pose (debug knob "verify_track": indices and, where the
pass stores them, the f32 squared distances bit for bit).  The pruning inequality  d(q, p_j0) + |q - q_ref| < d_other  carries a
rounding margin (track_bound_holds); these cases sit where it could break:
  * lattice targets with the source half a cell off: 2-4 exactly or nearly equidistant neighbours per query, ties decided by index;
  * the same 8 km from the origin, where one f32 ulp is ~0.5 mm and coordinates carry almost no fraction bits;
  * a noisy street scene 8 km out (irregular distances, far neighbours, list passes)."""
import numpy as np
import pytest
from qn_amd import synth

pytestmark = pytest.mark.gpu


def lattice_scene(step=0.3, half=18.0, wall_h=6.0):
    g = np.arange(-half, half + 1e-6, step)
    gx, gy = np.meshgrid(g, g)
    ground = np.c_[gx.ravel(), gy.ravel(), np.zeros(gx.size)]
    z = np.arange(step, wall_h, step)
    wy, wz = np.meshgrid(g, z)
    wall_a = np.c_[np.full(wy.size, -half * 0.5), wy.ravel(), wz.ravel()]           # x = const
    wall_b = np.c_[wy.ravel(), np.full(wy.size, half * 0.4), wz.ravel()]            # y = const
    return np.concatenate([ground, wall_a, wall_b], 0)


def rigid(yaw, t):
    c, s = np.cos(yaw), np.sin(yaw)
    T = np.eye(4); T[:3, :3] = [[c, -s, 0], [s, c, 0], [0, 0, 1]]; T[:3, 3] = t
    return T


def run_verified(engine, ctx, src, tgt, optimizer, force, track_from=None, k=20):
    g = engine.NanoGICP(ctx)
    g.setCorrespondenceRandomness(k); g.setMaximumIterations(32); g.setMaxCorrespondenceDistance(52.5); g.setTransformationEpsilon(1e-5)
    g.setRotationEpsilon(1e-6); g.setOptimizer(optimizer); g.setForceIterations(force)
    ctx.debug_set("verify_track", 1)
    if track_from is not None:
        ctx.debug_set("track_from_tick", track_from)
    g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
    r = g.align()
    bad, passes, first = ctx.debug_get("verify_mismatches"), ctx.debug_get("verify_passes"), ctx.debug_get("verify_first")
    ctx.debug_set("verify_track", 0); ctx.debug_set("track_from_tick", 3)
    return r, int(bad), int(passes), int(first), g


@pytest.fixture(scope="module")
def eng():
    from qn_amd import engine
    ctx = engine.Context(60000)
    yield engine, ctx
    ctx.close()


@pytest.mark.parametrize("offset", [0.0, 8000.0])
@pytest.mark.parametrize("optimizer,force", [("gn", 20), ("lm", 0)])
def test_lattice_half_cell_offset(eng, offset, optimizer, force):
    engine, ctx = eng
    P = lattice_scene()
    off = np.array([offset, -offset * 0.75, 0.0])
    tgt = (P + off).astype(np.float32)
    T = rigid(0.004, [0.15, 0.15, 0.0])                                   # half a lattice cell in x and y: every ground query starts between 4 neighbours
    Pc = P - P.mean(0)
    src = ((Pc @ T[:3, :3].T + T[:3, 3]) + P.mean(0) + off).astype(np.float32)
    r, bad, passes, first, g = run_verified(engine, ctx, src, tgt, optimizer, force, track_from=1)
    assert passes >= (15 if force else 2), passes
    assert bad == 0, "tracked pass differs from a fresh search at %d queries (first at sorted position %d)" % (bad, first - 1)


@pytest.mark.parametrize("pair_id", [340, 341])
def test_street_scene_8km_out(eng, oracle, pair_id):
    engine, ctx = eng
    src, tgt, T = synth.make_pair(pair_id, 40000)
    off = np.array([8000.0, 6000.0, 50.0])
    src8 = (src.astype(np.float64) + off).astype(np.float32); tgt8 = (tgt.astype(np.float64) + off).astype(np.float32)
    r, bad, passes, first, g = run_verified(engine, ctx, src8, tgt8, "gn", 20, track_from=1)
    assert passes >= 15 and bad == 0, (passes, bad, first)
    # and the answer is the oracle's (KD-tree search, no pruning): identical trajectory
    o = oracle.GicpOracle(k=20, max_iter=32, max_corr_dist=52.5, trans_eps=1e-5, rot_eps=1e-6, optimizer="gn", force_iterations=20)
    o.set_source(src8); o.compute_covariances(0); o.set_target(tgt8); o.compute_covariances(1)
    ro = o.align()
    dt, dr = synth.pose_error(np.array(r.T64).reshape(4, 4), ro["T"])
    assert dt <= 1e-4 and dr <= 1e-4, (dt, dr)
    assert abs(r.fitness - ro["fitness"]) <= 1e-6 * ro["fitness"]


def test_default_schedule_lm_reference_point(eng):
    """the reference's operating point (k = 15, LM, real stopping rule) with the production tick schedule, every tracked pass verified"""
    engine, ctx = eng
    src, tgt, T = synth.make_pair(342, 30000)
    g = engine.NanoGICP(ctx)
    g.setCorrespondenceRandomness(15); g.setMaximumIterations(32); g.setMaxCorrespondenceDistance(52.5); g.setTransformationEpsilon(0.001)
    ctx.debug_set("verify_track", 1)
    g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
    r = g.align()
    bad, passes = int(ctx.debug_get("verify_mismatches")), int(ctx.debug_get("verify_passes"))
    ctx.debug_set("verify_track", 0)
    assert bad == 0 and passes >= 1, (bad, passes)


@pytest.mark.parametrize("optimizer,force", [("gn", 20), ("lm", 0)])
def test_partial_overlap_far_queries(eng, oracle, optimizer, force):
    """80 % overlap: a fifth of the source has its nearest target point metres away (beyond the target's extent).  Those queries go through
    the candidate cache / k_far refresh path (qn_tick.cuh); every tracked pass is compared with a fresh search, and the answer with the oracle."""
    engine, ctx = eng
    src, tgt, T = synth.make_pair(343, 40000, shift=24.0)
    r, bad, passes, first, g = run_verified(engine, ctx, src, tgt, optimizer, force)
    assert passes >= (15 if force else 2) and bad == 0, (passes, bad, first)
    o = oracle.GicpOracle(k=20, max_iter=32, max_corr_dist=52.5, trans_eps=1e-5, rot_eps=1e-6, optimizer=optimizer, force_iterations=force)
    o.set_source(src); o.compute_covariances(0); o.set_target(tgt); o.compute_covariances(1)
    ro = o.align()
    assert r.iterations == ro["iterations"] and bool(r.converged) == ro["converged"]
    dt, dr = synth.pose_error(np.array(r.T64).reshape(4, 4), ro["T"])
    assert dt <= 1e-4 and dr <= 1e-4, (dt, dr)
    assert abs(r.fitness - ro["fitness"]) <= 1e-6 * ro["fitness"]


def test_far_refresh_distributions_agree():
    """k_far deals its refresh requests by global rank (balanced) or, for clouds beyond 262144 points, word by word to the blocks: both serve every request with the
    same exact search - same iteration count, same neighbours (score) and the same pose to rounding (the far contributions are summed in a different order)."""
    from qn_amd import engine
    src, tgt, _ = synth.make_pair(77, 40000, shift=24.0)
    out = []
    for ranked in (1, 0):
        ctx = engine.Context(41024)
        ctx.debug_set("far_ranked", ranked); ctx.debug_set("persist", 0)
        g = engine.NanoGICP(ctx)
        g.setCorrespondenceRandomness(15); g.setMaximumIterations(12); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer("gn"); g.setForceIterations(12)
        g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
        r = g.align()
        out.append((np.array(r.T64).reshape(4, 4).copy(), r.fitness, r.iterations))
        ctx.close()
    assert out[0][2] == out[1][2]
    assert np.max(np.abs(out[0][0] - out[1][0])) < 1e-9 and abs(out[0][1] - out[1][1]) <= 1e-9 * abs(out[0][1])
