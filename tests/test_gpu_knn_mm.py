"""k-NN selection with the squared distances of both histogram passes on the matrix cores (qn_knn_hist.cuh, knob knn_mm; calculateSource/TargetCovariances,
loop_closure.cpp:121,123).  The matrix-core values only SCREEN (error bound E = 2^-16 R^2); the k indices and f32 distances must stay the
oracle's bit for bit - also where the screen is weakest: coordinates kilometres from the origin (cancellation in |q|^2 - 2 q.c + |c|^2), neighbour distances far
below the wave's extent (E no longer small against tau: the list overflows and the sorted-list kernel takes over), distances that differ in their last bits."""
import numpy as np
import pytest
from qn_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from qn_amd import engine
    ctx = engine.Context(120000)
    yield engine, ctx
    ctx.close()


def _check(engine, ctx, oracle, cloud, ks):
    g = engine.NanoGICP(ctx); g.setInputSource(cloud)
    o = oracle.GicpOracle(); o.set_source(cloud)
    for k in ks:
        idx, d2 = g.knn(0, k); oi, od = o.knn(0, cloud, k)
        assert np.array_equal(idx, oi), k
        assert np.array_equal(d2, od), k


def test_street_scene_far_from_origin(eng, oracle):
    engine, ctx = eng
    src, _, _ = synth.make_pair(47, 20000, extent=60.0)
    for off in ((8000.0, -3000.0, 150.0), (-65000.0, 40000.0, 0.0)):      # f32 coordinates with 1 mm / 4 mm resolution
        cloud = (src.astype(np.float64) + np.array(off)).astype(np.float32)
        _check(engine, ctx, oracle, cloud, (15, 20, 24))


def test_dense_clumps_in_a_wide_scene(eng, oracle):
    """Millimetre clumps of 40 points scattered over a 60 m scene: the k nearest of a clump point are its clump (d2 ~ 1e-6 m^2) while the wave's candidate box
    spans metres - the screening margin E exceeds tau itself."""
    engine, ctx = eng
    rng = np.random.default_rng(3)
    base, _, _ = synth.make_pair(48, 6000, extent=60.0)
    centres = base[rng.choice(len(base), 60, replace=False)]
    clumps = (centres[:, None, :] + rng.normal(0, 1e-3, (60, 40, 3))).reshape(-1, 3).astype(np.float32)
    cloud = np.concatenate([base, clumps]).astype(np.float32)
    _check(engine, ctx, oracle, cloud, (15, 20))


def test_near_ties(eng, oracle):
    """A lattice whose points are moved by a few ulps: squared distances that differ in their last bits or not at all (ties to the lowest index)."""
    engine, ctx = eng
    rng = np.random.default_rng(11)
    gx, gy = np.meshgrid(np.arange(70, dtype=np.float32) * 0.25, np.arange(70, dtype=np.float32) * 0.25)
    lat = np.stack([gx.ravel() + 31.0, gy.ravel() - 17.125, np.full(4900, 2.0, np.float32)], 1).astype(np.float32)      # (no coordinate is 0: its bit pattern minus 2 would be a NaN)
    bits = lat.view(np.int32) + rng.integers(-2, 3, lat.shape).astype(np.int32)
    cloud = bits.view(np.float32).copy()
    assert np.isfinite(cloud).all()
    _check(engine, ctx, oracle, cloud, (16, 20))


def test_knn_of_a_non_finite_cloud_is_refused_without_a_fault(eng):
    """qn_gicp_knn on a cloud with a NaN: the grid is empty on the device, the call reports the cloud's error - and the covariance kernel behind the selection must not
    gather through an uninitialised index table (a GPU memory fault in a long-lived process, round 6)."""
    engine, ctx = eng
    src, _, _ = synth.make_pair(49, 5000, extent=40.0)
    src = src.copy(); src[77, 1] = np.nan
    g = engine.NanoGICP(ctx); g.setInputSource(src)
    with pytest.raises(engine.EngineError):
        g.knn(0, 20)
    good, _, _ = synth.make_pair(49, 5000, extent=40.0)
    g.setInputSource(good)
    idx, d2 = g.knn(0, 20)
    assert (idx[:, 0] == np.arange(len(good))).all()


@pytest.mark.parametrize("cell", [1.5, 3.0, 0.25])
def test_coarse_and_fine_cells(oracle, cell):
    """Cell edges far from the default (knob `cell`): 1.5 m / 3 m cells put ~900 / ~3000 candidates into a round - chunks beyond the four whose operands stay in registers, several
    segment tables per round -, 0.25 m cells make every query retry with larger radii (several clusters per wave)."""
    from qn_amd import engine
    ctx = engine.Context(40000)
    ctx.debug_set("cell", cell)
    src, _, _ = synth.make_pair(52, 30000, extent=80.0)
    _check(engine, ctx, oracle, src, (20,))
    ctx.close()


def test_matches_valu_scoring_at_full_size(eng):
    """100k-point street scene: the matrix-core path and the VALU path (knn_mm 0) write the same tables."""
    engine, ctx = eng
    src, _, _ = synth.make_pair(2, 100000)
    out = []
    for mm in (1, 0):
        ctx.debug_set("knn_mm", mm)
        g = engine.NanoGICP(ctx); g.setInputSource(src)
        out.append(g.knn(0, 20))
    ctx.debug_set("knn_mm", 1)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    # brute-force spot check of 64 rows
    rng = np.random.default_rng(0)
    for i in rng.choice(len(src), 64, replace=False):
        d = ((src.astype(np.float32) - src[i]) ** 2)
        d2 = (d[:, 0] + d[:, 1]) + d[:, 2]
        order = np.lexsort((np.arange(len(src)), d2))[:20]
        assert np.array_equal(out[0][0][i], order.astype(np.int32)), i
