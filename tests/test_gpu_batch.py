"""qn_gicp_align_batch: B candidate pairs per kernel launch, the pair as a grid dimension (the pairs are independent
icpAlignment calls, loop_closure.cpp:110-136).  The batched launches run the same kernel functors on the same arguments as the classic
one-registration-per-stream path, so every record must equal that path's BIT FOR BIT - forced Gauss-Newton and the reference's LM stopping rule,
ragged batches, lanes of different sizes, lanes that finish at different ticks, lanes in different far-query regimes, a shared source,
empty and non-finite clouds inside a batch - and a registration must cost <= 20 launches amortised."""
import numpy as np
import pytest
from qn_amd import synth

pytestmark = pytest.mark.gpu


def params(engine, *, k=15, max_iter=32, optimizer="lm", force=0, eps=0.01):
    import ctypes as C
    p = engine.GicpParams(); engine.lib().qn_gicp_default_params(C.byref(p))
    p.k_correspondences = k; p.max_iterations = max_iter; p.max_corr_dist = 52.5; p.transformation_epsilon = eps
    p.optimizer = 1 if optimizer == "gn" else 0; p.force_iterations = force
    return p


def set_params(engine, ctx, p):
    import ctypes as C
    ctx.check(engine.lib().qn_gicp_set_params(ctx.h, C.byref(p)))


def classic(engine, cap, p, pairs, knobs=None):
    """every pair on ONE classic context, one registration at a time, as a batch member (what each stream of the multi-stream batch does)"""
    ctx = engine.Context(cap)
    ctx.debug_set("batch_lanes", 1); ctx.debug_set("batch_member", 1); ctx.debug_set("pair_pipeline", 0)
    for k, v in (knobs or {}).items():
        ctx.debug_set(k, v)
    set_params(engine, ctx, p)
    res, val, st = engine.gicp_align_batch(ctx, pairs, score_thr=1.5)      # (batch_lanes = 1: the classic chain, pair by pair)
    out = [rec(r, v, s) for r, v, s in zip(res, val, st)]
    ctx.close()
    return out


def rec(r, v, s):
    return (s, v, r.iterations, r.converged, r.lm_failed, r.fitness, np.array(r.T64).tobytes(), np.array(r.H).tobytes(), np.array(r.T, dtype=np.float32).tobytes())


def batched(engine, cap, p, pairs, lanes, knobs=None):
    ctx = engine.Context(cap)
    ctx.debug_set("batch_lanes", lanes)
    for k, v in (knobs or {}).items():
        ctx.debug_set(k, v)
    set_params(engine, ctx, p)
    res, val, st = engine.gicp_align_batch(ctx, pairs, score_thr=1.5)
    out = [rec(r, v, s) for r, v, s in zip(res, val, st)]
    launches, npairs = ctx.debug_get("batch_launches"), ctx.debug_get("batch_pairs")
    ctx.close()
    return out, launches, npairs


def host_pairs(clouds):
    return [(s, len(s), t, len(t), 12, 0) for s, t in clouds]


@pytest.mark.parametrize("mode", ["gn_forced", "lm"])
def test_batch_records_equal_the_classic_path(mode):
    from qn_amd import engine
    # seven pairs of DIFFERENT sizes (lanes with different grids), lanes = 3: two full runs and a ragged one
    clouds = [synth.make_pair(700 + i, 6000 + 1500 * i, extent=40.0, shift=1.0 + 2.0 * i)[:2] for i in range(7)]
    p = params(engine, k=20, optimizer="gn", force=12) if mode == "gn_forced" else params(engine)
    ref = classic(engine, 20000, p, host_pairs(clouds))
    got, launches, npairs = batched(engine, 20000, p, host_pairs(clouds), lanes=3)
    assert npairs == 7 and launches > 0
    assert [g[0] for g in got] == [0] * 7
    for i, (g, r) in enumerate(zip(got, ref)):
        assert g == r, "pair %d differs from the classic path (mode %s)" % (i, mode)
    if mode == "lm":
        assert len({g[2] for g in got}) > 1, "the lanes were meant to finish at different iterations"


def test_batch_lanes_in_different_far_regimes_and_full_size():
    """100k x 100k, BASELINE configs[1] parameters: aligned scenes next to 80 %-overlap pairs (their lanes enter the k_far refresh regime, the others do not:
    per-lane chunk boundaries); launches per registration amortised over the batch."""
    from qn_amd import engine
    clouds = [synth.make_pair(40 + i, 100000, shift=(24.0 if i % 2 else None))[:2] for i in range(4)]
    p = params(engine, k=20, max_iter=20, optimizer="gn", force=20)
    ref = classic(engine, 101024, p, host_pairs(clouds))
    got, launches, npairs = batched(engine, 101024, p, host_pairs(clouds), lanes=4)
    for i, (g, r) in enumerate(zip(got, ref)):
        assert g == r, "pair %d differs from the classic path" % i
    assert launches / npairs <= 20.0, "launches per registration: %.1f" % (launches / npairs)
    again, _, _ = batched(engine, 101024, p, host_pairs(clouds), lanes=4)
    assert again == got, "a rerun of the batch must reproduce itself bit for bit"


def test_batch_shares_the_source_of_one_query():
    """the candidates of ONE loop-closure query: every pair names the same source buffer - a lane prepares it once and keeps it (qn_icp_alignment_same_source);
    a different source in between must be rebuilt."""
    from qn_amd import engine
    s0, t0, _ = synth.make_pair(810, 9000, extent=40.0)
    s1, t1, _ = synth.make_pair(811, 8000, extent=40.0)
    s0 = np.ascontiguousarray(s0, dtype=np.float32); s1 = np.ascontiguousarray(s1, dtype=np.float32)
    tg = []
    for v in range(9):
        a = 0.004 * (v + 1); R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
        tg.append(np.ascontiguousarray(t0 @ R.T + np.array([0.02 * v, -0.01 * v, 0], np.float32), dtype=np.float32))
    import torch
    ds0, ds1 = torch.from_numpy(s0).cuda(), torch.from_numpy(s1).cuda()
    dt = [torch.from_numpy(t).cuda() for t in tg] + [torch.from_numpy(np.ascontiguousarray(t1, dtype=np.float32)).cuda()]
    torch.cuda.synchronize()
    order = [(ds0, dt[0]), (ds0, dt[1]), (ds0, dt[2]), (ds1, dt[9]), (ds0, dt[3]), (ds0, dt[4]), (ds0, dt[5]), (ds0, dt[6]), (ds0, dt[7]), (ds0, dt[8])]
    pairs = [(s.data_ptr(), len(s), t.data_ptr(), len(t), 12, 1) for s, t in order]
    p = params(engine)
    # the reference: every pair as a registration of its own (fresh source every time)
    ctx = engine.Context(10024); ctx.debug_set("batch_lanes", 1); ctx.debug_set("batch_member", 1); ctx.debug_set("pair_pipeline", 0); set_params(engine, ctx, p)
    ref = []
    for pr in pairs:
        r, v, s = engine.gicp_align_batch(ctx, [pr], score_thr=1.5)
        ref.append(rec(r[0], v[0], s[0]))
    ctx.close()
    got, _, _ = batched(engine, 10024, p, pairs, lanes=2)
    assert got == ref


def test_batch_with_empty_and_non_finite_clouds():
    """an empty candidate is an invalid registration (QN_ERR_EMPTY_CLOUD, valid = 0), a cloud with a NaN is refused (QN_ERR_INVALID_ARG) - neither may disturb its batch mates"""
    from qn_amd import engine
    clouds = [synth.make_pair(830 + i, 5000, extent=36.0)[:2] for i in range(4)]
    bad = clouds[2][1].copy(); bad[17, 1] = np.nan
    pairs = host_pairs([clouds[0], (np.zeros((0, 3), np.float32), clouds[1][1]), (clouds[2][0], bad), clouds[3], (clouds[1][0], np.zeros((0, 3), np.float32))])
    p = params(engine)
    good = classic(engine, 6024, p, host_pairs([clouds[0], clouds[3]]))
    got, _, _ = batched(engine, 6024, p, pairs, lanes=4)
    assert got[0] == good[0] and got[3] == good[1]
    assert got[1][0] == engine.QN_ERR_EMPTY_CLOUD and got[1][1] == 0
    assert got[4][0] == engine.QN_ERR_EMPTY_CLOUD and got[4][1] == 0
    assert got[2][0] == engine.QN_ERR_INVALID_ARG and got[2][1] == 0


def test_multi_stream_batch_uses_the_lanes():
    """qn_icp_alignment_batch over two contexts: each context takes runs of pairs through its lanes; records equal the classic path"""
    from qn_amd import engine
    clouds = [synth.make_pair(850 + i, 7000, extent=40.0, shift=1.0 + i)[:2] for i in range(11)]
    p = params(engine, k=20, optimizer="gn", force=10)
    ref = classic(engine, 8024, p, host_pairs(clouds))
    ctxs = [engine.Context(8024) for _ in range(2)]
    for c in ctxs:
        c.debug_set("batch_lanes", 3); set_params(engine, c, p)
    res, val, st = engine.icp_alignment_batch(ctxs, host_pairs(clouds), score_thr=1.5)
    got = [rec(r, v, s) for r, v, s in zip(res, val, st)]
    assert got == ref
    assert sum(c.debug_get("batch_pairs") for c in ctxs) == 11
    for c in ctxs:
        c.close()


def test_batch_lm_with_partial_overlap_lanes():
    """the reference's optimiser (LM, real stopping rule) on a mix of aligned and 80 %-overlap pairs: lanes stop at different iterations AND sit in different far-query regimes"""
    from qn_amd import engine
    clouds = [synth.make_pair(870 + i, 20000 + 2000 * i, shift=(24.0 if i % 2 == 0 else None))[:2] for i in range(5)]
    p = params(engine)
    ref = classic(engine, 30000, p, host_pairs(clouds))
    got, _, _ = batched(engine, 30000, p, host_pairs(clouds), lanes=4)
    for i, (g, r) in enumerate(zip(got, ref)):
        assert g == r, "pair %d differs from the classic path" % i


def test_batch_strided_clouds_capacity_and_context_reuse():
    """pcl::PointXYZI-like strides (32 B, and a fat 48-byte point type), a pair beyond the context's capacity (QN_ERR_CAPACITY for that pair only), and the same context used for
    classic calls and several batches in turn (lanes are reused; a classic call in between must not disturb them)"""
    from qn_amd import engine
    clouds = [synth.make_pair(880 + i, 4000 + 300 * i, extent=36.0)[:2] for i in range(4)]
    p = params(engine)
    ref = classic(engine, 6024, p, host_pairs(clouds))

    def strided(a, stride_floats):
        out = np.full((len(a), stride_floats), 7.25, np.float32); out[:, :3] = a
        return out
    ctx = engine.Context(6024); ctx.debug_set("batch_lanes", 3); set_params(engine, ctx, p)
    for stride_floats in (8, 12):
        pr = [(strided(s, stride_floats), len(s), strided(t, stride_floats), len(t), 4 * stride_floats, 0) for s, t in clouds]
        res, val, st = engine.gicp_align_batch(ctx, pr, score_thr=1.5)
        assert [rec(r, v, s) for r, v, s in zip(res, val, st)] == ref, "stride %d bytes" % (4 * stride_floats)
        one = engine.icp_alignment(ctx, *clouds[1])                                     # a classic call on the batch context in between
        assert one["iterations"] == ref[1][2] and one["score"] == ref[1][5]
    big = synth.make_pair(889, 7000, extent=36.0)[:2]
    res, val, st = engine.gicp_align_batch(ctx, host_pairs([clouds[0], big, clouds[2]]), score_thr=1.5)
    assert st[1] == engine.QN_ERR_CAPACITY and val[1] == 0
    assert rec(res[0], val[0], st[0]) == ref[0] and rec(res[2], val[2], st[2]) == ref[2]
    ctx.close()


def test_batch_argument_arena_overflow_is_flushed_not_failed():
    """a forced Gauss-Newton run of 1500 iterations x 8 lanes needs ~6 MB of argument tables in one segment - more than the 4 MB arena: the plan is launched in pieces
    (it used to fail the whole batch with QN_ERR_HIP although the classic path handles the same parameters); records = the classic path's"""
    from qn_amd import engine
    clouds = [synth.make_pair(890 + i, 2000 + 100 * i, extent=30.0)[:2] for i in range(8)]
    p = params(engine, k=10, optimizer="gn", force=1500, max_iter=1500)
    ref = classic(engine, 4096, p, host_pairs(clouds[:2]))
    got, launches, npairs = batched(engine, 4096, p, host_pairs(clouds), lanes=8)
    assert [g[0] for g in got] == [0] * 8 and npairs == 8
    assert got[0] == ref[0] and got[1] == ref[1]
    assert all(g[2] == 1500 for g in got)


def test_lane_creation_failure_falls_back_to_the_one_pair_path_once():
    """when the lanes cannot be created (out of memory: batch_lanes x a full max_points slab; forced here by the fail_lane_create knob) the batch entry
    points register the pairs one at a time with the SAME records, the failed allocation's sticky HIP error does not surface as the first pair's status, and later calls do not
    try to create the lanes again."""
    from qn_amd import engine
    clouds = [synth.make_pair(760 + i, 5000 + 700 * i, extent=36.0)[:2] for i in range(4)]
    p = params(engine)
    ref = classic(engine, 12000, p, host_pairs(clouds))
    ctx = engine.Context(12000)
    ctx.debug_set("batch_lanes", 4); ctx.debug_set("fail_lane_create", 2)
    set_params(engine, ctx, p)
    for call in range(2):
        res, val, st = engine.gicp_align_batch(ctx, host_pairs(clouds), score_thr=1.5)
        got = [rec(r, v, s) for r, v, s in zip(res, val, st)]
        assert [g[0] for g in got] == [0] * 4, "call %d: statuses %s" % (call, [g[0] for g in got])
        for i, (g, r) in enumerate(zip(got, ref)):
            assert g == r, "call %d pair %d differs from the one-pair path" % (call, i)
        assert ctx.debug_get("lanes_failed") == 1.0 and ctx.debug_get("batch_pairs") == 0.0
    # setting batch_lanes again (with the fault gone) re-arms the lanes
    ctx.debug_set("fail_lane_create", 0); ctx.debug_set("batch_lanes", 4)
    res, val, st = engine.gicp_align_batch(ctx, host_pairs(clouds), score_thr=1.5)
    assert ctx.debug_get("lanes_failed") == 0.0 and ctx.debug_get("batch_pairs") == 4.0
    assert [rec(r, v, s) for r, v, s in zip(res, val, st)] == ref
    ctx.close()
