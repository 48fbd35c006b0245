"""qn_coarse_to_fine_align_batch: the reference's DEFAULT per-candidate path - LoopClosure::coarseToFineAlignment, Quatro -> transformPcd -> Nano-GICP ->
T_gicp * T_quatro (fast_lio_sam_qn/src/loop_closure.cpp:138-159; enable_quatro_ = true, include/loop_closure.h:54, dispatch :188-192) - for MANY candidate pairs:
per context the Quatro device stages of a run of pairs are enqueued back to back on the lanes' buffers, the host solver runs per pair, the accepted pairs go
through the GICP lanes.  Every record must equal the one-pair entry point's (qn_coarse_to_fine_alignment[_device]) BIT FOR BIT, and the oracle's
(oracle.coarse_to_fine_alignment) within the north star's 1e-4 m / 1e-4 rad."""
import ctypes as C
import numpy as np
import pytest
from qn_amd import synth

pytestmark = pytest.mark.gpu
TOL_T, TOL_R = 1e-4, 1e-4


def make_ctx(engine, cap, lanes):
    ctx = engine.Context(cap)
    ctx.debug_set("batch_lanes", lanes)
    p = engine.GicpParams(); engine.lib().qn_gicp_default_params(C.byref(p))
    p.k_correspondences = 15; p.max_iterations = 32; p.max_corr_dist = 52.5; p.transformation_epsilon = 0.01      # SURVEY App. C (loop_closure.cpp:9-16 + config.yaml)
    ctx.check(engine.lib().qn_gicp_set_params(ctx.h, C.byref(p)))
    engine.Quatro(ctx)                                                                                             # quatro<> ctor: the reference's 10 arguments (loop_closure.cpp:18-27)
    return ctx


def same_record(a, b):
    return (a["valid"] == b["valid"] and a["converged"] == b["converged"] and a["score"] == b["score"] and a["iterations"] == b["iterations"]
            and np.array_equal(a["T"], b["T"]) and np.array_equal(a["T_quatro"], b["T_quatro"])
            and ("T_gicp" not in a or "T_gicp" not in b or np.array_equal(a["T_gicp"], b["T_gicp"])))      # the Nano-GICP record's own T too: the identity for a pair Quatro could not register, on both paths


@pytest.mark.parametrize("n_ctx,lanes,device", [(1, 8, False), (2, 3, True)])
def test_c2f_batch_30k_equals_the_one_pair_path_and_the_oracle(oracle, n_ctx, lanes, device):
    import torch
    from qn_amd import engine
    N = 30000
    clouds = [synth.make_pair(520 + i, N, mode="quatro")[:2] for i in range(8)]
    ctxs = [make_ctx(engine, N + 1024, lanes) for _ in range(n_ctx)]
    keep = []
    if device:
        pairs = []
        for s, t in clouds:
            ds, dt_ = torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(); keep += [ds, dt_]
            pairs.append((ds.data_ptr(), len(s), dt_.data_ptr(), len(t), 12, 1))
        torch.cuda.synchronize()
    else:
        pairs = [(s, len(s), t, len(t), 12, 0) for s, t in clouds]
    got = engine.coarse_to_fine_align_batch(ctxs, pairs)
    again = engine.coarse_to_fine_align_batch(ctxs, pairs)
    assert all(g["status"] == 0 for g in got)
    assert sum(c.debug_get("batch_pairs") for c in ctxs) > 0, "the fine stage did not go through the lanes"
    one = engine.Context(N + 1024)
    for i, ((s, t), g) in enumerate(zip(clouds, got)):
        assert same_record(g, again[i]), "pair %d: a rerun of the batch must reproduce itself bit for bit" % i
        if device:
            r = engine.coarse_to_fine_alignment_device(one, pairs[i][0], len(s), pairs[i][2], len(t), 12)
        else:
            r = engine.coarse_to_fine_alignment(one, s, t)
        assert same_record(g, r), "pair %d differs from the one-pair path: %r vs %r" % (i, {k: g[k] for k in ("valid", "score", "iterations")}, {k: r[k] for k in ("valid", "score", "iterations")})
        o = oracle.coarse_to_fine_alignment(s, t)
        assert g["valid"] == o["valid"] and g["converged"] == o["converged"], (i, g["valid"], o["valid"])
        if o["valid"]:
            dt, dr = synth.pose_error(g["T"], o["T"])
            assert dt <= TOL_T and dr <= TOL_R, (i, dt, dr)
            assert abs(g["score"] - o["score"]) <= 1e-6 * o["score"], (i, g["score"], o["score"])
    one.close()
    for c in ctxs:
        c.close()


def test_c2f_batch_with_invalid_quatro_empty_and_ragged_pairs(oracle):
    """a pair Quatro cannot register (no correspondences: is_converged = false -> invalid, no fine stage, loop_closure.cpp:145-148), an empty candidate, pairs of different
    sizes, 7 pairs through 2 contexts x 3 lanes: nobody disturbs a batch mate"""
    from qn_amd import engine
    rng = np.random.default_rng(3)
    clouds = [synth.make_pair(560 + i, 5000 + 700 * i, extent=42.0, mode="quatro")[:2] for i in range(5)]
    sparse = (rng.uniform(-20, 20, size=(300, 3)).astype(np.float32), rng.uniform(-20, 20, size=(280, 3)).astype(np.float32))      # too sparse for any normal
    items = [clouds[0], sparse, clouds[1], (np.zeros((0, 3), np.float32), clouds[2][1]), clouds[2], clouds[3], clouds[4]]
    ctxs = [make_ctx(engine, 9000, 3) for _ in range(2)]
    got = engine.coarse_to_fine_align_batch(ctxs, [(s, len(s), t, len(t), 12, 0) for s, t in items])
    one = engine.Context(9000)
    for i, ((s, t), g) in enumerate(zip(items, got)):
        if len(s) == 0:
            assert g["status"] == engine.QN_ERR_EMPTY_CLOUD and not g["valid"]
            continue
        assert g["status"] == 0
        r = engine.coarse_to_fine_alignment(one, s, t)
        assert same_record(g, r), i
        o = oracle.coarse_to_fine_alignment(s, t)
        assert g["valid"] == o["valid"], i
        if o["valid"]:
            dt, dr = synth.pose_error(g["T"], o["T"])
            assert dt <= TOL_T and dr <= TOL_R, (i, dt, dr)
    assert not got[1]["valid"] and np.array_equal(got[1]["T"], np.eye(4))
    one.close()
    for c in ctxs:
        c.close()


def test_qn_multi_routes_through_coarse_to_fine_when_quatro_is_enabled(oracle):
    """qn_multi_set_quatro_params = enable_quatro_: every pair of qn_multi_align_best is a coarseToFineAlignment; records carry T_gicp * T_quatro (f32); NULL switches back"""
    import torch
    from qn_amd import engine
    clouds = [synth.make_pair(580 + i, 6000, extent=42.0, mode="quatro")[:2] for i in range(5)]
    mg = engine.MultiGpu(torch.cuda.device_count(), 8192, in_flight=2)
    p = engine.GicpParams(); engine.lib().qn_gicp_default_params(C.byref(p))
    p.k_correspondences = 15; p.max_iterations = 32; p.max_corr_dist = 52.5; p.transformation_epsilon = 0.01
    mg.set_params(p); mg.debug_set("batch_lanes", 2)
    mg.set_quatro_params(engine.quatro_default_params())
    recs, best = mg.align_best([(s, len(s), t, len(t), 12, 0) for s, t in clouds])
    nvalid = 0
    for i, ((s, t), r) in enumerate(zip(clouds, recs)):
        o = oracle.coarse_to_fine_alignment(s, t)
        assert r.status == 0 and bool(r.valid) == o["valid"], (i, r.status, r.valid, o["valid"])
        if o["valid"]:
            nvalid += 1
            T = np.array(r.T, dtype=np.float32).reshape(4, 4).astype(np.float64)
            assert np.abs(T - o["T"]).max() <= 2e-5, (i, np.abs(T - o["T"]).max())          # the record's f32 cast of pose_between (translations of tens of metres: 1 ulp = 4e-6)
            assert abs(r.fitness - o["score"]) <= 1e-6 * o["score"]
    assert nvalid >= 3 and best is not None
    mg.set_quatro_params(None)                                                               # back to icpAlignment: Quatro-scale offsets do not converge to a valid loop with GICP alone
    recs2, _ = mg.align_best([(s, len(s), t, len(t), 12, 0) for s, t in clouds])
    ctx = engine.Context(8192)
    for (s, t), r in zip(clouds, recs2):
        e = engine.icp_alignment(ctx, s, t)
        assert r.fitness == e["score"] and r.iterations == e["iterations"]
    ctx.close(); mg.close()


def test_c2f_batch_candidates_of_one_query_share_the_source_features(oracle):
    """the candidates of ONE loop-closure query name the same source buffer: per run of lanes the source's grid, normals, SPFH and FPFH are prepared once and borrowed by the
    other lanes (batch_share_source, as in qn_gicp_align_batch).  Records must equal the one-pair entry point's bit for bit - with the sharing on and off - and the oracle's;
    a different source in between must be prepared on its own."""
    from qn_amd import engine
    src, tgt0, _ = synth.make_pair(402, 12000, mode="quatro")
    other = synth.make_pair(403, 11000, mode="quatro")[:2]
    src = np.ascontiguousarray(src, dtype=np.float32)
    tg = []
    for v in range(7):
        a = 0.004 * v; c, s = np.cos(a), np.sin(a)
        R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
        tg.append(np.ascontiguousarray((tgt0.astype(np.float64) @ R.T + np.array([0.02 * v, -0.01 * v, 0.0])).astype(np.float32)))
    items = [(src, tg[0]), (src, tg[1]), (src, tg[2]), other, (src, tg[3]), (src, tg[4]), (src, tg[5]), (src, tg[6])]
    pairs = [(s_, len(s_), t_, len(t_), 12, 0) for s_, t_ in items]
    one = engine.Context(13024)
    ref = [engine.coarse_to_fine_alignment(one, s_, t_) for s_, t_ in items]
    for share, lanes_fpfh in ((1, 0), (0, 0), (1, 1), (0, 1)):      # lanes_fpfh: grid builds + K9-K11 of all lanes in nine k_lanes launches per run (knob c2f_lanes_fpfh; measured neutral, off by default)
        ctxs = [make_ctx(engine, 13024, 4) for _ in range(2)]
        for c in ctxs:
            c.debug_set("batch_share_source", share); c.debug_set("c2f_lanes_fpfh", lanes_fpfh)
        got = engine.coarse_to_fine_align_batch(ctxs, pairs)
        for i, (g, r) in enumerate(zip(got, ref)):
            assert g["status"] == 0 and same_record(g, r), "pair %d (share %d, lanes_fpfh %d) differs from the one-pair path" % (i, share, lanes_fpfh)
        for c in ctxs:
            c.close()
    for (s_, t_), r in list(zip(items, ref))[:4]:
        o = oracle.coarse_to_fine_alignment(s_, t_)
        assert r["valid"] == o["valid"]
        if o["valid"]:
            dt, dr = synth.pose_error(r["T"], o["T"])
            assert dt <= TOL_T and dr <= TOL_R, (dt, dr)
    one.close()


def test_concurrent_callers_share_the_worker_pool():
    """two host threads, each with contexts of its own, call the batch entry points at the same time (ctypes releases the GIL): the parked workers of csrc/qn_pool.h are
    shared by both calls (and grow with them); every record equals the same call made alone"""
    import threading
    from qn_amd import engine
    cq = [synth.make_pair(590 + i, 5000, extent=42.0, mode="quatro")[:2] for i in range(6)]
    cg = [synth.make_pair(596 + i, 6000, extent=40.0)[:2] for i in range(7)]
    pq = [(s, len(s), t, len(t), 12, 0) for s, t in cq]; pg = [(s, len(s), t, len(t), 12, 0) for s, t in cg]

    def run_c2f(out):
        ctxs = [make_ctx(engine, 8192, 2) for _ in range(2)]
        out.append(engine.coarse_to_fine_align_batch(ctxs, pq))
        for c in ctxs:
            c.close()

    def run_gicp(out):
        ctxs = [make_ctx(engine, 8192, 3) for _ in range(2)]
        res, val, st = engine.icp_alignment_batch(ctxs, pg, score_thr=1.5)
        out.append([(s_, v_, r_.iterations, r_.fitness, np.array(r_.T64).tobytes()) for r_, v_, s_ in zip(res, val, st)])
        for c in ctxs:
            c.close()

    alone_q, alone_g = [], []
    run_c2f(alone_q); run_gicp(alone_g)
    for _ in range(3):
        oq, og = [], []
        th = [threading.Thread(target=run_c2f, args=(oq,)), threading.Thread(target=run_gicp, args=(og,))]
        [t.start() for t in th]; [t.join() for t in th]
        assert len(oq) == 1 and len(og) == 1
        assert all(same_record(a, b) and a["status"] == b["status"] for a, b in zip(oq[0], alone_q[0]))
        assert og[0] == alone_g[0]
