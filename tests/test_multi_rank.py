"""N > 1 path on CPU (gloo, world_size 2): pairs shard pair i -> rank i mod N with no data-path collective,
one all_gather of fixed-size result records, rank 0 picks the winning loop.  The
registrations themselves need a GPU, so the records here come from the CPU oracle - what is under
test is the sharding + gather + arg-min logic bench.py uses."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shard(n_pairs, rank, world):
    return [i for i in range(n_pairs) if i % world == rank]


def pick_winner(records, score_thr):
    ok = [r for r in records if r[1] > 0.5 and r[2] < score_thr]
    return min(ok, key=lambda r: r[2]) if ok else None


def _worker(rank, world, port, n_pairs, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
    from qn_amd import synth
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    recs = []
    for pid in shard(n_pairs, rank, world):
        src, tgt, _ = synth.make_pair(200 + pid, 1500, extent=30.0, shift=1.0 + 2.0 * pid)
        r = orc.icp_alignment(src, tgt)
        recs.append([float(pid), float(r["valid"]), r["score"]] + list(r["T"].reshape(-1)))
    mine = torch.tensor(recs, dtype=torch.float64)
    allrec = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allrec, mine)
    if rank == 0:
        q.put(torch.cat(allrec).numpy())
    dist.barrier(); dist.destroy_process_group()


def test_shard_covers_every_pair_once():
    for world in (1, 2, 4, 8):
        got = sorted(sum((shard(64, r, world) for r in range(world)), []))
        assert got == list(range(64))


def test_two_rank_gather_picks_same_winner_as_single_process(oracle):
    from qn_amd import synth
    n_pairs, world, port = 4, 2, 29533
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, q)) for r in range(world)]
    [p.start() for p in procs]
    table = q.get(timeout=300)
    [p.join(timeout=60) for p in procs]
    assert sorted(table[:, 0].astype(int).tolist()) == list(range(n_pairs))
    single = []
    for pid in range(n_pairs):
        src, tgt, _ = synth.make_pair(200 + pid, 1500, extent=30.0, shift=1.0 + 2.0 * pid)
        r = oracle.icp_alignment(src, tgt)
        single.append([float(pid), float(r["valid"]), r["score"]] + list(r["T"].reshape(-1)))
    w_multi = pick_winner(table.tolist(), 1.5); w_single = pick_winner(single, 1.5)
    assert (w_multi is None) == (w_single is None)
    if w_single is not None:
        assert int(w_multi[0]) == int(w_single[0]) and np.allclose(w_multi[2:], w_single[2:])


# ------------------------------------------------------------------ bench.py's own launcher: loud failures, no silent 1-rank runs
def _run_bench(args, env=None, timeout=600):
    import subprocess
    e = dict(os.environ); e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=timeout)


def test_bench_refuses_more_ranks_than_gpus():
    """`--gpus N` used to be parsed and ignored.  Now a plain launch starts N ranks itself and, with fewer than N GPUs
    visible, fails loudly instead of measuring one rank."""
    ngpu = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run_bench(["--gpus", str(ngpu + 2), "--steps", "4", "--warmup", "1"])
    assert r.returncode != 0 and ("needs %d GPUs" % (ngpu + 2)) in (r.stderr + r.stdout)


def test_bench_refuses_world_size_mismatch():
    r = _run_bench(["--gpus", "1", "--steps", "4"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


# ------------------------------------------------------------------ GPU: the N > 1 path THROUGH THE ENGINE
@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_through_the_engine():
    """2 ranks (gloo for the collectives, both on cuda:0) run bench.py's own sharding / gather / arg-min with real registrations."""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "8", "--warmup", "2", "--pairs", "4", "--in-flight", "2", "--batch-pairs", "8", "--no-cpu-baseline", "--no-quatro"],
                   env={"QN_BENCH_BACKEND": "gloo"}, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["steps"] == 8
    b = j["config"]["batch64"]
    assert b["pairs"] == 8 and b["pairs_per_s"] > 0 and 0 <= b["winner_pair"] < 8


@pytest.mark.gpu
def test_qn_multi_align_best_matches_single_context(oracle):
    """qn_multi_init / qn_multi_align_best (single process, RCCL communicator, one all-gather of the 96-byte records) on the GPUs of
    this box: every record equals a plain qn_icp_alignment of that pair, the winner is the arg-min over valid records."""
    from qn_amd import engine, synth
    ngpu = torch.cuda.device_count()
    mg = engine.MultiGpu(ngpu, 8192, in_flight=2)
    gg = engine.GicpParams(); engine.lib().qn_gicp_default_params(__import__("ctypes").byref(gg))
    gg.k_correspondences = 15; gg.max_iterations = 32; gg.max_corr_dist = 52.5; gg.transformation_epsilon = 0.01
    mg.set_params(gg)
    clouds = [synth.make_pair(220 + pid, 3000 + 100 * pid, extent=36.0, shift=1.0 + 3.0 * pid)[:2] for pid in range(7)]
    recs, best = mg.align_best([(s, len(s), t, len(t), 12, 0) for s, t in clouds] + [(np.zeros((0, 3), np.float32), 0, clouds[0][1], len(clouds[0][1]), 12, 0)])
    ctx = engine.Context(8192)
    ref = [engine.icp_alignment(ctx, s, t) for s, t in clouds]
    for pid, (r, e) in enumerate(zip(recs, ref)):
        assert r.pair_id == pid and r.status == 0 and bool(r.valid) == e["valid"] and bool(r.converged) == e["converged"] and r.iterations == e["iterations"]
        assert r.fitness == e["score"] and np.array_equal(np.array(r.T, dtype=np.float32).reshape(4, 4).astype(np.float64), e["T"])
        o = oracle.icp_alignment(*clouds[pid])
        assert bool(r.valid) == o["valid"] and abs(r.fitness - o["score"]) <= 1e-6 * o["score"]
    assert recs[7].status == engine.QN_ERR_EMPTY_CLOUD and not recs[7].valid           # an empty candidate is an invalid registration, not a failure
    ok = [r for r in recs[:7] if r.valid]
    if ok:
        w = min(ok, key=lambda r: (r.fitness, r.pair_id))
        assert best is not None and best.pair_id == w.pair_id and best.fitness == w.fitness
    else:
        assert best is None
    with pytest.raises(engine.EngineError) as ei:
        engine.MultiGpu(ngpu + 1, 1024)
    assert ei.value.status == engine.QN_ERR_NO_DEVICE
    mg.close(); ctx.close()


@pytest.mark.gpu
def test_candidates_of_one_query_share_the_prepared_source():
    """64-candidate style batch: every pair has the SAME source buffer - qn_multi prepares it once per context (grid + covariances) and
    registers the other targets against it (qn_icp_alignment_same_source); records must equal independent full registrations bit for bit,
    also when a different source is interleaved."""
    from qn_amd import engine, synth
    import ctypes as C
    mg = engine.MultiGpu(1, 8192, in_flight=2)
    gg = engine.GicpParams(); engine.lib().qn_gicp_default_params(C.byref(gg))
    gg.k_correspondences = 15; gg.max_iterations = 32; gg.max_corr_dist = 52.5; gg.transformation_epsilon = 0.01
    mg.set_params(gg)
    src, tgt0, _ = synth.make_pair(240, 4000, extent=36.0)
    other_src, other_tgt, _ = synth.make_pair(241, 3500, extent=36.0)
    tgts = []
    for v in range(6):
        a = 0.004 * v; c, s = np.cos(a), np.sin(a)
        R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
        tgts.append((tgt0.astype(np.float64) @ R.T + np.array([0.05 * v, -0.02 * v, 0.0])).astype(np.float32))
    pairs = [(src, len(src), t, len(t), 12, 0) for t in tgts[:3]] + [(other_src, len(other_src), other_tgt, len(other_tgt), 12, 0)] + [(src, len(src), t, len(t), 12, 0) for t in tgts[3:]]
    recs, best = mg.align_best(pairs)
    ctx = engine.Context(8192)
    for r, p in zip(recs, pairs):
        e = engine.icp_alignment(ctx, p[0], p[2])
        assert r.status == 0 and r.iterations == e["iterations"] and bool(r.converged) == e["converged"]
        assert r.fitness == e["score"] and np.array_equal(np.array(r.T, dtype=np.float32).reshape(4, 4).astype(np.float64), e["T"])
    # and the entry point itself: NOT_READY without a prepared source, same answer with one
    c2 = engine.Context(8192)
    g2 = engine.NanoGICP(c2); g2.p = gg; g2.bind()                # the context's own parameters = the reference's (the helper below restores what it finds)
    res = engine.GicpResult(); valid = C.c_int()
    st = c2._l.qn_icp_alignment_same_source(c2.h, engine._p(tgts[0]), C.c_uint32(len(tgts[0])), C.c_uint32(12), C.c_int(0), C.c_double(1.5), C.byref(res), C.byref(valid))
    assert st == engine.QN_ERR_NOT_READY
    e0 = engine.icp_alignment(c2, src, tgts[0])
    c2.check(c2._l.qn_icp_alignment_same_source(c2.h, engine._p(tgts[1]), C.c_uint32(len(tgts[1])), C.c_uint32(12), C.c_int(0), C.c_double(1.5), C.byref(res), C.byref(valid)))
    e1 = engine.icp_alignment(ctx, src, tgts[1])
    assert res.fitness == e1["score"] and res.iterations == e1["iterations"]
    mg.close(); ctx.close(); c2.close()
