"""The persistent align kernel (csrc/qn_persist.cuh: the tracked regime of NanoGICP::align(), loop_closure.cpp:124, as ONE launch with granule
hand-offs between resident blocks) against the k_tick chain it replaces for a single registration: the same bits - H, pose, score, iteration
trace - for GN, LM, partial overlap and LM runs with rejected trials; it really runs (counter); concurrent aligns of several contexts
neither deadlock nor change a result."""
import threading
import numpy as np
import pytest
from qn_amd import synth

pytestmark = pytest.mark.gpu


def run(engine, src, tgt, optimizer="lm", force=0, k=15, max_iter=32, guess=None, knobs=None, mcd=52.5, eps=0.01, want=None):
    ctx = engine.Context(max(len(src), len(tgt)) + 1024)
    for kk, v in (knobs or {}).items():
        ctx.debug_set(kk, v)
    g = engine.NanoGICP(ctx)
    g.setCorrespondenceRandomness(k); g.setMaximumIterations(max_iter); g.setMaxCorrespondenceDistance(mcd); g.setTransformationEpsilon(eps)
    g.setOptimizer(optimizer); g.setForceIterations(force)
    g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
    r = g.align(guess)
    out = (np.array(r.H).tobytes(), np.array(r.T64).tobytes(), r.fitness, r.iterations, r.converged, np.asarray(g.trace()).tobytes(), g.alignedCloud().tobytes())
    launches = ctx.debug_get(want or "persist_launches")
    ctx.close()
    return out, launches


CASES = {
    "gn_forced_100k": lambda: (synth.make_pair(0, 100000)[:2], dict(optimizer="gn", force=20, k=20, max_iter=20)),
    "lm_30k": lambda: (synth.make_pair(1, 30000)[:2], dict()),
    "gn_overlap80_40k": lambda: (synth.make_pair(262, 40000, shift=24.0)[:2], dict(optimizer="gn", force=10)),
    "lm_tiny": lambda: (synth.make_pair(7, 700, extent=25.0)[:2], dict()),
    "lm_rejections": lambda: (synth.lever_arm_pair(0, rot_sigma=0.1)[:2], dict(guess=synth.lever_arm_pair(0, rot_sigma=0.1)[2].astype(np.float32))),
    "lm_long_30k": lambda: (synth.make_pair(3, 30000)[:2], dict(eps=1e-12, max_iter=14)),      # the reference's optimiser, provably beyond the unseeded chunk: an epsilon no step satisfies
    "lm_rejections_forced": lambda: (synth.lever_arm_pair(5, rot_sigma=0.1)[:2], dict(guess=synth.lever_arm_pair(5, rot_sigma=0.1)[2].astype(np.float32), force=12, max_iter=12)),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_persistent_kernel_equals_the_tick_chain_bit_for_bit(case):
    from qn_amd import engine
    (src, tgt), kw = CASES[case]()
    a, la = run(engine, src, tgt, **kw)
    b, lb = run(engine, src, tgt, knobs={"persist": 0}, **kw)
    assert lb == 0
    if case not in ("gn_overlap80_40k", "lm_30k", "lm_tiny", "lm_rejections"):      # lm_long_30k, gn_forced_100k, lm_rejections_forced must reach it      # (partial overlap: the host may keep the chain for the far-query refresh kernel; a quick LM run may finish inside the unseeded chunk)
        assert la == 1, "the persistent kernel did not run"
    assert a == b
    c, _ = run(engine, src, tgt, **kw)
    assert a == c                             # and it reproduces itself
    d, _ = run(engine, src, tgt, knobs={"device_look": 0}, **kw)
    assert a == d                             # the hand-over decision taken on the device (k_look) is the host's


def test_a_persistent_launch_that_gives_up_is_rerun_on_the_chain():
    """Another tenant on the GPU, a partitioned device: the persistent kernel's blocks are not all resident and its bounded spins run out.  The reference would just be
    slow there (loop_closure.cpp:124) - the engine re-arms the kernel's buffers and runs the align again on the k_tick chain: same bits, no error."""
    from qn_amd import engine
    (src, tgt), kw = CASES["gn_forced_100k"]()
    ref, _ = run(engine, src, tgt, knobs={"persist": 0}, **kw)
    got, gave_up = run(engine, src, tgt, knobs={"persist_timeout": 1}, want="persist_gave_up", **kw)      # 10 ns per spin: the first bounded wait expires
    assert gave_up >= 1, "the time-out knob did not force a give-up"
    assert got == ref
    (src, tgt), kw = CASES["lm_long_30k"]()
    ref, _ = run(engine, src, tgt, knobs={"persist": 0}, **kw)
    got, gave_up = run(engine, src, tgt, knobs={"persist_timeout": 1}, want="persist_gave_up", **kw)
    assert gave_up >= 1 and got == ref


def test_clouds_beyond_one_point_per_lane_keep_the_chain():
    """more than 512 x 240 source points: the persistent kernel's tracking records would live in memory in its top-2 form (nn_ref.w = bound on the THIRD-nearest point),
    which is not what k_tick reads - such clouds take the chain; and the occupancy the launch needs was measured at context creation"""
    from qn_amd import engine
    src, tgt, _ = synth.make_pair(11, 130000)
    kw = dict(optimizer="gn", force=8, k=20, max_iter=8)
    a, la = run(engine, src, tgt, **kw)
    b, _ = run(engine, src, tgt, knobs={"persist": 0}, **kw)
    assert la == 0 and a == b
    ctx = engine.Context(4096)
    assert ctx.debug_get("persist_fits") == 1 and ctx.debug_get("persist_resident_blocks") >= 241
    ctx.close()


def test_concurrent_aligns_of_several_contexts():
    """Four host threads, one context each, aligning at the same time (contexts are per thread, qn_engine.h): persistent launches are for a
    registration that is alone on the GPU - whatever mix of persistent and chained aligns the race produces, nothing hangs and every
    result equals the single-threaded one."""
    from qn_amd import engine
    pairs = [synth.make_pair(300 + i, 20000)[:2] for i in range(4)]
    ref = [run(engine, s, t)[0] for s, t in pairs]
    ctxs = [engine.Context(21024) for _ in range(4)]
    out = [None] * 4

    def work(i):
        g = engine.NanoGICP(ctxs[i])
        g.setCorrespondenceRandomness(15); g.setMaximumIterations(32); g.setMaxCorrespondenceDistance(52.5); g.setTransformationEpsilon(0.01)
        res = []
        for _ in range(6):
            s, t = pairs[i]
            g.setInputSource(s); g.calculateSourceCovariances(); g.setInputTarget(t); g.calculateTargetCovariances()
            r = g.align()
            res.append((np.array(r.H).tobytes(), np.array(r.T64).tobytes(), r.fitness, r.iterations, r.converged))
        out[i] = res

    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert all(not t.is_alive() for t in th), "an align hung"
    for i in range(4):
        assert all(r == ref[i][:5] for r in out[i])
    for c in ctxs:
        c.close()
