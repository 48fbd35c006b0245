"""The target cloud is prepared on a second stream while the source's k-NN still runs (TargetScope / join_target, qn_engine.hip).
Whatever the call order, every result must be bit-identical to a context with the pipeline switched off."""
import numpy as np
import pytest
from qn_amd import synth

pytestmark = pytest.mark.gpu


def make(engine, pipeline):
    ctx = engine.Context(20000)
    ctx.debug_set("pair_pipeline", 1 if pipeline else 0)
    g = engine.NanoGICP(ctx)
    g.setCorrespondenceRandomness(15); g.setMaximumIterations(32); g.setMaxCorrespondenceDistance(52.5); g.setTransformationEpsilon(0.01)
    return ctx, g


def same(a, b):
    return a.iterations == b.iterations and a.converged == b.converged and a.fitness == b.fitness and list(a.T64) == list(b.T64)


@pytest.mark.parametrize("order", ["reference", "target_first", "target_twice", "covariances_between", "new_source_same_target"])
def test_call_orders_match_the_single_stream_path(order):
    from qn_amd import engine
    src, tgt, _ = synth.make_pair(260, 9000, extent=40.0)
    src2, tgt2, _ = synth.make_pair(261, 7000, extent=40.0)
    out = []
    for pipeline in (True, False):
        ctx, g = make(engine, pipeline)
        res = []
        if order == "reference":                                        # loop_closure.cpp:120-124, twice in a row
            for s, t in ((src, tgt), (src2, tgt2)):
                g.setInputSource(s); g.calculateSourceCovariances(); g.setInputTarget(t); g.calculateTargetCovariances(); res.append(g.align())
        elif order == "target_first":                                   # the usual PCL order
            g.setInputTarget(tgt); g.calculateTargetCovariances(); g.setInputSource(src); g.calculateSourceCovariances(); res.append(g.align())
            g.setInputTarget(tgt2); g.setInputSource(src2); g.calculateTargetCovariances(); g.calculateSourceCovariances(); res.append(g.align())
        elif order == "target_twice":                                   # a target replaced before it was ever used; covariances computed twice
            g.setInputSource(src); g.calculateSourceCovariances()
            g.setInputTarget(tgt2); g.calculateTargetCovariances()
            g.setInputTarget(tgt); g.calculateTargetCovariances(); g.calculateTargetCovariances()
            res.append(g.align())
        elif order == "covariances_between":                            # read-backs that touch the target on the first stream
            g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
            ct = g.covariances(1); cs = g.covariances(0)
            res.append(g.align())
            f = g.getFitnessScore(4.0)
            res.append((ct.tobytes(), cs.tobytes(), f))
        else:                                                           # candidates of one query: a new source against the target already there
            g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances(); res.append(g.align())
            g.setInputSource(src2); g.calculateSourceCovariances(); res.append(g.align())
            r1 = engine.icp_alignment(ctx, src, tgt); r2 = engine.icp_alignment(ctx, src2, tgt2)
            res.append((r1["score"], r1["iterations"], r1["T"].tobytes(), r2["score"], r2["iterations"], r2["T"].tobytes()))
        out.append(res); ctx.close()
    for a, b in zip(*out):
        if isinstance(a, tuple):
            assert a == b
        else:
            assert same(a, b)
