"""BASELINE.json configs[1] LITERALLY - 100k x 100k points, k = 20 covariances, 20 FORCED Gauss-Newton iterations - HIP path vs the
C++ oracle (call site fast_lio_sam_qn/src/loop_closure.cpp:124-129): the workload bench.py times, on bench pair 0 and on an
80 %-overlap pair (SURVEY 8d's generator case, target window shifted 24 m).  Plus the driver-facing pieces around it: the
bench's helper must not clobber a context's parameters, forced LM runs terminate, and the single-process N-GPU bench mode runs."""
import json
import os
import subprocess
import sys
import numpy as np
import pytest
from qn_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, K, ITERS = 100000, 20, 20
TOL_T, TOL_R = 1e-4, 1e-4          # north-star tolerance: <= 1e-4 m, <= 1e-4 rad


@pytest.fixture(scope="module")
def eng():
    from qn_amd import engine
    ctx = engine.Context(N + 1024)
    yield engine, ctx
    ctx.close()


def _forced_gn(engine, ctx, oracle, src, tgt, device=False):
    g = engine.NanoGICP(ctx)
    g.setCorrespondenceRandomness(K); g.setMaximumIterations(ITERS); g.setMaxCorrespondenceDistance(52.5)
    g.setOptimizer("gn"); g.setForceIterations(ITERS)
    keep = None
    if device:                       # the entry points bench.py uses: raw clouds resident in HBM
        import torch
        keep = (torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()); torch.cuda.synchronize()
        g.setInputSourceDevice(keep[0].data_ptr(), len(src), 12); g.calculateSourceCovariances()
        g.setInputTargetDevice(keep[1].data_ptr(), len(tgt), 12); g.calculateTargetCovariances()
    else:
        g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
    g.align(); r = g.result_dict()
    o = oracle.GicpOracle(k=K, max_iter=ITERS, max_corr_dist=52.5, optimizer="gn", force_iterations=ITERS)
    o.set_source(src); o.compute_covariances(0); o.set_target(tgt); o.compute_covariances(1)
    return r, o.align()


@pytest.mark.parametrize("pair_id,shift,device", [(0, None, True), (0, None, False), (9000, 24.0, True), (3, None, True)])
def test_bench_workload_vs_oracle(eng, oracle, pair_id, shift, device):
    engine, ctx = eng
    src, tgt, _ = synth.make_pair(pair_id, N, shift=shift)
    r, ro = _forced_gn(engine, ctx, oracle, src, tgt, device)
    assert r["iterations"] == ITERS == ro["iterations"]
    tr, tro = r["trace"], ro["trace"]
    assert tr.shape == tro.shape and tr.shape[0] == ITERS
    assert np.allclose(tr[:, 0], tro[:, 0], rtol=1e-8), np.abs(tr[:, 0] / tro[:, 0] - 1).max()       # y0 (cost) trajectory, every iteration
    dt, dr = synth.pose_error(r["T"], ro["T"])
    assert dt <= TOL_T and dr <= TOL_R, (dt, dr)
    assert np.abs(r["T"] - ro["T"]).max() <= 1e-9, np.abs(r["T"] - ro["T"]).max()                     # what bench.py's spot check asserts
    assert abs(r["fitness"] - ro["fitness"]) <= 1e-6 * ro["fitness"]


def test_helpers_leave_the_context_parameters_alone(eng):
    """bench.py runs engine.icp_alignment (k = 15, LM) on the context its forced-GN object is bound to; the object must keep working
    without re-pushing (round 2's spot check compared a k = 15 LM run with the 20-GN oracle result because of this)."""
    engine, ctx = eng
    src, tgt, _ = synth.make_pair(21, 6000, extent=45.0)
    g = engine.NanoGICP(ctx)
    g.setCorrespondenceRandomness(K); g.setMaximumIterations(7); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer("gn"); g.setForceIterations(7)
    def run():
        g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
        return g.align()
    a = run()
    engine.icp_alignment(ctx, src, tgt)
    engine.coarse_to_fine_alignment(ctx, src, tgt)
    b = run()
    assert a.iterations == b.iterations == 7 and list(a.T64) == list(b.T64) and a.fitness == b.fitness


@pytest.mark.parametrize("seed,rot_sigma,scene", [(0, 0.03, True), (0, 0.1, True), (5, 0.1, True), (1, 0.1, False), (4, 0.1, False)])
@pytest.mark.parametrize("force", [0, 12])
def test_lm_with_rejected_trials(eng, oracle, seed, rot_sigma, scene, force):
    """LM runs whose trial steps are REJECTED (inner tries up to 5; synth.lever_arm_pair), with the real stopping rule and with
    force_iterations > 0: every rejected trial needs one more device tick than 2 x iterations (the host loop of a forced LM
    run used to spin forever once its fixed tick allowance was spent).  Same branches as the oracle, step by step."""
    engine, ctx = eng
    src, tgt, guess = synth.lever_arm_pair(seed, rot_sigma=rot_sigma, scene=scene, n=3000 if scene else 1000)
    mi = 12 if force else 32
    g = engine.NanoGICP(ctx); g.setCorrespondenceRandomness(15); g.setMaximumIterations(mi); g.setMaxCorrespondenceDistance(52.5); g.setTransformationEpsilon(0.01); g.setForceIterations(force)
    g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
    g.align(guess.astype(np.float32)); r = g.result_dict()
    o = oracle.GicpOracle(k=15, max_iter=mi, max_corr_dist=52.5, trans_eps=0.01, force_iterations=force)
    o.set_source(src); o.compute_covariances(0); o.set_target(tgt); o.compute_covariances(1)
    ro = o.align(guess.astype(np.float32).astype(np.float64))
    assert (ro["trace"][:, 5] > 1).any(), "the case must contain a rejected LM trial"
    assert r["iterations"] == ro["iterations"] and r["converged"] == ro["converged"]
    assert np.array_equal(r["trace"][:, 5:], ro["trace"][:, 5:]), (r["trace"][:, 5], ro["trace"][:, 5])
    assert np.allclose(r["trace"][:, 0], ro["trace"][:, 0], rtol=1e-8)
    dt, dr = synth.pose_error(r["T"], ro["T"])
    assert dt <= TOL_T and dr <= TOL_R, (dt, dr)


def test_single_process_bench_mode_runs_qn_multi_over_every_visible_gpu():
    """`bench.py --gpus N --single-process` = one process, qn_multi_init(N) (ncclCommInitAll(N)) + the grouped all-gather, N = every GPU of the box
    (1 on the test box; the same command exercises 8 ranks on a node)."""
    import torch
    n = torch.cuda.device_count()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--single-process", "--steps", "6", "--warmup", "2", "--pairs", "2"],
                         capture_output=True, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    line = json.loads(out.stdout.decode().strip().splitlines()[-1])
    c = line["config"]
    assert line["n_gpus"] == n and c["rccl_ranks"] == n and len(c["per_gpu_pairs_per_s"]) == n and line["value"] > 0
    assert line["roofline"]["frac"] > 0 and c["gather_ms"] >= 0
