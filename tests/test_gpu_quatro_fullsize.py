"""Quatro at BASELINE.json's sizes (configs[2]: 30k x 30k, cap 200, thr 35 m; SURVEY 8d adds 100k x 100k), GPU vs the C++ oracle through the
C-ABI: quatro<>::align (fast_lio_sam_qn/src/loop_closure.cpp:144) and coarseToFineAlignment (:138-159).

Tolerances:
  * normals: equal to <= 1e-6, and the number of points whose f32 normal differs in the last bit is REPORTED (the GPU sums the
    neighbourhood covariance in cell order, the oracle in index order: f64 sums that differ by ~1e-16 relative);
  * SPFH (integer counts x 100/(n-1)): bit-equal except where such a normal moved a pair feature across a bin edge - counted;
  * FPFH: <= 1e-4 per bin on EVERY point whose r_f-neighbourhood contains no SPFH row from that count (the rest is reported as a
    count, not silently tolerated);
  * matcher (feature NN, cross-check, gate, tuple test) on the GPU's own descriptors: correspondences identical to the oracle's;
  * solver: same clique, T <= 1e-9;  coarse-to-fine: T <= 1e-4 m / 1e-4 rad, score <= 1e-6 relative."""
import time
import numpy as np
import pytest
from scipy.spatial import cKDTree
from qn_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from qn_amd import engine
    ctx = engine.Context(101024)
    yield engine, ctx
    ctx.close()


def _fpfh_report(cloud, gpu, orc, rf):
    nrm, sp, fp = gpu; on, osp, ofp = orc
    assert np.array_equal(np.isnan(nrm), np.isnan(on))
    ok = ~np.isnan(on[:, 0])
    assert np.abs(nrm[ok] - on[ok]).max() < 1e-6
    d0 = int((np.abs(nrm[ok] - on[ok]).max(1) > 0).sum())
    dirty1 = ~(sp.view(np.uint32) == osp.view(np.uint32)).all(1)
    assert np.array_equal(np.isnan(fp), np.isnan(ofp))
    tree = cKDTree(cloud.astype(np.float64))
    tainted = np.zeros(len(cloud), bool)
    for i in np.flatnonzero(dirty1):
        tainted[tree.query_ball_point(cloud[i].astype(np.float64), rf * 1.0001)] = True
    good = ~np.isnan(ofp[:, 0]) & ~tainted
    err = np.abs(fp[good] - ofp[good]).max() if good.any() else 0.0
    return dict(normals_last_bit=d0, spfh_rows_off=int(dirty1.sum()), fpfh_rows_excluded=int(tainted.sum()), fpfh_max_err=float(err), n=len(cloud))


@pytest.mark.parametrize("pair_id,n", [(330, 30000), (331, 30000), (332, 100000)])
def test_quatro_align_parity_fullsize(eng, oracle, pair_id, n):
    engine, ctx = eng
    src, tgt, T = synth.make_pair(pair_id, n, mode="quatro")
    q = engine.Quatro(ctx)                                              # reference's effective config: r_n 0.9, r_f 1.5, cap 200, thr 35 m, noise 0.3
    t0 = time.time(); r = q.align(src, tgt, debug=True); t_gpu = time.time() - t0
    rep = []
    feats = []
    for which, cloud in ((0, src), (1, tgt)):
        g = q.features(which); feats.append(g[2])
        o = oracle.quatro_fpfh(cloud, 0.9, 1.5)
        rp = _fpfh_report(cloud, g, o, 1.5); rep.append(rp)
        assert rp["fpfh_max_err"] <= 1e-4, rp                           # every untainted point, every bin
        assert rp["spfh_rows_off"] <= max(3, n // 2000), rp              # the count that is excluded must stay marginal
    print("quatro %dk pair %d: %s | gpu align %.1f ms (first call)" % (n // 1000, pair_id, rep, 1e3 * t_gpu))
    # matcher + solver on the GPU's descriptors
    t0 = time.time(); mutual, corres = oracle.quatro_match(src, tgt, feats[0], feats[1]); t_match = time.time() - t0
    assert np.array_equal(r["mutual"], mutual) and np.array_equal(r["corres"], corres)
    o = oracle.quatro_solve(src, tgt, corres)
    assert r["valid"] == o["valid"] and r["clique"].tolist() == o["clique"].tolist() and np.abs(r["T"] - o["T"]).max() < 1e-9
    # whole coarse stage against the oracle's OWN descriptors.  Strict (same correspondences, pose <= 1e-4 m / rad) whenever no SPFH row differs - always at 30k;
    # at 100k a last-bit f32 normal may move one pair feature across a bin edge (spfh_rows_off, capped at 3 above and at 0 + strict for the 30k pairs): then the
    # correspondence sets may differ by a few pairs and the bar is SURVEY 7.7-7's for the coarse stage, 0.3 m / 2 degrees.  The branch taken is printed and recorded.
    oa = oracle.quatro_align(src, tgt)
    strict = all(rp["spfh_rows_off"] == 0 for rp in rep) or n <= 30000
    assert r["valid"] == oa["valid"]
    dt, dr = synth.pose_error(r["T"], oa["T"])
    if strict:
        assert np.array_equal(r["corres"], oa["corres"])
        assert dt <= 1e-4 and dr <= 1e-4, (dt, dr)
    else:
        assert max(rp["spfh_rows_off"] for rp in rep) <= 3, rep
        common = len(set(map(tuple, r["corres"].tolist())) & set(map(tuple, oa["corres"].tolist())))
        assert common >= 0.9 * len(oa["corres"]), (common, len(oa["corres"]))
        assert dt <= 0.3 and dr <= np.radians(2.0), (dt, dr)
    print("  coarse stage vs the oracle's own descriptors: branch %s, |dT| = %.2e m / %.2e rad; oracle matcher on %d x %d descriptors: %.1f s" % ("STRICT" if strict else "LOOSE (an SPFH bin differs)", dt, dr, len(src), len(tgt), t_match))
    if not strict:      # visible in a -q run: the loose bar was met, the strict one could not be applied (recorded as an expected failure, not as a pass)
        pytest.xfail("LOOSE branch: %d SPFH row(s) differ from the oracle's by a bin edge; coarse pose within %.2e m / %.2e rad" % (max(rp["spfh_rows_off"] for rp in rep), dt, dr))


def test_advanced_matching_parity_30k(eng, oracle):
    """use_optimized_matching = false (loop_closure.h:40, README.md:21): Matcher::advancedMatching - no gate, no cap."""
    engine, ctx = eng
    src, tgt, T = synth.make_pair(333, 30000, mode="quatro")
    q = engine.Quatro(ctx, use_optimized_matching=False)
    r = q.align(src, tgt, debug=True)
    _, _, fs = q.features(0); _, _, ft = q.features(1)
    p = oracle.QuatroParams(use_optimized_matching=False)
    mutual, corres = oracle.quatro_match(src, tgt, fs, ft, p)
    assert np.array_equal(r["mutual"], mutual) and np.array_equal(r["corres"], corres)
    o = oracle.quatro_solve(src, tgt, corres, p)
    assert r["valid"] == o["valid"] and r["clique"].tolist() == o["clique"].tolist() and np.abs(r["T"] - o["T"]).max() < 1e-9


@pytest.mark.parametrize("pair_id", [334, 335])
def test_coarse_to_fine_parity_30k(eng, oracle, pair_id):
    """LoopClosure::coarseToFineAlignment at BASELINE's size, host-buffer and device-pointer entry points."""
    import torch
    engine, ctx = eng
    src, tgt, T = synth.make_pair(pair_id, 30000, mode="quatro")
    r = engine.coarse_to_fine_alignment(ctx, src, tgt)
    o = oracle.coarse_to_fine_alignment(src, tgt)
    assert r["valid"] == o["valid"] and r["converged"] == o["converged"]
    if o["quatro"]["valid"]:
        dt, dr = synth.pose_error(r["T"], o["T"])
        assert dt <= 1e-4 and dr <= 1e-4, (dt, dr)
        assert abs(r["score"] - o["score"]) <= 1e-6 * o["score"]
    s_d = torch.from_numpy(src).cuda(); t_d = torch.from_numpy(tgt).cuda()
    rd = engine.coarse_to_fine_alignment_device(ctx, s_d.data_ptr(), len(src), t_d.data_ptr(), len(tgt), 12)
    assert rd["valid"] == r["valid"] and np.array_equal(rd["T"], r["T"]) and rd["score"] == r["score"]      # bitwise the same path
