"""Feature matching on the matrix cores (csrc/qn_feat_mm.cuh) against the VALU search it replaces (k_feat_nn) and against the oracle.
The screening GEMM may only ever add work: every survivor is re-evaluated with the defining f32 arithmetic, so the matches must be
IDENTICAL - for every query of both search directions (debug knob feat_verify), on the awkward inputs too:
  * duplicate descriptors on both sides (a cloud matched against a copy of itself: every exact-plane point has the same FPFH row);
  * clouds smaller than one 32-row tile, ragged sizes;
  * every sampling step of the lower-bound pass, including one so coarse that the survivor list overflows and the search is
    repeated with the VALU kernel (counted by feat_fallbacks);
  * advancedMatching (no gate, no cap)."""
import numpy as np
import pytest
from qn_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from qn_amd import engine
    ctx = engine.Context(40000)
    yield engine, ctx
    ctx.debug_set("feat_mfma", 1); ctx.debug_set("feat_sample", 4); ctx.debug_set("feat_verify", 0)
    ctx.close()


def run(engine, ctx, src, tgt, mm, sample=4, verify=False, **kw):
    ctx.debug_set("feat_mfma", 1 if mm else 0); ctx.debug_set("feat_sample", sample); ctx.debug_set("feat_verify", 1 if (verify and mm) else 0)
    q = engine.Quatro(ctx, **kw)
    r = q.align(src, tgt, debug=True)
    r["fallbacks"] = int(ctx.debug_get("feat_fallbacks"))
    if verify and mm:
        r["verified"], r["mismatches"] = int(ctx.debug_get("feat_verified")), int(ctx.debug_get("feat_mismatches"))
    ctx.debug_set("feat_verify", 0)
    return r


def same(a, b):
    return np.array_equal(a["mutual"], b["mutual"]) and np.array_equal(a["corres"], b["corres"]) and np.array_equal(a["T"], b["T"]) and a["valid"] == b["valid"]


@pytest.mark.parametrize("n,pair_id", [(3000, 7), (20000, 320), (30000, 331)])
@pytest.mark.parametrize("sample", [1, 4, 8])
def test_every_query_matches_the_valu_search(eng, n, pair_id, sample):
    engine, ctx = eng
    src, tgt, _ = synth.make_pair(pair_id, n, mode="quatro")
    fb0 = int(ctx.debug_get("feat_fallbacks"))
    a = run(engine, ctx, src, tgt, True, sample=sample, verify=True)
    assert a["verified"] >= n and a["mismatches"] == 0, a
    assert a["fallbacks"] == fb0, "the survivor list overflowed on an ordinary pair"
    b = run(engine, ctx, src, tgt, False)
    assert same(a, b)


def test_against_the_oracle_matcher(eng, oracle):
    engine, ctx = eng
    src, tgt, _ = synth.make_pair(321, 12000, mode="quatro")
    a = run(engine, ctx, src, tgt, True)
    q = engine.Quatro(ctx); q.align(src, tgt)
    fs, ft = q.features(0)[2], q.features(1)[2]
    mutual, corres = oracle.quatro_match(src, tgt, fs, ft)                  # the oracle's matcher on the GPU's descriptors
    assert np.array_equal(a["mutual"], mutual) and np.array_equal(a["corres"], corres)


def test_cloud_against_itself_all_duplicates(eng):
    """noise-free cloud vs a rigidly moved copy: the descriptor sets are (nearly) identical row sets with thousands of exact duplicates"""
    engine, ctx = eng
    src, _, _ = synth.make_pair(322, 15000, mode="quatro")
    c, s = np.cos(0.3), np.sin(0.3)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    tgt = (src.astype(np.float64) @ R.T + np.array([1.0, -2.0, 0.1])).astype(np.float32)
    a = run(engine, ctx, src, tgt, True, verify=True)
    assert a["mismatches"] == 0 and a["verified"] >= len(src)
    assert same(a, run(engine, ctx, src, tgt, False))
    a2 = run(engine, ctx, src, src.copy(), True, verify=True)             # literally the same cloud: every row has its duplicate on the other side
    assert a2["mismatches"] == 0 and same(a2, run(engine, ctx, src, src.copy(), False))


@pytest.mark.parametrize("ns,nt", [(20, 25), (33, 31), (700, 95), (64, 4000)])
def test_tiny_and_ragged_clouds(eng, ns, nt):
    engine, ctx = eng
    src, tgt, _ = synth.make_pair(323, 5000, mode="quatro")
    rng = np.random.default_rng(ns * 1000 + nt)
    s = np.ascontiguousarray(src[np.sort(rng.choice(len(src), ns, replace=False))]); t = np.ascontiguousarray(tgt[np.sort(rng.choice(len(tgt), nt, replace=False))])
    a = run(engine, ctx, s, t, True, verify=True)
    assert a["mismatches"] == 0
    assert same(a, run(engine, ctx, s, t, False))


def test_survivor_overflow_falls_back_to_the_valu_search(eng):
    engine, ctx = eng
    src, tgt, _ = synth.make_pair(331, 30000, mode="quatro")
    fb0 = int(ctx.debug_get("feat_fallbacks"))
    a = run(engine, ctx, src, tgt, True, sample=100000)                    # one sampled tile per segment: almost every candidate "survives"
    assert a["fallbacks"] > fb0, "expected the overflow path"
    assert same(a, run(engine, ctx, src, tgt, False))


def test_advanced_matching(eng):
    engine, ctx = eng
    src, tgt, _ = synth.make_pair(324, 9000, mode="quatro")
    a = run(engine, ctx, src, tgt, True, verify=True, use_optimized_matching=False)
    assert a["mismatches"] == 0
    assert same(a, run(engine, ctx, src, tgt, False, use_optimized_matching=False))
