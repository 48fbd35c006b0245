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


def _adversarial_descriptors(seed, ns, nt):
    """FPFH-shaped rows (three 11-bin groups summing to 100) built to sit on the screening bound of csrc/qn_feat_mm.cuh: exact duplicates and rows that differ
    in ONE bin by 1-2 ulp (near ties: the defining f32 sum must decide, ties to the lowest index), rows within 1e-5..1e-3 of the exact-plane row P (the centred
    value x' = x - P falls into f16's subnormal range and below), rows with their 100s in OTHER bins (|x'| ~ 140: the largest products), all-equal blocks, NaN rows."""
    rng = np.random.default_rng(seed)
    P = np.zeros(33, np.float32); P[[5, 16, 27]] = 100.0

    def generic(n):
        r = rng.gamma(0.6, 1.0, (n, 3, 11)).astype(np.float64)
        r = 100.0 * r / r.sum(2, keepdims=True)
        return r.reshape(n, 33).astype(np.float32)

    def near_plane(n):
        e = (10.0 ** rng.uniform(-5.5, -2.5, (n, 33))).astype(np.float32) * rng.choice([0.0, 1.0], (n, 33), p=[0.5, 0.5]).astype(np.float32)
        r = np.abs(P[None, :] - e * (P[None, :] > 0)) + e * (P[None, :] == 0)
        return r.astype(np.float32)

    def far_peaks(n):
        r = np.zeros((n, 33), np.float32)
        for g in range(3):
            r[np.arange(n), 11 * g + rng.integers(0, 11, n)] = 100.0
        return r

    def ulp_variants(base, k):
        out = []
        for _ in range(k):
            v = base.copy(); j = rng.integers(0, 33)
            v[j] = np.nextafter(v[j], np.float32(np.inf if rng.random() < 0.5 else -np.inf)) if v[j] != 0 else np.float32(1e-45) * rng.integers(0, 3)
            if rng.random() < 0.5:
                v[j] = np.nextafter(v[j], np.float32(np.inf))
            out.append(v)
        return np.array(out, np.float32)

    tg = [generic(nt // 3), near_plane(nt // 6), far_peaks(nt // 12)]
    seeds = np.concatenate([tg[0][:40], tg[1][:40], tg[2][:10]])
    tg.append(np.concatenate([ulp_variants(b, 3) for b in seeds]))                # clusters of near-tied candidates
    tg.append(np.repeat(generic(4), 60, axis=0))                                  # all-equal blocks
    tg.append(np.repeat(P[None, :], 200, axis=0))                                 # the plane row itself, many times
    t = np.concatenate(tg)
    t = np.concatenate([t, generic(max(0, nt - len(t)))])[:nt]
    t[rng.choice(nt, 15, replace=False)] = np.nan
    rng.shuffle(t, axis=0)
    ok = np.flatnonzero(~np.isnan(t[:, 0]))
    sq = [t[rng.choice(ok, ns // 3)], np.concatenate([ulp_variants(t[i], 1) for i in rng.choice(ok, ns // 3)]), near_plane(ns // 8), far_peaks(ns // 16), np.repeat(P[None, :], 50, axis=0)]
    s = np.concatenate(sq)
    s = np.concatenate([s, generic(max(0, ns - len(s)))])[:ns]
    s[rng.choice(ns, 10, replace=False)] = np.nan
    rng.shuffle(s, axis=0)
    return np.ascontiguousarray(s, np.float32), np.ascontiguousarray(t, np.float32)


@pytest.mark.parametrize("seed,ns,nt", [(1, 2500, 3000), (2, 4100, 2700), (3, 700, 5000)])
def test_adversarial_descriptors_through_match_optimized(eng, oracle, seed, ns, nt):
    """qn_match_optimized takes the caller's descriptors: rows crafted against the matrix-core screening (its bound rests on a measured property of the MFMA
    rounding, csrc/qn_feat_mm.cuh:11-15) - every query of both search directions must agree with the VALU search (feat_verify) and the outcome with the
    oracle's matcher, whose nearest neighbour is the plain f32 sum."""
    engine, ctx = eng
    fs, ft = _adversarial_descriptors(seed, ns, nt)
    rng = np.random.default_rng(100 + seed)
    src = rng.uniform(-20, 20, (ns, 3)).astype(np.float32); tgt = rng.uniform(-20, 20, (nt, 3)).astype(np.float32)
    engine.Quatro(ctx)                                                    # default Quatro parameters (seed 1) in the context
    for sample in (4, 1):
        ctx.debug_set("feat_mfma", 1); ctx.debug_set("feat_sample", sample); ctx.debug_set("feat_verify", 1)
        got = engine.match_optimized(ctx, src, tgt, fs, ft, thr_dist=1e9, num_max_corres=400, tuple_scale=0.95)
        assert int(ctx.debug_get("feat_verified")) >= ns - 10 and int(ctx.debug_get("feat_mismatches")) == 0
        ctx.debug_set("feat_verify", 0)
        p = oracle.QuatroParams(distance_threshold=1e9, max_num_corres=400)
        mutual, corres = oracle.quatro_match(src, tgt, fs, ft, p)
        assert np.array_equal(got, corres), (len(got), len(corres))
        ctx.debug_set("feat_mfma", 0)
        assert np.array_equal(engine.match_optimized(ctx, src, tgt, fs, ft, thr_dist=1e9, num_max_corres=400, tuple_scale=0.95), corres)      # and the VALU search alone
    # the forward nearest neighbours themselves, one query at a time, against the oracle's exact search
    nn = oracle.quatro_feature_nn(fs, ft)
    assert len(nn) == ns


@pytest.mark.parametrize("n,pair_id", [(3000, 7), (30000, 331)])
def test_query_side_deduplication_changes_nothing(eng, n, pair_id):
    """The forward search looks up only the lowest index of every distinct QUERY row and copies its key to the duplicates (57 % of a noise-free synthetic
    source is the exact-plane row): the same mutual pairs, correspondences and transform as searching every query."""
    engine, ctx = eng
    src, tgt, _ = synth.make_pair(pair_id, n, mode="quatro")
    ctx.debug_set("feat_query_dedupe", 1); a = run(engine, ctx, src, tgt, True)
    ctx.debug_set("feat_query_dedupe", 0); b = run(engine, ctx, src, tgt, True)
    ctx.debug_set("feat_query_dedupe", 1)
    assert same(a, b) and same(a, run(engine, ctx, src, tgt, False))
    a2 = run(engine, ctx, src, src.copy(), True); ctx.debug_set("feat_query_dedupe", 0); b2 = run(engine, ctx, src, src.copy(), True); ctx.debug_set("feat_query_dedupe", 1)
    assert same(a2, b2)
