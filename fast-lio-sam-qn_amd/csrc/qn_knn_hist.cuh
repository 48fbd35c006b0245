// qn_knn_hist.cuh - k-NN by histogram selection (the default calculateSource/TargetCovariances path, SURVEY.md A.1.3;
// loop_closure.cpp:121,123).  Separate from qn_device.cuh so that tuning it does not rebuild the sorted-list k-NN units.
#pragma once
#include "qn_device.cuh"

namespace qn {

// The sorted-list k-NN sink (BestK) pays an insertion chain per accepted candidate;
// wave_knn_hist instead walks the candidate stream TWICE (16 queries x 4 candidate sub-slots per wave, same clusters
// and stream as wave_search):
//   pass 1  histogram of the squared distances per query in LDS, bins = the top 11 bits of the f32 pattern
//           (8 bins per octave), 62 regular bins below (2 r)^2, one underflow bin;
//   tau     the upper edge of the first bin whose cumulative count reaches k: at least k and typically k + 1..3
//           candidates lie below it, and the k nearest are certainly among them;
//   pass 2  the candidates below tau are appended to a short per-query list in LDS (<= HCAP);
//   rank    each list entry's rank = number of smaller (d2, idx) keys in the list; rank < k -> output slot `rank`.
// Exact: the output is the k smallest keys of the scanned box in ascending (d2, idx) order, certified against the
// nearest unseen box face exactly as in wave_search.  Queries this scheme does not cover (fewer than k points within
// 2 r although the whole grid was scanned, more than HCAP candidates below tau, non-finite) return status 2 and
// go to the general sorted-list path (wave_search + BestK).
#define QN_HB 64
#define QN_HW (QN_HB + 5)
template <int HCAP>
struct WaveLdsH {
  WaveLds s;
  union {
    uint32_t hist[16][QN_HW];                     // 64 bins + 4 reject columns (one per sub-slot) + 1: rows start in different banks
    unsigned long long list[16][HCAP + 1];
  } u;
  uint32_t cnt[16];
  uint32_t kth[16];
};

// All 64 lanes call; lane l serves query (l & 15) as sub-slot (l >> 4); the 4 lanes of a query pass identical q, r,
// out pointers.  Returns the query's status in all of its lanes: 0 = done (k indices written, ascending),
// 1 = not certified within max_rounds (continue from the returned r), 2 = needs the general path.
template <int HCAP>
__device__ __forceinline__ int wave_knn_hist(const GridView& g, float qx, float qy, float qz, bool active, float& r, int k, int max_rounds,
                                             WaveLdsH<HCAP>* L, int32_t* __restrict__ idx_out, float* __restrict__ d2_out) {
  const int lane = threadIdx.x & 63, qs = lane & 15, sub = lane >> 4;
  WaveLds* lds = &L->s;
  const int cx = cell_coord(qx, g.ox, g.inv_cell, g.nx);
  const int cy = cell_coord(qy, g.oy, g.inv_cell, g.ny);
  const int cz = cell_coord(qz, g.oz, g.inv_cell, g.nz);
  const float INF = __int_as_float(0x7f800000);
  int status = active ? 1 : 0;
  unsigned long long todo = __ballot(active);
  for (int round = 0; todo != 0 && round < max_rounds; round++) {
    const bool mine = (todo >> lane) & 1ull;
    uint32_t cid; int ncl; uint32_t nseg_all;
    build_clusters<4>(g, lds, todo, cx, cy, cz, qx, qy, qz, r, cid, ncl, nseg_all);
    // ---- pass 1: histogram
    { uint32_t* hz = &L->u.hist[0][0];
      for (int e = lane; e < 16 * QN_HW; e += 64) hz[e] = 0; }
    wave_lds_fence();
    const int base = (int)(__float_as_uint(4.f * r * r) >> 20) - (QN_HB - 2);      // bin QN_HB-2 ends at (2 r)^2, bin QN_HB-1 = beyond (not counted)
    const uint32_t ncand = stream_clusters<4>(g, lds, ncl, nseg_all, [&](const float4& cp, bool in_tile, uint32_t ccid) __attribute__((always_inline)) {
      const uint32_t bits = __float_as_uint(sqdist(qx, qy, qz, cp.x, cp.y, cp.z));
      const int bin = max((int)(bits >> 20) - base, 0);
      const bool ok = mine && in_tile && ccid == cid && bin < QN_HB - 1;
      atomicAdd(&L->u.hist[qs][ok ? bin : QN_HB + sub], 1u);                      // branch-free: rejected candidates land in the sub-slot's reject column
    });
    wave_lds_fence();
    // ---- tau: sub-slot s sums bins [16 s, 16 s + 16), then looks for the crossing in its own range
    uint32_t hv[16], mysum = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) { hv[j] = L->u.hist[qs][sub * 16 + j]; mysum += hv[j]; }
    const uint32_t s0 = __shfl(mysum, qs), s1 = __shfl(mysum, qs + 16), s2 = __shfl(mysum, qs + 32), s3 = __shfl(mysum, qs + 48);
    uint32_t run = (sub > 0 ? s0 : 0u) + (sub > 1 ? s1 : 0u) + (sub > 2 ? s2 : 0u);
    const bool enough = s0 + s1 + s2 + s3 >= (uint32_t)k;
    int cross = 1 << 20;
#pragma unroll
    for (int j = 0; j < 16; j++) { run += hv[j]; if (cross == (1 << 20) && run >= (uint32_t)k) cross = sub * 16 + j; }
    cross = min(cross, __shfl_xor(cross, 16)); cross = min(cross, __shfl_xor(cross, 32));
    const uint32_t tau_bits = (uint32_t)(base + cross + 1) << 20;                  // d2 bit patterns below this pass
    wave_lds_fence();                                                               // hist is dead: the list shares its storage
    if (lane < 16) { L->cnt[lane] = 0; L->kth[lane] = 0x7f800000u; }
    wave_lds_fence();
    // ---- pass 2: collect the candidates below tau
    const bool collect = mine && enough;
    stream_clusters<4>(g, lds, ncl, nseg_all, [&](const float4& cp, bool in_tile, uint32_t ccid) __attribute__((always_inline)) {
      const float d2 = sqdist(qx, qy, qz, cp.x, cp.y, cp.z);
      if (collect && in_tile && ccid == cid && __float_as_uint(d2) < tau_bits) {
        const uint32_t pos = atomicAdd(&L->cnt[qs], 1u);
        if (pos < HCAP) L->u.list[qs][pos] = pack_key(d2, __float_as_uint(cp.w));
      }
    });
    wave_lds_fence();
    const uint32_t P = L->cnt[qs];
    const bool ok = collect && P <= HCAP;                                        // (P >= k by construction)
    // ---- rank: own entries sub, sub + 4, ...; every entry of the query's list is compared against them
    const int maxP = wave_max_i(ok ? (int)P : 0);
    unsigned long long own[HCAP / 4]; int rank[HCAP / 4];
#pragma unroll
    for (int j = 0; j < HCAP / 4; j++) { own[j] = (ok && (uint32_t)(sub + 4 * j) < P) ? L->u.list[qs][sub + 4 * j] : QN_INF_KEY; rank[j] = 0; }
    // 4 list entries per step, all four LDS reads issued before the compares (the entry -> compare chain was latency-bound)
    for (int f = 0; f < maxP; f += 4) {
      unsigned long long kf[4];
#pragma unroll
      for (int u = 0; u < 4; u++) kf[u] = L->u.list[qs][min(f + u, HCAP)];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const unsigned long long kv = (ok && (uint32_t)(f + u) < P) ? kf[u] : QN_INF_KEY;
        if (HCAP > 32 && maxP > 32) {
#pragma unroll
          for (int j = 0; j < HCAP / 4; j++) rank[j] += kv < own[j] ? 1 : 0;
        } else {
#pragma unroll
          for (int j = 0; j < 8; j++) rank[j] += kv < own[j] ? 1 : 0;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < HCAP / 4; j++) if (own[j] != QN_INF_KEY && rank[j] == k - 1) L->kth[qs] = (uint32_t)(own[j] >> 32);
    wave_lds_fence();
    const float kth_d2 = __uint_as_float(L->kth[qs]);
    if (g.dbg && lane == 0) { atomicAdd(&g.dbg[0], (uint32_t)ncl); atomicAdd(&g.dbg[1], ncand); }
    // ---- certification (as in wave_search)
    bool retry = false;
    if (mine) {
      const int* b = lds->box[cid];
      const int ex0 = b[0], ex1 = b[1], ey0 = b[2], ey1 = b[3], ez0 = b[4], ez1 = b[5];
      float d = INF;
      if (ex0 > 0) d = fminf(d, qx - (g.ox + ex0 * g.cell));
      if (ex1 < g.nx - 1) d = fminf(d, (g.ox + (ex1 + 1) * g.cell) - qx);
      if (ey0 > 0) d = fminf(d, qy - (g.oy + ey0 * g.cell));
      if (ey1 < g.ny - 1) d = fminf(d, (g.oy + (ey1 + 1) * g.cell) - qy);
      if (ez0 > 0) d = fminf(d, qz - (g.oz + ez0 * g.cell));
      if (ez1 < g.nz - 1) d = fminf(d, (g.oz + (ez1 + 1) * g.cell) - qz);
      const bool whole = d == INF || !(r == r);
      if (!enough) {
        if (whole) status = 2;                                                      // fewer than k points within 2 r of the whole cloud
        else { r = 2.f * r + g.cell; retry = true; }
      } else if (!ok) status = 2;                                                   // list overflow
      else {
        d -= g.eps;
        if (whole || (d > 0.f && kth_d2 < d * d)) {
          status = 0;
#pragma unroll
          for (int j = 0; j < HCAP / 4; j++) if (own[j] != QN_INF_KEY && rank[j] < k) {
            idx_out[rank[j]] = (int32_t)key_idx(own[j]);
            if (d2_out) d2_out[rank[j]] = key_d2(own[j]);
          }
        } else { r = fmaxf(sqrtf(kth_d2) * 1.000001f + g.eps, r); retry = true; }
      }
    }
    todo = __ballot(retry && round + 1 < max_rounds);
    if (g.dbg && lane == 0 && todo) atomicAdd(&g.dbg[3], (uint32_t)__popcll(todo));
  }
  return status;
}


}  // namespace qn
