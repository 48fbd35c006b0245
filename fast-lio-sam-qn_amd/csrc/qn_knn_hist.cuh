// qn_knn_hist.cuh - k-NN by histogram selection (the default calculateSource/TargetCovariances path, SURVEY.md A.1.3;
// loop_closure.cpp:121,123).  Separate from qn_device.cuh so that tuning it does not rebuild the sorted-list k-NN units.
#pragma once
#include "qn_device.cuh"
#include <type_traits>

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
  union alignas(16) {
    uint32_t hist[16][QN_HW];                     // 64 bins + 4 reject columns (one per sub-slot) + 1: rows start in different banks
    unsigned long long list[16][HCAP + 1];
    uint4 zero_[(16 * QN_HW + 3) / 4];             // (the histogram is cleared with 16-byte stores)
  } u;
  uint32_t cnt[16];
  uint32_t kth[16];
  alignas(16) uint32_t tile_sc[64 * 4];                       // MM: position in pts[] (26 bits) | cluster id << 26 of the staged candidates of the round's first QN_MM_NCH chunks
};

// ---- the distance matrix of a chunk on the matrix cores (MM)
// 16 queries x 64 staged candidates = four v_mfma_f32_16x16x4_f32: A = candidates (x', y', z', |c'|^2), B = queries (-2 x', -2 y', -2 z', 1), C = |q'|^2, primes = relative
// to an origin O near the wave's queries: D = |q'|^2 - 2 q'.c' + |c'|^2, an fmaf chain in f32 (MI355X_MICROARCH.md: bitwise an fmaf chain).  Lane l gets its OWN query's
// (l & 15) distances to candidates 4 (l >> 4) + 0..3 of the group: 256 squared distances per issue slot instead of 6 VALU instructions per candidate and lane, and
// one ds_read_b32 per lane and group instead of four ds_read_b128.  The values are APPROXIMATE (cancellation: absolute error <= 50 * 2^-24 * R^2 against the oracle's
// f32 `sqdist` of the original coordinates, R = the largest |q'|, |c'| of the round - the derivation is in DESIGN.md section 4f); they only SCREEN:
//   pass 1  bins them; tau_a = upper edge of the first bin whose cumulative count reaches k  =>  at least k candidates have exact d2 < tau_a + E   (E = 2^-16 R^2);
//   pass 2  lists every candidate with approximate d2 < tau_a + 2 E  =>  the list holds EVERY candidate with exact d2 < tau_a + E, at least k of them;
//   exact   the listed candidates are re-evaluated with the defining arithmetic (sqdist on the original coordinates), keyed and ranked as before: the k smallest keys of
//           the list are the k smallest keys of the scanned box, in the oracle's order, bit for bit.
typedef float qn_f4v __attribute__((ext_vector_type(4)));
#define QN_MM_FAR 1.0e15f                          // an empty slot of the last chunk: a finite point far beyond every bin and every threshold
#define QN_MM_NCH 4                                // chunks of a round whose A operands stay in REGISTERS between the two passes (256 candidates: three rounds in four)
static_assert(QN_MM_NCH == 4, "tile_sc holds 64 x QN_MM_NCH words and is cleared with one 16-byte store per lane; the hit masks are two 32-bit words");
// stages the 64 candidates [cb, cb + 64) of the segment table in LDS: lds->tile[c] = (p - O, |p - O|^2), sc_out[c] = position in pts[] | cluster id << 26; returns
// how many of them are real (all 64 lanes call, convergent)
// (j: the segment of this lane's slot - chunk_segment, or the marker scan of the cached chunks)
__device__ __forceinline__ uint32_t stage_chunk_mm(const GridView& g, WaveLds* lds, uint32_t* __restrict__ sc_out, const uint32_t cb, const uint32_t total,
                                                   const float Ox, const float Oy, const float Oz, float& c2max, const int j) {
  const int lane = threadIdx.x & 63;
  const uint32_t slot = cb + lane;
  float4 rel = make_float4(QN_MM_FAR, 0.f, 0.f, QN_MM_FAR * QN_MM_FAR); uint32_t sc = 63u << 26;      // (an empty slot belongs to no cluster: ids are < 16)
  if (slot < total) {
    const uint32_t sidx = lds->seg_start[j] + (slot - lds->seg_excl[j]);
    sc = sidx | (lds->seg_cid[j] << 26);
    const float4 p = g.pts[sidx];
    rel.x = p.x - Ox; rel.y = p.y - Oy; rel.z = p.z - Oz; rel.w = (rel.x * rel.x + rel.y * rel.y) + rel.z * rel.z;
    c2max = fmaxf(c2max, rel.w);
  }
  wave_lds_fence();
  lds->tile[lane] = rel; sc_out[lane] = sc;
  wave_lds_fence();
  return min(64u, total - cb);
}

// All 64 lanes call; lane l serves query (l & 15) as sub-slot (l >> 4); the 4 lanes of a query pass identical q, r,
// out pointers.  Returns the query's status in all of its lanes: 0 = done (k indices written, ascending),
// 1 = not certified within max_rounds (continue from the returned r), 2 = needs the general path.
template <int HCAP, bool MM = true>
__device__ __forceinline__ int wave_knn_hist(const GridView& g, float qx, float qy, float qz, bool active, float& r, int k, int max_rounds,
                                             WaveLdsH<HCAP>* L, int32_t* __restrict__ idx_base, float* __restrict__ d2_base, const uint32_t row) {
  // (outputs: row `row` of the [n][k] tables idx_base / d2_base - the addresses are formed where the row is written: as two per-lane 64-bit pointers they were carried
  //  across the whole search and spilled)
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
    { uint4* hz = (uint4*)&L->u.hist[0][0];                                      // (16-byte stores, unrolled: the word-by-word loop was 200 instructions per round)
#pragma unroll
      for (int e = 0; e < (16 * QN_HW + 3) / 4; e += 64) if (e + lane < (16 * QN_HW + 3) / 4) hz[e + lane] = make_uint4(0u, 0u, 0u, 0u); }
    wave_lds_fence();
    const int base = (int)(__float_as_uint(4.f * r * r) >> 20) - (QN_HB - 2);      // bin QN_HB-2 ends at (2 r)^2, bin QN_HB-1 = beyond (not counted)
    // (ONE cluster - 94 % of the rounds: neighbours in the sorted order - needs no cluster test: every real candidate is everybody's, and the empty slots of the last
    //  chunk sit at infinity, beyond every bin)
    const bool one_cluster = ncl == 1;
    const int base1 = mine ? base : -(1 << 28);
    uint32_t ncand;
    // MM: origin of the relative coordinates = the first open query; B operand and C of this lane's query
    float Ox = 0.f, Oy = 0.f, Oz = 0.f, bsel = 0.f, q2 = 0.f, c2max = 0.f;
    if constexpr (MM) {
      const int l0 = __ffsll((long long)todo) - 1;
      Ox = rdlane(qx, l0); Oy = rdlane(qy, l0); Oz = rdlane(qz, l0);
      const float rx = qx - Ox, ry = qy - Oy, rz = qz - Oz;
      q2 = (rx * rx + ry * ry) + rz * rz;
      bsel = sub == 0 ? -2.f * rx : (sub == 1 ? -2.f * ry : (sub == 2 ? -2.f * rz : 1.f));
    }
    // the four groups of a chunk: d[e] = approximate squared distance from this lane's query to candidate 16 gi + 4 sub + e; av = the chunk's A operands
    const float* tilef = (const float*)&lds->tile[0];
    const qn_f4v cin = {q2, q2, q2, q2};
    auto mm_groups = [&](const float (&av)[4], const uint32_t cnt, auto&& per_group) __attribute__((always_inline)) {
      // (software-pipelined by one group: the next group's instruction is issued before this group's results are used - 40 cycles of matrix-core latency behind ~17
      //  instructions of work; the groups of a chunk's empty tail are computed too - far points - and not used)
      qn_f4v d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bsel, cin, 0, 0, 0);
#pragma unroll
      for (int gi = 0; gi < 4; gi++) {
        qn_f4v dn = d;
        if (gi < 3) dn = __builtin_amdgcn_mfma_f32_16x16x4f32(av[gi + 1], bsel, cin, 0, 0, 0);
        if ((uint32_t)(16 * gi) < cnt) per_group(gi, d);                           // (wave-uniform)
        d = dn;
      }
    };
    auto load_av = [&](float (&av)[4]) __attribute__((always_inline)) {
#pragma unroll
      for (int gi = 0; gi < 4; gi++) av[gi] = tilef[(16 * gi + qs) * 4 + sub];
    };
    // (one segment table, one cluster - three rounds in four: the A operands of its first QN_MM_NCH chunks are kept for pass 2, which then needs no staging at all)
    const bool cached = nseg_all <= 64u && one_cluster;
    float avc[QN_MM_NCH][4];
    if constexpr (MM) {
      auto p1_group = [&](const uint32_t* __restrict__ scb, const int gi, const qn_f4v d) __attribute__((always_inline)) {
        if (one_cluster) {
#pragma unroll
          for (int e = 0; e < 4; e++) {                                           // (arithmetic shift: a rounding-negative distance - the query itself - lands in bin 0)
            const int bin = min(max((__float_as_int(d[e]) >> 20) - base1, 0), QN_HB - 1);
            atomicAdd(&L->u.hist[qs][bin], 1u);
          }
        } else {
          const uint4 cc = *(const uint4*)&scb[16 * gi + 4 * sub];
          const uint32_t ccv[4] = {cc.x >> 26, cc.y >> 26, cc.z >> 26, cc.w >> 26};
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const int bin = max((__float_as_int(d[e]) >> 20) - base, 0);
            const bool ok = mine && ccv[e] == cid && bin < QN_HB - 1;
            atomicAdd(&L->u.hist[qs][ok ? bin : QN_HB + sub], 1u);
          }
        }
      };
      ncand = stream_tables<false>(g, lds, ncl, nseg_all, [&](const uint32_t total) __attribute__((always_inline)) {
        uint32_t cb0 = 0u;
        if (cached) {
          // slot -> segment for the cached chunks without a binary search per chunk: every non-empty segment marks its first slot with (index + 1) in tile_sc (the slots
          // are overwritten chunk by chunk as they are staged), a running maximum over the slots - six DPP steps per chunk and a carry - is the segment of each slot
          ((uint4*)L->tile_sc)[lane] = make_uint4(0u, 0u, 0u, 0u);
          wave_lds_fence();
          { const uint32_t ex = lds->seg_excl[lane], nx = lane < 63 ? lds->seg_excl[(lane + 1) & 63] : total;
            if (nx > ex && ex < 64u * QN_MM_NCH) L->tile_sc[ex] = (uint32_t)lane + 1u; }
          wave_lds_fence();
          uint32_t carry = 0u;
#pragma unroll
          for (int c = 0; c < QN_MM_NCH; c++) {
            if ((uint32_t)(64 * c) < total) {
              const uint32_t mk = max(wave_incl_max_u32(L->tile_sc[64 * c + lane]), carry);
              carry = rdlane(mk, 63);
              const uint32_t cnt = stage_chunk_mm(g, lds, L->tile_sc + 64 * c, 64u * c, total, Ox, Oy, Oz, c2max, (int)mk - 1);
              load_av(avc[c]);
              mm_groups(avc[c], cnt, [&](const int, const qn_f4v d) __attribute__((always_inline)) {
#pragma unroll
                for (int e = 0; e < 4; e++) atomicAdd(&L->u.hist[qs][min(max((__float_as_int(d[e]) >> 20) - base1, 0), QN_HB - 1)], 1u);
              });
            }
          }
          cb0 = 64u * QN_MM_NCH;
        }
        const uint32_t exs = lds->seg_excl[lane], nxs = lane < 63 ? lds->seg_excl[(lane + 1) & 63] : total;
        for (uint32_t cb = cb0; cb < total; cb += 64) {
          const int jg = chunk_segment(lds->tile_cid, cb, exs, nxs);
          const uint32_t cnt = stage_chunk_mm(g, lds, lds->tile_cid, cb, total, Ox, Oy, Oz, c2max, jg);
          float av[4]; load_av(av);
          mm_groups(av, cnt, [&](const int gi, const qn_f4v d) __attribute__((always_inline)) { p1_group(lds->tile_cid, gi, d); });
        }
      });
    } else {
    auto bin_of = [&](const float4& cp) __attribute__((always_inline)) { return max((int)(__float_as_uint(sqdist(qx, qy, qz, cp.x, cp.y, cp.z)) >> 20) - base, 0); };
    // (one cluster: the bin index clamped into [0, QN_HB - 1] IS the column - column QN_HB - 1 collects what lies beyond (2 r)^2 and is left out of the sums below;
    //  lanes without an open query get a base that sends everything there)
    ncand = one_cluster
      ? stream_clusters<4>(g, lds, ncl, nseg_all, [&](const float4& cp, bool, uint32_t) __attribute__((always_inline)) {
          const int bin = min(max((int)(__float_as_uint(sqdist(qx, qy, qz, cp.x, cp.y, cp.z)) >> 20) - base1, 0), QN_HB - 1);
          atomicAdd(&L->u.hist[qs][bin], 1u);
        })
      : stream_clusters<4>(g, lds, ncl, nseg_all, [&](const float4& cp, bool in_tile, uint32_t ccid) __attribute__((always_inline)) {
          const int bin = bin_of(cp);
          const bool ok = mine && in_tile && ccid == cid && bin < QN_HB - 1;
          atomicAdd(&L->u.hist[qs][ok ? bin : QN_HB + sub], 1u);                      // branch-free: rejected candidates land in the sub-slot's reject column
        });
    }
    wave_lds_fence();
    // ---- tau: sub-slot s sums bins [16 s, 16 s + 16), then looks for the crossing in its own range
    uint32_t hv[16], mysum = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) { hv[j] = L->u.hist[qs][sub * 16 + j]; if (j == 15 && sub == 3) hv[j] = 0u; mysum += hv[j]; }      // (column QN_HB - 1: beyond (2 r)^2, not counted)
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
    { uint4* lz = (uint4*)&L->u.list[0][0];                                       // every list slot starts as the infinite key: what pass 2 does not fill never counts below
      constexpr int NZ = (int)(sizeof(L->u.list) / 16);
      static_assert(sizeof(L->u.list) % 16 == 0, "list rows are cleared with 16-byte stores");
#pragma unroll
      for (int e = 0; e < NZ; e += 64) if (e + lane < NZ) lz[e + lane] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu); }
    wave_lds_fence();
    // ---- pass 2: collect the candidates below tau
    const bool collect = mine && enough;
    if constexpr (MM) {
      // threshold on the approximate distances: tau_a + 2 E, E = 2^-16 R^2 (R^2 = the largest |q'|^2, |c'|^2 of this round), rounded up
      const float R2 = wave_max_f(fmaxf(c2max, mine ? q2 : 0.f));
      const float thr = collect ? (__uint_as_float(tau_bits) + 2.f * (R2 * 1.52587890625e-5f)) * 1.000001f : -__int_as_float(0x7f800000);
      float dummy = 0.f;
      const uint32_t cidm = cid;
      if (cached) {
        // the cached chunks straight from registers: one hit bit per result (sign of d - thr), first result on top; m_lo = chunks 0, 1, m_hi = chunks 2, 3
        uint32_t mw[QN_MM_NCH / 2] = {0u, 0u};
#pragma unroll
        for (int c = 0; c < QN_MM_NCH; c++) {
          uint32_t& m = mw[c >> 1];
          if ((uint32_t)(64 * c) < ncand) {
            const uint32_t cnt = min(64u, ncand - 64u * c);
            qn_f4v d = __builtin_amdgcn_mfma_f32_16x16x4f32(avc[c][0], bsel, cin, 0, 0, 0);
#pragma unroll
            for (int gi = 0; gi < 4; gi++) {                                       // (pipelined by one group, as in pass 1; an empty tail group is far points: no hit bits)
              qn_f4v dn = d;
              if (gi < 3) dn = __builtin_amdgcn_mfma_f32_16x16x4f32(avc[c][gi + 1], bsel, cin, 0, 0, 0);
#pragma unroll
              for (int e = 0; e < 4; e++) m = __builtin_amdgcn_alignbit(m, __float_as_uint(d[e] - thr), 31);
              d = dn;
            }
          } else m <<= 16;
        }
        // list positions without atomics: the four lanes of a query exchange their hit counts
        const uint32_t cntl = (uint32_t)(__popc(mw[0]) + __popc(mw[1]));
        const uint32_t h0 = __shfl(cntl, qs), h1 = __shfl(cntl, qs + 16), h2 = __shfl(cntl, qs + 32), h3 = __shfl(cntl, qs + 48);
        uint32_t pos = (sub > 0 ? h0 : 0u) + (sub > 1 ? h1 : 0u) + (sub > 2 ? h2 : 0u);
        if (sub == 0) L->cnt[qs] = h0 + h1 + h2 + h3;                             // (chunks beyond the cached ones append behind, with the counter)
        while (__any((mw[0] | mw[1]) != 0u)) {
          const bool lo = mw[0] != 0u;
          uint32_t w = lo ? mw[0] : mw[1];
          if (w != 0u) {
            const uint32_t I = (uint32_t)__clz((int)w);                           // hit of chunk (I >> 4) of the pair, group (I >> 2) & 3, result I & 3
            const uint32_t at = (lo ? 0u : 128u) + ((I >> 4) << 6) + (((I >> 2) & 3u) << 4) + 4u * (uint32_t)sub + (I & 3u);
            if (pos < HCAP) L->u.list[qs][pos] = (unsigned long long)(L->tile_sc[at] & 0x03ffffffu);
            pos++;
            w &= ~(0x80000000u >> I);
            if (lo) mw[0] = w; else mw[1] = w;
          }
        }
        wave_lds_fence();
      }
      if (!cached || ncand > 64u * QN_MM_NCH) {
      stream_tables<true>(g, lds, ncl, nseg_all, [&](const uint32_t total) __attribute__((always_inline)) {
        const uint32_t exs = lds->seg_excl[lane], nxs = lane < 63 ? lds->seg_excl[(lane + 1) & 63] : total;
        for (uint32_t cb = cached ? 64u * QN_MM_NCH : 0u; cb < total; cb += 64) {
          const int jg = chunk_segment(lds->tile_cid, cb, exs, nxs);
          const uint32_t cnt = stage_chunk_mm(g, lds, lds->tile_cid, cb, total, Ox, Oy, Oz, dummy, jg);
          float av[4]; load_av(av);
          mm_groups(av, cnt, [&](const int gi, const qn_f4v d) __attribute__((always_inline)) {
            bool hit[4];
#pragma unroll
            for (int e = 0; e < 4; e++) hit[e] = d[e] < thr;
            const uint4 cc = *(const uint4*)&lds->tile_cid[16 * gi + 4 * sub];
            if (!one_cluster) { hit[0] = hit[0] && (cc.x >> 26) == cidm; hit[1] = hit[1] && (cc.y >> 26) == cidm; hit[2] = hit[2] && (cc.z >> 26) == cidm; hit[3] = hit[3] && (cc.w >> 26) == cidm; }
            const uint32_t scv[4] = {cc.x, cc.y, cc.z, cc.w};
            if (hit[0] || hit[1] || hit[2] || hit[3]) {
#pragma unroll
              for (int e = 0; e < 4; e++) if (hit[e]) {
                const uint32_t pos = atomicAdd(&L->cnt[qs], 1u);
                if (pos < HCAP) L->u.list[qs][pos] = (unsigned long long)(scv[e] & 0x03ffffffu);      // (position in pts[]: the exact stage fetches the point from there)
              }
            }
          });
        }
      }, ncand);
      }
      wave_lds_fence();
      // ---- exact stage: the listed candidates with the defining arithmetic
      const uint32_t Pn = min(L->cnt[qs], (uint32_t)HCAP);
      const int pmax = wave_max_i(collect ? (int)Pn : 0);
#pragma unroll
      for (int j = 0; j < HCAP / 4; j++) {
        if (4 * j >= pmax) break;                                                  // (wave-uniform: the longest list of the wave's queries)
        const uint32_t slot = (uint32_t)(sub + 4 * j);
        if (collect && slot < Pn) {
          const float4 p = g.pts[(uint32_t)L->u.list[qs][slot]];
          L->u.list[qs][slot] = pack_key(sqdist(qx, qy, qz, p.x, p.y, p.z), __float_as_uint(p.w));
        }
      }
    } else {
    auto collect_one = [&](const float4& cp) __attribute__((always_inline)) {
      const float d2 = sqdist(qx, qy, qz, cp.x, cp.y, cp.z);
      if (collect && __float_as_uint(d2) < tau_bits) {
        const uint32_t pos = atomicAdd(&L->cnt[qs], 1u);
        if (pos < HCAP) L->u.list[qs][pos] = pack_key(d2, __float_as_uint(cp.w));
      }
    };
    // (the segment table of pass 1 is still in LDS when it fits one: `ncand`)
    if (one_cluster) stream_clusters<4, true>(g, lds, ncl, nseg_all, [&](const float4& cp, bool, uint32_t) __attribute__((always_inline)) { collect_one(cp); }, ncand);
    else stream_clusters<4, true>(g, lds, ncl, nseg_all, [&](const float4& cp, bool in_tile, uint32_t ccid) __attribute__((always_inline)) { if (in_tile && ccid == cid) collect_one(cp); }, ncand);
    }
    wave_lds_fence();
    const uint32_t P = L->cnt[qs];
    const bool ok = collect && P <= HCAP;                                        // (P >= k by construction)
    // ---- rank: own entries sub, sub + 4, ...; every entry of the query's list is compared against them
    const int maxP = wave_max_i(ok ? (int)P : 0);
    unsigned long long own[HCAP / 4]; int rank[HCAP / 4];
#pragma unroll
    for (int j = 0; j < HCAP / 4; j++) { own[j] = L->u.list[qs][sub + 4 * j]; rank[j] = 0; }      // (infinite beyond P; an overflowed list - !ok - is ranked for nothing: its query takes status 2)
    // 4 list entries per step, all four LDS reads issued before the compares (the entry -> compare chain was latency-bound); NJ = own entries a lane can hold for
    // the longest list of the wave (the usual case - k + a few entries, <= 24 - has no own[6], own[7]): one loop per width, chosen once
    auto rank_pass = [&](auto NJ) __attribute__((always_inline)) {
      for (int f = 0; f < maxP; f += 4) {
        unsigned long long kf[4];
#pragma unroll
        for (int u = 0; u < 4; u++) kf[u] = L->u.list[qs][min(f + u, HCAP)];
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
          for (int j = 0; j < decltype(NJ)::value; j++) rank[j] += kf[u] < own[j] ? 1 : 0;
        }
      }
    };
    if (HCAP > 32 && maxP > 32) rank_pass(std::integral_constant<int, HCAP / 4>());
    else if (maxP > 28) rank_pass(std::integral_constant<int, 8>());
    else if (maxP > 24) rank_pass(std::integral_constant<int, 7>());      // (k = 20: the longest list of a wave's 16 queries is 25...28 entries every other time)
    else rank_pass(std::integral_constant<int, 6>());
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
        else {
          const uint32_t tot = s0 + s1 + s2 + s3;
          if (4u * tot < (uint32_t)k) {                                           // an isolated point (a quarter of k within 2 r): do not drag the wave through a
            int kk = k; asm volatile("" : "+s"(kk));                          // (opaque: as a loop invariant (float)k was kept - spilled - across the whole search for this rare branch)
            r = tot > 0 ? 2.f * r * sqrtf((float)kk / (float)tot) * 1.1f : 4.f * r + g.cell;   // big-box round - hand it over with an extrapolated radius (count ~ r^2);
          } else { r = 2.f * r + g.cell; retry = true; }                          // the caller sends radii beyond 2.5 r0 to the one-query-per-wave pass
        }
      } else if (!ok) status = 2;                                                   // list overflow
      else {
        d -= g.eps;
        if (whole || (d > 0.f && kth_d2 < d * d)) {
          status = 0;
          uint32_t row_l = row; asm volatile("" : "+v"(row_l));
          int32_t* __restrict__ idx_out = idx_base + (size_t)row_l * k; float* __restrict__ d2_out = d2_base ? d2_base + (size_t)row_l * k : nullptr;
#pragma unroll
          for (int j = 0; j < HCAP / 4; j++) if (own[j] != QN_INF_KEY && rank[j] < k) {
            idx_out[rank[j]] = (int32_t)key_idx(own[j]);
            if (d2_out) d2_out[rank[j]] = key_d2(own[j]);
          }
        } else { r = fmaxf(sqrtf(kth_d2) * 1.000001f + g.eps, r + g.eps); retry = true; }
      }
    }
    todo = __ballot(retry && round + 1 < max_rounds);
    if (g.dbg && lane == 0 && todo) atomicAdd(&g.dbg[3], (uint32_t)__popcll(todo));
  }
  return status;
}


// ------------------------------------------------------------------ one k-NN query per WAVE
// For the queries whose k-th neighbour is far away (isolated points, sparse far field of a real scan): in the 16-query layout
// such queries form 16 disjoint clusters per wave and every lane walks all 16 candidate streams.  Here all 64 lanes score ONE
// query's candidates (stream_box: one candidate per lane and step) with the same histogram selection: pass 1 bins the squared
// distances into one LDS histogram, tau = first bin edge whose cumulative count reaches k, pass 2 collects the candidates below
// tau (<= QN_HCAP1), each lane ranks up to two of them.  Same exactness argument and certification as wave_knn_hist.
// Returns 0 = done (k indices written in ascending (d2, idx) order), 2 = hand over to the sorted-list path.
#define QN_HCAP1 128
struct WaveLdsH1 {
  WaveLds s;
  uint32_t hist[QN_HB];
  unsigned long long list[QN_HCAP1];
  uint32_t cnt, kth;
  unsigned long long best;          // outputs of the last successful call: smallest key,
  uint32_t second, pad;             // d2 bits of the runner-up (kth: d2 bits of the k-th)
};

__device__ __forceinline__ int wave_knn_single(const GridView& g, float qx, float qy, float qz, float r, int k, WaveLdsH1* L,
                                               int32_t* __restrict__ idx_out, float* __restrict__ d2_out) {
  const int lane = threadIdx.x & 63;
  const float INF = __int_as_float(0x7f800000);
  if (!(qx == qx) || !(qy == qy) || !(qz == qz) || !(r == r)) return 2;
  const CapBox cap = cap_of(g, qx, qy, qz);                               // (all zeros for a query inside the grid box: the plain ball)
  if (cap.S > 0.f) r = fmaxf(r, sqrtf(cap.S) + 0.5f * g.cell);
  const float r_in = r;
  for (int round = 0; round < 64; round++) {
    const float rx = cap_extent(g, cap, r, 0), ry = cap_extent(g, cap, r, 1), rz = cap_extent(g, cap, r, 2);
    int x0 = rfl(cell_coord(qx - rx, g.ox, g.inv_cell, g.nx)), x1 = rfl(cell_coord(qx + rx, g.ox, g.inv_cell, g.nx));
    int y0 = rfl(cell_coord(qy - ry, g.oy, g.inv_cell, g.ny)), y1 = rfl(cell_coord(qy + ry, g.oy, g.inv_cell, g.ny));
    int z0 = rfl(cell_coord(qz - rz, g.oz, g.inv_cell, g.nz)), z1 = rfl(cell_coord(qz + rz, g.oz, g.inv_cell, g.nz));
    const bool tile_mode = ((x1 >> 3) - (x0 >> 3) + 1) * (y1 - y0 + 1) * (z1 - z0 + 1) > 128;
    if (tile_mode) {                                                          // whole tiles: the certification below sees the larger scanned box
      x0 = (x0 >> 3) << 3; x1 = min(((x1 >> 3) << 3) + 7, g.nx - 1);
      y0 = (y0 >> 2) << 2; y1 = min(((y1 >> 2) << 2) + 3, g.ny - 1);
      z0 = (z0 >> 2) << 2; z1 = min(((z1 >> 2) << 2) + 3, g.nz - 1);
    }
    wave_lds_fence();
    L->hist[lane] = 0; if (lane == 0) { L->cnt = 0; L->kth = 0x7f800000u; }
    wave_lds_fence();
    const int base = (int)(__float_as_uint(4.f * r * r) >> 20) - (QN_HB - 2);
    stream_box(g, x0, x1, y0, y1, z0, z1, tile_mode, &L->s, [&](const float4& p, bool valid, uint32_t) __attribute__((always_inline)) {
      const int bin = max((int)(__float_as_uint(sqdist(qx, qy, qz, p.x, p.y, p.z)) >> 20) - base, 0);
      if (valid && bin < QN_HB - 1) atomicAdd(&L->hist[bin], 1u);
    });
    wave_lds_fence();
    const uint32_t hv = lane < QN_HB - 1 ? L->hist[lane] : 0u;
    const uint32_t cum = wave_incl_scan_u32(hv, lane);
    const uint32_t total = rdlane(cum, 63);
    const bool enough = total >= (uint32_t)k;
    const unsigned long long reach = __ballot(cum >= (uint32_t)k);
    const int cross = reach ? __ffsll((long long)reach) - 1 : QN_HB - 2;
    uint32_t tau_bits = (uint32_t)(base + cross + 1) << 20;
    // A far query's crossing bin can hold hundreds of points (the bins are 9 % wide in d2): refine it with 64 linear sub-bins
    // (bits 14..19 of the f32 pattern) so that the list of pass 2 stays short.
    if (enough && rdlane(cum, cross) > (uint32_t)QN_HCAP1 && cross > 0) {
      const uint32_t below = rdlane(cum, cross - 1), need = (uint32_t)k - below;       // below < k: all of them are among the k nearest
      const uint32_t hi20 = (uint32_t)(base + cross);
      wave_lds_fence();
      L->hist[lane] = 0;
      wave_lds_fence();
      stream_box(g, x0, x1, y0, y1, z0, z1, tile_mode, &L->s, [&](const float4& p, bool valid, uint32_t) __attribute__((always_inline)) {
        const uint32_t bits = __float_as_uint(sqdist(qx, qy, qz, p.x, p.y, p.z));
        if (valid && (bits >> 20) == hi20) atomicAdd(&L->hist[(bits >> 14) & 63u], 1u);
      });
      wave_lds_fence();
      const uint32_t cum2 = wave_incl_scan_u32(L->hist[lane], lane);
      const unsigned long long reach2 = __ballot(cum2 >= need);
      const int sub = reach2 ? __ffsll((long long)reach2) - 1 : 63;
      tau_bits = (hi20 << 20) + ((uint32_t)(sub + 1) << 14);
    }
    float d = INF;                                                          // nearest face of the scanned box with unseen cells behind it
    if (x0 > 0) d = fminf(d, cap_face_dist(cap, qx - (g.ox + x0 * g.cell), 0));
    if (x1 < g.nx - 1) d = fminf(d, cap_face_dist(cap, (g.ox + (x1 + 1) * g.cell) - qx, 0));
    if (y0 > 0) d = fminf(d, cap_face_dist(cap, qy - (g.oy + y0 * g.cell), 1));
    if (y1 < g.ny - 1) d = fminf(d, cap_face_dist(cap, (g.oy + (y1 + 1) * g.cell) - qy, 1));
    if (z0 > 0) d = fminf(d, cap_face_dist(cap, qz - (g.oz + z0 * g.cell), 2));
    if (z1 < g.nz - 1) d = fminf(d, cap_face_dist(cap, (g.oz + (z1 + 1) * g.cell) - qz, 2));
    const bool whole = d == INF;
    if (!enough) {
      if (whole) return 2;                                                   // fewer than k points within 2 r of the whole cloud: general path
      if (r_in > 6.f * g.cell) r += (1.f + round) * g.cell;                  // a far query (its k nearest sit within centimetres to decimetres of each other): grow by cells
      else r = total > 0 ? fmaxf(2.f * r * sqrtf((float)k / (float)total) * 1.1f, r + g.cell) : 2.f * r + g.cell;
      continue;
    }
    stream_box(g, x0, x1, y0, y1, z0, z1, tile_mode, &L->s, [&](const float4& p, bool valid, uint32_t) __attribute__((always_inline)) {
      const float d2 = sqdist(qx, qy, qz, p.x, p.y, p.z);
      if (valid && __float_as_uint(d2) < tau_bits) {
        const uint32_t pos = atomicAdd(&L->cnt, 1u);
        if (pos < QN_HCAP1) L->list[pos] = pack_key(d2, __float_as_uint(p.w));
      }
    });
    wave_lds_fence();
    const uint32_t P = L->cnt;
    if (P > QN_HCAP1) return 2;                                             // too many ties below tau: general path
    const unsigned long long own0 = (uint32_t)lane < P ? L->list[lane] : QN_INF_KEY, own1 = (uint32_t)lane + 64u < P ? L->list[lane + 64] : QN_INF_KEY;
    int rank0 = 0, rank1 = 0;
    for (uint32_t f = 0; f < P; f++) { const unsigned long long kf = L->list[f]; rank0 += kf < own0 ? 1 : 0; rank1 += kf < own1 ? 1 : 0; }
    if (own0 != QN_INF_KEY && rank0 == k - 1) L->kth = (uint32_t)(own0 >> 32);
    if (own1 != QN_INF_KEY && rank1 == k - 1) L->kth = (uint32_t)(own1 >> 32);
    wave_lds_fence();
    const float kth_d2 = __uint_as_float(L->kth);
    const float df = d - g.eps;
    if (whole || (df > 0.f && kth_d2 < df * df)) {
      if (own0 != QN_INF_KEY && rank0 < k) { idx_out[rank0] = (int32_t)key_idx(own0); if (d2_out) d2_out[rank0] = key_d2(own0); }
      if (own1 != QN_INF_KEY && rank1 < k) { idx_out[rank1] = (int32_t)key_idx(own1); if (d2_out) d2_out[rank1] = key_d2(own1); }
      if (lane == 0) { L->second = 0x7f800000u; }
      wave_lds_fence();
      if (own0 != QN_INF_KEY && rank0 == 0) L->best = own0;
      if (own1 != QN_INF_KEY && rank1 == 0) L->best = own1;
      if (own0 != QN_INF_KEY && rank0 == 1) L->second = (uint32_t)(own0 >> 32);
      if (own1 != QN_INF_KEY && rank1 == 1) L->second = (uint32_t)(own1 >> 32);
      wave_lds_fence();
      return 0;
    }
    r = fmaxf(sqrtf(kth_d2) * 1.000001f + g.eps, r + g.eps);                        // the k-th best is known: the next ball certifies
  }
  return 2;
}

}  // namespace qn
