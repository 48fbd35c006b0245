// qn_feat_mm.cuh - K12 on the matrix cores: exact nearest neighbour in FPFH space (33-D, f32 sequential sum, ties -> lowest index;
// Matcher::searchKDTree) by SCREENING with v_mfma_f32_32x32x16_f16 and re-evaluating the survivors with the defining arithmetic.
//
//   d(q, c) = |q - c|^2 = |q'|^2 - 2 S(q, c),   S = q'.c' - |c'|^2 / 2,   x' = x - P  (P = the FPFH row of an exact plane: 100 in bins 5, 16, 27 -
//   the mode of every structured scene; any P is CORRECT, it only decides how tight the bound below is where the rows are dense)
//
// For a fixed query the nearest candidate maximises S.  Rows are split x' = h + l + e (h, l in f16, |e| <= 2^-24 |x'|) and laid out along K = 112:
//        candidate row  [ h(33) n1 n2 n3 | l(33) b|c'|^2+g|c'|  |c'|     1      | h(33) 0 0 0 | 0 0 0 0 ]      n1 + n2 + n3 = -|c'|^2 / 2 (three f16 pieces)
//        query row      [ h(33) 1  1  1  | h(33) s              s a|q'|  s g|q'| | l(33) 0 0 0 | 0 0 0 0 ]      s = -1 (pass 1) / +1 (pass 2)
// so one chain of 7 MFMAs per 32 x 32 tile yields  S~ -/+ delta  with  delta(q, c) = a |q'||c'| + b |c'|^2 + g (|q'| + |c'|)  an upper bound of |S~ - S|:
//        split:       |q'.c' - (hh + hl + lh)| <= 3.1 x 2^-24 |q'||c'|
//        matrix core: each instruction returns C + (16 products) rounded ONCE (measured: tools/micro/mfma_probe.hip - f16 subnormals kept, the
//                     sum carried wider than f32); assumed with an 8 x margin: 2^-21 (|C| + sum |products|) per instruction, 7 instructions
//        subnormal low pieces (|x'| < 0.25): absolute error <= 2^-25 per element, <= 2^-25 sqrt(33) (|q'| + |c'|) on the product
//        =>  a = 3.6e-6, b = 1.7e-6, g = 2e-7  (the f16 images of the bound terms are rounded UP)
// Pass 1:  L(q) = max (S~ - delta) over a SAMPLE of the candidates (every QN_MM_SAMPLE-th tile)  <= S of the best candidate.
// Pass 2:  every c with  S~ + delta >= L(q) - X(q)  is a SURVIVOR;  X = 2.3e-6 (|q'|^2 - 2 L) covers the rounding of the defining f32 sum itself
//          (36 x 2^-24 relative on both candidates compared).  The defining nearest neighbour is always a survivor ("feature matching").
//          (Sampling makes pass 1 cost 1 / QN_MM_SAMPLE of pass 2; the price is ~QN_MM_SAMPLE survivors per query - the candidates that beat the best of
//          the sample - instead of ~1.)
// Exact:   survivors are evaluated with the defining arithmetic; 64-bit atomicMin on (distance bits, index).  Bit-identical results.
// Duplicate rows (every point of an exact plane has the same FPFH row) are removed from the CANDIDATE side first: a hash table keeps the lowest
// index of each distinct row, the others become dead rows (S~ = -196512, below every live value) - their distance is the representative's and
// their index is higher, so they can never win.  Dead rows also stand for NaN rows and tile padding.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qn {

#define QN_MM_KS 7                      // K = 112 = 7 x 16
#define QN_MM_QT 4                      // query tiles (32 queries each) per wave
#define QN_MM_WAVES 4                   // waves per block: 512 queries per block
#define QN_MM_SAMPLE 4                  // pass 1 visits every 4th candidate tile (measured: 2 -> 3.00, 4 -> 2.79, 8 -> 3.03, 16 -> 5.2 ms at 100k)
#define QN_MM_ALPHA 3.6e-6
#define QN_MM_BETA 1.7e-6
#define QN_MM_GAMMA 2.0e-7             // absolute part: an f16 low piece in the subnormal range is off by up to 2^-25 per element whatever the element's size
#define QN_MM_DEAD (-65504.0f)          // x 3 pieces
#define QN_MM_EMPTY 0xFFFFFFFFFFFFFFFFull
typedef _Float16 qn_h8 __attribute__((ext_vector_type(8)));
typedef uint32_t qn_u4 __attribute__((ext_vector_type(4)));      // a fragment as the kernels hold it: raw 128 bits (copies of f16 VECTORS were compiled to per-element shifts and v_perm)
typedef float qn_f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t mm_enc(float f) { const uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float mm_dec(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }
// max of the 16 results of a tile and a seed (plain fmaxf: an inline-asm v_max3_f32 here reads the MFMA result registers without the wait
// states the compiler's hazard recognizer inserts for its own instructions - measured: wrong maxima)
__device__ __forceinline__ float mm_max16(const float __attribute__((ext_vector_type(16)))& v, float seed) {
  float x = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
  x = fmaxf(x, fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
  x = fmaxf(x, fmaxf(fmaxf(v[8], v[9]), fmaxf(v[10], v[11])));
  x = fmaxf(x, fmaxf(fmaxf(v[12], v[13]), fmaxf(v[14], v[15])));
  return fmaxf(seed, x);
}
__device__ __forceinline__ float mm_plane(int d) { return (d == 5 || d == 16 || d == 27) ? 100.f : 0.f; }

// ---- candidate de-duplication: open addressing, entry = (row hash << 32 | lowest index of that row); rows compared bit for bit
static __global__ void k_feat_dedupe(const float* __restrict__ rows, uint32_t n, unsigned long long* __restrict__ table, uint32_t mask, uint32_t* __restrict__ plane_min /* zeroed */) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t* r = (const uint32_t*)(rows + (size_t)min(i, n - 1) * QN_FROW);
  // the exact-plane row (half of a synthetic cloud, every flat patch of a real one) does not go through the table - tens of thousands of
  // atomics on one slot took 0.3 ms: the lowest lane of each wave that holds one reports its index (plane_min keeps ~min as a max)
  bool plane = i < n;
#pragma unroll
  for (int d = 0; d < 33; d++) plane = plane && (__uint_as_float(r[d]) == mm_plane(d));
  const unsigned long long pw = __ballot(plane);
  if (pw != 0ull && (int)(threadIdx.x & 63) == __ffsll((long long)pw) - 1 && ~*plane_min > i) atomicMax(plane_min, ~i);
  if (i >= n || plane) return;
  if ((r[0] & 0x7fffffffu) > 0x7f800000u) return;                      // NaN row (k_fpfh marks a dead point by NaN in bin 0)
  const uint32_t h = r[34];
  const unsigned long long mine = ((unsigned long long)h << 32) | i;
  for (uint32_t slot = h & mask;; slot = (slot + 1) & mask) {
    unsigned long long e = table[slot];
    if (e == QN_MM_EMPTY) { e = atomicCAS(&table[slot], QN_MM_EMPTY, mine); if (e == QN_MM_EMPTY) return; }
    if ((uint32_t)(e >> 32) == h) {
      const uint32_t* o = (const uint32_t*)(rows + (size_t)(uint32_t)e * QN_FROW);
      bool same = true;
#pragma unroll
      for (int d = 0; d < 33; d++) same = same && (o[d] == r[d]);
      if (same) { if ((uint32_t)e > i) atomicMin(&table[slot], mine); return; }      // (most duplicates find a lower index already there: no atomic on the hot slot)
    }
  }
}
// the representative (lowest index) of row i's distinct row, or 0xffffffff for a NaN row (never inserted)
__device__ __forceinline__ uint32_t feat_rep_of(const float* __restrict__ rows, uint32_t i, const unsigned long long* __restrict__ table, uint32_t mask, const uint32_t* __restrict__ plane_min) {
  const uint32_t* r = (const uint32_t*)(rows + (size_t)i * QN_FROW);
  bool plane = true;
#pragma unroll
  for (int d = 0; d < 33; d++) plane = plane && (__uint_as_float(r[d]) == mm_plane(d));
  if (plane) return ~*plane_min;
  if ((r[0] & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;
  const uint32_t h = r[34];
  for (uint32_t slot = h & mask;; slot = (slot + 1) & mask) {
    const unsigned long long e = table[slot];
    if (e == QN_MM_EMPTY) return i;                                     // (not reached for an inserted row)
    if ((uint32_t)(e >> 32) == h) {
      if ((uint32_t)e == i) return i;
      const uint32_t* o = (const uint32_t*)(rows + (size_t)(uint32_t)e * QN_FROW);
      bool same = true;
#pragma unroll
      for (int d = 0; d < 33; d++) same = same && (o[d] == r[d]);
      if (same) return (uint32_t)e;
    }
  }
}
// QUERY-side de-duplication: rows that are bit for bit equal have the same nearest neighbour (same distances, ties to the lowest candidate index either way), so
// only the lowest index of every distinct query row is searched (k_query_reps builds that list) and the others copy its key afterwards (k_query_propagate).
// Every point of an exact plane has the same FPFH row - 57 % of a noise-free synthetic source; a noisy real cloud has next to none, and pays two small kernels.
static __global__ void k_query_reps(const float* __restrict__ rows, uint32_t n, const unsigned long long* __restrict__ table, uint32_t mask, const uint32_t* __restrict__ plane_min,
                                    uint32_t* __restrict__ list, uint32_t* __restrict__ count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool rep = i < n && feat_rep_of(rows, i, table, mask, plane_min) == i;
  const unsigned long long m = __ballot(rep);
  if (m == 0ull) return;
  const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(m));
  base = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl(base, leader));
  if (rep) list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
}
static __global__ void k_query_propagate(const float* __restrict__ rows, uint32_t n, const unsigned long long* __restrict__ table, uint32_t mask, const uint32_t* __restrict__ plane_min,
                                         unsigned long long* __restrict__ keys) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t r = feat_rep_of(rows, i, table, mask, plane_min);
  if (r != i && r != 0xffffffffu) keys[i] = keys[r];
}
__device__ __forceinline__ bool feat_is_rep(const float* __restrict__ rows, uint32_t i, const unsigned long long* __restrict__ table, uint32_t mask, const uint32_t* __restrict__ plane_min) {
  const uint32_t* r = (const uint32_t*)(rows + (size_t)i * QN_FROW);
  bool plane = true;
#pragma unroll
  for (int d = 0; d < 33; d++) plane = plane && (__uint_as_float(r[d]) == mm_plane(d));
  if (plane) return ~*plane_min == i;
  const uint32_t h = r[34];
  for (uint32_t slot = h & mask;; slot = (slot + 1) & mask) {
    const unsigned long long e = table[slot];
    if (e == QN_MM_EMPTY) return true;                                  // (not reached for an inserted row)
    if ((uint32_t)(e >> 32) == h) {
      if ((uint32_t)e == i) return true;
      const uint32_t* o = (const uint32_t*)(rows + (size_t)(uint32_t)e * QN_FROW);
      bool same = true;
#pragma unroll
      for (int d = 0; d < 33; d++) same = same && (o[d] == r[d]);
      if (same) return false;
    }
  }
}

// ---- operand images.  Tile-major: element (row r, k) of a set sits at  (((r / 32) * 7 + k / 16) * 64 + (r % 32) + 32 * ((k % 16) / 8)) * 8 + k % 8,
// i.e. the 16 bytes lane l of a wave needs for k-step ks of tile t are at  ((t * 7 + ks) * 64 + l) * 16: every fragment load is one coalesced KB.
// One thread per row (padding rows of the last tile included).  side 0: candidate rows (index = row), side 1: query slots (row = qlist[slot]).
__device__ __forceinline__ _Float16 mm_up(double v) { return (_Float16)(float)(v * (1.0 + 1.0 / 1024.0) + 1e-7); }      // f16 image >= v (v >= 0)
static __global__ void k_feat_prep(const float* __restrict__ rows, uint32_t n, const uint32_t* __restrict__ qlist, const uint32_t* __restrict__ qlist_n, int side,
                                   const unsigned long long* __restrict__ table, uint32_t mask, const uint32_t* __restrict__ plane_min, _Float16* __restrict__ out, float* __restrict__ qn_up, uint32_t n_pad) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_pad) return;
  const uint32_t count = (side == 1 && qlist) ? *qlist_n : n;
  bool live = s < count;
  uint32_t row = s;
  if (live && side == 1 && qlist) row = qlist[s];
  double xp[33]; double nn = 0;
  if (live) {
    const float* r = rows + (size_t)row * QN_FROW;
    live = r[0] == r[0];
    if (live && side == 0 && table) live = feat_is_rep(rows, row, table, mask, plane_min);
#pragma unroll
    for (int d = 0; d < 33; d++) { xp[d] = live ? (double)r[d] - (double)mm_plane(d) : 0.0; nn += xp[d] * xp[d]; }
  } else {
#pragma unroll
    for (int d = 0; d < 33; d++) xp[d] = 0.0;
  }
  _Float16 v[112];
#pragma unroll
  for (int k = 0; k < 112; k++) v[k] = (_Float16)0.f;
#pragma unroll
  for (int d = 0; d < 33; d++) {
    const _Float16 h = (_Float16)(float)xp[d];
    const _Float16 l = (_Float16)(float)(xp[d] - (double)(float)h);
    v[d] = h;
    if (side == 0) { v[36 + d] = l; v[72 + d] = h; } else { v[36 + d] = h; v[72 + d] = l; }
  }
  if (side == 0) {
    if (live) {
      const double half = -0.5 * nn;
      const _Float16 n1 = (_Float16)(float)half; const double r1 = half - (double)(float)n1;
      const _Float16 n2 = (_Float16)(float)r1; const double r2 = r1 - (double)(float)n2;
      v[33] = n1; v[34] = n2; v[35] = (_Float16)(float)r2;
      v[69] = mm_up(QN_MM_BETA * nn + QN_MM_GAMMA * sqrt(nn)); v[70] = mm_up(sqrt(nn)); v[71] = (_Float16)1.f;
    } else { v[33] = (_Float16)QN_MM_DEAD; v[34] = (_Float16)QN_MM_DEAD; v[35] = (_Float16)QN_MM_DEAD; }
  } else {
    v[33] = (_Float16)1.f; v[34] = (_Float16)1.f; v[35] = (_Float16)1.f;
    v[69] = (_Float16)1.f; v[70] = mm_up(QN_MM_ALPHA * sqrt(nn)); v[71] = mm_up(QN_MM_GAMMA * sqrt(nn));
    qn_up[s] = live ? (float)(nn * (1.0 + 1e-6)) : __int_as_float(0x7fc00000);       // NaN: this slot admits nothing
  }
  const uint32_t tile = s >> 5, rr = s & 31;
  qn_h8* o = (qn_h8*)out;
#pragma unroll
  for (int ks = 0; ks < QN_MM_KS; ks++)
#pragma unroll
    for (int half = 0; half < 2; half++) {
      qn_h8 w;
#pragma unroll
      for (int j = 0; j < 8; j++) w[j] = v[ks * 16 + half * 8 + j];
      o[((size_t)tile * QN_MM_KS + ks) * 64 + rr + 32 * half] = w;
    }
}

// ---- the screening passes.  Candidates are the A operand (rows), queries the B operand (columns): in the 32 x 32 result a lane holds ONE query
// (column lane & 31) and 16 candidates (rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)), so the running maximum / the threshold is a per-lane scalar.
// A wave keeps the fragments of QN_MM_QT query tiles in registers (112 VGPRs) and streams the candidate tiles of its grid.y segment; each
// candidate fragment (one coalesced KB per k-step) feeds QN_MM_QT x 7 MFMAs.
template <int PASS>
static __global__ void __launch_bounds__(64 * QN_MM_WAVES, 2) k_feat_mm(const qn_u4* __restrict__ Qm, uint32_t nq_max, const uint32_t* __restrict__ qcount_p,
                                                                       const qn_u4* __restrict__ Cm, uint32_t nc_tiles, uint32_t tiles_per_seg, uint32_t tile_step,
                                                                       uint32_t* __restrict__ Lq, const float* __restrict__ qn_up, uint2* __restrict__ pairs,
                                                                       uint32_t* __restrict__ counts, uint32_t cap_block, uint2* __restrict__ spill, uint32_t cap_spill) {
  // survivors of this block go to its own region pairs[block * cap_block ..] through a counter in LDS (one global counter for the whole grid
  // serialises a million atomics on one address: measured 3 ms); what does not fit there goes to the shared spill list (global counter counts[3]).
  // counts[0] = all survivors, counts[1] = lost survivors (the caller repeats the search with the VALU kernel), counts[4 + block] = the block's count
  __shared__ uint32_t bcnt;
  const uint32_t blk = blockIdx.y * gridDim.x + blockIdx.x;
  if (PASS == 2) { if (threadIdx.x == 0) bcnt = 0; __syncthreads(); }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const uint32_t nqueries = qcount_p ? *qcount_p : nq_max;
  const uint32_t nq_tiles = (nqueries + 31) >> 5;
  const uint32_t qt0 = (blockIdx.x * QN_MM_WAVES + wid) * QN_MM_QT;
  const bool idle = qt0 >= nq_tiles;                                      // (no early return in pass 2: the block meets again at the end)
  if (idle && PASS == 1) return;
  if (!idle) {
  qn_u4 bq[QN_MM_QT][QN_MM_KS];
#pragma unroll
  for (int u = 0; u < QN_MM_QT; u++)
#pragma unroll
    for (int ks = 0; ks < QN_MM_KS; ks++) {
      const uint32_t t = min(qt0 + u, nq_tiles - 1);                        // a tile past the end repeats the last one (its results are dropped)
      bq[u][ks] = Qm[((size_t)t * QN_MM_KS + ks) * 64 + lane];
    }
  if (PASS == 1 && lane < 32) {                                           // k = 69, 70, 71: the delta terms enter with a minus sign
#pragma unroll
    for (int u = 0; u < QN_MM_QT; u++) { bq[u][4][2] ^= 0x80000000u; bq[u][4][3] ^= 0x80008000u; }      // f16 elements 5, 6 and 7 of the fragment
  }
  float m[QN_MM_QT];                                                     // pass 1: running max; pass 2: the admission threshold
#pragma unroll
  for (int u = 0; u < QN_MM_QT; u++) {
    m[u] = __int_as_float(0xff800000);
    if (PASS == 2) {
      const uint32_t slot = (qt0 + u) * 32 + (lane & 31);
      float thr = __int_as_float(0x7f800000);
      if (qt0 + u < nq_tiles && slot < nqueries) {
        const float L = mm_dec(Lq[slot]); const float qn = qn_up[slot];
        if (qn == qn) thr = L > -150000.f ? L - 2.3e-6f * fmaxf(qn - 2.f * L, 0.f) * 1.01f - 2.4e-7f * fabsf(L) - 1e-30f
                                          : -150000.f;                      // no live candidate in the sample: every live candidate survives
      }
      m[u] = thr;
    }
  }
  // the segment's tiles, every tile_step-th one (pass 1 looks at a SAMPLE of the candidates: any lower bound L is a valid one)
  const uint32_t t0 = blockIdx.y * tiles_per_seg * tile_step, t1 = min(nc_tiles, t0 + tiles_per_seg * tile_step);
  qn_u4 a[QN_MM_KS], an[QN_MM_KS];
  if (t0 < t1) {
#pragma unroll
    for (int ks = 0; ks < QN_MM_KS; ks++) an[ks] = Cm[((size_t)t0 * QN_MM_KS + ks) * 64 + lane];
  }
  for (uint32_t t = t0; t < t1; t += tile_step) {
#pragma unroll
    for (int ks = 0; ks < QN_MM_KS; ks++) a[ks] = an[ks];
    if (t + tile_step < t1) {
#pragma unroll
      for (int ks = 0; ks < QN_MM_KS; ks++) an[ks] = Cm[((size_t)(t + tile_step) * QN_MM_KS + ks) * 64 + lane];
    }
    // the four query tiles' MFMA chains are independent (the compiler interleaves them: a dependent chain alone runs the matrix pipe at half
    // rate); in pass 2 ONE branch per candidate tile decides whether any survivor exists - a branch per chain serialises the chains
    qn_f16v acc[QN_MM_QT];
#pragma unroll
    for (int u = 0; u < QN_MM_QT; u++) {
      acc[u] = (qn_f16v){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < QN_MM_KS; ks++) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(qn_h8, a[ks]), __builtin_bit_cast(qn_h8, bq[u][ks]), acc[u], 0, 0, 0);
      if (PASS == 1) m[u] = mm_max16(acc[u], m[u]);
    }
    if (PASS == 2) {
      bool hu[QN_MM_QT], hit = false;
#pragma unroll
      for (int u = 0; u < QN_MM_QT; u++) { hu[u] = mm_max16(acc[u], __int_as_float(0xff800000)) >= m[u]; hit = hit || hu[u]; }
      if (__ballot(hit) != 0ull) {                                        // rare: some lane of the wave has a survivor in this tile
#pragma unroll
        for (int u = 0; u < QN_MM_QT; u++) {
          if (__ballot(hu[u]) == 0ull) continue;
          const uint32_t slot = (qt0 + u) * 32 + (lane & 31);
          uint32_t bits = 0;                                              // this lane's survivors of the tile as a 16-bit mask, then one loop
#pragma unroll                                                            // iteration per survivor (a compare-and-branch per result cost as much
          for (int r = 0; r < 16; r++) bits |= (acc[u][r] >= m[u] ? 1u : 0u) << r;      // as two tiles of MFMAs every time the path was entered)
          while (bits != 0u) {
            const int r = __ffs((int)bits) - 1; bits &= bits - 1u;
            const uint32_t cand = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const uint32_t pos = atomicAdd(&bcnt, 1u);
            if (pos < cap_block) pairs[(size_t)blk * cap_block + pos] = make_uint2(slot, cand);
            else { const uint32_t g = atomicAdd(&counts[3], 1u); if (g < cap_spill) spill[g] = make_uint2(slot, cand); else counts[1] = 1u; }
          }
        }
      }
    }
  }
  if (PASS == 1) {
#pragma unroll
    for (int u = 0; u < QN_MM_QT; u++) {
      const float x = fmaxf(m[u], __shfl_xor(m[u], 32));
      const uint32_t slot = (qt0 + u) * 32 + (lane & 31);
      if (lane < 32 && qt0 + u < nq_tiles && slot < nqueries && x == x) atomicMax(&Lq[slot], mm_enc(x));
    }
  }
  }
  if (PASS == 2) {
    __syncthreads();
    if (threadIdx.x == 0) { const uint32_t n = bcnt; counts[4 + blk] = min(n, cap_block); if (n) atomicAdd(&counts[0], n); }
  }
}

// ---- survivors: the defining arithmetic (f32, bins in order, no contraction), winner by 64-bit atomicMin on (distance bits << 32 | candidate index)
static __global__ void k_feat_exact(const uint2* __restrict__ pairs, const uint32_t* __restrict__ counts, uint32_t cap_block, uint32_t nregions, const uint2* __restrict__ spill,
                                    uint32_t cap_spill, const float* __restrict__ Q, const uint32_t* __restrict__ qlist, const float* __restrict__ C, uint32_t nc,
                                    unsigned long long* __restrict__ best_key, uint32_t* __restrict__ overflow) {
  if (counts[1] && blockIdx.x == 0 && threadIdx.x == 0) *overflow = 1u;
  // blocks [0, nregions): one region each; the blocks behind them share the spill list
  const bool reg = blockIdx.x < nregions;
  const uint32_t m = reg ? counts[4 + blockIdx.x] : min(counts[3], cap_spill);
  const uint32_t i0 = reg ? threadIdx.x : (blockIdx.x - nregions) * blockDim.x + threadIdx.x, di = reg ? blockDim.x : (gridDim.x - nregions) * blockDim.x;
  for (uint32_t i = i0; i < m; i += di) {
    const uint2 pr = reg ? pairs[(size_t)blockIdx.x * cap_block + i] : spill[i];
    if (pr.y >= nc) continue;
    const uint32_t qi = qlist ? qlist[pr.x] : pr.x;
    const float4* q = (const float4*)(Q + (size_t)qi * QN_FROW); const float4* c = (const float4*)(C + (size_t)pr.y * QN_FROW);
    float e = 0.f;
#pragma unroll
    for (int v = 0; v < 9; v++) {
      const float4 a = q[v], b = c[v];
      { const float t = a.x - b.x; e = e + t * t; }
      if (4 * v + 1 < 33) { const float t = a.y - b.y; e = e + t * t; }
      if (4 * v + 2 < 33) { const float t = a.z - b.z; e = e + t * t; }
      if (4 * v + 3 < 33) { const float t = a.w - b.w; e = e + t * t; }
    }
    if (e == e) atomicMin(&best_key[qi], ((unsigned long long)__float_as_uint(e) << 32) | pr.y);
  }
}

// developer check ("feat_verify"): the matrix-core search against the VALU search, every query
static __global__ void k_feat_compare(const unsigned long long* __restrict__ a, const unsigned long long* __restrict__ b, uint32_t n, const uint32_t* __restrict__ qlist,
                                      const uint32_t* __restrict__ qlist_n, uint32_t* __restrict__ out /* mismatches, first + 1, queries */) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t cnt = qlist ? *qlist_n : n;
  if (s >= cnt) return;
  const uint32_t i = qlist ? qlist[s] : s;
  atomicAdd(&out[2], 1u);
  if (a[i] != b[i]) { atomicAdd(&out[0], 1u); atomicMax(&out[1], i + 1); }
}

}  // namespace qn
