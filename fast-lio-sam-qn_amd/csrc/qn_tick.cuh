// qn_tick.cuh - ONE kernel per optimiser tick in the tracked regime of NanoGICP::align() (call site
// fast_lio_sam_qn/src/loop_closure.cpp:124; LsqRegistration loop restated in SURVEY.md A.1.4-A.1.5).
//
// Round 1 ran  [k_nn_track, k_solve]  per Gauss-Newton iteration and eight kernels per LM iteration; k_solve - one thread doing the
// controller between two launches - was 20 % of the GPU time at 0.14 % of the roof.  k_tick removes that launch and
// the serial hop from the chain:
//
//   prologue  every block reduces the previous tick's partial rows (rows x 28 f64, L2/MALL resident) in a fixed order and thread 0 runs the
//             LM / GN controller on an LDS copy of the optimiser state - redundantly in every block, bit-identical by construction;
//             block 0 publishes the new state.  The loads that do not depend on the pose (point, tracking seed, its target record,
//             normal) are ISSUED BEFORE the prologue, so the gather latency hides behind the controller.
//   body      phase 0 (linearise): tracked / bound-pruned exact 1-NN of every source point at the new pose + its contribution to the
//             28 sums (J^T M J, J^T M e, e^T M e);  phase 1 (LM trial): the error at the trial pose with the cached correspondences;
//   epilogue  the block's 28 sums through an LDS transpose (28 ds_write + 32 ds_read per thread instead of 28 x 6 shuffle steps),
//             one row of the OTHER partial buffer (double buffered with the state: no block waits for another).
//
// PLANE-regularised covariances are exactly C = I - 0.999 n n^T, so a point carries its normal (3 f64) instead of six
// covariance entries:  C_B + R C_A R^T = 2 I - 0.999 (n_B n_B^T + m m^T),  m = R n_A.
#pragma once
#include "qn_gicp_kernels.cuh"

namespace qn {


struct TickArgs {
  GridView src, tgt;
  TailArgs tail;                       // tail.st_in: the state this launch runs under; tail.st_out / cfg / trace / ticket: the controller step its last block runs (enabled = 0: rows left
                                       // for k_far_reduce - the far-query refresh adds a row first - or, MODE 1, nothing to step).  The persistent kernel reads st_in / st_out / cfg / trace from here too.
  double* part_out;                    // this launch's partial rows (one per block)
  double thr2;
  int32_t* nn_idx; float4* nn_ref;
  const double* nrm_s;                 // [n][3] source normals, cell-sorted order
  const TargetRec* tgt_rec;
  uint32_t ppt;                        // source points per thread AND ROW (1 up to 131072 points)
  uint32_t rpb, rows;                  // k_tick: a block forms `rpb` consecutive rows one after the other (batch members: fewer, longer blocks - their prologue, barrier ladder and ticket once per
                                       // rpb x 512 points - while the rows, hence every bit downstream, stay those of rpb = 1); rows = ceil(n / (block size x ppt))
  // far queries (neighbour several cells away: no overlap there, occlusion): candidate cache + refresh requests (below)
  int far_mode;                        // 0: resolve big balls in the kernel; 1: cache, misses go to k_far (request bits); 2: cache, misses resolved in the kernel
  const float4* tgt_raw;
  int32_t* cand; float4* cand_ref;     // [n][QN_FAR_M] candidate indices in ascending distance from q_ref, [n] (q_ref, bound); cell-sorted source order
  float2* cand_b;                      // [n] prefix bounds of the list: everything but its first 4 / first 16 candidates is at least this far from q_ref
  unsigned long long* far_req;         // [ceil(n / 64)] request bits, one word per 64 consecutive source positions
  uint32_t* far_stats;                 // [0] refresh requests, [1] cache hits
  // MODE 1 (closing pass)
  float4* aligned; double* fit_psum; uint32_t* fit_pcnt;
  unsigned long long* clk;              // developer probe (null in production): wall_clock64() stamps of block 0, 8 per launch
  unsigned long long* clk_blk;          // ... and [start, +prologue, +nn, end] of every block of the latest tick launch
};

// ------------------------------------------------------------------ far queries
// A source point whose nearest target point is many cells away (the clouds do not overlap there, or the spot is occluded in the target)
// defeats bound pruning: its runner-up is practically as close as its neighbour (for a neighbour at distance d on a surface of density
// rho the m-th nearest lies only  m / (2 pi rho d)  further), so the inequality  d(q, p_j0) + delta < d_other  never holds and the ball
// around the query - mostly empty space - is searched again at every iteration.  With 20 % of the source outside the target that
// made one tick cost 145 us instead of 10 (profiles/r2_*).  Instead such a query keeps its QN_FAR_M nearest target points and the
// radius B of the ball they were collected from (ONE pass over the ball around the query, wave_ball_collect): every point NOT in the
// list is at least B away from where the list was made, so after a move by delta the true nearest neighbour is the best of the M candidates whenever
// best_distance + delta < B - exact, and M gathers instead of a search.  The margin grows with M (0.1 m at d = 10 m for M = 64) and is sized to the step the query just made;
// a miss (early iterations, big steps) requests a refresh, served chip-wide by k_far, one query per wave.
#define QN_FAR_M 64
#define QN_FAR_RMIN_CELLS 6.f           // a neighbour farther than this many cells makes a query "far"
#define QN_FAR_BLOCKS 512            // x 8 waves: the refreshes are latency-bound (dependent LDS / global round trips), they need many waves in flight
#define QN_FAR_THREADS 512
#define QN_FAR_WORDS 4096            // request words ranked in LDS (clouds of up to 262144 points; larger ones keep the word-per-block distribution)


// Block-level sum of the 28 per-thread accumulators.  Per wave: an LDS transpose in 4 rounds of 7 components through the wave's own
// search scratch (7 x 64 f64 = 3.5 KB; DS operations of one wave execute in order, so no block barrier): lane (c, s) sums 8 of the 64
// values of component c, three DPP steps fold the 8 sub-sums.  Then the 4 wave sums meet in a [4][28] LDS table.  ~150 wave
// instructions instead of 28 x 6 shuffle steps (~600), and no LDS beyond what the search already owns, so 4+ blocks fit a CU.
// Fixed order throughout: bitwise reproducible.
__device__ __forceinline__ double dpp_xor_add(double v, const int which) {       // which: 0 -> lane ^ 1, 1 -> lane ^ 2 (quad_perm); 2 -> row_half_mirror (pairs l, 7 - l)
  union { double d; int i[2]; } a, b; a.d = v;
  if (which == 0) { b.i[0] = __builtin_amdgcn_mov_dpp(a.i[0], 0xB1, 0xF, 0xF, true); b.i[1] = __builtin_amdgcn_mov_dpp(a.i[1], 0xB1, 0xF, 0xF, true); }
  else if (which == 1) { b.i[0] = __builtin_amdgcn_mov_dpp(a.i[0], 0x4E, 0xF, 0xF, true); b.i[1] = __builtin_amdgcn_mov_dpp(a.i[1], 0x4E, 0xF, 0xF, true); }
  else { b.i[0] = __builtin_amdgcn_mov_dpp(a.i[0], 0x141, 0xF, 0xF, true); b.i[1] = __builtin_amdgcn_mov_dpp(a.i[1], 0x141, 0xF, 0xF, true); }
  return v + b.d;
}
// fold 7 per-lane values across the wave (transpose through wbuf, 8 sub-sums per component, three DPP steps) into wsum[wid][7 G ..]
__device__ __forceinline__ void fold7(const double g[7], const int G, const bool first, double* __restrict__ wbuf, double* __restrict__ wrow) {
  const int lane = threadIdx.x & 63, c8 = lane >> 3, s8 = lane & 7;
#pragma unroll
  for (int c = 0; c < 7; c++) wbuf[c * 64 + lane] = g[c];
  wave_lds_fence();
  double v = 0;
  if (c8 < 7) {
#pragma unroll
    for (int u = 0; u < 8; u++) v += wbuf[c8 * 64 + u * 8 + s8];
  }
  v = dpp_xor_add(v, 0); v = dpp_xor_add(v, 1); v = dpp_xor_add(v, 2);   // the 8 sub-sums of a component sit in one aligned group of 8 lanes
  if (c8 < 7 && s8 == 0) wrow[7 * G + c8] = first ? v : wrow[7 * G + c8] + v;
  wave_lds_fence();
}

// G = R R^T of the pose in LDS (9 threads; the caller puts a barrier behind it): accumulate_point_n's expression, entry by entry
__device__ __forceinline__ void pose_gram(const double* __restrict__ sx0, double* __restrict__ sG, const int t) {
  if (t < 9) { const int a = t / 3, b = t - 3 * a; sG[t] = sx0[4 * a] * sx0[4 * b] + sx0[4 * a + 1] * sx0[4 * b + 1] + sx0[4 * a + 2] * sx0[4 * b + 2]; }
}
// One correspondence's contribution to the 28 sums (accumulate_point_n's arithmetic, term for term), produced SEVEN values at a time
// and folded across the wave at once: a lane never holds the 28 f64 accumulators (56 VGPRs) next to the matrices they are made of,
// which is what pushed this kernel to 240 VGPRs.  have = false: the lane contributes zeros.
__device__ __forceinline__ void emit_point(const bool have, const bool lin, const bool first, const double* __restrict__ sRx, const double* __restrict__ sTx, const double* __restrict__ sG, const float4 pa, const float4 pb,
                                           const double na[3], const double nb[3], double* __restrict__ wbuf, double* __restrict__ wrow) {
  // sRx / sTx: the 3 x 4 poses (row-major, stride 4) in LDS - Rx rotates the source normal (x0), Tx is the pose the residual is evaluated at (x0; xi in an LM trial pass);
  // sG = Rx Rx^T, formed once per block with the expression below.  Read where they are used (broadcast ds_read_b64): a lane never holds a pose in registers - the 24 to
  // 48 VGPRs that cost were what this kernel spilled at its 128-register budget.
  double m[3];
#pragma unroll
  for (int a = 0; a < 3; a++) m[a] = sRx[4 * a] * na[0] + sRx[4 * a + 1] * na[1] + sRx[4 * a + 2] * na[2];
  // R C_A R^T = R R^T - 0.999 m m^T.  R R^T is I to rounding for every pose the optimiser builds from the identity guess of
  // loop_closure.cpp:124, but NOT for a caller's f32 guess with a rotation (pcl::Registration::align(output, guess)): its rows are
  // orthonormal to 6e-8 only, the reference multiplies with the matrix as it is, and the difference is 1e-7 of the cost.
  M3 rcr;
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = a; b < 3; b++) { rcr.m[a][b] = ((a == b ? 1.0 : 0.0) - 0.999 * nb[a] * nb[b]) + (sG[3 * a + b] - 0.999 * m[a] * m[b]); rcr.m[b][a] = rcr.m[a][b]; }      // (upper triangle, mirrored: accumulate_point_n)
  M3 M = m3_inverse(rcr);
  // a lane without a correspondence contributes zeros: every one of the 28 values is linear in M, so M = 0 (and a finite residual: the neighbour record of such a lane may
  // be anything) zeroes them all - 21 selects instead of two per value
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) M.m[a][b] = have ? M.m[a][b] : 0.0;
  const double mA[3] = {(double)pa.x, (double)pa.y, (double)pa.z};
  double tA[3], e[3], Me[3];
#pragma unroll
  for (int r = 0; r < 3; r++) tA[r] = sTx[4 * r] * mA[0] + sTx[4 * r + 1] * mA[1] + sTx[4 * r + 2] * mA[2] + sTx[4 * r + 3];
#pragma unroll
  for (int r = 0; r < 3; r++) tA[r] = have ? tA[r] : 0.0;                           // (a non-finite pose: 0 x NaN must not reach the sums of a lane that has nothing to add)
  e[0] = (double)(have ? pb.x : 0.f) - tA[0]; e[1] = (double)(have ? pb.y : 0.f) - tA[1]; e[2] = (double)(have ? pb.z : 0.f) - tA[2];
#pragma unroll
  for (int r = 0; r < 3; r++) Me[r] = M.m[r][0] * e[0] + M.m[r][1] * e[1] + M.m[r][2] * e[2];
  const double x = tA[0], y = tA[1], z = tA[2];
  double g[7];
  if (lin) {
    double MS[3][3];
#pragma unroll
    for (int r = 0; r < 3; r++) { MS[r][0] = M.m[r][1] * z + M.m[r][2] * (-y); MS[r][1] = M.m[r][0] * (-z) + M.m[r][2] * x; MS[r][2] = M.m[r][0] * y + M.m[r][1] * (-x); }
    g[0] = z * MS[1][0] + (-y) * MS[2][0]; g[1] = z * MS[1][1] + (-y) * MS[2][1]; g[2] = z * MS[1][2] + (-y) * MS[2][2];
    g[3] = -MS[0][0]; g[4] = -MS[1][0]; g[5] = -MS[2][0];                     // (-skew(tA)^T M = -MS^T, bit for bit: M is symmetric - accumulate_point_n)
    g[6] = (-z) * MS[0][1] + x * MS[2][1];
    fold7(g, 0, first, wbuf, wrow);
    g[0] = (-z) * MS[0][2] + x * MS[2][2];
    g[1] = -MS[0][1]; g[2] = -MS[1][1]; g[3] = -MS[2][1];
    g[4] = y * MS[0][2] + (-x) * MS[1][2];
    g[5] = -MS[0][2]; g[6] = -MS[1][2];
    fold7(g, 1, first, wbuf, wrow);
    g[0] = -MS[2][2];
    g[1] = M.m[0][0]; g[2] = M.m[0][1]; g[3] = M.m[0][2]; g[4] = M.m[1][1]; g[5] = M.m[1][2]; g[6] = M.m[2][2];
    fold7(g, 2, first, wbuf, wrow);
    g[0] = z * Me[1] + (-y) * Me[2]; g[1] = (-z) * Me[0] + x * Me[2]; g[2] = y * Me[0] + (-x) * Me[1];
    g[3] = -Me[0]; g[4] = -Me[1]; g[5] = -Me[2];
  } else {
    // an LM trial pass: the cost alone.  Component 6 of the last group through the same slots and the same order of additions as fold7 (the cost of a linearisation pass and
    // of a trial pass are compared by the controller), the other 27 sums of the row are zero - no seven-wide fold of six zeros (a zero f64 pair kept alive across the whole
    // kernel for it was this kernel's last spill)
    const int lane = threadIdx.x & 63, c8 = lane >> 3, s8 = lane & 7;
    double zero = 0.0; asm volatile("" : "+v"(zero));                   // (materialised HERE: hoisted out of the per-point loop the constant was kept alive - in scratch - across the search)
    if (first && lane < 27) wrow[lane] = zero;
    wbuf[6 * 64 + lane] = e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2];
    wave_lds_fence();
    double v = zero;
    if (c8 == 6) {
#pragma unroll
      for (int u = 0; u < 8; u++) v += wbuf[6 * 64 + u * 8 + s8];
    }
    v = dpp_xor_add(v, 0); v = dpp_xor_add(v, 1); v = dpp_xor_add(v, 2);
    if (c8 == 6 && s8 == 0) wrow[27] = first ? v : wrow[27] + v;
    wave_lds_fence();
    return;
  }
  g[6] = e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2];
  fold7(g, 3, first, wbuf, wrow);
}

// All target points within R of q (one query per wave, ONE pass over the cap box of radius R - which covers ball(q, R) completely, so
// every point that is NOT collected is farther than R): indices into list[], the smallest key and the runner-up's d2.  Returns the
// number of points found (> cap: the list overflowed, the caller shrinks R).
__device__ __forceinline__ uint32_t wave_ball_collect(const GridView& g, float qx, float qy, float qz, float R, WaveLds* ws, unsigned long long* __restrict__ wlist /* [QN_HCAP1], LDS */,
                                                      uint32_t* __restrict__ wcnt /* LDS */, unsigned long long& best, float& second) {
  const CapBox cb = cap_of(g, qx, qy, qz);
  const float rx = cap_extent(g, cb, R, 0), ry = cap_extent(g, cb, R, 1), rz = cap_extent(g, cb, R, 2);
  int x0 = rfl(cell_coord(qx - rx, g.ox, g.inv_cell, g.nx)), x1 = rfl(cell_coord(qx + rx, g.ox, g.inv_cell, g.nx));
  int y0 = rfl(cell_coord(qy - ry, g.oy, g.inv_cell, g.ny)), y1 = rfl(cell_coord(qy + ry, g.oy, g.inv_cell, g.ny));
  int z0 = rfl(cell_coord(qz - rz, g.oz, g.inv_cell, g.nz)), z1 = rfl(cell_coord(qz + rz, g.oz, g.inv_cell, g.nz));
  const bool tile_mode = ((x1 >> 3) - (x0 >> 3) + 1) * (y1 - y0 + 1) * (z1 - z0 + 1) > 128;
  if (tile_mode) {
    x0 = (x0 >> 3) << 3; x1 = min(((x1 >> 3) << 3) + 7, g.nx - 1);
    y0 = (y0 >> 2) << 2; y1 = min(((y1 >> 2) << 2) + 3, g.ny - 1);
    z0 = (z0 >> 2) << 2; z1 = min(((z1 >> 2) << 2) + 3, g.nz - 1);
  }
  const int lane = threadIdx.x & 63;
  const float R2 = R * R;
  unsigned long long b = QN_INF_KEY; float s2 = __int_as_float(0x7f800000);
  uint32_t found = 0;                                                  // (wave-uniform: list positions from a ballot's prefix count - up to 64 returning atomics on ONE LDS word per chunk before)
  wave_lds_fence();
  const uint32_t streamed = stream_box(g, x0, x1, y0, y1, z0, z1, tile_mode, ws, [&](const float4& p, bool valid, uint32_t) __attribute__((always_inline)) {
    const float d2 = sqdist(qx, qy, qz, p.x, p.y, p.z);
    const bool in = valid && d2 <= R2;
    const unsigned long long mk = __ballot(in);
    if (in) {
      const unsigned long long k = pack_key(d2, __float_as_uint(p.w));
      const uint32_t pos = found + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
      if (pos < QN_HCAP1) wlist[pos] = k;
      if (k < b) { if (b != QN_INF_KEY) s2 = key_d2(b); b = k; } else if (d2 < s2) s2 = d2;
    }
    found += (uint32_t)__popcll(mk);
  }, qx, qy, qz, tile_mode ? R2 * 1.000002f : -1.f);                    // (tiles outside the ball are not streamed: only points with d2 <= R2 are used)
  if (lane == 0) *wcnt = found;
  wave_lds_fence();
  if (g.dbg && lane == 0) { atomicAdd(&g.dbg[14], 1u); atomicAdd(&g.dbg[15], streamed); atomicAdd(&g.dbg[9], (uint32_t)(tile_mode ? ((x1 >> 3) - (x0 >> 3) + 1) * ((y1 >> 2) - (y0 >> 2) + 1) * ((z1 >> 2) - (z0 >> 2) + 1) : ((x1 >> 3) - (x0 >> 3) + 1) * (y1 - y0 + 1) * (z1 - z0 + 1))); }      // developer counters: calls, candidates, segments
  const unsigned long long wb = wave_min_u64(b);
  float c = (b == wb) ? s2 : key_d2(b);
  if (b == QN_INF_KEY) c = __int_as_float(0x7f800000);
  c = wave_min_f(c);
  best = wb; second = c;
  return found;
}

// Tracking state of one query beyond (j0, q_ref, bound): the RUNNER-UP.  Used by the persistent kernel only (TOP2), where it lives in registers.
// A query whose runner-up is as close as its neighbour - two target points equidistant to 1e-5 m: a source point metres off the target's surface sees many -
// fails the pruning inequality  d(q, p_j0) + delta < d_other  at EVERY tick however little it moves, and its block then spends 2-6 us per tick in a rescan or
// a wave-level search while the other 195 blocks finish in 3.6: the tick ends with its slowest block (device-clock timeline, profiles/r3_*).  With the
// runner-up j1 kept beside j0, d_other bounds everything ELSE (the third-nearest point, the unscanned region): the nearer of the two is exact whenever
// min(d0, d1) + delta < d_other, ties by index as everywhere.  The results are the same neighbours (exact both ways), so the sums and the pose are the
// same bits as the k_tick chain's.
struct Top2 { int32_t j1; float4 p1; };                                // j1 = -1: none

// One source point of a tick (the body of k_tick's loop; also the body of the persistent align kernel, qn_persist.cuh).
//   lin = false   LM trial error: the cached correspondence rec0 at the trial pose xi, gate as at the linearisation;
//   lin = true    tracked exact 1-NN at the pose x0 (k_nn_track's logic: bound pruning, in-lane rescan of <= QN_TRACK_SEG segments, far-query cache,
//                 cooperative big-ball search 16 queries at a time) and the correspondence's contribution to the 28 sums, or (MODE 1) its squared distance
//                 and the transformed point of the output cloud.
// sx0 / sxi: the f64 poses in LDS (fetched only where the sums are formed: the search and the accumulation are the two register-hungry parts, their live
// ranges are kept apart).  wl / red: the wave's search scratch and the transpose buffer that aliases it; wrow: the wave's row of the block's 28 sums.
// Must be called convergently by the whole wave.  PROBE: developer clock stamps (compiled out of the production kernels).
// NA_LATE: the source normal is fetched here, right before the sums are formed (k_tick: six registers fewer across the search); the persistent kernel hands it over preloaded
template <int MODE, bool PROBE, bool TOP2, bool NA_LATE = false>
__device__ __forceinline__ void tick_point(const TickArgs& a, const float (&Tf)[12], const double* __restrict__ sx0, const double* __restrict__ sxi, const double* __restrict__ sG, const bool lin, const bool first,
                                           const uint32_t t, const bool valid, const float4 p, int32_t& j0s, float4& ref, const double (&na_in)[3], TargetRec& rec0, Top2& t2,
                                           WaveLds* __restrict__ wl, double* __restrict__ red, double* __restrict__ wrow, unsigned long long* __restrict__ wlist, uint32_t* __restrict__ wcnt, const bool probe) {
  const int tid = threadIdx.x, lane = tid & 63;
  const float INF = __int_as_float(0x7f800000);
  float qx, qy, qz; xform_query<MODE>(Tf, p.x, p.y, p.z, qx, qy, qz);
  const bool finite_q = (qx - qx == 0.f) && (qy - qy == 0.f) && (qz - qz == 0.f);
  const uint32_t j0 = (uint32_t)j0s;
  unsigned long long best = QN_INF_KEY;
  if (!lin) {                                                    // LM trial error: cached correspondence, gate as at the linearisation
    bool have = false;
    if (valid && finite_q && j0 < a.tgt.n) have = (double)sqdist(qx, qy, qz, rec0.p.x, rec0.p.y, rec0.p.z) < a.thr2;
    double na[3] = {NA_LATE ? 0.0 : na_in[0], NA_LATE ? 0.0 : na_in[1], NA_LATE ? 0.0 : na_in[2]};      // (NA_LATE: the caller's placeholder is never read - a literal here, not a value kept alive across the search)
    if (NA_LATE && valid) { na[0] = a.nrm_s[(size_t)t * 3]; na[1] = a.nrm_s[(size_t)t * 3 + 1]; na[2] = a.nrm_s[(size_t)t * 3 + 2]; }
    wave_lds_fence();
    emit_point(have, false, first, sx0, sxi, sG, make_float4(p.x, p.y, p.z, 1.f), rec0.p, na, rec0.n, red, wrow);
    return;
  }
  // ---- phase 0: tracked exact 1-NN (k_nn_track's logic: bound pruning, small in-lane rescan, cooperative big-ball search)
  float second = INF, d_unseen = INF, r = 0.f, delta = 0.f;
  bool rescanned = false, big = false;
  unsigned long long key2 = QN_INF_KEY; float third = INF;          // TOP2: the runner-up's key and the squared distance of the third-nearest, as far as this tick's scan knows them
  const GridView& tg = a.tgt;
  if (valid && finite_q && j0 >= tg.n) { big = true; r = tg.cell; }         // no usable seed: unseeded search
  if (valid && finite_q && j0 < tg.n) {
    float d0 = sqdist(qx, qy, qz, rec0.p.x, rec0.p.y, rec0.p.z);
    best = pack_key(d0, j0);
    if (TOP2 && (uint32_t)t2.j1 < tg.n) {                            // the nearer of the neighbour and its runner-up (ref.w bounds everything else)
      const float d1 = sqdist(qx, qy, qz, t2.p1.x, t2.p1.y, t2.p1.z);
      const unsigned long long k1 = pack_key(d1, (uint32_t)t2.j1);
      if (k1 < best) { best = k1; d0 = d1; }
    }
    delta = sqrtf(sqdist(qx, qy, qz, ref.x, ref.y, ref.z));
    if (!track_bound_holds(d0, delta, ref.w)) {
      r = sqrtf(d0) * 1.000001f + tg.eps;
      const int bx0 = cell_coord(qx - r, tg.ox, tg.inv_cell, tg.nx), bx1 = cell_coord(qx + r, tg.ox, tg.inv_cell, tg.nx);
      const int by0 = cell_coord(qy - r, tg.oy, tg.inv_cell, tg.ny), by1 = cell_coord(qy + r, tg.oy, tg.inv_cell, tg.ny);
      const int bz0 = cell_coord(qz - r, tg.oz, tg.inv_cell, tg.nz), bz1 = cell_coord(qz + r, tg.oz, tg.inv_cell, tg.nz);
      const int tx0 = bx0 >> 3, ntr = (bx1 >> 3) - tx0 + 1, nyr = by1 - by0 + 1;
      const int nseg = ntr * nyr * (bz1 - bz0 + 1);
      if (!(d0 == d0) || nseg > QN_TRACK_SEG) big = true;
      else {
        uint32_t s[QN_TRACK_SEG], e[QN_TRACK_SEG];
#pragma unroll
        for (int sg = 0; sg < QN_TRACK_SEG; sg++) {
          s[sg] = 0; e[sg] = 0;
          if (sg < nseg) {
            int tt, rr; divmod_small(sg, ntr, rr, tt);
            int qz_, ry_; divmod_small(rr, nyr, qz_, ry_);
            const int ry = by0 + ry_, rz = bz0 + qz_, tx = tx0 + tt;
            const int xa = max(bx0, tx << 3), xb = min(bx1, (tx << 3) + 7);
            const uint32_t k0 = cell_key(tg, xa, ry, rz);
            s[sg] = tg.cell_start[k0]; e[sg] = tg.cell_start[k0 + (xb - xa) + 1];
          }
        }
#pragma unroll
        for (int sg = 0; sg < QN_TRACK_SEG; sg++) {
          for (uint32_t u = s[sg]; u < e[sg]; u++) {
            const float4 c = tg.pts[u];
            const float da = sqdist(qx, qy, qz, c.x, c.y, c.z);
            const unsigned long long ka = pack_key(da, __float_as_uint(c.w));
            if (TOP2) {                                            // best < key2 (keys), third = squared distance of the nearest point that is neither
              if (ka < best) { third = key2 != QN_INF_KEY ? key_d2(key2) : third; key2 = best; best = ka; }
              else if (ka != best) { if (ka < key2) { third = key2 != QN_INF_KEY ? key_d2(key2) : third; key2 = ka; } else if (ka != key2 && da < third) third = da; }
            } else {
              if (ka < best) { second = key_d2(best); best = ka; }
              else if (ka != best && da < second) second = da;
            }
          }
        }
        if (TOP2) second = key2 != QN_INF_KEY ? key_d2(key2) : INF;
        float d = INF;
        if (bx0 > 0) d = fminf(d, qx - (tg.ox + bx0 * tg.cell));
        if (bx1 < tg.nx - 1) d = fminf(d, (tg.ox + (bx1 + 1) * tg.cell) - qx);
        if (by0 > 0) d = fminf(d, qy - (tg.oy + by0 * tg.cell));
        if (by1 < tg.ny - 1) d = fminf(d, (tg.oy + (by1 + 1) * tg.cell) - qy);
        if (bz0 > 0) d = fminf(d, qz - (tg.oz + bz0 * tg.cell));
        if (bz1 < tg.nz - 1) d = fminf(d, (tg.oz + (bz1 + 1) * tg.cell) - qz);
        d_unseen = d - tg.eps; rescanned = true;
      }
    }
  }
  bool requested = false;
  const int far_mode = MODE == 0 ? a.far_mode : (a.far_mode != 0 ? 2 : 0);      // the closing pass has no refresh kernel behind it
  const bool far_lane = big && r > QN_FAR_RMIN_CELLS * tg.cell;       // only truly far neighbours: a ball of a few cells is cheaper to search with the wave (shared stream)
  if (far_mode != 0) {
    if (far_lane) {
      const float4 cr = a.cand_ref[t];
      if (cr.w > 0.f) {                                            // candidate list made at cr.xyz, everything else is >= cr.w away from there
        const float dc = sqrtf(sqdist(qx, qy, qz, cr.x, cr.y, cr.z));
        const float2 pb = a.cand_b[t];                               // ... and everything but the first 4 / 16 candidates is >= pb.x / pb.y away
        unsigned long long b1 = QN_INF_KEY; float s2 = INF, used = 0.f;
        const int4* cl = (const int4*)(a.cand + (size_t)t * QN_FAR_M);
        bool proven = false;
#pragma unroll 1
        for (int stage = 0; stage < 3 && !proven; stage++) {           // 4, then 16, then all 64 candidates
          const int u0 = stage == 0 ? 0 : (stage == 1 ? 1 : 4), u1 = stage == 0 ? 1 : (stage == 1 ? 4 : QN_FAR_M / 4);
          for (int u = u0; u < u1; u++) {
            const int4 c4 = cl[u];
            const int cj[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
            for (int v = 0; v < 4; v++) {
              if ((uint32_t)cj[v] < tg.n) {
                const float4 cp = a.tgt_raw[cj[v]];
                const float dd = sqdist(qx, qy, qz, cp.x, cp.y, cp.z);
                const unsigned long long kk = pack_key(dd, (uint32_t)cj[v]);
                if (kk < b1) { s2 = b1 != QN_INF_KEY ? key_d2(b1) : s2; b1 = kk; } else if (dd < s2) s2 = dd;
              }
            }
          }
          used = stage == 0 ? pb.x : (stage == 1 ? pb.y : cr.w);
          proven = b1 != QN_INF_KEY && track_bound_holds(key_d2(b1), dc, used);
        }
        if (proven) {                                                // the nearest neighbour is one of the candidates looked at; every other point is >= used - dc away
          best = b1; second = s2; d_unseen = used - dc - tg.eps; rescanned = true; big = false;
        }
      }
    }
    if (far_mode == 1) {
      requested = far_lane && big; big = big && !requested;        // misses: k_far searches them (and adds their share of the sums)
      const unsigned long long word = __ballot(requested);
      if (lane == 0) {
        const uint32_t chunk = t >> 6;
        if (chunk * 64u < a.src.n) a.far_req[chunk] = word;
        if (word) atomicAdd(&a.far_stats[0], (uint32_t)__popcll(word));
      }
    } else if (MODE == 0) {                                        // mode 2: misses are searched right here; counted so that the host can bring k_far back
      const unsigned long long word = __ballot(big && far_lane);
      if (lane == 0 && word) atomicAdd(&a.far_stats[0], (uint32_t)__popcll(word));
    }
  }
  if (PROBE) {
    if (probe) a.clk[3] = wall_clock64();
    if (MODE == 0) { const unsigned long long wb = __ballot(big), wr = __ballot(rescanned), wf = __ballot(big && far_lane);
      if (big || rescanned) { a.clk_blk[8 * blockIdx.x + 7] = ((unsigned long long)t << 32) | (unsigned long long)(uint32_t)j0s; a.clk_blk[8 * 1024 + 4 * blockIdx.x] = __float_as_uint(key_d2(best)); a.clk_blk[8 * 1024 + 4 * blockIdx.x + 1] = __float_as_uint(ref.w); a.clk_blk[8 * 1024 + 4 * blockIdx.x + 2] = __float_as_uint(delta); a.clk_blk[8 * 1024 + 4 * blockIdx.x + 3] = __float_as_uint(r); }
      if (lane == 0) { atomicAdd(&a.clk_blk[8 * blockIdx.x + 4], (unsigned long long)__popcll(wb)); atomicAdd(&a.clk_blk[8 * blockIdx.x + 5], (unsigned long long)__popcll(wr)); atomicAdd(&a.clk_blk[8 * blockIdx.x + 6], (unsigned long long)__popcll(wf)); } }
  }
  // the wave's big-ball queries, 16 at a time, cooperatively (neighbouring queries' balls overlap: one shared candidate stream)
  const bool was_big = big;
  for (unsigned long long pend = __ballot(big); pend != 0;) {
    unsigned long long grp = 0, tmp = pend; int srcl = -1;
    for (int sl = 0; sl < 16 && tmp != 0; sl++) { const int L = __ffsll((long long)tmp) - 1; if (sl == (lane & 15)) srcl = L; grp |= 1ull << L; tmp &= tmp - 1; }
    pend &= ~grp;
    const bool act = srcl >= 0;
    const int sl_ = act ? srcl : 0;
    const float ax = __shfl(qx, sl_), ay = __shfl(qy, sl_), az = __shfl(qz, sl_), ar = __shfl(r, sl_), ad = __shfl(delta, sl_);
    float rs = (ad < 0.25f * ar) ? ar * 1.1f + 0.5f * tg.cell : ar;           // tight seed: scan a little wider (bound pruning next time)
    Best1 sink; sink.init();
    float du = INF;
    wave_search<4>(tg, ax, ay, az, act, rs, INF, 64, sink, wl, du);
    const int slot = __popcll(grp & ((1ull << lane) - 1ull));
    const unsigned long long rk = __shfl(sink.key, slot); const float rsec = __shfl(sink.second, slot), rdu = __shfl(du, slot);
    if ((grp >> lane) & 1ull) { best = rk; second = rsec; d_unseen = rdu; rescanned = true; key2 = QN_INF_KEY; third = INF; }
  }
  // TOP2: a query that needed the wave although it barely moved, and whose fresh scan left no room either (a tie to 2e-4 cells), gets its runner-up from ONE
  // pass over the ball of radius d_nn + m with all 64 lanes on that query: the three smallest keys inside it -> (j0, j1, bound on everything else).
  if (TOP2 && MODE == 0) {
    bool want = false;
    if (valid && was_big && rescanned && best != QN_INF_KEY && !requested)
      want = delta < 0.02f * tg.cell && fminf(sqrtf(second), d_unseen) - sqrtf(key_d2(best)) < fmaxf(1.5f * delta, 2e-4f * tg.cell);
    for (unsigned long long pend = __ballot(want); pend != 0; pend &= pend - 1) {
      const int L = __ffsll((long long)pend) - 1;
      const float ax = __shfl(qx, L), ay = __shfl(qy, L), az = __shfl(qz, L), amoved = __shfl(delta, L);
      const float dnn = sqrtf(key_d2(__shfl(best, L)));
      float m = fminf(fmaxf(8.f * amoved, 0.05f * tg.cell), 0.5f * tg.cell);
      unsigned long long lb = QN_INF_KEY; float lsec = INF;
      for (int attempt = 0; attempt < 4; attempt++, m *= 0.25f) {
        const float R = dnn * 1.000002f + tg.eps + m;
        const uint32_t cnt = wave_ball_collect(tg, ax, ay, az, R, wl, wlist, wcnt, lb, lsec);
        if (cnt <= (uint32_t)QN_HCAP1 && cnt > 0) {                  // every point of the ball is in the list: the three smallest keys
          const unsigned long long o0 = (uint32_t)lane < cnt ? wlist[lane] : QN_INF_KEY, o1 = (uint32_t)lane + 64u < cnt ? wlist[lane + 64] : QN_INF_KEY;
          const unsigned long long k1 = wave_min_u64(o0 < o1 ? o0 : o1);
          const unsigned long long e0 = o0 == k1 ? QN_INF_KEY : o0, e1 = o1 == k1 ? QN_INF_KEY : o1;
          const unsigned long long k2 = wave_min_u64(e0 < e1 ? e0 : e1);
          const unsigned long long f0 = e0 == k2 ? QN_INF_KEY : e0, f1 = e1 == k2 ? QN_INF_KEY : e1;
          const unsigned long long k3 = wave_min_u64(f0 < f1 ? f0 : f1);
          if (lane == L) {
            best = k1; key2 = k2; second = k2 != QN_INF_KEY ? key_d2(k2) : INF;
            third = k3 != QN_INF_KEY ? key_d2(k3) : INF;
            d_unseen = R * 0.9999995f;                               // nothing outside the ball is closer than its radius
          }
          break;
        }
      }
    }
  }
  if (MODE == 1) {                                               // fitness + output cloud (original point order)
    double d2 = 0.0; uint32_t cnt = 0;
    if (valid) {
      a.aligned[__float_as_uint(p.w)] = make_float4(qx, qy, qz, 1.0f);
      if (best != QN_INF_KEY) { d2 = (double)key_d2(best); cnt = 1; }      // max_range = DBL_MAX: every source point with a neighbour counts
    }
    d2 = wave_sum_f64_dpp(d2);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0) { if (first) { wrow[0] = d2; wrow[1] = (double)cnt; } else { wrow[0] += d2; wrow[1] += (double)cnt; } }
    return;
  }
  bool have = false;
  if (valid && !requested) {
    const int32_t jn = best != QN_INF_KEY ? (int32_t)key_idx(best) : -1;
    if (TOP2) {
      if (rescanned) {                                             // a fresh scan: the runner-up it found (or none), its point fetched once
        t2.j1 = key2 != QN_INF_KEY ? (int32_t)key_idx(key2) : -1;
        if (t2.j1 >= 0) t2.p1 = a.tgt_raw[t2.j1];
      } else if (jn != j0s) {                                      // proven, but the runner-up has become the nearer one: the two swap roles, ref still bounds everything else
        t2.j1 = j0s; t2.p1 = rec0.p;
      }
    }
    if (jn != j0s) a.nn_idx[t] = jn;                               // (only a CHANGED neighbour goes to memory: a converging align re-proves nearly every neighbour, and 100k unconditional stores per tick were a third of the kernel's write traffic)
    j0s = jn;                                                      // (the tracking record also stays in the caller's registers: the persistent kernel never re-reads it)
    if (rescanned) { ref = make_float4(qx, qy, qz, fminf(sqrtf(TOP2 && key2 != QN_INF_KEY ? third : second), d_unseen)); a.nn_ref[t] = ref; }
    if (best != QN_INF_KEY) {
      const uint32_t j = key_idx(best);                              // rec0 always is the record of j0s (gated out or not)
      if (j != j0) { const TargetRec* rr = a.tgt_rec + j; rec0.p = rr->p; rec0.n[0] = rr->n[0]; rec0.n[1] = rr->n[1]; rec0.n[2] = rr->n[2]; }
      have = (double)key_d2(best) < a.thr2;
    }
  }
  if (PROBE) {
    if (probe) a.clk[4] = wall_clock64();
    if (MODE == 0 && threadIdx.x == 0) a.clk_blk[8 * blockIdx.x + 2] = wall_clock64();
  }
  double na[3] = {NA_LATE ? 0.0 : na_in[0], NA_LATE ? 0.0 : na_in[1], NA_LATE ? 0.0 : na_in[2]};
  if (NA_LATE && valid) { na[0] = a.nrm_s[(size_t)t * 3]; na[1] = a.nrm_s[(size_t)t * 3 + 1]; na[2] = a.nrm_s[(size_t)t * 3 + 2]; }
  wave_lds_fence();                                              // the wave's search scratch becomes its transpose buffer
  emit_point(have, true, first, sx0, sx0, sG, make_float4(p.x, p.y, p.z, 1.f), rec0.p, na, rec0.n, red, wrow);
}

// MODE 0: an optimiser tick.  MODE 1: the closing pass of align() in ONE kernel - the last controller step in the prologue, then (only
// when the state machine is done) pcl::Registration::getFitnessScore's nearest-neighbour sweep at the FINAL pose (f32 transform in PCL's
// SSE order, tracked from the last tick's neighbours; per-block f64 sums of the f32 squared distances) and the output cloud
// pcl::transformPointCloud(*input_, output, final_transformation_) (loop_closure.cpp:124, 127).  It replaces six launches
// (controller, tracked NN, its list pass, two fitness reductions, the transform); k_finalize_fit folds the block sums and fills the result.
// PROBE = true: the developer variant with device-clock stamps (knob clk_probe); the production instantiations carry none of it.
union WaveScratch { WaveLds w; double red[7 * 64]; };
static_assert(sizeof(WaveLds) >= 7 * 64 * sizeof(double), "the transpose buffer aliases the wave's search scratch");
template <int TB_, int OCC_, int MODE, bool PROBE>
struct TickK {
  static constexpr int TB = TB_, OCC = OCC_;
  using Args = TickArgs;
  static __device__ __forceinline__ void run(const TickArgs& a_in, const uint32_t bx, const uint32_t nbx_) {
    TickArgs a = a_in;

  __shared__ WaveScratch sc[TB / 64];
  __shared__ double wsum[2][TB / 64][QN_NPART];                      // (two buffers: the row a block has just finished is summed while its waves start the next one)
  __shared__ TailLds tl;                                            // the state this launch runs under; in the last block: the controller's workspace and the next state
  __shared__ double sG[9];                                          // R R^T of the pose (emit_point)
  const int tid = threadIdx.x;
  const uint32_t nblk = nbx_;
  a.src = grid_resolve(a.src); a.tgt = grid_resolve(a.tgt);
  const bool probe = PROBE && a.clk != nullptr && bx == 0 && threadIdx.x == 0;
  if (PROBE) {
    if (probe) a.clk[0] = wall_clock64();
    if (threadIdx.x == 0) atomicMax(&a.clk[7], ~wall_clock64());       // earliest block start (as a max of the complement)
    if (MODE == 0 && threadIdx.x == 0) { a.clk_blk[8 * bx] = wall_clock64(); a.clk_blk[8 * bx + 4] = 0; a.clk_blk[8 * bx + 5] = 0; a.clk_blk[8 * bx + 6] = 0; }
  }
  const uint32_t lblk = xcd_block(bx, nblk);               // XCD x works on one contiguous eighth of the cell-sorted source
  // ---- pose-independent loads of this thread's first point, in flight during the prologue
  const uint32_t row0 = lblk * a.rpb;
  uint32_t t = (row0 * a.ppt) * TB + tid;
  bool valid = t < a.src.n;
  float4 p = valid ? a.src.pts[t] : make_float4(0, 0, 0, 0);
  int32_t j0s = valid ? a.nn_idx[t] : -1;
  float4 ref = valid ? a.nn_ref[t] : make_float4(0, 0, 0, 0);
  const double na[3] = {0, 0, 0};                                   // (fetched inside tick_point: NA_LATE)
  TargetRec rec0; rec0.p = make_float4(0, 0, 0, 0); rec0.n[0] = rec0.n[1] = rec0.n[2] = 0;
  if (valid && (uint32_t)j0s < a.tgt.n) { const TargetRec* r = a.tgt_rec + j0s; rec0.p = r->p; rec0.n[0] = r->n[0]; rec0.n[1] = r->n[1]; rec0.n[2] = r->n[2]; }
  Top2 no_t2; no_t2.j1 = -1; no_t2.p1 = make_float4(0, 0, 0, 0);     // (the chain keeps no runner-up: its tracking record is re-read from memory every tick)

  // ---- the state this launch runs under: written by the controller tail of the previous launch (or by k_init_state / k_solve).  ONE 300-byte read, requested back
  // to back with the point loads above - no row reduction, no controller in front of the body any more (controller_tail).
  for (int i = tid; i < (int)(sizeof(GicpState) / 8); i += TB) ((unsigned long long*)&tl.sh)[i] = ((const unsigned long long*)a.tail.st_in)[i];
  __syncthreads();
  if (PROBE) {
    if (probe) { a.clk[1] = wall_clock64(); a.clk[2] = a.clk[1]; }
    if (MODE == 0 && threadIdx.x == 0) a.clk_blk[8 * bx + 1] = wall_clock64();
  }
  const int phase = tl.sh.phase;
  if (MODE == 0 ? phase == 2 : phase != 2) { if (MODE == 0 && a.tail.enabled) state_pass_through<TB>(a.tail, bx); return; }      // ticks past convergence / a closing pass before it: nothing to do
  const bool lin = MODE == 1 || phase == 0;
  float Tf[12];
#pragma unroll
  for (int j = 0; j < 12; j++) Tf[j] = (float)tl.sh.x0[j];
  pose_gram(tl.sh.x0, sG, tid);
  __syncthreads();
  for (uint32_t rr = 0; rr < a.rpb; rr++) {
    const uint32_t row = row0 + rr;
    if (row >= a.rows) break;
    double (*ws)[QN_NPART] = wsum[rr & 1u];
    for (uint32_t it = 0; it < a.ppt; it++) {
      if (rr > 0 || it > 0) {                                        // (more than one point per thread: batch members, clouds beyond 131072 points)
        t = (row * a.ppt + it) * TB + tid; valid = t < a.src.n;
        p = valid ? a.src.pts[t] : make_float4(0, 0, 0, 0);
        j0s = valid ? a.nn_idx[t] : -1;
        ref = valid ? a.nn_ref[t] : make_float4(0, 0, 0, 0);
        if (valid && (uint32_t)j0s < a.tgt.n) { const TargetRec* r = a.tgt_rec + j0s; rec0.p = r->p; rec0.n[0] = r->n[0]; rec0.n[1] = r->n[1]; rec0.n[2] = r->n[2]; }
      }
      tick_point<MODE, PROBE, false, true>(a, Tf, tl.sh.x0, tl.sh.xi, sG, lin, it == 0, t, valid, p, j0s, ref, na, rec0, no_t2, &sc[tid >> 6].w, sc[tid >> 6].red, ws[tid >> 6], nullptr, nullptr, probe);
    }
    __syncthreads();
    if (MODE == 1) {
      if (tid == 0) { double sv = 0, cv = 0; for (int w = 0; w < TB / 64; w++) { sv += ws[w][0]; cv += ws[w][1]; } a.fit_psum[row] = sv; a.fit_pcnt[row] = (uint32_t)cv; }
    } else if (tid < QN_NPART) { double v = 0;
#pragma unroll
      for (int w = 0; w < TB / 64; w++) v += ws[w][tid];
      row_store(&a.part_out[(size_t)row * QN_NPART + late((uint32_t)tid)], v, a.tail.enabled != 0); }      // (late: the store address is formed here, not carried - spilled - across the body)
  }
  if (MODE == 1) return;
  if (PROBE) {
    if (probe) a.clk[5] = wall_clock64();
    if (threadIdx.x == 0) atomicMax(&a.clk[6], wall_clock64());        // latest block end
    if (MODE == 0 && threadIdx.x == 0) a.clk_blk[8 * bx + 3] = wall_clock64();
  }
  // ---- the controller step on this launch's rows, by the last block to get here (far_mode 1: k_far adds a row first - k_far_reduce runs the step)
  if (a.tail.enabled) controller_tail<TB>(a.tail, a.part_out, nblk, &tl);
  }
};
// K4b / K5 for the UNSEEDED ticks and the debug linearisation: the correspondences corr[] of the search that just ran (API order) -> one partial row per
// block, and the controller step in the last block (controller_tail).  phase 0: linearize - M = (C_B + R C_A R^T)^-1, e = mu_B - T mu_A, H += J^T M J, b += J^T M e,
// cost += e^T M e; phase 1: compute_error at the trial transform xi with the cached correspondences.  The sums are formed by emit_point - seven values at a time, folded
// across the wave at once - as in the tracked ticks: a thread never holds 28 f64 accumulators (this kernel ran at 150 VGPRs = three waves per SIMD with them, and its
// final 28 wave-wide DPP sums were 500 of its 800 instructions per wave).  Fixed grid, fixed striding, fixed fold order => bitwise reproducible rows.
struct AccumulateK {
  static constexpr int TB = QN_BLOCK, OCC = 4;
  struct Args { const float4* src_raw; uint32_t ns; const double* nrm_s; const TargetRec* tgt_rec; const int32_t* corr; const GicpState* st; double* partials; int cond; TailArgs tail; };
  static __device__ __forceinline__ void run(const Args& a, const uint32_t bx, const uint32_t nbx) {
    __shared__ double red[QN_BLOCK / 64][7 * 64];
    __shared__ double wsum[QN_BLOCK / 64][QN_NPART];
    __shared__ TailLds tl;
    __shared__ double sG[9];
    const int tid = threadIdx.x;
    for (int i = tid; i < (int)(sizeof(GicpState) / 8); i += QN_BLOCK) ((unsigned long long*)&tl.sh)[i] = ((const unsigned long long*)a.st)[i];
    __syncthreads();
    const int phase = tl.sh.phase;
    const bool skip = phase == 2 || (a.cond && !(tl.sh.reserved & a.cond));      // done, or a conditional launch behind look_decide whose flag is not set
    if (skip) { if (a.tail.enabled) state_pass_through<QN_BLOCK>(a.tail, bx); return; }
    pose_gram(tl.sh.x0, sG, tid);
    __syncthreads();
    const bool lin = phase == 0;
    const double* sTx = lin ? tl.sh.x0 : tl.sh.xi;
    const uint32_t ns = a.ns;
    bool first = true;
    for (uint32_t base = bx * QN_BLOCK; base < ns; base += nbx * QN_BLOCK, first = false) {      // (wave-uniform trip count: emit_point folds across the wave)
      const uint32_t i = base + (uint32_t)tid;
      const int j = i < ns ? a.corr[i] : -1;
      const bool have = j >= 0;
      float4 pa = make_float4(0.f, 0.f, 0.f, 1.f), pb = make_float4(0.f, 0.f, 0.f, 0.f);
      double na[3] = {0.0, 0.0, 0.0}, nb[3] = {0.0, 0.0, 0.0};
      if (have) {
        const TargetRec* rec = a.tgt_rec + j;
        const float4 ps = a.src_raw[i];
        pa = make_float4(ps.x, ps.y, ps.z, 1.f); pb = rec->p;
        na[0] = a.nrm_s[(size_t)i * 3]; na[1] = a.nrm_s[(size_t)i * 3 + 1]; na[2] = a.nrm_s[(size_t)i * 3 + 2];
        nb[0] = rec->n[0]; nb[1] = rec->n[1]; nb[2] = rec->n[2];
      }
      wave_lds_fence();
      emit_point(have, lin, first, tl.sh.x0, sTx, sG, pa, pb, na, nb, red[tid >> 6], wsum[tid >> 6]);
    }
    __syncthreads();
    if (tid < QN_NPART) { double v = 0;
#pragma unroll
      for (int w = 0; w < QN_BLOCK / 64; w++) v += wsum[w][tid];
      row_store(&a.partials[(size_t)bx * QN_NPART + tid], v, a.tail.enabled != 0); }
    if (a.tail.enabled) controller_tail<QN_BLOCK>(a.tail, a.partials, nbx, &tl);
  }
};
static __global__ void __launch_bounds__(QN_BLOCK, 4) k_accumulate(AccumulateK::Args a) { AccumulateK::run(a, blockIdx.x, gridDim.x); }

template <int TB, int OCC, int MODE, bool PROBE>
__global__ void __launch_bounds__(TB, OCC) k_tick(TickArgs a) { TickK<TB, OCC, MODE, PROBE>::run(a, blockIdx.x, gridDim.x); }

// result block of an align(): state + fitness = (sum of the block sums, fixed order) / (points with a neighbour), written to pinned host memory
struct FinalizeFitK {
  static constexpr int TB = 64, OCC = 1;
  struct Args { const GicpState* st; ResultBlock* out; uint32_t* far_stats; const double* fit_psum; const uint32_t* fit_pcnt; int nblk; };
  static __device__ __forceinline__ void run(const Args& a, const uint32_t, const uint32_t) {
    const GicpState* __restrict__ st = a.st; ResultBlock* out = a.out; uint32_t* __restrict__ far_stats = a.far_stats;
    const double* __restrict__ fit_psum = a.fit_psum; const uint32_t* __restrict__ fit_pcnt = a.fit_pcnt; const int nblk = a.nblk;

  const int lane = threadIdx.x;                                     // one wave
  double s = 0; uint32_t c = 0;
  if (st->phase == 2) {
    for (int b = lane * 8; b < min(lane * 8 + 8, nblk); b++) { s += fit_psum[b]; c += fit_pcnt[b]; }      // nblk <= 512 = 64 lanes x 8
    s = wave_sum_f64_dpp(s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  }
  if (lane != 0) return;
  for (int i = 0; i < 16; i++) { out->r.T64[i] = st->x0[i]; out->r.T[i] = (float)st->x0[i]; }
  for (int i = 0; i < 36; i++) out->r.H[i] = st->final_H[i];
  out->r.fitness = c > 0 ? s / c : 1.7976931348623157e308;
  out->r.iterations = st->outer; out->r.converged = st->converged; out->r.lm_failed = st->lm_failed; out->r.reserved = 0;
  out->phase = st->phase; out->trace_len = st->trace_len;
  out->far_requests = far_stats ? far_stats[1] : 0u; out->far_misses = far_stats ? far_stats[0] : 0u; out->far_queries = far_stats ? far_stats[3] : 0u;
  if (far_stats) { far_stats[0] = 0u; far_stats[1] = 0u; far_stats[3] = 0u; }
  double mr = 0, mt = 0;                                             // the latest pose step, as k_finalize reports it (host: does the unseeded phase go on - unseeded_goes_on)
  for (int u = 0; u < 3; u++) { for (int b = 0; b < 3; b++) mr = fmax(mr, fabs(st->delta[4 * u + b] - (u == b ? 1.0 : 0.0))); mt = fmax(mt, fabs(st->delta[4 * u + 3])); }
  out->step_dt = mt; out->step_dr = mr;
  }
};
static __global__ void __launch_bounds__(64) k_finalize_fit(const GicpState* __restrict__ st, ResultBlock* out, uint32_t* __restrict__ far_stats,
                                      const double* __restrict__ fit_psum, const uint32_t* __restrict__ fit_pcnt, int nblk) {
  const FinalizeFitK::Args a{st, out, far_stats, fit_psum, fit_pcnt, nblk};
  FinalizeFitK::run(a, blockIdx.x, gridDim.x);
}

// k_far: the refresh requests of the tick that just ran (bits in far_req), one query per WAVE, chip-wide.  Chunks of 64 source positions
// are dealt round-robin to the blocks (far points are spatially clustered = contiguous in the cell-sorted order; this spreads them) and
// a block's requests go to its waves in position order - a fixed assignment, so the sums below are bitwise reproducible.  Per request:
// exact QN_FAR_M-NN of the transformed point in the target grid (wave_knn_single), candidate list + bound for the following ticks,
// nearest neighbour + tracking record, and the correspondence's contribution to the 28 sums (lane 0).  Each block writes one row of a
// side table; k_far_reduce folds the table (fixed order) into ONE partial row behind the rows of k_tick.
struct FarArgs {
  GridView src, tgt;
  const GicpState* st;                 // the state k_tick published (the pose its body used)
  double thr2;
  int32_t* nn_idx; float4* nn_ref;
  const double* nrm_s; const TargetRec* tgt_rec; const float4* tgt_raw;
  int32_t* cand; float4* cand_ref; float2* cand_b;
  const unsigned long long* far_req;
  double* far_rows;                    // [QN_FAR_BLOCKS][28] side table
  uint32_t ranked_max;                 // request words up to which the requests are ranked globally (QN_FAR_WORDS; 0 = word-per-block distribution: clouds beyond 262144 points, and the test of that path)
  uint32_t* far_stats;
};
struct FarK {
  static constexpr int TB = QN_FAR_THREADS, OCC = 1;
  using Args = FarArgs;
  static __device__ __forceinline__ void run(const FarArgs& a_in, const uint32_t bx, const uint32_t) {
    FarArgs a = a_in;

  __shared__ WaveLdsH1 lds[QN_FAR_THREADS / 64];
  __shared__ double wsum[QN_FAR_THREADS / 64][QN_NPART];
  __shared__ uint32_t wpre[QN_FAR_WORDS];                           // exclusive popcount prefix of the request words
  __shared__ uint32_t wtot[QN_FAR_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  constexpr int NW = QN_FAR_THREADS / 64;
  a.src = grid_resolve(a.src); a.tgt = grid_resolve(a.tgt);
  const int phase = uni(a.st->phase);
  double acc[QN_NPART];
#pragma unroll
  for (int u = 0; u < QN_NPART; u++) acc[u] = 0;
  if (phase == 0 && uni(a.st->pending)) {
    float Tf[12]; double X0[3][4];
#pragma unroll
    for (int j = 0; j < 12; j++) Tf[j] = uni((float)a.st->x0[j]);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 4; c++) X0[r][c] = uni(a.st->x0[4 * r + c]);
    const uint32_t nchunks = (a.src.n + 63u) >> 6;
    // Work distribution.  The requests are bits in far_req (one word per 64 consecutive source positions) and they are CLUSTERED: the part of the source that has
    // no counterpart in the target is one stretch of the cell-sorted order.  Handing words to blocks (the first version) left most of the chip idle while a few
    // blocks worked through dozens of requests per wave.  Instead every block ranks the requests globally (popcount prefix over all words, in LDS) and request e
    // goes to wave e mod (all waves of the launch): balanced, and still a fixed assignment - the sums below are formed in a reproducible order.
    const bool ranked = nchunks <= min((uint32_t)QN_FAR_WORDS, a.ranked_max);
    uint32_t total_req = 0;
    if (ranked && uni(a.far_stats[0]) != 0u) {                            // ([0]: the requests the tick counted - none: nothing to rank, nothing to serve)
      for (uint32_t w = tid; w < nchunks; w += QN_FAR_THREADS) wpre[w] = (uint32_t)__popcll(a.far_req[w]);
      __syncthreads();
      // exclusive prefix over nchunks <= QN_FAR_WORDS values: thread i owns the run [i * PER, (i + 1) * PER)
      constexpr uint32_t PER = (QN_FAR_WORDS + QN_FAR_THREADS - 1) / QN_FAR_THREADS;
      uint32_t loc[PER], sum = 0;
#pragma unroll
      for (uint32_t u = 0; u < PER; u++) { const uint32_t w = tid * PER + u; loc[u] = w < nchunks ? wpre[w] : 0u; sum += loc[u]; }
      uint32_t inc = wave_incl_scan_u32(sum, lane);
      if (lane == 63) wtot[wid] = inc;
      __syncthreads();
      uint32_t base = inc - sum;
      for (int w2 = 0; w2 < wid; w2++) base += wtot[w2];
      for (int w2 = 0; w2 < NW; w2++) total_req += wtot[w2];
      __syncthreads();
#pragma unroll
      for (uint32_t u = 0; u < PER; u++) { const uint32_t w = tid * PER + u; if (w < nchunks) wpre[w] = base; base += loc[u]; }
      __syncthreads();
    }
    const uint32_t gw = bx * NW + wid, ngw = QN_FAR_BLOCKS * NW;
    uint32_t seen = 0;                                              // requests of this block so far (block-uniform; the unranked path)
    uint32_t e = gw;                                                  // ranked path: this wave's next request
    uint32_t ch = ranked ? 0u : bx;
    unsigned long long word = 0;
    for (;;) {
      uint32_t t;
      if (ranked) {
        if (e >= total_req) break;
        uint32_t lo = 0;                                               // last word with wpre[w] <= e
        for (uint32_t step = 1u << 11; step > 0; step >>= 1) { const uint32_t w = lo + step; if (w < nchunks && wpre[w] <= e) lo = w; }
        unsigned long long wd = a.far_req[lo];
        for (uint32_t skip = e - wpre[lo]; skip > 0; skip--) wd &= wd - 1;       // the (e - wpre[lo])-th set bit
        t = lo * 64u + (uint32_t)(__ffsll((long long)wd) - 1);
        e += ngw;
      } else {
        while (word == 0) { if (ch >= nchunks) break; word = a.far_req[ch]; if (word == 0) ch += QN_FAR_BLOCKS; }
        if (word == 0) break;
        t = ch * 64u + (uint32_t)(__ffsll((long long)word) - 1);
        word &= word - 1; if (word == 0) ch += QN_FAR_BLOCKS;
        const bool mine_req = (int)(seen % NW) == wid; seen++;
        if (!mine_req) continue;                                    // request e of the block goes to wave e mod NW
      }
      {
        const float4 p = a.src.pts[t];
        float qx, qy, qz; xform_query<0>(Tf, p.x, p.y, p.z, qx, qy, qz);
        const uint32_t j0 = (uint32_t)a.nn_idx[t];
        float r = 2.f * a.tgt.cell;
        if (j0 < a.tgt.n) { const float4 p0 = a.tgt_raw[j0]; r = sqrtf(sqdist(qx, qy, qz, p0.x, p0.y, p0.z)) * 1.002f + 0.4f * a.tgt.cell; }    // the M nearest of a far query lie within centimetres of the nearest
        int32_t* cl = a.cand + (size_t)t * QN_FAR_M;
        unsigned long long best = QN_INF_KEY; float second = __int_as_float(0x7f800000), bound = 0.f, other = __int_as_float(0x7f800000), b4 = 0.f, b16 = 0.f;
        float dnn = -1.f;                                              // an upper bound on the nearest-neighbour distance
        if (j0 < a.tgt.n) { const float4 p0 = a.tgt_raw[j0]; dnn = sqrtf(sqdist(qx, qy, qz, p0.x, p0.y, p0.z)); }
        else { float du; wave_search_single(a.tgt, qx, qy, qz, r, __int_as_float(0x7f800000), best, second, du, &lds[wid].s); if (best != QN_INF_KEY) dnn = sqrtf(key_d2(best)); other = fminf(sqrtf(second), du); }
        if (dnn >= 0.f) {
          // Everything within R = dnn + m of q becomes the candidate list; everything else is farther than R.  On a surface of ~rho points/m^2 at
          // distance d about 2 pi rho d m points qualify: m ~ 0.35 / d keeps the list at two dozen; a list that overflows halves m.
          // 1 / rho ~ cell^2 / 4 (the grid is sized to ~4 points per surface cell): ~48 points expected at m0; and the list should survive the NEXT
          // step, which is about as long as the one just made (the optimiser converges): at least 2.5 x the move since the last scan
          const float4 lastq = a.nn_ref[t];
          const float moved = sqrtf(sqdist(qx, qy, qz, lastq.x, lastq.y, lastq.z));
          float m = fminf(fmaxf(fmaxf(2.f * a.tgt.cell * a.tgt.cell / fmaxf(dnn, a.tgt.cell), 2.5f * moved), 0.01f), 0.5f * a.tgt.cell);
          for (int attempt = 0; attempt < 5; attempt++, m *= 0.5f) {
            const float R = dnn * 1.000002f + a.tgt.eps + m;
            const uint32_t cnt = wave_ball_collect(a.tgt, qx, qy, qz, R, &lds[wid].s, lds[wid].list, &lds[wid].cnt, best, second);
            other = fminf(sqrtf(second), R * 0.9999995f);              // the runner-up inside the ball, or the ball's radius: nothing else is closer
            if (cnt <= (uint32_t)QN_FAR_M && cnt > 0) {
              // the list in ascending (distance, index) order, and two prefix bounds: every point but the first 4 (16) is at least as far from q as the 5th (17th)
              // - so once the optimiser's steps are millimetres a tick checks FOUR candidates of a far query, not 64
              const unsigned long long own = (uint32_t)lane < cnt ? lds[wid].list[lane] : QN_INF_KEY;
              int rank = 0;
              for (uint32_t f = 0; f < cnt; f++) rank += lds[wid].list[f] < own ? 1 : 0;
              if ((uint32_t)lane < cnt) cl[rank] = (int32_t)key_idx(own); else if (lane < QN_FAR_M) cl[lane] = -1;
              bound = R * 0.9999995f;
              const unsigned long long k5 = wave_min_u64(rank == 4 && own != QN_INF_KEY ? own : QN_INF_KEY), k17 = wave_min_u64(rank == 16 && own != QN_INF_KEY ? own : QN_INF_KEY);
              b4 = k5 != QN_INF_KEY ? fminf(sqrtf(key_d2(k5)) * 0.9999995f, bound) : bound;
              b16 = k17 != QN_INF_KEY ? fminf(sqrtf(key_d2(k17)) * 0.9999995f, bound) : bound;
              break;
            }
          }
        }
        if (lane == 0) {
          a.nn_idx[t] = best != QN_INF_KEY ? (int32_t)key_idx(best) : -1;
          a.nn_ref[t] = make_float4(qx, qy, qz, other);
          a.cand_ref[t] = make_float4(qx, qy, qz, bound); a.cand_b[t] = make_float2(b4, b16);
          if (best != QN_INF_KEY && (double)key_d2(best) < a.thr2) {
            const TargetRec* rec = a.tgt_rec + key_idx(best);
            const double na[3] = {a.nrm_s[(size_t)t * 3], a.nrm_s[(size_t)t * 3 + 1], a.nrm_s[(size_t)t * 3 + 2]};
            const double nb[3] = {rec->n[0], rec->n[1], rec->n[2]};
            accumulate_point_n(X0, X0, make_float4(p.x, p.y, p.z, 1.f), rec->p, na, nb, true, acc);
          }
        }
      }
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int u = 0; u < QN_NPART; u++) wsum[wid][u] = acc[u];
  }
  __syncthreads();
  if (tid < QN_NPART) { double v = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) v += wsum[w][tid]; a.far_rows[(size_t)bx * QN_NPART + tid] = v; }
  }
};
static __global__ void __launch_bounds__(QN_FAR_THREADS) k_far(FarArgs a) { FarK::run(a, blockIdx.x, gridDim.x); }
struct FarReduceK {      // folds k_far's side table into ONE more partial row behind the tick's, then runs the controller step on all of them (the tick left it to this launch)
  static constexpr int TB = QN_FAR_BLOCKS, OCC = 1;
  struct Args { const double* far_rows; double* part; int rows; uint32_t* far_stats; TailArgs tail; };      // part: the tick's row buffer, rows: the rows the tick wrote
  static __device__ __forceinline__ void run(const Args& a, const uint32_t, const uint32_t) {
    __shared__ double sh[QN_FAR_BLOCKS / 32][QN_NPART];
    __shared__ TailLds tl;
    const double* __restrict__ far_rows = a.far_rows; double* part_row = a.part + (size_t)a.rows * QN_NPART; uint32_t* __restrict__ far_stats = a.far_stats;
    const int tid = threadIdx.x, c = tid % 32, seg = tid / 32;          // 16 segments of 32 rows; threads c >= 28 idle
    if (tid == 0) { far_stats[1] = far_stats[0]; far_stats[0] = 0u; }   // [1] = requests of the tick just served (here, not in k_far: every block of k_far reads [0] when it starts)
    if (c < QN_NPART) {
      double v[32];
#pragma unroll
      for (int u = 0; u < 32; u++) v[u] = far_rows[(size_t)(seg * 32 + u) * QN_NPART + c];       // 32 loads in flight
      double acc = 0;
#pragma unroll
      for (int u = 0; u < 32; u++) acc += v[u];
      sh[seg][c] = acc;
    }
    __syncthreads();
    if (tid < QN_NPART) { double v = 0; for (int s2 = 0; s2 < QN_FAR_BLOCKS / 32; s2++) v += sh[s2][tid]; row_store(&part_row[tid], v, true); }
    if (!a.tail.enabled) return;
    // the controller step over the tick's rows + this one (single block: it is its own "last block"; the row above is read back with the same agent-scope loads)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < (int)(sizeof(GicpState) / 8); i += TB) ((unsigned long long*)&tl.sh)[i] = ((const unsigned long long*)a.tail.st_in)[i];
    reduce_rows<TB, true>(a.part, a.rows + 1, tl.part, tl.sums);
    if (tid == 0) {
      const int phase = tl.sh.phase;
      if (tl.sh.pending && phase != 2) { const GicpConfig cfg = a.tail.cfg; solve_controller(&tl.sh, tl.sums, cfg, a.tail.trace, 0, phase, &tl.work); }
      tl.sh.fb_count = 0; tl.sh.big_count = 0; tl.sh.pending = tl.sh.phase != 2 ? 1 : 0;
    }
    __syncthreads();
    for (int i = tid; i < (int)(sizeof(GicpState) / 8); i += TB) ((unsigned long long*)a.tail.st_out)[i] = ((const unsigned long long*)&tl.sh)[i];
  }
};
static __global__ void __launch_bounds__(QN_FAR_BLOCKS) k_far_reduce(FarReduceK::Args a) { FarReduceK::run(a, blockIdx.x, gridDim.x); }

}  // namespace qn
