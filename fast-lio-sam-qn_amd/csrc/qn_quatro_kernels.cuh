// qn_quatro_kernels.cuh - gfx950 kernels of the Quatro coarse stage:
// FPFH descriptors (normals -> SPFH -> FPFH) and the 33-D feature nearest-neighbour search behind
// Matcher::optimizedMatching.  Reference call site: quatro_handler_->align(src, dst, ok),
// fast_lio_sam_qn/src/loop_closure.cpp:144; behaviour restated in SURVEY.md Appendix A.2 (the
// Quatro submodule is empty in the reference tree).
//
// Per-point arrays live in CELL-SORTED order (position t of GridView::pts), so a radius query walks
// contiguous segments of the point array and every neighbour attribute (normal, SPFH row) is fetched
// from the same neighbourhood of memory; k_rows_to_original permutes the final descriptors back.
// No MFMA: the only contraction-like stage (33-D distances) must reproduce a sequential f32 sum
// bit for bit, which the f32 MFMA (different summation tree) would not.
#pragma once
#include "qn_device.cuh"

namespace qn {

#define QN_FROW 36                      // 33 histogram bins padded to 9 x float4

// deterministic f32 atan2 (Cephes atanf reduction + polynomial), identical to oracle qn_atan2f
__device__ __forceinline__ float qn_atanf_pos(float x) {
  float y;
  if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
  else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
  else y = 0.0f;
  const float z = x * x;
  y += (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
  return y;
}
__device__ __forceinline__ float qn_atan2f(float y, float x) {
  const float PI_F = 3.14159265358979323846f;
  if (x == 0.0f) { if (y == 0.0f) return 0.0f; return y > 0.0f ? 1.5707963267948966f : -1.5707963267948966f; }
  const float a = qn_atanf_pos(fabsf(y) / fabsf(x));
  const float r = x > 0.0f ? a : PI_F - a;
  return y < 0.0f ? -r : r;
}

// Walk every point u of the grid whose cell intersects the ball (q, r); body(u, point).  One query per lane.
template <class Body>
__device__ __forceinline__ void for_each_in_ball_cells(const GridView& g, float qx, float qy, float qz, float r, Body&& body) {
  const int bx0 = cell_coord(qx - r, g.ox, g.inv_cell, g.nx), bx1 = cell_coord(qx + r, g.ox, g.inv_cell, g.nx);
  const int by0 = cell_coord(qy - r, g.oy, g.inv_cell, g.ny), by1 = cell_coord(qy + r, g.oy, g.inv_cell, g.ny);
  const int bz0 = cell_coord(qz - r, g.oz, g.inv_cell, g.nz), bz1 = cell_coord(qz + r, g.oz, g.inv_cell, g.nz);
  for (int rz = bz0; rz <= bz1; rz++) for (int ry = by0; ry <= by1; ry++) for (int tx = bx0 >> 3; tx <= (bx1 >> 3); tx++) {
    const int xa = max(bx0, tx << 3), xb = min(bx1, (tx << 3) + 7);
    const uint32_t k0 = cell_key(g, xa, ry, rz);
    const uint32_t s = g.cell_start[k0], e = g.cell_start[k0 + (xb - xa) + 1];
    for (uint32_t u = s; u < e; u++) body(u, g.pts[u]);
  }
}

// The same walk shared by a GROUP of QN_FG consecutive lanes that serve ONE query: lane g of the group takes the candidates g, g + QN_FG, ...
// of every segment (coalesced 16-byte loads).  With one query per lane a 30k-point cloud is 470 waves for 1024 SIMDs and every wave
// walks ~150 candidates through a chain of dependent loads: latency-bound with half the chip empty.  Eight lanes per query give 8x the
// waves and 8x the loads in flight per query.
#define QN_FG 8
template <int FG = QN_FG, class Body>
__device__ __forceinline__ void for_each_in_ball_cells_group(const GridView& g, float qx, float qy, float qz, float r, int gl, Body&& body) {
  const int bx0 = cell_coord(qx - r, g.ox, g.inv_cell, g.nx), bx1 = cell_coord(qx + r, g.ox, g.inv_cell, g.nx);
  const int by0 = cell_coord(qy - r, g.oy, g.inv_cell, g.ny), by1 = cell_coord(qy + r, g.oy, g.inv_cell, g.ny);
  const int bz0 = cell_coord(qz - r, g.oz, g.inv_cell, g.nz), bz1 = cell_coord(qz + r, g.oz, g.inv_cell, g.nz);
  for (int rz = bz0; rz <= bz1; rz++) for (int ry = by0; ry <= by1; ry++) for (int tx = bx0 >> 3; tx <= (bx1 >> 3); tx++) {
    const int xa = max(bx0, tx << 3), xb = min(bx1, (tx << 3) + 7);
    const uint32_t k0 = cell_key(g, xa, ry, rz);
    const uint32_t s = g.cell_start[k0], e = g.cell_start[k0 + (xb - xa) + 1];
    for (uint32_t u = s + gl; u < e; u += FG) body(u, g.pts[u]);
  }
}
// The same walk with the segment bounds fetched UP FRONT (round 5).  The loop above pays two dependent cell_start loads per segment before it can touch a point, segment
// after segment: a 5 x 5 x 5-cell box is ~50 segments of which ~8 hold points (surfaces), so a group spent most of its life waiting for the bounds of empty rows - 70 us for
// a kernel whose arithmetic is a few microseconds.  Here the FG lanes of the group fetch the bounds of ALL segments in parallel (lane gl takes segments gl, gl + FG, ...)
// into a small LDS table of the group, then walk the segments in the SAME order with the SAME dealing of the points to the lanes: body() sees exactly the sequence it saw
// before - bit-identical sums - minus ~40 dependent round trips and ~120 vector instructions of per-segment bookkeeping in every lane (tools/isa_loops.py: the segment loop of
// k_spfh is 121 VALU instructions around a 200-instruction point loop).  Measured, both clouds of a pair: at 100k points k_spfh 0.366 -> 0.339 ms, k_fpfh 0.476 -> 0.435,
// k_normals_group 0.125 -> 0.117; at 30k k_fpfh 0.157 -> 0.145, k_normals_group 0.069 -> 0.064, k_spfh unchanged (0.14: one round of waves, as long as its densest
// neighbourhoods).  Prefetching the next candidate's point on top of it measured SLOWER (0.170 / 0.078) and was dropped.  `seg`: this group's table (QN_SEG_CAP entries; a box with more segments takes the loop above).
// Every lane of the wave must call (the table hand-over is a wave-level LDS fence); on = false: nothing to walk.
#define QN_SEG_CAP 80
template <int FG, class Body>
__device__ __forceinline__ void for_each_in_ball_cells_group_pre(const GridView& g, bool on, float qx, float qy, float qz, float r, int gl, uint2* __restrict__ seg, Body&& body) {
  const int bx0 = cell_coord(qx - r, g.ox, g.inv_cell, g.nx), bx1 = cell_coord(qx + r, g.ox, g.inv_cell, g.nx);
  const int by0 = cell_coord(qy - r, g.oy, g.inv_cell, g.ny), by1 = cell_coord(qy + r, g.oy, g.inv_cell, g.ny);
  const int bz0 = cell_coord(qz - r, g.oz, g.inv_cell, g.nz), bz1 = cell_coord(qz + r, g.oz, g.inv_cell, g.nz);
  const int tx0 = bx0 >> 3, ntx = (bx1 >> 3) - tx0 + 1, ny = by1 - by0 + 1, nz = bz1 - bz0 + 1;
  const int nseg = on ? nz * ny * ntx : 0;
  const bool fits = on && nseg <= QN_SEG_CAP;      // (an inactive lane walks nothing and must not FILL anything either: its box is whatever its placeholder query gives - ADVICE r5: with fits true for nseg = 0 the
                                                   //  fill loop below ran over that box and could write past the group's QN_SEG_CAP entries)
  // lane gl takes the (y, z) rows gl, gl + FG, ... of the box (each row = ntx segments, 1 or 2).  No integer division: a 32-bit division is ~40 instructions on this
  // machine and four of them per segment cost what the table saves; row -> (y, z) through a float reciprocal, exact for these sizes ((r + 0.5) / ny is never an integer).
  const float inv_ny = 1.0f / (float)max(ny, 1);
  if (fits) for (int r = gl; r < ny * nz; r += FG) {
    const int rzi = (int)(((float)r + 0.5f) * inv_ny), ryi = r - rzi * ny;
    for (int it = 0; it < ntx; it++) {
      const int tx = tx0 + it;
      const int xa = max(bx0, tx << 3), xb = min(bx1, (tx << 3) + 7);
      const uint32_t k0 = cell_key(g, xa, by0 + ryi, bz0 + rzi);
      seg[r * ntx + it] = make_uint2(g.cell_start[k0], g.cell_start[k0 + (xb - xa) + 1]);
    }
  }
  wave_lds_fence();
  if (fits) {
    for (int si = 0; si < nseg; si++) { const uint2 se = seg[si]; for (uint32_t u = se.x + gl; u < se.y; u += FG) body(u, g.pts[u]); }
  } else if (on) {
    for (int rz = bz0; rz <= bz1; rz++) for (int ry = by0; ry <= by1; ry++) for (int tx = bx0 >> 3; tx <= (bx1 >> 3); tx++) {
      const int xa = max(bx0, tx << 3), xb = min(bx1, (tx << 3) + 7);
      const uint32_t k0 = cell_key(g, xa, ry, rz);
      const uint32_t s = g.cell_start[k0], e = g.cell_start[k0 + (xb - xa) + 1];
      for (uint32_t u = s + gl; u < e; u += FG) body(u, g.pts[u]);
    }
  }
  wave_lds_fence();                                                          // (the table is reused by the group's next walk, if any)
}
// sums over the FG (8 or 16) consecutive lanes of a group: fixed butterfly, deterministic
template <int FG = QN_FG> __device__ __forceinline__ int group_sum_i(int v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); if (FG > 8) v += __shfl_xor(v, 8); return v; }
template <int FG = QN_FG> __device__ __forceinline__ double group_sum_d(double v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); if (FG > 8) v += __shfl_xor(v, 8); return v; }

// K9: PCL NormalEstimation, radius search.  normals[t] = (nx, ny, nz, 1) or NaNs.
static __global__ void __launch_bounds__(QN_BLOCK) k_normals(GridView g, float r, float r2, float4* __restrict__ normals) {
  g = grid_resolve(g);
  const uint32_t t = blockIdx.x * QN_BLOCK + threadIdx.x;
  if (t >= g.n) return;
  const float4 p = g.pts[t];
  int cnt = 0; double s[3] = {0, 0, 0}, c[6] = {0, 0, 0, 0, 0, 0};
  for_each_in_ball_cells(g, p.x, p.y, p.z, r, [&](uint32_t, float4 q) __attribute__((always_inline)) {
    if (sqdist(p.x, p.y, p.z, q.x, q.y, q.z) < r2) {
      const double dx = (double)q.x - (double)p.x, dy = (double)q.y - (double)p.y, dz = (double)q.z - (double)p.z;   // relative to p: no cancellation
      cnt++; s[0] += dx; s[1] += dy; s[2] += dz;
      c[0] += dx * dx; c[1] += dx * dy; c[2] += dx * dz; c[3] += dy * dy; c[4] += dy * dz; c[5] += dz * dz;
    }
  });
  const float qnan = __int_as_float(0x7fc00000);
  if (cnt < 3) { normals[t] = make_float4(qnan, qnan, qnan, 0.f); return; }
  const double inv = 1.0 / cnt, mx = s[0] * inv, my = s[1] * inv, mz = s[2] * inv;
  const double cov[6] = {c[0] * inv - mx * mx, c[1] * inv - mx * my, c[2] * inv - mx * mz, c[3] * inv - my * my, c[4] * inv - my * mz, c[5] * inv - mz * mz};
  double w[3], V[3][3]; sym_eig3(cov, w, V);
  double nx = V[0][2], ny = V[1][2], nz = V[2][2];
  if (-(nx * (double)p.x + ny * (double)p.y + nz * (double)p.z) < 0) { nx = -nx; ny = -ny; nz = -nz; }   // flip towards the viewpoint (0,0,0)
  normals[t] = make_float4((float)nx, (float)ny, (float)nz, 1.f);
}

// The same with FG lanes per query (round 5): at 30k points the one-query-per-lane kernel is 470 waves for 1024 SIMDs, each lane walking ~60 candidates through dependent
// loads - 54 us of pure latency, 0.8 % of the nominal roof.  Lane gl of a group takes the candidates gl, gl + FG, ... of every segment; the count is an integer, the nine
// f64 sums of each lane meet in a fixed butterfly.  The sums are formed in a different association than the sequential kernel's (and the oracle's index order): normals
// agree to the last bit of f64 rounding, i.e. a handful of f32 normals per cloud differ in their last bit - the tolerance the FPFH parity tests already carry against the oracle.
template <int FG>
__device__ __forceinline__ void normals_group_body(GridView g, const float r, const float r2, float4* __restrict__ normals, const uint32_t bx) {
  g = grid_resolve(g);
  const uint32_t t0 = (bx * QN_BLOCK + threadIdx.x) / FG; const int gl = threadIdx.x & (FG - 1);
  const bool in_range = t0 < g.n;
  const uint32_t t = in_range ? t0 : 0;
  const float4 p = g.pts[t];
  int cnt = 0; double s[3] = {0, 0, 0}, c[6] = {0, 0, 0, 0, 0, 0};
  __shared__ uint2 seg_tab[QN_BLOCK / FG][QN_SEG_CAP];
  for_each_in_ball_cells_group_pre<FG>(g, in_range, p.x, p.y, p.z, r, gl, seg_tab[threadIdx.x / FG], [&](uint32_t, float4 q) __attribute__((always_inline)) {
    if (sqdist(p.x, p.y, p.z, q.x, q.y, q.z) < r2) {
      const double dx = (double)q.x - (double)p.x, dy = (double)q.y - (double)p.y, dz = (double)q.z - (double)p.z;
      cnt++; s[0] += dx; s[1] += dy; s[2] += dz;
      c[0] += dx * dx; c[1] += dx * dy; c[2] += dx * dz; c[3] += dy * dy; c[4] += dy * dz; c[5] += dz * dz;
    }
  });
  cnt = group_sum_i<FG>(cnt);
#pragma unroll
  for (int a = 0; a < 3; a++) s[a] = group_sum_d<FG>(s[a]);
#pragma unroll
  for (int a = 0; a < 6; a++) c[a] = group_sum_d<FG>(c[a]);
  if (!in_range || gl != 0) return;
  const float qnan = __int_as_float(0x7fc00000);
  if (cnt < 3) { normals[t] = make_float4(qnan, qnan, qnan, 0.f); return; }
  const double inv = 1.0 / cnt, mx = s[0] * inv, my = s[1] * inv, mz = s[2] * inv;
  const double cov[6] = {c[0] * inv - mx * mx, c[1] * inv - mx * my, c[2] * inv - mx * mz, c[3] * inv - my * my, c[4] * inv - my * mz, c[5] * inv - mz * mz};
  double w[3], V[3][3]; sym_eig3(cov, w, V);
  double nx = V[0][2], ny = V[1][2], nz = V[2][2];
  if (-(nx * (double)p.x + ny * (double)p.y + nz * (double)p.z) < 0) { nx = -nx; ny = -ny; nz = -nz; }
  normals[t] = make_float4((float)nx, (float)ny, (float)nz, 1.f);
}
template <int FG>
static __global__ void __launch_bounds__(QN_BLOCK) k_normals_group(GridView g, float r, float r2, float4* __restrict__ normals) { normals_group_body<FG>(g, r, r2, normals, blockIdx.x); }
// the same as a functor for k_lanes<F> (the batched coarse-to-fine path carries the clouds of a whole run of candidate pairs in ONE launch)
template <int FG> struct NormalsK {
  static constexpr int TB = QN_BLOCK, OCC = 1;
  struct Args { GridView g; float r, r2; float4* normals; };
  static __device__ __forceinline__ void run(const Args& a, const uint32_t bx, const uint32_t) { normals_group_body<FG>(a.g, a.r, a.r2, a.normals, bx); }
};

// pcl::computePairFeatures in f32 with the oracle's operation order
__device__ __forceinline__ bool pair_features(const float4 p1, const float4 n1, const float4 p2, const float4 n2, float& f1, float& f2, float& f3) {
  float dx = p2.x - p1.x, dy = p2.y - p1.y, dz = p2.z - p1.z;
  const float f4 = sqrtf((dx * dx + dy * dy) + dz * dz);
  if (f4 == 0.0f) return false;
  const float angle1 = ((n1.x * dx + n1.y * dy) + n1.z * dz) / f4;
  const float angle2 = ((n2.x * dx + n2.y * dy) + n2.z * dz) / f4;
  float ax = n1.x, ay = n1.y, az = n1.z, bx = n2.x, by = n2.y, bz = n2.z;
  if (fabsf(angle1) < fabsf(angle2)) { ax = n2.x; ay = n2.y; az = n2.z; bx = n1.x; by = n1.y; bz = n1.z; dx = -dx; dy = -dy; dz = -dz; f3 = -angle2; }
  else f3 = angle1;
  float vx = dy * az - dz * ay, vy = dz * ax - dx * az, vz = dx * ay - dy * ax;
  const float vn = sqrtf((vx * vx + vy * vy) + vz * vz);
  if (vn == 0.0f) return false;
  vx /= vn; vy /= vn; vz /= vn;
  const float wx = ay * vz - az * vy, wy = az * vx - ax * vz, wz = ax * vy - ay * vx;
  f2 = (vx * bx + vy * by) + vz * bz;
  f1 = qn_atan2f((wx * bx + wy * by) + wz * bz, (ax * bx + ay * by) + az * bz);
  return true;
}

// K10: SPFH - 3 x 11-bin histograms of (theta, alpha, phi) over the r_f neighbourhood; bin = count * 100 / (n_nbrs - 1).
// QN_FG lanes per query: integer counts, so the split of the neighbours over the lanes changes nothing.
template <int FG>
__device__ __forceinline__ void spfh_body(GridView g, const float r, const float r2, const float4* __restrict__ normals, float* __restrict__ spfh, const uint32_t bx) {
  g = grid_resolve(g);
  const uint32_t t0 = (bx * QN_BLOCK + threadIdx.x) / FG; const int gl = threadIdx.x & (FG - 1);
  const bool in_range = t0 < g.n;
  const uint32_t t = in_range ? t0 : 0;
  const float4 p = g.pts[t], np = normals[t];
  float* out = spfh + (size_t)t * QN_FROW;
  const bool active = in_range && (np.x == np.x);
  int cnt[33];
#pragma unroll
  for (int b = 0; b < 33; b++) cnt[b] = 0;
  int nn = 0;
  const float d_pi = 1.0f / (2.0f * 3.14159265358979323846f);
  __shared__ uint2 seg_tab[QN_BLOCK / FG][QN_SEG_CAP];
  for_each_in_ball_cells_group_pre<FG>(g, active, p.x, p.y, p.z, r, gl, seg_tab[threadIdx.x / FG], [&](uint32_t u, float4 q) __attribute__((always_inline)) {
    if (!(sqdist(p.x, p.y, p.z, q.x, q.y, q.z) < r2)) return;
    nn++;
    if (u == t) return;
    const float4 nq = normals[u];
    if (!(nq.x == nq.x)) return;
    float f1, f2, f3;
    if (!pair_features(p, np, q, nq, f1, f2, f3)) return;
    const int h1 = min(max((int)floor(11.0 * (((double)f1 + 3.14159265358979323846) * (double)d_pi)), 0), 10);
    const int h2 = min(max((int)floor(11.0 * (((double)f2 + 1.0) * 0.5)), 0), 10);
    const int h3 = min(max((int)floor(11.0 * (((double)f3 + 1.0) * 0.5)), 0), 10);
#pragma unroll
    for (int b = 0; b < 11; b++) { cnt[b] += (h1 == b); cnt[11 + b] += (h2 == b); cnt[22 + b] += (h3 == b); }
  });
  nn = group_sum_i<FG>(nn);
#pragma unroll
  for (int b = 0; b < 33; b++) cnt[b] = group_sum_i<FG>(cnt[b]);
  if (!in_range) return;
  const float incr = 100.0f / (float)(nn - 1);
  // the group writes the row together: lane gl takes the slots gl, gl + FG, ...
#pragma unroll
  for (int b = 0; b < QN_FROW; b++) if ((b & (FG - 1)) == gl) out[b] = (active && b < 33 && cnt[b < 33 ? b : 0] > 0) ? (float)cnt[b < 33 ? b : 0] * incr : 0.f;
}
template <int FG>
static __global__ void __launch_bounds__(QN_BLOCK) k_spfh(GridView g, float r, float r2, const float4* __restrict__ normals, float* __restrict__ spfh) { spfh_body<FG>(g, r, r2, normals, spfh, blockIdx.x); }
template <int FG> struct SpfhK {
  static constexpr int TB = QN_BLOCK, OCC = 1;
  struct Args { GridView g; float r, r2; const float4* normals; float* spfh; };
  static __device__ __forceinline__ void run(const Args& a, const uint32_t bx, const uint32_t) { spfh_body<FG>(a.g, a.r, a.r2, a.normals, a.spfh, bx); }
};

// K11: FPFH(p) = sum_q SPFH(q) / d2(p, q) over the r_f neighbourhood (d2 > 0), each 11-bin group normalised to 100.
// QN_FG lanes per query: each lane sums its share of the neighbours in f64, the eight partial sums meet in a fixed butterfly.  (The f64
// sums of f32-sized terms are rounded to f32 at the end: the result does not depend on the order except when a sum sits within 1e-16 of a
// rounding boundary - no difference against the sequential oracle on any cloud tried; the parity tests hold it to 1e-4 per bin.)
template <int FG>
__device__ __forceinline__ void fpfh_body(GridView g, const float r, const float r2, const float4* __restrict__ normals, const float* __restrict__ spfh, float* __restrict__ fpfh, const uint32_t bx) {
  g = grid_resolve(g);
  const uint32_t t0 = (bx * QN_BLOCK + threadIdx.x) / FG; const int gl = threadIdx.x & (FG - 1);
  const bool in_range = t0 < g.n;
  const uint32_t t = in_range ? t0 : 0;
  const float4 p = g.pts[t], np = normals[t];
  float* out = fpfh + (size_t)t * QN_FROW;
  const float qnan = __int_as_float(0x7fc00000);
  double acc[33];
#pragma unroll
  for (int b = 0; b < 33; b++) acc[b] = 0.0;
  __shared__ uint2 seg_tab[QN_BLOCK / FG][QN_SEG_CAP];
  {
    for_each_in_ball_cells_group_pre<FG>(g, in_range && np.x == np.x, p.x, p.y, p.z, r, gl, seg_tab[threadIdx.x / FG], [&](uint32_t u, float4 q) __attribute__((always_inline)) {
      const float d2 = sqdist(p.x, p.y, p.z, q.x, q.y, q.z);
      if (!(d2 < r2) || d2 == 0.0f) return;
      const float w = 1.0f / d2;
      const float4* s = (const float4*)(spfh + (size_t)u * QN_FROW);
#pragma unroll
      for (int v = 0; v < 9; v++) {
        const float4 x = s[v];
        if (4 * v + 0 < 33) acc[4 * v + 0] += (double)(x.x * w);
        if (4 * v + 1 < 33) acc[4 * v + 1] += (double)(x.y * w);
        if (4 * v + 2 < 33) acc[4 * v + 2] += (double)(x.z * w);
        if (4 * v + 3 < 33) acc[4 * v + 3] += (double)(x.w * w);
      }
    });
  }
#pragma unroll
  for (int b = 0; b < 33; b++) acc[b] = group_sum_d<FG>(acc[b]);
  if (!in_range) return;
  double sum[3] = {0, 0, 0};
#pragma unroll
  for (int b = 0; b < 33; b++) sum[b / 11] += acc[b];
  const bool dead = !(np.x == np.x) || sum[0] == 0.0;
#pragma unroll
  for (int b = 0; b < QN_FROW; b++) if ((b & (FG - 1)) == gl)
    out[b] = b >= 33 ? 0.f : (dead ? qnan : (float)(acc[b < 33 ? b : 0] * (sum[(b < 33 ? b : 0) / 11] != 0.0 ? 100.0 / sum[(b < 33 ? b : 0) / 11] : 0.0)));
}
template <int FG>
static __global__ void __launch_bounds__(QN_BLOCK) k_fpfh(GridView g, float r, float r2, const float4* __restrict__ normals, const float* __restrict__ spfh, float* __restrict__ fpfh) { fpfh_body<FG>(g, r, r2, normals, spfh, fpfh, blockIdx.x); }
template <int FG> struct FpfhK {
  static constexpr int TB = QN_BLOCK, OCC = 1;
  struct Args { GridView g; float r, r2; const float4* normals; const float* spfh; float* fpfh; };
  static __device__ __forceinline__ void run(const Args& a, const uint32_t bx, const uint32_t) { fpfh_body<FG>(a.g, a.r, a.r2, a.normals, a.spfh, a.fpfh, bx); }
};

struct RowsToOriginalK {      // k_rows_to_original (below) for k_lanes<F>
  static constexpr int TB = QN_BLOCK, OCC = 1;
  struct Args { const float4* pts; uint32_t n; const float* in; float* out; int w_in, w_out; };
  static __device__ __forceinline__ void run(const Args& a, const uint32_t bx, const uint32_t) {
    const uint32_t t = bx * QN_BLOCK + threadIdx.x;
    if (t >= a.n) return;
    const uint32_t i = __float_as_uint(a.pts[t].w);
    for (int b = 0; b < a.w_out; b++) a.out[(size_t)i * a.w_out + b] = b < a.w_in ? a.in[(size_t)t * a.w_in + b] : 0.f;
  }
};
// sorted-position rows -> original-index rows (rows of `w` floats)
static __global__ void k_rows_to_original(const float4* __restrict__ pts, uint32_t n, const float* __restrict__ in, float* __restrict__ out, int w_in, int w_out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const uint32_t i = __float_as_uint(pts[t].w);
  for (int b = 0; b < w_out; b++) out[(size_t)i * w_out + b] = b < w_in ? in[(size_t)t * w_in + b] : 0.f;
}

}  // namespace qn
#include "qn_feat_mm.cuh"
namespace qn {
// 32-bit hash of a descriptor row's 33 bit patterns into slot 34 of the row (k_feat_nn recognises duplicate rows by it)
static __global__ void k_row_hash(float* __restrict__ rows, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float* r = rows + (size_t)i * QN_FROW;
  uint32_t h = 2166136261u;
#pragma unroll
  for (int d = 0; d < 33; d++) { h ^= __float_as_uint(r[d]); h *= 16777619u; h ^= h >> 15; }
  r[34] = __uint_as_float(h);
}

// K12: exact nearest neighbour in 33-D (f32 sequential sum over the dimensions, ties -> lowest candidate index).
// grid.x tiles the queries (256 per block), grid.y splits the candidates; partial winners meet in a 64-bit
// atomicMin on (distance bits << 32 | index).  Queries optionally come through an index list.
//
// The answer is defined by the sequential non-fused sum (the oracle's arithmetic).  Computing THAT for every pair costs 99 VALU
// instructions per candidate (sub, mul, add x 33).  Instead every pair of candidates first gets a screening distance from packed-f32
// FMAs (v_pk_add_f32 + v_pk_fma_f32: two candidates per instruction, 33 instructions per candidate) - same dimension order, so it
// differs from the defining sum by at most ~33 x 2^-24 relative - and only a candidate whose screening distance is below the current
// best x (1 + 1e-5) is re-evaluated with the defining arithmetic before it may replace the best.  No improvement can be missed
// (exact < best implies screen < best (1 + 4e-6)), and what is kept is always an exactly evaluated distance: bit-identical results.
// Duplicate descriptors (every point of an exact plane has the same FPFH row - 40 % of a noise-free synthetic cloud) would sit inside that
// band for ever: a candidate whose screening value AND 32-bit row hash (slot 34 of its row, k_row_hash) equal those of the current best
// is the same row, has the same distance, and - coming later in the ascending scan - can never replace it: skipped without the
// re-evaluation.  (Two DIFFERENT rows with equal screening value, equal hash and a sub-1e-5 relative distance difference would be
// mistaken for duplicates: a ~1e-13 event per search, far below the ambiguity of the specification itself - FLANN sums in a different
// order - and the only non-proof in this kernel.)
#define QN_FM_TILE 64                   // candidate ranges of the grid.y split are multiples of this
typedef float qn_v2f __attribute__((ext_vector_type(2)));
// Candidate rows re-laid in PAIRS for the screening: pair p holds candidates 2p, 2p+1 as 34 x (c0[d], c1[d]) (dimension 33 = padding 0;
// a missing second candidate = 3e18, never below any threshold) + their row hashes.  The candidate pointer of k_feat_nn is wave-uniform,
// so the rows arrive through SCALAR loads (s_load_dwordx16) into SGPRs and feed v_pk_add_f32 / v_pk_fma_f32 directly: no LDS tile - a
// broadcast ds_read_b128 still writes 64 x 16 bytes of registers, and with four SIMDs sharing one LDS the tile version of this kernel
// was LDS-bound at a quarter of the packed-f32 rate.
static __global__ void k_pair_rows(const float* __restrict__ rows, uint32_t n, qn_v2f* __restrict__ pairs, uint32_t* __restrict__ hashes) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;          // one thread per (pair, dimension)
  const uint32_t np = (n + 1) / 2;
  if (e >= np * 34) return;
  const uint32_t p = e / 34, d = e - p * 34;
  const uint32_t c0 = 2 * p, c1 = 2 * p + 1;
  qn_v2f v;
  v.x = d < 33 ? rows[(size_t)c0 * QN_FROW + d] : 0.f;
  v.y = d < 33 ? (c1 < n ? rows[(size_t)c1 * QN_FROW + d] : 3.0e18f) : 0.f;
  pairs[(size_t)p * 34 + d] = v;
  if (d == 0) { hashes[c0] = __float_as_uint(rows[(size_t)c0 * QN_FROW + 34]); hashes[c1] = c1 < n ? __float_as_uint(rows[(size_t)c1 * QN_FROW + 34]) : 0u; }
}

// QN_FM_QPL queries per lane: every scalar-loaded candidate pair (272 bytes) is used for 64 x QPL queries - with one query per lane the
// kernel was bound by the scalar cache (each wave streams the whole candidate set), not by the packed-f32 pipe.
#define QN_FM_QPL 1
static __global__ void __launch_bounds__(QN_BLOCK) k_feat_nn(const float* __restrict__ Q, uint32_t nq, const uint32_t* __restrict__ qlist, const uint32_t* __restrict__ qlist_n,
                                                      const qn_v2f* __restrict__ Cp, const uint32_t* __restrict__ Ch, uint32_t nc, uint32_t chunk, unsigned long long* __restrict__ best_key) {
  const uint32_t nqueries = qlist ? *qlist_n : nq;
  if (blockIdx.x * (QN_BLOCK * QN_FM_QPL) >= nqueries) return;
  float q[QN_FM_QPL][34]; uint32_t qi[QN_FM_QPL]; bool qok[QN_FM_QPL];
#pragma unroll
  for (int u = 0; u < QN_FM_QPL; u++) {
    const uint32_t slot = (blockIdx.x * QN_FM_QPL + u) * QN_BLOCK + threadIdx.x;
    const bool active = slot < nqueries;
    qi[u] = active ? (qlist ? qlist[slot] : slot) : 0;
    const float4* qrow = (const float4*)(Q + (size_t)qi[u] * QN_FROW);
#pragma unroll
    for (int v = 0; v < 9; v++) { const float4 x = qrow[v]; q[u][4 * v] = x.x; if (4 * v + 1 < 34) q[u][4 * v + 1] = x.y; if (4 * v + 2 < 34) q[u][4 * v + 2] = x.z; if (4 * v + 3 < 34) q[u][4 * v + 3] = x.w; }
    q[u][33] = 0.f;
    qok[u] = active && (q[u][0] == q[u][0]);
  }
  const uint32_t c0 = blockIdx.y * chunk, c1 = min(nc, c0 + chunk);        // chunk is even
  float best[QN_FM_QPL], bs[QN_FM_QPL]; uint32_t bi[QN_FM_QPL], bh[QN_FM_QPL];      // best exact distance / its screening value and row hash
#pragma unroll
  for (int u = 0; u < QN_FM_QPL; u++) { best[u] = __int_as_float(0x7f7fffff); bi[u] = 0xffffffffu; bs[u] = -1.f; bh[u] = 0; }
  for (uint32_t p = c0 >> 1; p < (c1 + 1) >> 1; p++) {
    const qn_v2f* __restrict__ row = Cp + (size_t)p * 34;                 // wave-uniform: scalar loads
    const uint32_t h0 = Ch[2 * p], h1 = Ch[2 * p + 1];
    qn_v2f s[QN_FM_QPL], s2[QN_FM_QPL];                              // two accumulation chains (even / odd dimensions): the FMA chain is latency-bound otherwise
#pragma unroll
    for (int u = 0; u < QN_FM_QPL; u++) { s[u] = (qn_v2f){0.f, 0.f}; s2[u] = (qn_v2f){0.f, 0.f}; }
#pragma unroll
    for (int d = 0; d < 34; d += 2) {
      const qn_v2f r = row[d], r2 = row[d + 1];
#pragma unroll
      for (int u = 0; u < QN_FM_QPL; u++) {
        const qn_v2f df = (qn_v2f){q[u][d], q[u][d]} - r; s[u] = __builtin_elementwise_fma(df, df, s[u]);
        const qn_v2f dg = (qn_v2f){q[u][d + 1], q[u][d + 1]} - r2; s2[u] = __builtin_elementwise_fma(dg, dg, s2[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < QN_FM_QPL; u++) s[u] = s[u] + s2[u];
#pragma unroll
    for (int u = 0; u < QN_FM_QPL; u++) {
      const float thr = best[u] * 1.00001f;
      if ((s[u].x < thr && !(s[u].x == bs[u] && h0 == bh[u])) || (s[u].y < thr && !(s[u].y == bs[u] && h1 == bh[u]))) {     // rare after the first few candidates: the defining arithmetic
#pragma unroll 1
        for (int h = 0; h < 2; h++) {
          const float sv = h == 0 ? s[u].x : s[u].y; const uint32_t hv = h == 0 ? h0 : h1;
          if (sv < thr && !(sv == bs[u] && hv == bh[u]) && 2 * p + h < nc) {
            float e = 0.f;
#pragma unroll
            for (int d = 0; d < 33; d++) { const float t = q[u][d] - (h == 0 ? row[d].x : row[d].y); e = e + t * t; }
            if (e < best[u]) { best[u] = e; bi[u] = 2 * p + h; bs[u] = sv; bh[u] = hv; }      // NaN never wins; ascending scan keeps the lowest index
          }
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < QN_FM_QPL; u++)
    if (qok[u] && bi[u] != 0xffffffffu) atomicMin(&best_key[qi[u]], ((unsigned long long)__float_as_uint(best[u]) << 32) | bi[u]);
}

static __global__ void k_fill_u64(unsigned long long* p, uint32_t n, unsigned long long v) { uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }

// mark the candidates that were somebody's nearest neighbour and compact them into a query list
static __global__ void k_mark_hits(const unsigned long long* __restrict__ j_key, uint32_t nj, uint32_t* __restrict__ hit) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nj) return;
  const unsigned long long k = j_key[j];
  if (k != QN_INF_KEY) hit[(uint32_t)k] = 1u;
}
static __global__ void k_compact_hits(const uint32_t* __restrict__ hit, uint32_t ni, uint32_t* __restrict__ list, uint32_t* __restrict__ count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ni && hit[i]) list[atomicAdd(count, 1u)] = i;
}
// ------------------------------------------------------------------ K13: the Matcher tail on the device
// (cross-check, distance gate, seeded tuple test, cap - Matcher::optimizedMatching / advancedMatching).
// Everything that is O(N) or O(ncorr x 100) stays on the GPU; what crosses PCIe at the end is one record per SELECTED
// correspondence (<= ~203 x 32 bytes with the reference's cap).

// normalizePoints (absolute scale): each cloud's own mean, f64 sums in a fixed tree, rounded to f32
#define QN_MEAN_BLOCKS 64
static __global__ void __launch_bounds__(QN_BLOCK) k_mean_partial(const float4* __restrict__ pts, uint32_t n, double* __restrict__ psum) {
  __shared__ double sh[QN_BLOCK / 64][3];
  double s[3] = {0, 0, 0};
  for (uint32_t i = blockIdx.x * QN_BLOCK + threadIdx.x; i < n; i += QN_MEAN_BLOCKS * QN_BLOCK) { const float4 p = pts[i]; s[0] += (double)p.x; s[1] += (double)p.y; s[2] += (double)p.z; }
#pragma unroll
  for (int d = 0; d < 3; d++) s[d] = wave_sum_f64_dpp(s[d]);
  if ((threadIdx.x & 63) == 0) { for (int d = 0; d < 3; d++) sh[threadIdx.x >> 6][d] = s[d]; }
  __syncthreads();
  if (threadIdx.x < 3) { double t = 0; for (int w = 0; w < QN_BLOCK / 64; w++) t += sh[w][threadIdx.x]; psum[blockIdx.x * 3 + threadIdx.x] = t; }
}
static __global__ void k_mean_final(const double* __restrict__ psum, uint32_t n, float* __restrict__ mean3) {
  if (threadIdx.x >= 3 || blockIdx.x != 0) return;
  double t = 0; for (int b = 0; b < QN_MEAN_BLOCKS; b++) t += psum[b * 3 + threadIdx.x];
  mean3[threadIdx.x] = (float)(t / (double)n);
}

// ---- fused bookkeeping launches of the matching stage (round 5).  A quatro::align at 30k points used to issue 26 launches of ~5 us each (memsets, fills, row
// hashes, counters, means) between its real kernels: 130 us of a 0.95 ms align.  The independent ones are one grid-stride kernel each now.
// k_match_init: everything the stage needs zeroed / filled / hashed before the forward search.
struct MatchInitArgs {
  uint32_t* counts; uint32_t n_counts;                     // q_counts: hit counter, query-list counter, mean tickets
  uint32_t* hit; uint32_t ni;                              // one flag per point of the larger cloud
  unsigned long long* key_j; uint32_t nj; unsigned long long* key_i;      // best keys of both searches, start at QN_INF_KEY
  float* rows0; uint32_t n0; float* rows1; uint32_t n1;    // descriptor sets: 32-bit row hash into slot 34 (k_row_hash)
  unsigned long long* mm_table; uint32_t mm_table_n; uint32_t* mm_L; uint32_t mm_L_n; uint32_t* mm_cnt; uint32_t mm_cnt_n;      // matrix-core search: de-duplication table (all ones), lower bounds, counters (null: VALU search)
};
__device__ __forceinline__ void row_hash_one(float* __restrict__ r) {
  uint32_t h = 2166136261u;
#pragma unroll
  for (int d = 0; d < 33; d++) { h ^= __float_as_uint(r[d]); h *= 16777619u; h ^= h >> 15; }
  r[34] = __uint_as_float(h);
}
static __global__ void __launch_bounds__(256) k_match_init(MatchInitArgs a) {
  const uint32_t t0 = blockIdx.x * 256u + threadIdx.x, stride = gridDim.x * 256u;
  for (uint32_t i = t0; i < a.n_counts; i += stride) a.counts[i] = 0u;
  for (uint32_t i = t0; i < a.ni; i += stride) { a.hit[i] = 0u; a.key_i[i] = QN_INF_KEY; }
  for (uint32_t i = t0; i < a.nj; i += stride) a.key_j[i] = QN_INF_KEY;
  for (uint32_t i = t0; i < a.n0; i += stride) row_hash_one(a.rows0 + (size_t)i * QN_FROW);
  for (uint32_t i = t0; i < a.n1; i += stride) row_hash_one(a.rows1 + (size_t)i * QN_FROW);
  if (a.mm_table) {
    for (uint32_t i = t0; i < a.mm_table_n; i += stride) a.mm_table[i] = ~0ull;
    for (uint32_t i = t0; i < a.mm_L_n; i += stride) a.mm_L[i] = 0u;
    for (uint32_t i = t0; i < a.mm_cnt_n; i += stride) a.mm_cnt[i] = 0u;
  }
}
// between the two searches: the forward winners mark their candidates (k_mark_hits), the forward survivor count is handed over, and the matrix-core search's table and
// lower bounds are reset for the reverse search (its counters are reset by k_compact_hits_reset, the next launch: the count is read here first)
static __global__ void __launch_bounds__(256) k_mark_hits_reset(const unsigned long long* __restrict__ j_key, uint32_t nj, uint32_t* __restrict__ hit,
                                                                const uint32_t* __restrict__ mm_cnt, uint32_t* __restrict__ survivors_out,
                                                                unsigned long long* __restrict__ mm_table, uint32_t mm_table_n, uint32_t* __restrict__ mm_L, uint32_t mm_L_n) {
  const uint32_t t0 = blockIdx.x * 256u + threadIdx.x, stride = gridDim.x * 256u;
  if (t0 == 0 && mm_cnt) *survivors_out = mm_cnt[0];
  for (uint32_t j = t0; j < nj; j += stride) { const unsigned long long k = j_key[j]; if (k != QN_INF_KEY) hit[(uint32_t)k] = 1u; }
  if (mm_table) {
    for (uint32_t i = t0; i < mm_table_n; i += stride) mm_table[i] = ~0ull;
    for (uint32_t i = t0; i < mm_L_n; i += stride) mm_L[i] = 0u;
  }
}
static __global__ void __launch_bounds__(256) k_compact_hits_reset(const uint32_t* __restrict__ hit, uint32_t ni, uint32_t* __restrict__ list, uint32_t* __restrict__ count,
                                                                   uint32_t* __restrict__ mm_cnt, uint32_t mm_cnt_n) {
  const uint32_t t0 = blockIdx.x * 256u + threadIdx.x, stride = gridDim.x * 256u;
  for (uint32_t i = t0; i < ni; i += stride) if (hit[i]) list[atomicAdd(count, 1u)] = i;
  if (mm_cnt) for (uint32_t i = t0; i < mm_cnt_n; i += stride) mm_cnt[i] = 0u;
}
// normalizePoints' means of BOTH clouds in one launch: 2 x QN_MEAN_BLOCKS blocks form the partial sums (k_mean_partial's), the last block of each cloud to draw its
// ticket sums them in the fixed order of k_mean_final - same bits - and zeroes the ticket for the next align
static __global__ void __launch_bounds__(QN_BLOCK) k_means2(const float4* __restrict__ p0, uint32_t n0, const float4* __restrict__ p1, uint32_t n1,
                                                            double* __restrict__ psum /* [2][QN_MEAN_BLOCKS][3] */, uint32_t* __restrict__ tickets /* [2], zero */, float* __restrict__ mean /* [2][4] */) {
  __shared__ double sh[QN_BLOCK / 64][3];
  __shared__ uint32_t last_sh;
  const int w = blockIdx.x >= QN_MEAN_BLOCKS ? 1 : 0; const uint32_t b = blockIdx.x - (uint32_t)w * QN_MEAN_BLOCKS;
  const float4* __restrict__ pts = w ? p1 : p0; const uint32_t n = w ? n1 : n0;
  double* __restrict__ ps = psum + (size_t)w * QN_MEAN_BLOCKS * 3;
  double s[3] = {0, 0, 0};
  for (uint32_t i = b * QN_BLOCK + threadIdx.x; i < n; i += QN_MEAN_BLOCKS * QN_BLOCK) { const float4 p = pts[i]; s[0] += (double)p.x; s[1] += (double)p.y; s[2] += (double)p.z; }
#pragma unroll
  for (int d = 0; d < 3; d++) s[d] = wave_sum_f64_dpp(s[d]);
  if ((threadIdx.x & 63) == 0) { for (int d = 0; d < 3; d++) sh[threadIdx.x >> 6][d] = s[d]; }
  __syncthreads();
  if (threadIdx.x < 3) { double t = 0; for (int v = 0; v < QN_BLOCK / 64; v++) t += sh[v][threadIdx.x]; __hip_atomic_store(&ps[b * 3 + threadIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last_sh = atomicAdd(&tickets[w], 1u) == QN_MEAN_BLOCKS - 1 ? 1u : 0u;
  __syncthreads();
  if (!last_sh) return;
  __threadfence();
  if (threadIdx.x < 3) {
    double t = 0; for (int v = 0; v < QN_MEAN_BLOCKS; v++) t += __hip_atomic_load(&ps[v * 3 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    mean[4 * w + threadIdx.x] = (float)(t / (double)n);
  }
  if (threadIdx.x == 0) tickets[w] = 0u;
}

__device__ __forceinline__ float norm_dist(const float4 a, const float* __restrict__ ma, const float4 b, const float* __restrict__ mb) {
  const float x = (a.x - ma[0]) - (b.x - mb[0]), y = (a.y - ma[1]) - (b.y - mb[1]), z = (a.z - ma[2]) - (b.z - mb[2]);
  return sqrtf((x * x + y * y) + z * z);
}

// cross-check + distance gate: flag[j] = 1 and partner[j] = (i, j) when i's own nearest neighbour is j and (optimizedMatching only)
// the mean-subtracted points are not further apart than thr.  One thread per j, so the compacted list is in ascending j.
static __global__ void k_mutual_gate(const unsigned long long* __restrict__ j_key, uint32_t nj, const unsigned long long* __restrict__ i_key,
                                     const float4* __restrict__ Pi, const float4* __restrict__ Pj, const float* __restrict__ mean_i, const float* __restrict__ mean_j,
                                     float thr, int gate, uint32_t* __restrict__ flag, uint2* __restrict__ partner) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nj) return;
  const unsigned long long kj = j_key[j];
  uint32_t f = 0, i = 0;
  if (kj != QN_INF_KEY) {
    i = (uint32_t)kj;
    const unsigned long long ki = i_key[i];
    f = (ki != QN_INF_KEY && (uint32_t)ki == j) ? 1u : 0u;
    if (f && gate && norm_dist(Pi[i], mean_i, Pj[j], mean_j) > thr) f = 0;
  }
  flag[j] = f; partner[j] = make_uint2(i, j);
}
static __global__ void k_compact_cand(const uint32_t* __restrict__ flag, const uint32_t* __restrict__ pos, const uint2* __restrict__ partner, uint32_t nj, uint2* __restrict__ cand) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < nj && flag[j]) cand[pos[j]] = partner[j];
}

// n steps of the LCG  s <- 1664525 s + 1013904223 (mod 2^32)  as one affine map (A, C): s_n = A s_0 + C
__device__ __forceinline__ void lcg_jump(unsigned long long n, uint32_t& A, uint32_t& C) {
  uint32_t am = 1u, ap = 0u, cm = 1664525u, cp = 1013904223u;
  while (n) { if (n & 1ull) { am *= cm; ap = ap * cm + cp; } cp = (cm + 1u) * cp; cm *= cm; n >>= 1; }
  A = am; C = ap;
}

// FGR tuple test with the seeded LCG (trial t consumes draws 3t+1 .. 3t+3 of the sequence, so every trial can be evaluated
// independently after a jump-ahead): ONE block walks the ncorr * 100 trials 1024 at a time in trial order; accepted trials mark
// their three correspondences in sel[].  optimizedMatching stops after the accepted trial that pushes 3 * accepted past the cap
// (`if (corres_tuple.size() > num_max_corres) break`), advancedMatching runs every trial.
#define QN_TUPLE_THREADS 1024
static __global__ void __launch_bounds__(QN_TUPLE_THREADS) k_tuple_test(const uint2* __restrict__ cand, const uint32_t* __restrict__ ncand_p,
                                                                        const float4* __restrict__ Pi, const float4* __restrict__ Pj, const float* __restrict__ mean_i, const float* __restrict__ mean_j,
                                                                        float scale, uint32_t seed, int capped, int max_corres, uint32_t* __restrict__ sel) {
  __shared__ uint32_t wcnt[QN_TUPLE_THREADS / 64];
  __shared__ uint32_t total_sh;
  const uint32_t ncorr = *ncand_p;
  if (ncorr < 3) return;
  const unsigned long long trials = (unsigned long long)ncorr * 100ull;
  const uint32_t a_max = capped ? (uint32_t)(max_corres / 3 + 1) : 0xffffffffu;      // accepted trials until 3 a > cap
  uint32_t A, C, As, Cs;
  lcg_jump(3ull * threadIdx.x, A, C);
  uint32_t s = A * seed + C;                                                            // state before this thread's first trial
  lcg_jump(3ull * QN_TUPLE_THREADS, As, Cs);
  uint32_t accepted = 0;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (unsigned long long base = 0; base < trials; base += QN_TUPLE_THREADS) {
    const bool live = base + threadIdx.x < trials;
    uint32_t t = s, r[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { t = t * 1664525u + 1013904223u; r[k] = (t >> 8) % ncorr; }
    bool ok = false;
    if (live) {
      const uint2 c0 = cand[r[0]], c1 = cand[r[1]], c2 = cand[r[2]];
      const float4 a0 = Pi[c0.x], a1 = Pi[c1.x], a2 = Pi[c2.x], b0 = Pj[c0.y], b1 = Pj[c1.y], b2 = Pj[c2.y];
      const float li0 = norm_dist(a0, mean_i, a1, mean_i), li1 = norm_dist(a1, mean_i, a2, mean_i), li2 = norm_dist(a2, mean_i, a0, mean_i);
      const float lj0 = norm_dist(b0, mean_j, b1, mean_j), lj1 = norm_dist(b1, mean_j, b2, mean_j), lj2 = norm_dist(b2, mean_j, b0, mean_j);
      ok = (li0 * scale < lj0) && (lj0 < li0 / scale) && (li1 * scale < lj1) && (lj1 < li1 / scale) && (li2 * scale < lj2) && (lj2 < li2 / scale);
    }
    const unsigned long long m = __ballot(ok);
    if (lane == 0) wcnt[wid] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = accepted;
    for (int w = 0; w < wid; w++) before += wcnt[w];
    if (threadIdx.x == 0) { uint32_t tot = 0; for (int w = 0; w < QN_TUPLE_THREADS / 64; w++) tot += wcnt[w]; total_sh = tot; }
    const uint32_t rank = before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));      // position of this trial among the accepted ones, trial order
    if (ok && rank < a_max) { sel[r[0]] = 1u; sel[r[1]] = 1u; sel[r[2]] = 1u; }
    __syncthreads();
    accepted += total_sh;
    if (accepted >= a_max) break;
    __syncthreads();
    s = As * s + Cs;
  }
}

// one 32-byte record per selected correspondence, written straight to pinned host memory in candidate (ascending j) order
struct QuatroCorr { uint32_t i, j; float pi[3], pj[3]; };
struct QuatroTailOut { uint32_t n_cand, n_sel, overflow, pad /* survivor-list overflow of the matrix-core search */, survivors, r0, r1, r2; };
static __global__ void k_feat_survivors(const uint32_t* __restrict__ cnt, uint32_t* __restrict__ out) { *out = *cnt; }
static __global__ void __launch_bounds__(QN_TUPLE_THREADS) k_collect_corres(const uint2* __restrict__ cand, const uint32_t* __restrict__ ncand_p, const uint32_t* __restrict__ sel,
                                                                            const float4* __restrict__ Pi, const float4* __restrict__ Pj, uint32_t cap, QuatroTailOut* __restrict__ head, QuatroCorr* __restrict__ out) {
  __shared__ uint32_t wcnt[QN_TUPLE_THREADS / 64];
  const uint32_t n = *ncand_p;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t total = 0;
  for (uint32_t base = 0; base < n; base += QN_TUPLE_THREADS) {
    const uint32_t e = base + threadIdx.x;
    const bool on = e < n && sel[e] != 0;
    const unsigned long long m = __ballot(on);
    if (lane == 0) wcnt[wid] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = total, tot = 0;
    for (int w = 0; w < QN_TUPLE_THREADS / 64; w++) { if (w < wid) before += wcnt[w]; tot += wcnt[w]; }
    const uint32_t slot = before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (on && slot < cap) {
      const uint2 c = cand[e]; const float4 a = Pi[c.x], b = Pj[c.y];
      QuatroCorr rec; rec.i = c.x; rec.j = c.y; rec.pi[0] = a.x; rec.pi[1] = a.y; rec.pi[2] = a.z; rec.pj[0] = b.x; rec.pj[1] = b.y; rec.pj[2] = b.z;
      out[slot] = rec;
    }
    total += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) { head->n_cand = n; head->n_sel = total < cap ? total : cap; head->overflow = total > cap ? 1u : 0u; }
}

// transformPcd(src, T_q): pcl::transformPointCloud with a Matrix4d on f32 points (utilities.hpp:164-175,
// loop_closure.cpp:152): double arithmetic ((t0 x + t1 y) + t2 z) + t3, rounded to f32
static __global__ void k_transform_cloud_f64(const float4* __restrict__ in, uint32_t n, const double* __restrict__ T, float4* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = in[i];
  const double x = p.x, y = p.y, z = p.z;
  out[i] = make_float4((float)(((T[0] * x + T[1] * y) + T[2] * z) + T[3]), (float)(((T[4] * x + T[5] * y) + T[6] * z) + T[7]),
                       (float)(((T[8] * x + T[9] * y) + T[10] * z) + T[11]), 1.0f);
}

}  // namespace qn
