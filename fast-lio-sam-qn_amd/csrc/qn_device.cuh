// qn_device.cuh - device-side building blocks of the gfx950 registration engine.
//
// Voxel grid-hash exact nearest-neighbour search (replaces nano_gicp's nanoflann KD-tree and the
// PCL FLANN tree behind getFitnessScore; reference call sites loop_closure.cpp:120-127).
//
// Layout (all resident in HBM, owned by the context):
//   pts[n]        float4, sorted by linear cell index (x fastest, then y, then z); .w carries the
//                 ORIGINAL point index as raw bits - ties between equal f32 distances are broken
//                 towards the lowest original index, exactly as the CPU oracle does.
//   cell_start[]  uint32 [ncells + 1], exclusive prefix of per-cell counts.
// A (y, z) row of cells [x0..x1] is therefore ONE contiguous run pts[cell_start[row + x0] ..
// cell_start[row + x1 + 1]) - what makes coalesced float4 staging of candidate points possible.
//
// Search, pass A (one query per lane): the 64 queries of a wavefront are spatially coherent
// (callers feed them in cell-sorted order).  Lanes are grouped into clusters around an anchor
// lane; the cluster's cell bounding box, grown by `margin` cells, is streamed row by row through
// a wave-private LDS tile (coalesced global float4 loads -> ds_write_b128 -> broadcast
// ds_read_b128), and every lane of the cluster scores every staged candidate.  A lane's result
// is CERTIFIED exact when its (k-th) best distance is smaller than its distance to the nearest
// face of the scanned box that still has unseen cells behind it.
// Pass B (uncertified leftovers): exact ball query, see wave_ball_nn1 / lane_ball_knn.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qn {

struct GridView {
  const float4* pts;
  const uint32_t* cell_start;
  float ox, oy, oz, cell, inv_cell, eps;
  int nx, ny, nz;
  uint32_t n;
};

#define QN_INF_KEY 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int cell_coord(float v, float o, float inv, int n) {
  return clampi((int)floorf((v - o) * inv), 0, n - 1);
}
__device__ __forceinline__ unsigned long long pack_key(float d2, uint32_t idx) {
  return ((unsigned long long)__float_as_uint(d2) << 32) | idx;
}
__device__ __forceinline__ float key_d2(unsigned long long k) { return __uint_as_float((uint32_t)(k >> 32)); }
__device__ __forceinline__ uint32_t key_idx(unsigned long long k) { return (uint32_t)k; }

// plain f32 mul/add in source order (the library is compiled -ffp-contract=off): the oracle's
// `dx*dx + dy*dy + dz*dz` bit for bit.
__device__ __forceinline__ float sqdist(float qx, float qy, float qz, float px, float py, float pz) {
  float dx = qx - px, dy = qy - py, dz = qz - pz;
  return dx * dx + dy * dy + dz * dz;
}

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = t < v ? t : v; }
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = t > v ? t : v; }
  return v;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { unsigned long long t = __shfl_xor(v, o); v = t < v ? t : v; }
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

// ------------------------------------------------------------------ result sinks
struct Best1 {                         // 1-NN
  unsigned long long key;
  __device__ __forceinline__ void init() { key = QN_INF_KEY; }
  __device__ __forceinline__ void consider(bool on, float d2, uint32_t idx) {
    unsigned long long k = pack_key(d2, idx);
    if (on && k < key) key = k;
  }
  __device__ __forceinline__ bool full() const { return key != QN_INF_KEY; }
  __device__ __forceinline__ float worst_d2() const { return key_d2(key); }
};

template <int KMAX>
struct BestK {                         // k-NN, ascending (d2, idx) list kept in registers
  unsigned long long a[KMAX];
  unsigned long long w;                // cached a[k-1] (the k-th best), refreshed after every insert
  int k;
  __device__ __forceinline__ void init(int k_) {
    k = k_; w = QN_INF_KEY;
#pragma unroll
    for (int j = 0; j < KMAX; j++) a[j] = QN_INF_KEY;
  }
  __device__ __forceinline__ void refresh() {
    unsigned long long t = a[KMAX - 1];
#pragma unroll
    for (int j = KMAX - 2; j >= 0; j--) if (j == k - 1) t = a[j];
    w = t;
  }
  __device__ __forceinline__ void insert(unsigned long long key) {   // compare-exchange chain: sorted insert, largest falls off
#pragma unroll
    for (int j = 0; j < KMAX; j++) {
      unsigned long long lo = key < a[j] ? key : a[j];
      unsigned long long hi = key < a[j] ? a[j] : key;
      a[j] = lo; key = hi;
    }
    refresh();
  }
  __device__ __forceinline__ void consider(bool on, float d2, uint32_t idx) {   // wave-cooperative: called by all lanes
    unsigned long long key = pack_key(d2, idx);
    const bool ins = on && key < w;
    if (__any(ins)) insert(ins ? key : QN_INF_KEY);
  }
  __device__ __forceinline__ bool full() const { return w != QN_INF_KEY; }
  __device__ __forceinline__ float worst_d2() const { return key_d2(w); }
};

// ------------------------------------------------------------------ pass A
// Cluster extents (cells) around the anchor lane.
#define QN_CL_DX 8
#define QN_CL_DY 2
#define QN_CL_DZ 2

// All 64 lanes must call this (inactive lanes pass active = false).  `lds` points at this wave's
// private 64-entry float4 tile.  Returns per lane whether the sink content is certified exact.
template <class Sink>
__device__ __forceinline__ bool wave_cluster_search(const GridView& g, float qx, float qy, float qz, bool active,
                                                    int margin, Sink& sink, float4* lds) {
  const int lane = threadIdx.x & 63;
  const int cx = cell_coord(qx, g.ox, g.inv_cell, g.nx);
  const int cy = cell_coord(qy, g.oy, g.inv_cell, g.ny);
  const int cz = cell_coord(qz, g.oz, g.inv_cell, g.nz);
  bool certified = false;
  unsigned long long remaining = __ballot(active);
  while (remaining) {
    const int leader = __ffsll((long long)remaining) - 1;
    const int ax = __shfl(cx, leader), ay = __shfl(cy, leader), az = __shfl(cz, leader);
    const bool in = active && ((remaining >> lane) & 1ull) && abs(cx - ax) <= QN_CL_DX && abs(cy - ay) <= QN_CL_DY && abs(cz - az) <= QN_CL_DZ;
    remaining &= ~__ballot(in);
    const int BIG = 0x3fffffff;
    int ex0 = wave_min_i(in ? cx : BIG) - margin, ex1 = wave_max_i(in ? cx : -BIG) + margin;
    int ey0 = wave_min_i(in ? cy : BIG) - margin, ey1 = wave_max_i(in ? cy : -BIG) + margin;
    int ez0 = wave_min_i(in ? cz : BIG) - margin, ez1 = wave_max_i(in ? cz : -BIG) + margin;
    ex0 = max(ex0, 0); ey0 = max(ey0, 0); ez0 = max(ez0, 0);
    ex1 = min(ex1, g.nx - 1); ey1 = min(ey1, g.ny - 1); ez1 = min(ez1, g.nz - 1);
    const int nyr = ey1 - ey0 + 1, nrows = nyr * (ez1 - ez0 + 1);
    for (int rbase = 0; rbase < nrows; rbase += 64) {
      const int r = rbase + lane;
      uint32_t s = 0, e = 0;
      if (r < nrows) {
        const int ry = ey0 + r % nyr, rz = ez0 + r / nyr;
        const uint32_t c0 = ((uint32_t)rz * g.ny + ry) * g.nx;
        s = g.cell_start[c0 + ex0]; e = g.cell_start[c0 + ex1 + 1];
      }
      unsigned long long nonempty = __ballot(e > s);
      while (nonempty) {
        const int rl = __ffsll((long long)nonempty) - 1; nonempty &= nonempty - 1;
        const uint32_t rs = __shfl(s, rl), re = __shfl(e, rl);
        for (uint32_t base = rs; base < re; base += 64) {
          const uint32_t cnt = min(64u, re - base);
          if ((uint32_t)lane < cnt) lds[lane] = g.pts[base + lane];       // coalesced 16 B/lane -> ds_write_b128
          wave_lds_fence();
          for (uint32_t c = 0; c < cnt; c++) {
            const float4 p = lds[c];                                      // broadcast ds_read_b128
            sink.consider(in, sqdist(qx, qy, qz, p.x, p.y, p.z), __float_as_uint(p.w));
          }
          wave_lds_fence();
        }
      }
    }
    if (in) {   // certification: nearest face of the scanned box that has unseen cells behind it
      const float INF = __int_as_float(0x7f800000);
      float d = INF;
      if (ex0 > 0) d = fminf(d, qx - (g.ox + ex0 * g.cell));
      if (ex1 < g.nx - 1) d = fminf(d, (g.ox + (ex1 + 1) * g.cell) - qx);
      if (ey0 > 0) d = fminf(d, qy - (g.oy + ey0 * g.cell));
      if (ey1 < g.ny - 1) d = fminf(d, (g.oy + (ey1 + 1) * g.cell) - qy);
      if (ez0 > 0) d = fminf(d, qz - (g.oz + ez0 * g.cell));
      if (ez1 < g.nz - 1) d = fminf(d, (g.oz + (ez1 + 1) * g.cell) - qz);
      if (d == INF) certified = true;                        // the whole grid was scanned
      else { d -= g.eps; certified = d > 0.f && sink.full() && sink.worst_d2() < d * d; }
    }
  }
  return certified;
}

// ------------------------------------------------------------------ pass B, 1-NN: one query per WAVE
// Exact ball query: scans every cell that intersects the ball of radius r around q, all 64 lanes
// striding over each row's contiguous run (coalesced), then a wave-wide min of the packed keys.
// r starts from a known upper bound on the NN distance (pass A's uncertified best) or, when
// nothing was found yet, from r0 and doubles until the best distance found fits inside it.
__device__ __forceinline__ unsigned long long wave_ball_nn1(const GridView& g, float qx, float qy, float qz, float r) {
  const int lane = threadIdx.x & 63;
  for (int round = 0;; round++) {
    const int bx0 = cell_coord(qx - r, g.ox, g.inv_cell, g.nx), bx1 = cell_coord(qx + r, g.ox, g.inv_cell, g.nx);
    const int by0 = cell_coord(qy - r, g.oy, g.inv_cell, g.ny), by1 = cell_coord(qy + r, g.oy, g.inv_cell, g.ny);
    const int bz0 = cell_coord(qz - r, g.oz, g.inv_cell, g.nz), bz1 = cell_coord(qz + r, g.oz, g.inv_cell, g.nz);
    const bool all = bx0 == 0 && by0 == 0 && bz0 == 0 && bx1 == g.nx - 1 && by1 == g.ny - 1 && bz1 == g.nz - 1;
    const int nyr = by1 - by0 + 1, nrows = nyr * (bz1 - bz0 + 1);
    unsigned long long best = QN_INF_KEY;
    for (int rbase = 0; rbase < nrows; rbase += 64) {
      const int rr = rbase + lane;
      uint32_t s = 0, e = 0;
      if (rr < nrows) {
        const int ry = by0 + rr % nyr, rz = bz0 + rr / nyr;
        const uint32_t c0 = ((uint32_t)rz * g.ny + ry) * g.nx;
        s = g.cell_start[c0 + bx0]; e = g.cell_start[c0 + bx1 + 1];
      }
      unsigned long long nonempty = __ballot(e > s);
      while (nonempty) {
        const int rl = __ffsll((long long)nonempty) - 1; nonempty &= nonempty - 1;
        const uint32_t rs = __shfl(s, rl), re = __shfl(e, rl);
        for (uint32_t i = rs + lane; i < re; i += 64) {
          const float4 p = g.pts[i];
          const unsigned long long k = pack_key(sqdist(qx, qy, qz, p.x, p.y, p.z), __float_as_uint(p.w));
          best = k < best ? k : best;
        }
      }
    }
    best = wave_min_u64(best);
    if (all || round > 160) return best;          // round cap: non-finite queries cannot spin forever
    if (best != QN_INF_KEY) {
      const float bd = sqrtf(key_d2(best)) * 1.000001f + g.eps;   // every point that could beat or tie `best` lies within bd
      if (bd <= r) return best;
      r = bd;                                                      // one more scan at exactly the needed radius
    } else {
      r = 2.f * r + g.cell;
    }
  }
}

// ------------------------------------------------------------------ pass B, k-NN: one query per LANE
// Same ball logic, per lane (divergent; only the k-NN leftovers of the covariance stage use it).
template <int KMAX>
__device__ __forceinline__ void lane_ball_knn(const GridView& g, float qx, float qy, float qz, float r, BestK<KMAX>& sink) {
  const int k = sink.k;
  for (int round = 0;; round++) {
    sink.init(k);
    const int bx0 = cell_coord(qx - r, g.ox, g.inv_cell, g.nx), bx1 = cell_coord(qx + r, g.ox, g.inv_cell, g.nx);
    const int by0 = cell_coord(qy - r, g.oy, g.inv_cell, g.ny), by1 = cell_coord(qy + r, g.oy, g.inv_cell, g.ny);
    const int bz0 = cell_coord(qz - r, g.oz, g.inv_cell, g.nz), bz1 = cell_coord(qz + r, g.oz, g.inv_cell, g.nz);
    const bool all = bx0 == 0 && by0 == 0 && bz0 == 0 && bx1 == g.nx - 1 && by1 == g.ny - 1 && bz1 == g.nz - 1;
    for (int rz = bz0; rz <= bz1; rz++) for (int ry = by0; ry <= by1; ry++) {
      const uint32_t c0 = ((uint32_t)rz * g.ny + ry) * g.nx;
      const uint32_t s = g.cell_start[c0 + bx0], e = g.cell_start[c0 + bx1 + 1];
      for (uint32_t i = s; i < e; i++) {
        const float4 p = g.pts[i];
        const unsigned long long key = pack_key(sqdist(qx, qy, qz, p.x, p.y, p.z), __float_as_uint(p.w));
        if (key < sink.w) sink.insert(key);                       // per-lane sorted insert
      }
    }
    if (all || round > 160) return;
    if (sink.full()) {
      const float bd = sqrtf(sink.worst_d2()) * 1.000001f + g.eps;
      if (bd <= r) return;
      r = bd;
    } else {
      r = 2.f * r + g.cell;
    }
  }
}

// ------------------------------------------------------------------ small dense f64 math
struct M3 { double m[3][3]; };

__device__ __forceinline__ M3 m3_inverse(const M3& a) {
  const double (*m)[3] = a.m;
  double c00 = m[1][1]*m[2][2] - m[1][2]*m[2][1];
  double c01 = m[1][2]*m[2][0] - m[1][0]*m[2][2];
  double c02 = m[1][0]*m[2][1] - m[1][1]*m[2][0];
  double det = m[0][0]*c00 + m[0][1]*c01 + m[0][2]*c02;
  double id = 1.0 / det;
  M3 r;
  r.m[0][0] = c00*id; r.m[1][0] = c01*id; r.m[2][0] = c02*id;
  r.m[0][1] = (m[0][2]*m[2][1] - m[0][1]*m[2][2])*id;
  r.m[1][1] = (m[0][0]*m[2][2] - m[0][2]*m[2][0])*id;
  r.m[2][1] = (m[0][1]*m[2][0] - m[0][0]*m[2][1])*id;
  r.m[0][2] = (m[0][1]*m[1][2] - m[0][2]*m[1][1])*id;
  r.m[1][2] = (m[0][2]*m[1][0] - m[0][0]*m[1][2])*id;
  r.m[2][2] = (m[0][0]*m[1][1] - m[0][1]*m[1][0])*id;
  return r;
}

// symmetric 3x3 eigen-decomposition (cyclic Jacobi, f64); eigenvalues descending, V columns.
__device__ inline void sym_eig3(const double A[6] /* xx xy xz yy yz zz */, double w[3], double V[3][3]) {
  double a[3][3] = {{A[0], A[1], A[2]}, {A[1], A[3], A[4]}, {A[2], A[4], A[5]}};
  double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 32; sweep++) {
    double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
    double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
    if (off <= 1e-300 || off <= 1e-17 * diag) break;
#pragma unroll
    for (int pq = 0; pq < 3; pq++) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
      if (a[p][q] == 0.0) continue;
      double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
      double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
      double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
      for (int k = 0; k < 3; k++) { double akp = a[k][p], akq = a[k][q]; a[k][p] = c*akp - s*akq; a[k][q] = s*akp + c*akq; }
#pragma unroll
      for (int k = 0; k < 3; k++) { double apk = a[p][k], aqk = a[q][k]; a[p][k] = c*apk - s*aqk; a[q][k] = s*apk + c*aqk; }
#pragma unroll
      for (int k = 0; k < 3; k++) { double vkp = v[k][p], vkq = v[k][q]; v[k][p] = c*vkp - s*vkq; v[k][q] = s*vkp + c*vkq; }
    }
  }
  // sort descending (3-element network), carrying columns
  double e0 = a[0][0], e1 = a[1][1], e2 = a[2][2];
  int i0 = 0, i1 = 1, i2 = 2;
  if (e0 < e1) { double t = e0; e0 = e1; e1 = t; int ti = i0; i0 = i1; i1 = ti; }
  if (e1 < e2) { double t = e1; e1 = e2; e2 = t; int ti = i1; i1 = i2; i2 = ti; }
  if (e0 < e1) { double t = e0; e0 = e1; e1 = t; int ti = i0; i0 = i1; i1 = ti; }
  w[0] = e0; w[1] = e1; w[2] = e2;
#pragma unroll
  for (int r = 0; r < 3; r++) {
    V[r][0] = i0 == 0 ? v[r][0] : (i0 == 1 ? v[r][1] : v[r][2]);
    V[r][1] = i1 == 0 ? v[r][0] : (i1 == 1 ? v[r][1] : v[r][2]);
    V[r][2] = i2 == 0 ? v[r][0] : (i2 == 1 ? v[r][1] : v[r][2]);
  }
}

}  // namespace qn
