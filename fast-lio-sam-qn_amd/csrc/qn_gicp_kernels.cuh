// qn_gicp_kernels.cuh - gfx950 kernels of the Nano-GICP path.
// Reference behaviour restated per kernel: SURVEY.md Appendix A.1; call sites
// fast_lio_sam_qn/src/loop_closure.cpp:120-133.  No MFMA anywhere: the path has no dense
// contraction - it is gather / scan / reduce work bound by memory latency and LDS bandwidth.
#pragma once
#include "qn_device.cuh"
#include "qn_knn_hist.cuh"
#include "../../include/qn_engine.h"
#include "qn_util_kernels.cuh"

namespace qn {

#define QN_NPART 28            // 21 (upper H) + 6 (b) + 1 (cost)
#define QN_ACC_MAX_BLOCKS 512  // accumulate grid = min(ceil(n / 256), this): fixed for a given n -> deterministic reduction tree
#define QN_MAX_TRACE 1024

// Device-resident optimiser state: the LM / GN controller of LsqRegistration runs
// entirely on the GPU (k_solve), so an align() needs no per-iteration host round trip.
struct GicpState {
  double x0[16], xi[16], delta[16];
  double H[36], b[6], d[6];
  double y0, yi, den;
  double lambda, nu;
  double final_H[36];
  double fitness;
  int outer, inner, phase, converged, lm_failed;   // phase: 0 = linearize at x0, 1 = error at xi, 2 = done
  uint32_t fb_count;                                // 16-query list length
  uint32_t trace_len;
  uint32_t big_count;                               // one-query-per-wave list length
  int pending;                                      // 1: the partial rows written under THIS state have not been consumed by the controller yet
  int reserved;                                     // look flags (look_decide): bit 0 = one more unseeded iteration, bit 1 = the persistent launch may go ahead
};
// The state is double buffered: generation g lives in state[g & 1] and the partial rows produced under it in partials[g & 1].  A
// controller step (k_solve, or the prologue of k_tick in every block) reads generation g and writes generation g + 1, so no block
// ever reads what another block of the same launch is writing.

struct GicpConfig {                                 // by-value kernel argument
  int k, max_iterations, optimizer, lm_max_iterations, force_iterations;
  double max_corr_dist_sq, transformation_epsilon, rotation_epsilon, lm_init_lambda_factor;
};

// XCD-aware block order.  The hardware deals consecutive workgroups round-robin to the 8 XCDs, each with its own L2; the
// queries are in tile-major cell order, so with the identity mapping every XCD touches the whole target cloud.  Remapped,
// XCD x serves one contiguous eighth of the (spatially sorted) queries and its L2 only has to hold that region of the
// target grid.  Bijection on [0, nb): logical block = first block of XCD (b & 7) + (b >> 3).
__device__ __forceinline__ uint32_t xcd_block(uint32_t b, uint32_t nb) {
  const uint32_t per = nb >> 3, rem = nb & 7u, x = b & 7u;
  return x * per + min(x, rem) + (b >> 3);
}

// ------------------------------------------------------------------ K1 grid build (utility kernels: qn_util_kernels.cuh)
struct CellCountK {
  static constexpr int TB = QN_BLOCK, OCC = 1;
  struct Args { const float4* pts; uint32_t n; GridView g; uint32_t* counts; uint32_t* cell_of_pt; };
  static __device__ __forceinline__ void run(const Args& a, const uint32_t bx, const uint32_t) {
    const GridView g = grid_resolve(a.g);
    const uint32_t i = bx * QN_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    const float4 p = a.pts[i];
    const int cx = cell_coord(p.x, g.ox, g.inv_cell, g.nx), cy = cell_coord(p.y, g.oy, g.inv_cell, g.ny), cz = cell_coord(p.z, g.oz, g.inv_cell, g.nz);
    const uint32_t c = cell_key(g, cx, cy, cz);
    a.cell_of_pt[i] = c;
    atomicAdd(&a.counts[c], 1u);
  }
};
static __global__ void __launch_bounds__(QN_BLOCK) k_cell_count(const float4* __restrict__ pts, uint32_t n, GridView g, uint32_t* __restrict__ counts, uint32_t* __restrict__ cell_of_pt) {
  const CellCountK::Args a{pts, n, g, counts, cell_of_pt};
  CellCountK::run(a, blockIdx.x, gridDim.x);
}

// The grid's numbers from the bounding box, on the device: cell edge ~4 points per ground-plane cell (the clouds are voxel-grid centroids sampled on
// surfaces, loop_closure.cpp:107), coarsened until the dense, tile-padded cell table fits max_cells; eps = the slack of cell_coord's f32 rounding.
__device__ inline GridDims grid_dims_from_bbox(const int (&omn)[3], const int (&omx)[3], bool bad, uint32_t n, uint32_t max_cells, double cell_override) {
  GridDims d;
  float mn[3], mx[3];
  for (int a = 0; a < 3; a++) { mn[a] = ord2f(omn[a]); mx[a] = ord2f(omx[a]); }
  bad = bad || !(mx[0] >= mn[0]);                                    // non-finite coordinates (or nothing finite at all)
  if (bad) { for (int a = 0; a < 3; a++) { mn[a] = 0.f; mx[a] = 0.f; } }
  double L[3]; for (int a = 0; a < 3; a++) L[a] = fmax((double)mx[a] - (double)mn[a], 0.0);
  const double Lmax = fmax(L[0], fmax(L[1], L[2]));
  const double area = fmax(L[0] * L[1], fmax(L[0] * L[2], L[1] * L[2]));
  double cell = sqrt(4.0 * area / (double)n);
  cell = fmax(cell, fmax(Lmax / 2048.0, 1e-6));
  if (cell_override > 0) cell = cell_override;
  int dims[3] = {1, 1, 1}, tdims[3] = {1, 1, 1}; const int tile[3] = {QN_TX, QN_TY, QN_TZ};
  for (int iter = 0; iter < 64; iter++) {
    double tot = 1;                                                  // the dense cell table is padded to whole 8x4x4 tiles
    for (int a = 0; a < 3; a++) { dims[a] = (int)floor(L[a] / cell) + 1; tdims[a] = (dims[a] + tile[a] - 1) / tile[a]; tot *= (double)tdims[a] * tile[a]; }
    if (tot <= (double)max_cells) break;
    cell *= fmax(cbrt(tot / (double)max_cells), 1.02);
    if (iter == 63) { cell = Lmax + 1.0; for (int a = 0; a < 3; a++) { dims[a] = 1; tdims[a] = 1; } }      // (never seen: one tile always fits - the table must not outgrow its allocation whatever the box)
  }
  d.ox = mn[0]; d.oy = mn[1]; d.oz = mn[2]; d.cell = (float)cell; d.inv_cell = 1.0f / d.cell;
  d.nx = dims[0]; d.ny = dims[1]; d.nz = dims[2]; d.ntx = tdims[0]; d.nty = tdims[1]; d.ntz = tdims[2];
  float amax = 0; for (int a = 0; a < 3; a++) amax = fmaxf(amax, fmaxf(fabsf(mn[a]), fabsf(mx[a])));
  d.eps = 1e-3f * d.cell + 1e-6f * (amax + (float)Lmax);
  d.n = bad ? 0u : n; d.ncells = (uint32_t)tdims[0] * tdims[1] * tdims[2] * QN_TILE_CELLS; d.nonfinite = bad ? 1u : 0u; d.pad = 0;
  return d;
}

// Grid build, first kernel: pack the caller's strided points into float4 (x, y, z, 1), fold their bounding box into the accumulator (ordered-int atomics, one set
// per block), and let the LAST block to arrive (ticket) turn the box into the grid's numbers - written to device memory for the kernels that follow and straight
// into the pinned mirror the host reads after its next synchronisation.  That block also puts the accumulator back into its initial state: no upload, no reset
// kernel, no read-back in front of or behind the build.  (Accumulator words are only ever touched with agent-scope atomics: the blocks sit on different XCDs.)
#define QN_BBOX_MAX_BLOCKS 256
struct BBoxAcc { int mn[3], mx[3]; uint32_t nonfinite, ticket; };
struct PackBBoxK {
  static constexpr int TB = QN_BLOCK, OCC = 1;
  struct Args { const char* in; uint32_t stride, n; float4* raw; BBoxAcc* acc; uint32_t max_cells; double cell_override; GridDims* dims_dev; GridDims* dims_host; };
  static __device__ __forceinline__ void run(const Args& a, const uint32_t bx, const uint32_t nbx) {
    __shared__ int smn[QN_BLOCK / 64][3], smx[QN_BLOCK / 64][3], sbad[QN_BLOCK / 64];
    const char* __restrict__ in = a.in; float4* __restrict__ raw = a.raw; BBoxAcc* acc = a.acc;
    const uint32_t n = a.n, stride = a.stride;
    int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    int bad = 0;
    for (uint32_t i = bx * QN_BLOCK + threadIdx.x; i < n; i += nbx * QN_BLOCK) {
      const float* q = (const float*)(in + (size_t)i * stride);
      const float4 p = make_float4(q[0], q[1], q[2], 1.0f);
      raw[i] = p;
      if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) { bad = 1; continue; }
      const int ox = f2ord(p.x), oy = f2ord(p.y), oz = f2ord(p.z);
      mn[0] = min(mn[0], ox); mn[1] = min(mn[1], oy); mn[2] = min(mn[2], oz);
      mx[0] = max(mx[0], ox); mx[1] = max(mx[1], oy); mx[2] = max(mx[2], oz);
    }
#pragma unroll
    for (int d = 0; d < 3; d++) { mn[d] = wave_min_i(mn[d]); mx[d] = wave_max_i(mx[d]); }
    bad = wave_max_i(bad);
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { for (int d = 0; d < 3; d++) { smn[wid][d] = mn[d]; smx[wid][d] = mx[d]; } sbad[wid] = bad; }
    __syncthreads();
    __shared__ int s_last;
    if (threadIdx.x == 0) {
      for (int w = 1; w < QN_BLOCK / 64; w++) { for (int d = 0; d < 3; d++) { mn[d] = min(mn[d], smn[w][d]); mx[d] = max(mx[d], smx[w][d]); } bad |= sbad[w]; }
      // the block's box goes to a slot of its own (six contended atomics per block on the same words cost more than the whole reduction: 64 blocks, 10 us);
      // the one contended operation left is the ticket
      BBoxAcc* mine = acc + 1 + bx;
      for (int d = 0; d < 3; d++) { __hip_atomic_store(&mine->mn[d], mn[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(&mine->mx[d], mx[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      __hip_atomic_store(&mine->nonfinite, (uint32_t)bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the slot's write-through stores are acknowledged before the ticket is drawn (no L2-wide fence: 4096 of them per batched launch were most of its time)
      s_last = __hip_atomic_fetch_add(&acc->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nbx - 1u ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    // the last block: its first wave folds the slots (one per lane, QN_BBOX_MAX_BLOCKS <= 256: four rounds at most), lane 0 derives the numbers
    if (threadIdx.x >= 64) return;
    for (int d = 0; d < 3; d++) { mn[d] = 0x7fffffff; mx[d] = (int)0x80000000; }      // (the slots are read with agent-scope loads below)
    int nf = 0;
    for (uint32_t b = threadIdx.x; b < nbx; b += 64) {
      const BBoxAcc* o = acc + 1 + b;
      for (int d = 0; d < 3; d++) { mn[d] = min(mn[d], __hip_atomic_load(&o->mn[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); mx[d] = max(mx[d], __hip_atomic_load(&o->mx[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
      nf |= (int)__hip_atomic_load(&o->nonfinite, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int d = 0; d < 3; d++) { mn[d] = wave_min_i(mn[d]); mx[d] = wave_max_i(mx[d]); }
    nf = wave_max_i(nf);
    if (threadIdx.x != 0) return;
    const GridDims g = grid_dims_from_bbox(mn, mx, nf != 0, n, a.max_cells, a.cell_override);
    *a.dims_dev = g; *a.dims_host = g;
    __hip_atomic_store(&acc->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
};
static __global__ void __launch_bounds__(QN_BLOCK) k_pack_bbox_dims(const char* __restrict__ in, uint32_t stride, uint32_t n, float4* __restrict__ raw, BBoxAcc* acc,
                                                                    uint32_t max_cells, double cell_override, GridDims* __restrict__ dims_dev, GridDims* __restrict__ dims_host) {
  const PackBBoxK::Args a{in, stride, n, raw, acc, max_cells, cell_override, dims_dev, dims_host};
  PackBBoxK::run(a, blockIdx.x, gridDim.x);
}
static __global__ void k_bbox_acc_init(BBoxAcc* acc, int count) {
  for (int i = threadIdx.x; blockIdx.x == 0 && i < count; i += blockDim.x) { BBoxAcc a; for (int d = 0; d < 3; d++) { a.mn[d] = 0x7fffffff; a.mx[d] = (int)0x80000000; } a.nonfinite = 0; a.ticket = 0; acc[i] = a; }
}

// Exclusive scan of the cell counts in ONE launch (decoupled look-back): tile t = block t publishes its aggregate, sums its predecessors' aggregates back to the
// nearest published inclusive prefix, publishes its own.  Status words are {epoch : 30 | state : 2 | value : 32} so that nothing has to be reset between builds:
// a word of another epoch reads "nothing yet".  The launch uses the grid of the LARGEST table (the host does not know this cloud's); blocks past the table leave
// at once.  All blocks are co-resident (<= 2048 tiles of 256 threads on 256 CUs), so a tile waiting on its predecessors cannot starve them.  out[m] = total.
#define QN_LB_STATE_AGG 1ull
#define QN_LB_STATE_INC 2ull
struct ScanLookbackK {
  static constexpr int TB = QN_BLOCK, OCC = 1;
  struct Args { const uint32_t* in; const GridDims* dims; uint32_t* out; unsigned long long* status; uint32_t epoch, total; };
  static __device__ __forceinline__ void run(const Args& a, const uint32_t bx, const uint32_t) {
    __shared__ uint32_t wsum[QN_BLOCK / 64];
    __shared__ uint32_t s_prefix;
    const uint32_t* __restrict__ in = a.in; uint32_t* __restrict__ out = a.out; unsigned long long* status = a.status;
    const uint32_t total = a.total;
    const uint32_t m = uni(a.dims->ncells);
    const uint32_t tile = bx;
    if (tile * (uint32_t)(QN_BLOCK * QN_SCAN_ITEMS) >= m) return;
    const uint32_t base = (tile * QN_BLOCK + threadIdx.x) * QN_SCAN_ITEMS;
    uint32_t v[QN_SCAN_ITEMS], s = 0;
    if (base + QN_SCAN_ITEMS <= m) {
      const uint4* q = (const uint4*)(in + base);
#pragma unroll
      for (int j = 0; j < QN_SCAN_ITEMS / 4; j++) { const uint4 x = q[j]; v[4 * j] = x.x; v[4 * j + 1] = x.y; v[4 * j + 2] = x.z; v[4 * j + 3] = x.w; }
    } else {
#pragma unroll
      for (int j = 0; j < QN_SCAN_ITEMS; j++) v[j] = (base + j < m) ? in[base + j] : 0u;
    }
#pragma unroll
    for (int j = 0; j < QN_SCAN_ITEMS; j++) s += v[j];
    uint32_t inc = s;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint32_t woff = 0, agg = 0;
    for (int w = 0; w < QN_BLOCK / 64; w++) { if (w < wid) woff += wsum[w]; agg += wsum[w]; }
    const unsigned long long tag = (unsigned long long)(a.epoch & 0x3fffffffu) << 34;
    if (wid == 0) {
      uint32_t prefix = 0;
      if (tile == 0) {
        if (lane == 0) __hip_atomic_store(status, tag | (QN_LB_STATE_INC << 32) | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        if (lane == 0) __hip_atomic_store(status + tile, tag | (QN_LB_STATE_AGG << 32) | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int look = (int)tile - 1;                                      // lane l watches tile look - l
        for (;;) {
          const int mine = look - lane;
          unsigned long long x = 0;
          if (mine >= 0) { do { x = __hip_atomic_load(status + mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((x >> 34) != (tag >> 34)); }
          const bool is_inc = mine >= 0 && ((x >> 32) & 3ull) == QN_LB_STATE_INC;
          const unsigned long long incs = __ballot(is_inc);
          const int first = incs ? __ffsll((long long)incs) - 1 : 64;  // nearest tile with an inclusive prefix, as a lane number
          uint32_t part = (mine >= 0 && lane <= first) ? (uint32_t)x : 0u;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
          prefix += part;
          if (incs || look - 64 < 0) break;
          look -= 64;
        }
        if (lane == 0) __hip_atomic_store(status + tile, tag | (QN_LB_STATE_INC << 32) | (prefix + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (lane == 0) s_prefix = prefix;
    }
    __syncthreads();
    uint32_t run = s_prefix + woff + inc - s;
    if (base + QN_SCAN_ITEMS <= m) {
      uint4* q = (uint4*)(out + base);
#pragma unroll
      for (int j = 0; j < QN_SCAN_ITEMS / 4; j++) { uint4 x; x.x = run; run += v[4 * j]; x.y = run; run += v[4 * j + 1]; x.z = run; run += v[4 * j + 2]; x.w = run; run += v[4 * j + 3]; q[j] = x; }
    } else {
#pragma unroll
      for (int j = 0; j < QN_SCAN_ITEMS; j++) { if (base + j < m) out[base + j] = run; run += v[j]; }
    }
    if (base <= m && m < base + QN_SCAN_ITEMS) out[m] = total;       // (m is a multiple of the 128-cell tile: this is the thread whose first item would be index m ...)
    if (m == (tile + 1u) * (uint32_t)(QN_BLOCK * QN_SCAN_ITEMS) && threadIdx.x == QN_BLOCK - 1) out[m] = total;      // (... or the table ends with this tile)
  }
};
static __global__ void __launch_bounds__(QN_BLOCK) k_scan_lookback(const uint32_t* __restrict__ in, const GridDims* __restrict__ dims, uint32_t* __restrict__ out,
                                                                   unsigned long long* status, uint32_t epoch, uint32_t total) {
  const ScanLookbackK::Args a{in, dims, out, status, epoch, total};
  ScanLookbackK::run(a, blockIdx.x, gridDim.x);
}

struct ScatterK {
  static constexpr int TB = QN_BLOCK, OCC = 1;
  struct Args { const float4* pts; uint32_t n; const uint32_t* cell_of_pt; const uint32_t* cell_start; uint32_t* counts; float4* sorted; };
  static __device__ __forceinline__ void run(const Args& a, const uint32_t bx, const uint32_t) {
    const uint32_t i = bx * QN_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    const uint32_t c = a.cell_of_pt[i];
    const uint32_t slot = a.cell_start[c] + atomicSub(&a.counts[c], 1u) - 1u;
    float4 p = a.pts[i];
    p.w = __uint_as_float(i);
    a.sorted[slot] = p;
  }
};
static __global__ void __launch_bounds__(QN_BLOCK) k_scatter(const float4* __restrict__ pts, uint32_t n, const uint32_t* __restrict__ cell_of_pt,
                          const uint32_t* __restrict__ cell_start, uint32_t* __restrict__ counts, float4* __restrict__ sorted) {
  const ScatterK::Args a{pts, n, cell_of_pt, cell_start, counts, sorted};
  ScatterK::run(a, blockIdx.x, gridDim.x);
}

// The scatter's atomics hand out the slots of a cell in arrival order: the order INSIDE a cell differs from run to run.  Searches do not care
// (ties resolve on the original index), but the optimiser ticks sum the correspondences in sorted order, block by block and lane by lane, and
// the FPFH kernels sum neighbours in walk order: with an arbitrary order inside the cells those f64 sums - and so H, b and the pose - differed
// in the last bits between two runs of the same registration.  This pass puts every cell's points in ascending original index (rank = number of
// smaller indices in the cell: ~4 points per cell, one thread per point).  Cells above 256 points (degenerate clouds) keep the arrival order.
struct StableCellsK {
  static constexpr int TB = QN_BLOCK, OCC = 1;
  struct Args { const float4* in; uint32_t n; const uint32_t* cell_of_pt; const uint32_t* cell_start; float4* out; };
  static __device__ __forceinline__ void run(const Args& a, const uint32_t bx, const uint32_t) {
    const float4* __restrict__ in = a.in; float4* __restrict__ out = a.out;
    const uint32_t t = bx * QN_BLOCK + threadIdx.x;
    if (t >= a.n) return;
    const float4 p = in[t];
    const uint32_t i = __float_as_uint(p.w), c = a.cell_of_pt[i];
    const uint32_t s = a.cell_start[c], e = a.cell_start[c + 1];
    if (e - s > 256u) { out[t] = p; return; }
    uint32_t rank = 0;
    for (uint32_t u = s; u < e; u++) rank += __float_as_uint(in[u].w) < i ? 1u : 0u;
    out[s + rank] = p;
  }
};
static __global__ void __launch_bounds__(QN_BLOCK) k_stable_cells(const float4* __restrict__ in, uint32_t n, const uint32_t* __restrict__ cell_of_pt, const uint32_t* __restrict__ cell_start,
                                      float4* __restrict__ out) {
  const StableCellsK::Args a{in, n, cell_of_pt, cell_start, out};
  StableCellsK::run(a, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------ K2+K3 k-NN + covariance (k-NN selection: k_knn_hist below, k_knn_cov in qn_knn_kernels.cuh)
// k nearest (self included), mean/cov in f64 (cov = X X^T / k), PLANE regularisation:
// C = V diag(1, 1, 1e-3) V^T with V the eigenvectors of cov (eigenvalues descending); from stored neighbour indices (ascending (d2, idx) order, -1 = missing): one point per lane.
// Target-side record for the fused optimiser ticks: point and covariance of a target point in ONE 64-byte line
// (a correspondence then costs one scattered cache line instead of two).
// PLANE covariances are C = I - 0.999 n n^T: the record carries the normal (24 bytes) instead of six covariance entries.
struct __attribute__((aligned(64))) TargetRec { float4 p; double n[3]; double pad[3]; };
static_assert(sizeof(TargetRec) == 64, "TargetRec is one 64-byte line");
// Threads walk the points in cell-sorted order (a block's points are spatial neighbours, so their k-NN gathers overlap in
// L1/L2), blocks in XCD-aware order.
#define QN_COV_BATCH 10
#define QN_COV_REG 20
struct CovFromIdxK {
  static constexpr int TB = QN_BLOCK, OCC = 1;
  struct Args { const float4* raw; const float4* sorted; uint32_t n; int k; const int32_t* knn_idx; double* nrm; double* nrm_sorted; TargetRec* rec; uint32_t* list_counts; };
  static __device__ __forceinline__ void run(const Args& a, const uint32_t bx, const uint32_t nbx) {
    const float4* __restrict__ raw = a.raw; const int32_t* __restrict__ knn_idx = a.knn_idx;
    double* __restrict__ nrm = a.nrm; double* __restrict__ nrm_sorted = a.nrm_sorted; TargetRec* __restrict__ rec = a.rec;
    const int k = a.k;
    if (bx == 0 && threadIdx.x < 4) a.list_counts[threadIdx.x] = 0u;      // the selection passes' list counters go back to zero behind their last reader (no memset in front of the next cloud)
    const uint32_t spos = xcd_block(bx, nbx) * QN_BLOCK + threadIdx.x;
    if (spos >= a.n) return;
    const uint32_t i = __float_as_uint(a.sorted[spos].w);
    const int32_t* nb = knn_idx + (size_t)i * k;
    int found = 0;
    double mean[3] = {0, 0, 0};
    double nv[3] = {0, 0, 0};                  // found == 0 cannot happen for a finite point (it is its own neighbour); a zero normal reads as C = I
    // the layouts of the optimiser ticks are written here as well (nrm_sorted: source, cell-sorted order; rec: target, 64-byte records)
    auto store = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < 3; u++) nrm[(size_t)i * 3 + u] = nv[u];
      if (nrm_sorted) {
#pragma unroll
        for (int u = 0; u < 3; u++) nrm_sorted[(size_t)spos * 3 + u] = nv[u];
      }
      if (rec) { TargetRec r; r.p = raw[i];
#pragma unroll
        for (int u = 0; u < 3; u++) { r.n[u] = nv[u]; r.pad[u] = 0; }
        rec[i] = r; }
    };
    double c[6] = {0, 0, 0, 0, 0, 0};
    if (k <= QN_COV_REG) {
      // k <= 20 (the reference's 15, the bench's 20): every neighbour is gathered ONCE and kept in registers for both sweeps (the kernel is bound by its scattered
      // 64-byte line fetches - 2 x k per point - not by arithmetic: one gather pass instead of two); sums in neighbour order, as below
      int32_t u[QN_COV_REG]; float4 q[QN_COV_REG];
      static_assert(QN_COV_REG % 4 == 0, "the index row is fetched four entries at a time");
      if ((k & 3) == 0) {                                                // a row of k indices starts on a 16-byte boundary then: five 16-byte loads instead of twenty 4-byte ones
        const int4* __restrict__ nb4 = (const int4*)nb;
#pragma unroll
        for (int e = 0; e < QN_COV_REG; e += 4) {
          const int4 v = e < k ? nb4[e >> 2] : make_int4(-1, -1, -1, -1);
          u[e] = v.x; u[e + 1] = v.y; u[e + 2] = v.z; u[e + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int e = 0; e < QN_COV_REG; e++) u[e] = e < k ? nb[e] : -1;
      }
#pragma unroll
      for (int e = 0; e < QN_COV_REG; e++) q[e] = raw[u[e] < 0 ? 0 : u[e]];
#pragma unroll
      for (int e = 0; e < QN_COV_REG; e++) if (u[e] >= 0) { mean[0] += (double)q[e].x; mean[1] += (double)q[e].y; mean[2] += (double)q[e].z; found++; }
      if (found == 0) { store(); return; }
      mean[0] /= found; mean[1] /= found; mean[2] /= found;
#pragma unroll
      for (int e = 0; e < QN_COV_REG; e++) if (u[e] >= 0) {
        const double dx = (double)q[e].x - mean[0], dy = (double)q[e].y - mean[1], dz = (double)q[e].z - mean[2];
        c[0] += dx * dx; c[1] += dx * dy; c[2] += dx * dz; c[3] += dy * dy; c[4] += dy * dz; c[5] += dz * dz;
      }
    } else {
    // neighbours QN_COV_BATCH (10) at a time: the index loads, then the point gathers are issued together (a loop of dependent idx -> point
    // round trips was the whole cost of this kernel: 4 per batch = 10 round trips at k = 20, 10 per batch = 4); the sums are still formed in neighbour order
    for (int j = 0; j < k; j += QN_COV_BATCH) {
      int32_t u[QN_COV_BATCH]; float4 q[QN_COV_BATCH];
#pragma unroll
      for (int e = 0; e < QN_COV_BATCH; e++) u[e] = j + e < k ? nb[j + e] : -1;
#pragma unroll
      for (int e = 0; e < QN_COV_BATCH; e++) q[e] = raw[u[e] < 0 ? 0 : u[e]];
#pragma unroll
      for (int e = 0; e < QN_COV_BATCH; e++) if (u[e] >= 0) { mean[0] += (double)q[e].x; mean[1] += (double)q[e].y; mean[2] += (double)q[e].z; found++; }
    }
    if (found == 0) { store(); return; }
    mean[0] /= found; mean[1] /= found; mean[2] /= found;
    for (int j = 0; j < k; j += QN_COV_BATCH) {
      int32_t u[QN_COV_BATCH]; float4 q[QN_COV_BATCH];
#pragma unroll
      for (int e = 0; e < QN_COV_BATCH; e++) u[e] = j + e < k ? nb[j + e] : -1;
#pragma unroll
      for (int e = 0; e < QN_COV_BATCH; e++) q[e] = raw[u[e] < 0 ? 0 : u[e]];
#pragma unroll
      for (int e = 0; e < QN_COV_BATCH; e++) if (u[e] >= 0) {
        const double dx = (double)q[e].x - mean[0], dy = (double)q[e].y - mean[1], dz = (double)q[e].z - mean[2];
        c[0] += dx * dx; c[1] += dx * dy; c[2] += dx * dz; c[3] += dy * dy; c[4] += dy * dz; c[5] += dz * dz;
      }
    }
    }
#pragma unroll
    for (int t = 0; t < 6; t++) c[t] /= found;
    double w[3], V[3][3];
    sym_eig3(c, w, V);
    // C = V diag(1, 1, 1e-3) V^T = I - 0.999 v3 v3^T  (V orthonormal): keep v3, the eigenvector of the smallest eigenvalue
    nv[0] = V[0][2]; nv[1] = V[1][2]; nv[2] = V[2][2];
    store();
  }
};
static __global__ void __launch_bounds__(QN_BLOCK) k_cov_from_idx(const float4* __restrict__ raw, const float4* __restrict__ sorted, uint32_t n, int k, const int32_t* __restrict__ knn_idx, double* __restrict__ nrm,
                                                                  double* __restrict__ nrm_sorted, TargetRec* __restrict__ rec, uint32_t* __restrict__ list_counts) {
  const CovFromIdxK::Args a{raw, sorted, n, k, knn_idx, nrm, nrm_sorted, rec, list_counts};
  CovFromIdxK::run(a, blockIdx.x, gridDim.x);
}

// parity read-back: the 3x3 covariances the reference stores (A.1.3), rebuilt from the normals: xx xy xz yy yz zz
static __global__ void k_cov_from_normals(const double* __restrict__ nrm, uint32_t n, double* __restrict__ cov6) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double a = nrm[(size_t)i * 3], b = nrm[(size_t)i * 3 + 1], c = nrm[(size_t)i * 3 + 2];
  double* o = cov6 + (size_t)i * 6;
  o[0] = 1.0 - 0.999 * a * a; o[1] = -0.999 * a * b; o[2] = -0.999 * a * c; o[3] = 1.0 - 0.999 * b * b; o[4] = -0.999 * b * c; o[5] = 1.0 - 0.999 * c * c;
}


// k-NN by histogram selection (wave_knn_hist), 16 queries per wave.  LIST = false: every point, radius margin * cell,
// `max_rounds` rounds; near leftovers are appended to fb_list as (t, r), far ones (next radius > 2.5 r0) and list overflows to
// gen_list.  LIST = true: the fb_list entries, rounds until exact; what it cannot finish is passed on to gen_list as well
// (served by k_knn_single, then k_knn_cov<KMAX, true, 4>).
// HCAP = capacity of the per-query candidate list of pass 2 (>= k + the few extras below tau): 32 for k <= 24 keeps the
// kernel at 4 waves/SIMD (LDS 34 KB/block, <= 128 VGPRs), 48 serves k <= 32 at 3 waves/SIMD.
// QN_KNN_BLOCK: threads per block of the selection kernel (one wave serves 16 queries; smaller blocks refill the CUs at a
// finer grain, which shortens the tail of the 6250-wave launch)
#ifndef QN_KNN_BLOCK
#define QN_KNN_BLOCK 64
#endif
struct KnnHistArgs { GridView g; int k; float r0; int max_rounds; int32_t* knn_idx; float* knn_d2; uint2* fb_list; uint32_t* fb_count; uint2* gen_list; uint32_t* gen_count; };
__device__ __forceinline__ float late_s(float v) { asm volatile("" : "+s"(v)); return v; }      // (a scalar made opaque here: products with it are formed where they are used, not kept in a vector register across a search)
template <bool LIST, int HCAP, bool MM = true>      // MM: distances of the two selection passes on the matrix cores (qn_knn_hist.cuh); false = the VALU scoring (knob knn_mm 0, A/B and verification)
struct KnnHistK {
  static constexpr int TB = QN_KNN_BLOCK, OCC = HCAP <= 32 ? 4 : 3;
  using Args = KnnHistArgs;
  static __device__ __forceinline__ void run(const Args& a, const uint32_t bx, const uint32_t nbx) {
    __shared__ WaveLdsH<HCAP> lds[QN_KNN_BLOCK / 64];
    const GridView g = grid_resolve(a.g);
    const int k = a.k, max_rounds = a.max_rounds;
    int32_t* __restrict__ knn_idx = a.knn_idx; float* __restrict__ knn_d2 = a.knn_d2;
    uint2* __restrict__ fb_list = a.fb_list; uint32_t* __restrict__ fb_count = a.fb_count; uint2* __restrict__ gen_list = a.gen_list; uint32_t* __restrict__ gen_count = a.gen_count;
    float r0 = a.r0; if (r0 < 0.f) r0 = -r0 * g.cell;              // (a negative radius is in cells: the host does not know the cell edge)
    r0 = uni(r0);                                                   // (a scalar: as a vector register it was carried - spilled - across every group's search)
    WaveLdsH<HCAP>* my = &lds[threadIdx.x >> 6];
    const uint32_t nq = LIST ? uni(*fb_count) : g.n;
    if (LIST && g.dbg && bx == 0 && threadIdx.x == 0) atomicAdd(&g.dbg[4], nq);
    const uint32_t wave0 = (LIST ? bx : xcd_block(bx, nbx)) * (QN_KNN_BLOCK / 64) + (threadIdx.x >> 6), nwaves = nbx * (QN_KNN_BLOCK / 64);
    for (uint32_t base = wave0 * 16; base < nq; base += nwaves * 16) {
      const uint32_t slot = base + (threadIdx.x & 15);
      bool active = slot < nq;
      uint32_t t = slot; float r = r0; bool general = false;
      if (LIST && active) { const uint2 rec = fb_list[slot]; t = rec.x; r = __uint_as_float(rec.y & 0x7fffffffu); general = (rec.y >> 31) != 0; }
      const float4 q = active ? g.pts[t] : make_float4(0, 0, 0, 0);
      const uint32_t i = __float_as_uint(q.w);
      int status = 2;
      {
        const int st = wave_knn_hist<HCAP, MM>(g, q.x, q.y, q.z, active && !general, r, k, max_rounds < 0 ? -max_rounds : max_rounds, my, knn_idx, knn_d2, i);
        if (!general) status = st;
      }
      if (!active || (threadIdx.x & 63) >= 16 || status == 0) continue;
      if (LIST) { const uint32_t fs = atomicAdd(gen_count, 1u); gen_list[fs] = make_uint2(t, __float_as_uint(r)); }
      else if (status == 2 || r > 2.5f * late_s(r0) || max_rounds < 0 /* every leftover: launch_knn_cov */) { const uint32_t fs = atomicAdd(gen_count, 1u); gen_list[fs] = make_uint2(t, __float_as_uint(r)); }   // far or overflowing: one query per wave (k_knn_single)
      else { const uint32_t fs = atomicAdd(fb_count, 1u); fb_list[fs] = make_uint2(t, __float_as_uint(r)); }
    }
  }
};
template <bool LIST, int HCAP, bool MM = true>
__global__ void __launch_bounds__(QN_KNN_BLOCK, HCAP <= 32 ? 4 : 3) k_knn_hist(GridView g, int k, float r0, int max_rounds, int32_t* __restrict__ knn_idx, float* __restrict__ knn_d2,
                                                       uint2* __restrict__ fb_list, uint32_t* __restrict__ fb_count, uint2* __restrict__ gen_list, uint32_t* __restrict__ gen_count) {
  const KnnHistArgs a{g, k, r0, max_rounds, knn_idx, knn_d2, fb_list, fb_count, gen_list, gen_count};
  KnnHistK<LIST, HCAP, MM>::run(a, blockIdx.x, gridDim.x);
}

// The far / overflowing k-NN queries, one per wave (wave_knn_single); what it cannot finish goes to the sorted-list kernel.
struct KnnSingleK {
  static constexpr int TB = QN_BLOCK, OCC = 1;
  struct Args { GridView g; int k; int32_t* knn_idx; float* knn_d2; const uint2* list; const uint32_t* count; uint2* gen_list; uint32_t* gen_count; };
  static __device__ __forceinline__ void run(const Args& a, const uint32_t bx, const uint32_t nbx) {
    __shared__ WaveLdsH1 lds[QN_BLOCK / 64];
    const GridView g = grid_resolve(a.g);
    const int k = a.k;
    int32_t* __restrict__ knn_idx = a.knn_idx; float* __restrict__ knn_d2 = a.knn_d2; const uint2* __restrict__ list = a.list; uint2* __restrict__ gen_list = a.gen_list; uint32_t* __restrict__ gen_count = a.gen_count;
    WaveLdsH1* my = &lds[threadIdx.x >> 6];
    const uint32_t nq = uni(*a.count);
    if (g.dbg && bx == 0 && threadIdx.x == 0) atomicAdd(&g.dbg[8], nq);
    const uint32_t wave0 = bx * (QN_BLOCK / 64) + (threadIdx.x >> 6), nwaves = nbx * (QN_BLOCK / 64);
    for (uint32_t e = wave0; e < nq; e += nwaves) {
      const uint2 rec = list[e];
      const float4 q = g.pts[rec.x];
      const uint32_t i = __float_as_uint(q.w);
      const int st = wave_knn_single(g, q.x, q.y, q.z, __uint_as_float(rec.y & 0x7fffffffu), k, my, knn_idx + (size_t)i * k, knn_d2 ? knn_d2 + (size_t)i * k : nullptr);
      if (st != 0 && (threadIdx.x & 63) == 0) { if (g.dbg) atomicAdd(&g.dbg[9], 1u); const uint32_t fs = atomicAdd(gen_count, 1u); gen_list[fs] = make_uint2(rec.x, rec.y & 0x7fffffffu); }
    }
  }
};
static __global__ void __launch_bounds__(QN_BLOCK) k_knn_single(GridView g, int k, int32_t* __restrict__ knn_idx, float* __restrict__ knn_d2,
                                                                const uint2* __restrict__ list, const uint32_t* __restrict__ count, uint2* __restrict__ gen_list, uint32_t* __restrict__ gen_count) {
  const KnnSingleK::Args a{g, k, knn_idx, knn_d2, list, count, gen_list, gen_count};
  KnnSingleK::run(a, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------ K4a nearest-neighbour search
// MODE 0: update_correspondences: q = T_f * p in f32, Eigen order ((c0 x + c1 y) + c2 z) + c3.
// MODE 1: getFitnessScore: q = pcl::transformPointCloud, SSE order c0 x + (c1 y + (c2 z + c3)).
template <int MODE>
__device__ __forceinline__ void xform_query(const float Tf[12], float x, float y, float z, float& qx, float& qy, float& qz) {
  if (MODE == 0) {
    qx = ((Tf[0] * x + Tf[1] * y) + Tf[2] * z) + Tf[3];
    qy = ((Tf[4] * x + Tf[5] * y) + Tf[6] * z) + Tf[7];
    qz = ((Tf[8] * x + Tf[9] * y) + Tf[10] * z) + Tf[11];
  } else {
    qx = Tf[0] * x + (Tf[1] * y + (Tf[2] * z + Tf[3]));
    qy = Tf[4] * x + (Tf[5] * y + (Tf[6] * z + Tf[7]));
    qz = Tf[8] * x + (Tf[9] * y + (Tf[10] * z + Tf[11]));
  }
}

// corr / sqd are indexed by the ORIGINAL source index i (API order); the tracking state (nn_idx, nn_ref) by the source's
// cell-sorted position t, the order every NN kernel walks the queries in - coalesced.
// The indices a search stores under are known before it starts; the compiler then forms the five 64-bit store addresses up front and carries them - ten VGPRs - across the
// whole search (they were what the list kernels spilled).  late(): the value is opaque until this point, so the addresses are formed where they are used.
__device__ __forceinline__ uint32_t late(uint32_t v) { asm volatile("" : "+v"(v)); return v; }
template <int MODE>
__device__ __forceinline__ void store_nn(unsigned long long key, uint32_t i, uint32_t t, double thr2, int32_t* __restrict__ corr, float* __restrict__ sqd, int32_t* __restrict__ nn_idx) {
  const float d2 = key_d2(key);
  if (MODE == 0) nn_idx[t] = key != QN_INF_KEY ? (int32_t)key_idx(key) : -1;     // ungated NN: next iteration's search seed (k_nn_track); -1 = none (non-finite query): never a stale index
  if (MODE == 0) {
    const bool found = key != QN_INF_KEY;
    sqd[i] = found ? d2 : 0.f;
    corr[i] = (found && (double)d2 < thr2) ? (int32_t)key_idx(key) : -1;
  } else {
    sqd[i] = key != QN_INF_KEY ? d2 : __int_as_float(0x7f800000);
  }
}

struct NnOpt { float4* clear_ref; int cond; int group, group_min; unsigned long long* probe; uint32_t fb_small; };       // fb_small: a 16-per-wave list of at most this many entries is served one entry per wave like the far list (0: never); probe: developer timing of the one-per-wave entries (knob list_probe): [4 w] = slowest entry << 32 | entries, [4 w + 1] = busy time of wave w (100 MHz ticks), [4 w + 2] = rounds << 48 | segments << 24 | candidates and [4 w + 3] = first radius | neighbour distance (f32 bits) of that slowest entry; group: far lists of at least group_min entries are served ceil(length / group) consecutive entries per wave, neighbours sharing a scan (0: one per wave); clear_ref: the first search of an align also resets the far-candidate references (one per query); cond: run only if this look flag is set
// LIST = false: first search of an align (no seed): every source point, radius margin * cell, two rounds,
// leftovers to fb_list.  LIST = true: the fb_list entries (leftovers of the first search, or the big-ball
// queries of k_nn_track with their seed radius), 16 per wave, rounds until exact.
// In the LIST launch the last `big_blocks` blocks serve big_list instead, one query per wave (wave_search_single).
// BLOCK = threads per block: the grid passes run one wave per block (finer refill of the CUs), the list passes four.
struct NnSearchArgs { GridView src, tgt; const GicpState* st; double thr2; float r0; int max_rounds; int32_t* corr; float* sqd; int32_t* nn_idx; float4* nn_ref;
                      uint2* fb_list; uint32_t* fb_count; uint2* big_list; uint32_t* big_count; int big_blocks; float big_ratio; uint32_t* far_stats; NnOpt opt; };
template <int MODE, bool LIST, int BLOCK, bool GROUP = false>      // GROUP: the far list may be served in groups of neighbours (its own instantiation: the grouped search costs registers)
struct NnSearchK {
  static constexpr int TB = BLOCK, OCC = GROUP ? 5 : 6;
  using Args = NnSearchArgs;
  static __device__ __forceinline__ void run(const Args& a, const uint32_t bx, const uint32_t nbx) {
    GridView src = a.src, tgt = a.tgt; const GicpState* __restrict__ st = a.st; const double thr2 = a.thr2; float r0 = a.r0; const int max_rounds = a.max_rounds;
    int32_t* __restrict__ corr = a.corr; float* __restrict__ sqd = a.sqd; int32_t* __restrict__ nn_idx = a.nn_idx; float4* __restrict__ nn_ref = a.nn_ref;
    uint2* __restrict__ fb_list = a.fb_list; uint32_t* __restrict__ fb_count = a.fb_count; uint2* __restrict__ big_list = a.big_list; uint32_t* __restrict__ big_count = a.big_count;
    const int big_blocks = a.big_blocks; const float big_ratio = a.big_ratio; uint32_t* __restrict__ far_stats = a.far_stats; const NnOpt opt = a.opt;

  __shared__ WaveLds lds[BLOCK / 64];
  if (MODE == 0 && uni(st->phase) != 0) return;
  if (MODE == 1 && uni(st->phase) != 2) return;
  if (opt.cond && !(uni(st->reserved) & opt.cond)) return;           // a conditional launch behind look_decide
  src = grid_resolve(src); tgt = grid_resolve(tgt); if (r0 < 0.f) r0 = -r0 * tgt.cell;
  // the f32 pose lives in LDS and is read where a query is transformed (broadcast reads): as twelve loop-invariant values it competed with ~90 other uniform numbers for the
  // scalar registers, lost, and was kept in VECTOR registers across the searches - spilled to scratch there
  __shared__ float sTf[12];
  if (threadIdx.x < 12) sTf[threadIdx.x] = (float)st->x0[threadIdx.x];
  __syncthreads();
  if (LIST && (int)bx >= (int)nbx - big_blocks) {            // ---- big entries: one query per wave
    const uint32_t nbig = uni(*big_count);
    if (tgt.dbg && bx == nbx - 1 && threadIdx.x == 0) atomicAdd(&tgt.dbg[7], nbig);
    uint32_t nfar = 0;                                                     // far queries of this pass (the host hands far_stats only to the LAST unseeded pass, when the clouds are nearly aligned)
    const uint32_t bw0 = (bx - (nbx - big_blocks)) * (BLOCK / 64) + (threadIdx.x >> 6), nbw = big_blocks * (BLOCK / 64);
    // Long lists (the first unseeded ticks of a misaligned pair, the non-overlapping part of a partial overlap): a wave takes E CONSECUTIVE entries - the list keeps
    // the cell-sorted order of the queries, so they are neighbours in space - and serves the ones whose balls overlap with ONE shared scan (wave_search_far16);
    // loners still get the whole wave (wave_search_single).  Short lists stay one entry per wave: nothing to share, and every entry starts at once.
    const uint32_t E = (GROUP && opt.group && nbig >= (uint32_t)opt.group_min) ? min(16u, (nbig + (uint32_t)opt.group - 1u) / (uint32_t)opt.group) : 1u;
    if (GROUP && E > 1u) {
      const int lane = threadIdx.x & 63, qs = lane & 15;
      WaveLds* wl = &lds[threadIdx.x >> 6];
      const float INF = __int_as_float(0x7f800000);
      for (uint32_t w0 = bw0 * E; w0 < nbig; w0 += nbw * E) {
        const uint32_t slot = w0 + (uint32_t)qs;
        const bool active = (uint32_t)qs < E && slot < nbig;
        const uint2 rec = active ? big_list[slot] : make_uint2(0u, 0u);
        const float4 p = active ? src.pts[rec.x] : make_float4(0.f, 0.f, 0.f, 0.f);
        float qx, qy, qz; { float Tf[12];
#pragma unroll
          for (int j = 0; j < 12; j++) Tf[j] = sTf[j];
          xform_query<MODE>(Tf, p.x, p.y, p.z, qx, qy, qz); }
        const float v = __uint_as_float(rec.y);
        const float r = v > 0.f ? v * 1.1f + 0.5f * tgt.cell : -v;
        const bool finite_q = (qx - qx == 0.f) && (qy - qy == 0.f) && (qz - qz == 0.f) && (r - r == 0.f);
        unsigned long long key = QN_INF_KEY; float second = INF, d_unseen = INF;
        unsigned long long todo = __ballot(active);
        while (todo) {
          const int al = __ffsll((long long)todo) - 1;                 // anchor: the first open query (its sub-slot 0 lane)
          const float ax = rdlane(qx, al), ay = rdlane(qy, al), az = rdlane(qz, al), ar = rdlane(r, al);
          const bool afin = rdlane((int)finite_q, al) != 0;
          const bool open_q = (todo >> lane) & 1ull;
          // members: within half the anchor's radius of it (their balls then share most of their volume); a non-finite entry is served alone
          const bool member = open_q && (qs == (al & 15) || (afin && finite_q && sqdist(qx, qy, qz, ax, ay, az) <= 0.25f * ar * ar));
          const unsigned long long mm = __ballot(member);
          if (__popcll(mm) <= 4) {                                     // a loner (4 lanes): the whole wave on its ball
            unsigned long long k1; float s1, du1;
            wave_search_single(tgt, ax, ay, az, ar, INF, k1, s1, du1, wl);
            if (member) { key = k1; second = s1; d_unseen = du1; }
          } else {
            wave_search_far16(tgt, qx, qy, qz, member, r, key, second, d_unseen, wl);
          }
          todo &= ~mm;
        }
        if (active && lane < 16) {
          const uint32_t tl = late(rec.x);
          store_nn<MODE>(key, late(__float_as_uint(p.w)), tl, thr2, corr, sqd, nn_idx);
          // (the bound every OTHER target point respects is the canonical one here - the neighbour's own distance - not the scan's runner-up / unseen radius: those
          //  depend on which entries shared a scan, i.e. on the order the list was appended in, and the tracked ticks' regime decisions must not)
          if (MODE == 0) nn_ref[tl] = make_float4(qx, qy, qz, key != QN_INF_KEY ? sqrtf(key_d2(key)) : INF);
        }
        if (far_stats && MODE == 0) {
          const unsigned long long fm = __ballot(active && lane < 16 && key != QN_INF_KEY && key_d2(key) > 36.f * tgt.cell * tgt.cell);
          if (lane == 0) nfar += (uint32_t)__popcll(fm);
        }
      }
      if (far_stats && MODE == 0 && (threadIdx.x & 63) == 0 && nfar) atomicAdd(&far_stats[3], nfar);
      return;
    }
    unsigned long long pr_sum = 0, pr_max = 0, pr_a = 0, pr_b = 0; uint32_t pr_n = 0;
    // A SHORT 16-per-wave list is served here as well, one entry per wave: a few hundred list waves grinding through growth rounds for all their queries were the
    // long pole of the later unseeded passes (55 us against 20).  The decision is taken from the list's actual length: a pair that is still metres off at its
    // second iteration leaves tens of thousands of leftovers there, and those need the 16-per-wave pass (435 us otherwise).
    const uint32_t nfb_all = uni(*fb_count);
    const uint32_t nfb1 = (opt.fb_small && nfb_all <= opt.fb_small) ? nfb_all : 0u;
    for (uint32_t w = bw0; w < nbig + nfb1; w += nbw) {
      const uint2 rec = w < nbig ? big_list[w] : fb_list[w - nbig];
      const float4 p = src.pts[rec.x];
      float qx, qy, qz; { float Tf[12];
#pragma unroll
          for (int j = 0; j < 12; j++) Tf[j] = sTf[j];
          xform_query<MODE>(Tf, p.x, p.y, p.z, qx, qy, qz); }
      const float v = __uint_as_float(rec.y);
      // v > 0: tight seed (the query barely moved since its last scan): scan a little wider than the bound so that the
      // following iterations can prove the neighbour unchanged; v < 0: unseeded, continue from |v|
      const float r = v > 0.f ? v * 1.1f + 0.5f * tgt.cell : -v;
      unsigned long long key; float second, d_unseen;
      const unsigned long long pt0 = opt.probe ? wall_clock64() : 0ull;
      SingleStats sst;
      wave_search_single(tgt, qx, qy, qz, r, __int_as_float(0x7f800000), key, second, d_unseen, &lds[threadIdx.x >> 6], sst, opt.probe != nullptr);
      if (opt.probe) { const unsigned long long dt = wall_clock64() - pt0; pr_sum += dt; pr_n++;
        if (dt > pr_max) { pr_max = dt; pr_a = ((unsigned long long)sst.rounds << 48) | ((unsigned long long)min(sst.segs, 0xffffffu) << 24) | (unsigned long long)min(sst.cand, 0xffffffu);
          pr_b = ((unsigned long long)__float_as_uint(sst.r_first) << 32) | __float_as_uint(key != QN_INF_KEY ? sqrtf(key_d2(key)) : -1.f); } }
      if ((threadIdx.x & 63) == 0) {
        const uint32_t tl = late(rec.x);
        store_nn<MODE>(key, late(__float_as_uint(p.w)), tl, thr2, corr, sqd, nn_idx);
        if (MODE == 0) nn_ref[tl] = make_float4(qx, qy, qz, fminf(sqrtf(second), d_unseen));
        if (key != QN_INF_KEY && key_d2(key) > 36.f * tgt.cell * tgt.cell) nfar++;        // neighbour beyond 6 cells (QN_FAR_RMIN_CELLS)
      }
    }
    if (far_stats && MODE == 0 && (threadIdx.x & 63) == 0 && nfar) atomicAdd(&far_stats[3], nfar);
    if (opt.probe && (threadIdx.x & 63) == 0 && bw0 < 16384u) { opt.probe[4 * bw0] = (pr_max << 32) | pr_n; opt.probe[4 * bw0 + 1] = pr_sum; opt.probe[4 * bw0 + 2] = pr_a; opt.probe[4 * bw0 + 3] = pr_b; }
    return;
  }
  const uint32_t nq = LIST ? uni(*fb_count) : src.n;
  if (LIST && opt.fb_small && nq <= opt.fb_small) return;          // (served one per wave by the other blocks of this launch)
  if (LIST && tgt.dbg && bx == 0 && threadIdx.x == 0) atomicAdd(&tgt.dbg[5], nq);
  const uint32_t wave0 = bx * (BLOCK / 64) + (threadIdx.x >> 6), nwaves = (nbx - (LIST ? big_blocks : 0)) * (BLOCK / 64);   // (XCD remap measured slower here: the leftover lists lose their locality)
  for (uint32_t base = wave0 * 16; base < nq; base += nwaves * 16) {
    const uint32_t slot = base + (threadIdx.x & 15);
    const bool active = slot < nq;
    uint32_t t = slot; float r = r0, r_cap = __int_as_float(0x7f800000);
    if (LIST && active) {        // rec.y > 0: seeded entry (proven bound on the NN distance); < 0: continue from |rec.y|
      const uint2 rec = fb_list[slot]; t = rec.x;
      const float v = __uint_as_float(rec.y);
      if (v < 0.f) r = -v; else { r_cap = v; r = fminf(v, r0); }   // seeded (possibly loose after a big pose step): start small, never beyond the bound
    }
    const float4 p = active ? src.pts[t] : make_float4(0, 0, 0, 0);
    float qx, qy, qz; { float Tf[12];
#pragma unroll
          for (int j = 0; j < 12; j++) Tf[j] = sTf[j];
          xform_query<MODE>(Tf, p.x, p.y, p.z, qx, qy, qz); }
    Best1 sink; sink.init();
    float d_unseen;
    const bool cert = wave_search<4>(tgt, qx, qy, qz, active, r, r_cap, max_rounds, sink, &lds[threadIdx.x >> 6], d_unseen);
    const bool mine = active && (threadIdx.x & 48) == 0;
    const bool done = cert || LIST;
    if (!LIST && MODE == 0 && opt.clear_ref && mine) opt.clear_ref[t] = make_float4(0.f, 0.f, 0.f, 0.f);      // no far-candidate list yet (w = 0)
    if (mine && done) {
      const uint32_t tl = late(t);
      store_nn<MODE>(sink.key, late(__float_as_uint(p.w)), tl, thr2, corr, sqd, nn_idx);
      // bound-pruning reference: where this query was scanned and how far away every other point is at least
      if (MODE == 0) nn_ref[tl] = make_float4(qx, qy, qz, fminf(sqrtf(sink.second), d_unseen));
    }
    if (!LIST) {
      const float rn = r;
      const bool far = rn > big_ratio * r0;
      wave_append(big_list, big_count, mine && !done && far, make_uint2(t, __float_as_uint(-rn)));     // far: one query per wave
      wave_append(fb_list, fb_count, mine && !done && !far, make_uint2(t, __float_as_uint(-rn)));      // continue from rn
    }
  }
  }
};
template <int MODE, bool LIST, int BLOCK, bool GROUP = false>
__global__ void __launch_bounds__(BLOCK, GROUP ? 5 : 6) k_nn_search(GridView src, GridView tgt, const GicpState* __restrict__ st, double thr2, float r0, int max_rounds,
                                                        int32_t* __restrict__ corr, float* __restrict__ sqd, int32_t* __restrict__ nn_idx, float4* __restrict__ nn_ref,
                                                        uint2* __restrict__ fb_list, uint32_t* __restrict__ fb_count,
                                                        uint2* __restrict__ big_list, uint32_t* __restrict__ big_count, int big_blocks, float big_ratio, uint32_t* __restrict__ far_stats, NnOpt opt) {
  const NnSearchArgs a{src, tgt, st, thr2, r0, max_rounds, corr, sqd, nn_idx, nn_ref, fb_list, fb_count, big_list, big_count, big_blocks, big_ratio, far_stats, opt};
  NnSearchK<MODE, LIST, BLOCK, GROUP>::run(a, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------ K4b / K5 accumulate
// phase 0: linearize: M = (C_B + R C_A R^T)^-1, e = mu_B - T mu_A, J = [skew(T mu_A) | -I],
//          H += J^T M J (21 unique), b += J^T M e, cost += e^T M e.
// phase 1: compute_error at the trial transform xi with the CACHED correspondences and the M of x0.
// Fixed grid, fixed per-thread striding, fixed reduction tree => bitwise reproducible partials.
// one correspondence's contribution: M = (C_B + R C_A R^T)^-1, e = mu_B - T mu_A, J = [skew(T mu_A) | -I];
// acc[0..20] += upper J^T M J, acc[21..26] += J^T M e (both only when `lin`), acc[27] += e^T M e.
// PLANE-regularised covariances are exactly C = I - 0.999 n n^T, so the kernels carry the normals (3 f64 per point
// instead of 6):  C_B + R C_A R^T = (I - 0.999 n_B n_B^T) + (I - 0.999 m m^T),  m = R n_A.
// Rx: 3x4 pose whose rotation block transforms the source normal (x0); Tx: 3x4 pose the residual is evaluated at (x0 when linearising,
// xi in an LM trial pass).  J = [skew(T mu_A) | -I] is mostly zeros and ones: the products are written out term by term, in the order
// of the dense formula, so the sums are bit-identical to it (adding an exact 0 or multiplying by -1 does not round).
__device__ __forceinline__ void accumulate_point_n(const double (*Rx)[4], const double (*Tx)[4], const float4 pa, const float4 pb,
                                                   const double na[3], const double nb[3], const bool lin, double acc[QN_NPART]) {
  double m[3];
#pragma unroll
  for (int a = 0; a < 3; a++) m[a] = Rx[a][0] * na[0] + Rx[a][1] * na[1] + Rx[a][2] * na[2];
  // R C_A R^T = R R^T - 0.999 m m^T.  R R^T is I to rounding for every pose the optimiser builds from the identity guess of
  // loop_closure.cpp:124, but NOT for a caller's f32 guess with a rotation (pcl::Registration::align(output, guess)): its rows are
  // orthonormal to 6e-8 only, the reference multiplies with the matrix as it is, and the difference is 1e-7 of the cost.
  M3 rcr;
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = a; b < 3; b++) {      // the upper triangle, mirrored: the sum of two symmetric matrices, formed once per pair (a, b) - its inverse then has three cofactors less to form
      const double g = Rx[a][0] * Rx[b][0] + Rx[a][1] * Rx[b][1] + Rx[a][2] * Rx[b][2];
      rcr.m[a][b] = ((a == b ? 1.0 : 0.0) - 0.999 * nb[a] * nb[b]) + (g - 0.999 * m[a] * m[b]);
      rcr.m[b][a] = rcr.m[a][b];
    }
  const M3 M = m3_inverse(rcr);
  const double mA[3] = {(double)pa.x, (double)pa.y, (double)pa.z};
  double tA[3], e[3], Me[3];
#pragma unroll
  for (int r = 0; r < 3; r++) tA[r] = Tx[r][0] * mA[0] + Tx[r][1] * mA[1] + Tx[r][2] * mA[2] + Tx[r][3];
  e[0] = (double)pb.x - tA[0]; e[1] = (double)pb.y - tA[1]; e[2] = (double)pb.z - tA[2];
#pragma unroll
  for (int r = 0; r < 3; r++) Me[r] = M.m[r][0] * e[0] + M.m[r][1] * e[1] + M.m[r][2] * e[2];
  acc[27] += e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2];
  if (lin) {
    const double x = tA[0], y = tA[1], z = tA[2];
    // MS = M skew(tA): column 0 = M (0, z, -y), column 1 = M (-z, 0, x), column 2 = M (y, -x, 0);  M J = [MS | -M]
    double MS[3][3];
#pragma unroll
    for (int r = 0; r < 3; r++) { MS[r][0] = M.m[r][1] * z + M.m[r][2] * (-y); MS[r][1] = M.m[r][0] * (-z) + M.m[r][2] * x; MS[r][2] = M.m[r][0] * y + M.m[r][1] * (-x); }
    // upper J^T M J, row-major over (r, c >= r):  rows 0..2 = skew(tA)^T [MS | -M], rows 3..5 = M
    // row 0: J[:,0] = (0, z, -y)
    acc[0] += z * MS[1][0] + (-y) * MS[2][0]; acc[1] += z * MS[1][1] + (-y) * MS[2][1]; acc[2] += z * MS[1][2] + (-y) * MS[2][2];
    // (the rotation-translation block -skew(tA)^T M is minus the TRANSPOSE of MS: M is symmetric bit for bit - the inverse of a mirrored matrix - so z (-M10) + (-y)(-M20)
    //  and -(M01 z + M02 (-y)) are the same products and the same rounded sum: nine entries without a multiplication)
    acc[3] += -MS[0][0]; acc[4] += -MS[1][0]; acc[5] += -MS[2][0];
    // row 1: J[:,1] = (-z, 0, x)
    acc[6] += (-z) * MS[0][1] + x * MS[2][1]; acc[7] += (-z) * MS[0][2] + x * MS[2][2];
    acc[8] += -MS[0][1]; acc[9] += -MS[1][1]; acc[10] += -MS[2][1];
    // row 2: J[:,2] = (y, -x, 0)
    acc[11] += y * MS[0][2] + (-x) * MS[1][2];
    acc[12] += -MS[0][2]; acc[13] += -MS[1][2]; acc[14] += -MS[2][2];
    // rows 3..5: (-I)^T (-M) = M
    acc[15] += M.m[0][0]; acc[16] += M.m[0][1]; acc[17] += M.m[0][2]; acc[18] += M.m[1][1]; acc[19] += M.m[1][2]; acc[20] += M.m[2][2];
    // J^T M e
    acc[21] += z * Me[1] + (-y) * Me[2]; acc[22] += (-z) * Me[0] + x * Me[2]; acc[23] += y * Me[0] + (-x) * Me[1];
    acc[24] += -Me[0]; acc[25] += -Me[1]; acc[26] += -Me[2];
  }
}

// Rows that another block of the SAME launch reads (the last block to finish runs the controller: controller_tail) leave their block as agent-scope write-through
// stores and are read with agent-scope loads - the per-XCD L2s are never asked to be coherent (the hand-off of the persistent kernel, qn_persist.cuh).
typedef __attribute__((address_space(1))) unsigned long long qn_gu64;
__device__ __forceinline__ void pr_store(unsigned long long* p, unsigned long long bits) { __hip_atomic_store((qn_gu64*)p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long pr_load(const unsigned long long* p) { return __hip_atomic_load((qn_gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void row_store(double* p, double v, const bool coherent) {
  if (coherent) pr_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v)); else *p = v;
}

// block-level reduction of the 28 per-thread sums (DPP row sums + 2 cross-row shuffles per wave, then the 4 waves in order)
__device__ __forceinline__ void reduce_block_partials(const double acc[QN_NPART], const bool lin, double* __restrict__ partials, double (*red)[QN_NPART], const uint32_t blk, const bool coherent = false) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lin) {
#pragma unroll
    for (int t = 0; t < QN_NPART; t++) { double v = wave_sum_f64_dpp(acc[t]); if (lane == 63) red[wid][t] = v; }
  } else {
    double v = wave_sum_f64_dpp(acc[27]);
    if (lane == 63) { for (int t = 0; t < 27; t++) red[wid][t] = 0; red[wid][27] = v; }
  }
  __syncthreads();
  if (threadIdx.x < QN_NPART) {
    double s = 0;
#pragma unroll
    for (int w = 0; w < QN_BLOCK / 64; w++) s += red[w][threadIdx.x];
    row_store(&partials[(size_t)blk * QN_NPART + threadIdx.x], s, coherent);   // blk = LOGICAL block: the reduction tree does not depend on the block order
  }
}

// ------------------------------------------------------------------ K4a'' first (unseeded) pass, one query per LANE
// The unseeded passes of an align start with ONE round at radius r0 = one cell (margin_nn): the query's ball box is 3 x 3 x 3 cells, i.e. <= 9 (y, z) rows of <= 2
// segments each (a row crosses a tile boundary at most once), ~40-60 candidates.  The cooperative search (wave_search: 16 queries per wave, cluster boxes, one dense
// candidate stream staged through LDS) spends ~80 wave-instructions per query on that machinery; here every lane walks ITS OWN box - segment bounds of a whole
// pass fetched in one go, candidates two at a time - for ~15-20 per query, four times fewer waves and no LDS.  Neighbouring lanes are neighbours in space (cell-sorted
// order), so their rows are the same cache lines and their trip counts agree.  Same certificate as wave_search (best distance < distance to the nearest box face
// with unseen cells behind it), same radius rule and the same routing of the leftovers (far -> one-per-wave list, near -> 16-per-wave list); results are exact
// either way, so everything downstream sees the same correspondences (tests: test_knn_bit_exact's 1-NN side, test_linearize_and_error, the oracle parity sweeps).
#define QN_LANE_ROWS 9
template <int MODE>
struct NnLaneK {
  static constexpr int TB = 256, OCC = 6;
  using Args = NnSearchArgs;
  static __device__ __forceinline__ void run(const Args& a, const uint32_t bx, const uint32_t) {
    const GicpState* __restrict__ st = a.st; const NnOpt opt = a.opt;
    if (MODE == 0 && uni(st->phase) != 0) return;
    if (MODE == 1 && uni(st->phase) != 2) return;
    if (opt.cond && !(uni(st->reserved) & opt.cond)) return;           // a conditional launch behind look_decide
    const GridView src = grid_resolve(a.src), tgt = grid_resolve(a.tgt);
    float r0 = a.r0; if (r0 < 0.f) r0 = -r0 * tgt.cell;
    float Tf[12];
#pragma unroll
    for (int j = 0; j < 12; j++) Tf[j] = uni((float)st->x0[j]);
    const uint32_t t = bx * TB + threadIdx.x;
    const bool active = t < src.n;
    const float4 p = active ? src.pts[t] : make_float4(0, 0, 0, 0);
    float qx, qy, qz; xform_query<MODE>(Tf, p.x, p.y, p.z, qx, qy, qz);
    if (MODE == 0 && opt.clear_ref && active) opt.clear_ref[t] = make_float4(0.f, 0.f, 0.f, 0.f);      // no far-candidate list yet (w = 0)
    float r = r0;
    const int x0 = cell_coord(qx - r, tgt.ox, tgt.inv_cell, tgt.nx), x1 = cell_coord(qx + r, tgt.ox, tgt.inv_cell, tgt.nx);
    const int y0 = cell_coord(qy - r, tgt.oy, tgt.inv_cell, tgt.ny), y1 = cell_coord(qy + r, tgt.oy, tgt.inv_cell, tgt.ny);
    const int z0 = cell_coord(qz - r, tgt.oz, tgt.inv_cell, tgt.nz), z1 = cell_coord(qz + r, tgt.oz, tgt.inv_cell, tgt.nz);
    const int tx0 = x0 >> 3, ntr = (x1 >> 3) - tx0 + 1, nyr = y1 - y0 + 1, nzr = z1 - z0 + 1;
    const bool scan = active && nyr <= 3 && nzr <= 3 && ntr <= 2;       // (a wider first radius - knob margin_nn - may not fit: such a query goes to the lists unscanned, with its radius)
    Best1 sink; sink.init();
    const float4* __restrict__ pts = tgt.pts; const uint32_t* __restrict__ cs = tgt.cell_start;
    // cell_key(x, y, z) = (tile << 7) | (z & 3) << 5 | (y & 3) << 3 | (x & 7) with tile = ((z >> 2) nty + (y >> 2)) ntx + (x >> 3): the z, y and x parts occupy disjoint bits and the
    // tile parts add, so a row's key is ONE three-operand add of per-axis terms formed once (the generic per-row form - a division, the key, two 64-bit addresses - was a fifth of the kernel)
    const uint32_t tile_zy = (uint32_t)tgt.nty * (uint32_t)tgt.ntx;
    uint32_t kz[3], ky[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
      const int z = z0 + d, y = y0 + d;
      kz[d] = (((uint32_t)(z >> 2) * tile_zy) << 7) | ((uint32_t)(z & 3) << 5);
      ky[d] = (((uint32_t)(y >> 2) * (uint32_t)tgt.ntx) << 7) | ((uint32_t)(y & 3) << 3);
    }
#pragma unroll 1
    for (int part = 0; part < 2; part++) {
      if (!__any(scan && part < ntr)) break;
      const int tx = tx0 + part, xa = max(x0, tx << 3), xb = min(x1, (tx << 3) + 7);
      const uint32_t kx = ((uint32_t)tx << 7) | (uint32_t)(xa & 7), xlen = (uint32_t)(xb - xa) + 1u;
      const bool pv = scan && part < ntr;
      uint32_t s[QN_LANE_ROWS], e[QN_LANE_ROWS];
#pragma unroll
      for (int rr = 0; rr < QN_LANE_ROWS; rr++) {
        const int dz = rr / 3, dy = rr % 3;                              // (compile-time: the rows of a 3 x 3 box, those beyond the query's own box empty)
        s[rr] = 0; e[rr] = 0;
        if (pv && dy < nyr && dz < nzr) { const uint32_t k0 = kz[dz] + ky[dy] + kx; s[rr] = cs[k0]; e[rr] = cs[k0 + xlen]; }
      }
#pragma unroll
      for (int rr = 0; rr < QN_LANE_ROWS; rr++) {
        for (uint32_t u = s[rr]; u < e[rr]; u += 4) {                    // four candidates per trip, all four loads in flight (a row is 3 cells: one trip, rarely two)
          const uint32_t last = e[rr] - 1u;
          const float4 c0 = pts[u], c1 = pts[min(u + 1u, last)], c2 = pts[min(u + 2u, last)], c3 = pts[min(u + 3u, last)];
          sink.consider<true>(true, sqdist(qx, qy, qz, c0.x, c0.y, c0.z), __float_as_uint(c0.w));
          sink.consider<true>(u + 1u <= last, sqdist(qx, qy, qz, c1.x, c1.y, c1.z), __float_as_uint(c1.w));
          sink.consider<true>(u + 2u <= last, sqdist(qx, qy, qz, c2.x, c2.y, c2.z), __float_as_uint(c2.w));
          sink.consider<true>(u + 3u <= last, sqdist(qx, qy, qz, c3.x, c3.y, c3.z), __float_as_uint(c3.w));
        }
      }
    }
    // certification: nearest face of the scanned box that has unseen cells behind it (wave_search's rule, on the query's own box)
    const float INF = __int_as_float(0x7f800000);
    bool certified = false; float d_unseen = INF;
    if (scan) {
      float d = INF;
      if (x0 > 0) d = fminf(d, qx - (tgt.ox + x0 * tgt.cell));
      if (x1 < tgt.nx - 1) d = fminf(d, (tgt.ox + (x1 + 1) * tgt.cell) - qx);
      if (y0 > 0) d = fminf(d, qy - (tgt.oy + y0 * tgt.cell));
      if (y1 < tgt.ny - 1) d = fminf(d, (tgt.oy + (y1 + 1) * tgt.cell) - qy);
      if (z0 > 0) d = fminf(d, qz - (tgt.oz + z0 * tgt.cell));
      if (z1 < tgt.nz - 1) d = fminf(d, (tgt.oz + (z1 + 1) * tgt.cell) - qz);
      if (d == INF || !(r == r)) { certified = true; d_unseen = INF; }      // the whole grid was scanned (or a non-finite query: nothing to find)
      else { d -= tgt.eps; d_unseen = d; certified = d > 0.f && sink.full() && sink.worst_d2() < d * d; }
      if (!certified) r = sink.full() ? fmaxf(sqrtf(sink.worst_d2()) * 1.000001f + tgt.eps, r + tgt.eps) : 2.f * r + tgt.cell;      // the radius the list pass continues from
    }
    if (active && certified) {
      store_nn<MODE>(sink.key, __float_as_uint(p.w), t, a.thr2, a.corr, a.sqd, a.nn_idx);
      if (MODE == 0) a.nn_ref[t] = make_float4(qx, qy, qz, fminf(sqrtf(sink.second), d_unseen));      // bound-pruning reference: where this query was scanned and how far away every other point is at least
    }
    const bool far = r > a.big_ratio * r0;
    wave_append(a.big_list, a.big_count, active && !certified && far, make_uint2(t, __float_as_uint(-r)));     // far: one query per wave
    wave_append(a.fb_list, a.fb_count, active && !certified && !far, make_uint2(t, __float_as_uint(-r)));      // continue from r
  }
};
template <int MODE>
__global__ void __launch_bounds__(256, 6) k_nn_lane(NnSearchArgs a) { NnLaneK<MODE>::run(a, blockIdx.x, gridDim.x); }

// (AccumulateK / k_accumulate: below the controller, whose step the last block of the launch may run - controller_tail)

// ------------------------------------------------------------------ K4a', temporal tracking
// Iterations after the first (and the fitness pass) start from the previous iteration's nearest
// neighbour j0: d(q, p_j0) is an upper bound on the NN distance, so the exact NN lies in the ball of that
// radius - after the first Gauss-Newton step that is one to eight grid cells.  One query per lane,
// at most QN_TRACK_SEG segments (bounds gathered first, then the points); larger balls go to the
// list passes with the same bound.  Exact by construction: every point that could beat or
// tie j0 (lower index wins) is scanned.
//
// Bound pruning: nn_ref[i] = (position q_ref at which query i was last SCANNED, lower bound d_other on the
// distance from q_ref to every target point other than its neighbour j0).  If the query has moved by delta
// since, every other point is still >= d_other - delta away, so  d(q, p_j0) + delta < d_other  PROVES that
// j0 is still the unique nearest neighbour and the scan is skipped - bit-identical result, no search.
// As the optimiser converges delta -> 0 and almost every query takes this path.
//
// The optimiser ticks of the tracked regime use the same logic inside k_tick (qn_tick.cuh); this kernel serves the fitness pass
// (MODE 1) and the unfused / verification path (MODE 0).
#define QN_TRACK_SEG 8
// The pruning inequality  d(q, p_j0) + |q - q_ref| < d_other  evaluated in f32 with a margin that dominates every rounding
// error involved.  Exact quantities: D0 = |q - p_j0|, DELTA = |q - q_ref|, B = the stored bound.  Computed: d0 (3 subtractions,
// 3 products, 2 sums: relative error <= 3.5 u with u = 2^-24 - the subtractions of f32 coordinates are exact to 1 ulp of the
// DIFFERENCE), its sqrtf (<= 0.5 u more after halving), the same for delta, one f32 sum (1 u).  So
// sqrt(d0) + delta >= (D0 + DELTA) (1 - 4 u) and the stored bound B was itself rounded DOWN from a quantity with the same
// error budget (<= 4 u); a factor 1 + 2^-18 (= 1 + 64 u) on the left covers both with a 8x reserve.  The bound B is a lower
// bound on the distance from q_ref to every OTHER target point in exact arithmetic up to those 4 u, so the triangle
// inequality gives  |q - p| >= B - DELTA > D0  for every other p: j0 stays the unique nearest neighbour, ties included
// (strict inequality).  tests/test_gpu_adversarial.py exercises it on lattices and at +8 km offsets against a fresh search.
__device__ __forceinline__ bool track_bound_holds(float d0, float delta, float bound) {
  return (sqrtf(d0) + delta) * 1.000004f < bound;
}
template <int MODE>
__global__ void __launch_bounds__(QN_BLOCK, 6) k_nn_track(GridView src, GridView tgt, const float4* __restrict__ tgt_raw, const GicpState* __restrict__ st,
                                                       double thr2, int32_t* __restrict__ corr, float* __restrict__ sqd, int32_t* __restrict__ nn_idx,
                                                       float4* __restrict__ nn_ref, uint2* __restrict__ fb_list, uint32_t* __restrict__ fb_count,
                                                       uint2* __restrict__ big_list, uint32_t* __restrict__ big_count) {
  if (MODE == 0 && st->phase != 0) return;
  if (MODE == 1 && st->phase != 2) return;
  src = grid_resolve(src); tgt = grid_resolve(tgt);
  float Tf[12];
#pragma unroll
  for (int j = 0; j < 12; j++) Tf[j] = (float)st->x0[j];
  const uint32_t lblk = xcd_block(blockIdx.x, gridDim.x);          // XCD x tracks one contiguous eighth of the sorted queries
  const uint32_t t = lblk * QN_BLOCK + threadIdx.x;
  if (t >= src.n) return;
  const float INF = __int_as_float(0x7f800000);
  const float4 p = src.pts[t];
  const uint32_t i = __float_as_uint(p.w);
  float qx, qy, qz; xform_query<MODE>(Tf, p.x, p.y, p.z, qx, qy, qz);
  unsigned long long best = QN_INF_KEY; float second = INF, d_unseen = INF;
  bool rescanned = false, big = false, noseed = false;              // noseed: nn_idx = -1 after a non-finite pose, or an index from another target
  float r = 0.f, delta = 0.f;
  const bool finite_q = (qx - qx == 0.f) && (qy - qy == 0.f) && (qz - qz == 0.f);   // a non-finite query has no neighbour: corr = -1, like the first search
  const uint32_t j0 = (uint32_t)nn_idx[t];
  if (finite_q && j0 >= tgt.n) { noseed = true; big = true; r = tgt.cell; }
  if (finite_q && !noseed) {
    const float4 ref = nn_ref[t];
    const float4 p0 = tgt_raw[j0];
    const float d0 = sqdist(qx, qy, qz, p0.x, p0.y, p0.z);
    best = pack_key(d0, j0);
    delta = sqrtf(sqdist(qx, qy, qz, ref.x, ref.y, ref.z));
    if (track_bound_holds(d0, delta, ref.w)) {                      // proven: j0 is still the unique NN
      if (tgt.dbg && (threadIdx.x & 63) == 0) atomicAdd(&tgt.dbg[6], (uint32_t)__popcll(__ballot(1)));
    } else {
      r = sqrtf(d0) * 1.000001f + tgt.eps;
      const int bx0 = cell_coord(qx - r, tgt.ox, tgt.inv_cell, tgt.nx), bx1 = cell_coord(qx + r, tgt.ox, tgt.inv_cell, tgt.nx);
      const int by0 = cell_coord(qy - r, tgt.oy, tgt.inv_cell, tgt.ny), by1 = cell_coord(qy + r, tgt.oy, tgt.inv_cell, tgt.ny);
      const int bz0 = cell_coord(qz - r, tgt.oz, tgt.inv_cell, tgt.nz), bz1 = cell_coord(qz + r, tgt.oz, tgt.inv_cell, tgt.nz);
      const int tx0 = bx0 >> 3, ntr = (bx1 >> 3) - tx0 + 1, nyr = by1 - by0 + 1;
      const int nseg = ntr * nyr * (bz1 - bz0 + 1);
      if (!(d0 == d0) || nseg > QN_TRACK_SEG) big = true;           // big ball
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
            const uint32_t k0 = cell_key(tgt, xa, ry, rz);
            s[sg] = tgt.cell_start[k0]; e[sg] = tgt.cell_start[k0 + (xb - xa) + 1];
          }
        }
#pragma unroll
        for (int sg = 0; sg < QN_TRACK_SEG; sg++) {
          for (uint32_t u = s[sg]; u < e[sg]; u++) {
            const float4 a = tgt.pts[u];
            const float da = sqdist(qx, qy, qz, a.x, a.y, a.z);
            const unsigned long long ka = pack_key(da, __float_as_uint(a.w));
            if (ka < best) { second = key_d2(best); best = ka; }
            else if (ka != best && da < second) second = da;
          }
        }
        // distance to the faces of the scanned cell box that have unseen cells behind them
        float d = INF;
        if (bx0 > 0) d = fminf(d, qx - (tgt.ox + bx0 * tgt.cell));
        if (bx1 < tgt.nx - 1) d = fminf(d, (tgt.ox + (bx1 + 1) * tgt.cell) - qx);
        if (by0 > 0) d = fminf(d, qy - (tgt.oy + by0 * tgt.cell));
        if (by1 < tgt.ny - 1) d = fminf(d, (tgt.oy + (by1 + 1) * tgt.cell) - qy);
        if (bz0 > 0) d = fminf(d, qz - (tgt.oz + bz0 * tgt.cell));
        if (bz1 < tgt.nz - 1) d = fminf(d, (tgt.oz + (bz1 + 1) * tgt.cell) - qz);
        d_unseen = d - tgt.eps; rescanned = true;
      }
    }
  }
  // list passes, seeded with the bound.  tight seed (the query barely moved since it was scanned) AND far neighbour: one query per wave
  const bool tight_far = r > 2.5f * tgt.cell && delta < 0.25f * r;
  wave_append(big_list, big_count, big && tight_far, make_uint2(t, __float_as_uint(r)));
  wave_append(fb_list, fb_count, big && !tight_far, make_uint2(t, __float_as_uint(noseed ? -r : r)));    // negative: unseeded, continue from |r|
  if (big) return;
  store_nn<MODE>(best, i, t, thr2, corr, sqd, nn_idx);
  if (MODE == 0 && rescanned) nn_ref[t] = make_float4(qx, qy, qz, fminf(sqrtf(second), d_unseen));
}

// ------------------------------------------------------------------ verification of the tracked passes (debug knob "verify_track")
// After a tracked / bound-pruned NN pass, a FRESH unseeded search of the same pose runs into scratch buffers and every query is
// compared: nn_idx (ungated NN per cell-sorted position) and, where the pass wrote them, the f32 squared distances bit for bit.
static __global__ void k_reset_lists(GicpState* st) { if (threadIdx.x == 0 && blockIdx.x == 0) { st->fb_count = 0; st->big_count = 0; } }
static __global__ void k_verify_nn(uint32_t n, const GicpState* __restrict__ st, const int32_t* __restrict__ nn_a, const int32_t* __restrict__ nn_b,
                                   const float* __restrict__ sqd_a, const float* __restrict__ sqd_b, const int32_t* __restrict__ corr_a, const int32_t* __restrict__ corr_b,
                                   uint32_t* __restrict__ counters /* [0] mismatching queries, [1] verified passes, [2] first mismatching position + 1 */) {
  if (st->phase != 0) return;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) atomicAdd(&counters[1], 1u);
  if (t >= n) return;
  bool bad = nn_a[t] != nn_b[t];
  if (sqd_a) bad = bad || __float_as_uint(sqd_a[t]) != __float_as_uint(sqd_b[t]) || corr_a[t] != corr_b[t];   // these two are indexed by original point: any bijection of [0, n) will do
  if (bad) { atomicAdd(&counters[0], 1u); atomicCAS(&counters[2], 0u, t + 1u); }
}

// ------------------------------------------------------------------ K6 solver / LM-GN controller
__device__ inline void d_so3_exp(const double om[3], double R[3][3]) {
  double theta_sq = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
  double imag, real;
  if (theta_sq < 1e-10) {
    double theta_quad = theta_sq * theta_sq;
    imag = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * theta_quad;
    real = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * theta_quad;
  } else {
    double theta = sqrt(theta_sq), half = 0.5 * theta;
    imag = sin(half) / theta; real = cos(half);
  }
  double w = real, x = imag * om[0], y = imag * om[1], z = imag * om[2];
  double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0][0] = 1 - (tyy + tzz); R[0][1] = txy - twz;       R[0][2] = txz + twy;
  R[1][0] = txy + twz;       R[1][1] = 1 - (txx + tzz); R[1][2] = tyz - twx;
  R[2][0] = txz - twy;       R[2][1] = tyz + twx;       R[2][2] = 1 - (txx + tyy);
}

// Unpivoted LDL^T, fully unrolled (static register indexing).  H + lambda I is SPD in every healthy
// registration; returns false when a pivot is not strictly positive so the caller can take the
// pivoted path (semi-definite systems: too few correspondences).  Same solution as the pivoted
// factorisation up to rounding.
__device__ __forceinline__ bool d_ldlt_solve6_fast(const double Ain[36], double diag_add, double rhs[6], double x[6], const double* b) {      // rhs = -b, fetched AFTER the factorisation (it is not needed before: six f64 less across it)
  double A[6][6];
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j < 6; j++) A[i][j] = Ain[6 * i + j] + (i == j ? diag_add : 0.0);
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const double d = A[k][k];
    ok = ok && (d > 0.0) && (d < 1.7976931348623157e308);
    const double inv = 1.0 / d;
    double l[6];
#pragma unroll
    for (int i = k + 1; i < 6; i++) l[i] = A[i][k] * inv;
#pragma unroll
    for (int i = k + 1; i < 6; i++)
#pragma unroll
      for (int j = k + 1; j <= i; j++) A[i][j] -= l[i] * d * l[j];
#pragma unroll
    for (int i = k + 1; i < 6; i++) A[i][k] = l[i];
  }
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; i++) rhs[i] = -b[i];
#pragma unroll
  for (int i = 0; i < 6; i++) { double s = rhs[i];
#pragma unroll
    for (int j = 0; j < i; j++) s -= A[i][j] * y[j]; y[i] = s; }
#pragma unroll
  for (int i = 0; i < 6; i++) y[i] = y[i] / A[i][i];
#pragma unroll
  for (int i = 5; i >= 0; i--) { double s = y[i];
#pragma unroll
    for (int j = i + 1; j < 6; j++) s -= A[j][i] * x[j]; x[i] = s; }
  return ok;
}

// LDL^T with diagonal pivoting (what Eigen::LDLT does), 6x6, f64.  The rarely taken general path: all its dynamically indexed work
// arrays live in LDS (SolveWork) - a kernel whose private arrays go to scratch pays ~5 us more per dispatch on gfx950.
struct SolveWork { double A[6][6]; double l[6], y[6], z[6], rhs[6]; int perm[6]; int pad[2]; };
__device__ inline void d_ldlt_solve6(const double Ain[36], double diag_add, double x[6] /* LDS */, SolveWork* w /* LDS; w->rhs set by the caller */) {
  double (*A)[6] = w->A; int* perm = w->perm; double* l = w->l; double* y = w->y; double* z = w->z;
#pragma unroll 1
  for (int i = 0; i < 6; i++) {                                        // (row by row, not unrolled: unrolled, the 36 entries were all in flight at once - the spill of the kernels that carry this step)
    perm[i] = i;
#pragma unroll
    for (int j = 0; j < 6; j++) A[i][j] = Ain[6 * i + j] + (i == j ? diag_add : 0.0);
  }
  for (int k = 0; k < 6; k++) {
    int piv = k; double best = fabs(A[k][k]);
    for (int i = k + 1; i < 6; i++) if (fabs(A[i][i]) > best) { best = fabs(A[i][i]); piv = i; }
    if (piv != k) {
      for (int j = 0; j < 6; j++) { double t = A[k][j]; A[k][j] = A[piv][j]; A[piv][j] = t; }
      for (int i = 0; i < 6; i++) { double t = A[i][k]; A[i][k] = A[i][piv]; A[i][piv] = t; }
      int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
    }
    double d = A[k][k];
    if (d == 0.0) continue;
    for (int i = k + 1; i < 6; i++) l[i] = A[i][k] / d;
    for (int i = k + 1; i < 6; i++) for (int j = k + 1; j <= i; j++) { A[i][j] -= l[i] * d * l[j]; A[j][i] = A[i][j]; }
    for (int i = k + 1; i < 6; i++) A[i][k] = l[i];
  }
  for (int i = 0; i < 6; i++) { double s = w->rhs[perm[i]]; for (int j = 0; j < i; j++) s -= A[i][j] * y[j]; y[i] = s; }
  for (int i = 0; i < 6; i++) y[i] = (A[i][i] != 0.0) ? y[i] / A[i][i] : 0.0;
  for (int i = 5; i >= 0; i--) { double s = y[i]; for (int j = i + 1; j < 6; j++) s -= A[j][i] * z[j]; z[i] = s; }
  for (int i = 0; i < 6; i++) x[perm[i]] = z[i];
}

__device__ __forceinline__ void d_iso_mul(const double A[16], const double B[16], double C[16]) {
  double r[16];
#pragma unroll
  for (int i = 0; i < 16; i++) r[i] = 0;
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) r[4 * i + j] = A[4 * i] * B[j] + A[4 * i + 1] * B[4 + j] + A[4 * i + 2] * B[8 + j];
    r[4 * i + 3] = A[4 * i] * B[3] + A[4 * i + 1] * B[7] + A[4 * i + 2] * B[11] + A[4 * i + 3];
  }
  r[15] = 1.0;
#pragma unroll
  for (int i = 0; i < 16; i++) C[i] = r[i];
}

__device__ inline bool d_is_converged(const double delta[16], const GicpConfig& cfg, double* mr_out, double* mt_out) {
  double mr = 0, mt = 0;
  for (int a = 0; a < 3; a++) { for (int b = 0; b < 3; b++) mr = fmax(mr, fabs(delta[4 * a + b] - (a == b ? 1.0 : 0.0))); mt = fmax(mt, fabs(delta[4 * a + 3])); }
  if (mr_out) { *mr_out = mr; *mt_out = mt; }
  return fmax(mr / cfg.rotation_epsilon, mt / cfg.transformation_epsilon) < 1.0;
}

__device__ __forceinline__ void d_propose(GicpState* st, double lambda, SolveWork* A) {      // d = LDLT(H + lambda I).solve(-b); delta; xi = delta * x0
  double Hl[36], rhs[6], dl[6];
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j < 6; j++) Hl[6 * i + j] = j <= i ? st->H[6 * i + j] : 0.0;      // (the fast factorisation reads the lower triangle only)
  if (!d_ldlt_solve6_fast(Hl, lambda, rhs, dl, st->b)) {
#pragma unroll
    for (int i = 0; i < 6; i++) A->rhs[i] = rhs[i];
    d_ldlt_solve6(st->H, lambda, st->d, A);
#pragma unroll
    for (int i = 0; i < 6; i++) dl[i] = st->d[i];
  }
  double R[3][3]; d_so3_exp(dl, R);
  double delta[16], xi[16];
#pragma unroll
  for (int i = 0; i < 16; i++) delta[i] = 0;
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int b = 0; b < 3; b++) delta[4 * a + b] = R[a][b];
    delta[4 * a + 3] = dl[3 + a];
  }
  delta[15] = 1.0;
  double x0l[16];                                                      // (fetched here, not next to H: 36 + 16 f64 live at once were what the kernels that carry this step spilled)
#pragma unroll
  for (int i = 0; i < 16; i++) x0l[i] = st->x0[i];
  d_iso_mul(delta, x0l, xi);
#pragma unroll
  for (int i = 0; i < 6; i++) st->d[i] = dl[i];
#pragma unroll
  for (int i = 0; i < 16; i++) { st->delta[i] = delta[i]; st->xi[i] = xi[i]; }
}

// after an outer iteration finished (accepted or the rho<0 && converged early return)
__device__ inline void d_finish_outer(GicpState* st, const GicpConfig& cfg, qn_iter_trace* trace, qn_iter_trace tr) {
  bool conv = d_is_converged(st->delta, cfg, &tr.max_dR, &tr.max_dt);
  if (cfg.force_iterations > 0) conv = false;
  if (st->trace_len < QN_MAX_TRACE) { if (trace) trace[st->trace_len] = tr; st->trace_len++; }
  st->outer += 1;
  const int maxit = cfg.force_iterations > 0 ? cfg.force_iterations : cfg.max_iterations;
  if (conv) { st->converged = 1; st->phase = 2; }
  else if (st->outer >= maxit) { st->phase = 2; }
  else st->phase = 0;
}

// mode 0: full controller.  mode 1: reduce a linearisation only (H, b, y0).  mode 2: reduce an error pass only (yi).
__device__ inline void solve_controller(GicpState* st, const double* sums, const GicpConfig& cfg, qn_iter_trace* trace, int mode, int phase, SolveWork* Awork) {
  st->fb_count = 0; st->big_count = 0;
  const bool lin = (mode == 1) || (mode == 0 && phase == 0);
  if (lin) {
    int t = 0;
    for (int r = 0; r < 6; r++) for (int c = r; c < 6; c++, t++) { st->H[6 * r + c] = sums[t]; st->H[6 * c + r] = sums[t]; }
    for (int r = 0; r < 6; r++) st->b[r] = sums[21 + r];
    st->y0 = sums[27];
  } else {
    st->yi = sums[27];
  }
  if (mode != 0) return;

  // One d_propose call site, inlined (a real call would give the kernel a stack = scratch: ~5 us more per dispatch on gfx950).
  bool propose = false, gn_finish = false;
  if (phase == 0) {
    if (cfg.optimizer == QN_OPT_GN) {                                   // step_gn
      propose = true; gn_finish = true;
    } else {                                                            // step_lm, first try of this outer iteration
      if (st->lambda < 0.0) {
        double mx = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) mx = fmax(mx, fabs(st->H[7 * i]));
        st->lambda = cfg.lm_init_lambda_factor * mx;
      }
      st->nu = 2.0; st->inner = 0; st->phase = 1; propose = true;
    }
  } else {                                                              // phase 1: an error pass at xi just finished
    double den = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) den += st->d[i] * (st->lambda * st->d[i] - st->b[i]);
    const double rho = (st->y0 - st->yi) / den;
    st->inner += 1;
    qn_iter_trace tr; tr.y0 = st->y0; tr.lambda = st->lambda; tr.rho = rho; tr.inner = st->inner; tr.accepted = 0; tr.max_dR = tr.max_dt = 0;
    if (rho < 0) {
      if (d_is_converged(st->delta, cfg, nullptr, nullptr)) d_finish_outer(st, cfg, trace, tr);               // `return true` without accepting
      else {
        st->lambda = st->nu * st->lambda; st->nu = 2 * st->nu;
        if (st->inner >= cfg.lm_max_iterations) {                        // "lm not converged!!"
          if (st->trace_len < QN_MAX_TRACE) { if (trace) trace[st->trace_len] = tr; st->trace_len++; }
          st->outer += 1; st->lm_failed = 1; st->phase = 2;
        } else propose = true;                                           // stay in phase 1
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; i++) st->x0[i] = st->xi[i];
      const double c3 = (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
      st->lambda = st->lambda * fmax(1.0 / 3.0, 1 - c3);
#pragma unroll
      for (int i = 0; i < 36; i++) st->final_H[i] = st->H[i];
      tr.accepted = 1;
      d_finish_outer(st, cfg, trace, tr);
    }
  }
  if (propose) d_propose(st, gn_finish ? 0.0 : st->lambda, Awork);
  if (gn_finish) {
#pragma unroll
    for (int i = 0; i < 16; i++) st->x0[i] = st->xi[i];
#pragma unroll
    for (int i = 0; i < 36; i++) st->final_H[i] = st->H[i];
    qn_iter_trace tr; tr.y0 = st->y0; tr.lambda = 0; tr.rho = 0; tr.inner = 1; tr.accepted = 1; tr.max_dR = tr.max_dt = 0;
    d_finish_outer(st, cfg, trace, tr);
  }
}


// Deterministic sum of `rows` partial rows (28 f64 each): QN_ROW_SEGS = 18 strided sub-sums per component (sub-sum (s, c) adds rows s, s + 18, ... in order; a wave
// reads whole rows: coalesced 224-byte runs), combined in a fixed order.  A block of 512 threads gives every (s, c) pair a thread of its own, a block of 256
// threads takes two pairs per thread - the ORDER of the additions is the same for every block size, so every caller (the controller tail of k_tick / k_accumulate /
// k_far_reduce, k_solve, the persistent kernel's reducer) gets the same bits.  The loads of up to 32 rows per sub-sum are issued back to back (ONE memory round
// trip - the rows sit in L2 / MALL, ~1 us away).  COHERENT: the rows were written by other blocks of THIS launch (row_store) - agent-scope loads.
// Ends with a __syncthreads(); sums[] is valid for every thread afterwards.
#define QN_ROWS_BATCH 32
#define QN_ROW_SEGS 18
template <int NT, bool COHERENT>
__device__ __forceinline__ void reduce_rows(const double* __restrict__ part, const int rows, double (*part_s)[QN_ROW_SEGS + 1], double* sums) {
  constexpr int PER = NT / QN_NPART;                                  // (s, c) pairs one pass of the block covers: 18 at 512 threads, 9 at 256
  static_assert(PER >= 1 && QN_ROW_SEGS % PER == 0, "block size must tile the 18 segments");
  const int tid = threadIdx.x;
  if (tid < QN_NPART * PER) {
    const int s0 = tid / QN_NPART, c = tid - s0 * QN_NPART;
    for (int s = s0; s < QN_ROW_SEGS; s += PER) {
      double a = 0;
      for (int r0 = s; r0 < rows; r0 += QN_ROW_SEGS * QN_ROWS_BATCH) {
        double v[QN_ROWS_BATCH];
#pragma unroll
        for (int u = 0; u < QN_ROWS_BATCH; u++) {
          const int r = r0 + QN_ROW_SEGS * u;
          v[u] = r < rows ? (COHERENT ? __longlong_as_double((long long)pr_load((const unsigned long long*)(part + (size_t)r * QN_NPART + c))) : part[(size_t)r * QN_NPART + c]) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < QN_ROWS_BATCH; u++) a += v[u];
      }
      part_s[c][s] = a;
    }
  }
  __syncthreads();
  if (tid < QN_NPART) { double v = 0;
#pragma unroll
    for (int s = 0; s < QN_ROW_SEGS; s++) v += part_s[tid][s]; sums[tid] = v; }
  __syncthreads();
}

struct ResultBlock { qn_gicp_result r; int32_t phase; uint32_t trace_len; uint32_t far_requests, far_misses, far_queries, look; double step_dt, step_dr; };   // step_*: max |t| and max |R - I| of the latest pose step (host: hand-over policy)
// The hand-over decision of a forced Gauss-Newton run, on the device (the host's "look" without the round trip): k_finalize's statistics block, plus
//   bit 0  one more unseeded iteration: the step just taken (translation + rotation x the source cloud's reach from the origin) would move the points by more
//          than 0.4 target cells - a tracked tick would re-search most neighbourhoods;
//   bit 1  the persistent launch may go ahead: at most ~6 % of the source has a far neighbour (otherwise the k_far refresh regime of the chain pays).
// The flags go into the state (the conditional launches behind the controller step read them: k_nn_search / k_accumulate with cond, k_align_persist) and to the
// host.  Runs at the end of the stand-alone controller step (k_solve with LookArgs): no launch of its own.
#define QN_LOOK_EXTRA 1
#define QN_LOOK_GO 2
struct LookArgs { ResultBlock* out; uint32_t* far_stats; const GridDims* sdims; const GridDims* tdims; int allow_extra; int enabled; };
__device__ inline void look_decide(GicpState* st, ResultBlock* out, uint32_t* __restrict__ far_stats, const GridDims* __restrict__ sdims, const GridDims* __restrict__ tdims, int allow_extra) {      // one thread
  const uint32_t fq = far_stats ? far_stats[3] : 0u;
  out->far_requests = far_stats ? far_stats[1] : 0u; out->far_misses = far_stats ? far_stats[0] : 0u; out->far_queries = fq;
  if (far_stats) { far_stats[0] = 0u; far_stats[1] = 0u; far_stats[3] = 0u; }
  out->r.iterations = st->outer; out->r.converged = st->converged; out->r.lm_failed = st->lm_failed; out->r.reserved = 0;
  out->phase = st->phase; out->trace_len = st->trace_len;
  double mr = 0, mt = 0;
  for (int a = 0; a < 3; a++) { for (int b = 0; b < 3; b++) mr = fmax(mr, fabs(st->delta[4 * a + b] - (a == b ? 1.0 : 0.0))); mt = fmax(mt, fabs(st->delta[4 * a + 3])); }
  out->step_dt = mt; out->step_dr = mr;
  const GridDims sg = *sdims;
  double reach2 = 0; const double lo[3] = {sg.ox, sg.oy, sg.oz}, ext[3] = {sg.nx * (double)sg.cell, sg.ny * (double)sg.cell, sg.nz * (double)sg.cell};
  for (int d = 0; d < 3; d++) { const double m = fmax(fabs(lo[d]), fabs(lo[d] + ext[d])); reach2 += m * m; }
  const double moved = mt + mr * sqrt(reach2), ok = 0.4 * (double)tdims->cell;
  int flags = 0;
  if (allow_extra && moved > ok) flags |= QN_LOOK_EXTRA;             // (NaN compares false: no extra iteration)
  if (fq * 16u <= sg.n && st->phase != 2) flags |= QN_LOOK_GO;
  st->reserved = flags; out->look = (uint32_t)flags | 0x100u;        // (0x100: "a device look ran")
}

// The SECOND look of a lone forced run (LookArgs::enabled = 2), at the tail of the conditional extra unseeded iteration: the persistent launch behind it may go ahead only if the
// step that iteration's controller just took is small as well.  A pair that starts a few degrees further off than the bench's default still moves by METRES at its third and
// fourth iteration (tools/gpu_step_trace.py); a tracked tick - inside the persistent kernel too - re-searches every neighbourhood cooperatively then (one such registration:
// 3.4 ms instead of 0.75).  With GO cleared the launch declines, the host carries on unseeded, one iteration per look, until the steps are small (unseeded_goes_on), and
// starts the persistent kernel then.  Keeps the first look's EXTRA bit and its far-query statistics (the extra iteration does not count them again).
__device__ inline void look_again(GicpState* st, ResultBlock* out, const GridDims* __restrict__ sdims, const GridDims* __restrict__ tdims) {      // one thread
  out->r.iterations = st->outer; out->r.converged = st->converged; out->r.lm_failed = st->lm_failed; out->r.reserved = 0;
  out->phase = st->phase; out->trace_len = st->trace_len;
  double mr = 0, mt = 0;
  for (int a = 0; a < 3; a++) { for (int b = 0; b < 3; b++) mr = fmax(mr, fabs(st->delta[4 * a + b] - (a == b ? 1.0 : 0.0))); mt = fmax(mt, fabs(st->delta[4 * a + 3])); }
  out->step_dt = mt; out->step_dr = mr;
  const GridDims sg = *sdims;
  double reach2 = 0; const double lo[3] = {sg.ox, sg.oy, sg.oz}, ext[3] = {sg.nx * (double)sg.cell, sg.ny * (double)sg.cell, sg.nz * (double)sg.cell};
  for (int d = 0; d < 3; d++) { const double m = fmax(fabs(lo[d]), fabs(lo[d] + ext[d])); reach2 += m * m; }
  const double moved = mt + mr * sqrt(reach2), ok = 0.4 * (double)tdims->cell;
  int flags = st->reserved & QN_LOOK_EXTRA;
  if ((st->reserved & QN_LOOK_GO) && !(moved > ok) && st->phase != 2) flags |= QN_LOOK_GO;
  st->reserved = flags; out->look = (uint32_t)flags | 0x300u;       // (0x100: a device look ran, 0x200: the second one too)
}

// The controller step at the TAIL of the launch that produced the partial rows (round 4; before: in the prologue of the NEXT launch, run redundantly by every one
// of its blocks after each had re-read every row - 196 x 196 x 224 B and 196 one-lane f64 solves per tick).  Every block stores its row (row_store, write-through),
// waits for the stores' acknowledgement and takes a ticket; the block that draws the last ticket reads the state the launch ran under, sums the rows in the fixed
// order of reduce_rows, runs the LM / GN controller and writes the NEXT state: generation g -> g + 1 inside the producer.  The following launch starts from one
// 300-byte state.  Same controller code, same order of additions as k_solve and the persistent kernel's reducer => the same bits.
struct TailArgs { const GicpState* st_in; GicpState* st_out; GicpConfig cfg; qn_iter_trace* trace; uint32_t* ticket; int enabled; int rows; LookArgs look; };      // rows: partial rows of the launch
struct TailLds { GicpState sh; double part[QN_NPART][QN_ROW_SEGS + 1]; double sums[QN_NPART]; SolveWork work; int last; };
template <int NT>
__device__ __forceinline__ void controller_tail(const TailArgs& t, const double* __restrict__ rows, const uint32_t nblk, TailLds* L) {
  const int tid = threadIdx.x;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // this block's row (stored by its first wave, which also takes the ticket) has been acknowledged
  if (tid == 0) L->last = __hip_atomic_fetch_add(t.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblk - 1u ? 1 : 0;
  __syncthreads();
  if (!L->last) return;
  if (tid == 0) __hip_atomic_store(t.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (nobody else touches it any more: every other block has drawn)
  static_assert(sizeof(GicpState) % 8 == 0, "GicpState is copied as 8-byte words");
  for (int i = tid; i < (int)(sizeof(GicpState) / 8); i += NT) ((unsigned long long*)&L->sh)[i] = ((const unsigned long long*)t.st_in)[i];
  reduce_rows<NT, true>(rows, t.rows, L->part, L->sums);
  if (tid == 0) {
    const int phase = L->sh.phase;
    if (L->sh.pending && phase != 2) { const GicpConfig cfg = t.cfg; solve_controller(&L->sh, L->sums, cfg, t.trace, 0, phase, &L->work); }
    L->sh.fb_count = 0; L->sh.big_count = 0; L->sh.pending = L->sh.phase != 2 ? 1 : 0;      // the next launch's body writes rows under the new state
    if (t.look.enabled == 1) look_decide(&L->sh, t.look.out, t.look.far_stats, t.look.sdims, t.look.tdims, t.look.allow_extra);
    else if (t.look.enabled == 2) look_again(&L->sh, t.look.out, t.look.sdims, t.look.tdims);
  }
  __syncthreads();
  for (int i = tid; i < (int)(sizeof(GicpState) / 8); i += NT) ((unsigned long long*)t.st_out)[i] = ((const unsigned long long*)&L->sh)[i];
}
// a launch that leaves early (state machine done, or a conditional launch whose flag is not set) hands the state on unchanged: the host advances the generation regardless
template <int NT>
__device__ __forceinline__ void state_pass_through(const TailArgs& t, const uint32_t bx) {
  if (bx != 0) return;
  for (int i = threadIdx.x; i < (int)(sizeof(GicpState) / 8); i += NT) ((unsigned long long*)t.st_out)[i] = ((const unsigned long long*)t.st_in)[i];
}

// One controller step as its own launch (unseeded first ticks, the end of a chunk, the debug entry points): generation g -> g + 1.
// mode 0: the LM / GN controller, if partial rows are pending under st_in.  mode 1 / 2: reduce a linearisation / an error pass only.
// will_produce: a body that writes partial rows under the NEW state follows (so they are pending for the next controller step).
struct SolveArgs { const GicpState* st_in; GicpState* st_out; const double* partials; int rows; GicpConfig cfg; qn_iter_trace* trace; int mode, will_produce; LookArgs look; };
template <int NT>      // NT = the thread count of the k_tick variant in use: both run the same row reduction, bit for bit
struct SolveK {
  static constexpr int TB = NT, OCC = 1;
  using Args = SolveArgs;
  static __device__ __forceinline__ void run(const Args& a, const uint32_t, const uint32_t) {
    __shared__ double sums[QN_NPART];
    __shared__ double part8[QN_NPART][QN_ROW_SEGS + 1];
    __shared__ GicpState sh;                       // the controller works on an LDS copy: one coalesced read, one coalesced write-back
    __shared__ SolveWork Awork_s; SolveWork* Awork = &Awork_s;
    static_assert(sizeof(GicpState) % 8 == 0, "GicpState is copied as 8-byte words");
    const GicpState* __restrict__ st_in = a.st_in; GicpState* __restrict__ st_out = a.st_out;
    const int rows = a.rows, mode = a.mode;
    for (int i = threadIdx.x; i < (int)(sizeof(GicpState) / 8); i += NT) ((unsigned long long*)&sh)[i] = ((const unsigned long long*)st_in)[i];
    __syncthreads();
    const int phase = sh.phase;
    if (mode != 0 || (sh.pending && phase != 2 && rows >= 0)) {        // rows < 0: the pending rows were consumed by an earlier stand-alone controller step
      reduce_rows<NT, false>(a.partials, rows, part8, sums);
      if (threadIdx.x == 0) { const GicpConfig cfg = a.cfg; solve_controller(&sh, sums, cfg, a.trace, mode, phase, Awork); }
    }
    if (threadIdx.x == 0) { sh.fb_count = 0; sh.big_count = 0; sh.pending = (a.will_produce && sh.phase != 2) ? 1 : 0; }
    if (threadIdx.x == 0 && a.look.enabled == 1) look_decide(&sh, a.look.out, a.look.far_stats, a.look.sdims, a.look.tdims, a.look.allow_extra);
    else if (threadIdx.x == 0 && a.look.enabled == 2) look_again(&sh, a.look.out, a.look.sdims, a.look.tdims);
    __syncthreads();
    for (int i = threadIdx.x; i < (int)(sizeof(GicpState) / 8); i += NT) ((unsigned long long*)st_out)[i] = ((const unsigned long long*)&sh)[i];
  }
};
template <int NT>
static __global__ void __launch_bounds__(NT) k_solve(const GicpState* __restrict__ st_in, GicpState* __restrict__ st_out, const double* __restrict__ partials, int rows,
                                                                   GicpConfig cfg, qn_iter_trace* trace, int mode, int will_produce, LookArgs look) {
  const SolveArgs a{st_in, st_out, partials, rows, cfg, trace, mode, will_produce, look};
  SolveK<NT>::run(a, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------ K4b / K5 accumulate (+ the controller step in the launch's last block)
// (AccumulateK / k_accumulate: qn_tick.cuh - the unseeded ticks' sums are formed by emit_point like the tracked ticks')

struct InitStateK {
  static constexpr int TB = 64, OCC = 1;
  struct Args { GicpState* st; const float* guess /* 16 or null */; int has_guess, phase; uint32_t* far_stats /* 4 words or null: zeroed */; };
  static __device__ __forceinline__ void run(const Args& a, const uint32_t bx, const uint32_t) {
    if (bx != 0) return;
    GicpState* st = a.st; const float* __restrict__ guess = a.guess; const int phase = a.phase;
    const int i = threadIdx.x;                                         // one wave: lane i writes element i of every array
    if (i < 16) { const double v = a.has_guess ? (double)guess[i] : ((i % 5 == 0) ? 1.0 : 0.0); st->x0[i] = v; st->xi[i] = v; st->delta[i] = (i % 5 == 0) ? 1.0 : 0.0; }
    if (i < 36) { st->H[i] = 0; st->final_H[i] = (i % 7 == 0) ? 1.0 : 0.0; }
    if (i < 6) { st->b[i] = 0; st->d[i] = 0; }
    if (i >= 60 && a.far_stats) a.far_stats[i - 60] = 0u;
    if (i != 0) return;
    st->y0 = st->yi = st->den = 0; st->lambda = -1.0; st->nu = 2.0; st->fitness = 0;
    st->outer = st->inner = 0; st->phase = phase; st->converged = 0; st->lm_failed = 0; st->fb_count = 0; st->big_count = 0; st->trace_len = 0;
    st->pending = phase != 2 ? 1 : 0; st->reserved = 0;      // the first tick's body writes partial rows under this state
  }
};
static __global__ void __launch_bounds__(64) k_init_state(GicpState* st, const float* __restrict__ guess, int has_guess, int phase, uint32_t* __restrict__ far_stats) {
  const InitStateK::Args a{st, guess, has_guess, phase, far_stats};
  InitStateK::run(a, blockIdx.x, gridDim.x);
}
static __global__ void k_set_pose(GicpState* st, const double* __restrict__ T, int which /*0 x0, 1 xi, 2 neither*/, int phase) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int i = 0; i < 16; i++) { if (which == 0) st->x0[i] = T[i]; else if (which == 1) st->xi[i] = T[i]; }
  st->phase = phase; st->fb_count = 0; st->big_count = 0; st->pending = phase != 2 ? 1 : 0;
}

// ------------------------------------------------------------------ K7 fitness reduce, K8 transform
// pcl getFitnessScore: mean of the f32 squared NN distances <= max_range, summed in f64.
#define QN_FIT_BLOCKS 128
static __global__ void __launch_bounds__(QN_BLOCK) k_fitness_partial(const float* __restrict__ sqd, uint32_t n, double max_range, const GicpState* __restrict__ st,
                                                              double* __restrict__ psum, uint32_t* __restrict__ pcnt, int require_done) {
  __shared__ double ssum[QN_BLOCK / 64]; __shared__ uint32_t scnt[QN_BLOCK / 64];
  if (require_done && st->phase != 2) return;
  double s = 0; uint32_t c = 0;
  for (uint32_t i = blockIdx.x * QN_BLOCK + threadIdx.x; i < n; i += QN_FIT_BLOCKS * QN_BLOCK) { float d = sqd[i]; if ((double)d <= max_range) { s += (double)d; c++; } }
  s = wave_sum_f64_dpp(s);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) { ssum[threadIdx.x >> 6] = s; scnt[threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0; uint32_t tc = 0;
    for (int w = 0; w < QN_BLOCK / 64; w++) { t += ssum[w]; tc += scnt[w]; }
    psum[blockIdx.x] = t; pcnt[blockIdx.x] = tc;
  }
}
static __global__ void __launch_bounds__(QN_FIT_BLOCKS) k_fitness_final(const double* __restrict__ psum, const uint32_t* __restrict__ pcnt, GicpState* st, int require_done) {
  __shared__ double ssum[QN_FIT_BLOCKS]; __shared__ uint32_t scnt[QN_FIT_BLOCKS];
  if (require_done && st->phase != 2) return;
  ssum[threadIdx.x] = psum[threadIdx.x]; scnt[threadIdx.x] = pcnt[threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0; uint32_t tc = 0;
    for (int w = 0; w < QN_FIT_BLOCKS; w++) { t += ssum[w]; tc += scnt[w]; }       // fixed order
    st->fitness = tc > 0 ? t / tc : 1.7976931348623157e308;
    st->fb_count = 0; st->big_count = 0;
  }
}

// pcl::transformPointCloud(*input_, output, final_transformation_) inside align(); also used for the
// Quatro -> GICP hand-over transformPcd(src, T_q) (loop_closure.cpp:152, utilities.hpp:164-175) in f64 mode.
static __global__ void k_transform_cloud(const float4* __restrict__ in, uint32_t n, const GicpState* __restrict__ st, float4* __restrict__ out, int require_done) {
  if (require_done && st->phase != 2) return;
  float Tf[12];
#pragma unroll
  for (int j = 0; j < 12; j++) Tf[j] = (float)st->x0[j];
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = in[i]; float x, y, z;
  xform_query<1>(Tf, p.x, p.y, p.z, x, y, z);
  out[i] = make_float4(x, y, z, 1.0f);
}


struct FinalizeK {
  static constexpr int TB = 64, OCC = 1;
  struct Args { const GicpState* st; ResultBlock* out /* pinned host memory */; uint32_t* far_stats; };
  static __device__ __forceinline__ void run(const Args& a, const uint32_t bx, const uint32_t) {
    if (threadIdx.x != 0 || bx != 0) return;
    const GicpState* __restrict__ st = a.st; ResultBlock* out = a.out; uint32_t* __restrict__ far_stats = a.far_stats;
    out->far_requests = far_stats ? far_stats[1] : 0u; out->far_misses = far_stats ? far_stats[0] : 0u; out->far_queries = far_stats ? far_stats[3] : 0u;
    // [1]: requests of the last tick k_far served; [0]: misses counted since (mode 2); [3]: far queries of the chunk's last unseeded pass
    if (far_stats) { far_stats[0] = 0u; far_stats[1] = 0u; far_stats[3] = 0u; }
    for (int i = 0; i < 16; i++) { out->r.T64[i] = st->x0[i]; out->r.T[i] = (float)st->x0[i]; }
    for (int i = 0; i < 36; i++) out->r.H[i] = st->final_H[i];
    out->r.fitness = st->fitness; out->r.iterations = st->outer; out->r.converged = st->converged; out->r.lm_failed = st->lm_failed; out->r.reserved = 0;
    out->phase = st->phase; out->trace_len = st->trace_len;
    double mr = 0, mt = 0;
    for (int u = 0; u < 3; u++) { for (int b = 0; b < 3; b++) mr = fmax(mr, fabs(st->delta[4 * u + b] - (u == b ? 1.0 : 0.0))); mt = fmax(mt, fabs(st->delta[4 * u + 3])); }
    out->step_dt = mt; out->step_dr = mr;
  }
};
static __global__ void __launch_bounds__(64) k_finalize(const GicpState* __restrict__ st, ResultBlock* out, uint32_t* __restrict__ far_stats) {   // out lives in pinned host memory
  const FinalizeK::Args a{st, out, far_stats};
  FinalizeK::run(a, blockIdx.x, gridDim.x);
}


}  // namespace qn
