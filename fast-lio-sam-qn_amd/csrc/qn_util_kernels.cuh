// qn_util_kernels.cuh - small utility kernels shared by the translation units of the library (point packing,
// bounding box with ordered-int atomics, 3-kernel exclusive scan).  `static __global__`: every TU gets its own copy.
#pragma once
#include "qn_device.cuh"

namespace qn {

#ifndef QN_BLOCK
#define QN_BLOCK 256
#endif

struct BBoxOut { int mn[3], mx[3]; uint32_t nonfinite; };

__device__ __forceinline__ int f2ord(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__host__ __device__ __forceinline__ float ord2f(int i) { union { int i; float f; } u; u.i = i >= 0 ? i : i ^ 0x7fffffff; return u.f; }

// pack a strided xyz(+junk) host/device layout into float4 (x, y, z, 1) - PointXYZI's data[3] = 1
static __global__ void k_pack_points(const char* __restrict__ in, uint32_t stride, uint32_t n, float4* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = (const float*)(in + (size_t)i * stride);
  out[i] = make_float4(p[0], p[1], p[2], 1.0f);
}

static __global__ void __launch_bounds__(QN_BLOCK) k_bbox(const float4* __restrict__ pts, uint32_t n, BBoxOut* out) {
  __shared__ int smn[QN_BLOCK / 64][3], smx[QN_BLOCK / 64][3], sbad[QN_BLOCK / 64];
  int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  int bad = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float4 p = pts[i];
    if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) { bad = 1; continue; }
    int ox = f2ord(p.x), oy = f2ord(p.y), oz = f2ord(p.z);
    mn[0] = min(mn[0], ox); mn[1] = min(mn[1], oy); mn[2] = min(mn[2], oz);
    mx[0] = max(mx[0], ox); mx[1] = max(mx[1], oy); mx[2] = max(mx[2], oz);
  }
#pragma unroll
  for (int d = 0; d < 3; d++) { mn[d] = wave_min_i(mn[d]); mx[d] = wave_max_i(mx[d]); }
  bad = wave_max_i(bad);
  const int wid = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { for (int d = 0; d < 3; d++) { smn[wid][d] = mn[d]; smx[wid][d] = mx[d]; } sbad[wid] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < QN_BLOCK / 64; w++) { for (int d = 0; d < 3; d++) { mn[d] = min(mn[d], smn[w][d]); mx[d] = max(mx[d], smx[w][d]); } bad |= sbad[w]; }
    for (int d = 0; d < 3; d++) { atomicMin(&out->mn[d], mn[d]); atomicMax(&out->mx[d], mx[d]); }
    if (bad) atomicAdd(&out->nonfinite, 1u);
  }
}

// exclusive scan of counts[0..m) -> out[0..m], out[m] = total.  3 kernels, 4096 items per block.
#define QN_SCAN_ITEMS 16
// `in` and `out` may alias (the radix-sort histograms are scanned in place): no __restrict__ on them; every thread loads
// its QN_SCAN_ITEMS inputs before it stores any output and blocks own disjoint ranges, so in-place is well defined.
static __global__ void k_scan_block(const uint32_t* in, uint32_t m, uint32_t* out, uint32_t* __restrict__ block_sums) {
  __shared__ uint32_t wsum[QN_BLOCK / 64];
  const uint32_t base = (blockIdx.x * QN_BLOCK + threadIdx.x) * QN_SCAN_ITEMS;
  uint32_t v[QN_SCAN_ITEMS], s = 0;
#pragma unroll
  for (int j = 0; j < QN_SCAN_ITEMS; j++) { v[j] = (base + j < m) ? in[base + j] : 0u; s += v[j]; }
  // inclusive wave scan of s
  uint32_t inc = s;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  uint32_t woff = 0;
  for (int w = 0; w < wid; w++) woff += wsum[w];
  uint32_t run = woff + inc - s;
#pragma unroll
  for (int j = 0; j < QN_SCAN_ITEMS; j++) { if (base + j < m) out[base + j] = run; run += v[j]; }
  if (threadIdx.x == QN_BLOCK - 1) block_sums[blockIdx.x] = woff + inc;
}
static __global__ void k_scan_top(uint32_t* block_sums, uint32_t nb) {          // single block, serial over chunks of 256
  __shared__ uint32_t wsum[QN_BLOCK / 64];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nb; base += QN_BLOCK) {
    uint32_t i = base + threadIdx.x;
    uint32_t s = i < nb ? block_sums[i] : 0u, inc = s;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint32_t woff = carry;
    for (int w = 0; w < wid; w++) woff += wsum[w];
    if (i < nb) block_sums[i] = woff + inc - s;
    __syncthreads();
    if (threadIdx.x == QN_BLOCK - 1) carry = woff + inc;
    __syncthreads();
  }
}
static __global__ void k_scan_add(uint32_t* __restrict__ out, uint32_t m, const uint32_t* __restrict__ block_sums, uint32_t total) {
  const uint32_t base = (blockIdx.x * QN_BLOCK + threadIdx.x) * QN_SCAN_ITEMS;
  const uint32_t off = block_sums[blockIdx.x];
#pragma unroll
  for (int j = 0; j < QN_SCAN_ITEMS; j++) if (base + j < m) out[base + j] += off;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[m] = total;
}

// variant whose grand total = block_sums total + last flag (used by the voxel-grid leaf compaction: out[m] = number of set flags)
static __global__ void k_scan_add_total(uint32_t* __restrict__ out, uint32_t m, const uint32_t* __restrict__ block_sums, const uint32_t* __restrict__ in) {
  const uint32_t base = (blockIdx.x * QN_BLOCK + threadIdx.x) * QN_SCAN_ITEMS;
  const uint32_t off = block_sums[blockIdx.x];
#pragma unroll
  for (int j = 0; j < QN_SCAN_ITEMS; j++) if (base + j < m) { out[base + j] += off; if (base + j == m - 1) out[m] = out[base + j] + in[m - 1]; }
}


}  // namespace qn
