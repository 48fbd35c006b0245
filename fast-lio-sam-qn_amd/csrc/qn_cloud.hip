// qn_cloud.hip - the feeder of the hot path, on the GPU.  Own translation unit.
//   LoopClosure::setSrcAndDstCloud (fast_lio_sam_qn/src/loop_closure.cpp:58-108): per keyframe transformPcd
//   (include/utilities.hpp:164-175), concatenation, voxelizePcd = pcl::VoxelGrid (utilities.hpp:38-51);
//   LoopClosure::fetchClosestKeyframeIdx (loop_closure.cpp:34-56).
// Keyframe clouds (PosePcd::pcd_, sensor frame, immutable: include/pose_pcd.hpp:7-19) are uploaded ONCE into a store
// and stay resident in HBM; a loop attempt then assembles its source / target clouds on the device and hands the
// device pointers to qn_icp_alignment_device / the batch API - no point cloud crosses PCIe per attempt.
// VoxelGrid: 64-bit keys (leaf index << 32 | point index) are sorted by leaf index with a hand-written STABLE LSD radix
// sort (8-bit digits, only as many passes as the leaf grid has bits; the input is in ascending point order and stability
// keeps it so inside every leaf), leaf heads are compacted with the engine's own scan kernels and one thread per leaf sums
// its points in ascending point order in f32 - the order the oracle fixes (PCL's own order inside a leaf is unspecified).
#include <hip/hip_runtime.h>
#include <cmath>
#include <string>
#include <vector>
#include <algorithm>
#include "../../include/qn_engine.h"
#include "qn_util_kernels.cuh"

namespace qn {

__global__ void k_kf_transform(const float4* __restrict__ in, uint32_t n, const double* __restrict__ T, float4* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = in[i]; const double x = p.x, y = p.y, z = p.z;
  out[i] = make_float4((float)(((T[0] * x + T[1] * y) + T[2] * z) + T[3]), (float)(((T[4] * x + T[5] * y) + T[6] * z) + T[7]),
                       (float)(((T[8] * x + T[9] * y) + T[10] * z) + T[11]), 1.0f);
}
// pcl::VoxelGrid on a cloud that is not dense (utilities.hpp:38-51 -> pcl::VoxelGrid::applyFilter: `if (!input_->is_dense) if (!isXYZFinite(p)) continue;`,
// getMinMax3D skips them as well): non-finite points take no part - they are dropped by a stable compaction (flag, exclusive scan, scatter).
__global__ void k_finite_flags(const float4* __restrict__ pts, uint32_t n, uint32_t* __restrict__ flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  flag[i] = (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) ? 1u : 0u;
}
__global__ void k_compact_finite(const float4* __restrict__ pts, uint32_t n, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ pos, float4* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (flag[i]) out[pos[i]] = pts[i];
}
struct VoxelDims { float inv; int minb[3]; int div0, div01; };
__global__ void k_voxel_keys(const float4* __restrict__ pts, uint32_t n, VoxelDims d, unsigned long long* __restrict__ keys) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  const int i0 = (int)(floorf(p.x * d.inv) - (float)d.minb[0]);
  const int i1 = (int)(floorf(p.y * d.inv) - (float)d.minb[1]);
  const int i2 = (int)(floorf(p.z * d.inv) - (float)d.minb[2]);
  keys[i] = ((unsigned long long)(uint32_t)(i0 + i1 * d.div0 + i2 * d.div01) << 32) | i;
}
// ---- stable LSD radix sort pass over bits [shift, shift + 8) of the 64-bit key; tile = one 256-thread block, one key per thread
__global__ void __launch_bounds__(QN_BLOCK) k_radix_hist(const unsigned long long* __restrict__ keys, uint32_t n, int shift, uint32_t nblocks, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t i = blockIdx.x * QN_BLOCK + threadIdx.x;
  if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
  __syncthreads();
  hist[threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];             // digit-major: the scan yields global offsets directly
}
__global__ void __launch_bounds__(QN_BLOCK) k_radix_scatter(const unsigned long long* __restrict__ keys, uint32_t n, int shift, uint32_t nblocks,
                                                            const uint32_t* __restrict__ offs, unsigned long long* __restrict__ out) {
  __shared__ uint32_t wcount[QN_BLOCK / 64][256];
  for (int t = threadIdx.x; t < (QN_BLOCK / 64) * 256; t += QN_BLOCK) (&wcount[0][0])[t] = 0;
  __syncthreads();
  const uint32_t i = blockIdx.x * QN_BLOCK + threadIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const bool valid = i < n;
  const unsigned long long key = valid ? keys[i] : 0ull;
  const uint32_t d = (uint32_t)(key >> shift) & 255u;
  unsigned long long same = __ballot(valid);                            // lanes of this wave holding the same digit
#pragma unroll
  for (int b = 0; b < 8; b++) { const unsigned long long m = __ballot((d >> b) & 1u); same &= ((d >> b) & 1u) ? m : ~m; }
  const uint32_t rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
  if (valid && rank == 0) wcount[wid][d] = (uint32_t)__popcll(same);
  __syncthreads();
  if (!valid) return;
  uint32_t before = 0;
  for (int w = 0; w < wid; w++) before += wcount[w][d];
  out[offs[d * nblocks + blockIdx.x] + before + rank] = key;
}

__global__ void k_leaf_flags(const unsigned long long* __restrict__ keys, uint32_t n, uint32_t* __restrict__ flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flag[i] = (i == 0 || (keys[i] >> 32) != (keys[i - 1] >> 32)) ? 1u : 0u;
}
__global__ void k_leaf_heads(const uint32_t* __restrict__ flag, const uint32_t* __restrict__ pos, uint32_t n, uint32_t* __restrict__ heads) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (flag[i]) heads[pos[i]] = i;
  if (i == n - 1) heads[pos[n]] = n;          // pos[n] = number of leaves (exclusive scan total)
}
__global__ void k_leaf_centroids(const float4* __restrict__ pts, const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ heads,
                                 const uint32_t* __restrict__ nleaf_ptr, float4* __restrict__ out) {
  const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= *nleaf_ptr) return;
  const uint32_t a = heads[m], b = heads[m + 1];
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (uint32_t t = a; t < b; t++) { const float4 p = pts[(uint32_t)keys[t]]; sx = sx + p.x; sy = sy + p.y; sz = sz + p.z; }
  const float cnt = (float)(b - a);
  out[m] = make_float4(sx / cnt, sy / cnt, sz / cnt, 1.0f);
}

}  // namespace qn

struct qn_kf_store {
  int device = 0; hipStream_t stream = nullptr;
  std::vector<float4*> clouds; std::vector<uint32_t> sizes;
  float4* concat = nullptr; unsigned long long* keys = nullptr; unsigned long long* keys_alt = nullptr;
  uint32_t* flag = nullptr; uint32_t* pos = nullptr; uint32_t* heads = nullptr; uint32_t* sums = nullptr; size_t cap = 0;
  uint32_t* hist = nullptr; uint32_t* hist_sums = nullptr;
  float4* out[2] = {nullptr, nullptr}; size_t out_cap[2] = {0, 0}; uint32_t out_n[2] = {0, 0};
  double* poses = nullptr; size_t poses_cap = 0;
  qn::BBoxOut* bbox = nullptr; qn::BBoxOut* bbox_host = nullptr; uint32_t* count_host = nullptr; char* staging = nullptr; size_t staging_cap = 0;
  std::string last_error;
};
#define KFCHK(s, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (s)->last_error = std::string(#call) + " -> " + hipGetErrorString(e_); return QN_ERR_HIP; } } while (0)

extern "C" int qn_kf_store_create(int device, qn_kf_store** out) {
  if (!out) return QN_ERR_INVALID_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return QN_ERR_NO_DEVICE;
  qn_kf_store* s = new qn_kf_store(); s->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc(&s->bbox, sizeof(qn::BBoxOut)) != hipSuccess || hipHostMalloc(&s->bbox_host, sizeof(qn::BBoxOut), hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc(&s->count_host, sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) { delete s; return QN_ERR_HIP; }
  *out = s;
  return QN_OK;
}
extern "C" void qn_kf_store_destroy(qn_kf_store* s) {
  if (!s) return;
  (void)hipSetDevice(s->device); if (s->stream) (void)hipStreamSynchronize(s->stream);
  for (float4* p : s->clouds) (void)hipFree(p);
  (void)hipFree(s->concat); (void)hipFree(s->keys); (void)hipFree(s->keys_alt); (void)hipFree(s->flag); (void)hipFree(s->pos); (void)hipFree(s->heads); (void)hipFree(s->sums);
  (void)hipFree(s->hist); (void)hipFree(s->hist_sums); (void)hipFree(s->out[0]); (void)hipFree(s->out[1]); (void)hipFree(s->poses); (void)hipFree(s->bbox); (void)hipFree(s->staging);
  if (s->bbox_host) (void)hipHostFree(s->bbox_host); if (s->count_host) (void)hipHostFree(s->count_host);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}
extern "C" const char* qn_kf_last_error(const qn_kf_store* s) { return s ? s->last_error.c_str() : "null store"; }

// upload one keyframe cloud (sensor frame) - PosePcd::pcd_ - and keep it resident
extern "C" int qn_kf_add(qn_kf_store* s, const float* xyz, uint32_t n, uint32_t stride, int32_t* id_out) {
  if (!s || !id_out || (n && !xyz) || stride < 12 || (stride & 3)) return QN_ERR_INVALID_ARG;
  KFCHK(s, hipSetDevice(s->device));
  float4* d = nullptr;
  if (n) {
    const size_t bytes = (size_t)(n - 1) * stride + 12;
    if (bytes > s->staging_cap) { (void)hipFree(s->staging); s->staging = nullptr; s->staging_cap = 0; KFCHK(s, hipMalloc(&s->staging, bytes + bytes / 2)); s->staging_cap = bytes + bytes / 2; }
    KFCHK(s, hipMalloc(&d, sizeof(float4) * n));
    KFCHK(s, hipMemcpyAsync(s->staging, xyz, bytes, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(qn::k_pack_points, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->staging, stride, n, d);
    KFCHK(s, hipStreamSynchronize(s->stream));
  }
  s->clouds.push_back(d); s->sizes.push_back(n);
  *id_out = (int32_t)s->clouds.size() - 1;
  return QN_OK;
}

static int kf_reserve(qn_kf_store* s, size_t n) {
  if (n <= s->cap) return QN_OK;
  (void)hipFree(s->concat); (void)hipFree(s->keys); (void)hipFree(s->keys_alt); (void)hipFree(s->flag); (void)hipFree(s->pos); (void)hipFree(s->heads); (void)hipFree(s->sums); (void)hipFree(s->hist); (void)hipFree(s->hist_sums);
  s->concat = nullptr; s->keys = s->keys_alt = nullptr; s->flag = s->pos = s->heads = s->sums = nullptr; s->hist = s->hist_sums = nullptr; s->cap = 0;
  const size_t c = n + n / 2 + 1024;
  KFCHK(s, hipMalloc(&s->concat, sizeof(float4) * c)); KFCHK(s, hipMalloc(&s->keys, 8 * c)); KFCHK(s, hipMalloc(&s->keys_alt, 8 * c));
  KFCHK(s, hipMalloc(&s->flag, 4 * (c + 1))); KFCHK(s, hipMalloc(&s->pos, 4 * (c + 1))); KFCHK(s, hipMalloc(&s->heads, 4 * (c + 2)));
  KFCHK(s, hipMalloc(&s->sums, 4 * (c / (QN_BLOCK * QN_SCAN_ITEMS) + 2)));
  const size_t hb = (c + QN_BLOCK - 1) / QN_BLOCK * 256;                   // digit-major block histograms of one radix pass
  KFCHK(s, hipMalloc(&s->hist, 4 * (hb + 1))); KFCHK(s, hipMalloc(&s->hist_sums, 4 * (hb / (QN_BLOCK * QN_SCAN_ITEMS) + 2)));
  s->cap = c;
  return QN_OK;
}

// transform + concatenate `count` resident keyframes with their poses (row-major 4x4 f64) and voxel-grid them into
// output slot 0 (source) or 1 (target); returns the device pointer (float4, stride 16) and the point count.
extern "C" int qn_kf_assemble(qn_kf_store* s, const int32_t* ids, const double* poses, uint32_t count, double leaf, int slot,
                              const float** d_xyz_out, uint32_t* n_out) {
  if (!s || !ids || !poses || !d_xyz_out || !n_out || (slot != 0 && slot != 1) || !(leaf > 0)) return QN_ERR_INVALID_ARG;
  *d_xyz_out = nullptr; *n_out = 0;
  KFCHK(s, hipSetDevice(s->device));
  size_t total = 0;
  for (uint32_t k = 0; k < count; k++) { if (ids[k] < 0 || (size_t)ids[k] >= s->clouds.size()) return QN_ERR_INVALID_ARG; total += s->sizes[ids[k]]; }
  if (total == 0) return QN_ERR_EMPTY_CLOUD;
  if (total >= 0xffffffffull) return QN_ERR_CAPACITY;
  int rc = kf_reserve(s, total); if (rc != QN_OK) return rc;
  if ((size_t)count * 16 > s->poses_cap) { (void)hipFree(s->poses); s->poses = nullptr; s->poses_cap = 0; KFCHK(s, hipMalloc(&s->poses, sizeof(double) * 16 * (count + 8))); s->poses_cap = (size_t)16 * (count + 8); }
  hipStream_t st = s->stream;
  KFCHK(s, hipMemcpyAsync(s->poses, poses, sizeof(double) * 16 * count, hipMemcpyHostToDevice, st));
  size_t off = 0;
  for (uint32_t k = 0; k < count; k++) {                       // transformPcd + operator+= (loop_closure.cpp:76,83,89,92,102)
    const uint32_t n = s->sizes[ids[k]];
    if (n) hipLaunchKernelGGL(qn::k_kf_transform, dim3((n + 255) / 256), dim3(256), 0, st, s->clouds[ids[k]], n, s->poses + 16 * k, s->concat + off);
    off += n;
  }
  uint32_t n = (uint32_t)total;
  // pcl::VoxelGrid::applyFilter: bounds, leaf indices
  qn::BBoxOut init; for (int d = 0; d < 3; d++) { init.mn[d] = 0x7fffffff; init.mx[d] = (int)0x80000000; } init.nonfinite = 0;
  *s->bbox_host = init;
  KFCHK(s, hipMemcpyAsync(s->bbox, s->bbox_host, sizeof(qn::BBoxOut), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(qn::k_bbox, dim3(std::min<uint32_t>((n + QN_BLOCK - 1) / QN_BLOCK, 128)), dim3(QN_BLOCK), 0, st, s->concat, n, s->bbox);
  KFCHK(s, hipMemcpyAsync(s->bbox_host, s->bbox, sizeof(qn::BBoxOut), hipMemcpyDeviceToHost, st));
  KFCHK(s, hipStreamSynchronize(st));
  if (s->bbox_host->nonfinite) {                                  // rare path: drop the non-finite points like pcl::VoxelGrid does for a non-dense cloud (order of the others kept)
    const uint32_t nbf = (n + 255) / 256, sbf = (n + QN_BLOCK * QN_SCAN_ITEMS - 1) / (QN_BLOCK * QN_SCAN_ITEMS);
    hipLaunchKernelGGL(qn::k_finite_flags, dim3(nbf), dim3(256), 0, st, s->concat, n, s->flag);
    hipLaunchKernelGGL(qn::k_scan_block, dim3(sbf), dim3(QN_BLOCK), 0, st, s->flag, n, s->pos, s->sums);
    hipLaunchKernelGGL(qn::k_scan_top, dim3(1), dim3(QN_BLOCK), 0, st, s->sums, sbf);
    hipLaunchKernelGGL(qn::k_scan_add_total, dim3(sbf), dim3(QN_BLOCK), 0, st, s->pos, n, s->sums, s->flag);
    if (n > s->out_cap[slot]) { (void)hipFree(s->out[slot]); s->out[slot] = nullptr; s->out_cap[slot] = 0; KFCHK(s, hipMalloc(&s->out[slot], sizeof(float4) * (n + n / 2))); s->out_cap[slot] = n + n / 2; }
    hipLaunchKernelGGL(qn::k_compact_finite, dim3(nbf), dim3(256), 0, st, (const float4*)s->concat, n, (const uint32_t*)s->flag, (const uint32_t*)s->pos, s->out[slot]);      // (the output slot as scratch)
    uint32_t kept = 0;
    KFCHK(s, hipMemcpyAsync(&kept, s->pos + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    KFCHK(s, hipStreamSynchronize(st));
    if (kept == 0) return QN_ERR_EMPTY_CLOUD;
    KFCHK(s, hipMemcpyAsync(s->concat, s->out[slot], sizeof(float4) * kept, hipMemcpyDeviceToDevice, st));
    n = kept;
    s->last_error = "note: non-finite points dropped (pcl::VoxelGrid on a non-dense cloud)";
  }
  qn::VoxelDims vd; vd.inv = 1.0f / (float)leaf;
  long long cells = 1; int divb[3];
  for (int d = 0; d < 3; d++) {
    const float mn = qn::ord2f(s->bbox_host->mn[d]), mx = qn::ord2f(s->bbox_host->mx[d]);
    vd.minb[d] = (int)std::floor(mn * vd.inv); const int maxb = (int)std::floor(mx * vd.inv);
    divb[d] = maxb - vd.minb[d] + 1; cells *= divb[d];
  }
  {  // pcl::VoxelGrid::applyFilter's overflow guard in PCL's own arithmetic (f32 product, int64 cast): it warns and sets output = *input_
    long long pd = 1;
    for (int d = 0; d < 3; d++) { const float mn = qn::ord2f(s->bbox_host->mn[d]), mx = qn::ord2f(s->bbox_host->mx[d]); pd *= (long long)((mx - mn) * vd.inv) + 1; }
    if (pd > (long long)INT32_MAX || cells > (long long)INT32_MAX) {
      s->last_error = "warning: leaf size is too small for the input dataset, integer indices would overflow: cloud passed through unfiltered (as pcl::VoxelGrid does)";
      if (n > s->out_cap[slot]) { (void)hipFree(s->out[slot]); s->out[slot] = nullptr; s->out_cap[slot] = 0; KFCHK(s, hipMalloc(&s->out[slot], sizeof(float4) * (n + n / 2))); s->out_cap[slot] = n + n / 2; }
      KFCHK(s, hipMemcpyAsync(s->out[slot], s->concat, sizeof(float4) * n, hipMemcpyDeviceToDevice, st));
      KFCHK(s, hipStreamSynchronize(st));
      s->out_n[slot] = n; *d_xyz_out = (const float*)s->out[slot]; *n_out = n;
      return QN_OK;
    }
  }
  vd.div0 = divb[0]; vd.div01 = divb[0] * divb[1];
  const uint32_t nb = (n + 255) / 256;
  hipLaunchKernelGGL(qn::k_voxel_keys, dim3(nb), dim3(256), 0, st, s->concat, n, vd, s->keys);
  unsigned long long* sorted = s->keys; unsigned long long* other = s->keys_alt;
  int bits = 1; while ((1ll << bits) < cells) bits++;
  const uint32_t rb = (n + QN_BLOCK - 1) / QN_BLOCK, hn = rb * 256, hsb = (hn + QN_BLOCK * QN_SCAN_ITEMS - 1) / (QN_BLOCK * QN_SCAN_ITEMS);
  for (int shift = 32; shift < 32 + bits; shift += 8) {                  // stable LSD passes over the leaf-index bits only
    hipLaunchKernelGGL(qn::k_radix_hist, dim3(rb), dim3(QN_BLOCK), 0, st, sorted, n, shift, rb, s->hist);
    hipLaunchKernelGGL(qn::k_scan_block, dim3(hsb), dim3(QN_BLOCK), 0, st, s->hist, hn, s->hist, s->hist_sums);
    hipLaunchKernelGGL(qn::k_scan_top, dim3(1), dim3(QN_BLOCK), 0, st, s->hist_sums, hsb);
    hipLaunchKernelGGL(qn::k_scan_add, dim3(hsb), dim3(QN_BLOCK), 0, st, s->hist, hn, s->hist_sums, n);
    hipLaunchKernelGGL(qn::k_radix_scatter, dim3(rb), dim3(QN_BLOCK), 0, st, sorted, n, shift, rb, s->hist, other);
    std::swap(sorted, other);
  }
  hipLaunchKernelGGL(qn::k_leaf_flags, dim3(nb), dim3(256), 0, st, sorted, n, s->flag);
  const uint32_t sb = (n + QN_BLOCK * QN_SCAN_ITEMS - 1) / (QN_BLOCK * QN_SCAN_ITEMS);
  hipLaunchKernelGGL(qn::k_scan_block, dim3(sb), dim3(QN_BLOCK), 0, st, s->flag, n, s->pos, s->sums);
  hipLaunchKernelGGL(qn::k_scan_top, dim3(1), dim3(QN_BLOCK), 0, st, s->sums, sb);
  hipLaunchKernelGGL(qn::k_scan_add_total, dim3(sb), dim3(QN_BLOCK), 0, st, s->pos, n, s->sums, s->flag);
  hipLaunchKernelGGL(qn::k_leaf_heads, dim3(nb), dim3(256), 0, st, s->flag, s->pos, n, s->heads);
  if (n > s->out_cap[slot]) { (void)hipFree(s->out[slot]); s->out[slot] = nullptr; s->out_cap[slot] = 0; KFCHK(s, hipMalloc(&s->out[slot], sizeof(float4) * (n + n / 2))); s->out_cap[slot] = n + n / 2; }
  hipLaunchKernelGGL(qn::k_leaf_centroids, dim3(nb), dim3(256), 0, st, s->concat, sorted, s->heads, s->pos + n, s->out[slot]);
  KFCHK(s, hipMemcpyAsync(s->count_host, s->pos + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  KFCHK(s, hipGetLastError());
  KFCHK(s, hipStreamSynchronize(st));
  s->out_n[slot] = *s->count_host;
  *d_xyz_out = (const float*)s->out[slot]; *n_out = s->out_n[slot];
  return QN_OK;
}

extern "C" int qn_kf_download(qn_kf_store* s, int slot, float* xyz_out) {        // packed n x 3, for tests / visualisation
  if (!s || !xyz_out || (slot != 0 && slot != 1)) return QN_ERR_INVALID_ARG;
  const uint32_t n = s->out_n[slot];
  if (!n) return QN_OK;
  KFCHK(s, hipSetDevice(s->device));
  KFCHK(s, hipMemcpy2D(xyz_out, 12, s->out[slot], 16, 12, n, hipMemcpyDeviceToHost));
  return QN_OK;
}

// LoopClosure::fetchClosestKeyframeIdx (loop_closure.cpp:34-56) generalised to the K best candidates (host code: O(#keyframes))
extern "C" int qn_loop_candidates(const double* pos_xyz, const double* stamps, uint32_t n, uint32_t query, double radius, double tdiff,
                                  uint32_t max_k, int32_t* out, uint32_t* n_out) {
  if (!pos_xyz || !stamps || !out || !n_out || query >= n) return QN_ERR_INVALID_ARG;
  std::vector<std::pair<double, int32_t>> c;
  for (uint32_t i = 0; i + 1 < n; i++) {                      // `keyframes.size() - 1`: the newest keyframe is the query itself
    const double dx = pos_xyz[3 * i] - pos_xyz[3 * query], dy = pos_xyz[3 * i + 1] - pos_xyz[3 * query + 1], dz = pos_xyz[3 * i + 2] - pos_xyz[3 * query + 2];
    const double d = std::sqrt(dx * dx + dy * dy + dz * dz);
    if (radius > d && tdiff < (stamps[query] - stamps[i])) c.emplace_back(d, (int32_t)i);
  }
  std::sort(c.begin(), c.end());
  uint32_t m = 0;
  for (const auto& e : c) { if (m >= max_k) break; out[m++] = e.second; }
  *n_out = m;
  return QN_OK;
}
