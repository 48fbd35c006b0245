// qn_knn_kernels.cuh - general sorted-list k-NN kernel (register-resident BestK<KMAX> sink) of calculateSource/
// TargetCovariances (loop_closure.cpp:121,123).  Its own header because its instantiations are the
// slowest units to compile (unrolled BestK merges): the units that instantiate it (qn_instances.h groups 2-9) depend on nothing but this
// file and qn_device.cuh, so edits to the align kernels do not rebuild them.
#pragma once
#include "qn_device.cuh"
#ifndef QN_BLOCK
#define QN_BLOCK 256
#endif

namespace qn {

// k-NN selection and covariance are separate kernels: the selection keeps a 2k-register list alive, the
// covariance needs ~40 f64 registers for the Jacobi sweep - fused, the kernel sat at 2 waves/SIMD.
template <int KMAX>
__device__ __forceinline__ void store_knn(const BestK<KMAX>& sink, int32_t* __restrict__ knn_idx, float* __restrict__ knn_d2) {
  const int k = sink.k;
#pragma unroll
  for (int j = 0; j < KMAX; j++) if (j >= KMAX - k) {
    const bool ok = sink.a[j] != QN_INF_KEY;
    knn_idx[j - (KMAX - k)] = ok ? (int32_t)key_idx(sink.a[j]) : -1;
    if (knn_d2) knn_d2[j - (KMAX - k)] = ok ? key_d2(sink.a[j]) : 0.f;
  }
}

// One kernel body for both passes.  LIST = false: query t = global query slot, radius margin * cell, two
// rounds, leftovers appended to fb_list with the radius to continue from.  LIST = true: the queries are
// the fb_list entries of the first pass (16 per wave, wave-stride), rounds until exact.
struct KnnCovArgs { GridView g; const float4* raw; int k; float r0; int max_rounds; double* cov; int32_t* knn_idx; float* knn_d2; uint2* fb_list; uint32_t* fb_count; };
template <int KMAX, bool LIST, int S>
struct KnnCovK {
  static constexpr int TB = QN_BLOCK, OCC = 3;
  using Args = KnnCovArgs;
  static __device__ __forceinline__ void run(const Args& a, const uint32_t bx, const uint32_t nbx) {
    const GridView g = grid_resolve(a.g);
    const int k = a.k, max_rounds = a.max_rounds; float r0 = a.r0;
    int32_t* __restrict__ knn_idx = a.knn_idx; float* __restrict__ knn_d2 = a.knn_d2; uint2* __restrict__ fb_list = a.fb_list; uint32_t* __restrict__ fb_count = a.fb_count;

  __shared__ WaveLdsK lds[QN_BLOCK / 64];
  if (r0 < 0.f) r0 = -r0 * g.cell;              // (a negative radius is in cells)
  WaveLdsK* my = &lds[threadIdx.x >> 6];
  const uint32_t nq = LIST ? uni(*fb_count) : g.n;
  if (LIST && g.dbg && bx == 0 && threadIdx.x == 0) atomicAdd(&g.dbg[4], nq);
  const uint32_t wave0 = bx * (QN_BLOCK / 64) + (threadIdx.x >> 6), nwaves = nbx * (QN_BLOCK / 64);
  constexpr uint32_t QPW = 64 / S;                                          // queries per wave
  for (uint32_t base = wave0 * QPW; base < nq; base += nwaves * QPW) {      // (non-LIST grids cover nq in one trip)
    const uint32_t slot = base + (threadIdx.x & (QPW - 1));
    const bool active = slot < nq;
    uint32_t t = slot; float r = r0;
    if (LIST && active) { const uint2 rec = fb_list[slot]; t = rec.x; r = __uint_as_float(rec.y); }   // continue from r
    const float4 q = active ? g.pts[t] : make_float4(0, 0, 0, 0);
    BestK<KMAX> sink; sink.init(k, my->pend, g.dbg);
    float d_unseen;
    const bool cert = wave_search<S>(g, q.x, q.y, q.z, active, r, __int_as_float(0x7f800000), max_rounds, sink, &my->s, d_unseen);
    if (!active || (threadIdx.x & 63) >= QPW) continue;             // sub-slot 0 of each query finishes the job
    const uint32_t i = __float_as_uint(q.w);
    if (cert || LIST) store_knn(sink, knn_idx + (size_t)i * k, knn_d2 ? knn_d2 + (size_t)i * k : nullptr);
    else {
      const uint32_t fs = atomicAdd(fb_count, 1u);
      fb_list[fs] = make_uint2(t, __float_as_uint(r));
    }
  }
  }
};
template <int KMAX, bool LIST, int S>
__global__ void __launch_bounds__(QN_BLOCK, 3) k_knn_cov(GridView g, const float4* __restrict__ raw, int k, float r0, int max_rounds,
                                                      double* __restrict__ cov, int32_t* __restrict__ knn_idx, float* __restrict__ knn_d2,
                                                      uint2* __restrict__ fb_list, uint32_t* __restrict__ fb_count) {
  const KnnCovArgs a{g, raw, k, r0, max_rounds, cov, knn_idx, knn_d2, fb_list, fb_count};
  KnnCovK<KMAX, LIST, S>::run(a, blockIdx.x, gridDim.x);
}

}  // namespace qn
