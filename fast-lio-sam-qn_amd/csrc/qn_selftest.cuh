// qn_selftest.cuh - device self-test of the wave-level primitives the search loops stand on (qn_device.cuh: DPP prefix sums and running maxima, DPP reductions with v_readlane,
// the slot -> segment map of a candidate chunk, the f64 wave sum).  Each is compared, lane by lane, with a plain restatement - serial loops over an LDS copy, the shuffle tree
// the f64 sum replaced - on seeded random data; qn_debug_selftest returns the number of mismatches (tests/test_gpu_selftest.py expects 0).  Test infrastructure: nothing on
// the registration path calls it.
#pragma once
#include "qn_device.cuh"

namespace qn {

// one wave per block; block b works on in[64 b .. 64 b + 63] (and in2[] for the low words of the 64-bit keys)
static __global__ void __launch_bounds__(64) k_selftest_wave(const uint32_t* __restrict__ in, const uint32_t* __restrict__ in2, uint32_t* __restrict__ bad) {
  __shared__ uint32_t sh[64], sh2[64], marks[64], excl[65];
  const int lane = threadIdx.x & 63;
  const uint32_t v = in[blockIdx.x * 64 + lane], w = in2[blockIdx.x * 64 + lane];
  sh[lane] = v; sh2[lane] = w;
  __syncthreads();
  uint32_t nbad = 0;
  // ---- prefix sum / running maximum
  { uint32_t rs = 0, rm = 0;
    for (int i = 0; i <= lane; i++) { rs += sh[i] & 0xffffu; rm = max(rm, sh[i]); }
    if (wave_incl_scan_u32(v & 0xffffu, lane) != rs) nbad++;
    if (wave_incl_max_u32(v) != rm) nbad++; }
  // ---- reductions (every lane must hold the wave's result)
  { int mn = 0x7fffffff, mx = (int)0x80000000; uint32_t mnu = 0xffffffffu; float fmn = __int_as_float(0x7f800000), fmx = -__int_as_float(0x7f800000); unsigned long long k = QN_INF_KEY;
    for (int i = 0; i < 64; i++) {
      const int s = (int)sh[i]; mn = min(mn, s); mx = max(mx, s); mnu = min(mnu, sh[i]);
      const float f = (float)(sh[i] & 0xfffffu) * 0.37f - 150000.f; fmn = fminf(fmn, f); fmx = fmaxf(fmx, f);
      const unsigned long long ki = ((unsigned long long)(sh[i] & 7u) << 32) | sh2[i];      // (three-bit high words: many ties - the low words decide)
      k = ki < k ? ki : k;
    }
    const float fv = (float)(v & 0xfffffu) * 0.37f - 150000.f;
    if (wave_min_i((int)v) != mn) nbad++;
    if (wave_max_i((int)v) != mx) nbad++;
    if (wave_min_u32(v) != mnu) nbad++;
    if (__float_as_uint(wave_min_f(fv)) != __float_as_uint(fmn)) nbad++;
    if (__float_as_uint(wave_max_f(fv)) != __float_as_uint(fmx)) nbad++;
    if (wave_min_u64(((unsigned long long)(v & 7u) << 32) | w) != k) nbad++; }
  // ---- f64 wave sum: the bits of the tree ((row sums by quad / row mirrors), then (R0 + R1) + (R2 + R3)) the shuffle form left in every lane
  { const double d = (double)(int)(v >> 7) * 1.0000001e-3 + (double)w * 3.3e-11;
    double t = dpp_add_f64(d, 0); t = dpp_add_f64(t, 1); t = dpp_add_f64(t, 2); t = dpp_add_f64(t, 3);
    t += __shfl_xor(t, 16); t += __shfl_xor(t, 32);
    if (__double_as_longlong(wave_sum_f64_dpp(d)) != __double_as_longlong(t)) nbad++; }
  // ---- slot -> segment of every chunk of a random segment table (lengths 0..15, a third of them empty)
  { const uint32_t len = (v % 3u == 0u) ? 0u : ((v >> 8) & 15u);
    uint32_t ex = 0; for (int i = 0; i < lane; i++) { const uint32_t li = (sh[i] % 3u == 0u) ? 0u : ((sh[i] >> 8) & 15u); ex += li; }
    excl[lane] = ex; if (lane == 63) excl[64] = ex + len;
    __syncthreads();
    const uint32_t total = excl[64];
    for (uint32_t cb = 0; cb < total; cb += 64) {
      const int j = chunk_segment(marks, cb, ex, ex + len);
      const uint32_t slot = cb + (uint32_t)lane;
      if (slot < total) { int r = 0; for (int i = 0; i < 64; i++) if (excl[i] <= slot && excl[i + 1] > slot) r = i; if (j != r) nbad++; }
    } }
  if (nbad) atomicAdd(bad, nbad);
}

}  // namespace qn
