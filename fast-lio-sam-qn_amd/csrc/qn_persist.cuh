// qn_persist.cuh - the tracked regime of NanoGICP::align() (call site fast_lio_sam_qn/src/loop_closure.cpp:124; LsqRegistration loop restated in
// SURVEY.md A.1.4-A.1.5) as ONE persistent launch, for the registration the reference actually runs: one candidate pair per 2 Hz timer
// tick (fast_lio_sam_qn.cpp:213-219), nothing else on the GPU.
//
// The chain of k_tick launches (qn_tick.cuh) pays per optimiser tick: a kernel boundary, every block re-reading every block's partial row
// (196 x 196 x 224 B = 8.6 MB per tick at 100k points, more than the tick's whole algorithmic traffic) and the f64 controller run redundantly
// in 196 prologues.  Here the nblk worker blocks stay resident with their source points, normals and tracking records in REGISTERS, and one more
// block - the reducer - does nothing but the controller:
//
//   worker, tick g:   poll the pose granules tagged g + 1  ->  tick_point (tracked exact 1-NN + the 28 sums; LM trial passes alike)
//                     ->  publish the block's row: 28 f64 into buffer (g + 1) % 3, whose slots read "not arrived" (a signalling-NaN sentinel) until then
//   reducer, tick g:  spin on the nblk rows of buffer g % 3 until no slot holds the sentinel (tick 0: nothing to gather - the state the chain left is already stepped), put the
//                     previous buffer back to "not arrived"  ->  fixed-order sum,
//                     LM / GN controller (solve_controller: the same code, the same order of additions as k_tick's prologue, so both paths
//                     give the same bits)  ->  publish x0, xi, phase as granules tagged g + 1
//   done (phase 2):   workers run the closing pass (fitness sweep + output cloud) and publish (sum, count); the reducer folds them like
//                     k_finalize_fit and writes the result block (pinned host memory) and the final state.
//
// Hand-offs follow the guide's R2 form (cdna_hip_programming.md, Guideline 16): the data IS the flag - 8-byte {epoch, 32-bit value} granules, each
// written by ONE relaxed agent-scope (sc1, write-through) store and read by relaxed agent-scope loads; an f64 is two granules.  No fences, no
// counters, no dependence on where a block runs; the per-XCD L2s are never asked to be coherent.  Epochs are unique per launch and per tick
// (the host advances epoch0 by more than a launch can use), so nothing needs zeroing between launches.  Every spin is bounded (give-up code in
// `status`, the host then fails the align loudly instead of hanging the GPU).
#pragma once
#include "qn_tick.cuh"

namespace qn {

#define QN_PERSIST_TB 512
#define QN_PERSIST_MAX_BLOCKS 240          // worker blocks (+ 1 reducer): every block needs a CU of its own (256 on the chip)
#define QN_PERSIST_BC 49                   // pose granules: x0[12] and xi[12] (two each), phase
#define QN_PERSIST_ROWS (QN_PERSIST_MAX_BLOCKS + 1)
#define QN_PERSIST_RSTRIDE 32              // doubles per row slot: 28 sums + 4 pad = two 128-byte lines, written by ONE store instruction (16 lanes x 16 B)
#define QN_PERSIST_SENTINEL 0xFFFFFFFFFFFFFFFFull      // "row value not arrived": a NaN pattern no f64 sum of finite terms can produce (arithmetic yields the canonical quiet NaN) - and all bytes equal, so hipMemset can write it

struct PersistArgs {
  TickArgs t;                              // t.tail.st_in: the state the unseeded ticks left (already stepped by their controller tail: nothing pending); t.tail.st_out: the final state
  unsigned long long* rows_g;              // [3][QN_PERSIST_ROWS][32] partial rows as raw f64 bits, buffer = tick % 3; a slot holds QN_PERSIST_SENTINEL until its value lands
  unsigned long long* bc_g;                // [64] pose granules
  unsigned long long* fit_g;               // [nblk][4] closing pass: sum lo, sum hi, count
  uint32_t* status;                        // [0] give-up code (0 = none), [1] ticks run
  ResultBlock* result;                     // pinned host memory
  uint32_t nblk, epoch0, max_ticks;
  unsigned long long timeout;              // wall_clock64 units (100 MHz) a spin may last
  uint32_t* status_host;                   // pinned mirror of `status`, written by the reducer when it leaves (no memset in front of, no copy behind the launch)
  int cond;                                // != 0: launched behind look_decide without a host look - go ahead only if the state's QN_LOOK_GO flag is set
  unsigned long long* clk;                 // developer probe (PROBE = true): [tick < 64][16] wall-clock stamps: 0 rows complete, 1 sums, 2 controller, 3 pose published (reducer);
                                           // 4 pose seen, 5 body done, 6 row published (worker block 0); 8..10 the same for the last worker block; [64 * 16] = launch start
};

__device__ __forceinline__ void pg_store(unsigned long long* p, uint32_t epoch, uint32_t v) {
  __hip_atomic_store((qn_gu64*)p, ((unsigned long long)epoch << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long pg_load(const unsigned long long* p) {
  return __hip_atomic_load((qn_gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (pr_store / pr_load: qn_gicp_kernels.cuh - the controller tail of the chain kernels hands its rows over the same way)
// 16-byte write-through store (agent scope): a row leaves its block as ONE instruction of 16 lanes (8-byte sc1 stores are one fabric write each and took
// 4 us to be acknowledged with 196 blocks storing 28 of them at once).  Each 8-byte half is read on its own by the reducer: no tearing inside a half.
__device__ __forceinline__ void pr_store16(unsigned long long* p, unsigned long long lo, unsigned long long hi) {
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const u64x2 v = {lo, hi};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void pg_store_f64(unsigned long long* p, uint32_t epoch, double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  pg_store(p, epoch, (uint32_t)b); pg_store(p + 1, epoch, (uint32_t)(b >> 32));
}
__device__ __forceinline__ bool pg_expired(unsigned long long t0, unsigned long long timeout, uint32_t* status, uint32_t code) {
  if (wall_clock64() - t0 <= timeout) return false;
  atomicCAS(status, 0u, code);
  return true;
}

template <int TB, bool PROBE>
__global__ void __launch_bounds__(TB, TB / 256) k_align_persist(PersistArgs A) {
  __shared__ WaveScratch sc[TB / 64];
  __shared__ double wsum[TB / 64][QN_NPART];
  __shared__ GicpState sh;                                           // reducer: the optimiser state; workers use bc below
  __shared__ double part8[QN_NPART][QN_ROW_SEGS + 1];
  static_assert(TB / QN_NPART == QN_ROW_SEGS, "the reducer's thread -> (segment, component) map is reduce_rows' at 512 threads");
  __shared__ double sums[QN_NPART];
  __shared__ SolveWork Awork_s;
  __shared__ double bc_x0[16], bc_xi[16], bc_G[9];
  __shared__ int bc_phase, bc_fail;
  __shared__ unsigned long long tie_list[TB / 64][QN_HCAP1]; __shared__ uint32_t tie_cnt[TB / 64];
  A.t.src = grid_resolve(A.t.src); A.t.tgt = grid_resolve(A.t.tgt);
  if (A.cond) {                                                      // behind look_decide: its flags decide (uniform over the launch: the state is not written before the reducer's last step)
    const int flags = A.t.tail.st_in->reserved;
    if (!(flags & QN_LOOK_GO)) { if (blockIdx.x == A.nblk && threadIdx.x == 0) { A.status_host[0] = 5u; A.status_host[1] = 0u; } return; }      // declined: nothing touched
  }
  const TickArgs& a = A.t;
  const int tid = threadIdx.x, lane = tid & 63;
  const uint32_t nblk = A.nblk;
  const unsigned long long t_start = wall_clock64();
  if (PROBE && blockIdx.x == 0 && tid == 0) A.clk[64 * 16] = t_start;

  if (blockIdx.x == nblk) {
    // ------------------------------------------------------------------ the reducer block
    constexpr int SEGS = TB / QN_NPART;                              // = reduce_rows' partition (QN_ROW_SEGS at TB = 512): thread (s, c) sums rows s, s + SEGS, ...
    constexpr int UMAX = (QN_PERSIST_MAX_BLOCKS + SEGS - 1) / SEGS;
    const int rs = tid / QN_NPART, rc = tid - rs * QN_NPART;
    const bool rthread = tid < QN_NPART * SEGS;
    for (int i = tid; i < (int)(sizeof(GicpState) / 8); i += TB) ((unsigned long long*)&sh)[i] = ((const unsigned long long*)a.tail.st_in)[i];
    if (tid == 0) { bc_fail = 0; bc_phase = 0; }
    // (all three row buffers read "not arrived" at this point: the context memsets them once, and every launch puts back what it dirtied - below)
    __syncthreads();
    uint32_t g = 0;
    for (;; g++) {
      const unsigned long long t_tick = wall_clock64();                // (the time-out bounds ONE wait, not the launch: a long legitimate run - LM with many rejected trials - is not a hang)
      if (g > 0) {                                                     // (tick 0: the state the chain left is already stepped - its rows were consumed by the controller tail of the launch that wrote them)
        // the rows the workers wrote under pose g: buffer g % 3.  The data IS the flag: a slot holds the sentinel until its value lands.  Every thread
        // spins on its own slots (no block barrier per pass); arrived values stay in registers, only the missing ones are asked for again.
        unsigned long long* buf = A.rows_g + (size_t)(g % 3u) * QN_PERSIST_ROWS * QN_PERSIST_RSTRIDE;
        double v[UMAX]; uint32_t need = 0;
        if (rthread) {
#pragma unroll
          for (int u = 0; u < UMAX; u++) { v[u] = 0.0; if ((uint32_t)(rs + SEGS * u) < nblk) need |= 1u << u; }
        }
        // (a light first phase - one watched slot per row line before the full fetch - measured slower: 14.4 vs 13.4 us per tick; removed)
        const uint32_t need0 = need; bool seen_first = false;
        for (uint32_t spins = 0; need != 0; spins++) {
          if (PROBE && tid == 0 && g < 64) { A.clk[16 * g + 14] = spins + 1; if (!seen_first && need != need0) { seen_first = true; A.clk[16 * g + 15] = wall_clock64(); } }
          unsigned long long x[UMAX];
#pragma unroll
          for (int u = 0; u < UMAX; u++) if ((need >> u) & 1u) x[u] = pr_load(buf + (size_t)(rs + SEGS * u) * QN_PERSIST_RSTRIDE + rc);
#pragma unroll
          for (int u = 0; u < UMAX; u++) if (((need >> u) & 1u) && x[u] != QN_PERSIST_SENTINEL) { v[u] = __longlong_as_double((long long)x[u]); need &= ~(1u << u); }
          if ((spins & 255u) == 255u && pg_expired(t_tick, A.timeout, A.status, 1u)) { bc_fail = 1; break; }
        }
        if (PROBE && tid == 0 && g < 64) A.clk[16 * g + 7] = wall_clock64();
        // consumed: the buffer of the PREVIOUS tick goes back to "not arrived" (it is written again under pose g + 2, which this block publishes only after
        // the barrier at the top of the next iteration, i.e. after every wave has drained these stores)
        if (rthread && g >= 1) {
          unsigned long long* old = A.rows_g + (size_t)((g - 1u) % 3u) * QN_PERSIST_ROWS * QN_PERSIST_RSTRIDE;
#pragma unroll
          for (int u = 0; u < UMAX; u++) if ((uint32_t)(rs + SEGS * u) < nblk) pr_store(old + (size_t)(rs + SEGS * u) * QN_PERSIST_RSTRIDE + rc, QN_PERSIST_SENTINEL);
        }
        if (rthread) { double acc = 0;
#pragma unroll
          for (int u = 0; u < UMAX; u++) acc += v[u];                // rows s, s + SEGS, ... in order; the rows beyond nblk add +0.0 like reduce_rows
          part8[rc][rs] = acc; }
        __syncthreads();
        if (PROBE && tid == 0 && g < 64) A.clk[16 * g + 0] = wall_clock64();
        if (tid < QN_NPART) { double w = 0;
#pragma unroll
          for (int s2 = 0; s2 < SEGS; s2++) w += part8[tid][s2]; sums[tid] = w; }
        wave_lds_fence();                                            // (sums[] is written and read by wave 0 only)
      }
      if (tid < 64) {                                                // wave 0: the controller on lane 0, then the pose straight out - no block barrier in front of the publication
        if (tid == 0 && !bc_fail) {
          if (PROBE && g < 64) A.clk[16 * g + 1] = wall_clock64();
          const int phase_in = sh.phase;
          if (sh.pending && phase_in != 2 && g > 0) solve_controller(&sh, sums, a.tail.cfg, a.tail.trace, 0, phase_in, &Awork_s);
          sh.fb_count = 0; sh.big_count = 0; sh.pending = sh.phase != 2 ? 1 : 0;
          if (g + 1 >= A.max_ticks && sh.phase != 2) { atomicCAS(A.status, 0u, 2u); bc_fail = 1; }
          bc_phase = sh.phase;
          if (PROBE && g < 64) A.clk[16 * g + 2] = wall_clock64();
        }
        wave_lds_fence();
        const uint32_t epn = A.epoch0 + g + 1;                       // tag of the pose = tag of nothing else: rows carry no tag
        if (tid < 24) pg_store(A.bc_g + tid, epn, (uint32_t)((unsigned long long)__double_as_longlong(sh.x0[tid >> 1]) >> ((tid & 1) * 32)));
        else if (tid < 48) pg_store(A.bc_g + tid, epn, (uint32_t)((unsigned long long)__double_as_longlong(sh.xi[(tid - 24) >> 1]) >> ((tid & 1) * 32)));
        else if (tid == 48) pg_store(A.bc_g + 48, epn, bc_fail ? 3u : (uint32_t)sh.phase);   // (phase 3: give up, leave)
        if (PROBE && tid == 0 && g < 64) A.clk[16 * g + 3] = wall_clock64();
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's slot resets are in memory before the barrier that precedes the next publication
      __syncthreads();
      if (bc_fail || bc_phase == 2) break;
      __builtin_amdgcn_s_sleep(100);                                 // ~3 us: no row can land before the workers have seen the pose and run the body; polling meanwhile only loads the memory system
    }
    if (bc_fail) {
      if (tid == 0) { A.status_host[0] = __hip_atomic_load(A.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); A.status_host[1] = g; A.result->phase = -1;
        __hip_atomic_store(A.status, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }      // (a worker that gives up later may set it again: the host clears the word after a failed launch)
      return;
    }
    // the buffer gathered last goes back to "not arrived" for the next launch (the one before it was reset in the last iteration, the third was never written)
    if (rthread && g >= 1) {
      unsigned long long* last = A.rows_g + (size_t)(g % 3u) * QN_PERSIST_ROWS * QN_PERSIST_RSTRIDE;
#pragma unroll
      for (int u = 0; u < UMAX; u++) if ((uint32_t)(rs + SEGS * u) < nblk) pr_store(last + (size_t)(rs + SEGS * u) * QN_PERSIST_RSTRIDE + rc, QN_PERSIST_SENTINEL);
    }
    // ---- closing: fold the workers' (sum, count) like k_finalize_fit (lane L adds blocks 8 L .. 8 L + 7, then the wave sum), write the result and the state
    const uint32_t epf = A.epoch0 + g + 1;
    const unsigned long long t_close = wall_clock64();
    if (tid < 64) {
      double s = 0; uint32_t cn = 0;
      unsigned long long done = 0;
      double pv[8]; uint32_t pc[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { pv[u] = 0; pc[u] = 0; if ((uint32_t)(lane * 8 + u) >= nblk) done |= 1ull << u; }
      bool fail = false;
      for (;;) {
#pragma unroll
        for (int u = 0; u < 8; u++) if (!((done >> u) & 1ull)) {
          const unsigned long long* q = A.fit_g + (size_t)(lane * 8 + u) * 4;
          const unsigned long long lo = pg_load(q), hi = pg_load(q + 1), cc = pg_load(q + 2);
          if ((uint32_t)(lo >> 32) == epf && (uint32_t)(hi >> 32) == epf && (uint32_t)(cc >> 32) == epf) {
            pv[u] = __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffull))); pc[u] = (uint32_t)cc; done |= 1ull << u; }
        }
        if (__all(done == 0xffull)) break;
        if (__any(lane == 0 && pg_expired(t_close, A.timeout, A.status, 3u))) { fail = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int u = 0; u < 8; u++) { s += pv[u]; cn += pc[u]; }
      s = wave_sum_f64_dpp(s);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) cn += __shfl_xor(cn, o);
      if (lane == 0) {
        ResultBlock* out = A.result;
        for (int i = 0; i < 16; i++) { out->r.T64[i] = sh.x0[i]; out->r.T[i] = (float)sh.x0[i]; }
        for (int i = 0; i < 36; i++) out->r.H[i] = sh.final_H[i];
        out->r.fitness = cn > 0 ? s / cn : 1.7976931348623157e308;
        out->r.iterations = sh.outer; out->r.converged = sh.converged; out->r.lm_failed = sh.lm_failed; out->r.reserved = 0;
        out->trace_len = sh.trace_len;
        // (the workers' counters are agent-scope atomics: read and reset them the same way, not through this XCD's L2)
        out->far_requests = a.far_stats ? __hip_atomic_load(a.far_stats + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        out->far_misses = a.far_stats ? __hip_atomic_load(a.far_stats + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        out->far_queries = a.far_stats ? __hip_atomic_load(a.far_stats + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        if (a.far_stats) { for (int i = 0; i < 4; i++) if (i != 2) __hip_atomic_store(a.far_stats + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        out->phase = fail ? -1 : sh.phase;
        A.status_host[0] = fail ? __hip_atomic_load(A.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u; A.status_host[1] = g + 1;
        if (fail) __hip_atomic_store(A.status, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
    for (int i = tid; i < (int)(sizeof(GicpState) / 8); i += TB) ((unsigned long long*)a.tail.st_out)[i] = ((const unsigned long long*)&sh)[i];
    return;
  }

  // ------------------------------------------------------------------ a worker block
  const uint32_t lblk = xcd_block(blockIdx.x, nblk);                 // the same block -> points map and row order as k_tick
  uint32_t t = (lblk * a.ppt) * TB + tid;
  bool valid = t < a.src.n;
  float4 p = valid ? a.src.pts[t] : make_float4(0, 0, 0, 0);
  int32_t j0s = valid ? a.nn_idx[t] : -1;
  float4 ref = valid ? a.nn_ref[t] : make_float4(0, 0, 0, 0);
  double na[3] = {0, 0, 0};
  if (valid) { na[0] = a.nrm_s[(size_t)t * 3]; na[1] = a.nrm_s[(size_t)t * 3 + 1]; na[2] = a.nrm_s[(size_t)t * 3 + 2]; }
  TargetRec rec0; rec0.p = make_float4(0, 0, 0, 0); rec0.n[0] = rec0.n[1] = rec0.n[2] = 0;
  if (valid && (uint32_t)j0s < a.tgt.n) { const TargetRec* r = a.tgt_rec + j0s; rec0.p = r->p; rec0.n[0] = r->n[0]; rec0.n[1] = r->n[1]; rec0.n[2] = r->n[2]; }
  Top2 t2; t2.j1 = -1; t2.p1 = make_float4(0, 0, 0, 0);             // the runner-up (qn_tick.cuh): in registers, never in memory
  if (tid == 0) bc_fail = 0;
  for (uint32_t g = 0;; g++) {
    const uint32_t ep = A.epoch0 + g + 1;
    __syncthreads();                                                 // (the previous tick's readers of bc_* and wsum are done)
    if (tid < 64) {                                                  // ONE wave polls: lanes 0..48 hold one granule each
      unsigned long long x = 0; bool fail = false; uint32_t spins = 0;
      const unsigned long long t_wait = wall_clock64();
      for (;;) {
        if (lane < QN_PERSIST_BC) x = pg_load(A.bc_g + lane);
        if (__all(lane >= QN_PERSIST_BC || (uint32_t)(x >> 32) == ep)) break;
        if ((++spins & 255u) == 0u && __any(lane == 0 && pg_expired(t_wait, A.timeout, A.status, 4u))) { fail = true; break; }
      }
      const uint32_t lo = (uint32_t)x, hi = (uint32_t)__shfl_down(x, 1);   // lane 2 i: low word of value i, lane 2 i + 1: its high word
      if (lane < 48 && !(lane & 1)) { const double d = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); if (lane < 24) bc_x0[lane >> 1] = d; else bc_xi[(lane - 24) >> 1] = d; }
      if (lane == 48) bc_phase = (int)lo;
      if (lane == 0 && fail) bc_fail = 1;
      wave_lds_fence();
      pose_gram(bc_x0, bc_G, lane);                                   // R R^T of the new pose (emit_point reads the poses and this from LDS)
    }
    __syncthreads();
    if (bc_fail || bc_phase == 3) return;
    const int pslot = blockIdx.x == 0 ? 4 : (blockIdx.x == nblk - 1 ? 8 : -1);
    if (PROBE && tid == 0 && pslot >= 0 && g < 64) A.clk[16 * g + pslot] = wall_clock64();
    const int phase = bc_phase;
    float Tf[12];
#pragma unroll
    for (int j = 0; j < 12; j++) Tf[j] = (float)bc_x0[j];
    if (phase == 2) {                                                // closing pass: getFitnessScore sweep + output cloud, (sum, count) to the reducer
      for (uint32_t it = 0; it < a.ppt; it++) {
        if (it > 0 || a.ppt > 1) {
          t = (lblk * a.ppt + it) * TB + tid; valid = t < a.src.n;
          p = valid ? a.src.pts[t] : make_float4(0, 0, 0, 0); j0s = valid ? a.nn_idx[t] : -1; ref = valid ? a.nn_ref[t] : make_float4(0, 0, 0, 0); t2.j1 = -1;
          if (valid && (uint32_t)j0s < a.tgt.n) { const TargetRec* r = a.tgt_rec + j0s; rec0.p = r->p; rec0.n[0] = r->n[0]; rec0.n[1] = r->n[1]; rec0.n[2] = r->n[2]; }
        }
        tick_point<1, false, true>(a, Tf, bc_x0, bc_xi, bc_G, true, it == 0, t, valid, p, j0s, ref, na, rec0, t2, &sc[tid >> 6].w, sc[tid >> 6].red, wsum[tid >> 6], tie_list[tid >> 6], &tie_cnt[tid >> 6], false);
      }
      __syncthreads();
      if (tid == 0) { double sv = 0, cv = 0; for (int w = 0; w < TB / 64; w++) { sv += wsum[w][0]; cv += wsum[w][1]; }
        unsigned long long* q = A.fit_g + (size_t)lblk * 4; pg_store_f64(q, ep, sv); pg_store(q + 2, ep, (uint32_t)cv); }
      return;
    }
    const bool lin = phase == 0;
    for (uint32_t it = 0; it < a.ppt; it++) {
      if (a.ppt > 1) {                                               // (clouds beyond QN_PERSIST_MAX_BLOCKS x TB points: the tracking records live in memory)
        t = (lblk * a.ppt + it) * TB + tid; valid = t < a.src.n;
        p = valid ? a.src.pts[t] : make_float4(0, 0, 0, 0); j0s = valid ? a.nn_idx[t] : -1; ref = valid ? a.nn_ref[t] : make_float4(0, 0, 0, 0); t2.j1 = -1;
        if (valid) { na[0] = a.nrm_s[(size_t)t * 3]; na[1] = a.nrm_s[(size_t)t * 3 + 1]; na[2] = a.nrm_s[(size_t)t * 3 + 2]; }
        if (valid && (uint32_t)j0s < a.tgt.n) { const TargetRec* r = a.tgt_rec + j0s; rec0.p = r->p; rec0.n[0] = r->n[0]; rec0.n[1] = r->n[1]; rec0.n[2] = r->n[2]; }
      }
      tick_point<0, false, true>(a, Tf, bc_x0, bc_xi, bc_G, lin, it == 0, t, valid, p, j0s, ref, na, rec0, t2, &sc[tid >> 6].w, sc[tid >> 6].red, wsum[tid >> 6], tie_list[tid >> 6], &tie_cnt[tid >> 6], false);
    }
    __syncthreads();
    if (PROBE && tid == 0 && pslot >= 0 && g < 64) A.clk[16 * g + pslot + 1] = wall_clock64();
    if (PROBE && tid == 0 && g < 64) { const unsigned long long st = ((wall_clock64() - t_start) << 16) | blockIdx.x; atomicMax(&A.clk[16 * g + 11], st); atomicMin(&A.clk[16 * g + 12], st); }      // slowest / fastest block of the tick
    if (tid < 32) {                                                  // lanes 0..27: the block's 28 sums (the 8 wave rows in order); even lanes store two of them
      double v = 0;
      if (tid < QN_NPART) {
#pragma unroll
        for (int w = 0; w < TB / 64; w++) v += wsum[w][tid];
      }
      unsigned long long bits = (unsigned long long)__double_as_longlong(v); if (bits == QN_PERSIST_SENTINEL) bits ^= 1ull;      // (never the "not arrived" pattern)
      const unsigned long long hi = __shfl_down(bits, 1);
      if (!(tid & 1)) pr_store16(A.rows_g + ((size_t)((g + 1u) % 3u) * QN_PERSIST_ROWS + lblk) * QN_PERSIST_RSTRIDE + tid, bits, hi);
    }
    if (PROBE && tid == 0 && pslot >= 0 && g < 64) A.clk[16 * g + pslot + 2] = wall_clock64();
    if (PROBE && blockIdx.x == 0 && g < 64) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (tid == 0) A.clk[16 * g + 13] = wall_clock64(); }      // row stores acknowledged
  }
}

}  // namespace qn
