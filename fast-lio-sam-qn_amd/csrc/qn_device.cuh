// qn_device.cuh - device-side building blocks of the gfx950 registration engine.
//
// Voxel grid-hash exact nearest-neighbour search (replaces nano_gicp's nanoflann KD-tree and the
// PCL FLANN tree behind getFitnessScore; reference call sites loop_closure.cpp:120-127).
//
// Layout (all resident in HBM, owned by the context):
//   pts[n]        float4, sorted by TILE-MAJOR cell key: the grid is cut into tiles of 8 x 4 x 4
//                 cells; tiles are ordered x-fastest, and the 128 cells inside a tile are ordered
//                 x-fastest too.  .w carries the ORIGINAL point index as raw bits - ties between
//                 equal f32 distances resolve to the lowest original index, as in the CPU oracle.
//   cell_start[]  uint32 [ncells + 1], exclusive prefix of per-cell counts in key order.
// Two properties follow:
//   * 64 consecutive sorted points (one wavefront of queries) sit in one or two neighbouring tiles,
//     whether they lie on the ground, on a wall or on a box (compact clusters);
//   * the cells [xa..xb] of one (y, z) row inside one tile are ONE contiguous run of pts[] - a
//     "segment" - so candidates are fetched with coalesced 16 B/lane loads.
//
// Search, pass A (one query per lane).  Lanes are grouped into clusters around an anchor lane; the
// cluster's cell bounding box, grown by a margin, is decomposed into segments (one per lane), the
// segment lengths are prefix-summed across the wave and the candidates are pulled as ONE dense
// stream: slot s of the stream maps to (segment, offset) by a binary search over the prefix in
// LDS, so every 64-wide fetch is full and all fetches of a cluster are independent.  Each chunk is
// staged in a wave-private LDS tile (ds_write_b128) and scored by every lane of the cluster with
// broadcast ds_read_b128.  A lane's result is CERTIFIED exact when its (k-th) best distance is
// smaller than its distance to the nearest face of the scanned box that has unseen cells behind
// it; uncertified lanes retry with a larger margin, and whatever is left goes to pass B
// (exact ball queries: wave_ball_nn1 / lane_ball_knn).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qn {

#define QN_TX 8
#define QN_TY 4
#define QN_TZ 4
#define QN_TILE_CELLS 128

// The numbers of a grid are computed ON THE DEVICE (k_grid_dims: bounding box -> cell edge -> dimensions) so that setInputSource / setInputTarget need no host round
// trip; kernels receive a GridView whose pointers the host filled in and whose numbers they fetch themselves (grid_resolve: one uniform 64-byte load).
struct GridDims {
  float ox, oy, oz, cell, inv_cell, eps;
  int nx, ny, nz, ntx, nty, ntz;
  uint32_t n;                          // points in the grid (0 when the cloud held non-finite coordinates: every kernel then sees an empty cloud)
  uint32_t ncells, nonfinite, pad;
};
struct GridView {
  const float4* pts;
  const uint32_t* cell_start;
  uint32_t* dbg;                       // optional counters (null in production): clusters, candidates, flushes, retries
  float ox, oy, oz, cell, inv_cell, eps;
  int nx, ny, nz;                      // cells per axis
  int ntx, nty, ntz;                   // tiles per axis
  uint32_t n;
  const GridDims* dims;                // device-resident numbers (never null: grid_resolve reads them)
};
// Values that are the same in every lane but were fetched through a pointer the compiler knows nothing about (a pointer that itself came out of a device-resident
// argument table: k_lanes) land in VECTOR registers; these put them back into scalar registers.  On a value that already lives in an SGPR they fold away.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ double uni(double v) { union { double d; int i[2]; } u; u.d = v; u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]); u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]); return u.d; }
__device__ __forceinline__ GridView grid_resolve(GridView g) {      // (dims is always set: build_grid.  Unconditional on purpose - a conditional copy became a select of POINTERS
  const GridDims d = *g.dims;                                       //  between the device numbers and a stack copy of the argument: 16 bytes of scratch in the hottest kernel)
  g.ox = uni(d.ox); g.oy = uni(d.oy); g.oz = uni(d.oz); g.cell = uni(d.cell); g.inv_cell = uni(d.inv_cell); g.eps = uni(d.eps);
  g.nx = uni(d.nx); g.ny = uni(d.ny); g.nz = uni(d.nz); g.ntx = uni(d.ntx); g.nty = uni(d.nty); g.ntz = uni(d.ntz); g.n = uni(d.n);
  return g;
}

// ------------------------------------------------------------------ pair as a grid dimension
// Every kernel of the registration chain is a functor F { TB, OCC, Args, run(args, bx, nbx) }: `bx` / `nbx` stand for blockIdx.x / gridDim.x.
// The classic launch (one registration per stream) wraps it in a __global__ of its own; the BATCHED launch is k_lanes<F>: blockIdx.y selects one
// entry of a device-resident table - one entry per (candidate pair, cloud) - so B independent registrations ride in ONE launch with their own
// buffers, grid sizes and state (SURVEY 7.1 step 8 "pair-as-grid-dimension"; loop_closure.cpp:116-124: the pairs share nothing).  The same
// instruction sequence runs on the same inputs either way: the records are bit-identical to the per-context path's.
// gridDim.x is a multiple of 8 in batched launches, so (linear workgroup id) mod 8 - the XCD a block lands on - is bx mod 8 like in a 1-D grid.
template <class A> struct alignas(16) LaneEntry { A a; uint32_t nbx; uint32_t pad_[3]; };
template <class F>
__global__ void __launch_bounds__(F::TB, F::OCC) k_lanes(const LaneEntry<typename F::Args>* __restrict__ tab) {
  const LaneEntry<typename F::Args>* e = tab + blockIdx.y;
  const uint32_t nbx = e->nbx;
  if (blockIdx.x >= nbx) return;
  F::run(e->a, blockIdx.x, nbx);
}

#define QN_INF_KEY 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int cell_coord(float v, float o, float inv, int n) {
  return clampi((int)floorf((v - o) * inv), 0, n - 1);
}
__device__ __forceinline__ uint32_t cell_key(const GridView& g, int x, int y, int z) {
  const uint32_t tile = ((uint32_t)(z >> 2) * g.nty + (y >> 2)) * g.ntx + (x >> 3);
  return (tile << 7) | ((z & 3) << 5) | ((y & 3) << 3) | (x & 7);
}
__device__ __forceinline__ unsigned long long pack_key(float d2, uint32_t idx) {
  return ((unsigned long long)__float_as_uint(d2) << 32) | idx;
}
__device__ __forceinline__ float key_d2(unsigned long long k) { return __uint_as_float((uint32_t)(k >> 32)); }
__device__ __forceinline__ uint32_t key_idx(unsigned long long k) { return (uint32_t)k; }

// plain f32 mul/add in source order (the library is compiled -ffp-contract=off): the oracle's
// `dx*dx + dy*dy + dz*dz` bit for bit.
// (x, y as ONE packed-f32 pair: v_pk_add_f32 / v_pk_mul_f32 do the two lanes' IEEE operations in one issue slot - same roundings, same sum order (dx^2 + dy^2) + dz^2,
//  six instructions instead of eight in every candidate loop of the engine)
typedef float qn_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float sqdist(float qx, float qy, float qz, float px, float py, float pz) {
  const qn_f2 q = {qx, qy}, p = {px, py};
  const qn_f2 d = q - p;
  const qn_f2 d2 = d * d;
  const float dz = qz - pz;
  return (d2.x + d2.y) + dz * dz;
}

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t rflu(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// the value a given lane holds, for a WAVE-UNIFORM lane number: one v_readlane_b32 (a __shfl with a computed lane is a ds_bpermute round trip)
__device__ __forceinline__ int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ float rdlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// Wave-wide reductions whose result every lane needs (all 64 lanes must call, convergent).  DPP ladder - row_shr 1, 2, 4, 8, row_bcast:15 into rows 1 / 3, row_bcast:31 into rows
// 2 / 3: lane 63 then holds the reduction of the wave - and one v_readlane; a lane without a source combines with the identity.  (Rounds 1-5: six __shfl_xor steps = six
// ds_bpermute round trips each - twelve for a 64-bit key - in front of every certification of the one-query-per-wave searches.)  The persistent align kernel keeps the shuffle
// forms (see wave_incl_scan_u32).
#if defined(QN_INST_GROUP) && QN_INST_GROUP == 10
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
__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
#else
#define QN_DPP_LADDER(STEP) STEP(0x111, 0xF) STEP(0x112, 0xF) STEP(0x114, 0xF) STEP(0x118, 0xF) STEP(0x142, 0xA) STEP(0x143, 0xC)
__device__ __forceinline__ int wave_min_i(int v) {
#define QN_S(ctrl, rm) { const int t = __builtin_amdgcn_update_dpp(0x7fffffff, v, ctrl, rm, 0xF, false); v = t < v ? t : v; }
  QN_DPP_LADDER(QN_S)
#undef QN_S
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max_i(int v) {
#define QN_S(ctrl, rm) { const int t = __builtin_amdgcn_update_dpp((int)0x80000000, v, ctrl, rm, 0xF, false); v = t > v ? t : v; }
  QN_DPP_LADDER(QN_S)
#undef QN_S
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#define QN_S(ctrl, rm) { const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, ctrl, rm, 0xF, false); v = t < v ? t : v; }
  QN_DPP_LADDER(QN_S)
#undef QN_S
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// (a 64-bit key = high word first: the minimum of the high words, then the minimum of the low words among the lanes that hold it)
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
  const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
  const uint32_t mh = wave_min_u32(hi);
  const uint32_t ml = wave_min_u32(hi == mh ? lo : 0xffffffffu);
  return ((unsigned long long)mh << 32) | ml;
}
__device__ __forceinline__ float wave_min_f(float v) {
#define QN_S(ctrl, rm) v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0x7f800000, __float_as_int(v), ctrl, rm, 0xF, false)));
  QN_DPP_LADDER(QN_S)
#undef QN_S
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_f(float v) {
#define QN_S(ctrl, rm) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp((int)0xff800000, __float_as_int(v), ctrl, rm, 0xF, false)));
  QN_DPP_LADDER(QN_S)
#undef QN_S
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
#endif
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// Wave64 sum with DPP for the 4 intra-row steps (VALU rate) and two cross-row bpermutes; the total
// lands in every lane (callers read lane 63 or any lane).  Fixed tree -> deterministic.
__device__ __forceinline__ double dpp_add_f64(double v, const int ctrl_sel) {
  union { double d; int i[2]; } a, b; a.d = v;
  switch (ctrl_sel) {
    case 0: b.i[0] = __builtin_amdgcn_mov_dpp(a.i[0], 0xB1, 0xF, 0xF, true); b.i[1] = __builtin_amdgcn_mov_dpp(a.i[1], 0xB1, 0xF, 0xF, true); break;   // quad_perm [1,0,3,2]
    case 1: b.i[0] = __builtin_amdgcn_mov_dpp(a.i[0], 0x4E, 0xF, 0xF, true); b.i[1] = __builtin_amdgcn_mov_dpp(a.i[1], 0x4E, 0xF, 0xF, true); break;   // quad_perm [2,3,0,1]
    case 2: b.i[0] = __builtin_amdgcn_mov_dpp(a.i[0], 0x141, 0xF, 0xF, true); b.i[1] = __builtin_amdgcn_mov_dpp(a.i[1], 0x141, 0xF, 0xF, true); break; // row_half_mirror
    default: b.i[0] = __builtin_amdgcn_mov_dpp(a.i[0], 0x140, 0xF, 0xF, true); b.i[1] = __builtin_amdgcn_mov_dpp(a.i[1], 0x140, 0xF, 0xF, true); break; // row_mirror
  }
  return v + b.d;
}
__device__ __forceinline__ double wave_sum_f64_dpp(double v) {
  v = dpp_add_f64(v, 0); v = dpp_add_f64(v, 1); v = dpp_add_f64(v, 2); v = dpp_add_f64(v, 3);   // 16-lane row sums in every lane
#if defined(QN_INST_GROUP) && QN_INST_GROUP == 10
  v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
  return v;
#else
  // the four row sums R0..R3 meet as (R3 + R2) + (R1 + R0) in lane 63 - bit for bit the value the two cross-row shuffle steps left in every lane (addition commutes exactly) -
  // on row_bcast:15 / row_bcast:31 and two v_readlane instead of four ds_bpermute round trips (28 sums per block of the unseeded accumulation: 112 of them)
  const int lane = threadIdx.x & 63;
  union { double d; int i[2]; } a, b;
  a.d = v; b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x142, 0xA, 0xF, false); b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x142, 0xA, 0xF, false);
  v = (lane & 16) ? v + b.d : v;
  a.d = v; b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x143, 0xC, 0xF, false); b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x143, 0xC, 0xF, false);
  v = (lane & 32) ? v + b.d : v;
  a.d = v; a.i[0] = __builtin_amdgcn_readlane(a.i[0], 63); a.i[1] = __builtin_amdgcn_readlane(a.i[1], 63);
  return a.d;
#endif
}
// Inclusive prefix sum over the 64 lanes (all lanes must call, convergent) on DPP: row_shr 1, 2, 4, 8 inside the rows of 16, then row_bcast:15 into rows 1 and 3 and
// row_bcast:31 into rows 2 and 3 - six VALU instructions, no LDS; a lane without a source adds the `old` operand, 0.  (Rounds 1-5: six __shfl_up steps = six ds_bpermute round
// trips and ~30 instructions, two to four times per search round of every kernel.)
#define QN_DPP_ADD(v, ctrl, rmask) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rmask, 0xF, false)
#if defined(QN_INST_GROUP) && QN_INST_GROUP == 10
// (the persistent align kernel keeps the shuffle form: with the DPP form its register allocation reserved a private segment - no instruction uses it, but a kernel with a
//  private segment pays extra at dispatch, tests/test_no_scratch.py; its cooperative searches are the rare path of a tick)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(v, o); if (lane >= o) v += t; }
  return v;
}
#else
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int) {
  QN_DPP_ADD(v, 0x111, 0xF); QN_DPP_ADD(v, 0x112, 0xF); QN_DPP_ADD(v, 0x114, 0xF); QN_DPP_ADD(v, 0x118, 0xF);
  QN_DPP_ADD(v, 0x142, 0xA); QN_DPP_ADD(v, 0x143, 0xC);
  return v;
}
#endif
// the same shape with max (u32, identity 0)
#define QN_DPP_MAX(v, ctrl, rmask) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rmask, 0xF, false))
__device__ __forceinline__ uint32_t wave_incl_max_u32(uint32_t v) {
  QN_DPP_MAX(v, 0x111, 0xF); QN_DPP_MAX(v, 0x112, 0xF); QN_DPP_MAX(v, 0x114, 0xF); QN_DPP_MAX(v, 0x118, 0xF);
  QN_DPP_MAX(v, 0x142, 0xA); QN_DPP_MAX(v, 0x143, 0xC);
  return v;
}
// Slot -> segment for one 64-candidate chunk of a segment table, without a search: every non-empty segment [ex, nx) that reaches into [cb, cb + 64) writes (its lane + 1) at its
// first slot inside the chunk; a running maximum over the chunk's slots (six DPP steps) is the segment each slot belongs to.  (Rounds 1-5: a six-step
// binary search over the prefix table per lane and chunk - six DEPENDENT LDS round trips in front of every candidate fetch of every search kernel.)  ex / nx: the
// first slot of this lane's OWN segment and of the next; marks: 64 words of wave-private LDS the caller does not need across the call.  All 64 lanes call.  (chunk_segment, below)
// DS operations of one wave execute in issue order; this only stops the compiler from reordering.
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
__device__ __forceinline__ int chunk_segment(uint32_t* __restrict__ marks, const uint32_t cb, const uint32_t ex, const uint32_t nx) {
  const int lane = threadIdx.x & 63;
  wave_lds_fence();
  marks[lane] = 0u;
  wave_lds_fence();
  if (nx > ex && ex < cb + 64u && nx > cb) marks[ex > cb ? ex - cb : 0u] = (uint32_t)lane + 1u;      // (a segment that began in an earlier chunk marks slot 0: every chunk stands alone)
  wave_lds_fence();
  return (int)wave_incl_max_u32(marks[lane]) - 1;
}

// Wave-aggregated list append: one atomic per wave, the wave's entries land contiguously and in lane order (so the
// list keeps the spatial coherence of the cell-sorted query order).  Must be called by the wave convergently.
__device__ __forceinline__ void wave_append(uint2* __restrict__ list, uint32_t* __restrict__ count, bool want, uint2 rec) {
  const unsigned long long mask = __ballot(want);
  if (mask == 0) return;
  const int lane = threadIdx.x & 63, leader = __ffsll((long long)mask) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(mask));
  base = rdlane(base, leader);
  if (want) list[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = rec;
}

// ------------------------------------------------------------------ wave-private LDS scratch
#define QN_PEND_CAP 16
struct WaveLds {                        // per-wave scratch: candidate tile + segment table + cluster boxes (~4 KiB)
  float4 tile[64];                      // staged candidates (ds_write_b128 / broadcast ds_read_b128)
  uint32_t tile_cid[64];                // cluster id of each staged candidate
  uint32_t seg_excl[64];                // exclusive prefix of segment lengths (candidate-stream slots)
  uint32_t seg_start[64];               // first pts[] index of each segment
  uint32_t seg_cid[64];                 // cluster each segment belongs to
  uint32_t cl_seg0[65];                 // first segment slot of each cluster (exclusive prefix of nseg)
  uint32_t cl_tile_mode[64];            // 1: the cluster's box is enumerated tile by tile (large boxes)
  int box[64][6];                       // per cluster: x0 x1 y0 y1 z0 z1 (cells, already grown by the margin and clipped)
};
struct WaveLdsK {
  WaveLds s;
  unsigned long long pend[QN_PEND_CAP][64];
};

// ------------------------------------------------------------------ result sinks
struct Best1 {                         // 1-NN, plus the squared distance of the runner-up (for bound pruning)
  unsigned long long key;
  float second;                        // smallest d2 among scanned points other than `key`'s point
  float bd;                            // key's d2 as a float (+inf while there is no key): best <= second always, so a new point's d2 updates the runner-up as the MEDIAN of the three
  __device__ __forceinline__ void init() { key = QN_INF_KEY; second = __int_as_float(0x7f800000); bd = __int_as_float(0x7f800000); }
  // UNIQUE: the caller never shows a point twice (every scan of the engine: one pass over disjoint segments per round) - the `same point again` test is dropped
  template <bool UNIQUE = false>
  __device__ __forceinline__ void consider(bool on, float d2, uint32_t idx) {
    const unsigned long long k = pack_key(d2, idx);                   // branch-free (selects): no exec-mask region for the scheduler to sink loads into
    const bool lt = on && k < key;                                     // (ties in d2: the lower index wins the key, the runner-up distance is that d2 either way)
    const float d2e = (on && (UNIQUE || k != key)) ? d2 : __int_as_float(0x7f800000);
    second = __builtin_amdgcn_fmed3f(bd, second, d2e);                // d2 < best: the old best; best <= d2 < second: d2; else unchanged
    bd = lt ? d2 : bd;
    key = lt ? k : key;
  }
  template <int S>
  __device__ __forceinline__ void finish(bool on) {                   // combine the S candidate sub-slots of a query
    if (S == 1) return;
    unsigned long long b = key, t;
    if (S == 4) { t = __shfl_xor(b, 16); b = t < b ? t : b; }
    t = __shfl_xor(b, 32); b = t < b ? t : b;
    float c = (key == b) ? second : key_d2(key);                      // this sub-slot's best runner-up candidate
    if (key == QN_INF_KEY) c = __int_as_float(0x7f800000);
    if (S == 4) c = fminf(c, __shfl_xor(c, 16));
    c = fminf(c, __shfl_xor(c, 32));
    if (on) { key = b; second = c; bd = b != QN_INF_KEY ? key_d2(b) : __int_as_float(0x7f800000); }
  }
  __device__ __forceinline__ void reset() { init(); }
  __device__ __forceinline__ bool full() const { return key != QN_INF_KEY; }
  __device__ __forceinline__ float worst_d2() const { return key_d2(key); }
  __device__ __forceinline__ int found() const { return key != QN_INF_KEY ? 1 : 0; }
  __device__ __forceinline__ int wanted() const { return 1; }
};

// k-NN: ascending (d2, idx) list in registers.  Inserting costs a KMAX-long compare-exchange chain
// that the WHOLE wave executes, so candidates that pass the (stale) k-th-best filter are parked in a
// per-lane LDS queue and merged QN_PEND_CAP at a time (lazy insertion).
template <int KMAX>
struct BestK {
  // a[0 .. KMAX-k) hold the sentinel key 0 (smaller-or-equal to every real key) for ever, so the k real
  // entries live in a[KMAX-k .. KMAX) and the k-th best is ALWAYS a[KMAX-1]: no runtime-indexed
  // register array (which hipcc would demote to scratch memory).
  unsigned long long a[KMAX];
  unsigned long long w;                // k-th best as of the last flush
  unsigned long long (*pend)[64];      // this wave's pending queue in LDS
  uint32_t* dbg;
  int k, cnt;
  __device__ __forceinline__ void reset() {
    w = QN_INF_KEY; cnt = 0;
    int kk = k; asm volatile("" : "+s"(kk));                           // (opaque here: as loop invariants the KMAX initial keys were kept alive - in scratch - across the searches between two resets)
#pragma unroll
    for (int j = 0; j < KMAX; j++) a[j] = (j < KMAX - kk) ? 0ull : QN_INF_KEY;
  }
  __device__ __forceinline__ void init(int k_, unsigned long long (*pend_)[64], uint32_t* dbg_ = nullptr) {
    k = k_; pend = pend_; dbg = dbg_; reset();
  }
  __device__ __forceinline__ bool slot_valid(int j) const { return j >= KMAX - k && a[j] != QN_INF_KEY; }   // j static
  __device__ __forceinline__ void refresh() { w = a[KMAX - 1]; }
  __device__ __forceinline__ void insert(unsigned long long key) {   // compare-exchange chain: sorted insert, largest falls off
#pragma unroll
    for (int j = 0; j < KMAX; j++) {
      unsigned long long lo = key < a[j] ? key : a[j];
      unsigned long long hi = key < a[j] ? a[j] : key;
      a[j] = lo; key = hi;
    }
  }
  __device__ __forceinline__ void flush() {                          // wave-cooperative
    const int lane = threadIdx.x & 63;
    const int m = wave_max_i(cnt);
    for (int e = 0; e < m; e++) {
      unsigned long long key = e < cnt ? pend[e][lane] : QN_INF_KEY;
      insert(key);
    }
    cnt = 0;
    refresh();
    if (dbg && lane == 0) atomicAdd(&dbg[2], 1u);
  }
  __device__ __forceinline__ void consider(bool on, float d2, uint32_t idx) {   // wave-cooperative: called by all lanes
    const unsigned long long key = pack_key(d2, idx);
    if (on && key < w) { pend[cnt][threadIdx.x & 63] = key; cnt++; }
    if (__any(cnt == QN_PEND_CAP)) flush();
  }
  // One-directional merge: lanes whose `recv` is true insert the k real entries of lane ^ lane_xor;
  // the sending lanes insert the +inf key (a no-op), so their lists stay intact while being read.
  __device__ __forceinline__ void merge_from(int lane_xor, bool recv) {
    if (KMAX > 24) {                 // fully unrolled, the KMAX x KMAX merge takes ~6 MINUTES to compile at KMAX = 32 (3 s rolled); k > 24 is off the hot path
      unsigned long long tmp[KMAX];
#pragma unroll
      for (int j = 0; j < KMAX; j++) tmp[j] = __shfl_xor(a[j], lane_xor);
#pragma unroll 1
      for (int j = 0; j < KMAX; j++) {
        unsigned long long other = QN_INF_KEY;
#pragma unroll
        for (int u = 0; u < KMAX; u++) other = (u == j) ? tmp[u] : other;      // select chain instead of a runtime-indexed register array
        if (__any(j >= KMAX - k)) insert((recv && j >= KMAX - k) ? other : QN_INF_KEY);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < KMAX; j++) {
      const unsigned long long other = __shfl_xor(a[j], lane_xor);
      if (__any(j >= KMAX - k)) insert((recv && j >= KMAX - k) ? other : QN_INF_KEY);
    }
  }
  // flush, then fold the 4 candidate sub-slots of a query into sub-slot 0.  `on` = this lane's query took
  // part in the round that just ended (queries finished in an earlier round must not merge twice).
  template <int S>
  __device__ __forceinline__ void finish(bool on) {
    if (__any(cnt > 0)) flush();
    if (S == 1) return;
    const int lane = threadIdx.x & 63;
    merge_from(32, on && (lane & 32) == 0);        // sub 0 <- 2, sub 1 <- 3   (S = 2: sub 0 <- 1)
    if (S == 4) merge_from(16, on && (lane & 48) == 0);        // sub 0 <- 1
    refresh();
    const unsigned long long w0 = __shfl(w, lane & (64 / S - 1));
    if (on) w = w0;                                // every lane of the query sees the merged k-th best
  }
  __device__ __forceinline__ bool full() const { return w != QN_INF_KEY; }
  __device__ __forceinline__ float worst_d2() const { return key_d2(w); }
  __device__ __forceinline__ int found() const {                       // real entries in the (merged) list
    int f = 0;
#pragma unroll
    for (int j = 0; j < KMAX; j++) f += (j >= KMAX - k && a[j] != QN_INF_KEY) ? 1 : 0;
    return f;
  }
  __device__ __forceinline__ int wanted() const { return k; }
};

// Lower bound (squared, rounded down) of the distance from q to any point stored in tile (tx, ty, tz): the tile's cell box, shrunk by the
// grid's eps on every side (the slack of cell_coord's f32 rounding).  Big-ball scans skip the tiles this proves to be outside the ball:
// the bounding box of a 24 m ball around a query beside the target covers thousands of square metres of ground that no neighbour can be on.
__device__ __forceinline__ float tile_box_d2(const GridView& g, int tx, int ty, int tz, float qx, float qy, float qz) {
  const float xl = g.ox + (float)(tx << 3) * g.cell, xh = g.ox + (float)min((tx << 3) + 8, g.nx) * g.cell;
  const float yl = g.oy + (float)(ty << 2) * g.cell, yh = g.oy + (float)min((ty << 2) + 4, g.ny) * g.cell;
  const float zl = g.oz + (float)(tz << 2) * g.cell, zh = g.oz + (float)min((tz << 2) + 4, g.nz) * g.cell;
  const float dx = fmaxf(fmaxf(xl - qx, qx - xh) - g.eps, 0.f), dy = fmaxf(fmaxf(yl - qy, qy - yh) - g.eps, 0.f), dz = fmaxf(fmaxf(zl - qz, qz - zh) - g.eps, 0.f);
  return ((dx * dx + dy * dy) + dz * dz) * 0.999998f;
}

// Exact n / d and n % d for 0 <= n < 2^22, 1 <= d < 2^22 from one v_rcp_f32 and a +-1 correction (the generic u32 division
// is ~25 VALU instructions, and the segment tables need four of them per lane and pass).
__device__ __forceinline__ void divmod_small(int n, int d, int& q, int& r) {
  q = (int)((float)n * __builtin_amdgcn_rcpf((float)d));
  r = n - q * d;
  if (r < 0) { q--; r += d; }
  if (r >= d) { q++; r -= d; }
}

// ------------------------------------------------------------------ dense candidate stream over a cell box
// Calls body(p, valid, cnt) once per 64-candidate chunk with one candidate per lane (`valid` = lane
// holds a real one, `cnt` = candidates in this chunk, wave-uniform).  All 64 lanes must call.
template <class Body>
__device__ __forceinline__ uint32_t stream_box(const GridView& g, int x0, int x1, int y0, int y1, int z0, int z1, const bool tile_mode, WaveLds* lds, Body&& body,
                                               const float bqx = 0.f, const float bqy = 0.f, const float bqz = 0.f, const float ball_r2 = -1.f) {
  // ball_r2 >= 0 (tile mode only): tiles proven farther than sqrt(ball_r2) from (bqx, bqy, bqz) are skipped - the caller only wants points inside that ball
  // tile_mode: the box has been widened to whole 8x4x4-cell tiles by the caller and is enumerated tile by tile (a tile's 128
  // cells are one contiguous run of pts[]) - a big ball is mostly empty space, row-wise enumeration would spend its time on
  // cell_start look-ups of empty rows.
  const int lane = threadIdx.x & 63;
  const int tx0 = x0 >> 3, ntr = (x1 >> 3) - tx0 + 1, nyr = y1 - y0 + 1;
  const int ty0 = y0 >> 2, ntyr = (y1 >> 2) - ty0 + 1;
  const int nseg = tile_mode ? ntr * ntyr * ((z1 >> 2) - (z0 >> 2) + 1) : ntr * nyr * (z1 - z0 + 1);
  uint32_t grand = 0;
  for (int sb = 0; sb < nseg; sb += 64) {
    const int sidx = sb + lane;
    uint32_t s = 0, len = 0;
    if (sidx < nseg) {
      int t, r; divmod_small(sidx, ntr, r, t);                         // (t = sidx % ntr, r = sidx / ntr: the generic integer division is ~30 instructions, four of them per table)
      if (tile_mode) {
        int qz_, ry_; divmod_small(r, ntyr, qz_, ry_);
        const int tzz = (z0 >> 2) + qz_, tyy = ty0 + ry_, txx = tx0 + t;
        const uint32_t tile = ((uint32_t)tzz * g.nty + tyy) * g.ntx + txx;
        if (!(ball_r2 >= 0.f && tile_box_d2(g, txx, tyy, tzz, bqx, bqy, bqz) > ball_r2)) { s = g.cell_start[tile << 7]; len = g.cell_start[(tile + 1) << 7] - s; }
      } else {
        int qz_, ry_; divmod_small(r, nyr, qz_, ry_);
        const int ry = y0 + ry_, rz = z0 + qz_, tx = tx0 + t;
        const int xa = max(x0, tx << 3), xb = min(x1, (tx << 3) + 7);
        const uint32_t k0 = cell_key(g, xa, ry, rz);
        s = g.cell_start[k0]; len = g.cell_start[k0 + (xb - xa) + 1] - s;
      }
    }
    const uint32_t incl = wave_incl_scan_u32(len, lane);
    const uint32_t total = rdlane(incl, 63);
    if (total == 0) continue;
    wave_lds_fence();
    lds->seg_excl[lane] = incl - len; lds->seg_start[lane] = s;
    wave_lds_fence();
    for (uint32_t cb = 0; cb < total; cb += 64) {
      const uint32_t slot = cb + lane;
      const bool valid = slot < total;
      const int j = chunk_segment(lds->tile_cid, cb, incl - len, incl);
      float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid) p = g.pts[lds->seg_start[j] + (slot - lds->seg_excl[j])];
      body(p, valid, min(64u, total - cb));
    }
    grand += total;
  }
  return grand;
}

// ------------------------------------------------------------------ cooperative exact search
#define QN_CL_DX 8
#define QN_CL_DY 4
#define QN_CL_DZ 4

// Steps (1)-(3) of a search round.  The queries of the `todo` lanes are grouped into clusters by anchor proximity, each
// cluster's cell box (union of its members' ball boxes) is left in lds->box[cluster], and the boxes are cut into
// segments (runs of pts[]): lds->cl_seg0 / cl_tile_mode.  cid = this lane's cluster, ncl = clusters, nseg_all = segments.
template <int S>
__device__ __forceinline__ void build_clusters(const GridView& g, WaveLds* lds, unsigned long long todo, int cx, int cy, int cz,
                                               float qx, float qy, float qz, float r, uint32_t& cid, int& ncl, uint32_t& nseg_all) {
  const int lane = threadIdx.x & 63;
  const bool mine = (todo >> lane) & 1ull;
  // (1) cluster ids
  cid = 0xffffffffu; ncl = 0;
  for (unsigned long long rem = todo; rem != 0; ncl++) {
    const int leader = __ffsll((long long)rem) - 1;
    const int ax = rdlane(cx, leader), ay = rdlane(cy, leader), az = rdlane(cz, leader);
    const bool in = ((rem >> lane) & 1ull) && abs(cx - ax) <= QN_CL_DX && abs(cy - ay) <= QN_CL_DY && abs(cz - az) <= QN_CL_DZ;
    if (in) cid = ncl;
    rem &= ~__ballot(in);
  }
  // (2) cluster boxes = union of the member queries' ball boxes
  wave_lds_fence();
  if (lane < ncl) { lds->box[lane][0] = 0x3fffffff; lds->box[lane][1] = -1; lds->box[lane][2] = 0x3fffffff; lds->box[lane][3] = -1; lds->box[lane][4] = 0x3fffffff; lds->box[lane][5] = -1; }
  wave_lds_fence();
  if (mine && lane < 64 / S) {
    atomicMin(&lds->box[cid][0], cell_coord(qx - r, g.ox, g.inv_cell, g.nx)); atomicMax(&lds->box[cid][1], cell_coord(qx + r, g.ox, g.inv_cell, g.nx));
    atomicMin(&lds->box[cid][2], cell_coord(qy - r, g.oy, g.inv_cell, g.ny)); atomicMax(&lds->box[cid][3], cell_coord(qy + r, g.oy, g.inv_cell, g.ny));
    atomicMin(&lds->box[cid][4], cell_coord(qz - r, g.oz, g.inv_cell, g.nz)); atomicMax(&lds->box[cid][5], cell_coord(qz + r, g.oz, g.inv_cell, g.nz));
  }
  wave_lds_fence();
  // (3) segments per cluster, prefix over clusters
  // A box with many (y, z) rows is mostly empty space around a surface: enumerate it TILE by tile
  // instead (a tile's 128 cells are one contiguous run of pts[]), which needs far fewer cell_start
  // look-ups; the box is widened to whole tiles, so certification sees the larger scanned volume.
  uint32_t my_nseg = 0;
  if (lane < ncl) {
    int* b = lds->box[lane];
    my_nseg = (uint32_t)(((b[1] >> 3) - (b[0] >> 3) + 1) * (b[3] - b[2] + 1) * (b[5] - b[4] + 1));
    uint32_t tm = 0;
    if (my_nseg > 384) {
      tm = 1;
      b[0] = (b[0] >> 3) << 3; b[1] = min(((b[1] >> 3) << 3) + 7, g.nx - 1);
      b[2] = (b[2] >> 2) << 2; b[3] = min(((b[3] >> 2) << 2) + 3, g.ny - 1);
      b[4] = (b[4] >> 2) << 2; b[5] = min(((b[5] >> 2) << 2) + 3, g.nz - 1);
      my_nseg = (uint32_t)(((b[1] >> 3) - (b[0] >> 3) + 1) * ((b[3] >> 2) - (b[2] >> 2) + 1) * ((b[5] >> 2) - (b[4] >> 2) + 1));
    }
    lds->cl_tile_mode[lane] = tm;
  }
  const uint32_t seg_incl = wave_incl_scan_u32(my_nseg, lane);
  nseg_all = rdlane(seg_incl, 63);
  lds->cl_seg0[lane] = seg_incl - my_nseg;
  if (lane == 0) lds->cl_seg0[64] = nseg_all;
  wave_lds_fence();
}

// Step (4): the points of all cluster boxes as one dense candidate stream.  fn(cp, in_tile, ccid) is called once per
// step by all 64 lanes; the S sub-slots see S consecutive candidates (cp = point, .w = original index bits; in_tile =
// the slot holds a real candidate; ccid = the cluster whose box the candidate came from).  Returns the stream length.
// (the chunk loop of stream_clusters: the `total` candidates of the segment table that sits in lds->seg_excl / seg_start / seg_cid)
template <int S, class Fn>
__device__ __forceinline__ void stream_chunks(const GridView& g, WaveLds* lds, const uint32_t total, Fn&& fn) {
  const int lane = threadIdx.x & 63;
  const float INF = __int_as_float(0x7f800000);
  const uint32_t ex = lds->seg_excl[lane], nxs = lane < 63 ? lds->seg_excl[(lane + 1) & 63] : total;      // (this lane's own segment of the table: first slot, first slot of the next)
  for (uint32_t cb = 0; cb < total; cb += 64) {
    const uint32_t slot = cb + lane;
    const int j = chunk_segment(lds->tile_cid, cb, ex, nxs);
    const uint32_t cnt = min(64u, total - cb);
    wave_lds_fence();
    // empty slots belong to no cluster and sit infinitely far away: the `ccid == cid` test of the scorers rejects them, and so does every distance test
    lds->tile[lane] = slot < total ? g.pts[lds->seg_start[j] + (slot - lds->seg_excl[j])] : make_float4(INF, INF, INF, 0.f);
    lds->tile_cid[lane] = slot < total ? lds->seg_cid[j] : 0xffffffffu;
    wave_lds_fence();
    // S candidates per step (one per sub-slot), 4 steps per trip with the four ds_read_b128 issued BEFORE any scoring: with
    // the read inside a predicated body every step paid the LDS latency in full (read -> wait -> score -> branch).
    const uint32_t sub_off = (uint32_t)(S == 1 ? 0 : lane / (64 / S));
    for (uint32_t c = 0; c < cnt; c += 4 * S) {
      float4 cp[4]; uint32_t cc[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const uint32_t ci = (c + (uint32_t)(S * u) + sub_off) & 63u; cp[u] = lds->tile[ci]; cc[u] = lds->tile_cid[ci]; }
#pragma unroll
      for (int u = 0; u < 4; u++) fn(cp[u], true, cc[u]);
    }
  }
}
// REUSE: the caller streams the SAME clusters a second time (the two passes of the histogram selection): when all segments fit one table (nseg_all <= 64) the
// table of the first walk is still in LDS - `reuse_total` is what that walk returned - and the cell_start gather, the divisions and the scan are skipped.
// (stream_tables: the segment tables of all cluster boxes, 64 segments at a time; `chunks(total)` consumes the `total` candidates of the table that sits in LDS)
template <bool REUSE = false, class Chunks>
__device__ __forceinline__ uint32_t stream_tables(const GridView& g, WaveLds* lds, int ncl, uint32_t nseg_all, Chunks&& chunks, const uint32_t reuse_total = 0) {
  const int lane = threadIdx.x & 63;
  if (REUSE && nseg_all <= 64u) { chunks(reuse_total); return reuse_total; }
  uint32_t ncand = 0;
  for (uint32_t sb = 0; sb < nseg_all; sb += 64) {
    const uint32_t sidx = sb + lane;
    uint32_t s = 0, len = 0, scid = 0;
    if (sidx < nseg_all) {
      int c = 0;                                   // last cluster with cl_seg0[c] <= sidx  (ncl <= 64)
#pragma unroll
      for (int step = 32; step > 0; step >>= 1) { const int t = c + step; if (t < ncl && lds->cl_seg0[t] <= sidx) c = t; }
      const int* b = lds->box[c];
      const int x0 = b[0], x1 = b[1], y0 = b[2], y1 = b[3], z0 = b[4];
      const int tx0 = x0 >> 3, ntr = (x1 >> 3) - tx0 + 1;
      const int li = (int)(sidx - lds->cl_seg0[c]);
      int t, rr; divmod_small(li, ntr, rr, t);
      if (lds->cl_tile_mode[c]) {                  // segment = one whole tile
        const int ty0 = y0 >> 2, ntyr = (y1 >> 2) - ty0 + 1;
        int qz_, ry_; divmod_small(rr, ntyr, qz_, ry_);
        const uint32_t tile = ((uint32_t)((z0 >> 2) + qz_) * g.nty + (ty0 + ry_)) * g.ntx + (tx0 + t);
        s = g.cell_start[tile << 7]; len = g.cell_start[(tile + 1) << 7] - s;
      } else {                                     // segment = cells [xa..xb] of row (ry, rz) inside tile tx
        const int nyr = y1 - y0 + 1;
        int qz_, ry_; divmod_small(rr, nyr, qz_, ry_);
        const int ry = y0 + ry_, rz = z0 + qz_, tx = tx0 + t;
        const int xa = max(x0, tx << 3), xb = min(x1, (tx << 3) + 7);
        const uint32_t k0 = cell_key(g, xa, ry, rz);
        s = g.cell_start[k0]; len = g.cell_start[k0 + (xb - xa) + 1] - s;
      }
      scid = (uint32_t)c;
    }
    const uint32_t incl = wave_incl_scan_u32(len, lane);
    const uint32_t total = rdlane(incl, 63);
    if (total == 0) continue;
    wave_lds_fence();
    lds->seg_excl[lane] = incl - len; lds->seg_start[lane] = s; lds->seg_cid[lane] = scid;
    wave_lds_fence();
    chunks(total);
    ncand += total;
  }
  return ncand;
}
template <int S, bool REUSE = false, class Fn>
__device__ __forceinline__ uint32_t stream_clusters(const GridView& g, WaveLds* lds, int ncl, uint32_t nseg_all, Fn&& fn, const uint32_t reuse_total = 0) {
  return stream_tables<REUSE>(g, lds, ncl, nseg_all, [&](const uint32_t total) __attribute__((always_inline)) { stream_chunks<S>(g, lds, total, fn); }, reuse_total);
}

// wave_search: the ONE cooperative search routine (1-NN and k-NN, first search and seeded re-search).
//
// A wavefront serves 64 / S QUERIES (S = 4: lane l works for query (l & 15) as candidate sub-slot (l >> 4), i.e. the
// four lanes of a query split the candidate stream four ways and merge in sink.finish(); 100k queries then make
// 6250 waves of the chip's 8192, each with a 4x shorter candidate loop - best latency.  S = 1: one query per lane,
// 4x fewer wave-instructions per query - best throughput when several registrations are in flight).  All 64 lanes
// must call (idle lanes pass active = false; the S lanes of a query pass identical q, r).
//
// Every query carries a search radius r (world units): a seed bound when the caller knows one (distance
// to last iteration's neighbour), else margin * cell.  One ROUND handles every cluster of the wave at once
// so its memory latency is paid once: (1) lanes are partitioned into clusters around anchor lanes (ALU
// only); (2) each cluster's cell box = union of its queries' ball boxes, built with LDS min/max atomics;
// (3) all boxes are cut into segments, one per lane, whose bounds come from ONE gather of cell_start;
// (4) the candidates of all segments form one dense stream fetched 64 at a time (coalesced within a
// segment), staged in LDS with the owning cluster id, and scored by that cluster's lanes via
// ds_read_b128.  A query is CERTIFIED exact when its (k-th) best distance is smaller than its distance to
// the nearest box face that still has unseen cells behind it; otherwise its radius grows (to the k-th
// best distance if known - then the next round certifies - else 2 r + cell, never beyond the proven
// bound r_cap) and it retries, up to max_rounds.  Returns certified; r is updated to the radius the next
// round would use; d_unseen = lower bound on the distance of every point NOT scanned in the last round.
template <int S, class Sink>
__device__ __forceinline__ bool wave_search(const GridView& g, float qx, float qy, float qz, bool active, float& r, const float r_cap,
                                            int max_rounds, Sink& sink, WaveLds* lds, float& d_unseen) {
  const int lane = threadIdx.x & 63;
  const int cx = cell_coord(qx, g.ox, g.inv_cell, g.nx);
  const int cy = cell_coord(qy, g.oy, g.inv_cell, g.ny);
  const int cz = cell_coord(qz, g.oz, g.inv_cell, g.nz);
  bool certified = false;
  unsigned long long todo = __ballot(active);
  for (int round = 0; todo != 0 && round < max_rounds; round++) {
    const bool mine = (todo >> lane) & 1ull;
    uint32_t cid; int ncl; uint32_t nseg_all;
    build_clusters<S>(g, lds, todo, cx, cy, cz, qx, qy, qz, r, cid, ncl, nseg_all);
    const uint32_t ncand = stream_clusters<S>(g, lds, ncl, nseg_all, [&](const float4& cp, bool in_tile, uint32_t ccid) __attribute__((always_inline)) {
      sink.consider(mine && in_tile && ccid == cid, sqdist(qx, qy, qz, cp.x, cp.y, cp.z), __float_as_uint(cp.w));
    });
    sink.template finish<S>(mine);
    if (g.dbg && lane == 0) { atomicAdd(&g.dbg[0], (uint32_t)ncl); atomicAdd(&g.dbg[1], ncand); }
    // certification: nearest face of the scanned box that has unseen cells behind it
    bool retry = false;
    if (mine) {
      const int* b = lds->box[cid];
      const int ex0 = b[0], ex1 = b[1], ey0 = b[2], ey1 = b[3], ez0 = b[4], ez1 = b[5];
      const float INF = __int_as_float(0x7f800000);
      float d = INF;
      if (ex0 > 0) d = fminf(d, qx - (g.ox + ex0 * g.cell));
      if (ex1 < g.nx - 1) d = fminf(d, (g.ox + (ex1 + 1) * g.cell) - qx);
      if (ey0 > 0) d = fminf(d, qy - (g.oy + ey0 * g.cell));
      if (ey1 < g.ny - 1) d = fminf(d, (g.oy + (ey1 + 1) * g.cell) - qy);
      if (ez0 > 0) d = fminf(d, qz - (g.oz + ez0 * g.cell));
      if (ez1 < g.nz - 1) d = fminf(d, (g.oz + (ez1 + 1) * g.cell) - qz);
      if (d == INF || !(r == r)) { certified = true; d_unseen = INF; }   // the whole grid was scanned (or a non-finite query: nothing to find)
      else { d -= g.eps; d_unseen = d; certified = d > 0.f && sink.full() && sink.worst_d2() < d * d; }
      if (!certified) {
        if (sink.full()) r = fmaxf(sqrtf(sink.worst_d2()) * 1.000001f + g.eps, r + g.eps);      // (never the same radius twice: see wave_search_single)
        else {                                                // fewer than k points in the box: extrapolate from the count (surface-like data: count ~ r^2)
          const int f = sink.found();                         // instead of doubling blindly (a sparse-region k-NN query then re-scans far less)
          r = f > 0 ? fmaxf(r * sqrtf((float)sink.wanted() / (float)f) * 1.25f, r + g.cell) : 2.f * r + g.cell;
        }
        r = fminf(r, r_cap);                                 // r_cap: a proven upper bound on the (k-th) NN distance, +inf if none
        if (round + 1 < max_rounds) { retry = true; sink.reset(); }
      }
    }
    todo = __ballot(retry);
    if (g.dbg && lane == 0 && todo) atomicAdd(&g.dbg[3], (uint32_t)__popcll(todo));
  }
  return certified;
}

// ------------------------------------------------------------------ k-NN by histogram selection
// The k-NN stage is VALU-bound: keeping a sorted k-list per lane costs ~100 instructions per inserted candidate.
// ------------------------------------------------------------------ a ball clipped to the grid's bounding box
// Every point lies inside the grid box.  For a query OUTSIDE it (a source point beyond the target's extent) the part of the ball
// (q, r) that can hold points is a spherical cap: along axis a it reaches only rho_a = sqrt(r^2 - sum_{b != a} o_b^2) from q, where o_b
// is the query's distance to the box along axis b (0 when inside).  The search box shrinks from (2 r)^3 to the cap's bounding box -
// for a neighbour 15 m away, a few hundred candidates instead of thousands - and the "unseen" distance behind a face at axis offset f
// becomes sqrt(f^2 + sum_{b != a} o_b^2): whatever lies beyond that face AND inside the grid is at least that far away.
struct CapBox {
  float o2[3];                         // o_b^2 per axis
  float S;                             // their sum
};
__device__ __forceinline__ CapBox cap_of(const GridView& g, float qx, float qy, float qz) {
  CapBox c;
  const float ox = fmaxf(fmaxf(g.ox - qx, qx - (g.ox + g.nx * g.cell)), 0.f);
  const float oy = fmaxf(fmaxf(g.oy - qy, qy - (g.oy + g.ny * g.cell)), 0.f);
  const float oz = fmaxf(fmaxf(g.oz - qz, qz - (g.oz + g.nz * g.cell)), 0.f);
  c.o2[0] = ox * ox; c.o2[1] = oy * oy; c.o2[2] = oz * oz; c.S = (c.o2[0] + c.o2[1]) + c.o2[2];
  return c;
}
// half extent of the cap along `axis` (>= the true one: rounded up, plus the grid's eps)
__device__ __forceinline__ float cap_extent(const GridView& g, const CapBox& c, float r, int axis) {
  const float rest = c.S - c.o2[axis];
  if (rest <= 0.f) return r;                                         // inside the box along the other axes: the plain ball
  const float v = r * r - rest * 0.999999f;
  return v > 0.f ? sqrtf(v) * 1.000001f + g.eps : g.eps;
}
// lower bound on the distance from q to anything beyond a face at axis offset f (f = distance from q to the face plane along `axis`)
__device__ __forceinline__ float cap_face_dist(const CapBox& c, float f, int axis) {
  const float rest = (c.S - c.o2[axis]) * 0.999999f;
  return rest > 0.f ? sqrtf(f * f + rest) * 0.999999f : f;
}

// ------------------------------------------------------------------ single-query search: one query per WAVE
// For the few queries whose neighbour is far away (several cells): all 64 lanes share ONE query and each
// scores its own candidate of the dense stream (no LDS tile needed), so a big ball is scanned 16x faster
// than in the 16-query layout.  The ball's cell box is enumerated row by row, or tile by tile when it is
// large (the inside of the ball is empty space - the neighbour sits on its surface - so most tiles cost
// one look-up and contribute no candidates).  Same certification and growth rule as wave_search.
struct SingleStats { uint32_t rounds, cand, segs; float r_first, r_last; };      // developer probe (list_probe): what one search did
__device__ __forceinline__ void wave_search_single(const GridView& g, float qx, float qy, float qz, float r, const float r_cap,
                                                   unsigned long long& best_out, float& second_out, float& d_unseen, WaveLds* lds, SingleStats& st_out, const bool stats) {      // (stats: by reference + flag - a conditional POINTER to the caller's struct made it a stack object: scratch in a kernel of the chain)
  const int lane = threadIdx.x & 63;
  const float INF = __int_as_float(0x7f800000);
  const CapBox cap = cap_of(g, qx, qy, qz);
  if (cap.S > 0.f) r = fminf(fmaxf(r, sqrtf(cap.S) + g.cell), fmaxf(r_cap, r));   // nothing is closer than the grid box itself: do not spend rounds below that
  if (stats) { st_out.rounds = 0; st_out.cand = 0; st_out.segs = 0; st_out.r_first = r; }
  // Growth rounds scan SHELLS: a tile-mode round scans whole tiles, so once the tiles within r (box distance <= r^2, the inclusion rule below) are done, the next
  // round only needs the tiles between the two radii and the running best / runner-up carry over - the scanned set after any round is exactly the set a fresh
  // scan of that round's ball would visit, each point once.  (Row-mode rounds - small boxes, partial tiles - start afresh.)
  float done2 = -1.f;                                                // tiles of the previous round's range [ptx0..ptx1] x [pty0..pty1] x [ptz0..ptz1] with box distance <= done2 are scanned
  int ptx0 = 0, ptx1 = -1, pty0 = 0, pty1 = -1, ptz0 = 0, ptz1 = -1; // (the ranges nest: r never shrinks.  A tile the f32 cell arithmetic left out of an earlier range is NOT skipped)
  unsigned long long best = QN_INF_KEY; float second = INF;
  for (int round = 0;; round++) {
    const float rx = cap_extent(g, cap, r, 0), ry = cap_extent(g, cap, r, 1), rz = cap_extent(g, cap, r, 2);
    int x0 = rfl(cell_coord(qx - rx, g.ox, g.inv_cell, g.nx)), x1 = rfl(cell_coord(qx + rx, g.ox, g.inv_cell, g.nx));
    int y0 = rfl(cell_coord(qy - ry, g.oy, g.inv_cell, g.ny)), y1 = rfl(cell_coord(qy + ry, g.oy, g.inv_cell, g.ny));
    int z0 = rfl(cell_coord(qz - rz, g.oz, g.inv_cell, g.nz)), z1 = rfl(cell_coord(qz + rz, g.oz, g.inv_cell, g.nz));
    const int tx0 = x0 >> 3, ntr = (x1 >> 3) - tx0 + 1;
    int nseg = ntr * (y1 - y0 + 1) * (z1 - z0 + 1);
    const bool tile_mode = nseg > 128;
    int ty0 = 0, ntyr = 1;
    if (tile_mode) {
      x0 = (x0 >> 3) << 3; x1 = min(((x1 >> 3) << 3) + 7, g.nx - 1);
      y0 = (y0 >> 2) << 2; y1 = min(((y1 >> 2) << 2) + 3, g.ny - 1);
      z0 = (z0 >> 2) << 2; z1 = min(((z1 >> 2) << 2) + 3, g.nz - 1);
      ty0 = y0 >> 2; ntyr = (y1 >> 2) - ty0 + 1;
      nseg = ntr * ntyr * ((z1 >> 2) - (z0 >> 2) + 1);
    }
    const int nyr = y1 - y0 + 1;
    if (!tile_mode || done2 < 0.f) { best = QN_INF_KEY; second = INF; done2 = -1.f; }
    const float in2 = r * r * 1.000002f;
    for (int sb = 0; sb < nseg; sb += 64) {
      const int sidx = sb + lane;
      uint32_t s = 0, len = 0;
      if (sidx < nseg) {
        int t, rr; divmod_small(sidx, ntr, rr, t);
        if (tile_mode) {                                               // only the tiles that reach into the ball of radius r and were not scanned by an earlier round
          int qz_, ry_; divmod_small(rr, ntyr, qz_, ry_);
          const int tzz = (z0 >> 2) + qz_, tyy = ty0 + ry_, txx = tx0 + t;
          const uint32_t tile = ((uint32_t)tzz * g.nty + tyy) * g.ntx + txx;
          const float tb2 = tile_box_d2(g, txx, tyy, tzz, qx, qy, qz);
          const bool seen = tb2 <= done2 && txx >= ptx0 && txx <= ptx1 && tyy >= pty0 && tyy <= pty1 && tzz >= ptz0 && tzz <= ptz1;
          if (!(tb2 > in2) && !seen) { s = g.cell_start[tile << 7]; len = g.cell_start[(tile + 1) << 7] - s; }
        } else {
          int qz_, ry_; divmod_small(rr, nyr, qz_, ry_);
          const int ry = y0 + ry_, rz = z0 + qz_, tx = tx0 + t;
          const int xa = max(x0, tx << 3), xb = min(x1, (tx << 3) + 7);
          const uint32_t k0 = cell_key(g, xa, ry, rz);
          s = g.cell_start[k0]; len = g.cell_start[k0 + (xb - xa) + 1] - s;
        }
      }
      const uint32_t incl = wave_incl_scan_u32(len, lane);
      const uint32_t total = rdlane(incl, 63);
      if (total == 0) continue;
      if (g.dbg && lane == 0) atomicAdd(&g.dbg[11], total);          // developer counter: candidates scanned
      if (stats) st_out.cand += total;
      wave_lds_fence();
      lds->seg_excl[lane] = incl - len; lds->seg_start[lane] = s;
      wave_lds_fence();
      for (uint32_t cb = 0; cb < total; cb += 64) {
        const uint32_t slot = cb + lane;
        const int j = chunk_segment(lds->tile_cid, cb, incl - len, incl);
        if (slot < total) {
          const float4 p = g.pts[lds->seg_start[j] + (slot - lds->seg_excl[j])];
          const float d2 = sqdist(qx, qy, qz, p.x, p.y, p.z);
          const unsigned long long k = pack_key(d2, __float_as_uint(p.w));
          if (k < best) { if (best != QN_INF_KEY) second = key_d2(best); best = k; }
          else if (d2 < second) second = d2;
        }
      }
    }
    if (g.dbg && lane == 0) { atomicAdd(&g.dbg[10], 1u); atomicAdd(&g.dbg[12], (uint32_t)nseg); if (round == 0) atomicAdd(&g.dbg[13], 1u); }      // developer counters: rounds, enumerated segments, entries
    if (stats) { st_out.rounds++; st_out.segs += (uint32_t)nseg; st_out.r_last = r; }
    if (tile_mode) { done2 = in2; ptx0 = tx0; ptx1 = tx0 + ntr - 1; pty0 = ty0; pty1 = ty0 + ntyr - 1; ptz0 = z0 >> 2; ptz1 = z1 >> 2; }
    const unsigned long long b = wave_min_u64(best);
    float c = (best == b) ? second : key_d2(best);
    if (best == QN_INF_KEY) c = INF;
    c = wave_min_f(c);
    float d = INF;
    if (x0 > 0) d = fminf(d, cap_face_dist(cap, qx - (g.ox + x0 * g.cell), 0));
    if (x1 < g.nx - 1) d = fminf(d, cap_face_dist(cap, (g.ox + (x1 + 1) * g.cell) - qx, 0));
    if (y0 > 0) d = fminf(d, cap_face_dist(cap, qy - (g.oy + y0 * g.cell), 1));
    if (y1 < g.ny - 1) d = fminf(d, cap_face_dist(cap, (g.oy + (y1 + 1) * g.cell) - qy, 1));
    if (z0 > 0) d = fminf(d, cap_face_dist(cap, qz - (g.oz + z0 * g.cell), 2));
    if (z1 < g.nz - 1) d = fminf(d, cap_face_dist(cap, (g.oz + (z1 + 1) * g.cell) - qz, 2));
    bool cert;
    if (tile_mode && r == r && round <= 160) d = fminf(d, r * 0.999998f + g.eps);   // skipped tiles: everything in them is farther than r (the eps is taken off below)
    if (d == INF || !(r == r) || round > 160) { cert = true; d_unseen = INF; }
    else { d -= g.eps; d_unseen = d; cert = d > 0.f && b != QN_INF_KEY && key_d2(b) < d * d; }
    if (cert) { best_out = b; second_out = c; return; }
    // (found but not certified: the candidate's own radius - and always at least eps MORE than this round's, or a candidate within rounding noise of r, whose
    //  box face comes out a few ulps of the coordinates short, would repeat the same round until the round limit: one such entry cost a list pass 390 us)
    r = (b != QN_INF_KEY) ? fmaxf(sqrtf(key_d2(b)) * 1.000001f + g.eps, r + g.eps) : (r > 6.f * g.cell ? r + (2.f + round) * g.cell : 2.f * r + g.cell);   // far away: grow by cells, not by factors (the cap's width grows with sqrt(r^2 - o^2))
    r = fminf(r, r_cap);
  }
}

__device__ __forceinline__ void wave_search_single(const GridView& g, float qx, float qy, float qz, float r, const float r_cap,
                                                   unsigned long long& best_out, float& second_out, float& d_unseen, WaveLds* lds) {
  SingleStats none;
  wave_search_single(g, qx, qy, qz, r, r_cap, best_out, second_out, d_unseen, lds, none, false);
}

// ------------------------------------------------------------------ far queries that are neighbours: up to 16 per wave, ONE shared candidate stream
// The far leftovers of an unseeded pass come in the cell-sorted order of the source: consecutive list entries are spatial neighbours whose search balls (radius:
// several cells to tens of metres) overlap almost completely, and one-per-wave each of them pays the same dependent chain - tile table look-ups, point gathers -
// for the same tiles.  Here the `member` queries of a wave (lane l: query l & 15, candidate sub-slot l >> 4, as in wave_search<4>) share ONE tile-mode scan of the
// ball (c, R) around their bounding-box centre c, R >= r_i + |q_i - c|: every staged 64-candidate chunk is scored by all members from LDS.  Shell rounds as in
// wave_search_single.  Query i is certified when its best distance is below R - |q_i - c| (everything unscanned inside the grid is farther than R from c), or
// when nothing is left unscanned.  Exact for any spread of the members; efficient when the spread is small against the radii (the caller groups by that).
// Outputs per member (identical in its 4 lanes): best key, runner-up d2, lower bound on everything unscanned.  All 64 lanes must call.
__device__ __forceinline__ void wave_search_far16(const GridView& g, float qx_in, float qy_in, float qz_in, bool member_in, float r_in,
                                                  unsigned long long& best_out, float& second_out, float& d_unseen_out, WaveLds* lds) {
  const int lane = threadIdx.x & 63;
  const float INF = __int_as_float(0x7f800000);
  // ---- re-layout: the m members get 64 / m2 lanes each (m2 = m rounded up to a power of two): lane l works for the member of rank l & (m2 - 1) as candidate
  // sub-slot l / m2 - two members score 32 candidates per step each, sixteen members 4 (the caller's layout: query l & 15 in lanes l, l + 16, l + 32, l + 48)
  const unsigned long long mm = __ballot(member_in) & 0xffffull;     // member queries, by their sub-slot-0 lane
  const int m = __popcll(mm);
  int m2 = 1; while (m2 < m) m2 <<= 1;
  const int S = 64 / m2;
  const int my_rank = __popcll(mm & ((1ull << (lane & 15)) - 1ull)); // (meaningful in member lanes)
  wave_lds_fence();
  if (member_in && lane < 16) lds->seg_cid[my_rank] = (uint32_t)lane;
  wave_lds_fence();
  const int rk = lane & (m2 - 1);
  const bool has_q = rk < m;
  const int src_lane = has_q ? (int)lds->seg_cid[rk] : 0;
  const float qx = __shfl(qx_in, src_lane), qy = __shfl(qy_in, src_lane), qz = __shfl(qz_in, src_lane), r = __shfl(r_in, src_lane);
  const uint32_t sub = (uint32_t)(lane / m2);
  const float cqx = 0.5f * (wave_min_f(has_q ? qx : INF) + wave_max_f(has_q ? qx : -INF));
  const float cqy = 0.5f * (wave_min_f(has_q ? qy : INF) + wave_max_f(has_q ? qy : -INF));
  const float cqz = 0.5f * (wave_min_f(has_q ? qz : INF) + wave_max_f(has_q ? qz : -INF));
  const float off = has_q ? sqrtf(sqdist(qx, qy, qz, cqx, cqy, cqz)) * 1.000002f + 1e-30f : 0.f;      // >= |q - c|
  float R = wave_max_f(has_q ? r + off : 0.f);
  const CapBox cap = cap_of(g, cqx, cqy, cqz);
  if (cap.S > 0.f) R = fmaxf(R, sqrtf(cap.S) + g.cell);
  Best1 sink; sink.init();
  bool open = has_q;                                                 // not certified yet
  unsigned long long fin_b = QN_INF_KEY; float fin_s = INF, fin_d = INF;
  float done2 = -1.f; int ptx0 = 0, ptx1 = -1, pty0 = 0, pty1 = -1, ptz0 = 0, ptz1 = -1;
  for (int round = 0;; round++) {
    const float rx = cap_extent(g, cap, R, 0), ry = cap_extent(g, cap, R, 1), rz = cap_extent(g, cap, R, 2);
    int x0 = rfl(cell_coord(cqx - rx, g.ox, g.inv_cell, g.nx)), x1 = rfl(cell_coord(cqx + rx, g.ox, g.inv_cell, g.nx));
    int y0 = rfl(cell_coord(cqy - ry, g.oy, g.inv_cell, g.ny)), y1 = rfl(cell_coord(cqy + ry, g.oy, g.inv_cell, g.ny));
    int z0 = rfl(cell_coord(cqz - rz, g.oz, g.inv_cell, g.nz)), z1 = rfl(cell_coord(cqz + rz, g.oz, g.inv_cell, g.nz));
    x0 = (x0 >> 3) << 3; x1 = min(((x1 >> 3) << 3) + 7, g.nx - 1);   // whole tiles
    y0 = (y0 >> 2) << 2; y1 = min(((y1 >> 2) << 2) + 3, g.ny - 1);
    z0 = (z0 >> 2) << 2; z1 = min(((z1 >> 2) << 2) + 3, g.nz - 1);
    const int tx0 = x0 >> 3, ntr = (x1 >> 3) - tx0 + 1, ty0 = y0 >> 2, ntyr = (y1 >> 2) - ty0 + 1, tz0 = z0 >> 2;
    const int nseg = ntr * ntyr * ((z1 >> 2) - tz0 + 1);
    const float in2 = R * R * 1.000002f;
    bool left_out = false;                                           // a tile of the range lies beyond the ball
    for (int sb = 0; sb < nseg; sb += 64) {
      const int sidx = sb + lane;
      uint32_t s = 0, len = 0;
      if (sidx < nseg) {
        int t, rr; divmod_small(sidx, ntr, rr, t);
        int qz_, ry_; divmod_small(rr, ntyr, qz_, ry_);
        const int tzz = tz0 + qz_, tyy = ty0 + ry_, txx = tx0 + t;
        const uint32_t tile = ((uint32_t)tzz * g.nty + tyy) * g.ntx + txx;
        const float tb2 = tile_box_d2(g, txx, tyy, tzz, cqx, cqy, cqz);
        const bool seen = tb2 <= done2 && txx >= ptx0 && txx <= ptx1 && tyy >= pty0 && tyy <= pty1 && tzz >= ptz0 && tzz <= ptz1;
        if (tb2 > in2) left_out = true;
        else if (!seen) { s = g.cell_start[tile << 7]; len = g.cell_start[(tile + 1) << 7] - s; }
      }
      const uint32_t incl = wave_incl_scan_u32(len, lane);
      const uint32_t total = rdlane(incl, 63);
      if (total == 0) continue;
      if (g.dbg && lane == 0) atomicAdd(&g.dbg[11], total);
      wave_lds_fence();
      lds->seg_excl[lane] = incl - len; lds->seg_start[lane] = s;
      wave_lds_fence();
      for (uint32_t cb = 0; cb < total; cb += 64) {
        const uint32_t slot = cb + lane;
        const int j = chunk_segment(lds->tile_cid, cb, incl - len, incl);
        const uint32_t cnt = min(64u, total - cb);
        wave_lds_fence();
        if (slot < total) lds->tile[lane] = g.pts[lds->seg_start[j] + (slot - lds->seg_excl[j])];
        wave_lds_fence();
        for (uint32_t c = 0; c < cnt; c += 4u * (uint32_t)S) {       // S candidates per member and step, 4 steps per trip: the four LDS reads issued before any scoring
          float4 cp[4]; uint32_t ci[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { ci[u] = c + (uint32_t)(u * S) + sub; cp[u] = lds->tile[ci[u] & 63u]; }
#pragma unroll
          for (int u = 0; u < 4; u++) sink.consider(open && ci[u] < cnt, sqdist(qx, qy, qz, cp[u].x, cp[u].y, cp[u].z), __float_as_uint(cp[u].w));
        }
      }
    }
    if (g.dbg && lane == 0) { atomicAdd(&g.dbg[10], 1u); atomicAdd(&g.dbg[12], (uint32_t)nseg); if (round == 0) atomicAdd(&g.dbg[13], 1u); }
    done2 = in2; ptx0 = tx0; ptx1 = tx0 + ntr - 1; pty0 = ty0; pty1 = ty0 + ntyr - 1; ptz0 = tz0; ptz1 = z1 >> 2;
    const bool everything = !__any(left_out) && x0 == 0 && x1 == g.nx - 1 && y0 == 0 && y1 == g.ny - 1 && z0 == 0 && z1 == g.nz - 1;
    // the member's merged result so far (its S sub-slots hold disjoint candidates)
    unsigned long long b = sink.key;
    for (int o = m2; o < 64; o <<= 1) { const unsigned long long t = __shfl_xor(b, o); b = t < b ? t : b; }
    float sc = (sink.key == b) ? sink.second : key_d2(sink.key);
    if (sink.key == QN_INF_KEY) sc = INF;
    for (int o = m2; o < 64; o <<= 1) sc = fminf(sc, __shfl_xor(sc, o));
    float need = 0.f;
    if (open) {
      const float d = R * 0.999998f - off - g.eps;                   // everything unscanned (inside the grid) is farther than R from c
      bool cert;
      if (everything || round > 160 || !(R == R)) { cert = true; fin_d = INF; }
      else { cert = d > 0.f && b != QN_INF_KEY && key_d2(b) < d * d; fin_d = fmaxf(d, 0.f); }
      if (cert) { fin_b = b; fin_s = sc; open = false; }
      else if (b != QN_INF_KEY) need = sqrtf(key_d2(b)) * 1.000002f + 2.f * g.eps + off;      // the ball that proves this member's candidate
    }
    if (!__any(open)) break;
    const float want = wave_max_f(need);
    const float grown = want > 0.f ? want : (R > 6.f * g.cell ? R + (2.f + round) * g.cell : 2.f * R + g.cell);
    R = fmaxf(grown, R * 1.0001f + g.eps);
  }
  // back to the caller's layout: member of rank k reads lane k (sub-slot 0 of that rank)
  const unsigned long long ob = __shfl(fin_b, my_rank & 63); const float os = __shfl(fin_s, my_rank & 63), od = __shfl(fin_d, my_rank & 63);
  if (member_in) { best_out = ob; second_out = os; d_unseen_out = od; }
}

// ------------------------------------------------------------------ pass B, k-NN: one query per LANE
// Same ball logic, per lane (divergent; only the rare k-NN leftovers of the covariance stage use it).
template <int KMAX>
__device__ __forceinline__ void lane_ball_knn(const GridView& g, float qx, float qy, float qz, float r, BestK<KMAX>& sink) {
  for (int round = 0;; round++) {
    sink.reset();
    const int bx0 = cell_coord(qx - r, g.ox, g.inv_cell, g.nx), bx1 = cell_coord(qx + r, g.ox, g.inv_cell, g.nx);
    const int by0 = cell_coord(qy - r, g.oy, g.inv_cell, g.ny), by1 = cell_coord(qy + r, g.oy, g.inv_cell, g.ny);
    const int bz0 = cell_coord(qz - r, g.oz, g.inv_cell, g.nz), bz1 = cell_coord(qz + r, g.oz, g.inv_cell, g.nz);
    const bool all = bx0 == 0 && by0 == 0 && bz0 == 0 && bx1 == g.nx - 1 && by1 == g.ny - 1 && bz1 == g.nz - 1;
    for (int rz = bz0; rz <= bz1; rz++) for (int ry = by0; ry <= by1; ry++) for (int tx = bx0 >> 3; tx <= (bx1 >> 3); tx++) {
      const int xa = max(bx0, tx << 3), xb = min(bx1, (tx << 3) + 7);
      const uint32_t k0 = cell_key(g, xa, ry, rz);
      const uint32_t s = g.cell_start[k0], e = g.cell_start[k0 + (xb - xa) + 1];
      for (uint32_t i = s; i < e; i++) {
        const float4 p = g.pts[i];
        const unsigned long long key = pack_key(sqdist(qx, qy, qz, p.x, p.y, p.z), __float_as_uint(p.w));
        if (key < sink.w) { sink.insert(key); sink.refresh(); }                 // per-lane sorted insert
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
