// qn_engine.hip - host side of the C-ABI declared in include/qn_engine.h: context, HBM-resident
// buffers, kernel sequencing on ONE hipStream per context.  No CPU fallback: every entry point
// fails with QN_ERR_NO_DEVICE / QN_ERR_HIP when there is no usable gfx950 device.
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <atomic>
#include <thread>
#include <chrono>
#include "qn_instances.h"      // heavy template kernels: declared here, compiled in qn_inst.hip (one TU per group)
#include "qn_context.h"
#include "qn_selftest.cuh"
#include "qn_pool.h"

using namespace qn;

#define HIPCHK(ctx, call)                                                                 \
  do { hipError_t e_ = (call);                                                            \
       if (e_ != hipSuccess) { (ctx)->set_error(#call, e_, __LINE__); return QN_ERR_HIP; } \
  } while (0)

void qn_ctx::set_error(const char* what, hipError_t e, int line) {
  char buf[512]; snprintf(buf, sizeof(buf), "%s -> %s (qn_engine.hip:%d)", what, hipGetErrorString(e), line);
  last_error = buf;
}

// ------------------------------------------------------------------ profiling (bench roofline leg)
void qn_ctx::prof_begin(int family, int count) {
  if (!prof_on) return;
  if (prof_open) { prof_depth++; return; }      // spans do not nest: an inner scope is part of the span that is open (its family is the outer one's)
  ProfSpan sp; sp.family = family; sp.count = count;
  if (hipEventCreate(&sp.a) != hipSuccess) return;
  if (hipEventCreate(&sp.b) != hipSuccess) { hipEventDestroy(sp.a); return; }
  hipEventRecord(sp.a, stream);
  spans.push_back(sp);
  prof_open = true; prof_depth = 0;
}
void qn_ctx::prof_end() {
  if (!prof_on || spans.empty() || !prof_open) return;
  if (prof_depth > 0) { prof_depth--; return; }
  prof_open = false;
  hipEventRecord(spans.back().b, stream);
}
void qn_ctx::prof_collect() {
  for (auto& sp : spans) {
    float ms = 0; if (hipEventSynchronize(sp.b) == hipSuccess && hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) { stats[sp.family].total_ms += ms; stats[sp.family].launches += sp.count; }
    hipEventDestroy(sp.a); hipEventDestroy(sp.b);
  }
  spans.clear(); prof_open = false; prof_depth = 0;
}
struct ProfScope { qn_ctx* c; ProfScope(qn_ctx* c_, int fam) : c(c_) { c->prof_begin(fam); } ~ProfScope() { c->prof_end(); } };

// ------------------------------------------------------------------ lifetime
extern "C" const char* qn_status_str(int s) {
  switch (s) {
    case QN_OK: return "ok"; case QN_ERR_INVALID_ARG: return "invalid argument"; case QN_ERR_EMPTY_CLOUD: return "empty cloud";
    case QN_ERR_CAPACITY: return "cloud exceeds context capacity"; case QN_ERR_NOT_READY: return "clouds/covariances not set";
    case QN_ERR_HIP: return "HIP runtime error"; case QN_ERR_NO_DEVICE: return "no gfx950 device (no CPU fallback)";
  }
  return "unknown status";
}
extern "C" const char* qn_last_error(const qn_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

extern "C" void qn_gicp_default_params(qn_gicp_params* p) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->k_correspondences = 20; p->max_iterations = 64; p->max_corr_dist = (double)FLT_MAX;
  p->transformation_epsilon = 5e-4; p->rotation_epsilon = 2e-3; p->optimizer = QN_OPT_LM;
  p->lm_max_iterations = 10; p->lm_init_lambda_factor = 1e-9; p->force_iterations = 0;
  p->ransac_iterations = 0; p->ransac_outlier_threshold = 0.05; p->euclidean_fitness_epsilon = -DBL_MAX;
}

// Device memory of a context comes from ONE slab (a single hipMalloc, 2 MiB aligned pieces of it handed out below): fifty separate allocations made late in a
// process - after another allocator (PyTorch's, a host's own) has fragmented the address space - ended up on small pages, and the gather-heavy kernels paid for
// it in TLB misses: the same 4-in-flight benchmark gave 1975 registrations/s with the contexts created after the point clouds and 2260 with them created first.
struct Slab {
  char* base = nullptr; size_t off = 0, cap = 0;
  template <class T> void take(T*& p, size_t bytes) {
    const size_t a = (off + 255) & ~(size_t)255;
    if (base) p = (T*)(base + a);
    off = a + bytes;
  }
};
static void carve_cloud(qn_ctx* c, Slab& sl, CloudBuf& b) {
  sl.take(b.raw, sizeof(float4) * c->max_points);
  sl.take(b.sorted, sizeof(float4) * c->max_points);
  sl.take(b.sorted_tmp, sizeof(float4) * c->max_points);
  sl.take(b.cell_of_pt, sizeof(uint32_t) * c->max_points);
  sl.take(b.cell_start, sizeof(uint32_t) * ((size_t)c->max_cells + 1));
  sl.take(b.counts, sizeof(uint32_t) * ((size_t)c->max_cells + 1));
  sl.take(b.nrm, sizeof(double) * 3 * c->max_points);
  sl.take(b.dims, sizeof(GridDims));
}

static int ctx_create(int device, uint32_t max_points, hipStream_t shared_stream, qn_ctx** out);
extern "C" int qn_ctx_create(int device, uint32_t max_points, qn_ctx** out) { return ctx_create(device, max_points, nullptr, out); }
// shared_stream: the context is a LANE of a batch context and works on its owner's stream (no stream of its own: HIP deals streams to a handful of hardware
// queues in creation order, and idle extra streams make working streams share queues)
static int ctx_create(int device, uint32_t max_points, hipStream_t shared_stream, qn_ctx** out) {
  if (!out || max_points == 0) return QN_ERR_INVALID_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return QN_ERR_NO_DEVICE;
  qn_ctx* c = new qn_ctx();
  c->device = device; c->max_points = max_points;
  // cell table capacity: ~16 cells per point, within [256 Ki, 8 Mi] cells
  size_t mc = (size_t)max_points * 16; if (mc < (1u << 18)) mc = 1u << 18; if (mc > (1u << 23)) mc = 1u << 23;
  c->max_cells = (uint32_t)mc;
  qn_gicp_default_params(&c->params);
  int rc = QN_OK;
  auto fail = [&](int code) { qn_ctx_destroy(c); return code; };
  if (hipSetDevice(device) != hipSuccess) return fail(QN_ERR_NO_DEVICE);
  if (shared_stream) { c->stream = shared_stream; c->owns_stream = false; c->is_lane = true; }
  else if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return fail(QN_ERR_HIP);
#define CA(call) if ((call) != hipSuccess) return fail(QN_ERR_HIP)
  auto carve = [&](Slab& sl) {
    carve_cloud(c, sl, c->cloud[0]);
    carve_cloud(c, sl, c->cloud[1]);
    sl.take(c->staging, (size_t)max_points * 32);
    sl.take(c->scan_sums, sizeof(uint32_t) * (c->max_cells / (QN_BLOCK * QN_SCAN_ITEMS) + 2));
    sl.take(c->state, 2 * sizeof(GicpState));
    sl.take(c->partials, 2 * sizeof(double) * (QN_ACC_MAX_BLOCKS + 8) * QN_NPART);
    sl.take(c->nn_idx, sizeof(int32_t) * max_points);
    sl.take(c->knn_idx, sizeof(int32_t) * (size_t)max_points * 32);
    sl.take(c->nn_ref, sizeof(float4) * max_points);
    sl.take(c->nrm_s_sorted, sizeof(double) * 3 * max_points);
    sl.take(c->tgt_rec, sizeof(TargetRec) * max_points);
    sl.take(c->fit_psum, sizeof(double) * (QN_ACC_MAX_BLOCKS + 8));
    sl.take(c->fit_pcnt, sizeof(uint32_t) * (QN_ACC_MAX_BLOCKS + 8));
    sl.take(c->trace, sizeof(qn_iter_trace) * QN_MAX_TRACE);
    sl.take(c->corr, sizeof(int32_t) * max_points);
    sl.take(c->sqd, sizeof(float) * max_points);
    sl.take(c->sqd_fit, sizeof(float) * max_points);
    sl.take(c->far_cand, sizeof(int32_t) * QN_FAR_M * (size_t)max_points);
    sl.take(c->far_cand_ref, sizeof(float4) * max_points);
    sl.take(c->far_cand_b, sizeof(float2) * max_points);
    sl.take(c->far_req, sizeof(unsigned long long) * (max_points / 64 + 2));
    sl.take(c->far_stats, 4 * sizeof(uint32_t));
    sl.take(c->far_rows, sizeof(double) * QN_FAR_BLOCKS * QN_NPART);
    sl.take(c->fb_list, sizeof(uint2) * max_points);
    sl.take(c->big_list, sizeof(uint2) * max_points);
    sl.take(c->fb_count2, 4 * sizeof(uint32_t));
    sl.take(c->aligned, sizeof(float4) * max_points);
    sl.take(c->pose_tmp, sizeof(double) * 16);
    sl.take(c->guess_tmp, sizeof(float) * 16);
    // scratch of the second stream (TargetScope): scan sums, k-NN lists and index table, bounding box; then the persistent kernel's hand-off buffers
    sl.take(c->scan_sums2, sizeof(uint32_t) * (c->max_cells / (QN_BLOCK * QN_SCAN_ITEMS) + 2));
    sl.take(c->fb_list2, sizeof(uint2) * max_points);
    sl.take(c->big_list2, sizeof(uint2) * max_points);
    sl.take(c->fb_count2b, 4 * sizeof(uint32_t));
    sl.take(c->knn_idx2, sizeof(int32_t) * (size_t)max_points * 32);
    sl.take(c->staging2, (size_t)max_points * 32);
    sl.take(c->bbox_acc, 2 * (QN_BBOX_MAX_BLOCKS + 1) * sizeof(BBoxAcc)); c->bbox_acc2 = c->bbox_acc ? c->bbox_acc + (QN_BBOX_MAX_BLOCKS + 1) : nullptr;      // [0]: ticket, [1 + b]: block b's box
    sl.take(c->scan_status, sizeof(unsigned long long) * 2 * (c->max_cells / (QN_BLOCK * QN_SCAN_ITEMS) + 2)); c->scan_status2 = c->scan_status ? c->scan_status + (c->max_cells / (QN_BLOCK * QN_SCAN_ITEMS) + 2) : nullptr;
    sl.take(c->pg_rows, sizeof(unsigned long long) * 3 * QN_PERSIST_ROWS * QN_PERSIST_RSTRIDE);
    sl.take(c->pg_bc, sizeof(unsigned long long) * 64);
    sl.take(c->pg_fit, sizeof(unsigned long long) * (QN_PERSIST_MAX_BLOCKS + 1) * 4);
    sl.take(c->pg_status, 4 * sizeof(uint32_t));
    sl.take(c->tail_ticket, 4 * sizeof(uint32_t));
  };
  { Slab dry; carve(dry);                                             // pass 1: the size; pass 2: the pointers
    Slab sl; sl.cap = (dry.off + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    CA(hipMalloc(&c->slab, sl.cap)); sl.base = (char*)c->slab; carve(sl); }
  CA(hipMemsetAsync(c->far_stats, 0, 4 * sizeof(uint32_t), c->stream));
  CA(hipHostMalloc(&c->result_host, sizeof(ResultBlock), hipHostMallocDefault));
  for (int w = 0; w < 2; w++) { CA(hipHostMalloc(&c->cloud[w].dims_host, sizeof(GridDims), hipHostMallocDefault)); memset(c->cloud[w].dims_host, 0, sizeof(GridDims)); }
  CA(hipHostMalloc(&c->scalar_host, 64 * sizeof(double), hipHostMallocDefault));
  // scratch of the second stream (TargetScope): scan sums, k-NN lists and index table, bounding box
  CA(hipHostMalloc(&c->pg_status_host, 4 * sizeof(uint32_t), hipHostMallocDefault));
  CA(hipMemsetAsync(c->pg_rows, 0xFF, sizeof(unsigned long long) * 3 * QN_PERSIST_ROWS * QN_PERSIST_RSTRIDE, c->stream));      // every slot = QN_PERSIST_SENTINEL
  CA(hipMemsetAsync(c->pg_bc, 0, sizeof(unsigned long long) * 64, c->stream));
  CA(hipMemsetAsync(c->pg_fit, 0, sizeof(unsigned long long) * (QN_PERSIST_MAX_BLOCKS + 1) * 4, c->stream));
  CA(hipMemsetAsync(c->pg_status, 0, 4 * sizeof(uint32_t), c->stream));
  CA(hipMemsetAsync(c->tail_ticket, 0, 4 * sizeof(uint32_t), c->stream));      // the last-block ticket of the controller tail: handed back at zero by every launch that uses it
  CA(hipMemsetAsync(c->state, 0, 2 * sizeof(GicpState), c->stream));
  hipLaunchKernelGGL(k_bbox_acc_init, dim3(1), dim3(256), 0, c->stream, c->bbox_acc, 2 * (QN_BBOX_MAX_BLOCKS + 1));
  CA(hipMemsetAsync(c->fb_count2, 0, 4 * sizeof(uint32_t), c->stream)); CA(hipMemsetAsync(c->fb_count2b, 0, 4 * sizeof(uint32_t), c->stream));      // the k-NN list counters: every covariance stage hands them back at zero (k_cov_from_idx)
  CA(hipMemsetAsync(c->scan_status, 0, sizeof(unsigned long long) * 2 * (c->max_cells / (QN_BLOCK * QN_SCAN_ITEMS) + 2), c->stream));
  for (int w = 0; w < 2; w++) CA(hipMemsetAsync(c->cloud[w].counts, 0, sizeof(uint32_t) * ((size_t)c->max_cells + 1), c->stream));      // the cell counters are handed back at zero by every build (k_scatter)
  // the k-NN index tables start as `no neighbour` (-1): a cloud with non-finite coordinates is an empty grid on the device - the selection kernels write nothing - and the
  // covariance kernel behind them gathers through the table as it is; a fresh allocation may hold anything (round 6: a GPU memory fault on the developer entry point qn_gicp_knn)
  CA(hipMemsetAsync(c->knn_idx, 0xff, sizeof(int32_t) * (size_t)max_points * 32, c->stream));
  if (c->knn_idx2) CA(hipMemsetAsync(c->knn_idx2, 0xff, sizeof(int32_t) * (size_t)max_points * 32, c->stream));
  CA(hipStreamSynchronize(c->stream));
#undef CA
  { // the persistent align kernel needs all of its blocks resident at once (up to QN_PERSIST_MAX_BLOCKS + 1 blocks of 512 threads): on a smaller or compute-partitioned
    // device it would only spin until its time-out - measured here once per context, not assumed
    int per_cu = 0; hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&k_align_persist<QN_PERSIST_TB, false>), QN_PERSIST_TB, 0) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; prop.multiProcessorCount = 0; }
    c->persist_resident_blocks = per_cu * prop.multiProcessorCount;
    c->persist_fits = c->persist_resident_blocks >= QN_PERSIST_MAX_BLOCKS + 1;
  }
  *out = c;
  return QN_OK;
}

extern "C" void qn_ctx_destroy(qn_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->stream2) hipStreamSynchronize(c->stream2);                     // a target may still be in preparation there (TargetScope)
  for (qn_ctx* l : c->lanes) if (l != c) qn_ctx_destroy(l);
  c->lanes.clear();
  if (c->args_h) hipHostFree(c->args_h);
  hipFree(c->args_d);
  c->prof_collect();
  hipFree(c->slab);                                                      // every per-context device buffer of the GICP path lives in it (qn_ctx_create)
  hipFree(c->q_mm_c); hipFree(c->q_mm_q); hipFree(c->q_mm_qn); hipFree(c->q_mm_L); hipFree(c->q_mm_table); hipFree(c->q_mm_table_q); hipFree(c->q_mm_pairs); hipFree(c->q_mm_cnt); hipFree(c->q_mm_vkeys); hipFree(c->q_mm_vcnt);
  for (int w = 0; w < 2; w++) { hipFree(c->q_normals[w]); hipFree(c->q_spfh[w]); hipFree(c->q_fpfh_s[w]); hipFree(c->q_fpfh[w]); hipFree(c->q_key[w]); hipFree(c->q_pair[w]); hipFree(c->q_pair_hash[w]); }
  hipFree(c->q_hit); hipFree(c->q_list); hipFree(c->q_sel); hipFree(c->q_pairs); hipFree(c->q_counts); hipFree(c->q_T); hipFree(c->q_mean); hipFree(c->q_mean_psum);
  if (c->q_host) hipHostFree(c->q_host);
  hipFree(c->c2f_src); hipFree(c->c2f_dst);
  hipFree(c->dbg_knn_idx); hipFree(c->dbg_knn_d2); hipFree(c->dbg_counters);
  hipFree(c->v_corr); hipFree(c->v_nn_idx); hipFree(c->v_sqd); hipFree(c->v_nn_ref); hipFree(c->v_counters);
  if (c->result_host) hipHostFree(c->result_host);
  for (int w = 0; w < 2; w++) if (c->cloud[w].dims_host) hipHostFree(c->cloud[w].dims_host);
  if (c->scalar_host) hipHostFree(c->scalar_host);
  hipFree(c->pg_clk); if (c->list_probe) hipFree(c->list_probe); if (c->pg_status_host) hipHostFree(c->pg_status_host);
  if (c->ev_pair) hipEventDestroy(c->ev_pair);
  for (qn_ctx::PinZone* z : {&c->pin_up, &c->pin_up2}) { if (z->ev) hipEventDestroy(z->ev); if (z->h) hipHostFree(z->h); }
  if (c->stream2) { hipStreamSynchronize(c->stream2); hipStreamDestroy(c->stream2); }
  if (c->stream && c->owns_stream) hipStreamDestroy(c->stream);
  delete c;
}

static int clouds_valid(qn_ctx* c);
extern "C" void* qn_ctx_stream(qn_ctx* c) { return c ? (void*)c->stream : nullptr; }
extern "C" int qn_ctx_synchronize(qn_ctx* c) {
  if (!c) return QN_ERR_INVALID_ARG;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->stream2) HIPCHK(c, hipStreamSynchronize(c->stream2));          // a target still being prepared (TargetScope)
  c->prof_collect();
  return clouds_valid(c);                                              // (every synchronising entry point reports a cloud with non-finite coordinates: the setters do not wait for the GPU)
}

extern "C" int qn_gicp_set_params(qn_ctx* c, const qn_gicp_params* p) {
  if (!c || !p) return QN_ERR_INVALID_ARG;
  if (p->k_correspondences < 1 || p->k_correspondences > 32 || p->max_iterations < 0 || p->lm_max_iterations < 1 ||
      (p->optimizer != QN_OPT_LM && p->optimizer != QN_OPT_GN) || !(p->max_corr_dist > 0)) return QN_ERR_INVALID_ARG;
  if (p->k_correspondences != c->params.k_correspondences) { c->cloud[0].has_cov = c->cloud[1].has_cov = false; }
  c->params = *p;
  return QN_OK;
}

extern "C" int qn_gicp_get_params(const qn_ctx* c, qn_gicp_params* p) { if (!c || !p) return QN_ERR_INVALID_ARG; *p = c->params; return QN_OK; }

static GicpConfig make_cfg(const qn_ctx* c) {
  GicpConfig g; const qn_gicp_params& p = c->params;
  g.k = p.k_correspondences; g.max_iterations = p.max_iterations; g.optimizer = p.optimizer; g.lm_max_iterations = p.lm_max_iterations;
  g.force_iterations = p.force_iterations; g.max_corr_dist_sq = p.max_corr_dist * p.max_corr_dist;
  g.transformation_epsilon = p.transformation_epsilon; g.rotation_epsilon = p.rotation_epsilon; g.lm_init_lambda_factor = p.lm_init_lambda_factor;
  return g;
}

// double-buffered optimiser state / partial rows (qn_gicp_kernels.cuh): generation c->gen
static inline GicpState* st_cur(qn_ctx* c) { return c->state + (c->gen & 1u); }
static inline GicpState* st_nxt(qn_ctx* c) { return c->state + ((c->gen + 1u) & 1u); }
static inline double* part_cur(qn_ctx* c) { return c->partials + (size_t)(c->gen & 1u) * (QN_ACC_MAX_BLOCKS + 8) * QN_NPART; }
static inline double* part_nxt(qn_ctx* c) { return c->partials + (size_t)((c->gen + 1u) & 1u) * (QN_ACC_MAX_BLOCKS + 8) * QN_NPART; }

// ------------------------------------------------------------------ setInputSource / setInputTarget
// second stream of the pair pipeline, created on first use (its scratch is allocated with the context): a context that only ever works
// inside a batch (several contexts in flight) never gets one - HIP deals streams to a handful of hardware queues in creation order, and idle extra streams made working
// streams share queues (batch of 64 pairs: 1900 -> 1460 pairs/s)
static bool pair_pipeline_ready(qn_ctx* c) {
  if (c->stream2) return true;
  if (c->pair_failed || !c->knn_idx2) return false;
  const bool ok = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&c->ev_pair, hipEventDisableTiming) == hipSuccess;
  if (!ok) { c->pair_failed = true; if (c->stream2) { hipStreamDestroy(c->stream2); c->stream2 = nullptr; } (void)hipGetLastError(); }
  return ok;
}
// The TARGET cloud is prepared (upload, grid, k-NN + covariances) on the second stream with its own scratch - scan sums, k-NN lists and
// index table, bounding box - while the source's k-NN selection, enqueued by the previous call, is still running on the first stream:
// the target's upload, bounding-box round trip and grid build cost no wall time and its k-NN fills the half-empty second round of
// the source's.  This is the reference's own call order (setInputSource, calculateSourceCovariances, setInputTarget,
// calculateTargetCovariances, align: loop_closure.cpp:120-124), so the shim gets it without asking.  Everything that reads the target
// on the first stream joins first (join_target).  Off while profiling (spans are timed on the first stream), inside Quatro (its
// kernels read both grids on the first stream right away) and for contexts that work in a batch.
static void swap_scratch(qn_ctx* c) {
  std::swap(c->stream, c->stream2); std::swap(c->scan_sums, c->scan_sums2); std::swap(c->fb_list, c->fb_list2); std::swap(c->big_list, c->big_list2);
  std::swap(c->fb_count2, c->fb_count2b); std::swap(c->knn_idx, c->knn_idx2);
  std::swap(c->bbox_acc, c->bbox_acc2); std::swap(c->scan_status, c->scan_status2);
  std::swap(c->pin_up, c->pin_up2);
  std::swap(c->staging, c->staging2);       // (setInputSource no longer waits for its pack kernel: the target's upload must not land in the source's landing zone)
}
struct TargetScope {                    // RAII: the body of set_cloud / compute_cov runs with the second stream's scratch; the event marks its end
  qn_ctx* c; bool on;
  TargetScope(qn_ctx* c_, bool on_) : c(c_), on(on_) { if (on) swap_scratch(c); }
  ~TargetScope() { if (on) { (void)hipEventRecord(c->ev_pair, c->stream); swap_scratch(c); c->tgt_pending = true; } }
};
static int join_target(qn_ctx* c) {
  if (!c->tgt_pending) return QN_OK;
  c->tgt_pending = false;
  return hipStreamWaitEvent(c->stream, c->ev_pair, 0) == hipSuccess ? QN_OK : QN_ERR_HIP;
}

// K1: pack + bbox + grid numbers (on the device) -> count -> exclusive scan -> scatter.  No host round trip: the kernels read the numbers from
// device memory (GridView::dims, grid_resolve); the table-sized scan uses the grid of the largest table and leaves early.  What the host
// needs to know - "the cloud held non-finite coordinates" - arrives with the pinned mirror at the next synchronisation (clouds_valid).
// (arguments of the five launches for one cloud: the classic path launches them one by one, the batched path puts the entries of every cloud of a batch into ONE launch each)
struct GridLaunch { PackBBoxK::Args pack; uint32_t pack_nb; CellCountK::Args count; ScanLookbackK::Args scan; uint32_t scan_nb; ScatterK::Args scat; StableCellsK::Args stab; bool stable; uint32_t nb; };
static GridLaunch prep_grid(qn_ctx* c, CloudBuf& b, const char* dsrc, uint32_t stride, bool lanes = false) {
  const uint32_t n = b.n;
  GridView& g = b.grid;
  memset(&g, 0, sizeof(g));
  g.pts = b.sorted; g.cell_start = b.cell_start; g.dbg = c->dbg_counters; g.n = n; g.dims = b.dims;
  GridLaunch L;
  L.nb = (n + QN_BLOCK - 1) / QN_BLOCK;
  L.scan_nb = (c->max_cells + QN_BLOCK * QN_SCAN_ITEMS - 1) / (QN_BLOCK * QN_SCAN_ITEMS);
  if (((++c->build_epoch) & 0x3fffffffu) == 0u) ++c->build_epoch;     // (0 = the tag of the zero-initialised status words)
  // (a lone build wants few blocks - one contended ticket each; in a batched launch the clouds run side by side and the pack pass is bandwidth: all the slots)
  L.pack_nb = std::min<uint32_t>(L.nb, (uint32_t)std::min(lanes ? QN_BBOX_MAX_BLOCKS : c->bbox_blocks, QN_BBOX_MAX_BLOCKS));
  L.pack = PackBBoxK::Args{dsrc, stride, n, b.raw, c->bbox_acc, c->max_cells, c->cell_override, b.dims, b.dims_host};
  L.count = CellCountK::Args{b.raw, n, g, b.counts, b.cell_of_pt};
  L.scan = ScanLookbackK::Args{b.counts, b.dims, b.cell_start, c->scan_status, c->build_epoch, n};
  L.stable = c->stable_cells;
  L.scat = ScatterK::Args{b.raw, n, b.cell_of_pt, b.cell_start, b.counts, L.stable ? b.sorted_tmp : b.sorted};
  L.stab = StableCellsK::Args{b.sorted_tmp, n, b.cell_of_pt, b.cell_start, b.sorted};
  b.has_grid = true; b.has_cov = false;
  return L;
}
static int build_grid(qn_ctx* c, CloudBuf& b, const char* dsrc, uint32_t stride) {
  hipStream_t s = c->stream;
  const GridLaunch L = prep_grid(c, b, dsrc, stride);
  { ProfScope ps(c, QN_K_GRID_BUILD);
    // 5 launches, nothing else: the bounding-box accumulator and the cell counters clean up after themselves (the last block of the first kernel; k_scatter's
    // atomicSub hands every counter back at zero), the scan is single-pass, the numbers reach the host through the pinned mirror
    hipLaunchKernelGGL(k_pack_bbox_dims, dim3(L.pack_nb), dim3(QN_BLOCK), 0, s, L.pack.in, L.pack.stride, L.pack.n, L.pack.raw, L.pack.acc, L.pack.max_cells, L.pack.cell_override, L.pack.dims_dev, L.pack.dims_host);
    hipLaunchKernelGGL(k_cell_count, dim3(L.nb), dim3(QN_BLOCK), 0, s, L.count.pts, L.count.n, L.count.g, L.count.counts, L.count.cell_of_pt);
    hipLaunchKernelGGL(k_scan_lookback, dim3(L.scan_nb), dim3(QN_BLOCK), 0, s, L.scan.in, L.scan.dims, L.scan.out, L.scan.status, L.scan.epoch, L.scan.total);
    hipLaunchKernelGGL(k_scatter, dim3(L.nb), dim3(QN_BLOCK), 0, s, L.scat.pts, L.scat.n, L.scat.cell_of_pt, L.scat.cell_start, L.scat.counts, L.scat.sorted);
    if (L.stable) hipLaunchKernelGGL(k_stable_cells, dim3(L.nb), dim3(QN_BLOCK), 0, s, L.stab.in, L.stab.n, L.stab.cell_of_pt, L.stab.cell_start, L.stab.out); }
  HIPCHK(c, hipGetLastError());
  return QN_OK;
}
// after a synchronisation: did a cloud hold non-finite coordinates?  (is_dense == false clouds are not supported: the reference's KD-tree build would not survive them)
static int clouds_valid(qn_ctx* c) {
  for (int w = 0; w < 2; w++)
    if (c->cloud[w].has_grid && c->cloud[w].dims_host && c->cloud[w].dims_host->nonfinite) {
      c->last_error = w == 0 ? "source cloud contains non-finite coordinates (is_dense == false clouds are not supported)" : "target cloud contains non-finite coordinates (is_dense == false clouds are not supported)";
      c->cloud[w].has_grid = c->cloud[w].has_cov = false;
      return QN_ERR_INVALID_ARG;
    }
  return QN_OK;
}

// A HOST cloud into the context's device landing zone (c->staging; the caller has the right scratch set swapped in).  Page-locked caller memory is read by the DMA engine
// directly (asynchronously: the buffer must stay unchanged until the next synchronisation, include/qn_engine.h); pageable memory is packed by the CPU into the context's own
// page-locked zone first - consumed when this returns - and never handed to hipMemcpyAsync itself (see qn_ctx::PinZone).  -> *dsrc / stride: what the pack kernel reads.
static int upload_host_cloud(qn_ctx* c, const float* xyz, uint32_t n, uint32_t& stride, hipStream_t s, const char** dsrc) {
  hipPointerAttribute_t at;
  const bool pinned = hipPointerGetAttributes(&at, xyz) == hipSuccess && at.type == hipMemoryTypeHost;
  if (!pinned) (void)hipGetLastError();                             // (an unregistered pointer is an "invalid value" to the query: not an error of ours)
  if (pinned) {
    if (stride <= 32) { HIPCHK(c, hipMemcpyAsync(c->staging, xyz, (size_t)(n - 1) * stride + 12, hipMemcpyHostToDevice, s)); }      // one contiguous H2D; PointXYZI's intensity half is dropped by the pack kernel
    else { HIPCHK(c, hipMemcpy2DAsync(c->staging, 16, xyz, stride, 16, n, hipMemcpyHostToDevice, s)); stride = 16; }                  // fat point types: only the leading 16 B of each point cross PCIe
    *dsrc = (const char*)c->staging;
    return QN_OK;
  }
  qn_ctx::PinZone& z = c->pin_up;
  if (!z.h) { HIPCHK(c, hipHostMalloc((void**)&z.h, (size_t)c->max_points * 16, hipHostMallocDefault)); HIPCHK(c, hipEventCreateWithFlags(&z.ev, hipEventDisableTiming)); }
  if (z.busy) { HIPCHK(c, hipEventSynchronize(z.ev)); z.busy = false; }      // the zone's previous cloud has left it (it has, long ago, in every sequence but back-to-back setters)
  float* o = (float*)z.h; const char* in = (const char*)xyz;
  for (uint32_t i = 0; i < n; i++, in += stride, o += 3) { const float* p = (const float*)in; o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; }
  HIPCHK(c, hipMemcpyAsync(c->staging, z.h, (size_t)n * 12, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipEventRecord(z.ev, s)); z.busy = true;
  stride = 12; *dsrc = (const char*)c->staging;
  return QN_OK;
}

static int set_cloud(qn_ctx* c, int which, const float* xyz, uint32_t n, uint32_t stride, bool on_device) {
  if (!c || (which != 0 && which != 1)) return QN_ERR_INVALID_ARG;
  CloudBuf& b = c->cloud[which];
  b.n = 0; b.has_grid = b.has_cov = false;
  if (n == 0) return QN_ERR_EMPTY_CLOUD;
  if (!xyz || stride < 12 || (stride & 3)) return QN_ERR_INVALID_ARG;
  if (n > c->max_points) return QN_ERR_CAPACITY;
  HIPCHK(c, hipSetDevice(c->device));
  if (which == QN_SOURCE && join_target(c) != QN_OK) return QN_ERR_HIP;         // (a target still in flight reads the staging buffer)
  c->tgt_on_stream2 = which == QN_TARGET ? (c->pair_pipeline && !c->prof_on && !c->no_pipe && c->cloud[0].has_grid && pair_pipeline_ready(c)) : c->tgt_on_stream2;
  TargetScope scope(c, which == QN_TARGET && c->tgt_on_stream2);
  hipStream_t s = c->stream;
  const char* dsrc = (const char*)xyz;
  if (!on_device) { const int rc = upload_host_cloud(c, xyz, n, stride, s, &dsrc); if (rc != QN_OK) return rc; }
  b.n = n;
  return build_grid(c, b, dsrc, stride);
}

extern "C" int qn_gicp_set_source(qn_ctx* c, const float* xyz, uint32_t n, uint32_t stride) { return set_cloud(c, QN_SOURCE, xyz, n, stride, false); }
extern "C" int qn_gicp_set_target(qn_ctx* c, const float* xyz, uint32_t n, uint32_t stride) { return set_cloud(c, QN_TARGET, xyz, n, stride, false); }
extern "C" int qn_gicp_set_source_device(qn_ctx* c, const float* xyz, uint32_t n, uint32_t stride) { return set_cloud(c, QN_SOURCE, xyz, n, stride, true); }
extern "C" int qn_gicp_set_target_device(qn_ctx* c, const float* xyz, uint32_t n, uint32_t stride) { return set_cloud(c, QN_TARGET, xyz, n, stride, true); }

// ------------------------------------------------------------------ calculateSource/TargetCovariances
// the five launches of the histogram-selection path for one cloud (classic: one by one; batched: one entry per cloud in ONE launch each)
struct KnnLaunch { KnnHistArgs sel; uint32_t sel_nb; KnnHistArgs lst; uint32_t lst_nb; KnnSingleK::Args single; uint32_t single_nb; KnnCovArgs tail; uint32_t tail_nb; CovFromIdxK::Args cov; uint32_t cov_nb; };
static float knn_first_radius(const qn_ctx* c, const CloudBuf& b) {
  // first radius in cells.  A cloud of <= 65536 points is ONE round of waves (16 queries each, 4096 resident): the kernel then lasts as long as
  // its slowest wave, and a wave that has to retry with a doubled radius is 3-4x slower - a wider first radius (fewer retries) wins there
  // (30k points: 174 -> 116 us); beyond that the retries hide behind the next round of waves and the smaller radius wins (100k: 100 vs 131 us);
  // at 10k points the wider radius measured slower again (few waves, each with more candidates): 2.5 only between 16k and 64k.
  return -(c->margin_knn > 0.f ? c->margin_knn : (b.n > 16384u && b.n <= 65536u ? 2.5f : 2.0f));      // negative = in cells (the kernels know the cell edge, the host does not)
}
// lanes: the launch carries many clouds (k_lanes) - the list passes, which serve a handful of leftovers per cloud with wave-stride loops, get small grids
// (an empty block still costs the dispatcher ~4 ns: 2048 of them x 16 clouds were 140 us per launch)
static KnnLaunch prep_knn(qn_ctx* c, CloudBuf& b, int k, int32_t* kidx, float* kd2, bool lanes = false) {
  const float r0 = knn_first_radius(c, b);
  uint32_t* genc = c->fb_count2 + 1;
  KnnLaunch L;
  L.sel_nb = (b.n + QN_KNN_BLOCK / 4 - 1) / (QN_KNN_BLOCK / 4);     // 16 queries per wave
  if (lanes && c->knn_trips > 1) L.sel_nb = (L.sel_nb + (uint32_t)c->knn_trips - 1) / (uint32_t)c->knn_trips;      // batched launches: a wave serves several groups of 16 (its prologue - table entry, grid numbers - is a twentieth of a group's work)
  L.sel = KnnHistArgs{b.grid, k, r0, c->knn_single_all ? -1 : c->knn_rounds, kidx, kd2, c->fb_list, c->fb_count2, c->big_list, genc};
  L.lst_nb = std::min<uint32_t>((b.n + 63) / 64, lanes ? 96 : 512) * (QN_BLOCK / QN_KNN_BLOCK);
  L.lst = KnnHistArgs{b.grid, k, r0, 64, kidx, kd2, c->fb_list, c->fb_count2, c->big_list, genc};
  // far / overflowing queries (isolated points, sparse far field): one per wave; its leftovers -> the sorted-list kernel (fb_list is free again)
  L.single_nb = std::min<uint32_t>((b.n + 3) / 4, lanes ? 128 : 2048);
  L.single = KnnSingleK::Args{b.grid, k, kidx, kd2, c->big_list, genc, c->fb_list, c->fb_count2 + 2};
  L.tail_nb = std::min<uint32_t>((b.n + 63) / 64, lanes ? 64 : 1024);
  L.tail = KnnCovArgs{b.grid, b.raw, k, r0, 64, b.nrm, kidx, kd2, c->fb_list, c->fb_count2 + 2};
  L.cov_nb = (b.n + QN_BLOCK - 1) / QN_BLOCK;
  L.cov = CovFromIdxK::Args{b.raw, b.sorted, b.n, k, kidx, b.nrm, &b == &c->cloud[0] ? c->nrm_s_sorted : (double*)nullptr, &b == &c->cloud[0] ? (TargetRec*)nullptr : c->tgt_rec, c->fb_count2};   // + the optimiser ticks' layouts
  return L;
}
template <int KMAX>
static void launch_knn_cov(qn_ctx* c, CloudBuf& b, int k, int32_t* kidx, float* kd2) {
  hipStream_t s = c->stream;
  if (c->knn_hist) {                        // histogram selection (default); its leftovers -> hist list pass -> general sorted-list pass
    const KnnLaunch L = prep_knn(c, b, k, kidx, kd2);
    constexpr int HCAP = KMAX <= 24 ? 32 : 48;      // pass-2 list capacity: 32 keeps the selection kernel at 4 waves/SIMD
    { ProfScope sel(c, QN_K_KNN_SELECT);
      if (HCAP == 32 && !c->knn_mm) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_knn_hist<false, 32, false>), dim3(L.sel_nb), dim3(QN_KNN_BLOCK), 0, s, L.sel.g, L.sel.k, L.sel.r0, L.sel.max_rounds, L.sel.knn_idx, L.sel.knn_d2, L.sel.fb_list, L.sel.fb_count, L.sel.gen_list, L.sel.gen_count);
      else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_knn_hist<false, HCAP>), dim3(L.sel_nb), dim3(QN_KNN_BLOCK), 0, s, L.sel.g, L.sel.k, L.sel.r0, L.sel.max_rounds, L.sel.knn_idx, L.sel.knn_d2, L.sel.fb_list, L.sel.fb_count, L.sel.gen_list, L.sel.gen_count); }
    ProfScope ps(c, QN_K_KNN_COV);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_knn_hist<true, HCAP>), dim3(L.lst_nb), dim3(QN_KNN_BLOCK), 0, s, L.lst.g, L.lst.k, L.lst.r0, L.lst.max_rounds, L.lst.knn_idx, L.lst.knn_d2, L.lst.fb_list, L.lst.fb_count, L.lst.gen_list, L.lst.gen_count);
    hipLaunchKernelGGL(k_knn_single, dim3(L.single_nb), dim3(QN_BLOCK), 0, s, L.single.g, L.single.k, L.single.knn_idx, L.single.knn_d2, L.single.list, L.single.count, L.single.gen_list, L.single.gen_count);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_knn_cov<KMAX, true, 4>), dim3(L.tail_nb), dim3(QN_BLOCK), 0, s, L.tail.g, L.tail.raw, L.tail.k, L.tail.r0, L.tail.max_rounds, L.tail.cov, L.tail.knn_idx, L.tail.knn_d2, L.tail.fb_list, L.tail.fb_count);
    hipLaunchKernelGGL(k_cov_from_idx, dim3(L.cov_nb), dim3(QN_BLOCK), 0, s, L.cov.raw, L.cov.sorted, L.cov.n, L.cov.k, L.cov.knn_idx, L.cov.nrm, L.cov.nrm_sorted, L.cov.rec, L.cov.list_counts);
    return;
  }
  const float r0 = knn_first_radius(c, b);
  ProfScope ps(c, QN_K_KNN_COV);
  {                                         // sorted-list sink, 16 queries per wave x 4 candidate sub-slots (knn_hist = 0)
    const uint32_t nb = (b.n + QN_BLOCK / 4 - 1) / (QN_BLOCK / 4);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_knn_cov<KMAX, false, 4>), dim3(nb), dim3(QN_BLOCK), 0, s, b.grid, b.raw, k, r0, 2, b.nrm, kidx, kd2, c->fb_list, c->fb_count2);
  }
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_knn_cov<KMAX, true, 4>), dim3(std::min<uint32_t>((b.n + 63) / 64, 512)), dim3(QN_BLOCK), 0, s, b.grid, b.raw, k, r0, 64, b.nrm, kidx, kd2, c->fb_list, c->fb_count2);
  const uint32_t nbp = (b.n + QN_BLOCK - 1) / QN_BLOCK;
  hipLaunchKernelGGL(k_cov_from_idx, dim3(nbp), dim3(QN_BLOCK), 0, s, b.raw, b.sorted, b.n, k, kidx, b.nrm,
                     &b == &c->cloud[0] ? c->nrm_s_sorted : (double*)nullptr, &b == &c->cloud[0] ? (TargetRec*)nullptr : c->tgt_rec, c->fb_count2);   // + the optimiser ticks' layouts
}
static int compute_cov(qn_ctx* c, int which, int32_t* kidx, float* kd2) {
  if (!c || (which != 0 && which != 1)) return QN_ERR_INVALID_ARG;
  CloudBuf& b = c->cloud[which];
  if (!b.has_grid) return b.n == 0 ? QN_ERR_EMPTY_CLOUD : QN_ERR_NOT_READY;
  HIPCHK(c, hipSetDevice(c->device));
  const int k = c->params.k_correspondences;
  const bool on2 = which == QN_TARGET && c->tgt_on_stream2 && !c->prof_on && kidx == c->knn_idx && c->stream2 != nullptr;   // the grid was built there: stay behind it
  if (which == QN_TARGET && !on2 && join_target(c) != QN_OK) return QN_ERR_HIP;
  if (on2) kidx = c->knn_idx2;
  TargetScope scope(c, on2);
  if (k <= 16) launch_knn_cov<16>(c, b, k, kidx, kd2);
  else if (k <= 20) launch_knn_cov<20>(c, b, k, kidx, kd2);
  else if (k <= 24) launch_knn_cov<24>(c, b, k, kidx, kd2);
  else launch_knn_cov<32>(c, b, k, kidx, kd2);
  HIPCHK(c, hipGetLastError());
  b.has_cov = true;
  return QN_OK;
}
extern "C" int qn_gicp_compute_covariances(qn_ctx* c, int which) { return compute_cov(c, which, c->knn_idx, nullptr); }

// ------------------------------------------------------------------ align
// seeded = true: the previous iteration's NN indices are valid -> temporal tracking kernel; false -> full grid search
// tick = index of this NN pass within the align (list-pass grids shrink as the optimiser converges: the first search
// leaves ~10-20 % of the queries to the list passes, the first tracked pass most of them after the big initial pose
// step, later passes a handful - any grid is correct, the lists are walked wave-stride)
// (arguments of the grid pass and the list pass of one exact 1-NN search; `seeded` searches replace the grid pass by k_nn_track)
struct NnLaunch { NnSearchArgs grid; uint32_t grid_nb; NnSearchArgs list; uint32_t list_nb; bool group; bool lane; };      // lane: the first pass runs one query per lane (NnLaneK)
static NnLaunch prep_nn(qn_ctx* c, int mode /*0 align, 1 fitness*/, float* sqd_out, bool seeded, int tick, int cond) {
  CloudBuf &S = c->cloud[0], &T = c->cloud[1];
  const uint32_t nb = (S.n + QN_NN_BLOCK / 4 - 1) / (QN_NN_BLOCK / 4);   // grid passes: 16 queries per wave
  const uint32_t nb4 = (S.n + QN_BLOCK / 4 - 1) / (QN_BLOCK / 4);       // list passes: 4 waves per block
  const double thr2 = c->params.max_corr_dist * c->params.max_corr_dist;
  GicpState* st = st_cur(c);
  uint32_t* fbc = &st->fb_count; uint32_t* bgc = &st->big_count;
  uint32_t* far_stats = (mode == 0 && !seeded && c->count_far_now) ? c->far_stats : nullptr;      // the LAST unseeded pass before the host's look (the clouds are best aligned then)
  const bool wide = tick <= 2 || (mode == 0 && !seeded);                                              // every unseeded pass of an align leaves thousands of queries to the lists, whatever its index
  int big_blocks = wide ? c->big_blocks0 : 1024;                                                      // waves with one far query each (idle blocks exit at once)
  const uint32_t fbb = std::min<uint32_t>(nb4, wide ? (uint32_t)c->fb_blocks0 : 256u);                           // list pass: wave-stride over the leftovers
  const float r0 = -(tick == 0 && mode == 0 && c->margin_nn_t0 > 0.f ? c->margin_nn_t0 : c->margin_nn);      // negative = in cells
  const float big_ratio = c->big_ratio;
  NnOpt opt; opt.clear_ref = (mode == 0 && !seeded && c->clear_far_now) ? c->far_cand_ref : nullptr; opt.cond = cond;
  // far-list grouping (wave_search_far16): with several registrations in flight throughput counts and long far lists are shared (partial overlap: 1110 -> 1330
  // registrations/s; the aligned-scene headline is unchanged); a lone registration wants latency - its lists fit a few rounds of resident waves, one entry per
  // wave starts them all at once, and the leaner kernel keeps 6 blocks per CU (ms_per_align 0.552 against 0.570 / 0.609 with the grouped variant)
  const bool batch = c->persist_batch_off;
  opt.group = c->far_group >= 0 ? c->far_group : (batch ? 2048 : 0); opt.group_min = 0;      // (round 4, batched launches: 2048 - up to 16 neighbours per shared scan - measured +3 % over 4096; 8192 and up lose)
  // (a grouped far list is served ceil(length / group) entries per wave: `group` waves are all it can use - same entry -> wave assignment, three quarters fewer empty blocks)
  if (mode == 0 && opt.group > 0) big_blocks = std::min(big_blocks, std::max(64, (opt.group + QN_BLOCK / 64 - 1) / (QN_BLOCK / 64)));
  opt.probe = (mode == 0 && !seeded) ? c->list_probe : nullptr;
  opt.fb_small = (mode == 0 && !c->persist_batch_off) ? (uint32_t)c->list_small : 0u;      // (a batch member keeps the 16-per-wave lists: fewer wave-instructions per query, +2 % throughput)
  NnOpt opt0; opt0.clear_ref = nullptr; opt0.cond = 0; opt0.group = 0; opt0.group_min = 0; opt0.probe = nullptr; opt0.fb_small = 0;
  c->clear_far_now = mode == 0 && !seeded ? false : c->clear_far_now;
  const NnOpt& o = mode == 0 ? opt : opt0;
  NnLaunch L;
  // one query per lane for batch members (throughput: a quarter of the waves, a fifth of the instructions); a LONE registration keeps the cooperative pass - its 1563
  // one-per-lane waves are a single latency-bound round (48 us against 36-41)
  L.lane = (c->nn_lane > 0 || (c->nn_lane < 0 && c->persist_batch_off)) && mode == 0 && !seeded;
  L.grid_nb = L.lane ? (S.n + 255u) / 256u : nb; L.list_nb = fbb + (uint32_t)big_blocks; L.group = mode == 0 && opt.group > 0;
  L.grid = NnSearchArgs{S.grid, T.grid, st, thr2, r0, mode == 0 ? c->nn_rounds : 1, c->corr, sqd_out, c->nn_idx, c->nn_ref, c->fb_list, fbc, c->big_list, bgc, 0, big_ratio, far_stats, o};
  L.list = NnSearchArgs{S.grid, T.grid, st, thr2, r0, 64, c->corr, sqd_out, c->nn_idx, c->nn_ref, c->fb_list, fbc, c->big_list, bgc, big_blocks, big_ratio, far_stats, o};
  return L;
}
#define QN_NN_ARGS(A) A.src, A.tgt, A.st, A.thr2, A.r0, A.max_rounds, A.corr, A.sqd, A.nn_idx, A.nn_ref, A.fb_list, A.fb_count, A.big_list, A.big_count, A.big_blocks, A.big_ratio, A.far_stats, A.opt
static void enqueue_nn(qn_ctx* c, int mode /*0 align, 1 fitness*/, float* sqd_out, bool seeded, int tick = 0, int cond = 0) {
  hipStream_t s = c->stream;
  CloudBuf &S = c->cloud[0], &T = c->cloud[1];
  const uint32_t nbt = (S.n + QN_BLOCK - 1) / QN_BLOCK;               // tracking: one query per lane
  const NnLaunch L = prep_nn(c, mode, sqd_out, seeded, tick, cond);
  if (L.grid.opt.probe) (void)hipMemsetAsync(c->list_probe, 0, sizeof(unsigned long long) * 4 * 16384, s);
  const NnSearchArgs& G = L.grid; const NnSearchArgs& F = L.list;
  if (mode == 0) {
    { ProfScope ps(c, QN_K_NN_SEARCH);
      if (seeded) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nn_track<0>), dim3(nbt), dim3(QN_BLOCK), 0, s, S.grid, T.grid, T.raw, G.st, G.thr2, c->corr, sqd_out, c->nn_idx, c->nn_ref, c->fb_list, G.fb_count, c->big_list, G.big_count);
      else if (L.lane) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nn_lane<0>), dim3(L.grid_nb), dim3(256), 0, s, G);
      else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nn_search<0, false, QN_NN_BLOCK>), dim3(L.grid_nb), dim3(QN_NN_BLOCK), 0, s, QN_NN_ARGS(G)); }
    { ProfScope ps(c, QN_K_NN_FALLBACK);
      if (L.group) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nn_search<0, true, QN_BLOCK, true>), dim3(L.list_nb), dim3(QN_BLOCK), 0, s, QN_NN_ARGS(F));
      else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nn_search<0, true, QN_BLOCK>), dim3(L.list_nb), dim3(QN_BLOCK), 0, s, QN_NN_ARGS(F)); }
  } else {
    ProfScope ps(c, QN_K_FITNESS);
    if (seeded) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nn_track<1>), dim3(nbt), dim3(QN_BLOCK), 0, s, S.grid, T.grid, T.raw, G.st, G.thr2, c->corr, sqd_out, c->nn_idx, c->nn_ref, c->fb_list, G.fb_count, c->big_list, G.big_count);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nn_search<1, false, QN_NN_BLOCK>), dim3(L.grid_nb), dim3(QN_NN_BLOCK), 0, s, QN_NN_ARGS(G));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nn_search<1, true, QN_BLOCK>), dim3(L.list_nb), dim3(QN_BLOCK), 0, s, QN_NN_ARGS(F));
  }
}
// debug knob "verify_track": a fresh, unseeded search of the current pose into scratch buffers, compared query by query with what
// the tracked / bound-pruned pass just produced (tests/test_gpu_adversarial.py)
static void enqueue_verify(qn_ctx* c, bool fused) {
  hipStream_t s = c->stream;
  CloudBuf &S = c->cloud[0], &T = c->cloud[1];
  const uint32_t nb = (S.n + QN_NN_BLOCK / 4 - 1) / (QN_NN_BLOCK / 4), nb4 = (S.n + QN_BLOCK / 4 - 1) / (QN_BLOCK / 4);
  const double thr2 = c->params.max_corr_dist * c->params.max_corr_dist;
  GicpState* st = st_cur(c);
  uint32_t* fbc = &st->fb_count; uint32_t* bgc = &st->big_count;
  const float r0 = -c->margin_nn;
  hipLaunchKernelGGL(k_reset_lists, dim3(1), dim3(64), 0, s, st);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nn_search<0, false, QN_NN_BLOCK>), dim3(nb), dim3(QN_NN_BLOCK), 0, s, S.grid, T.grid, st, thr2, r0, c->nn_rounds, c->v_corr, c->v_sqd, c->v_nn_idx, c->v_nn_ref, c->fb_list, fbc, c->big_list, bgc, 0, c->big_ratio, (uint32_t*)nullptr, NnOpt{nullptr, 0, 0, 0, nullptr, 0});
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nn_search<0, true, QN_BLOCK>), dim3(std::min<uint32_t>(nb4, 512) + 4096), dim3(QN_BLOCK), 0, s, S.grid, T.grid, st, thr2, r0, 64, c->v_corr, c->v_sqd, c->v_nn_idx, c->v_nn_ref, c->fb_list, fbc, c->big_list, bgc, 4096, c->big_ratio, (uint32_t*)nullptr, NnOpt{nullptr, 0, 0, 0, nullptr, 0});
  hipLaunchKernelGGL(k_verify_nn, dim3((S.n + 255) / 256), dim3(256), 0, s, S.n, st, c->nn_idx, c->v_nn_idx, fused ? (const float*)nullptr : c->sqd, c->v_sqd, c->corr, c->v_corr, c->v_counters);
  hipLaunchKernelGGL(k_reset_lists, dim3(1), dim3(64), 0, s, st);
}
static uint32_t acc_blocks(const qn_ctx* c) { return std::min<uint32_t>((c->cloud[0].n + QN_BLOCK - 1) / QN_BLOCK, QN_ACC_MAX_BLOCKS); }
static uint32_t tick_ppt(const qn_ctx* c) { return std::max<uint32_t>(c->tick_ppt_min, (c->cloud[0].n + c->tick_tb * QN_ACC_MAX_BLOCKS - 1) / (c->tick_tb * QN_ACC_MAX_BLOCKS)); }
static uint32_t tick_rows(const qn_ctx* c) { const uint32_t per = c->tick_tb * tick_ppt(c); return (c->cloud[0].n + per - 1) / per; }      // partial rows of a tick: one per tick_tb x ppt source points
static uint32_t tick_rpb(const qn_ctx* c) { return c->persist_batch_off ? std::max(1u, c->tick_rpb) : 1u; }      // rows a block forms: batch members take fewer, longer blocks (the rows - every bit downstream - are the same)
static uint32_t tick_blocks(const qn_ctx* c) { const uint32_t rpb = tick_rpb(c); return (tick_rows(c) + rpb - 1) / rpb; }
// The controller step of a producer launch (controller_tail): generation g -> g + 1 inside the launch that wrote the rows.
static TailArgs tail_args(qn_ctx* c, int enabled, int rows, const LookArgs* look = nullptr) {
  TailArgs t; memset(&t, 0, sizeof(t));
  t.st_in = st_cur(c); t.st_out = st_nxt(c); t.cfg = make_cfg(c); t.trace = c->trace; t.ticket = c->tail_ticket; t.enabled = enabled; t.rows = rows;
  if (look) t.look = *look;
  return t;
}
// tail = true: an optimiser tick - the launch's last block runs the controller and the generation advances; false: the rows stay for a stand-alone k_solve (debug entry points)
static AccumulateK::Args prep_accumulate(qn_ctx* c, int cond, bool tail, const LookArgs* look = nullptr) {
  CloudBuf &S = c->cloud[0];
  const AccumulateK::Args a{S.raw, S.n, S.nrm, c->tgt_rec, c->corr, st_cur(c), part_cur(c), cond, tail_args(c, tail ? 1 : 0, (int)acc_blocks(c), look)};
  if (tail) { c->gen++; c->part_rows = -1; } else c->part_rows = (int)acc_blocks(c);
  return a;
}
static void enqueue_accumulate(qn_ctx* c, int cond, bool tail, const LookArgs* look = nullptr) {
  ProfScope ps(c, QN_K_ACCUMULATE);
  const uint32_t nb = acc_blocks(c);
  const AccumulateK::Args a = prep_accumulate(c, cond, tail, look);
  hipLaunchKernelGGL(k_accumulate, dim3(nb), dim3(QN_BLOCK), 0, c->stream, a);
}
// one controller step as its own launch: generation g -> g + 1 (k_solve)
static SolveArgs prep_solve(qn_ctx* c, int mode, int will_produce, const LookArgs* look) {
  LookArgs la; memset(&la, 0, sizeof(la)); if (look) la = *look;
  const SolveArgs a{st_cur(c), st_nxt(c), part_cur(c), c->part_rows, make_cfg(c), c->trace, mode, will_produce, la};
  c->gen++; c->part_rows = -1;                                      // consumed: whatever controller comes next (k_tick's prologue, the persistent kernel, another k_solve) must not step again
  return a;
}
static void enqueue_solve(qn_ctx* c, int mode, int will_produce, const LookArgs* look = nullptr) {
  ProfScope ps(c, QN_K_SOLVE);
  const SolveArgs a = prep_solve(c, mode, will_produce, look);
  if (c->tick_tb == 256) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_solve<256>), dim3(1), dim3(256), 0, c->stream, a.st_in, a.st_out, a.partials, a.rows, a.cfg, a.trace, a.mode, a.will_produce, a.look);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_solve<512>), dim3(1), dim3(512), 0, c->stream, a.st_in, a.st_out, a.partials, a.rows, a.cfg, a.trace, a.mode, a.will_produce, a.look);
}
// One "tick" of the device-side state machine = [controller step on the previous tick's partial rows] + [body under the new state].
// Tracked regime: ONE kernel (k_tick: controller in the prologue of every block, tracked NN + accumulation, LM trial passes included).
// mode 0: an optimiser tick (its last block steps the controller unless the far-query refresh kernels follow); mode 1: the closing pass (nothing to step)
static TickArgs tick_args(qn_ctx* c, int mode = 0) {
  CloudBuf &S = c->cloud[0], &T = c->cloud[1];
  TickArgs a;
  a.src = S.grid; a.tgt = T.grid; a.part_out = part_cur(c);
  a.thr2 = c->params.max_corr_dist * c->params.max_corr_dist;
  a.nn_idx = c->nn_idx; a.nn_ref = c->nn_ref; a.nrm_s = c->nrm_s_sorted; a.tgt_rec = c->tgt_rec; a.ppt = tick_ppt(c); a.rpb = tick_rpb(c); a.rows = tick_rows(c);
  a.far_mode = c->far_enabled ? c->far_mode : 0; a.tgt_raw = T.raw; a.cand = c->far_cand; a.cand_ref = c->far_cand_ref; a.cand_b = c->far_cand_b; a.far_req = c->far_req; a.far_stats = c->far_stats;
  a.tail = tail_args(c, (mode == 0 && a.far_mode != 1) ? 1 : 0, (int)tick_rows(c));
  a.aligned = c->aligned; a.fit_psum = c->fit_psum; a.fit_pcnt = c->fit_pcnt;
  a.clk = c->clk_probe ? c->clk_probe + 8 * (c->clk_n++ % 256) : nullptr; a.clk_blk = c->clk_probe ? c->clk_probe + 8 * 256 : nullptr;
  return a;
}
static FarArgs far_args(qn_ctx* c) {         // k_far behind the tick that just advanced the generation (st_cur = the state that tick published)
  CloudBuf &S = c->cloud[0], &T = c->cloud[1];
  FarArgs f;
  f.src = S.grid; f.tgt = T.grid; f.st = st_cur(c); f.thr2 = c->params.max_corr_dist * c->params.max_corr_dist; f.nn_idx = c->nn_idx; f.nn_ref = c->nn_ref; f.nrm_s = c->nrm_s_sorted; f.tgt_rec = c->tgt_rec; f.tgt_raw = T.raw;
  f.cand = c->far_cand; f.cand_ref = c->far_cand_ref; f.cand_b = c->far_cand_b; f.far_req = c->far_req; f.far_rows = c->far_rows; f.far_stats = c->far_stats; f.ranked_max = c->far_ranked ? (uint32_t)QN_FAR_WORDS : 0u;
  return f;
}
static FarReduceK::Args far_reduce_args(qn_ctx* c) {      // behind k_far: the side table -> one more row, then the controller step the tick left to it
  return FarReduceK::Args{c->far_rows, part_cur(c), (int)tick_rows(c), c->far_stats, tail_args(c, 1, (int)tick_rows(c) + 1)};
}
static void enqueue_tick_fused(qn_ctx* c) {
  TickArgs a = tick_args(c);
  { ProfScope ps(c, QN_K_GN_TICK_FUSED);
    const dim3 gr(tick_blocks(c)), bl(c->tick_tb);
#define QN_TICK_LAUNCH(TB, OCC, PR) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tick<TB, OCC, 0, PR>), gr, bl, 0, c->stream, a)
    if (a.clk) { if (c->tick_tb == 256) QN_TICK_LAUNCH(256, 4, true); else QN_TICK_LAUNCH(512, 4, true); }      // developer probe variant
    else if (c->tick_tb == 256) { if (c->tick_occ >= 4) QN_TICK_LAUNCH(256, 4, false); else if (c->tick_occ == 3) QN_TICK_LAUNCH(256, 3, false); else QN_TICK_LAUNCH(256, 2, false); }
    else { if (c->tick_occ >= 4) QN_TICK_LAUNCH(512, 4, false); else if (c->tick_occ == 3) QN_TICK_LAUNCH(512, 3, false); else QN_TICK_LAUNCH(512, 2, false); }
#undef QN_TICK_LAUNCH
  }
  if (a.far_mode == 1) {                       // the tick's refresh requests, chip-wide, one query per wave; its row follows the tick's, then the controller step
    const FarArgs f = far_args(c);
    { ProfScope ps(c, QN_K_FAR);
      hipLaunchKernelGGL(k_far, dim3(QN_FAR_BLOCKS), dim3(QN_FAR_THREADS), 0, c->stream, f);
      hipLaunchKernelGGL(k_far_reduce, dim3(1), dim3(QN_FAR_BLOCKS), 0, c->stream, far_reduce_args(c)); }
  }
  if (c->verify_track) enqueue_verify(c, true);         // (before the generation advances: st_cur is still the state - the pose - the tick's body used)
  c->gen++; c->part_rows = -1;
}
// Unseeded regime (the first outer iterations, while the pose still moves by more than a few cells): controller launch, grid search +
// list passes, accumulation.  `first`: the very first tick of an align has nothing to consume.
static void enqueue_tick(qn_ctx* c, bool seeded, int tick_no, const LookArgs* look = nullptr) {
  const int per_outer = c->params.optimizer == QN_OPT_LM ? 2 : 1, tick = tick_no / per_outer;      // tick = outer iteration (list-pass grid sizes)
  if (tick_no < c->unseeded_until) seeded = false;
  if (seeded && c->fused_ticks) { enqueue_tick_fused(c); return; }
  enqueue_nn(c, 0, c->sqd, seeded, tick);
  if (c->verify_track && seeded) enqueue_verify(c, false);
  enqueue_accumulate(c, 0, true, look);                              // + the controller step (and, behind the chunk's last unseeded tick of a lone forced run, the hand-over decision)
}
// The closing pass of align(): the last controller step + (once the state machine is done) fitness sweep + output cloud in ONE launch
// (k_tick<.., 1>) and the result block (k_finalize_fit).  `tracked`: the neighbours of the last tick seed the sweep; without them
// (no iteration ran, or the fused path is switched off) the unfused sequence runs: controller, search + list pass, two reductions, transform.
static void enqueue_epilogue(qn_ctx* c, double max_range, bool tracked) {
  if (tracked && c->fused_ticks && c->fused_final) {
    TickArgs a = tick_args(c, 1);
    { ProfScope ps(c, QN_K_FITNESS);
      const dim3 gr(tick_blocks(c)), bl(c->tick_tb);
      if (c->tick_tb == 256) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tick<256, 4, 1, false>), gr, bl, 0, c->stream, a);
      else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tick<512, 4, 1, false>), gr, bl, 0, c->stream, a); }
    hipLaunchKernelGGL(k_finalize_fit, dim3(1), dim3(64), 0, c->stream, st_cur(c), c->result_host, c->far_stats, c->fit_psum, c->fit_pcnt, (int)tick_rows(c));
    return;
  }
  GicpState* st = st_cur(c);                                    // (already stepped by the controller tail of the chunk's last tick)
  enqueue_nn(c, 1, c->sqd_fit, tracked, 1);
  { ProfScope ps(c, QN_K_FITNESS);
    hipLaunchKernelGGL(k_fitness_partial, dim3(QN_FIT_BLOCKS), dim3(QN_BLOCK), 0, c->stream, c->sqd_fit, c->cloud[0].n, max_range, st, c->fit_psum, c->fit_pcnt, 1);
    hipLaunchKernelGGL(k_fitness_final, dim3(1), dim3(QN_FIT_BLOCKS), 0, c->stream, c->fit_psum, c->fit_pcnt, st, 1); }
  { ProfScope ps(c, QN_K_TRANSFORM);
    hipLaunchKernelGGL(k_transform_cloud, dim3((c->cloud[0].n + QN_BLOCK - 1) / QN_BLOCK), dim3(QN_BLOCK), 0, c->stream, c->cloud[0].raw, c->cloud[0].n, st, c->aligned, 1); }
  hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, c->stream, st, c->result_host, c->far_stats);
}

// ---- the tracked regime as ONE persistent launch (qn_persist.cuh).  Only for the registration the reference runs - one pair at a time, nothing else on
// the GPU: every block must be resident at once (nblk + 1 <= 241 blocks of 512 threads on 256 CUs), so contexts that work in a batch, concurrent aligns of
// other contexts, the profiling / verification / probe modes and the far-query refresh regime (k_far between the ticks) keep the k_tick chain.
static std::atomic<int> g_aligns_in_flight{0};
static uint32_t persist_ppt(const qn_ctx* c) { const uint32_t cap = QN_PERSIST_TB * QN_PERSIST_MAX_BLOCKS; return (c->cloud[0].n + cap - 1) / cap; }
// (clouds beyond QN_PERSIST_TB x QN_PERSIST_MAX_BLOCKS points keep the chain: with more than one point per lane the persistent kernel's tracking records would live in memory, where
//  its top-2 form - nn_ref.w bounds the THIRD-nearest point - is not what k_tick reads; persist_fits: the launch's nblk + 1 blocks are all resident on this device, measured at context creation)
static bool persist_usable(const qn_ctx* c, bool alone) {
  return c->persist && c->persist_fits && persist_ppt(c) == 1 && !c->persist_batch_off && alone && c->fused_ticks && c->fused_final && (!c->prof_on || c->prof_persist) && !c->verify_track && !c->clk_probe && c->tick_tb == QN_PERSIST_TB &&
         (!c->far_enabled || c->far_mode == 2) && c->pg_rows != nullptr;
}
// A persistent launch that gave up (bounded spins: the reducer or a worker waited longer than `persist_timeout` - another tenant holds CUs, a partitioned device ...) leaves
// its row buffers in an unknown state and tracking records in its own top-2 form: re-arm the buffers; the caller then runs the whole align again on the k_tick chain
// (the reference would just be slow there, loop_closure.cpp:124 - never a lost loop closure).
static const int QN_INTERNAL_RETRY = -100;
static int persist_recover(qn_ctx* c) {
  hipStream_t s = c->stream;
  HIPCHK(c, hipMemsetAsync(c->pg_rows, 0xFF, sizeof(unsigned long long) * 3 * QN_PERSIST_ROWS * QN_PERSIST_RSTRIDE, s));
  HIPCHK(c, hipMemsetAsync(c->pg_status, 0, 4 * sizeof(uint32_t), s));
  HIPCHK(c, hipStreamSynchronize(s));
  char buf[200]; snprintf(buf, sizeof(buf), "align: the persistent kernel gave up (code %u after %u ticks: 1 rows, 2 tick budget, 3 closing sums, 4 pose) - the registration was re-run on the k_tick chain", c->pg_status_host[0], c->pg_status_host[1]);
  c->last_error = buf; c->persist_gave_up++;
  return QN_OK;
}
static int launch_persist(qn_ctx* c, uint32_t max_ticks, int cond = 0) {
  PersistArgs A;
  A.t = tick_args(c);                                                // (tail.st_in: the state the chain left, already stepped; tail.st_out: where the reducer leaves the final state)
  A.cond = cond; A.status_host = c->pg_status_host;
  A.t.ppt = persist_ppt(c); A.t.rpb = 1; A.t.rows = 0; A.t.clk = nullptr; A.t.clk_blk = nullptr;
  A.t.far_mode = c->far_enabled ? 2 : 0;
  const uint32_t per = QN_PERSIST_TB * A.t.ppt;
  A.nblk = (c->cloud[0].n + per - 1) / per;
  if (c->pg_epoch > 0xF0000000u) {                                   // epochs must never repeat within the life of the granule buffers
    HIPCHK(c, hipMemsetAsync(c->pg_bc, 0, sizeof(unsigned long long) * 64, c->stream));
    HIPCHK(c, hipMemsetAsync(c->pg_fit, 0, sizeof(unsigned long long) * (QN_PERSIST_MAX_BLOCKS + 1) * 4, c->stream));
    c->pg_epoch = 0;
  }
  A.rows_g = c->pg_rows; A.bc_g = c->pg_bc; A.fit_g = c->pg_fit; A.status = c->pg_status; A.result = c->result_host;
  A.epoch0 = c->pg_epoch; A.max_ticks = max_ticks; c->pg_epoch += max_ticks + 8;
  A.timeout = c->persist_timeout;                                    // per spin, 100 MHz wall clock; default 0.25 s: three orders of magnitude above any legitimate wait
  c->pg_status_host[0] = 0xffffffffu; c->pg_status_host[1] = 0;      // (the reducer overwrites it when it leaves)
  { ProfScope ps(c, QN_K_ALIGN_PERSIST);
    A.clk = c->pg_clk;
    if (c->pg_clk) { (void)hipMemsetAsync(c->pg_clk, 0, 8 * (64 * 16 + 16), c->stream); for (int g = 0; g < 64; g++) (void)hipMemsetAsync(c->pg_clk + 16 * g + 12, 0xff, 8, c->stream); }
    if (c->pg_clk) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_align_persist<QN_PERSIST_TB, true>), dim3(A.nblk + 1), dim3(QN_PERSIST_TB), 0, c->stream, A);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_align_persist<QN_PERSIST_TB, false>), dim3(A.nblk + 1), dim3(QN_PERSIST_TB), 0, c->stream, A); }
  c->gen++; c->part_rows = 0; c->persist_launches++;
  return QN_OK;
}

// Batch members: does the lane's unseeded phase go on for one more outer iteration?  The pose step the controller just took (step_dt / step_dr of the result block the chunk's
// k_finalize / closing pass wrote; the same quantity as look_decide's) would move the source points by more than 0.4 target cells: the NEXT step is then still large enough that a
// tracked tick - which re-finds every neighbour inside the ball of its previous one - degenerates into cooperative big-ball searches for most of the cloud inside the tick (an
// initial misalignment of a few degrees more than the bench's default turns one 25 us tick launch into 200-1800 us, and a launch ends with its slowest lane), while the dedicated
// unseeded search costs the same whatever the step was.  Searches are exact in both regimes; the rule only decides which kernels run, and it reads nothing but the controller's own
// step and the grids' numbers - the classic batch-member path (gicp_align) and the lanes (batch_register) take the same decision, so their records stay bit-identical.
static bool unseeded_goes_on(const qn_ctx* c, int tick_no, int per_outer, int ticks_left, bool lone = false) {      // lone: a registration on the adaptive lone path whose persistent launch declined on the second look (look_again)
  if ((!c->persist_batch_off && !lone) || !c->far_enabled || !c->fused_ticks) return false;
  if (tick_no != c->unseeded_until || tick_no + per_outer > c->unseeded_cap * per_outer || ticks_left < per_outer) return false;
  const qn::ResultBlock* rb = c->result_host;
  if (rb->phase == 2) return false;
  const GridDims& sg = *c->cloud[0].dims_host;                       // (valid: the stream has been synchronised since the grids were built)
  double reach2 = 0; const double lo[3] = {sg.ox, sg.oy, sg.oz}, ext[3] = {sg.nx * (double)sg.cell, sg.ny * (double)sg.cell, sg.nz * (double)sg.cell};
  for (int d = 0; d < 3; d++) { const double m = std::max(std::fabs(lo[d]), std::fabs(lo[d] + ext[d])); reach2 += m * m; }
  const double moved = rb->step_dt + rb->step_dr * std::sqrt(reach2), ok = 0.4 * (double)c->cloud[1].dims_host->cell;
  return moved > ok;                                                 // (NaN compares false: the fixed schedule)
}

static int ready(qn_ctx* c) {
  if (!c) return QN_ERR_INVALID_ARG;
  for (int w = 0; w < 2; w++) { if (c->cloud[w].n == 0) return QN_ERR_EMPTY_CLOUD; if (!c->cloud[w].has_grid || !c->cloud[w].has_cov) return QN_ERR_NOT_READY; }
  return QN_OK;
}

static int gicp_align(qn_ctx* c, const float guess[16], qn_gicp_result* out, bool alone);
extern "C" int qn_gicp_align(qn_ctx* c, const float guess[16], qn_gicp_result* out) {
  const bool alone = g_aligns_in_flight.fetch_add(1) == 0;           // (a heuristic for the persistent launch only: correctness never depends on it)
  int rc = gicp_align(c, guess, out, alone);
  if (rc == QN_INTERNAL_RETRY) rc = gicp_align(c, guess, out, false);      // (alone = false: no persistent launch this time)
  g_aligns_in_flight.fetch_sub(1);
  return rc;
}
static int gicp_align(qn_ctx* c, const float guess[16], qn_gicp_result* out, bool alone) {
  if (c && join_target(c) != QN_OK) return QN_ERR_HIP;
  int rc = ready(c); if (rc != QN_OK) return rc;
  if (!out) return QN_ERR_INVALID_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const qn_gicp_params& p = c->params;
  if (guess) HIPCHK(c, hipMemcpyAsync(c->guess_tmp, guess, sizeof(float) * 16, hipMemcpyHostToDevice, s));
  c->gen = 0; c->part_rows = 0;
  c->far_mode = 2;
  const int maxit = p.force_iterations > 0 ? p.force_iterations : p.max_iterations;
  // no far-candidate lists yet: the first (unseeded) search of the align resets the per-query references as it goes (enqueue_nn); an align without iterations has no such search
  c->clear_far_now = maxit > 0;
  if (maxit == 0) HIPCHK(c, hipMemsetAsync(c->far_cand_ref, 0, sizeof(float4) * c->cloud[0].n, s));
  hipLaunchKernelGGL(k_init_state, dim3(1), dim3(64), 0, s, st_cur(c), c->guess_tmp, guess ? 1 : 0, 0, c->far_stats);
  if (maxit == 0) hipLaunchKernelGGL(k_set_pose, dim3(1), dim3(64), 0, s, st_cur(c), c->pose_tmp, 2, 2);   // which=2: touch nothing, phase = done
  // Ticks are enqueued in chunks with NO host round trip inside a chunk; kernels of ticks past
  // convergence exit on the `phase` word.  LM needs two ticks per outer iteration (linearize, trial error).
  const int per_outer = p.optimizer == QN_OPT_LM ? 2 : 1;
  // Chunks: [the unseeded ticks] -> ONE host look at the state -> [everything else].  The look decides (i) whether the tracked ticks get the far-query refresh
  // kernel (k_far) behind them: how many source points have a FAR neighbour (no overlap there / occlusion) - a 4x faster registration of partially overlapping
  // clouds, and no extra launch in the chain when nobody needs it; (ii) for a registration that is alone on the GPU, WHEN tracking takes over.  A tracked tick
  // re-finds every neighbour inside the ball of its previous one: cheap when the pose step before it moved the points by less than half a cell, several times
  // the cost of the dedicated unseeded search when it moved them by metres (a 10-degree initial yaw error moves the far end of a 120 m scene by 10 m, and the
  // second step still by 0.5-2 m).  The steps shrink by 10x or more per iteration, so after two unseeded iterations and the controller step that follows them
  // (step_dt / step_dr in the result block) the host knows whether a third one should run unseeded as well.  A batch member keeps the fixed three (other
  // streams fill the chip; measured best for throughput).
  // (only runs that are known to be long pay for the look - a forced iteration count: with the real stopping rule the reference's operating point converges in
  //  3-7 iterations, and the extra round trip cost it 0.03-0.07 ms; knob single_from_tick = 0: the fixed hand-over everywhere; an explicitly earlier one wins too)
  const bool adaptive = (!c->persist_batch_off || (c->device_look && c->batch_look)) && c->fused_ticks && c->far_enabled && c->single_from_tick > 0 && p.force_iterations > 0 && p.optimizer == QN_OPT_GN && maxit >= 8 &&
                        std::min(c->track_from_tick, c->fused_from_tick) > c->single_from_tick;
  const int fixed_unseeded = std::max(1, std::min(c->track_from_tick, c->fused_from_tick)) * per_outer;
  const int unseeded = adaptive ? std::max(1, std::min(fixed_unseeded / per_outer, c->single_from_tick)) * per_outer : fixed_unseeded;
  // The number of ticks is known in advance only for forced Gauss-Newton runs (one tick per iteration).  A forced LM run needs one more tick
  // per rejected trial step, so it is driven like an unforced one: chunks until the device reports `done`, bounded by `budget`.
  const bool exact_ticks = p.force_iterations > 0 && p.optimizer == QN_OPT_GN;
  int ticks_left = exact_ticks ? maxit : 1 << 30;
  int chunk = c->far_enabled ? std::min(unseeded, ticks_left) : (exact_ticks ? ticks_left : c->ticks_per_chunk);
  bool first_chunk = true;
  long budget = (long)maxit * (p.optimizer == QN_OPT_LM ? (p.lm_max_iterations + 1) : 1) + 2;
  c->result_host->phase = 0;
  c->unseeded_until = adaptive ? chunk : fixed_unseeded;           // ticks below this index search unseeded (enqueue_tick)
  bool seeded = false; int tick_no = 0;   // the first linearisation runs the full grid search; every later NN pass tracks from it
  bool extending = false;                 // the unseeded phase goes on beyond the look, one outer iteration per host look (unseeded_goes_on)
  for (;;) {
    const bool look = adaptive && first_chunk && !extending && ticks_left > chunk;      // the adaptive look: the state behind the chunk's last tick (its controller tail has stepped it)
    // A registration that is alone on the GPU takes the look ON THE DEVICE (look_decide, at the end of that controller tail): the conditional third unseeded iteration and
    // the persistent launch are enqueued behind it and read its flags - no host round trip between the unseeded ticks and the tracked regime (it cost 15-30 us of a
    // 0.6 ms align).  If the flags say "not the persistent kernel" (many far neighbours: the k_far regime), that launch returns at once and the chain goes on from the host below.
    const bool dev_look = look && c->device_look && chunk > 0;
    const bool with_persist = dev_look && persist_usable(c, alone);   // (a batch member takes the same look - its third unseeded iteration is conditional too - and carries on with the chain)
    LookArgs la; memset(&la, 0, sizeof(la));
    if (dev_look) {
      la.out = c->result_host; la.far_stats = c->far_stats; la.sdims = c->cloud[0].dims; la.tdims = c->cloud[1].dims; la.enabled = 1;
      la.allow_extra = std::min(ticks_left - chunk - 1, per_outer) > 0 ? 1 : 0;
      c->result_host->look = 0;
    }
    for (int t = 0; t < chunk; t++) {
      c->count_far_now = first_chunk && (t / per_outer == (chunk - 1) / per_outer);      // far-query statistics: the chunk's last unseeded linearisation
      enqueue_tick(c, seeded, tick_no, (dev_look && t == chunk - 1) ? &la : nullptr); tick_no++; seeded = true;
    }
    c->count_far_now = false;
    bool declined = false;
    if (dev_look) {
      c->unseeded_until = tick_no + 1;
      enqueue_nn(c, 0, c->sqd, false, tick_no / per_outer, QN_LOOK_EXTRA);
      LookArgs la2 = la; la2.enabled = 2; la2.allow_extra = 0;      // the second look, at the tail of the extra iteration: the persistent launch goes ahead only if THAT step is small too (look_again)
      enqueue_accumulate(c, QN_LOOK_EXTRA, true, c->unseeded_cap > fixed_unseeded / per_outer ? &la2 : nullptr);      // (flag not set: the launch hands the state on unchanged)
      if (with_persist && (rc = launch_persist(c, (uint32_t)(budget - chunk) + 2u, QN_LOOK_GO)) != QN_OK) return rc;
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipStreamSynchronize(s));
      if ((rc = clouds_valid(c)) != QN_OK) return rc;
      const int extra = (c->result_host->look & QN_LOOK_EXTRA) ? per_outer : 0;
      c->last_extra_unseeded = extra;
      if (with_persist && c->result_host->phase == 2) break;        // the whole registration ran behind the look
      if (with_persist && c->pg_status_host[0] != 5u) { if ((rc = persist_recover(c)) != QN_OK) return rc; return QN_INTERNAL_RETRY; }      // gave up: the whole align again, on the chain
      // declined: the state is the one the look (and the extra iteration, if it ran) left - carry on with the chain
      declined = true;
      if (with_persist) { c->gen--; c->persist_launches--; }        // (the launch wrote no state)
      tick_no += extra; budget -= extra; ticks_left -= extra;
      c->unseeded_until = tick_no;
    } else {
    if (look || (exact_ticks && ticks_left > chunk)) {            // forced GN iterations cannot be done yet: only the statistics block is needed at this chunk end
      hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, s, st_cur(c), c->result_host, c->far_stats);
    } else {
      enqueue_epilogue(c, DBL_MAX, maxit > 0 && tick_no > 0);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(s));
    if (first_chunk && (rc = clouds_valid(c)) != QN_OK) return rc;
    }
    budget -= std::max(chunk, 1); ticks_left -= chunk;
    if (c->result_host->phase == 2) {
      if (!look) break;
      enqueue_epilogue(c, DBL_MAX, true);                          // the look's controller step finished the state machine: the closing pass is still to run
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipStreamSynchronize(s));
      if (c->result_host->phase == 2) break;
      c->last_error = "align: closing pass did not complete"; return QN_ERR_HIP;
    }
    // far queries: keep the refresh kernel in the chain only while ticks actually ask for refreshes (a launch costs ~4 us per tick)
    if (c->far_enabled) {
      const qn::ResultBlock* rb = c->result_host;
      if (first_chunk) c->far_mode = rb->far_queries * 16u > c->cloud[0].n ? 1 : 2;      // more than ~6 % of the source has a far neighbour
      else c->far_mode = (c->far_mode == 1 ? rb->far_requests : rb->far_misses / (uint32_t)std::max(1, chunk)) > 256u ? 1 : 2;   // stray misses are cheaper inside k_tick
    }
    if (!exact_ticks) chunk = first_chunk ? c->ticks_per_chunk : std::max(2 * per_outer, c->ticks_per_chunk / 2);   // convergence is usually near after the first chunks: ticks past it are wasted launches
    else chunk = c->far_mode == 1 ? std::min(ticks_left, c->far_chunk) : ticks_left;      // the refresh regime is needed for the first few tracked ticks only: an empty k_far + k_far_reduce behind every later tick cost 14 us each (80 %-overlap pairs: 1384 -> 1429 registrations/s with 4 instead of 8)
    if (budget <= 0) { c->last_error = "align: device state machine did not terminate"; return QN_ERR_HIP; }
    if (first_chunk && (!adaptive || declined || extending) && unseeded_goes_on(c, tick_no, per_outer, ticks_left, adaptive)) {
      // a batch member - or a lone registration whose persistent launch declined on its second look - whose pose still moves by more than a fraction of a cell: one more
      // unseeded outer iteration, then look again (first_chunk stays set: the far-query statistics are those of the LAST unseeded linearisation)
      c->unseeded_until = tick_no + per_outer; chunk = per_outer; extending = true;
      continue;
    }
    if (look && !declined) {
      // how far the NEXT step will move the source points at most: the step just taken (translation + rotation x the cloud's reach from the origin), shrunk
      // by 10 (what the optimiser does per iteration at this stage, measured on the synthetic pairs: 12x - 50x)
      const GridDims& sg = *c->cloud[0].dims_host;                   // (valid: this stream has been synchronised since the grids were built)
      double reach2 = 0; const double lo[3] = {sg.ox, sg.oy, sg.oz}, ext[3] = {sg.nx * (double)sg.cell, sg.ny * (double)sg.cell, sg.nz * (double)sg.cell};
      for (int d = 0; d < 3; d++) { const double m = std::max(std::fabs(lo[d]), std::fabs(lo[d] + ext[d])); reach2 += m * m; }
      const double moved = c->result_host->step_dt + c->result_host->step_dr * std::sqrt(reach2), ok = 0.4 * (double)c->cloud[1].dims_host->cell;
      // (one more unseeded iteration at most: a fourth one measured slower than the tracked tick it replaces - at a nearly converged pose the list pass ends with a
      // few one-per-wave far queries of 100-300 us each, tools/gpu_probe_lists4.py - so a large second step hands over at iteration 3 like the fixed schedule)
      int extra = moved <= ok ? 0 : 1;
      extra = std::min(extra * per_outer, std::max(0, std::min(ticks_left - 1, per_outer)));
      if (!(moved == moved)) extra = 0;
      c->last_extra_unseeded = extra;
      if (extra > 0 && c->unseeded_cap > fixed_unseeded / per_outer) {      // the extra iteration as a chunk of its own with a look behind it: the same decisions as the device route (look_again), one iteration per look
        c->unseeded_until = tick_no + extra; chunk = extra; extending = true;
        continue;
      }
      c->unseeded_until = tick_no + extra;
      for (int t = 0; t < extra; t++) { enqueue_tick(c, seeded, tick_no); tick_no++; }
      budget -= extra; ticks_left -= extra;
      if (exact_ticks) chunk = c->far_mode == 1 ? std::min(ticks_left, c->far_chunk) : ticks_left;
    }
    if (tick_no > 0 && persist_usable(c, alone)) {          // everything that is left - ticks, closing pass, result - in ONE persistent launch (first chunk end, or once the far-query refreshes have died down)
      if ((rc = launch_persist(c, (uint32_t)budget + 2u)) != QN_OK) return rc;
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipStreamSynchronize(s));
      if (c->result_host->phase == 2) break;
      if ((rc = persist_recover(c)) != QN_OK) return rc;            // gave up: the whole align again, on the chain
      return QN_INTERNAL_RETRY;
    }
    first_chunk = false;
  }
  c->prof_collect();
  *out = c->result_host->r;
  c->trace_len = c->result_host->trace_len;
  c->aligned_valid = true;
  return QN_OK;
}

extern "C" int qn_gicp_fitness(qn_ctx* c, double max_range, double* score) {
  if (c && join_target(c) != QN_OK) return QN_ERR_HIP;
  int rc = ready(c); if (rc != QN_OK) return rc;
  if (!score) return QN_ERR_INVALID_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  enqueue_nn(c, 1, c->sqd_fit, false);
  hipLaunchKernelGGL(k_fitness_partial, dim3(QN_FIT_BLOCKS), dim3(QN_BLOCK), 0, c->stream, c->sqd_fit, c->cloud[0].n, max_range, st_cur(c), c->fit_psum, c->fit_pcnt, 1);
  hipLaunchKernelGGL(k_fitness_final, dim3(1), dim3(QN_FIT_BLOCKS), 0, c->stream, c->fit_psum, c->fit_pcnt, st_cur(c), 1);
  hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, c->stream, st_cur(c), c->result_host, (uint32_t*)nullptr);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->prof_collect();
  if ((rc = clouds_valid(c)) != QN_OK) return rc;
  if (c->result_host->phase != 2) return QN_ERR_NOT_READY;
  *score = c->result_host->r.fitness;
  return QN_OK;
}

extern "C" int qn_gicp_transformed_source(qn_ctx* c, float* xyz_out, uint32_t stride) {
  if (!c || !xyz_out || stride < 12 || (stride & 3)) return QN_ERR_INVALID_ARG;
  if (!c->aligned_valid) return QN_ERR_NOT_READY;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t w = stride >= 16 ? 16 : 12;
  // through the context's page-locked zone (one contiguous D2H + a CPU scatter): a strided 2-D copy into pageable memory is 100k row transfers and pins the caller's pages
  qn_ctx::PinZone& z = c->pin_up;
  if (!z.h) { HIPCHK(c, hipHostMalloc((void**)&z.h, (size_t)c->max_points * 16, hipHostMallocDefault)); HIPCHK(c, hipEventCreateWithFlags(&z.ev, hipEventDisableTiming)); }
  if (z.busy) { HIPCHK(c, hipEventSynchronize(z.ev)); z.busy = false; }
  const uint32_t n = c->cloud[0].n;
  HIPCHK(c, hipMemcpyAsync(z.h, c->aligned, (size_t)n * 16, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const char* in = z.h; char* o = (char*)xyz_out;
  for (uint32_t i = 0; i < n; i++, in += 16, o += stride) memcpy(o, in, w);
  return QN_OK;
}

extern "C" int qn_gicp_get_trace(qn_ctx* c, qn_iter_trace* out, uint32_t cap, uint32_t* n) {
  if (!c || !out || !n) return QN_ERR_INVALID_ARG;
  uint32_t m = std::min(cap, c->trace_len);
  HIPCHK(c, hipSetDevice(c->device));
  if (m) HIPCHK(c, hipMemcpy(out, c->trace, sizeof(qn_iter_trace) * m, hipMemcpyDeviceToHost));
  *n = m;
  return QN_OK;
}

// the iteration trace of the registration lane `lane` carried in the latest run of qn_gicp_align_batch (lane l of a run = the l-th pair of that run;
// with n_pairs <= batch_lanes: pair l).  Parity tests compare a batched lane's y0 / lambda trajectory with the oracle's, iteration by iteration.
extern "C" int qn_gicp_get_lane_trace(qn_ctx* c, uint32_t lane, qn_iter_trace* out, uint32_t cap, uint32_t* n) {
  if (!c || !out || !n) return QN_ERR_INVALID_ARG;
  if (c->lanes.empty()) return lane == 0 ? qn_gicp_get_trace(c, out, cap, n) : QN_ERR_NOT_READY;
  if (lane >= c->lanes.size()) return QN_ERR_INVALID_ARG;
  qn_ctx* l = c->lanes[lane];
  uint32_t m = std::min(cap, l->trace_len);
  HIPCHK(c, hipSetDevice(c->device));
  if (m) HIPCHK(c, hipMemcpy(out, l->trace, sizeof(qn_iter_trace) * m, hipMemcpyDeviceToHost));
  *n = m;
  return QN_OK;
}

// LoopClosure::icpAlignment (loop_closure.cpp:110-136).  where: 0 = both clouds on the host, 1 = both on the device,
// 2 = source on the device as packed float4 (the coarse-aligned cloud of coarseToFineAlignment), target on the host,
// 3 = like 2 with the target on the device too.
static int icp_alignment(qn_ctx* c, const float* src, uint32_t ns, const float* dst, uint32_t nt, uint32_t stride, double thr,
                         qn_gicp_result* out, int* valid, int where, bool reuse_source = false, bool target_ready = false) {
  if (!c || !out || !valid) return QN_ERR_INVALID_ARG;
  *valid = 0;
  memset(out, 0, sizeof(*out)); out->fitness = DBL_MAX;
  for (int i = 0; i < 4; i++) { out->T[5 * i] = 1.f; out->T64[5 * i] = 1.0; }
  int rc;
  // reuse_source: the context still holds this source cloud's grid and covariances (the candidates of ONE loop-closure query share their
  // source; loop_closure.cpp:116-123 rebuilds it for every call because the reference only ever tries one candidate)
  const bool need_source = !(reuse_source && c->cloud[0].has_grid && c->cloud[0].has_cov && c->cloud[0].n == ns);
  // target_ready: coarseToFineAlignment prepared the target (grid + covariances, :122-123) while Quatro's matching ran - it depends on nothing Quatro computes (coarse_to_fine)
  const bool need_target = !(target_ready && c->cloud[1].has_grid && c->cloud[1].has_cov && c->cloud[1].n == nt);
  // tgt_early: the target's upload + grid build (second stream, own scratch) is ENQUEUED right behind the source's, before the source's k-NN launches: the host needs
  // ~50 us to submit those, and the second stream used to sit empty that long (lone registration: the target chain started 77 us into the call).  Same kernels, same data.
  const bool early = c->tgt_early && need_source && need_target;
  if (need_source) {
    if ((rc = set_cloud(c, QN_SOURCE, src, ns, where >= 2 ? 16 : stride, where != 0)) != QN_OK) return rc;   // :120
    if (early && (rc = set_cloud(c, QN_TARGET, dst, nt, stride, where == 1 || where == 3)) != QN_OK) return rc;     // :122, early
    if ((rc = qn_gicp_compute_covariances(c, QN_SOURCE)) != QN_OK) return rc;         // :121
  }
  if (need_target) {
    if (!early && (rc = set_cloud(c, QN_TARGET, dst, nt, stride, where == 1 || where == 3)) != QN_OK) return rc;     // :122 (on the second stream: see TargetScope)
    if ((rc = qn_gicp_compute_covariances(c, QN_TARGET)) != QN_OK) return rc;         // :123
  }
  if ((rc = qn_gicp_align(c, nullptr, out)) != QN_OK) return rc;                    // :124, :127
  *valid = (out->converged && out->fitness < thr) ? 1 : 0;                          // :129
  return QN_OK;
}
// the same registration against the source cloud the context already holds (set + covariances done by the previous call with THIS source)
extern "C" int qn_icp_alignment_same_source(qn_ctx* c, const float* dst, uint32_t nt, uint32_t stride, int dst_on_device, double thr, qn_gicp_result* out, int* valid) {
  if (!c) return QN_ERR_INVALID_ARG;
  if (!c->cloud[0].has_grid || !c->cloud[0].has_cov) return QN_ERR_NOT_READY;
  return icp_alignment(c, nullptr, c->cloud[0].n, dst, nt, stride, thr, out, valid, dst_on_device ? 1 : 0, true);
}
extern "C" int qn_icp_alignment(qn_ctx* c, const float* src, uint32_t ns, const float* dst, uint32_t nt, uint32_t stride, double thr, qn_gicp_result* out, int* valid) {
  return icp_alignment(c, src, ns, dst, nt, stride, thr, out, valid, 0);
}
extern "C" int qn_icp_alignment_device(qn_ctx* c, const float* src, uint32_t ns, const float* dst, uint32_t nt, uint32_t stride, double thr, qn_gicp_result* out, int* valid) {
  return icp_alignment(c, src, ns, dst, nt, stride, thr, out, valid, 1);
}

// batch over several contexts (streams): one host worker thread per context, dynamic pair assignment.  A context whose batch_lanes is >= 2 (the default) takes
// `batch_lanes` pairs at a time and registers them in lockstep, the pair as a grid dimension of every launch (qn_batch.inc); otherwise one pair at a time.
// The last round of a batch call is dealt in equal shares so that the contexts finish together - but not in shares smaller than this: a run of two or three lanes is
// latency, not throughput, and three contexts grinding through their k-NN phases at once slow each other down.  Measured on one MI355X, 8 pairs of 100k points (one rank's
// whole work at 8 GPUs, tools/gpu_share8_probe.py): 3 contexts x (3 + 3 + 2) 3.52 ms, 4 x 2 3.08, 1 x 8 2.78, 2 x 4 2.46 ms.
// (qn_ctx::batch_min_share of the FIRST context of the call, default 4; knob batch_min_share)
namespace { bool batch_supported(const qn_ctx* c); int batch_ensure_lanes(qn_ctx* c);
            int batch_register(qn_ctx* owner, const qn_pair_desc* pairs, const uint32_t* idx, uint32_t m, double thr, qn_gicp_result* results, int* valid, int* status,
                               std::vector<const float*>& last_src, std::vector<uint64_t>& last_key); }
extern "C" int qn_icp_alignment_batch(qn_ctx* const* ctxs, uint32_t n_ctx, const qn_pair_desc* pairs, uint32_t n_pairs, double thr,
                                      qn_gicp_result* results, int* valid, int* status) {
  if (!ctxs || n_ctx == 0 || (n_pairs && (!pairs || !results || !valid || !status))) return QN_ERR_INVALID_ARG;
  for (uint32_t i = 0; i < n_ctx; i++) if (!ctxs[i]) return QN_ERR_INVALID_ARG;
  std::atomic<uint32_t> next{0}, final_share{0};
  // several registrations in flight already fill the chip: the two-stream pair pipeline of a single registration (icp_alignment) would only
  // add streams to the hardware queues (measured: 2290 -> 1880 registrations/s with 4 contexts), so it is switched off for the batch
  std::vector<char> saved(n_ctx);
  std::vector<char> saved_p(n_ctx);
  std::vector<char> use_lanes(n_ctx);
  for (uint32_t i = 0; i < n_ctx; i++) {
    saved[i] = ctxs[i]->pair_pipeline; saved_p[i] = ctxs[i]->persist_batch_off;
    use_lanes[i] = n_pairs > 1 && batch_supported(ctxs[i]) && batch_ensure_lanes(ctxs[i]) == QN_OK;
    if (n_ctx > 1 || use_lanes[i]) { ctxs[i]->pair_pipeline = false; ctxs[i]->persist_batch_off = true; }
  }
  auto worker = [&](qn_ctx* c, bool lanes) {
    if (lanes) {
      (void)hipSetDevice(c->device);
      (void)join_target(c);                                               // a target an earlier qn_gicp_set_target left in flight on the second stream still owns scratch set 2
      const uint32_t B = (uint32_t)c->lanes.size();
      // (pairs are dealt in runs of B; towards the end of the batch - and when the batch is smaller than the lanes of all contexts together - the runs shrink to an n-th of
      //  what is left, so that the contexts finish together instead of one of them registering the last full run alone)
      std::vector<const float*> last_src(B, nullptr); std::vector<uint64_t> last_key(B, 0);
      std::vector<uint32_t> idx(B);
      for (;;) {
        uint32_t base = next.load(), m = 0;
        for (;;) {
          if (base >= n_pairs) break;
          const uint32_t rem = n_pairs - base;
          if (rem > n_ctx * B) m = B;
          else {                                                        // the last round: what is left, in n_ctx equal shares (fixed by the first context that gets here)
            uint32_t q = final_share.load();
            if (q == 0) { uint32_t want = std::min<uint32_t>(B, std::max<uint32_t>((uint32_t)std::max(1, ctxs[0]->batch_min_share), (rem + n_ctx - 1) / n_ctx)); if (final_share.compare_exchange_strong(q, want)) q = want; }
            m = std::min(q, rem);
          }
          m = std::min(m, B);                                           // (the share was fixed by whichever context got there first, with ITS lane count: never more than this context's - ADVICE r5)
          if (next.compare_exchange_weak(base, base + m)) break;
        }
        if (base >= n_pairs) break;
        for (uint32_t l = 0; l < m; l++) idx[l] = base + l;
        const int rc = batch_register(c, pairs, idx.data(), m, thr, results, valid, status, last_src, last_key);
        if (rc != QN_OK) { (void)hipStreamSynchronize(c->stream); for (uint32_t l = 0; l < m; l++) status[base + l] = rc; std::fill(last_src.begin(), last_src.end(), nullptr); }
      }
      return;
    }
    const qn_pair_desc* last = nullptr;                                 // the source this context holds (within THIS call: same pointer = same cloud): the candidates of one query share it
    for (;;) {
      const uint32_t i = next.fetch_add(1);
      if (i >= n_pairs) break;
      const qn_pair_desc& p = pairs[i];
      const bool same = c->batch_share_source && last && last->src == p.src && last->ns == p.ns && last->stride_bytes == p.stride_bytes && last->on_device == p.on_device;
      status[i] = icp_alignment(c, p.src, p.ns, p.dst, p.nt, p.stride_bytes, thr, &results[i], &valid[i], p.on_device ? 1 : 0, same);
      last = status[i] == QN_OK ? &p : nullptr;
    }
  };
  qn::WorkerPool::instance().run(n_ctx, [&](uint32_t i) { worker(ctxs[i], use_lanes[i] != 0); });      // parked threads, not one std::thread per context per call (qn_pool.h)
  for (uint32_t i = 0; i < n_ctx; i++) { ctxs[i]->pair_pipeline = saved[i] != 0; ctxs[i]->persist_batch_off = saved_p[i] != 0; }
  return QN_OK;
}

// ------------------------------------------------------------------ per-stage read-backs for parity tests
extern "C" int qn_gicp_get_covariances(qn_ctx* c, int which, double* out9) {
  if (c && join_target(c) != QN_OK) return QN_ERR_HIP;
  if (!c || !out9 || (which != 0 && which != 1)) return QN_ERR_INVALID_ARG;
  CloudBuf& b = c->cloud[which];
  if (!b.has_cov) return QN_ERR_NOT_READY;
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<double> h((size_t)b.n * 6);
  double* d6 = nullptr;                                                 // the engine keeps normals (C = I - 0.999 n n^T): rebuild the matrices for the read-back
  HIPCHK(c, hipMalloc(&d6, sizeof(double) * 6 * b.n));
  hipLaunchKernelGGL(k_cov_from_normals, dim3((b.n + 255) / 256), dim3(256), 0, c->stream, b.nrm, b.n, d6);
  hipError_t ce = hipMemcpyAsync(h.data(), d6, sizeof(double) * 6 * b.n, hipMemcpyDeviceToHost, c->stream);
  if (ce == hipSuccess) ce = hipStreamSynchronize(c->stream);
  (void)hipFree(d6);
  if (ce != hipSuccess) { c->set_error("covariance read-back", ce, __LINE__); return QN_ERR_HIP; }
  { const int vrc = clouds_valid(c); if (vrc != QN_OK) return vrc; }
  for (uint32_t i = 0; i < b.n; i++) {
    const double* s = &h[(size_t)i * 6]; double* o = out9 + (size_t)i * 9;
    o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[1]; o[4] = s[3]; o[5] = s[4]; o[6] = s[2]; o[7] = s[4]; o[8] = s[5];
  }
  return QN_OK;
}

extern "C" int qn_gicp_knn(qn_ctx* c, int which, int k, int32_t* idx_out, float* d2_out) {
  if (c && join_target(c) != QN_OK) return QN_ERR_HIP;
  if (!c || !idx_out || !d2_out || (which != 0 && which != 1) || k < 1 || k > 32) return QN_ERR_INVALID_ARG;
  CloudBuf& b = c->cloud[which];
  if (!b.has_grid) return QN_ERR_NOT_READY;
  HIPCHK(c, hipSetDevice(c->device));
  hipFree(c->dbg_knn_idx); hipFree(c->dbg_knn_d2); c->dbg_knn_idx = nullptr; c->dbg_knn_d2 = nullptr;
  HIPCHK(c, hipMalloc(&c->dbg_knn_idx, sizeof(int32_t) * (size_t)b.n * k));
  HIPCHK(c, hipMalloc(&c->dbg_knn_d2, sizeof(float) * (size_t)b.n * k));
  // (a cloud with non-finite coordinates is an EMPTY grid on the device: the selection kernels write nothing, and the covariance kernel behind them must not gather through
  //  whatever a fresh allocation holds - -1 reads as `no neighbour`; the call itself returns the cloud's error after its synchronisation)
  HIPCHK(c, hipMemsetAsync(c->dbg_knn_idx, 0xff, sizeof(int32_t) * (size_t)b.n * k, c->stream));
  const int ksave = c->params.k_correspondences; const bool had = b.has_cov;
  c->params.k_correspondences = k;
  int rc = compute_cov(c, which, c->dbg_knn_idx, c->dbg_knn_d2);
  c->params.k_correspondences = ksave;
  b.has_cov = had && (k == ksave);
  if (rc != QN_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(idx_out, c->dbg_knn_idx, sizeof(int32_t) * (size_t)b.n * k, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(d2_out, c->dbg_knn_d2, sizeof(float) * (size_t)b.n * k, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->prof_collect();
  return clouds_valid(c);
}

extern "C" int qn_gicp_linearize(qn_ctx* c, const double T[16], double H[36], double b6[6], double* err, int32_t* corr_out, float* sqd_out) {
  if (c && join_target(c) != QN_OK) return QN_ERR_HIP;
  int rc = ready(c); if (rc != QN_OK) return rc;
  if (!T || !H || !b6 || !err) return QN_ERR_INVALID_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  HIPCHK(c, hipMemcpyAsync(c->pose_tmp, T, sizeof(double) * 16, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_set_pose, dim3(1), dim3(64), 0, s, st_cur(c), c->pose_tmp, 0, 0);
  enqueue_nn(c, 0, c->sqd, false); enqueue_accumulate(c, 0, false); enqueue_solve(c, 1, 0);
  GicpState* hs = (GicpState*)malloc(sizeof(GicpState));
  hipError_t e = hipMemcpyAsync(hs, st_cur(c), sizeof(GicpState), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess && corr_out) e = hipMemcpyAsync(corr_out, c->corr, sizeof(int32_t) * c->cloud[0].n, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess && sqd_out) e = hipMemcpyAsync(sqd_out, c->sqd, sizeof(float) * c->cloud[0].n, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) { free(hs); c->set_error("linearize readback", e, __LINE__); return QN_ERR_HIP; }
  memcpy(H, hs->H, sizeof(double) * 36); memcpy(b6, hs->b, sizeof(double) * 6); *err = hs->y0;
  free(hs);
  c->prof_collect();
  return clouds_valid(c);
}

extern "C" int qn_gicp_compute_error(qn_ctx* c, const double T[16], double* err) {
  if (c && join_target(c) != QN_OK) return QN_ERR_HIP;
  int rc = ready(c); if (rc != QN_OK) return rc;
  if (!T || !err) return QN_ERR_INVALID_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  HIPCHK(c, hipMemcpyAsync(c->pose_tmp, T, sizeof(double) * 16, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_set_pose, dim3(1), dim3(64), 0, s, st_cur(c), c->pose_tmp, 1, 1);
  enqueue_accumulate(c, 0, false); enqueue_solve(c, 2, 0);
  HIPCHK(c, hipMemcpyAsync(c->scalar_host, &st_cur(c)->yi, sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  *err = c->scalar_host[0];
  c->prof_collect();
  return clouds_valid(c);
}

// ------------------------------------------------------------------ profiling hooks
extern "C" int qn_prof_enable(qn_ctx* c, int on) { if (!c) return QN_ERR_INVALID_ARG; c->prof_on = on != 0; return QN_OK; }
extern "C" int qn_prof_reset(qn_ctx* c) { if (!c) return QN_ERR_INVALID_ARG; c->prof_collect(); for (auto& s : c->stats) { s.total_ms = 0; s.launches = 0; } return QN_OK; }
extern "C" int qn_prof_get(qn_ctx* c, int fam, qn_kernel_stat* out) {
  if (!c || !out || fam < 0 || fam >= QN_K_COUNT) return QN_ERR_INVALID_ARG;
  c->prof_collect(); *out = c->stats[fam]; return QN_OK;
}
// test/tuning knobs (not part of the reference surface)
extern "C" int qn_debug_set(qn_ctx* c, const char* key, double v) {
  if (!c || !key) return QN_ERR_INVALID_ARG;
  std::string k(key);
  if (k == "cell") { c->cell_override = v; c->cloud[0].has_grid = c->cloud[1].has_grid = false; }
  else if (k == "margin_nn") c->margin_nn = (float)v;
  else if (k == "margin_nn_t0") c->margin_nn_t0 = (float)v;
  else if (k == "big_blocks0") c->big_blocks0 = v < 64 ? 64 : (int)v;
  else if (k == "fb_blocks0") c->fb_blocks0 = v < 64 ? 64 : (int)v;
  else if (k == "margin_knn") c->margin_knn = (float)v;
  else if (k == "knn_single_all") c->knn_single_all = v != 0;
  else if (k == "bbox_blocks") c->bbox_blocks = std::max(1, (int)v);
  else if (k == "device_look") c->device_look = v != 0;
  else if (k == "far_group") c->far_group = (int)v;
  else if (k == "c2f_overlap") c->c2f_overlap = v != 0;
  else if (k == "quatro_fused") c->quatro_fused = v != 0;
  else if (k == "tgt_early") c->tgt_early = v != 0;
  else if (k == "c2f_lanes_fpfh") c->c2f_lanes_fpfh = v != 0;
  else if (k == "normals_fg") c->normals_fg = (int)v;
  else if (k == "fpfh_fg") c->fpfh_fg = (int)v;
  else if (k == "list_small") c->list_small = std::max(0, (int)v);
  else if (k == "far_chunk") c->far_chunk = std::max(1, (int)v);
  else if (k == "far_ranked") c->far_ranked = v != 0;
  else if (k == "batch_look") c->batch_look = v != 0;
  else if (k == "unseeded_cap") c->unseeded_cap = (int)v;
  else if (k == "pair_pipeline") c->pair_pipeline = v != 0;
  else if (k == "persist") c->persist = v != 0;
  else if (k == "persist_timeout") c->persist_timeout = v < 1 ? 1ull : (unsigned long long)v;      // 100 MHz ticks a spin of the persistent kernel may last (tests force a give-up with a tiny value)
  else if (k == "prof_persist") c->prof_persist = v != 0;      // profiling (qn_prof_enable) normally times the k_tick chain; 1: let the persistent kernel run and time it as its own family
  else if (k == "persist_probe") {                              // developer probe: wall-clock stamps inside k_align_persist (qn_debug_get_persist_clk)
    if (v != 0 && !c->pg_clk) { if (hipMalloc(&c->pg_clk, 8 * (64 * 16 + 16)) != hipSuccess) return QN_ERR_HIP; }
    if (c->pg_clk) (void)hipMemset(c->pg_clk, 0, 8 * (64 * 16 + 16));
    if (v == 0) { (void)hipFree(c->pg_clk); c->pg_clk = nullptr; }
  }
  else if (k == "batch_lanes") {                                 // candidate pairs per kernel launch of the batch entry points (qn_batch.inc); < 2: one pair at a time on this context's stream
    const int b = std::max(1, std::min((int)v, 64));
    if (b < (int)c->lanes.size()) { if (hipStreamSynchronize(c->stream) != hipSuccess) return QN_ERR_HIP; while ((int)c->lanes.size() > std::max(b, 1)) { qn_ctx* l = c->lanes.back(); c->lanes.pop_back(); if (l != c) qn_ctx_destroy(l); } }
    c->batch_lanes = b; c->lanes_failed = false;
  }
  else if (k == "fail_lane_create") c->fail_lane_create = (int)v;   // test knob: creating lane index >= v fails (the out-of-memory fallback of the batch entry points)
  else if (k == "batch_share_source") c->batch_share_source = v != 0;      // batch entry points: pairs that name the same source buffer share its preparation (default on: the candidates of one query)
  else if (k == "batch_trace") c->batch_trace = v != 0;           // developer: host timeline of every batched segment on stderr
  else if (k == "batch_member") c->persist_batch_off = v != 0;      // this context registers beside others (qn_multi with in_flight > 1): no persistent launches
  else if (k == "stable_cells") c->stable_cells = v != 0;
  else if (k == "knn_hist") c->knn_hist = v != 0;
  else if (k == "nn_rounds") c->nn_rounds = v < 1 ? 1 : (int)v;
  else if (k == "nn_lane") c->nn_lane = v < 0 ? -1 : (v != 0 ? 1 : 0);                  // first (unseeded) 1-NN pass one query per lane (NnLaneK) instead of the cooperative 16-per-wave search
  else if (k == "track_from_tick") c->track_from_tick = v < 1 ? 1 : (int)v;
  else if (k == "single_from_tick") c->single_from_tick = v < 1 ? 1 : (int)v;
  else if (k == "fused_from_tick") c->fused_from_tick = v < 1 ? 1 : (int)v;
  else if (k == "knn_rounds") c->knn_rounds = v < 1 ? 1 : (int)v;
  else if (k == "knn_trips") c->knn_trips = v < 1 ? 1 : (int)v;
  else if (k == "knn_mm") c->knn_mm = v != 0;
  else if (k == "fused_ticks") c->fused_ticks = v != 0;
  else if (k == "big_ratio") c->big_ratio = (float)v;
  else if (k == "margin_nn_cap") c->margin_nn_cap = (int)v;
  else if (k == "margin_knn_cap") c->margin_knn_cap = (int)v;
  else if (k == "dbg_counters") {
    if (v != 0 && !c->dbg_counters) { if (hipMalloc(&c->dbg_counters, 16 * sizeof(uint32_t)) != hipSuccess) return QN_ERR_HIP; }
    if (c->dbg_counters) (void)hipMemset(c->dbg_counters, 0, 16 * sizeof(uint32_t));
    if (v == 0) { (void)hipFree(c->dbg_counters); c->dbg_counters = nullptr; }
    for (int w = 0; w < 2; w++) c->cloud[w].grid.dbg = c->dbg_counters;
  }
  else if (k == "ticks_per_chunk") c->ticks_per_chunk = std::max(1, (int)v);
  else if (k == "tick_occ") c->tick_occ = (int)v;
  else if (k == "far") c->far_enabled = v != 0;
  else if (k == "fused_final") c->fused_final = v != 0;
  else if (k == "feat_mfma") c->feat_mfma = v != 0;
  else if (k == "feat_sample") c->feat_sample = (int)v;
  else if (k == "feat_query_dedupe") c->feat_query_dedupe = v != 0;
  else if (k == "feat_min_blocks") c->feat_min_blocks = (int)v;
  else if (k == "feat_verify") { c->feat_verify = v != 0; if (c->q_mm_vcnt) (void)hipMemset(c->q_mm_vcnt, 0, 16); }
  else if (k == "list_probe") {                                  // developer probe: wall-clock time of the one-per-wave entries of the LAST unseeded list pass (qn_debug_get_list_probe)
    if (v != 0 && !c->list_probe) { if (hipMalloc(&c->list_probe, sizeof(unsigned long long) * 4 * 16384) != hipSuccess) return QN_ERR_HIP; }
    if (v == 0 && c->list_probe) { (void)hipFree(c->list_probe); c->list_probe = nullptr; }
  }
  else if (k == "clk_probe") {                                  // developer probe: device-clock stamps inside k_tick (qn_debug_get_clk)
    if (v != 0 && !c->clk_probe) { if (hipMalloc(&c->clk_probe, 8 * 8 * 256 + 8 * 12 * 1024) != hipSuccess) return QN_ERR_HIP; }
    if (c->clk_probe) (void)hipMemset(c->clk_probe, 0, 8 * 8 * 256 + 8 * 12 * 1024);
    c->clk_n = 0; if (v == 0) { (void)hipFree(c->clk_probe); c->clk_probe = nullptr; }
  }
  else if (k == "tick_tb") c->tick_tb = v >= 512 ? 512 : 256;
  else if (k == "tick_ppt_min") c->tick_ppt_min = v < 1 ? 1u : (uint32_t)v;
  else if (k == "tick_rpb") c->tick_rpb = v < 1 ? 1u : (uint32_t)v;
  else if (k == "tick_lds_pad") c->tick_lds_pad = v < 0 ? 0 : (int)v;
  else if (k == "knn_lds_pad") c->knn_lds_pad = v < 0 ? 0 : (int)v;
  else if (k == "batch_min_share") c->batch_min_share = v < 1 ? 1 : (int)v;
  else if (k == "verify_track") {
    if (v != 0 && !c->v_counters) {
      const size_t n = c->max_points;
      if (hipMalloc(&c->v_corr, 4 * n) != hipSuccess || hipMalloc(&c->v_nn_idx, 4 * n) != hipSuccess || hipMalloc(&c->v_sqd, 4 * n) != hipSuccess ||
          hipMalloc(&c->v_nn_ref, 16 * n) != hipSuccess || hipMalloc(&c->v_counters, 16) != hipSuccess) return QN_ERR_HIP;
    }
    if (c->v_counters) (void)hipMemset(c->v_counters, 0, 16);
    c->verify_track = v != 0;
  }
  else return QN_ERR_INVALID_ARG;
  return QN_OK;
}
// developer / test entry point: the wave-level primitives of the search loops against plain restatements (qn_selftest.cuh); *mismatches = 0 when they agree
extern "C" int qn_debug_selftest(qn_ctx* c, uint32_t n_waves, uint32_t seed, uint32_t* mismatches) {
  if (!c || !mismatches || n_waves == 0 || n_waves > 65536) return QN_ERR_INVALID_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<uint32_t> h((size_t)n_waves * 128);
  uint64_t x = 0x9e3779b97f4a7c15ull ^ seed;
  for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)(x >> 16); }
  uint32_t* d = nullptr;
  HIPCHK(c, hipMalloc(&d, (h.size() + 1) * sizeof(uint32_t)));
  int rc = QN_OK; uint32_t bad = 0xffffffffu;
  if (hipMemcpyAsync(d, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream) != hipSuccess || hipMemsetAsync(d + h.size(), 0, sizeof(uint32_t), c->stream) != hipSuccess) rc = QN_ERR_HIP;
  if (rc == QN_OK) {
    hipLaunchKernelGGL(k_selftest_wave, dim3(n_waves), dim3(64), 0, c->stream, d, d + (size_t)n_waves * 64, d + h.size());
    if (hipMemcpyAsync(&bad, d + h.size(), sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) rc = QN_ERR_HIP;
  }
  (void)hipFree(d);
  if (rc != QN_OK) { c->last_error = "selftest: HIP call failed"; return rc; }
  *mismatches = bad;
  return QN_OK;
}
extern "C" int qn_debug_get_counters(qn_ctx* c, uint32_t out[16]) {
  if (!c || !c->dbg_counters) return QN_ERR_INVALID_ARG;
  if (hipStreamSynchronize(c->stream) != hipSuccess) return QN_ERR_HIP;
  if (hipMemcpy(out, c->dbg_counters, 16 * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) return QN_ERR_HIP;
  return QN_OK;
}
extern "C" int qn_debug_get(qn_ctx* c, const char* key, double* value) {
  if (!c || !key || !value) return QN_ERR_INVALID_ARG;
  const std::string k(key);
  if (k == "verify_mismatches" || k == "verify_passes" || k == "verify_first") {
    if (!c->v_counters) return QN_ERR_NOT_READY;
    uint32_t h[4];
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(h, c->v_counters, 16, hipMemcpyDeviceToHost) != hipSuccess) return QN_ERR_HIP;
    *value = k == "verify_mismatches" ? h[0] : (k == "verify_passes" ? h[1] : h[2]);
    return QN_OK;
  }
  if (k == "feat_mismatches" || k == "feat_verified" || k == "feat_first_mismatch") {
    if (!c->q_mm_vcnt) return QN_ERR_NOT_READY;
    uint32_t h[4];
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(h, c->q_mm_vcnt, 16, hipMemcpyDeviceToHost) != hipSuccess) return QN_ERR_HIP;
    *value = k == "feat_mismatches" ? h[0] : (k == "feat_verified" ? h[2] : h[1]);
    return QN_OK;
  }
  if (k == "quatro_wall_features_ms") { *value = c->q_wall_ms[0]; return QN_OK; }   // host wall clock of the latest Quatro align, by section
  if (k == "quatro_wall_match_ms") { *value = c->q_wall_ms[1]; return QN_OK; }
  if (k == "quatro_wall_solve_ms") { *value = c->q_wall_ms[2]; return QN_OK; }
  if (k == "extra_unseeded") { *value = c->last_extra_unseeded; return QN_OK; }          // the adaptive hand-over's decision in the latest align (ticks)
  if (k == "persist_launches") { *value = c->persist_launches; return QN_OK; }
  if (k == "persist_gave_up") { *value = c->persist_gave_up; return QN_OK; }      // persistent launches that gave up and were re-run on the k_tick chain
  if (k == "persist_fits") { *value = c->persist_fits ? 1 : 0; return QN_OK; }
  if (k == "persist_resident_blocks") { *value = c->persist_resident_blocks; return QN_OK; }
  if (k == "batch_launches") { *value = (double)c->batch_launches; return QN_OK; }      // kernel launches / pairs of the batched path so far (launches per registration = the ratio)
  if (k == "batch_pairs") { *value = (double)c->batch_pairs; return QN_OK; }
  if (k == "lanes_failed") { *value = c->lanes_failed ? 1.0 : 0.0; return QN_OK; }
  if (k == "batch_lanes") { *value = (double)c->batch_lanes; return QN_OK; }      // aligns of this context that ran the persistent kernel
  if (k == "feat_fallbacks") { *value = c->feat_fallbacks; return QN_OK; }      // matrix-core feature searches repeated with the VALU kernel (survivor overflow)
  if (k == "feat_survivors") { *value = c->feat_survivors; return QN_OK; }      // survivors of the latest forward search (exactly re-evaluated pairs)
  return QN_ERR_INVALID_ARG;
}
extern "C" int qn_debug_get_clk(qn_ctx* c, unsigned long long* out /* 256 x 8, then 1024 x 4 per-block stamps of the latest tick */, uint32_t* n) {
  if (!c || !out || !n || !c->clk_probe) return QN_ERR_INVALID_ARG;
  if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(out, c->clk_probe, 8 * 8 * 256 + 8 * 12 * 1024, hipMemcpyDeviceToHost) != hipSuccess) return QN_ERR_HIP;
  *n = c->clk_n; return QN_OK;
}
extern "C" int qn_debug_get_list_probe(qn_ctx* c, unsigned long long* out /* [4 * 16384] */) {
  if (!c || !out || !c->list_probe) return QN_ERR_INVALID_ARG;
  if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(out, c->list_probe, sizeof(unsigned long long) * 4 * 16384, hipMemcpyDeviceToHost) != hipSuccess) return QN_ERR_HIP;
  return QN_OK;
}
extern "C" int qn_debug_get_persist_clk(qn_ctx* c, unsigned long long* out /* 64 x 16 + 16 */) {
  if (!c || !out || !c->pg_clk) return QN_ERR_INVALID_ARG;
  if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(out, c->pg_clk, 8 * (64 * 16 + 16), hipMemcpyDeviceToHost) != hipSuccess) return QN_ERR_HIP;
  return QN_OK;
}
extern "C" int qn_debug_get_partials(qn_ctx* c, double* out /* 2 x (QN_ACC_MAX_BLOCKS + 8) x 28 */, uint32_t* rows_per_buffer, double* state /* 2 x sizeof(GicpState) / 8 */) {
  if (!c || !out || !rows_per_buffer || !state) return QN_ERR_INVALID_ARG;
  *rows_per_buffer = QN_ACC_MAX_BLOCKS + 8;
  if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(out, c->partials, 2 * sizeof(double) * (QN_ACC_MAX_BLOCKS + 8) * QN_NPART, hipMemcpyDeviceToHost) != hipSuccess
      || hipMemcpy(state, c->state, 2 * sizeof(GicpState), hipMemcpyDeviceToHost) != hipSuccess) return QN_ERR_HIP;
  return QN_OK;
}
extern "C" int qn_debug_get_grid(qn_ctx* c, int which, double out[8]) {
  if (!c || (which != 0 && which != 1) || !c->cloud[which].has_grid) return QN_ERR_INVALID_ARG;
  if (hipStreamSynchronize(c->stream) != hipSuccess || (c->stream2 && hipStreamSynchronize(c->stream2) != hipSuccess)) return QN_ERR_HIP;
  const GridDims& g = *c->cloud[which].dims_host;
  out[0] = g.ox; out[1] = g.oy; out[2] = g.oz; out[3] = g.cell; out[4] = g.nx; out[5] = g.ny; out[6] = g.nz; out[7] = g.eps;
  return QN_OK;
}

#include "qn_batch.inc"
#include "qn_quatro_host.inc"
