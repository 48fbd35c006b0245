// qn_multi.hip - candidate pairs of one loop-closure query sharded over the GPUs of a node (BASELINE configs[3]).
//
// Every candidate pair is an independent LoopClosure::icpAlignment (fast_lio_sam_qn/src/loop_closure.cpp:116-123 rebuilds trees and
// covariances per call; the fan-out point in the reference is fast_lio_sam_qn.cpp:213-219, where ONE candidate is registered per timer
// tick), so the data path needs no collective: pair i runs on GPU i mod N, on one of `in_flight` contexts (= hipStreams) of that GPU.
// The single exchange step is the gather of the fixed-size result records (<= 96 bytes each) so that the host can pick the winning
// loop: ONE ncclAllGather over xGMI.  Point clouds never move between GPUs.  Single process, all GPUs (ncclCommInitAll) - the
// reference is a single process too.  Host code only: this unit launches no kernel of its own; it drives the C-ABI of qn_engine.h.
//
// RCCL is loaded with dlopen at qn_multi_init, so libqn_engine.so itself carries no load-time dependency on librccl: the single-GPU
// entry points work on a box without RCCL, qn_multi_init fails loudly there.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <atomic>
#include <chrono>
#include <cfloat>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "../../include/qn_engine.h"
#include "qn_pool.h"

static_assert(sizeof(qn_pair_record) == 96, "result records are 96 bytes (SURVEY 8e)");

struct qn_multi {
  int n_gpus = 0, in_flight = 1;
  uint32_t max_points = 0;
  std::vector<int> dev;
  std::vector<std::vector<qn_ctx*>> ctx;          // [gpu][slot]
  std::vector<ncclComm_t> comm;
  std::vector<hipStream_t> stream;
  std::vector<qn_pair_record*> d_send, d_recv;    // per GPU: its own records / everybody's
  uint32_t rec_cap = 0;                           // records per GPU the device buffers hold
  qn_pair_record* h_all = nullptr;                // pinned: gathered table as GPU 0 received it
  void* rccl = nullptr;
  decltype(&ncclCommInitAll) p_init_all = nullptr;
  decltype(&ncclCommDestroy) p_destroy = nullptr;
  decltype(&ncclAllGather) p_all_gather = nullptr;
  decltype(&ncclGroupStart) p_group_start = nullptr;
  decltype(&ncclGroupEnd) p_group_end = nullptr;
  decltype(&ncclGetErrorString) p_err = nullptr;
  decltype(&ncclCommCount) p_count = nullptr;
  uint32_t last_per = 0;                          // records per GPU of the latest gather (qn_multi_verify_gather)
  std::vector<double> gpu_ms; double gather_ms = 0;   // timing of the latest qn_multi_align_best: per GPU (first pair start .. last pair end), the gather
  bool quatro_on = false;                         // enable_quatro_: pairs are coarse-to-fine registrations (qn_multi_set_quatro_params)
  bool poisoned = false;                          // a collective failed: the communicator state is undefined, every later call is refused
  std::string last_error;
};

static thread_local std::string g_init_error;
extern "C" const char* qn_multi_last_error(const qn_multi* m) { return m ? m->last_error.c_str() : g_init_error.c_str(); }

extern "C" void qn_multi_destroy(qn_multi* m) {
  if (!m) return;
  for (int g = 0; g < (int)m->ctx.size(); g++) for (qn_ctx* c : m->ctx[g]) qn_ctx_destroy(c);
  for (int g = 0; g < (int)m->comm.size(); g++) if (m->comm[g] && m->p_destroy) { hipSetDevice(m->dev[g]); m->p_destroy(m->comm[g]); }
  for (int g = 0; g < (int)m->dev.size(); g++) {
    hipSetDevice(m->dev[g]);
    if (g < (int)m->d_send.size()) { hipFree(m->d_send[g]); hipFree(m->d_recv[g]); }
    if (g < (int)m->stream.size() && m->stream[g]) hipStreamDestroy(m->stream[g]);
  }
  if (m->h_all) hipHostFree(m->h_all);
  if (m->rccl) dlclose(m->rccl);
  delete m;
}

static int multi_reserve(qn_multi* m, uint32_t per_gpu) {
  if (per_gpu <= m->rec_cap) return QN_OK;
  const uint32_t cap = per_gpu + 8;
  for (int g = 0; g < m->n_gpus; g++) {
    if (hipSetDevice(m->dev[g]) != hipSuccess) return QN_ERR_HIP;
    hipFree(m->d_send[g]); hipFree(m->d_recv[g]); m->d_send[g] = m->d_recv[g] = nullptr;
    if (hipMalloc(&m->d_send[g], sizeof(qn_pair_record) * cap) != hipSuccess || hipMalloc(&m->d_recv[g], sizeof(qn_pair_record) * cap * m->n_gpus) != hipSuccess) { m->last_error = "hipMalloc of the record buffers failed"; return QN_ERR_HIP; }
  }
  if (m->h_all) hipHostFree(m->h_all);
  m->h_all = nullptr;
  if (hipHostMalloc(&m->h_all, sizeof(qn_pair_record) * cap * m->n_gpus, hipHostMallocDefault) != hipSuccess) { m->last_error = "hipHostMalloc failed"; return QN_ERR_HIP; }
  m->rec_cap = cap;
  return QN_OK;
}

extern "C" int qn_multi_init(int n_gpus, const int* device_ids, uint32_t max_points, int in_flight, qn_multi** out) {
  if (!out || n_gpus < 1 || max_points == 0 || in_flight < 1 || in_flight > 64) return QN_ERR_INVALID_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < n_gpus) {
    char buf[160]; snprintf(buf, sizeof(buf), "qn_multi_init: %d GPUs requested, %d visible (no CPU fallback, no GPU sharing between ranks)", n_gpus, ndev < 0 ? 0 : ndev);
    g_init_error = buf; return QN_ERR_NO_DEVICE;
  }
  qn_multi* m = new qn_multi();
  m->n_gpus = n_gpus; m->in_flight = in_flight; m->max_points = max_points;
  for (int g = 0; g < n_gpus; g++) {
    const int d = device_ids ? device_ids[g] : g;
    if (d < 0 || d >= ndev) { g_init_error = "qn_multi_init: device id out of range"; qn_multi_destroy(m); return QN_ERR_NO_DEVICE; }
    for (int e : m->dev) if (e == d) { g_init_error = "qn_multi_init: the same device listed twice (one rank per GPU)"; qn_multi_destroy(m); return QN_ERR_INVALID_ARG; }
    m->dev.push_back(d);
  }
  auto fail = [&](int code, const std::string& why) { g_init_error = why; qn_multi_destroy(m); return code; };
  m->rccl = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!m->rccl) m->rccl = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!m->rccl) m->rccl = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!m->rccl) return fail(QN_ERR_HIP, std::string("qn_multi_init: cannot load RCCL: ") + dlerror());
#define QN_SYM(field, name) m->field = (decltype(m->field))dlsym(m->rccl, name); if (!m->field) return fail(QN_ERR_HIP, std::string("qn_multi_init: RCCL lacks ") + name)
  QN_SYM(p_init_all, "ncclCommInitAll"); QN_SYM(p_destroy, "ncclCommDestroy"); QN_SYM(p_all_gather, "ncclAllGather");
  QN_SYM(p_group_start, "ncclGroupStart"); QN_SYM(p_group_end, "ncclGroupEnd"); QN_SYM(p_err, "ncclGetErrorString"); QN_SYM(p_count, "ncclCommCount");
#undef QN_SYM
  m->ctx.resize(n_gpus); m->stream.assign(n_gpus, nullptr); m->d_send.assign(n_gpus, nullptr); m->d_recv.assign(n_gpus, nullptr);
  for (int g = 0; g < n_gpus; g++) {
    if (hipSetDevice(m->dev[g]) != hipSuccess || hipStreamCreateWithFlags(&m->stream[g], hipStreamNonBlocking) != hipSuccess) return fail(QN_ERR_HIP, "qn_multi_init: stream creation failed");
    for (int s = 0; s < in_flight; s++) {
      qn_ctx* c = nullptr;
      const int rc = qn_ctx_create(m->dev[g], max_points, &c);
      if (rc != QN_OK) return fail(rc, std::string("qn_multi_init: qn_ctx_create: ") + qn_status_str(rc));
      if (in_flight > 1) { (void)qn_debug_set(c, "pair_pipeline", 0.0); (void)qn_debug_set(c, "batch_member", 1.0); }      // the streams in flight fill the chip: no second stream per registration, no persistent launches
      m->ctx[g].push_back(c);
    }
  }
  m->comm.assign(n_gpus, nullptr);
  const ncclResult_t nr = m->p_init_all(m->comm.data(), n_gpus, m->dev.data());      // one communicator over the node's GPUs (xGMI)
  if (nr != ncclSuccess) return fail(QN_ERR_HIP, std::string("qn_multi_init: ncclCommInitAll: ") + m->p_err(nr));
  const int rc = multi_reserve(m, 64);
  if (rc != QN_OK) return fail(rc, m->last_error);
  *out = m;
  return QN_OK;
}

extern "C" int qn_multi_set_params(qn_multi* m, const qn_gicp_params* p) {
  if (!m || !p) return QN_ERR_INVALID_ARG;
  for (auto& v : m->ctx) for (qn_ctx* c : v) { const int rc = qn_gicp_set_params(c, p); if (rc != QN_OK) return rc; }
  return QN_OK;
}

extern "C" int qn_multi_set_quatro_params(qn_multi* m, const qn_quatro_params* p) {      // loop_closure.cpp:18-27 on every context; NULL: back to Nano-GICP only
  if (!m) return QN_ERR_INVALID_ARG;
  if (!p) { m->quatro_on = false; return QN_OK; }
  for (auto& v : m->ctx) for (qn_ctx* c : v) { const int rc = qn_quatro_set_params(c, p); if (rc != QN_OK) return rc; }
  m->quatro_on = true;
  return QN_OK;
}

extern "C" int qn_multi_debug_set(qn_multi* m, const char* key, double value) {      // a developer knob (qn_debug_set) on every context, e.g. "batch_lanes"
  if (!m || !key) return QN_ERR_INVALID_ARG;
  for (auto& v : m->ctx) for (qn_ctx* c : v) { const int rc = qn_debug_set(c, key, value); if (rc != QN_OK) return rc; }
  return QN_OK;
}
extern "C" int qn_multi_gpu_count(const qn_multi* m) { return m ? m->n_gpus : 0; }
extern "C" int qn_multi_rccl_ranks(qn_multi* m) {      // what RCCL says, not what the caller asked for
  if (!m || m->comm.empty() || !m->p_count) return -1;
  int least = 1 << 30;
  for (int g = 0; g < m->n_gpus; g++) {
    int cnt = -1;
    if (!m->comm[g] || m->p_count(m->comm[g], &cnt) != ncclSuccess) { m->last_error = "qn_multi_rccl_ranks: ncclCommCount failed"; return -1; }
    least = cnt < least ? cnt : least;
  }
  return least;
}
extern "C" int qn_multi_verify_gather(qn_multi* m) {
  if (!m) return QN_ERR_INVALID_ARG;
  if (m->poisoned || m->last_per == 0 || !m->h_all) return QN_ERR_NOT_READY;
  const size_t bytes = sizeof(qn_pair_record) * m->last_per * (size_t)m->n_gpus;
  std::vector<char> got(bytes);
  for (int g = 1; g < m->n_gpus; g++) {
    if (hipSetDevice(m->dev[g]) != hipSuccess || hipMemcpy(got.data(), m->d_recv[g], bytes, hipMemcpyDeviceToHost) != hipSuccess) { m->last_error = "qn_multi_verify_gather: download failed"; return QN_ERR_HIP; }
    if (memcmp(got.data(), m->h_all, bytes) != 0) { char buf[120]; snprintf(buf, sizeof(buf), "qn_multi_verify_gather: GPU %d holds a different record table than GPU 0", g); m->last_error = buf; return QN_ERR_HIP; }
  }
  return QN_OK;
}
extern "C" int qn_multi_get_timing(const qn_multi* m, double* per_gpu_ms, double* gather_ms) {
  if (!m || !per_gpu_ms || !gather_ms) return QN_ERR_INVALID_ARG;
  if ((int)m->gpu_ms.size() != m->n_gpus) return QN_ERR_NOT_READY;
  for (int g = 0; g < m->n_gpus; g++) per_gpu_ms[g] = m->gpu_ms[g];
  *gather_ms = m->gather_ms;
  return QN_OK;
}

// pair i -> GPU i mod N (clouds are read where the caller put them: host memory, or - on_device - memory of THAT GPU); every GPU
// registers its pairs on `in_flight` streams; one ncclAllGather of the per-GPU record tables; the winner is the valid record
// (converged && score < score_thr, loop_closure.cpp:129) with the smallest score, ties to the lowest pair id.
extern "C" int qn_multi_align_best(qn_multi* m, const qn_pair_desc* pairs, uint32_t n_pairs, double score_thr,
                                   qn_pair_record* records, qn_pair_record* best, int* best_found) {
  if (!m || (n_pairs && !pairs) || !best || !best_found) return QN_ERR_INVALID_ARG;
  *best_found = 0; memset(best, 0, sizeof(*best)); best->pair_id = -1; best->fitness = DBL_MAX;
  if (m->poisoned) { m->last_error = "qn_multi handle poisoned by an earlier failed collective: destroy it and create a new one"; return QN_ERR_HIP; }
  if (n_pairs == 0) return QN_OK;
  const int N = m->n_gpus;
  const uint32_t per = (n_pairs + N - 1) / N;
  int rc = multi_reserve(m, per);
  if (rc != QN_OK) return rc;
  // ---- the data path: independent registrations, no collective
  std::vector<std::vector<qn_pair_record>> mine(N, std::vector<qn_pair_record>(per));
  for (int g = 0; g < N; g++) for (uint32_t l = 0; l < per; l++) { qn_pair_record& r = mine[g][l]; memset(&r, 0, sizeof(r)); r.pair_id = -1; r.status = QN_ERR_EMPTY_CLOUD; r.fitness = DBL_MAX; }
  // each GPU's pairs go through qn_icp_alignment_batch on that GPU's `in_flight` contexts: every context registers runs of pairs in lockstep, the pair as a
  // grid dimension of every kernel launch (qn_gicp_align_batch); candidates that share their source buffer share its grid and covariances per lane
  using clk = std::chrono::steady_clock;
  const clk::time_point t_start = clk::now();
  std::vector<long long> t_end(N, 0);
  std::vector<int> gpu_rc(N, QN_OK);
  auto gpu_worker = [&](int g) {
    std::vector<qn_pair_desc> mp; std::vector<uint32_t> ids;
    for (uint32_t l = 0; l < per; l++) { const uint64_t i = (uint64_t)g + (uint64_t)l * N; if (i < n_pairs) { mp.push_back(pairs[i]); ids.push_back((uint32_t)i); } }
    std::vector<qn_gicp_result> res(mp.size()); std::vector<int> val(mp.size(), 0), st(mp.size(), QN_ERR_HIP);
    std::vector<double> Tt;
    if (!mp.empty() && m->quatro_on) {                                          // coarseToFineAlignment per pair (loop_closure.cpp:188-192 with enable_quatro_)
      Tt.resize(16 * mp.size());
      gpu_rc[g] = qn_coarse_to_fine_align_batch(m->ctx[g].data(), (uint32_t)m->ctx[g].size(), mp.data(), (uint32_t)mp.size(), score_thr, res.data(), Tt.data(), nullptr, val.data(), st.data());
      for (size_t l = 0; l < mp.size(); l++) for (int k = 0; k < 16; k++) res[l].T[k] = (float)Tt[16 * l + k];      // the record's T = pose_between (:156), cast to the record's f32
    }
    else if (!mp.empty()) gpu_rc[g] = qn_icp_alignment_batch(m->ctx[g].data(), (uint32_t)m->ctx[g].size(), mp.data(), (uint32_t)mp.size(), score_thr, res.data(), val.data(), st.data());
    for (size_t l = 0; l < mp.size(); l++) {
      qn_pair_record& r = mine[g][l];
      const int s_ = gpu_rc[g] == QN_OK ? st[l] : gpu_rc[g];
      r.pair_id = (int32_t)ids[l]; r.status = s_; r.valid = (s_ == QN_OK && val[l]) ? 1 : 0; r.converged = s_ == QN_OK ? res[l].converged : 0;
      r.iterations = s_ == QN_OK ? res[l].iterations : 0; r.fitness = s_ == QN_OK ? res[l].fitness : DBL_MAX;
      if (s_ == QN_OK) memcpy(r.T, res[l].T, sizeof(r.T)); else for (int k = 0; k < 16; k++) r.T[k] = (k % 5 == 0) ? 1.f : 0.f;
    }
    t_end[g] = std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - t_start).count();
  };
  qn::WorkerPool::instance().run((uint32_t)N, [&](uint32_t g) { gpu_worker((int)g); });      // one parked worker per GPU (qn_pool.h); each fans out over its contexts
  m->gpu_ms.assign(N, 0.0);
  for (int g = 0; g < N; g++) m->gpu_ms[g] = 1e-6 * (double)t_end[g];
  const clk::time_point t_gather = clk::now();
  // ---- the one exchange step: gather the record tables (N x per x 96 bytes: latency-bound, ring or tree does not matter at this size)
  // A failure after the first asynchronous operation must not return while copies out of `mine` (pageable host memory) may still be staging:
  // every per-GPU stream is drained first.  A failed collective leaves the communicator in an undefined state: the handle is poisoned.
  auto bail = [&](const std::string& why, bool poison) {
    for (int g = 0; g < N; g++) { if (hipSetDevice(m->dev[g]) == hipSuccess) (void)hipStreamSynchronize(m->stream[g]); }
    m->last_error = why; if (poison) m->poisoned = true;
    return QN_ERR_HIP;
  };
  for (int g = 0; g < N; g++) {
    if (hipSetDevice(m->dev[g]) != hipSuccess || hipMemcpyAsync(m->d_send[g], mine[g].data(), sizeof(qn_pair_record) * per, hipMemcpyHostToDevice, m->stream[g]) != hipSuccess) return bail("upload of the record table failed", false);
  }
  ncclResult_t nr = m->p_group_start();
  for (int g = 0; g < N && nr == ncclSuccess; g++) nr = m->p_all_gather(m->d_send[g], m->d_recv[g], sizeof(qn_pair_record) * per, ncclChar, m->comm[g], m->stream[g]);
  const ncclResult_t ne = m->p_group_end();
  if (nr == ncclSuccess) nr = ne;
  if (nr != ncclSuccess) return bail(std::string("ncclAllGather: ") + m->p_err(nr) + " (the communicator is unusable: destroy this qn_multi)", true);
  if (hipSetDevice(m->dev[0]) != hipSuccess || hipMemcpyAsync(m->h_all, m->d_recv[0], sizeof(qn_pair_record) * per * N, hipMemcpyDeviceToHost, m->stream[0]) != hipSuccess) return bail("download of the gathered table failed", false);
  for (int g = 0; g < N; g++) { if (hipSetDevice(m->dev[g]) != hipSuccess || hipStreamSynchronize(m->stream[g]) != hipSuccess) return bail("stream synchronisation after the gather failed", true); }
  m->gather_ms = 1e-6 * (double)std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - t_gather).count();
  m->last_per = per;
  // ---- the winner, from the GATHERED table (what rank 0 sees after the collective)
  int status = QN_OK;
  for (uint32_t e = 0; e < per * (uint32_t)N; e++) {
    const qn_pair_record& r = m->h_all[e];
    if (r.pair_id < 0) continue;
    if (records) records[r.pair_id] = r;
    if (r.status != QN_OK && r.status != QN_ERR_EMPTY_CLOUD) status = r.status;
    if (r.valid && (r.fitness < best->fitness || (r.fitness == best->fitness && r.pair_id < best->pair_id) || !*best_found)) { *best = r; *best_found = 1; }
  }
  return status;
}
