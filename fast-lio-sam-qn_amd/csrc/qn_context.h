// qn_context.h - the opaque qn_ctx behind include/qn_engine.h (host-side bookkeeping only).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "qn_gicp_kernels.cuh"

struct CloudBuf {                       // one point cloud, resident in HBM
  uint32_t n = 0, ncells = 0;
  bool has_grid = false, has_cov = false;
  float4* raw = nullptr;                // [max_points] original order, (x, y, z, 1)
  float4* sorted = nullptr;             // [max_points] cell-sorted, .w = original index bits
  float4* sorted_tmp = nullptr;         // [max_points] the scatter's output before the cells are put in a run-independent order (k_stable_cells)
  uint32_t* cell_of_pt = nullptr;       // [max_points]
  uint32_t* cell_start = nullptr;       // [max_cells + 1]
  uint32_t* counts = nullptr;           // [max_cells + 1] histogram / scatter cursors
  double* nrm = nullptr;                // [max_points][3] plane normal (f64), original order: C = I - 0.999 n n^T
  qn::GridView grid{};                  // pointers + the pointer to the device-side numbers (`dims`); the numeric fields are NOT valid on the host
  qn::GridDims* dims = nullptr;         // device: the grid's numbers (k_grid_dims)
  qn::GridDims* dims_host = nullptr;    // pinned mirror, copied behind every grid build: valid after the next synchronisation of the stream that built the grid
};

struct ProfSpan { int family; hipEvent_t a, b; int count; };      // count: registrations a batched launch (k_lanes) carried - the family's `launches` are counted per registration

struct qn_ctx {
  int device = 0;
  uint32_t max_points = 0, max_cells = 0;
  hipStream_t stream = nullptr; bool owns_stream = true;      // (the lanes of a batch context work on their owner's stream)
  // pair-as-grid-dimension batches (qn_gicp_align_batch, qn_batch.inc): lane 0 is this context, lanes 1 .. are sub-contexts with buffers of their own;
  // every kernel of the chain is launched ONCE for all lanes (k_lanes<F>) with its per-lane arguments in a device-resident table (args_d, staged in args_h)
  std::vector<qn_ctx*> lanes; int batch_lanes = 8; char* args_h = nullptr; char* args_d = nullptr; size_t args_cap = 0; uint64_t batch_launches = 0, batch_pairs = 0; bool is_lane = false, batch_trace = false, batch_share_source = true;      // batch_share_source: pairs of one batch call that name the same source buffer share its grid and covariances (the candidates of ONE query); off = every pair rebuilds its source like loop_closure.cpp:120-121
  bool lanes_failed = false; int fail_lane_create = 0;      // lanes_failed: creating this context's lanes failed once (out of memory, typically): the batch entry points take the one-pair path until batch_lanes is set again - no
                                                             // retry (GBs of hipMalloc / hipFree) per call; fail_lane_create: test knob - the creation of lane index >= this fails (0 = off)
  void* slab = nullptr;                 // ONE device allocation behind every per-context buffer of the GICP path (qn_ctx_create)
  qn_gicp_params params{};
  CloudBuf cloud[2];
  qn::BBoxAcc* bbox_acc = nullptr; qn::BBoxAcc* bbox_acc2 = nullptr;     // bounding-box accumulators (k_pack_bbox_dims), one per stream
  unsigned long long* scan_status = nullptr; unsigned long long* scan_status2 = nullptr; uint32_t build_epoch = 0;   // look-back scan: tile status words, tagged with the build's epoch
  int far_group = -1;                   // far-list grouping (wave_search_far16): -1 = by regime (enqueue_nn), 0 = off, n = ceil(list length / n) entries per wave for every list
  int far_chunk = 2;                    // tracked ticks of a forced run that keep the far-query refresh kernel behind them before the host looks again (round 5: 4 -> 2: the refresh regime is over after
                                        // two ticks on the 80 %-overlap pairs - 2332 -> 2400 registrations/s on 3 x 8, lone registration 1.279 -> 1.252 ms; aligned pairs unchanged; 1 and 3 measured between)
  bool batch_look = false;              // batch members take the device-side look as well (third unseeded iteration conditional): measured neutral for throughput (2239-2253 vs 2246-2247), off
  bool far_ranked = true;               // k_far deals its requests by global rank (off: the word-per-block distribution that clouds beyond 262144 points use)
  unsigned long long* list_probe = nullptr;   // developer probe (knob list_probe)
  bool device_look = true;              // the hand-over decision of a lone forced-GN registration on the device (look_decide) instead of a host round trip
  bool clear_far_now = false;           // the next unseeded search resets the far-candidate references (first search of an align)
  char* staging2 = nullptr;             // the second stream's landing zone (TargetScope)
  // page-locked HOST landing zones (one per device landing zone; allocated on the first pageable host cloud): a pageable caller buffer is packed into it by the CPU (xyz, 12 B per
  // point) and crosses PCIe from there.  hipMemcpyAsync straight from pageable memory PINS the caller's pages for copies above ~1 MB; a caller that hands over a freshly allocated
  // cloud every call - LoopClosure::icpAlignment does, loop_closure.cpp:116-119 - paid 14-27 ms for that on every other call at 100k points (round 6, tests/shim_icp_alignment bench)
  struct PinZone { char* h = nullptr; hipEvent_t ev = nullptr; bool busy = false; } pin_up, pin_up2;
  char* staging = nullptr;              // [max_points * 32] H2D landing zone for strided host clouds
  uint32_t* scan_sums = nullptr;

  qn::GicpState* state = nullptr;       // [2], double buffered: generation g in state[g & 1] (qn_gicp_kernels.cuh)
  uint32_t gen = 0; int part_rows = 0;  // current generation; rows of the partial buffer written under it
  double* partials = nullptr;           // [2][QN_ACC_MAX_BLOCKS][28]
  uint32_t* tail_ticket = nullptr;      // the last-block ticket of the controller tail (controller_tail, qn_gicp_kernels.cuh)
  double* fit_psum = nullptr; uint32_t* fit_pcnt = nullptr;
  qn_iter_trace* trace = nullptr; uint32_t trace_len = 0;
  int32_t* corr = nullptr; int32_t* nn_idx = nullptr; int32_t* knn_idx = nullptr; float4* nn_ref = nullptr; double* nrm_s_sorted = nullptr; qn::TargetRec* tgt_rec = nullptr; float* sqd = nullptr; float* sqd_fit = nullptr;
  // far queries (qn_tick.cuh): candidate cache, refresh request bits, counters; far_mode: 0 off, 1 refresh kernel after every tick, 2 cache only
  int32_t* far_cand = nullptr; float4* far_cand_ref = nullptr; float2* far_cand_b = nullptr; unsigned long long* far_req = nullptr; uint32_t* far_stats = nullptr; double* far_rows = nullptr; int far_mode = 2; bool far_enabled = true;
  uint2* fb_list = nullptr; uint2* big_list = nullptr; uint32_t* fb_count2 = nullptr;
  float4* aligned = nullptr; bool aligned_valid = false;
  double* pose_tmp = nullptr; float* guess_tmp = nullptr;
  qn::ResultBlock* result_host = nullptr; double* scalar_host = nullptr;
  int32_t* dbg_knn_idx = nullptr; float* dbg_knn_d2 = nullptr;
  // Quatro (allocated on first use)
  qn_quatro_params qparams{}; bool qparams_set = false, q_ready = false; double q_last_scale = 1.0;      // q_last_scale: the scale of the latest host solve (1 unless estimate_scale)
  float4* q_normals[2] = {nullptr, nullptr}; float* q_spfh[2] = {nullptr, nullptr}; float* q_fpfh_s[2] = {nullptr, nullptr}; float* q_fpfh[2] = {nullptr, nullptr};
  unsigned long long* q_key[2] = {nullptr, nullptr};
  float* q_pair[2] = {nullptr, nullptr}; uint32_t* q_pair_hash[2] = {nullptr, nullptr};   // descriptors in candidate-pair layout + row hashes (k_feat_nn)
  uint32_t* q_hit = nullptr; uint32_t* q_list = nullptr; uint32_t* q_sel = nullptr; uint2* q_pairs = nullptr; uint32_t* q_counts = nullptr; double* q_T = nullptr;
  // feature matching on the matrix cores (qn_feat_mm.cuh): f16 operand images (candidates, queries), |q'|^2, lower bounds, de-duplication table, survivors
  void* q_mm_c = nullptr; void* q_mm_q = nullptr; float* q_mm_qn = nullptr; uint32_t* q_mm_L = nullptr; unsigned long long* q_mm_table = nullptr; unsigned long long* q_mm_table_q = nullptr; bool feat_query_dedupe = false;   // (knob: measured no gain - in the forward search the QUERIES are the smaller or, at equal size, the TARGET cloud, whose rows carry the sensor noise: no duplicates to remove)
  uint32_t q_mm_mask = 0;
  uint2* q_mm_pairs = nullptr; uint32_t* q_mm_cnt = nullptr; uint32_t q_mm_cap = 0, q_mm_ncnt = 0; bool feat_mfma = true, feat_mm_retry = false; int feat_sample = 4 /* = QN_MM_SAMPLE */; bool feat_verify = false; int feat_min_blocks = 1536; double q_wall_ms[3] = {0, 0, 0}; unsigned long long* q_mm_vkeys = nullptr; uint32_t* q_mm_vcnt = nullptr; uint32_t feat_fallbacks = 0, feat_survivors = 0;
  bool c2f_overlap = false;            // (measured neutral, 1.386 vs 1.393 ms per 30k pair: icpAlignment already prepares the target on the second stream beside the source's k-NN - off, the proven order)              // coarseToFineAlignment: the fine stage's TARGET preparation (grid, k-NN, covariances) is enqueued behind Quatro's matching instead of after its host solve
  bool c2f_lanes_fpfh = false;         // batched coarse-to-fine: grid builds and K9-K11 of every lane's clouds in nine k_lanes launches per run instead of eighteen launches per pair.  Built, bit-identical,
                                       // measured (64 true-loop 30k pairs): 4 x 8 1400 vs 1405, 8 x 4 1462 vs 1466, 3 x 8 1358 vs 1398 pairs/s - the FPFH kernels are VALU-issue-bound, a shared launch buys them nothing: off
  bool tgt_early = true;               // icpAlignment in one call: the target's grid build is enqueued before the source's k-NN launches (second stream busy ~50 us earlier)
  bool quatro_fused = true;            // the matching stage's bookkeeping (memsets, fills, row hashes, hit marking, means) in four fused launches instead of nineteen
  int normals_fg = 0, fpfh_fg = 0;     // lanes per query of k_normals (1 / 8 / 16) and of k_spfh + k_fpfh (8 / 16); 0 = by cloud size (quatro_fpfh)
  float4* c2f_src = nullptr; float4* c2f_dst = nullptr;   // batched coarse-to-fine (qn_coarse_to_fine_align_batch): this lane's coarse-aligned source (transformPcd, loop_closure.cpp:152) and its target, float4 in caller order, until the GICP lanes have packed them
  float* q_mean = nullptr; double* q_mean_psum = nullptr; void* q_host = nullptr;   // Matcher tail: cloud means, pinned hand-over block (header + one record per selected correspondence)
  // tuning knobs
  double cell_override = 0.0;
  int list_small = 6000;                // lone registration: a 16-per-wave leftover list of at most this many entries is served one entry per wave (k_nn_search LIST)
  float big_ratio = 2.5f;               // first-search leftovers whose next radius exceeds big_ratio * r0 go one-per-wave
  bool fused_ticks = true;              // GN ticks >= 3: tracking + leftovers + accumulation in one kernel
  int nn_rounds = 1;                    // rounds of an unseeded NN search before a query goes to the list pass
  int nn_lane = -1;                     // the first pass of an unseeded search runs one query per lane over its own 3 x 3 x 3 cell box (NnLaneK): -1 = batch members only, 0 = never, 1 = always
  int track_from_tick = 3;              // NN passes before this tick search unseeded (ball around the query) instead of tracking the previous neighbour
  int fused_from_tick = 3;
  int unseeded_until = 3; bool count_far_now = false; int last_extra_unseeded = 0;   // per align: ticks below this index search unseeded; far-query statistics of this pass; the adaptive decision (debug read-back)
  int unseeded_cap = 10;                // batch members: a lane whose latest pose step still moves the source by more than 0.4 target cells at the end of its unseeded ticks runs ONE MORE outer
                                        // iteration unseeded and the host looks again, up to this many unseeded outer iterations per align (<= track_from_tick: the fixed schedule).  A tracked tick behind a
                                        // step of metres re-searches every neighbourhood cooperatively inside the tick (100-1800 us per 4-lane launch, and the launch ends with its slowest lane); the
                                        // dedicated unseeded search does not care (round 6: an 8-pair call of the hardest re-pose variant 5.64 -> 3.2 ms)
  int single_from_tick = 2;             // the same hand-over for a registration that is alone on the GPU (not a batch member): one tick earlier - the tracked tick is the slower
                                        // kernel for the large early steps, but it saves four launches and the persistent kernel starts sooner (align 0.506 -> 0.477 ms; batches: 2268 -> 2031 /s)
  bool fused_final = true;              // closing pass (last controller step + fitness sweep + output cloud) in one launch
  int knn_rounds = 2;                   // rounds of the first k-NN pass before a query goes to the list pass
  int knn_mm = 1;                       // k-NN selection: squared distances of the two passes on the matrix cores (v_mfma_f32_16x16x4_f32 screen + exact re-evaluation of the listed candidates); 0 = VALU scoring (k <= 24 only)
  int knn_trips = 3;                    // batched launches: groups of 16 queries a wave of the k-NN selection pass serves (grid = n / 64 / knn_trips blocks)
  int knn_hist = 1;                     // 1: k-NN by histogram selection (wave_knn_hist), 0: sorted-list sink (wave_search + BestK)
  int big_blocks0 = 4096, fb_blocks0 = 512;   // grid of the list pass behind the first (unseeded) ticks: one-far-query-per-wave blocks, wave-stride leftover blocks
  float margin_nn_t0 = 0.f;             // first search radius of tick 0 only (0 = margin_nn)
  float margin_nn = 1.f, margin_knn = 0.f;   // first search radius in cells (1-NN of the first tick / k-NN of the covariances; 0 = by cloud size, launch_knn_cov)
  int margin_nn_cap = 3, margin_knn_cap = 5, ticks_per_chunk = 8; bool knn_single_all = false, stable_cells = true; int bbox_blocks = 64;
  // pair pipeline of icpAlignment: the target cloud is prepared on a second stream with its own scratch while the source's k-NN runs
  hipStream_t stream2 = nullptr; hipEvent_t ev_pair = nullptr; bool pair_pipeline = true, pair_failed = false, tgt_on_stream2 = false, tgt_pending = false, no_pipe = false;
  uint32_t* scan_sums2 = nullptr; uint2* fb_list2 = nullptr; uint2* big_list2 = nullptr; uint32_t* fb_count2b = nullptr; int32_t* knn_idx2 = nullptr;   // 128 blocks = 768 atomics on six words: 10.6 us per cloud; 32: 5 us   // experiment: one selection round, every leftover to the one-query-per-wave pass
  uint32_t tick_ppt_min = 1;            // source points per lane of k_tick (knob: fewer, longer blocks)
  uint32_t tick_rpb = 2;                // batch members: partial rows a k_tick block forms, one after the other (rows and results are those of 1; the launch has half the blocks)
  uint32_t tick_tb = 512;               // threads per block of k_tick (and of k_solve: both run the same row reduction)
  int batch_min_share = 4;              // the last round of a multi-context batch call is dealt in equal shares, but not smaller than this many pairs per context (qn_icp_alignment_batch)
  int knn_lds_pad = 0;                  // experiment: dynamic LDS bytes added to the batched k-NN selection launches (one-wave blocks, 8.5 KB each: 4300 -> 12 instead of 16 per CU, i.e. the issue-bound
                                        // selection leaves a wave slot per SIMD to the other contexts' latency-bound kernels, which otherwise queue behind it - profiles/r5_final_share8_timeline.txt)
  int tick_lds_pad = 0;                 // experiment: dynamic LDS bytes added to the batched k_tick launches (40000 = one block per CU: the latency-bound tick then leaves half of every CU's register file to the other contexts' kernels)
  int tick_occ = 4;                     // k_tick variant: waves per SIMD the register budget allows (4 = 128 VGPRs: other streams' kernels keep half of the register file)
  // persistent align kernel (qn_persist.cuh): granule buffers, give-up status, epoch counter; `persist` = knob, `persist_batch_off` = this context works in a batch
  unsigned long long* pg_rows = nullptr; unsigned long long* pg_bc = nullptr; unsigned long long* pg_fit = nullptr; uint32_t* pg_status = nullptr; uint32_t* pg_status_host = nullptr;
  unsigned long long* pg_clk = nullptr; uint32_t pg_epoch = 0; bool prof_persist = false, persist = true, persist_batch_off = false, persist_fits = true; int persist_resident_blocks = 0; uint32_t persist_launches = 0, persist_gave_up = 0; unsigned long long persist_timeout = 25000000ull;
  uint32_t* dbg_counters = nullptr;
  unsigned long long* clk_probe = nullptr; uint32_t clk_n = 0;   // developer probe: device-clock stamps of k_tick
  // verify_track (debug): scratch of the fresh search every tracked pass is compared with
  bool verify_track = false; int32_t* v_corr = nullptr; int32_t* v_nn_idx = nullptr; float* v_sqd = nullptr; float4* v_nn_ref = nullptr; uint32_t* v_counters = nullptr;
  // profiling
  bool prof_on = false, prof_open = false; int prof_depth = 0;      // prof_open / prof_depth: a span is open / scopes nested inside it (they join it)
  std::vector<ProfSpan> spans;
  qn_kernel_stat stats[QN_K_COUNT] = {};
  std::string last_error;

  void set_error(const char* what, hipError_t e, int line);
  void prof_begin(int family, int count = 1); void prof_end(); void prof_collect();
};
