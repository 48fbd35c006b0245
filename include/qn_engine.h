/* qn_engine.h - C-ABI of the MI355X-native loop-closure registration engine.
 *
 * This is the drop-in boundary for the ONE hot path of engcang/FAST-LIO-SAM-QN: the
 * Nano-GICP + Quatro scan matching that FastLioSamQn::loopTimerFunc triggers
 * (fast_lio_sam_qn/src/fast_lio_sam_qn.cpp:203-252 -> LoopClosure::icpAlignment /
 * coarseToFineAlignment, fast_lio_sam_qn/src/loop_closure.cpp:110-159).
 *
 * The reference has no FFI: its boundary is two C++ class templates,
 * nano_gicp::NanoGICP<> and quatro<> (includes at include/loop_closure.h:16-19, members at
 * :75-76).  The header-only C++ shims in fast-lio-sam-qn_amd/shim/ reproduce those classes and
 * call ONLY the functions declared here; each entry point below names the reference call it
 * stands behind.  Plain C: int status returns, caller-owned buffers, no exceptions, no torch
 * or PCL/Eigen types.  One context per host thread (the reference enters the engines from one
 * timer thread at a time); a context is NOT re-entrant.
 *
 * Matrices are 4x4 ROW-major.  Point buffers are `n` points of 3 leading floats (x, y, z) with
 * `stride_bytes` between points (32 for pcl::PointXYZI, 16 for float4, 12 for packed xyz).
 * "_device" variants take HIP device pointers (same layout) and do no host<->device copies.
 */
#ifndef QN_ENGINE_H
#define QN_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qn_ctx qn_ctx;

enum {
  QN_OK = 0,
  QN_ERR_INVALID_ARG = 1,     /* null pointer, bad stride, bad enum                          */
  QN_ERR_EMPTY_CLOUD = 2,     /* a cloud with 0 points (registration yields valid = 0)       */
  QN_ERR_CAPACITY = 3,        /* cloud larger than the context's max_points                  */
  QN_ERR_NOT_READY = 4,       /* align before both clouds and both covariance sets exist     */
  QN_ERR_HIP = 5,             /* a HIP runtime call failed; see qn_last_error()              */
  QN_ERR_NO_DEVICE = 6        /* no gfx950 device: the engine has NO CPU fallback            */
};

enum { QN_SOURCE = 0, QN_TARGET = 1 };
enum { QN_OPT_LM = 0, QN_OPT_GN = 1 };

/* Mirrors what LoopClosure's ctor pushes through the 8 NanoGICP setters
 * (loop_closure.cpp:9-16; struct NanoGICPConfig, include/loop_closure.h:25-36) plus the
 * LsqRegistration knobs the reference leaves at their defaults.          */
typedef struct {
  int32_t k_correspondences;       /* setCorrespondenceRandomness  (loop_closure.cpp:10); default 20  */
  int32_t max_iterations;          /* setMaximumIterations         (loop_closure.cpp:11); default 64  */
  double  max_corr_dist;           /* setMaxCorrespondenceDistance (loop_closure.cpp:13); default FLT_MAX */
  double  transformation_epsilon;  /* setTransformationEpsilon     (loop_closure.cpp:14); default 5e-4 */
  double  rotation_epsilon;        /* LsqRegistration default 2e-3 (no setter called by the reference) */
  int32_t optimizer;               /* QN_OPT_LM (reference default) | QN_OPT_GN                        */
  int32_t lm_max_iterations;       /* 10                                                               */
  double  lm_init_lambda_factor;   /* 1e-9                                                             */
  int32_t force_iterations;        /* bench only: > 0 runs exactly this many outer iterations          */
  int32_t ransac_iterations;       /* setRANSACIterations (loop_closure.cpp:12): stored, unused by the LSQ path */
  double  ransac_outlier_threshold;/* setRANSACOutlierRejectionThreshold (loop_closure.cpp:16): stored, unused  */
  double  euclidean_fitness_epsilon;/* setEuclideanFitnessEpsilon (loop_closure.cpp:15): stored, unused         */
} qn_gicp_params;

/* What icpAlignment reads back (loop_closure.cpp:127-133): getFitnessScore(), hasConverged(),
 * getFinalTransformation() - plus the f64 state and iteration trace used by the parity tests. */
typedef struct {
  float   T[16];          /* final_transformation_ (f32, as the reference returns it)            */
  double  T64[16];        /* the f64 estimate before the cast                                     */
  double  H[36];          /* final_hessian_                                                       */
  double  fitness;        /* pcl getFitnessScore(): mean squared NN distance over all src points  */
  int32_t iterations;     /* outer iterations executed                                            */
  int32_t converged;      /* hasConverged()                                                       */
  int32_t lm_failed;      /* "lm not converged!!" path taken                                      */
  int32_t reserved;
} qn_gicp_result;

/* one row per outer iteration, for trajectory parity */
typedef struct { double y0, lambda, rho, max_dR, max_dt; int32_t inner, accepted; } qn_iter_trace;

typedef struct {           /* device-side accounting of one kernel family (bench roofline leg) */
  double  total_ms;
  int64_t launches;
} qn_kernel_stat;

enum {                     /* kernel families for qn_prof_get */
  QN_K_GRID_BUILD = 0, QN_K_KNN_COV = 1, QN_K_NN_SEARCH = 2, QN_K_NN_FALLBACK = 3,
  QN_K_ACCUMULATE = 4, QN_K_SOLVE = 5, QN_K_FITNESS = 6, QN_K_TRANSFORM = 7,
  QN_K_FPFH_NORMALS = 8, QN_K_FPFH_SPFH = 9, QN_K_FPFH_FPFH = 10, QN_K_FEAT_MATCH = 11,
  QN_K_GN_TICK_FUSED = 12,   /* fused Gauss-Newton tick: tracking NN + leftovers + accumulation in one kernel       */
  QN_K_KNN_SELECT = 13,      /* k-NN selection kernel alone (QN_K_KNN_COV then holds its list tail + covariances)  */
  QN_K_FAR = 15,             /* refresh of far queries' candidate lists (k_far)                                     */
  QN_K_MATCH_TAIL = 14,      /* Matcher tail on the device: means, cross-check + gate, tuple test, hand-over        */
  QN_K_ALIGN_PERSIST = 16,   /* the persistent align kernel: every tracked tick + closing pass of one align in ONE launch */
  QN_K_COUNT = 17
};

/* ---- lifetime ------------------------------------------------------------------------- */
int  qn_ctx_create(int device, uint32_t max_points, qn_ctx** out);
void qn_ctx_destroy(qn_ctx* ctx);
const char* qn_status_str(int status);
const char* qn_last_error(const qn_ctx* ctx);
/* The context's primary hipStream_t.  NOT an ordering guarantee for inputs: the engine also works on a private second stream (the target
 * cloud of a pair is prepared there while the source's k-NN runs), so device buffers handed to a *_device entry point must be COMPLETE
 * before the call (synchronise the producer, or make it wait on an event of yours); outputs are complete when the call returns.     */
void* qn_ctx_stream(qn_ctx* ctx);
int  qn_ctx_synchronize(qn_ctx* ctx);

/* ---- Nano-GICP ------------------------------------------------------------------------ */
void qn_gicp_default_params(qn_gicp_params* p);                       /* NanoGICP()/LsqRegistration() ctor defaults */
int  qn_gicp_set_params(qn_ctx*, const qn_gicp_params*);              /* loop_closure.cpp:9-16  */
int  qn_gicp_get_params(const qn_ctx*, qn_gicp_params* out);          /* what the setters above last stored (the getters of pcl::Registration / NanoGICP) */
/* setInputSource / setInputTarget do not wait for the GPU: upload, packing and the grid build (its numbers are derived from the bounding box ON the
 * device) are enqueued and the call returns.  Consequences at this boundary: (1) a cloud with non-finite coordinates is refused by the first call that
 * synchronises (align, fitness, a read-back) with QN_ERR_INVALID_ARG, not by the setter; (2) pageable host buffers are consumed when the setter returns
 * (hipMemcpyAsync stages them before returning); PAGE-LOCKED host buffers and the device buffers of the *_device variants are read asynchronously and
 * must stay unchanged until the next synchronising call of this context (qn_gicp_align, qn_ctx_synchronize, ...).                                  */
int  qn_gicp_set_source(qn_ctx*, const float* xyz, uint32_t n, uint32_t stride_bytes);          /* setInputSource, loop_closure.cpp:120 */
int  qn_gicp_set_target(qn_ctx*, const float* xyz, uint32_t n, uint32_t stride_bytes);          /* setInputTarget, loop_closure.cpp:122 */
int  qn_gicp_set_source_device(qn_ctx*, const float* d_xyz, uint32_t n, uint32_t stride_bytes);
int  qn_gicp_set_target_device(qn_ctx*, const float* d_xyz, uint32_t n, uint32_t stride_bytes);
int  qn_gicp_compute_covariances(qn_ctx*, int which);                 /* calculateSource/TargetCovariances, loop_closure.cpp:121,123 */
int  qn_gicp_align(qn_ctx*, const float guess[16], qn_gicp_result* out);  /* align(), loop_closure.cpp:124 (guess NULL = identity); also fills fitness */
int  qn_gicp_fitness(qn_ctx*, double max_range, double* score);       /* getFitnessScore(), loop_closure.cpp:127 */
int  qn_gicp_transformed_source(qn_ctx*, float* xyz_out, uint32_t stride_bytes);  /* the `aligned_` cloud align() fills, loop_closure.cpp:124 */
int  qn_gicp_get_trace(qn_ctx*, qn_iter_trace* out, uint32_t cap, uint32_t* n);
/* the same for lane `lane` of the latest qn_gicp_align_batch run on this context (lane l of a run carries the l-th pair of that run) */
int  qn_gicp_get_lane_trace(qn_ctx*, uint32_t lane, qn_iter_trace* out, uint32_t cap, uint32_t* n);

/* LoopClosure::icpAlignment in one call (loop_closure.cpp:110-136): set x2, cov x2, align, score,
 * accept test `converged && score < score_thr` (loop_closure.cpp:129).  *valid receives is_valid_. */
int  qn_icp_alignment(qn_ctx*, const float* src, uint32_t ns, const float* dst, uint32_t nt,
                      uint32_t stride_bytes, double score_thr, qn_gicp_result* out, int* valid);
int  qn_icp_alignment_device(qn_ctx*, const float* d_src, uint32_t ns, const float* d_dst, uint32_t nt,
                             uint32_t stride_bytes, double score_thr, qn_gicp_result* out, int* valid);
/* The candidates of ONE loop-closure query share their source cloud: after a qn_icp_alignment[_device] call the context still holds the
   source's grid and covariances; this registers another target against it (the reference rebuilds the source per call,
   loop_closure.cpp:116-123, because it only ever tries one candidate).  QN_ERR_NOT_READY without a prepared source. */
int  qn_icp_alignment_same_source(qn_ctx*, const float* dst, uint32_t nt, uint32_t stride_bytes, int dst_on_device, double score_thr, qn_gicp_result* out, int* valid);

/* ---- batch of independent candidate pairs (BASELINE config "batch of 64 candidate keyframe pairs") ----
 * Every candidate pair of a loop-closure query is an independent icpAlignment (loop_closure.cpp:116-123 rebuilds
 * everything per call), so a batch is spread over several contexts = several hipStreams of one GPU: worker i drives
 * ctxs[i] and pulls the next unprocessed pair.  Parameters are those already set on each context.  status[i] receives
 * the per-pair status code.  Returns QN_OK when every pair ran (individual pairs may still be invalid).          */
typedef struct {
  const float* src; uint32_t ns;
  const float* dst; uint32_t nt;
  uint32_t stride_bytes;
  int32_t  on_device;              /* 1: src/dst are HIP device pointers */
} qn_pair_desc;
int  qn_icp_alignment_batch(qn_ctx* const* ctxs, uint32_t n_ctx, const qn_pair_desc* pairs, uint32_t n_pairs, double score_thr,
                            qn_gicp_result* results, int* valid, int* status);
/* The same batch on ONE context with the PAIR AS A GRID DIMENSION (SURVEY.md 8b qn_gicp_align_batch; 7.1 step 8): the pairs are registered `lanes` at a
 * time (default 8; qn_debug_set(ctx, "batch_lanes", B)) in lockstep - every kernel of the chain is launched once for all of them, blockIdx.y selecting the
 * pair's entry of a device-resident argument table - on the context's one stream.  Each pair is an independent icpAlignment (loop_closure.cpp:110-136) with the
 * context's parameters; pairs of ONE call that name the same source buffer (pointer, size, stride) share one preparation of it - grid and covariances - the way the
 * candidates of one loop-closure query share the query cloud (qn_debug_set(ctx, "batch_share_source", 0): every pair rebuilds its source like loop_closure.cpp:120-121).
 * Records are bit-identical to qn_icp_alignment_batch's one-pair-per-stream path.  qn_icp_alignment_batch itself uses this per context.
 * MEMORY: a context that registers batches owns `batch_lanes` - 1 sub-contexts, each with a full max_points slab (~1.2 KB per point of max_points: 119 MB at 100k), created on the
 * first batch call: batch_lanes x in_flight x slab in total (8 x 3 x 119 MB = 2.9 GB at the bench's setting; batch_lanes = 64 at 100k is 7.6 GB per context).  If the lanes cannot be
 * allocated, the ones created so far are freed again and both entry points register the pairs one at a time on the context itself (same records).                        */
int  qn_gicp_align_batch(qn_ctx*, const qn_pair_desc* pairs, uint32_t n_pairs, double score_thr, qn_gicp_result* results, int* valid, int* status);

/* ---- candidate pairs sharded over the GPUs of one node (BASELINE "batch of 64 candidate keyframe pairs sharded
 * across 8 MI355X, RCCL gather of best loop").  The reference registers ONE candidate per timer tick
 * (fast_lio_sam_qn.cpp:213-219 -> loop_closure.cpp:168-205); the generalisation keeps every registration independent: pair i runs on
 * GPU i mod N on one of `in_flight` contexts, no data-path collective, then ONE ncclAllGather (RCCL over xGMI) of the fixed-size
 * records below and the host picks the valid record with the smallest score.  Single process, all GPUs (ncclCommInitAll), like the
 * reference's single process.  qn_multi_init fails with QN_ERR_NO_DEVICE when fewer than n_gpus devices are visible.              */
typedef struct qn_multi qn_multi;
typedef struct {            /* 96 bytes */
  int32_t pair_id;          /* index into `pairs`; -1 = padding slot of the gathered table                    */
  int32_t status;           /* qn status of this pair's icpAlignment                                            */
  int32_t valid;            /* is_valid_ (converged && score < score_thr, loop_closure.cpp:129)                 */
  int32_t converged;        /* hasConverged()                                                                   */
  int32_t iterations;
  int32_t reserved;
  double  fitness;          /* getFitnessScore()                                                                */
  float   T[16];            /* getFinalTransformation(), row-major                                              */
} qn_pair_record;
int  qn_multi_init(int n_gpus, const int* device_ids /* NULL: 0 .. n_gpus-1 */, uint32_t max_points, int in_flight, qn_multi** out);
void qn_multi_destroy(qn_multi*);
const char* qn_multi_last_error(const qn_multi*);    /* NULL argument: why the last qn_multi_init on this thread failed */
int  qn_multi_gpu_count(const qn_multi*);
/* ranks of the RCCL communicator as RCCL itself reports them (ncclCommCount on every GPU's communicator; the smallest answer, -1 if a query fails): a caller that
 * asked for N GPUs checks this equals N before it trusts a multi-GPU figure.  qn_multi_verify_gather: after a qn_multi_align_best, QN_OK iff EVERY GPU's receive buffer
 * holds the same gathered record table as GPU 0's (the all-gather delivered every rank's records to every rank, not just to the one the host reads)                     */
int  qn_multi_rccl_ranks(qn_multi*);
int  qn_multi_verify_gather(qn_multi*);
int  qn_multi_set_params(qn_multi*, const qn_gicp_params*);            /* loop_closure.cpp:9-16, on every context */
int  qn_multi_debug_set(qn_multi*, const char* key, double value);     /* qn_debug_set on every context (e.g. "batch_lanes": pairs per kernel launch of each context) */
/* host wall clock of the latest qn_multi_align_best: per GPU from the call's start to its last pair's end [n_gpus], and the gather step */
int  qn_multi_get_timing(const qn_multi*, double* per_gpu_ms, double* gather_ms);
/* pairs[i].src/dst: host buffers, or (on_device) buffers resident on GPU device_ids[i mod n_gpus].  records (optional, n_pairs
 * entries) receives every pair's record as rank 0 holds them after the gather; *best / *best_found the winning loop.             */
int  qn_multi_align_best(qn_multi*, const qn_pair_desc* pairs, uint32_t n_pairs, double score_thr,
                         qn_pair_record* records, qn_pair_record* best, int* best_found);

/* ---- Quatro coarse registration ---------------------------------------------------------- */
/* The 10 constructor arguments of quatro<PointType>, in the order LoopClosure passes them
 * (loop_closure.cpp:18-27; struct QuatroConfig, include/loop_closure.h:38-50), plus the seed of the
 * tuple test (the reference seeds rand() from wall-clock and does not reproduce itself).       */
typedef struct {
  double  fpfh_normal_radius;      /* loop_closure.cpp:18 */
  double  fpfh_radius;             /* :19 */
  double  noise_bound;             /* :20 */
  double  rot_gnc_factor;          /* :21 */
  double  rot_cost_diff_thr;       /* :22 */
  int32_t rot_max_iter;            /* :23 */
  int32_t estimate_scale;          /* :24 estimat_scale_ (include/loop_closure.h:44; config.yaml ships false): 1 = TEASER++'s TLS scale solver over the TIM norm ratios runs in front of the
                                      consistency graph; rotation on dst TIMs / scale with the bound x 2 / scale, translation on dst - scale R src.  T stays [R | t] (upstream's quatro::align
                                      returns rotation and translation only); the scale itself: qn_quatro_get_scale */
  int32_t use_optimized_matching;  /* :25  1: Matcher::optimizedMatching (gate + cap), 0: Matcher::advancedMatching */
  double  distance_threshold;      /* :26 */
  int32_t max_num_corres;          /* :27 */
  uint32_t rng_seed;
  double  tuple_scale;             /* Matcher::calculateCorrespondences argument, 0.95 upstream */
} qn_quatro_params;

void qn_quatro_default_params(qn_quatro_params* p);                  /* the reference's effective values */
int  qn_quatro_set_params(qn_ctx*, const qn_quatro_params*);         /* quatro<PointType> ctor, loop_closure.cpp:18-27 */
/* quatro<PointType>::align(src, dst, is_converged), loop_closure.cpp:144: T = 4x4 f64 row-major, *valid = is_converged */
int  qn_quatro_align(qn_ctx*, const float* src, uint32_t ns, const float* dst, uint32_t nt, uint32_t stride_bytes, double T[16], int* valid);
int  qn_quatro_align_device(qn_ctx*, const float* d_src, uint32_t ns, const float* d_dst, uint32_t nt, uint32_t stride_bytes, double T[16], int* valid);
/* LoopClosure::coarseToFineAlignment, loop_closure.cpp:138-159: Quatro, transformPcd, icpAlignment, T_gicp * T_quatro */
int  qn_coarse_to_fine_alignment(qn_ctx*, const float* src, uint32_t ns, const float* dst, uint32_t nt, uint32_t stride_bytes, double score_thr,
                                 qn_gicp_result* gicp_out, double T_total[16], double T_quatro[16], int* valid);
/* same with both clouds resident on the device (the output of qn_kf_assemble = setSrcAndDstCloud, loop_closure.cpp:177-192):
 * only the selected correspondences (<= ~6 KB) and the result record cross PCIe                                            */
int  qn_coarse_to_fine_alignment_device(qn_ctx*, const float* d_src, uint32_t ns, const float* d_dst, uint32_t nt, uint32_t stride_bytes, double score_thr,
                                        qn_gicp_result* gicp_out, double T_total[16], double T_quatro[16], int* valid);
/* The reference's DEFAULT per-candidate path for MANY candidates (enable_quatro_ = true, include/loop_closure.h:54 + config.yaml:31; dispatch loop_closure.cpp:188-192 ->
 * coarseToFineAlignment :138-159): n_pairs independent coarse-to-fine registrations over n_ctx contexts (= streams, one pooled host worker each).  Each context takes runs of
 * `batch_lanes` pairs: the Quatro device stages of a run are enqueued back to back (one lane's buffers per pair, ONE synchronisation per run), the host solver runs per pair,
 * transformPcd stays on the device, and the run's accepted pairs go through the GICP lanes (qn_gicp_align_batch's machinery).  Parameters: each context's own
 * (qn_gicp_set_params / qn_quatro_set_params).  results[i] = the fine stage's record, T_total[16 i ..] = T_gicp * T_quatro (row-major f64), T_quatro (optional) = the coarse
 * estimate, valid[i] = Quatro converged && GICP converged && score < score_thr, status[i] = the pair's own status.  Records equal qn_coarse_to_fine_alignment[_device] of
 * the same pair bit for bit.  Pairs of one run that name the same source buffer (the candidates of ONE loop-closure query) share the source's Quatro preparation - grid, normals,
 * SPFH, FPFH are made once per run of lanes and borrowed read-only by the other lanes (qn_debug_set(ctx, "batch_share_source", 0): every pair prepares its own, same records).
 * Memory: every lane allocates its own Quatro buffers on first use (~0.7 KB per point of max_points per lane).                                                          */
int  qn_coarse_to_fine_align_batch(qn_ctx* const* ctxs, uint32_t n_ctx, const qn_pair_desc* pairs, uint32_t n_pairs, double score_thr,
                                   qn_gicp_result* results, double* T_total, double* T_quatro, int* valid, int* status);
/* enable_quatro_ (include/loop_closure.h:54): non-NULL = every pair of qn_multi_align_best is a coarseToFineAlignment (qn_coarse_to_fine_align_batch per GPU) with these
 * Quatro parameters, and the records carry T_gicp * T_quatro (cast to f32: the record layout is fixed); NULL = Nano-GICP only (icpAlignment), the default.             */
int  qn_multi_set_quatro_params(qn_multi*, const qn_quatro_params* p);
/* The two stages upstream Quatro exposes on its own:
 *  qn_fpfh            = FPFH descriptors of one cloud (pcl::FPFHEstimationOMP with the context's radii): n x 33 f32, caller order
 *  qn_match_optimized = teaser::Matcher::optimizedMatching(thr_dist, num_max_corres, tuple_scale) on two clouds + their
 *                       descriptors: mutual 33-D NN (GPU), distance gate, tuple test, cap; pairs = (src idx, dst idx)            */
int  qn_fpfh(qn_ctx*, const float* xyz, uint32_t n, uint32_t stride_bytes, float* fpfh33_out);
int  qn_match_optimized(qn_ctx*, const float* src, uint32_t ns, const float* dst, uint32_t nt, uint32_t stride_bytes,
                        const float* src_fpfh33, const float* dst_fpfh33, float thr_dist, int num_max_corres, float tuple_scale,
                        int32_t* pairs_out, uint32_t cap, uint32_t* n_out);

/* stages, for the parity tests: descriptors of the last qn_quatro_align (original point order; n x 3, n x 33, n x 33),
 * the align with its intermediate products, and the host solver alone (Matcher + TEASER++/Quatro solve)            */
int  qn_quatro_get_features(qn_ctx*, int which, float* normals3, float* spfh33, float* fpfh33);
int  qn_quatro_align_debug(qn_ctx*, const float* src, uint32_t ns, const float* dst, uint32_t nt, uint32_t stride_bytes, double T[16], int* valid,
                           int32_t* mutual_pairs, uint32_t* n_mutual, int32_t* corres_pairs, uint32_t* n_corres, uint32_t cap,
                           int32_t* clique, uint32_t* n_clique, int32_t* rot_iterations);
int  qn_quatro_solve(const float* src, const float* dst, uint32_t stride_bytes, const int32_t* corres_pairs, uint32_t n_corres,
                     const qn_quatro_params* p, double T[16], int* valid, int32_t* clique, uint32_t* n_clique);
/* the scale TEASER++'s solver estimated (1 unless estimate_scale): of the context's latest qn_quatro_align[_device/_debug] / coarse-to-fine call, and of the host solver alone */
int  qn_quatro_get_scale(qn_ctx*, double* scale);
int  qn_quatro_solve_scaled(const float* src, const float* dst, uint32_t stride_bytes, const int32_t* corres_pairs, uint32_t n_corres,
                            const qn_quatro_params* p, double T[16], int* valid, int32_t* clique, uint32_t* n_clique, double* scale);

/* ---- feeder of the path, kept on the device -------------------------------
 * Keyframe clouds (PosePcd::pcd_, sensor frame, include/pose_pcd.hpp:7-19) are uploaded once and stay resident;
 * qn_kf_assemble = the inner loops of LoopClosure::setSrcAndDstCloud (loop_closure.cpp:70-107): transformPcd of each
 * listed keyframe with its pose, concatenation, voxelizePcd (pcl::VoxelGrid, include/utilities.hpp:38-51), entirely
 * on the GPU; the returned device pointer (float4, stride 16) feeds qn_icp_alignment_device / the batch API.     */
typedef struct qn_kf_store qn_kf_store;
int  qn_kf_store_create(int device, qn_kf_store** out);
void qn_kf_store_destroy(qn_kf_store*);
const char* qn_kf_last_error(const qn_kf_store*);
int  qn_kf_add(qn_kf_store*, const float* xyz, uint32_t n, uint32_t stride_bytes, int32_t* id_out);
int  qn_kf_assemble(qn_kf_store*, const int32_t* ids, const double* poses16, uint32_t count, double leaf, int slot,
                    const float** d_xyz_out, uint32_t* n_out);
int  qn_kf_download(qn_kf_store*, int slot, float* xyz_out /* n x 3 packed */);
/* LoopClosure::fetchClosestKeyframeIdx (loop_closure.cpp:34-56) generalised to the max_k nearest admissible keyframes,
 * ascending distance; out[0] is the reference's single choice.  Host code (O(#keyframes)).                        */
int  qn_loop_candidates(const double* pos_xyz, const double* stamps, uint32_t n, uint32_t query, double radius, double tdiff,
                        uint32_t max_k, int32_t* out, uint32_t* n_out);

/* ---- per-stage read-backs used by the parity tests (not needed by the shims) ------------ */
int  qn_gicp_get_covariances(qn_ctx*, int which, double* cov9_out);   /* n x 9 f64, original point order */
int  qn_gicp_knn(qn_ctx*, int which, int k, int32_t* idx_out, float* d2_out);   /* self k-NN of a cloud, n x k */
int  qn_gicp_linearize(qn_ctx*, const double T[16], double H[36], double b[6], double* err,
                       int32_t* corr_out, float* sqd_out);            /* one update_correspondences + linearize */
int  qn_gicp_compute_error(qn_ctx*, const double T[16], double* err); /* cached correspondences (LM trial)   */

/* ---- profiling hooks (bench.py roofline leg) -------------------------------------------- */
int  qn_prof_enable(qn_ctx*, int on);                /* records a hipEvent pair around every kernel family launch */
int  qn_prof_reset(qn_ctx*);
int  qn_prof_get(qn_ctx*, int kernel_family, qn_kernel_stat* out);
/* developer knobs (cell size, margins, debug counters); not part of the reference surface */
int  qn_debug_set(qn_ctx*, const char* key, double value);
int  qn_debug_get_counters(qn_ctx*, uint32_t out[16]);
int  qn_debug_selftest(qn_ctx*, uint32_t n_waves, uint32_t seed, uint32_t* mismatches);      /* developer / tests: the device's wave-level search primitives against plain restatements (csrc/qn_selftest.cuh) */
int  qn_debug_get(qn_ctx*, const char* key, double* value);   /* "verify_mismatches" / "verify_passes" / "verify_first" after qn_debug_set("verify_track", 1); "feat_survivors" / "feat_fallbacks" (matrix-core feature matching) */
int  qn_debug_get_grid(qn_ctx*, int which, double out[8]);
int  qn_debug_get_partials(qn_ctx*, double* out, uint32_t* rows_per_buffer, double* state);   /* developer: both partial-row buffers and both state buffers after an align */
int  qn_debug_get_list_probe(qn_ctx*, unsigned long long* out /* [4 * 16384]: per wave of the last unseeded list pass: slowest entry << 32 | entries, busy time (100 MHz ticks), and of that slowest entry rounds << 48 | segments << 24 | candidates, first radius | neighbour distance as f32 bits (knob list_probe) */);
int  qn_debug_get_persist_clk(qn_ctx*, unsigned long long* out /* 64 x 16 + 16 wall-clock stamps of the latest persistent align */);   /* after qn_debug_set("persist_probe", 1) */
int  qn_debug_get_clk(qn_ctx*, unsigned long long* out /* 256 x 8 + 1024 x 12 device-clock stamps / counters */, uint32_t* n);   /* after qn_debug_set("clk_probe", 1) */

#ifdef __cplusplus
}
#endif
#endif /* QN_ENGINE_H */
