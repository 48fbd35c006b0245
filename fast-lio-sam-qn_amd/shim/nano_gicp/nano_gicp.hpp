// Drop-in for <nano_gicp/nano_gicp.hpp> (included at fast_lio_sam_qn/include/loop_closure.h:17).
//
// nano_gicp::NanoGICP<PointSource, PointTarget> with exactly the members LoopClosure uses
// (fast_lio_sam_qn/src/loop_closure.cpp:9-16, 120-124, 127, 129, 133; member at
// include/loop_closure.h:75).  Header-only; every member forwards to ONE function of the C-ABI in
// include/qn_engine.h - all work happens on the MI355X behind it.  Link with -lqn_engine.
// Compiled here against the minimal pcl/Eigen stand-ins in tests/standins (the real headers are
// not installed in this image); the same header builds against real PCL >= 1.8 and Eigen >= 3.2:
// it only touches cloud.points / size() / Ptr and Matrix4f's (row, col) accessor.
#pragma once
#include <cfloat>
#include <cstdint>
#include <cstdio>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <Eigen/Core>
#include "qn_engine.h"

namespace nano_gicp {

template <typename PointSource, typename PointTarget>
class NanoGICP {
 public:
  using PointCloudSource = pcl::PointCloud<PointSource>;
  using PointCloudSourcePtr = typename PointCloudSource::Ptr;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTarget = pcl::PointCloud<PointTarget>;
  using PointCloudTargetPtr = typename PointCloudTarget::Ptr;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;

  NanoGICP() { qn_gicp_default_params(&params_); }
  ~NanoGICP() { if (ctx_) qn_ctx_destroy(ctx_); }
  NanoGICP(const NanoGICP&) = delete;
  NanoGICP& operator=(const NanoGICP&) = delete;

  // ---- the eight setters of loop_closure.cpp:9-16
  void setNumThreads(int /*n*/) {}                                            // CPU threads: meaningless on the GPU
  void setCorrespondenceRandomness(int k) { params_.k_correspondences = k; dirty_ = true; }
  void setMaximumIterations(int n) { params_.max_iterations = n; dirty_ = true; }
  void setRANSACIterations(int n) { params_.ransac_iterations = n; dirty_ = true; }
  void setMaxCorrespondenceDistance(double d) { params_.max_corr_dist = d; dirty_ = true; }
  void setTransformationEpsilon(double e) { params_.transformation_epsilon = e; dirty_ = true; }
  void setEuclideanFitnessEpsilon(double e) { params_.euclidean_fitness_epsilon = e; dirty_ = true; }
  void setRANSACOutlierRejectionThreshold(double t) { params_.ransac_outlier_threshold = t; dirty_ = true; }
  void setRotationEpsilon(double e) { params_.rotation_epsilon = e; dirty_ = true; }

  // ---- loop_closure.cpp:120-123
  void setInputSource(const PointCloudSourceConstPtr& cloud) {
    source_ = cloud; has_result_ = false; src_cov_done_ = false;
    if (!prepare(cloud ? cloud->size() : 0, false)) return;
    status_ = cloud->size() ? qn_gicp_set_source(ctx_, &cloud->points[0].x, (uint32_t)cloud->size(), (uint32_t)sizeof(PointSource)) : QN_ERR_EMPTY_CLOUD;
  }
  void setInputTarget(const PointCloudTargetConstPtr& cloud) {
    target_ = cloud; has_result_ = false; tgt_cov_done_ = false;
    if (!prepare(cloud ? cloud->size() : 0, true)) return;
    status_ = cloud->size() ? qn_gicp_set_target(ctx_, &cloud->points[0].x, (uint32_t)cloud->size(), (uint32_t)sizeof(PointTarget)) : QN_ERR_EMPTY_CLOUD;
  }
  bool calculateSourceCovariances() { src_cov_done_ = ctx_ && push() && cov(QN_SOURCE); return src_cov_done_; }
  bool calculateTargetCovariances() { tgt_cov_done_ = ctx_ && push() && cov(QN_TARGET); return tgt_cov_done_; }

  // ---- loop_closure.cpp:124: pcl::Registration::align(output) == align(output, Identity)
  void align(PointCloudSource& output) { align_impl(output, nullptr); }
  void align(PointCloudSource& output, const Eigen::Matrix4f& guess) {
    float g[16];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) g[4 * r + c] = guess(r, c);
    align_impl(output, g);
  }

  // ---- loop_closure.cpp:127, 129, 133.  No exceptions, no error codes: a failed or impossible
  // registration reads as hasConverged() == false and a huge score, as with the reference.
  double getFitnessScore(double max_range = DBL_MAX) {
    if (!has_result_) return DBL_MAX;
    if (max_range >= DBL_MAX) return result_.fitness;
    double s = DBL_MAX;
    return qn_gicp_fitness(ctx_, max_range, &s) == QN_OK ? s : DBL_MAX;
  }
  bool hasConverged() const { return has_result_ && result_.converged != 0; }
  Eigen::Matrix4f getFinalTransformation() const {
    Eigen::Matrix4f T = Eigen::Matrix4f::Identity();
    if (has_result_) for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T(r, c) = result_.T[4 * r + c];
    return T;
  }
  int lastStatus() const { return status_; }     // not part of the reference surface: qn_status_str() of the last call

 private:
  bool prepare(size_t n, bool for_target) {   // (re)create the device context when a cloud outgrows it
    if (n > capacity_ || !ctx_) {
      if (ctx_) { qn_ctx_destroy(ctx_); ctx_ = nullptr; }
      capacity_ = n + n / 2 + 4096;
      if (source_ && source_->size() + 4096 > capacity_) capacity_ = source_->size() + 4096;
      if (target_ && target_->size() + 4096 > capacity_) capacity_ = target_->size() + 4096;
      status_ = qn_ctx_create(0, (uint32_t)capacity_, &ctx_);
      if (status_ != QN_OK) { std::fprintf(stderr, "[nano_gicp shim] %s\n", qn_status_str(status_)); ctx_ = nullptr; capacity_ = 0; return false; }
      dirty_ = true;
      if (!push()) return false;
      // the OTHER cloud (already set, maybe with covariances) lived in the old context: restore it, whichever order the
      // caller uses (LoopClosure sets the source first, loop_closure.cpp:120-123; PCL code usually the target first)
      if (for_target && source_ && source_->size()) {
        int rc = qn_gicp_set_source(ctx_, &source_->points[0].x, (uint32_t)source_->size(), (uint32_t)sizeof(PointSource));
        if (rc == QN_OK && src_cov_done_) rc = qn_gicp_compute_covariances(ctx_, QN_SOURCE);
        if (rc != QN_OK) { status_ = rc; src_cov_done_ = false; std::fprintf(stderr, "[nano_gicp shim] restoring the source after a regrow failed: %s\n", qn_status_str(rc)); return false; }
      }
      if (!for_target && target_ && target_->size()) {
        int rc = qn_gicp_set_target(ctx_, &target_->points[0].x, (uint32_t)target_->size(), (uint32_t)sizeof(PointTarget));
        if (rc == QN_OK && tgt_cov_done_) rc = qn_gicp_compute_covariances(ctx_, QN_TARGET);
        if (rc != QN_OK) { status_ = rc; tgt_cov_done_ = false; std::fprintf(stderr, "[nano_gicp shim] restoring the target after a regrow failed: %s\n", qn_status_str(rc)); return false; }
      }
    }
    return push();
  }
  bool cov(int which) { status_ = qn_gicp_compute_covariances(ctx_, which); return status_ == QN_OK; }
  bool push() { if (dirty_ && ctx_) { status_ = qn_gicp_set_params(ctx_, &params_); dirty_ = status_ != QN_OK; } return !dirty_; }
  void align_impl(PointCloudSource& output, const float* guess) {
    has_result_ = false;
    if (!ctx_ || !source_ || !push()) return;
    status_ = qn_gicp_align(ctx_, guess, &result_);
    if (status_ != QN_OK) return;
    has_result_ = true;
    output = *source_;                                               // keeps intensity etc.; xyz(+1) are overwritten below
    if (output.size()) status_ = qn_gicp_transformed_source(ctx_, &output.points[0].x, (uint32_t)sizeof(PointSource));
  }

  qn_ctx* ctx_ = nullptr;
  size_t capacity_ = 0;
  qn_gicp_params params_;
  qn_gicp_result result_;
  bool dirty_ = true, has_result_ = false, src_cov_done_ = false, tgt_cov_done_ = false;
  int status_ = QN_OK;
  PointCloudSourceConstPtr source_;
  PointCloudTargetConstPtr target_;
};

}  // namespace nano_gicp
