// Drop-in for <quatro/quatro_module.h> (included at fast_lio_sam_qn/include/loop_closure.h:19).
//
// template <typename PointType> class quatro with exactly what LoopClosure uses: the 10-argument
// constructor in the reference's argument order (fast_lio_sam_qn/src/loop_closure.cpp:18-27; held by
// std::shared_ptr at include/loop_closure.h:76) and
//   Eigen::Matrix4d align(const pcl::PointCloud<PointType>& src, const pcl::PointCloud<PointType>& dst, bool& is_converged)
// (loop_closure.cpp:144).  Header-only over the C-ABI of include/qn_engine.h; link with -lqn_engine.
// The upstream package also has the matcher on its own: teaser::Matcher with
//   void optimizedMatching(float thr_dist, int num_max_corres, float tuple_scale)      (result in corres_)
// on two clouds and their 33-D FPFH sets; `quatro_matcher` below keeps that signature over qn_fpfh / qn_match_optimized.
// quatro<>::optimizedMatching(...) with the same arguments stores them for the next align().
#pragma once
#include <cstdint>
#include <cstdio>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <Eigen/Core>
#include "qn_engine.h"

#include <utility>
#include <vector>

// teaser::Matcher as Quatro uses it: holds the two clouds and their descriptors, optimizedMatching fills corres_.
template <typename PointType>
class quatro_matcher {
 public:
  typedef std::vector<std::vector<float>> Feature;                       // upstream: vector of 33-float descriptors
  explicit quatro_matcher(qn_ctx* ctx) : ctx_(ctx) {}
  // FPFH of a cloud with the context's radii (upstream FPFHEstimation::computeFPFHFeatures); rows of NaN where PCL yields none
  int computeFPFH(const pcl::PointCloud<PointType>& cloud, std::vector<float>& fpfh33) {
    fpfh33.assign(cloud.size() * 33, 0.f);
    return cloud.size() == 0 ? QN_ERR_EMPTY_CLOUD : qn_fpfh(ctx_, &cloud.points[0].x, (uint32_t)cloud.size(), (uint32_t)sizeof(PointType), fpfh33.data());
  }
  void setInput(const pcl::PointCloud<PointType>& src, const pcl::PointCloud<PointType>& dst, const std::vector<float>& src_fpfh33, const std::vector<float>& dst_fpfh33) {
    src_ = &src; dst_ = &dst; fs_ = &src_fpfh33; ft_ = &dst_fpfh33;
  }
  void optimizedMatching(float thr_dist, int num_max_corres, float tuple_scale) {
    corres_.clear();
    if (!src_ || !dst_ || src_->size() == 0 || dst_->size() == 0) return;
    std::vector<int32_t> pairs(2 * (size_t)num_max_corres); uint32_t n = 0;
    status_ = qn_match_optimized(ctx_, &src_->points[0].x, (uint32_t)src_->size(), &dst_->points[0].x, (uint32_t)dst_->size(), (uint32_t)sizeof(PointType),
                                 fs_->data(), ft_->data(), thr_dist, num_max_corres, tuple_scale, pairs.data(), (uint32_t)num_max_corres, &n);
    if (status_ != QN_OK) return;
    for (uint32_t e = 0; e < n && e < (uint32_t)num_max_corres; e++) corres_.emplace_back(pairs[2 * e], pairs[2 * e + 1]);
  }
  std::vector<std::pair<int, int>> corres_;
  int lastStatus() const { return status_; }
 private:
  qn_ctx* ctx_;
  const pcl::PointCloud<PointType>* src_ = nullptr; const pcl::PointCloud<PointType>* dst_ = nullptr;
  const std::vector<float>* fs_ = nullptr; const std::vector<float>* ft_ = nullptr;
  int status_ = QN_OK;
};

template <typename PointType>
class quatro {
 public:
  quatro(const double& fpfh_normal_radi, const double& fpfh_radi, const double noise_bound, const double& rot_gnc_fact,
         const double& rot_cost_thr, const int& rot_max_iter, const bool& estimat_scale,
         const bool& use_optimized_matching = true, const double& distance_threshold = 30.0, const int& max_correspondences = 200) {
    qn_quatro_default_params(&p_);
    p_.fpfh_normal_radius = fpfh_normal_radi; p_.fpfh_radius = fpfh_radi; p_.noise_bound = noise_bound;
    p_.rot_gnc_factor = rot_gnc_fact; p_.rot_cost_diff_thr = rot_cost_thr; p_.rot_max_iter = rot_max_iter;
    p_.estimate_scale = estimat_scale ? 1 : 0; p_.use_optimized_matching = use_optimized_matching ? 1 : 0;
    p_.distance_threshold = distance_threshold; p_.max_num_corres = max_correspondences;
  }
  ~quatro() { if (ctx_) qn_ctx_destroy(ctx_); }
  quatro(const quatro&) = delete;
  quatro& operator=(const quatro&) = delete;

  void setSeed(uint32_t seed) { p_.rng_seed = seed; }        // the reference seeds the tuple test from wall-clock; here it is explicit
  void optimizedMatching(float thr_dist, int num_max_corres, float tuple_scale) {
    p_.distance_threshold = thr_dist; p_.max_num_corres = num_max_corres; p_.tuple_scale = tuple_scale; p_.use_optimized_matching = 1;
  }

  Eigen::Matrix4d align(const pcl::PointCloud<PointType>& src, const pcl::PointCloud<PointType>& dst, bool& is_converged) {
    Eigen::Matrix4d out = Eigen::Matrix4d::Identity();
    is_converged = false;
    const size_t need = src.size() > dst.size() ? src.size() : dst.size();
    if (need == 0) return out;
    if (!ctx_ || need > capacity_) {
      if (ctx_) { qn_ctx_destroy(ctx_); ctx_ = nullptr; }
      capacity_ = need + need / 2 + 4096;
      status_ = qn_ctx_create(0, (uint32_t)capacity_, &ctx_);
      if (status_ != QN_OK) { std::fprintf(stderr, "[quatro shim] %s\n", qn_status_str(status_)); ctx_ = nullptr; capacity_ = 0; return out; }
    }
    if ((status_ = qn_quatro_set_params(ctx_, &p_)) != QN_OK) return out;
    if (src.size() == 0 || dst.size() == 0) return out;
    double T[16]; int valid = 0;
    status_ = qn_quatro_align(ctx_, &src.points[0].x, (uint32_t)src.size(), &dst.points[0].x, (uint32_t)dst.size(), (uint32_t)sizeof(PointType), T, &valid);
    if (status_ != QN_OK) return out;
    is_converged = valid != 0;
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) out(r, c) = T[4 * r + c];
    return out;
  }
  int lastStatus() const { return status_; }

 private:
  qn_quatro_params p_;
  qn_ctx* ctx_ = nullptr;
  size_t capacity_ = 0;
  int status_ = QN_OK;
};
