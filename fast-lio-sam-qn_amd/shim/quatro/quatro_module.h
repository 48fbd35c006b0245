// Drop-in for <quatro/quatro_module.h> (included at fast_lio_sam_qn/include/loop_closure.h:19).
//
// template <typename PointType> class quatro with exactly what LoopClosure uses: the 10-argument
// constructor in the reference's argument order (fast_lio_sam_qn/src/loop_closure.cpp:18-27; held by
// std::shared_ptr at include/loop_closure.h:76) and
//   Eigen::Matrix4d align(const pcl::PointCloud<PointType>& src, const pcl::PointCloud<PointType>& dst, bool& is_converged)
// (loop_closure.cpp:144).  Header-only over the C-ABI of include/qn_engine.h; link with -lqn_engine.
// The upstream class also exposes the matcher's optimizedMatching(thr_dist, num_max_corres, tuple_scale);
// here it is a public method with the same arguments that stores them for the next align().
#pragma once
#include <cstdint>
#include <cstdio>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <Eigen/Core>
#include "qn_engine.h"

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
