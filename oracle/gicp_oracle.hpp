// ORACLE - TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).  PARITY UNPINNED.
//
// CPU restatement of the Nano-GICP fine registration that
// LoopClosure::icpAlignment drives (fast_lio_sam_qn/src/loop_closure.cpp:110-136):
//   setInputSource / calculateSourceCovariances / setInputTarget / calculateTargetCovariances
//   / align / getFitnessScore / hasConverged / getFinalTransformation.
// The bodies are un-vendored (third_party/nano_gicp is empty); they follow the published
// engcang/nano_gicp == fast_gicp LsqRegistration + PCL Registration algorithms as written
// down in SURVEY.md Appendix A.1.  Dependency-free C++17 + OpenMP.
#pragma once
#include <cstdint>
#include <vector>
#include <array>
#include "oracle_math.hpp"

namespace orc {

// ---- exact k-NN: KD-tree over f32 points, f32 squared distances (dx*dx+dy*dy+dz*dz, no FMA),
// ties broken towards the lowest point index.  Stands in for nanoflann (nano_gicp) and for
// PCL's FLANN tree (getFitnessScore): both are exact L2 trees, so only tie order is ours.
class KdTree {
 public:
  void build(const float* xyz, int n);          // xyz: n x 3 contiguous
  // k nearest (sorted by (d2, index) ascending); returns number found (= min(k, n))
  int knn(const float q[3], int k, int* idx, float* d2) const;
  int size() const { return n_; }
 private:
  struct Node { int left, right, begin, end, axis; float split; };
  int build_rec(int begin, int end);
  const float* pts_ = nullptr; int n_ = 0;
  std::vector<int> order_; std::vector<Node> nodes_;
};

struct GicpParams {
  int k_correspondences = 20;            // NanoGICP default; reference sets 15 (loop_closure.cpp:10, config.yaml:22)
  int max_iterations = 64;               // LsqRegistration default; reference 32 (loop_closure.cpp:11)
  double max_corr_dist = 3.4028234663852886e38;  // float max; reference 52.5 m (loop_closure.cpp:13)
  double transformation_epsilon = 5e-4;  // reference 0.01 (loop_closure.cpp:14)
  double rotation_epsilon = 2e-3;        // no setter called by the reference
  int optimizer = 0;                     // 0 = LevenbergMarquardt (default), 1 = GaussNewton
  int lm_max_iterations = 10;
  double lm_init_lambda_factor = 1e-9;
  int force_iterations = 0;              // bench only: >0 runs exactly this many outer iterations
  int num_threads = 0;                   // 0 = all cores (loop_closure.cpp:9, config.yaml:20)
};

struct IterTrace { double y0, lambda, rho, max_dR, max_dt; int inner; int accepted; };

struct GicpResult {
  double T[16];            // x0 after the loop, f64 row-major
  float Tf[16];            // final_transformation_ = x0.cast<float>(), row-major
  double H[36];            // final_hessian_
  int iterations;          // outer iterations executed
  int converged;
  double fitness;          // pcl getFitnessScore(): mean squared NN distance over ALL source points
  std::vector<IterTrace> trace;
};

class NanoGicpOracle {
 public:
  GicpParams params;
  void setInputSource(const float* xyz, int n);
  void setInputTarget(const float* xyz, int n);
  bool calculateSourceCovariances();
  bool calculateTargetCovariances();
  void align(const double guess[16], GicpResult* out);      // guess row-major 4x4
  double getFitnessScore(const float Tf[16], double max_range) const;
  void transformedSource(const float Tf[16], float* out_xyz) const;  // pcl::transformPointCloud (f32, SSE order)

  // exposed for per-stage parity tests
  double linearize(const double T[16], double H[36], double b[6]);
  double compute_error(const double T[16]) const;
  const std::vector<std::array<double, 9>>& sourceCovs() const { return src_cov_; }
  const std::vector<std::array<double, 9>>& targetCovs() const { return tgt_cov_; }
  const std::vector<int>& correspondences() const { return corr_; }
  const std::vector<float>& sqDistances() const { return sqd_; }
  const KdTree& sourceTree() const { return src_tree_; }
  const KdTree& targetTree() const { return tgt_tree_; }

 private:
  bool calc_cov(const std::vector<float>& pts, const KdTree& tree, std::vector<std::array<double, 9>>& covs);
  void update_correspondences(const double T[16]);
  bool step_lm(double x0[16], double delta[16], IterTrace& tr);
  bool step_gn(double x0[16], double delta[16], IterTrace& tr);
  bool is_converged(const double delta[16], IterTrace* tr) const;
  int threads() const;

  std::vector<float> src_, tgt_;
  KdTree src_tree_, tgt_tree_;
  std::vector<std::array<double, 9>> src_cov_, tgt_cov_;
  std::vector<int> corr_; std::vector<float> sqd_; std::vector<std::array<double, 9>> mahal_;
  double lm_lambda_ = -1.0;
  double final_H_[36];
};

}  // namespace orc
