// LoopClosure's constructor and coarseToFineAlignment (fast_lio_sam_qn/src/loop_closure.cpp:3-30, 138-159)
// written against the two drop-in shims, in the reference's own shape.
// usage: shim_coarse_to_fine src.bin dst.bin [reps] -> prints valid converged score T(16) (reps > 0: coarseToFineAlignment is repeated and a second line
// "BENCH reps median_ms min_ms" reports its wall time - what an unmodified LoopClosure would see: second context for Quatro, CPU transformPcd, target uploaded twice)
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <vector>
#include <cstdio>
#include <limits>
#include <memory>
#include <nano_gicp/point_type_nano_gicp.hpp>
#include <nano_gicp/nano_gicp.hpp>
#include <quatro/quatro_module.h>

using PointType = pcl::PointXYZI;

struct RegistrationOutput {
  bool is_valid_ = false, is_converged_ = false;
  double score_ = std::numeric_limits<double>::max();
  Eigen::Matrix4d pose_between_eig_ = Eigen::Matrix4d::Identity();
};

static pcl::PointCloud<PointType> load(const char* path) {
  pcl::PointCloud<PointType> c; FILE* f = std::fopen(path, "rb"); float v[3];
  while (f && std::fread(v, sizeof(float), 3, f) == 3) { PointType p; p.x = v[0]; p.y = v[1]; p.z = v[2]; c.push_back(p); }
  if (f) std::fclose(f);
  return c;
}

// utilities.hpp:164-175 (pcl::transformPointCloud with a Matrix4d)
static pcl::PointCloud<PointType> transformPcd(const pcl::PointCloud<PointType>& in, const Eigen::Matrix4d& T) {
  pcl::PointCloud<PointType> out = in;
  for (size_t i = 0; i < in.size(); i++) {
    const double x = in[i].x, y = in[i].y, z = in[i].z;
    out[i].x = (float)(((T(0, 0) * x + T(0, 1) * y) + T(0, 2) * z) + T(0, 3));
    out[i].y = (float)(((T(1, 0) * x + T(1, 1) * y) + T(1, 2) * z) + T(1, 3));
    out[i].z = (float)(((T(2, 0) * x + T(2, 1) * y) + T(2, 2) * z) + T(2, 3));
  }
  return out;
}

struct LoopClosureLike {
  nano_gicp::NanoGICP<PointType, PointType> nano_gicp_;
  std::shared_ptr<quatro<PointType>> quatro_handler_ = nullptr;
  pcl::PointCloud<PointType> coarse_aligned_, aligned_;
  LoopClosureLike() {
    nano_gicp_.setNumThreads(0); nano_gicp_.setCorrespondenceRandomness(15); nano_gicp_.setMaximumIterations(32);
    nano_gicp_.setRANSACIterations(5); nano_gicp_.setMaxCorrespondenceDistance(52.5); nano_gicp_.setTransformationEpsilon(0.01);
    nano_gicp_.setEuclideanFitnessEpsilon(0.01); nano_gicp_.setRANSACOutlierRejectionThreshold(1.0);
    quatro_handler_ = std::make_shared<quatro<PointType>>(0.9, 1.5, 0.3, 1.4, 0.0001, 50, false, true, 35.0, 200);   // loop_closure.cpp:18-27, SURVEY App. C values
  }
  RegistrationOutput icpAlignment(const pcl::PointCloud<PointType>& src, const pcl::PointCloud<PointType>& dst) {
    RegistrationOutput reg_output;
    aligned_.clear();
    pcl::PointCloud<PointType>::Ptr src_cloud(new pcl::PointCloud<PointType>()), dst_cloud(new pcl::PointCloud<PointType>());
    *src_cloud = src; *dst_cloud = dst;
    nano_gicp_.setInputSource(src_cloud); nano_gicp_.calculateSourceCovariances();
    nano_gicp_.setInputTarget(dst_cloud); nano_gicp_.calculateTargetCovariances();
    nano_gicp_.align(aligned_);
    reg_output.score_ = nano_gicp_.getFitnessScore();
    if (nano_gicp_.hasConverged() && reg_output.score_ < 1.5) {
      reg_output.is_valid_ = true; reg_output.is_converged_ = true;
      reg_output.pose_between_eig_ = nano_gicp_.getFinalTransformation().cast<double>();
    }
    return reg_output;
  }
  RegistrationOutput coarseToFineAlignment(const pcl::PointCloud<PointType>& src, const pcl::PointCloud<PointType>& dst) {
    RegistrationOutput reg_output;
    coarse_aligned_.clear();
    reg_output.pose_between_eig_ = (quatro_handler_->align(src, dst, reg_output.is_converged_));
    if (!reg_output.is_converged_) return reg_output;
    coarse_aligned_ = transformPcd(src, reg_output.pose_between_eig_);
    const auto& fine_output = icpAlignment(coarse_aligned_, dst);
    const auto quatro_tf_ = reg_output.pose_between_eig_;
    reg_output = fine_output;
    reg_output.pose_between_eig_ = fine_output.pose_between_eig_ * quatro_tf_;
    return reg_output;
  }
};

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  LoopClosureLike lc;
  const pcl::PointCloud<PointType> src_in = load(argv[1]), dst_in = load(argv[2]);
  const RegistrationOutput r = lc.coarseToFineAlignment(src_in, dst_in);
  std::printf("%d %d %.17g", (int)r.is_valid_, (int)r.is_converged_, r.score_);
  for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) std::printf(" %.17g", r.pose_between_eig_(a, b));
  // the matcher on its own, with upstream's optimizedMatching(thr_dist, num_max_corres, tuple_scale) signature
  {
    qn_ctx* mctx = nullptr;
    const pcl::PointCloud<PointType> a = load(argv[1]), b = load(argv[2]);
    if (qn_ctx_create(0, (uint32_t)(std::max(a.size(), b.size()) + 4096), &mctx) != QN_OK) return 3;
    quatro_matcher<PointType> matcher(mctx);
    std::vector<float> fa, fb;
    if (matcher.computeFPFH(a, fa) != QN_OK || matcher.computeFPFH(b, fb) != QN_OK) return 4;
    matcher.setInput(a, b, fa, fb);
    matcher.optimizedMatching(35.0f, 200, 0.95f);
    long long chk = 0; for (const auto& pr : matcher.corres_) chk += 31LL * pr.first + pr.second;
    std::printf(" %zu %lld", matcher.corres_.size(), chk);
    qn_ctx_destroy(mctx);
  }
  std::printf("\n");
  const int reps = argc > 3 ? std::atoi(argv[3]) : 0;
  if (reps > 0) {
    std::vector<double> ms;
    for (int i = 0; i < reps; i++) {
      const auto t0 = std::chrono::steady_clock::now(); (void)lc.coarseToFineAlignment(src_in, dst_in);
      ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    std::sort(ms.begin(), ms.end());
    std::printf("BENCH %d %.4f %.4f\n", reps, ms[ms.size() / 2], ms[0]);
  }
  return 0;
}
