// LoopClosure::icpAlignment (fast_lio_sam_qn/src/loop_closure.cpp:110-136) written against the drop-in
// shim exactly as the reference writes it against nano_gicp - proves the shim's surface is sufficient.
// usage: shim_icp_alignment src.bin dst.bin [t|s [reps]]  (raw float32 xyz triplets; t = set the target first; reps > 0: the icpAlignment sequence is repeated and a second
// line "BENCH reps median_ms min_ms" reports its wall time - the latency an unmodified LoopClosure would see: 32-byte PointXYZI stride, deep copies, aligned_ filled) -> prints valid converged score T(16)
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <limits>
#include <nano_gicp/point_type_nano_gicp.hpp>
#include <nano_gicp/nano_gicp.hpp>

using PointType = pcl::PointXYZI;

struct RegistrationOutput {
  bool is_valid_ = false, is_converged_ = false;
  double score_ = std::numeric_limits<double>::max();
  Eigen::Matrix4d pose_between_eig_ = Eigen::Matrix4d::Identity();
};

static pcl::PointCloud<PointType> load(const char* path) {
  pcl::PointCloud<PointType> c; FILE* f = std::fopen(path, "rb"); float v[3];
  while (f && std::fread(v, sizeof(float), 3, f) == 3) { PointType p; p.x = v[0]; p.y = v[1]; p.z = v[2]; p.intensity = 42.f; c.push_back(p); }
  if (f) std::fclose(f);
  return c;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  nano_gicp::NanoGICP<PointType, PointType> nano_gicp_;
  // LoopClosure ctor, loop_closure.cpp:9-16 with the reference's effective config
  nano_gicp_.setNumThreads(0);
  nano_gicp_.setCorrespondenceRandomness(15);
  nano_gicp_.setMaximumIterations(32);
  nano_gicp_.setRANSACIterations(5);
  nano_gicp_.setMaxCorrespondenceDistance(52.5);
  nano_gicp_.setTransformationEpsilon(0.01);
  nano_gicp_.setEuclideanFitnessEpsilon(0.01);
  nano_gicp_.setRANSACOutlierRejectionThreshold(1.0);
  const pcl::PointCloud<PointType> src = load(argv[1]), dst = load(argv[2]);
  // icpAlignment, loop_closure.cpp:113-135
  RegistrationOutput reg_output;
  pcl::PointCloud<PointType> aligned_;
  const bool target_first = argc > 3 && argv[3][0] == 't';
  const bool trace = std::getenv("QN_SHIM_TRACE") != nullptr;      // developer: wall time of the sections of every call on stderr
  auto icpAlignment = [&]() {
  const auto c0 = std::chrono::steady_clock::now();
  reg_output = RegistrationOutput();
  aligned_.clear();
  pcl::PointCloud<PointType>::Ptr src_cloud(new pcl::PointCloud<PointType>());
  pcl::PointCloud<PointType>::Ptr dst_cloud(new pcl::PointCloud<PointType>());
  *src_cloud = src;
  *dst_cloud = dst;
  const auto s0 = std::chrono::steady_clock::now();
  if (target_first) {   // the usual PCL order (target first); with a larger source this regrows the shim's context AFTER the target was set
    nano_gicp_.setInputTarget(dst_cloud);
    nano_gicp_.calculateTargetCovariances();
    nano_gicp_.setInputSource(src_cloud);
    nano_gicp_.calculateSourceCovariances();
  } else {
  nano_gicp_.setInputSource(src_cloud);
  nano_gicp_.calculateSourceCovariances();
  nano_gicp_.setInputTarget(dst_cloud);
  nano_gicp_.calculateTargetCovariances();
  }
  const auto s1 = std::chrono::steady_clock::now();
  nano_gicp_.align(aligned_);
  if (trace) std::fprintf(stderr, "shim trace: copies %.3f ms, set + covariances %.3f ms, align + aligned_ %.3f ms\n", std::chrono::duration<double, std::milli>(s0 - c0).count(),
                          std::chrono::duration<double, std::milli>(s1 - s0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - s1).count());
  reg_output.score_ = nano_gicp_.getFitnessScore();
  if (nano_gicp_.hasConverged() && reg_output.score_ < 1.5) {
    reg_output.is_valid_ = true;
    reg_output.is_converged_ = true;
    reg_output.pose_between_eig_ = nano_gicp_.getFinalTransformation().cast<double>();
  }
  };
  icpAlignment();
  std::printf("%d %d %.17g", (int)reg_output.is_valid_, (int)nano_gicp_.hasConverged(), reg_output.score_);
  const Eigen::Matrix4f T = nano_gicp_.getFinalTransformation();
  for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) std::printf(" %.9g", T(r, c));
  std::printf(" %zu %.9g %.9g\n", aligned_.size(), aligned_.size() ? aligned_[0].x : 0.f, aligned_.size() ? aligned_[0].intensity : 0.f);
  const int reps = argc > 4 ? std::atoi(argv[4]) : 0;
  if (reps > 0) {
    std::vector<double> ms;
    for (int i = 0; i < reps; i++) {
      const auto t0 = std::chrono::steady_clock::now(); icpAlignment();
      ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    std::sort(ms.begin(), ms.end());
    std::printf("BENCH %d %.4f %.4f\n", reps, ms[ms.size() / 2], ms[0]);
  }
  return 0;
}
