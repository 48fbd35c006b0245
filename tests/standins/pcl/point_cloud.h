// Minimal stand-in for <pcl/point_cloud.h>.
#pragma once
#include <memory>
#include <vector>
namespace pcl {
template <typename PointT> class PointCloud {
 public:
  using Ptr = std::shared_ptr<PointCloud<PointT>>;
  using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
  std::vector<PointT> points;
  size_t size() const { return points.size(); }
  void clear() { points.clear(); }
  void reserve(size_t n) { points.reserve(n); }
  void push_back(const PointT& p) { points.push_back(p); }
  PointT& operator[](size_t i) { return points[i]; }
  const PointT& operator[](size_t i) const { return points[i]; }
};
}
