// Minimal stand-in for <pcl/point_types.h> (PCL is not installed in this image): the layout of
// pcl::PointXYZI - 32 bytes, xyz + data[3] = 1, then intensity + padding.
#pragma once
namespace pcl {
struct alignas(16) PointXYZI {
  union { float data[4]; struct { float x, y, z; }; };
  union { struct { float intensity; }; float data_c[4]; };
  PointXYZI() : data{0.f, 0.f, 0.f, 1.f}, data_c{0.f, 0.f, 0.f, 0.f} {}
};
static_assert(sizeof(PointXYZI) == 32, "PointXYZI layout");
}
