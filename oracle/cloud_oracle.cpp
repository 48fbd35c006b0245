// ORACLE - TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).  PARITY UNPINNED.
// The feeder of the hot path: LoopClosure::setSrcAndDstCloud
// (fast_lio_sam_qn/src/loop_closure.cpp:58-108) = transformPcd per keyframe (include/utilities.hpp:164-175,
// pcl::transformPointCloud with a Matrix4d), concatenation, voxelizePcd (utilities.hpp:38-51, pcl::VoxelGrid),
// and LoopClosure::fetchClosestKeyframeIdx (loop_closure.cpp:34-56).
// pcl::VoxelGrid restated from the published PCL algorithm: leaf index = floor(x * (1/leaf)) - floor(min * (1/leaf))
// in f32, linear index x + y*dx + z*dx*dy, output ordered by index, one centroid per leaf.  PCL sorts
// (index, point) pairs with an unstable std::sort and sums each leaf in f32 in that order; here the order inside a
// leaf is fixed to ascending point index (one of the orders PCL may produce), f32 sequential sums, centroid = sum / n.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>
#include <numeric>

extern "C" {

void orc_transform_pcd(const float* xyz, int n, const double* T, float* out) {
  for (int i = 0; i < n; i++) {
    const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    for (int r = 0; r < 3; r++) out[3 * i + r] = (float)(((T[4 * r] * x + T[4 * r + 1] * y) + T[4 * r + 2] * z) + T[4 * r + 3]);
  }
}

// returns the number of output points (<= n); when the leaf grid would overflow 2^31 cells PCL warns and passes the input through unfiltered
int orc_voxel_grid(const float* xyz, int n, float leaf, float* out) {
  if (n == 0) return 0;
  const float inv = 1.0f / leaf;
  float mn[3] = {xyz[0], xyz[1], xyz[2]}, mx[3] = {xyz[0], xyz[1], xyz[2]};
  for (int i = 0; i < n; i++) for (int d = 0; d < 3; d++) { mn[d] = std::min(mn[d], xyz[3 * i + d]); mx[d] = std::max(mx[d], xyz[3 * i + d]); }
  int minb[3], divb[3];
  for (int d = 0; d < 3; d++) { minb[d] = (int)std::floor(mn[d] * inv); const int maxb = (int)std::floor(mx[d] * inv); divb[d] = maxb - minb[d] + 1; }
  // pcl::VoxelGrid::applyFilter's overflow guard, in PCL's own arithmetic: dx = int64((max - min) * inverse_leaf) + 1 per axis (f32
  // product), "Leaf size is too small for the input dataset. Integer indices would overflow." -> output = *input_ (unfiltered).
  {
    const int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1, dy = (int64_t)((mx[1] - mn[1]) * inv) + 1, dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
    if (dx * dy * dz > (int64_t)INT32_MAX) { std::memcpy(out, xyz, sizeof(float) * 3 * (size_t)n); return n; }
  }
  std::vector<uint64_t> key(n);
  for (int i = 0; i < n; i++) {
    const int i0 = (int)(std::floor(xyz[3 * i] * inv) - (float)minb[0]);
    const int i1 = (int)(std::floor(xyz[3 * i + 1] * inv) - (float)minb[1]);
    const int i2 = (int)(std::floor(xyz[3 * i + 2] * inv) - (float)minb[2]);
    const uint32_t idx = (uint32_t)(i0 + i1 * divb[0] + i2 * divb[0] * divb[1]);
    key[i] = ((uint64_t)idx << 32) | (uint32_t)i;
  }
  std::sort(key.begin(), key.end());
  int m = 0;
  for (int a = 0; a < n;) {
    int b = a; float s[3] = {0.f, 0.f, 0.f};
    while (b < n && (key[b] >> 32) == (key[a] >> 32)) { const uint32_t i = (uint32_t)key[b]; for (int d = 0; d < 3; d++) s[d] = s[d] + xyz[3 * i + d]; b++; }
    const float cnt = (float)(b - a);
    for (int d = 0; d < 3; d++) out[3 * m + d] = s[d] / cnt;
    m++; a = b;
  }
  return m;
}

// LoopClosure::fetchClosestKeyframeIdx (loop_closure.cpp:34-56) generalised: every keyframe idx < n-1 within `radius`
// of the query position and more than `tdiff` older, ascending distance (ties: lower index), at most max_k.
// out[0] is exactly the reference's single choice.
int orc_loop_candidates(const double* pos, const double* stamp, int n, int query, double radius, double tdiff, int max_k, int* out) {
  std::vector<std::pair<double, int>> c;
  for (int i = 0; i + 1 < n; i++) {
    const double dx = pos[3 * i] - pos[3 * query], dy = pos[3 * i + 1] - pos[3 * query + 1], dz = pos[3 * i + 2] - pos[3 * query + 2];
    const double d = std::sqrt(dx * dx + dy * dy + dz * dz);
    if (radius > d && tdiff < (stamp[query] - stamp[i])) c.emplace_back(d, i);
  }
  std::sort(c.begin(), c.end());
  int m = 0;
  for (auto& e : c) { if (m >= max_k) break; out[m++] = e.second; }
  return m;
}

}  // extern "C"
