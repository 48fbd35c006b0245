// ORACLE - TEST INFRASTRUCTURE ONLY (see oracle_math.hpp / quatro_oracle.hpp headers).  PARITY UNPINNED.
// Quatro coarse registration restated per SURVEY.md Appendix A.2; call site
// fast_lio_sam_qn/src/loop_closure.cpp:144, parameters :18-27.
#include "quatro_oracle.hpp"
#include "oracle_math.hpp"
#include <omp.h>
#include <cmath>
#include <cfloat>
#include <algorithm>
#include <unordered_map>
#include <numeric>
#include <functional>

namespace orc {

// ------------------------------------------------------------------ deterministic f32 atan2
// Cephes atanf reduction + polynomial, plain f32 mul/add/div in a fixed order (the library is built
// -ffp-contract=off); the GPU path runs the same sequence, so histogram bins agree bit for bit.
static inline float qn_atanf_pos(float x) {          // x >= 0
  float y;
  if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
  else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
  else y = 0.0f;
  const float z = x * x;
  y += (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
  return y;
}
float qn_atan2f(float y, float x) {
  const float PI_F = 3.14159265358979323846f;
  if (x == 0.0f) { if (y == 0.0f) return 0.0f; return y > 0.0f ? 1.5707963267948966f : -1.5707963267948966f; }
  const float a = qn_atanf_pos(std::fabs(y) / std::fabs(x));
  float r = x > 0.0f ? a : PI_F - a;
  return y < 0.0f ? -r : r;
}

// ------------------------------------------------------------------ radius search (hash grid, cell = radius)
namespace {
struct RadiusGrid {
  const float* p; int n; float cell, r2; float mn[3];
  std::unordered_map<long long, std::vector<int>> cells;
  static long long key(int x, int y, int z) { return ((long long)(x + (1 << 20)) << 42) | ((long long)(y + (1 << 20)) << 21) | (long long)(z + (1 << 20)); }
  RadiusGrid(const float* xyz, int n_, double radius) : p(xyz), n(n_) {
    cell = (float)radius; r2 = (float)(radius * radius);
    for (int d = 0; d < 3; d++) { mn[d] = FLT_MAX; for (int i = 0; i < n; i++) mn[d] = std::min(mn[d], p[3 * i + d]); }
    for (int i = 0; i < n; i++) cells[key(c(i, 0), c(i, 1), c(i, 2))].push_back(i);
  }
  int c(int i, int d) const { return (int)std::floor((p[3 * i + d] - mn[d]) / cell); }
  // neighbours with f32 d2 < r2, ascending index, the query itself included
  void query(int i, std::vector<int>& idx, std::vector<float>& d2) const {
    idx.clear(); d2.clear();
    const int cx = c(i, 0), cy = c(i, 1), cz = c(i, 2);
    for (int dz = -1; dz <= 1; dz++) for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
      auto it = cells.find(key(cx + dx, cy + dy, cz + dz));
      if (it == cells.end()) continue;
      for (int j : it->second) {
        float ex = p[3*i] - p[3*j], ey = p[3*i+1] - p[3*j+1], ez = p[3*i+2] - p[3*j+2];
        float d = ex * ex + ey * ey + ez * ez;
        if (d < r2) idx.push_back(j);
      }
    }
    std::sort(idx.begin(), idx.end());
    for (int j : idx) { float ex = p[3*i] - p[3*j], ey = p[3*i+1] - p[3*j+1], ez = p[3*i+2] - p[3*j+2]; d2.push_back(ex * ex + ey * ey + ez * ez); }
  }
};
}  // namespace

// ------------------------------------------------------------------ normals
void compute_normals(const float* xyz, int n, double radius, std::vector<float>& normals) {
  normals.assign((size_t)3 * n, std::nanf(""));
  RadiusGrid g(xyz, n, radius);
#pragma omp parallel
  {
    std::vector<int> idx; std::vector<float> d2;
#pragma omp for schedule(dynamic, 64)
    for (int i = 0; i < n; i++) {
      g.query(i, idx, d2);
      if (idx.size() < 3) continue;
      double mean[3] = {0, 0, 0};
      for (int j : idx) for (int d = 0; d < 3; d++) mean[d] += (double)xyz[3 * j + d];
      for (int d = 0; d < 3; d++) mean[d] /= (double)idx.size();
      Mat3 cov{};
      for (int j : idx) { double c[3]; for (int d = 0; d < 3; d++) c[d] = (double)xyz[3 * j + d] - mean[d];
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) cov.m[a][b] += c[a] * c[b]; }
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) cov.m[a][b] /= (double)idx.size();
      double w[3]; Mat3 V; sym_eig3(cov, w, V);
      double nx = V.m[0][2], ny = V.m[1][2], nz = V.m[2][2];
      // flipNormalTowardsViewpoint with viewpoint (0,0,0): keep n . (vp - p) >= 0
      if (-(nx * (double)xyz[3*i] + ny * (double)xyz[3*i+1] + nz * (double)xyz[3*i+2]) < 0) { nx = -nx; ny = -ny; nz = -nz; }
      normals[3*i] = (float)nx; normals[3*i+1] = (float)ny; normals[3*i+2] = (float)nz;
    }
  }
}

// pcl::computePairFeatures, f32, fixed operation order
static bool pair_features(const float* p1, const float* n1, const float* p2, const float* n2, float& f1, float& f2, float& f3) {
  float dp[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  const float f4 = std::sqrt((dp[0] * dp[0] + dp[1] * dp[1]) + dp[2] * dp[2]);
  if (f4 == 0.0f) return false;
  const float angle1 = ((n1[0] * dp[0] + n1[1] * dp[1]) + n1[2] * dp[2]) / f4;
  const float angle2 = ((n2[0] * dp[0] + n2[1] * dp[1]) + n2[2] * dp[2]) / f4;
  const float* a = n1; const float* b = n2;
  if (std::fabs(angle1) < std::fabs(angle2)) {       // acos(|angle1|) > acos(|angle2|): the other point becomes the source
    a = n2; b = n1; dp[0] = -dp[0]; dp[1] = -dp[1]; dp[2] = -dp[2]; f3 = -angle2;
  } else f3 = angle1;
  float v[3] = {dp[1] * a[2] - dp[2] * a[1], dp[2] * a[0] - dp[0] * a[2], dp[0] * a[1] - dp[1] * a[0]};
  const float vn = std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
  if (vn == 0.0f) return false;
  v[0] /= vn; v[1] /= vn; v[2] /= vn;
  const float w[3] = {a[1] * v[2] - a[2] * v[1], a[2] * v[0] - a[0] * v[2], a[0] * v[1] - a[1] * v[0]};
  f2 = (v[0] * b[0] + v[1] * b[1]) + v[2] * b[2];
  f1 = qn_atan2f((w[0] * b[0] + w[1] * b[1]) + w[2] * b[2], (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]);
  return true;
}

static inline bool finite3(const float* v) { return std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]); }

void compute_fpfh(const float* xyz, int n, double normal_radius, double fpfh_radius, std::vector<float>& normals,
                  std::vector<float>& spfh, std::vector<float>& fpfh) {
  compute_normals(xyz, n, normal_radius, normals);
  spfh.assign((size_t)33 * n, 0.f); fpfh.assign((size_t)33 * n, std::nanf(""));
  RadiusGrid g(xyz, n, fpfh_radius);
  const float d_pi = 1.0f / (2.0f * (float)M_PI);
#pragma omp parallel
  {
    std::vector<int> idx; std::vector<float> d2;
#pragma omp for schedule(dynamic, 64)
    for (int i = 0; i < n; i++) {                       // SPFH
      if (!finite3(&normals[3 * i])) continue;
      g.query(i, idx, d2);
      int cnt[33] = {0};
      for (int j : idx) {
        if (j == i || !finite3(&normals[3 * j])) continue;
        float f1, f2, f3;
        if (!pair_features(&xyz[3 * i], &normals[3 * i], &xyz[3 * j], &normals[3 * j], f1, f2, f3)) continue;
        int h = (int)std::floor(11.0 * (((double)f1 + M_PI) * (double)d_pi)); h = std::min(std::max(h, 0), 10); cnt[h]++;
        h = (int)std::floor(11.0 * (((double)f2 + 1.0) * 0.5)); h = std::min(std::max(h, 0), 10); cnt[11 + h]++;
        h = (int)std::floor(11.0 * (((double)f3 + 1.0) * 0.5)); h = std::min(std::max(h, 0), 10); cnt[22 + h]++;
      }
      const float incr = 100.0f / (float)((int)idx.size() - 1);
      for (int b = 0; b < 33; b++) spfh[(size_t)33 * i + b] = cnt[b] > 0 ? (float)cnt[b] * incr : 0.f;
    }
#pragma omp for schedule(dynamic, 64)
    for (int i = 0; i < n; i++) {                       // FPFH = sum_q SPFH(q) / d2, each 11-bin group normalised to 100
      if (!finite3(&normals[3 * i])) continue;
      g.query(i, idx, d2);
      double acc[33] = {0};
      for (size_t t = 0; t < idx.size(); t++) {
        if (d2[t] == 0.0f) continue;
        const float w = 1.0f / d2[t];
        const float* s = &spfh[(size_t)33 * idx[t]];
        for (int b = 0; b < 33; b++) acc[b] += (double)(s[b] * w);
      }
      double sum[3] = {0, 0, 0};
      for (int b = 0; b < 33; b++) sum[b / 11] += acc[b];
      if (sum[0] == 0.0) continue;                      // no usable neighbourhood: descriptor stays NaN (excluded from matching)
      for (int b = 0; b < 33; b++) fpfh[(size_t)33 * i + b] = (float)(acc[b] * (sum[b / 11] != 0.0 ? 100.0 / sum[b / 11] : 0.0));
    }
  }
}

// ------------------------------------------------------------------ matching
void feature_nn(const float* q, int nq, const float* c, int nc, std::vector<int>& nn) {
  nn.assign(nq, -1);
#pragma omp parallel for schedule(dynamic, 16)
  for (int i = 0; i < nq; i++) {
    const float* a = q + (size_t)33 * i;
    if (!std::isfinite(a[0])) continue;
    float best = FLT_MAX; int bi = -1;
    for (int j = 0; j < nc; j++) {
      const float* b = c + (size_t)33 * j;
      float s = 0.f;
      for (int d = 0; d < 33; d++) { float t = a[d] - b[d]; s = s + t * t; }
      if (s < best) { best = s; bi = j; }                 // NaN candidates never win; ties keep the lowest index
    }
    nn[i] = bi;
  }
}

static inline uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

// optimized = Quatro's Matcher::optimizedMatching(thr_dist, num_max_corres, tuple_scale): distance gate + correspondence cap;
// !optimized = TEASER++'s Matcher::advancedMatching (the use_optimized_matching_ == false branch, loop_closure.h:40,
// loop_closure.cpp:25; README.md:21 quotes its cost): the same lazily cross-checked feature matches, no distance gate, and the
// tuple test runs all ncorr * 100 trials (TEASER++'s matcher has no early exit; SURVEY A.2.3 marks the cap there "believed").
static void match_impl(const float* src, int ns, const float* dst, int nt, const float* fs, const float* ft, const QuatroParams& p, bool optimized,
                       std::vector<std::pair<int, int>>& mutual, std::vector<std::pair<int, int>>& corres) {
  mutual.clear(); corres.clear();
  const bool swapped = nt > ns;                          // fi = the larger cloud, fj = the smaller
  const float* Pi = swapped ? dst : src; const float* Pj = swapped ? src : dst;
  const float* Fi = swapped ? ft : fs;   const float* Fj = swapped ? fs : ft;
  const int ni = swapped ? nt : ns, nj = swapped ? ns : nt;
  if (ni == 0 || nj == 0) return;
  std::vector<int> j_to_i; feature_nn(Fj, nj, Fi, ni, j_to_i);
  std::vector<uint8_t> hit(ni, 0);
  for (int j = 0; j < nj; j++) if (j_to_i[j] >= 0) hit[j_to_i[j]] = 1;
  std::vector<int> list; for (int i = 0; i < ni; i++) if (hit[i]) list.push_back(i);
  std::vector<float> sub((size_t)33 * list.size());
  for (size_t t = 0; t < list.size(); t++) std::copy(Fi + (size_t)33 * list[t], Fi + (size_t)33 * (list[t] + 1), sub.begin() + 33 * t);
  std::vector<int> sub_nn; feature_nn(sub.data(), (int)list.size(), Fj, nj, sub_nn);
  std::vector<int> i_to_j(ni, -1);
  for (size_t t = 0; t < list.size(); t++) i_to_j[list[t]] = sub_nn[t];
  // normalizePoints (absolute scale): subtract each cloud's own mean
  auto mean_of = [](const float* P, int n, float m[3]) { double s[3] = {0, 0, 0}; for (int i = 0; i < n; i++) for (int d = 0; d < 3; d++) s[d] += (double)P[3 * i + d]; for (int d = 0; d < 3; d++) m[d] = (float)(s[d] / n); };
  float mi[3], mj[3]; mean_of(Pi, ni, mi); mean_of(Pj, nj, mj);
  auto npt = [](const float* P, const float m[3], int i, float o[3]) { for (int d = 0; d < 3; d++) o[d] = P[3 * i + d] - m[d]; };
  auto dist = [](const float a[3], const float b[3]) { float x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2]; return std::sqrt((x * x + y * y) + z * z); };
  std::vector<std::pair<int, int>> cand;                 // (i, j), ascending j
  for (int j = 0; j < nj; j++) {
    const int i = j_to_i[j];
    if (i < 0 || i_to_j[i] != j) continue;               // cross-check
    float a[3], b[3]; npt(Pi, mi, i, a); npt(Pj, mj, j, b);
    if (optimized && dist(a, b) > (float)p.distance_threshold) continue;
    cand.emplace_back(i, j);
  }
  for (auto& c : cand) mutual.emplace_back(swapped ? c.second : c.first, swapped ? c.first : c.second);
  std::sort(mutual.begin(), mutual.end());
  // tuple test (FGR): seeded LCG instead of rand()
  const int ncorr = (int)cand.size();
  if (ncorr < 3) return;
  const float scale = (float)p.tuple_scale;
  uint32_t rng = p.rng_seed;
  std::vector<std::pair<int, int>> tup;
  for (long trial = 0; trial < (long)ncorr * 100; trial++) {
    const int r0 = (int)(lcg(rng) % (uint32_t)ncorr), r1 = (int)(lcg(rng) % (uint32_t)ncorr), r2 = (int)(lcg(rng) % (uint32_t)ncorr);
    float a0[3], a1[3], a2[3], b0[3], b1[3], b2[3];
    npt(Pi, mi, cand[r0].first, a0); npt(Pi, mi, cand[r1].first, a1); npt(Pi, mi, cand[r2].first, a2);
    npt(Pj, mj, cand[r0].second, b0); npt(Pj, mj, cand[r1].second, b1); npt(Pj, mj, cand[r2].second, b2);
    const float li0 = dist(a0, a1), li1 = dist(a1, a2), li2 = dist(a2, a0);
    const float lj0 = dist(b0, b1), lj1 = dist(b1, b2), lj2 = dist(b2, b0);
    if ((li0 * scale < lj0) && (lj0 < li0 / scale) && (li1 * scale < lj1) && (lj1 < li1 / scale) && (li2 * scale < lj2) && (lj2 < li2 / scale)) {
      tup.push_back(cand[r0]); tup.push_back(cand[r1]); tup.push_back(cand[r2]);
    }
    if (optimized && (int)tup.size() > p.max_num_corres) break;
  }
  for (auto& c : tup) corres.emplace_back(swapped ? c.second : c.first, swapped ? c.first : c.second);
  std::sort(corres.begin(), corres.end());
  corres.erase(std::unique(corres.begin(), corres.end()), corres.end());
}

void optimized_matching(const float* src, int ns, const float* dst, int nt, const float* fs, const float* ft,
                        const QuatroParams& p, std::vector<std::pair<int, int>>& mutual, std::vector<std::pair<int, int>>& corres) {
  match_impl(src, ns, dst, nt, fs, ft, p, true, mutual, corres);
}
void advanced_matching(const float* src, int ns, const float* dst, int nt, const float* fs, const float* ft,
                       const QuatroParams& p, std::vector<std::pair<int, int>>& mutual, std::vector<std::pair<int, int>>& corres) {
  match_impl(src, ns, dst, nt, fs, ft, p, false, mutual, corres);
}
// Matcher::calculateCorrespondences: dispatch on use_optimized_matching
void calculate_correspondences(const float* src, int ns, const float* dst, int nt, const float* fs, const float* ft,
                               const QuatroParams& p, std::vector<std::pair<int, int>>& mutual, std::vector<std::pair<int, int>>& corres) {
  match_impl(src, ns, dst, nt, fs, ft, p, p.use_optimized_matching, mutual, corres);
}

// ------------------------------------------------------------------ max clique (Bron-Kerbosch with pivot, all maximal cliques)
std::vector<int> max_clique_lex(const std::vector<uint8_t>& adj, int n) {
  std::vector<int> best;
  std::function<void(std::vector<int>&, std::vector<int>, std::vector<int>)> bk = [&](std::vector<int>& R, std::vector<int> P, std::vector<int> X) {
    if (P.empty() && X.empty()) {
      std::vector<int> c = R; std::sort(c.begin(), c.end());
      if (c.size() > best.size() || (c.size() == best.size() && c < best)) best = c;
      return;
    }
    if (R.size() + P.size() < best.size()) return;        // cannot beat (ties still explored: '<')
    int pivot = -1, pc = -1;
    for (int u : P) { int c = 0; for (int v : P) c += adj[(size_t)u * n + v]; if (c > pc) { pc = c; pivot = u; } }
    for (int u : X) { int c = 0; for (int v : P) c += adj[(size_t)u * n + v]; if (c > pc) { pc = c; pivot = u; } }
    std::vector<int> cand; for (int v : P) if (!adj[(size_t)pivot * n + v]) cand.push_back(v);
    for (int v : cand) {
      std::vector<int> P2, X2;
      for (int u : P) if (adj[(size_t)v * n + u]) P2.push_back(u);
      for (int u : X) if (adj[(size_t)v * n + u]) X2.push_back(u);
      R.push_back(v); bk(R, P2, X2); R.pop_back();
      P.erase(std::find(P.begin(), P.end(), v)); X.push_back(v);
    }
  };
  std::vector<int> R, P(n), X; std::iota(P.begin(), P.end(), 0);
  bk(R, P, X);
  return best;
}

// ------------------------------------------------------------------ TEASER++ solve with Quatro rotation
static double tls_estimate(const std::vector<double>& X, double alpha) {
  const int N = (int)X.size();
  struct H { double v; int tag; };
  std::vector<H> h; h.reserve(2 * N);
  for (int i = 0; i < N; i++) { h.push_back({X[i] - alpha, i + 1}); h.push_back({X[i] + alpha, -i - 1}); }
  std::sort(h.begin(), h.end(), [](const H& a, const H& b) { return a.v < b.v || (a.v == b.v && a.tag > b.tag); });
  const double w = 1.0 / (alpha * alpha);
  double ranges_inverse_sum = alpha * N, dot_X_weights = 0, dot_weights_consensus = 0, sum_xi = 0, sum_xi_square = 0;
  int card = 0; double best_cost = DBL_MAX, best_x = 0; bool have = false;
  for (int i = 0; i < 2 * N; i++) {
    const int idx = std::abs(h[i].tag) - 1, eps = h[i].tag > 0 ? 1 : -1;
    card += eps; dot_weights_consensus += eps * w; dot_X_weights += eps * w * X[idx]; ranges_inverse_sum -= eps * alpha;
    sum_xi += eps * X[idx]; sum_xi_square += eps * X[idx] * X[idx];
    const double x_hat = dot_X_weights / dot_weights_consensus;
    const double residual = card * x_hat * x_hat + sum_xi_square - 2 * sum_xi * x_hat;
    const double cost = residual + ranges_inverse_sum;
    if (cost == cost && (!have || cost < best_cost)) { best_cost = cost; best_x = x_hat; have = true; }
  }
  return best_x;
}

// ScalarTLSEstimator::estimate with one range per measurement (TLSScaleSolver::solveForScale: s_ij = |b_ij| / |a_ij|, range beta / |a_ij|): the sweep of tls_estimate with
// weights 1 / range^2 in the estimate, unweighted squared residuals + the ranges of the measurements outside the consensus set in the cost (TEASER++ registration.cc as recalled:
// un-vendored, parity unpinned).  inliers: |X - estimate| <= range.
static double tls_estimate_ranges(const std::vector<double>& X, const std::vector<double>& ranges, std::vector<char>& inliers) {
  const int N = (int)X.size();
  struct H { double v; int tag; };
  std::vector<H> h; h.reserve(2 * (size_t)N);
  double ranges_inverse_sum = 0;
  for (int i = 0; i < N; i++) { h.push_back({X[i] - ranges[i], i + 1}); h.push_back({X[i] + ranges[i], -i - 1}); ranges_inverse_sum += ranges[i]; }
  std::sort(h.begin(), h.end(), [](const H& a, const H& b) { return a.v < b.v || (a.v == b.v && a.tag > b.tag); });
  double dot_X_weights = 0, dot_weights_consensus = 0, sum_xi = 0, sum_xi_square = 0;
  int card = 0; double best_cost = DBL_MAX, best_x = 1.0; bool have = false;
  for (size_t i = 0; i < h.size(); i++) {
    const int idx = std::abs(h[i].tag) - 1, eps = h[i].tag > 0 ? 1 : -1;
    const double w = 1.0 / (ranges[idx] * ranges[idx]);
    card += eps; dot_weights_consensus += eps * w; dot_X_weights += eps * w * X[idx]; ranges_inverse_sum -= eps * ranges[idx];
    sum_xi += eps * X[idx]; sum_xi_square += eps * X[idx] * X[idx];
    const double x_hat = dot_X_weights / dot_weights_consensus;
    const double residual = card * x_hat * x_hat + sum_xi_square - 2 * sum_xi * x_hat;
    const double cost = residual + ranges_inverse_sum;
    if (cost == cost && (!have || cost < best_cost)) { best_cost = cost; best_x = x_hat; have = true; }
  }
  inliers.assign(N, 0);
  for (int i = 0; i < N; i++) inliers[i] = std::fabs(X[i] - best_x) <= ranges[i];
  return best_x;
}

void solve(const float* src, const float* dst, const std::vector<std::pair<int, int>>& corres, const QuatroParams& p, QuatroResult* out) {
  for (int i = 0; i < 16; i++) out->T[i] = (i % 5 == 0) ? 1.0 : 0.0;
  out->valid = 0; out->clique.clear(); out->rot_iterations = 0; out->scale = 1.0;
  const int M = (int)corres.size();
  if (M == 0) return;
  std::vector<std::array<double, 3>> S(M), D(M);
  for (int k = 0; k < M; k++) for (int d = 0; d < 3; d++) { S[k][d] = (double)src[3 * corres[k].first + d]; D[k][d] = (double)dst[3 * corres[k].second + d]; }
  auto norm3 = [](const double a[3]) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); };
  // TIMs + scale-consistency graph (scale fixed to 1 when estimate_scale = false - the shipped config, loop_closure.cpp:24 / config.yaml:37)
  const double beta = 2.0 * p.noise_bound * std::sqrt(1.0);
  std::vector<uint8_t> adj((size_t)M * M, 0);
  double scale = 1.0;
  if (p.estimate_scale) {      // TLSScaleSolver::solveForScale on all TIMs; its inliers are the graph's edges (a TIM of two coincident source points has no scale: no edge)
    std::vector<double> X, ranges; std::vector<std::pair<int, int>> ij;
    for (int i = 0; i < M; i++) for (int j = i + 1; j < M; j++) {
      double a[3], b[3]; for (int d = 0; d < 3; d++) { a[d] = S[j][d] - S[i][d]; b[d] = D[j][d] - D[i][d]; }
      const double na = norm3(a), nb = norm3(b);
      if (!(na > 0.0) || !(nb == nb)) continue;
      X.push_back(nb / na); ranges.push_back(beta / na); ij.push_back({i, j});
    }
    if (X.empty()) return;
    std::vector<char> inl;
    scale = tls_estimate_ranges(X, ranges, inl);
    if (!(scale > 0.0)) return;
    for (size_t k = 0; k < ij.size(); k++) if (inl[k]) adj[(size_t)ij[k].first * M + ij[k].second] = adj[(size_t)ij[k].second * M + ij[k].first] = 1;
    out->scale = scale;
  } else
  for (int i = 0; i < M; i++) for (int j = i + 1; j < M; j++) {
    double a[3], b[3]; for (int d = 0; d < 3; d++) { a[d] = S[j][d] - S[i][d]; b[d] = D[j][d] - D[i][d]; }
    if (std::fabs(norm3(b) - norm3(a)) <= beta) adj[(size_t)i * M + j] = adj[(size_t)j * M + i] = 1;
  }
  std::vector<int> C = max_clique_lex(adj, M);
  out->clique = C;
  const int m = (int)C.size();
  if (m <= 1) return;
  // chain TIMs over the clique
  std::vector<std::array<double, 3>> A(m), B(m);
  for (int i = 0; i < m; i++) { const int root = C[i], leaf = C[(i + 1) % m]; for (int d = 0; d < 3; d++) { A[i][d] = S[leaf][d] - S[root][d]; B[i][d] = D[leaf][d] - D[root][d]; } }
  // Quatro rotation: GNC-TLS restricted to yaw
  // TEASER++ RobustRegistrationSolver::solve(): "params.noise_bound *= (2 / solution_.scale)" on the rotation solver's params
  // before solveForRotation (TIMs are differences of two bounded measurements), which then uses pow(params_.noise_bound, 2).
  double nb2 = (2.0 * p.noise_bound) * (2.0 * p.noise_bound);
  if (p.estimate_scale) {      // pruned_dst_tims_ *= (1 / solution_.scale); params.noise_bound *= (2 / solution_.scale)
    for (int i = 0; i < m; i++) for (int d = 0; d < 3; d++) B[i][d] *= 1.0 / scale;
    const double nbr = p.noise_bound * (2.0 / scale); nb2 = nbr * nbr;
  }
  if (nb2 < 1e-16) nb2 = 1e-2;
  std::vector<double> wgt(m, 1.0), res(m);
  double mu = 1.0, prev_cost = std::numeric_limits<double>::infinity(), c = 1.0, s = 0.0;
  for (int it = 0; it < p.rot_max_iter; it++) {
    out->rot_iterations = it + 1;
    double sxy = 0, cxy = 0;
    for (int i = 0; i < m; i++) { sxy += wgt[i] * (A[i][0] * B[i][1] - A[i][1] * B[i][0]); cxy += wgt[i] * (A[i][0] * B[i][0] + A[i][1] * B[i][1]); }
    const double th = std::atan2(sxy, cxy); c = std::cos(th); s = std::sin(th);
    double maxres = 0;
    for (int i = 0; i < m; i++) {
      const double rx = B[i][0] - (c * A[i][0] - s * A[i][1]), ry = B[i][1] - (s * A[i][0] + c * A[i][1]), rz = B[i][2] - A[i][2];
      res[i] = rx * rx + ry * ry + rz * rz; maxres = std::max(maxres, res[i]);
    }
    if (it == 0) { mu = 1.0 / (2.0 * maxres / nb2 - 1.0); if (mu <= 0) break; }
    const double th1 = (mu + 1) / mu * nb2, th2 = mu / (mu + 1) * nb2;
    double cost = 0;
    for (int i = 0; i < m; i++) {
      cost += wgt[i] * res[i];
      if (res[i] >= th1) wgt[i] = 0; else if (res[i] <= th2) wgt[i] = 1; else wgt[i] = std::sqrt(nb2 * mu * (mu + 1) / res[i]) - mu;
    }
    const double diff = std::fabs(cost - prev_cost);
    mu *= p.rot_gnc_factor; prev_cost = cost;
    if (diff < p.rot_cost_diff_thr) break;
  }
  // translation: component-wise TLS on d_k - R s_k over the clique
  double t[3];
  for (int d = 0; d < 3; d++) {
    std::vector<double> X(m);
    for (int i = 0; i < m; i++) {
      const double* sp = S[C[i]].data();
      const double r[3] = {c * sp[0] - s * sp[1], s * sp[0] + c * sp[1], sp[2]};
      X[i] = D[C[i]][d] - (p.estimate_scale ? scale * r[d] : r[d]);      // solveForTranslation(scale * R * src, dst)
    }
    t[d] = tls_estimate(X, p.noise_bound);
  }
  const double R[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
  for (int a = 0; a < 3; a++) { for (int b = 0; b < 3; b++) out->T[4 * a + b] = R[3 * a + b]; out->T[4 * a + 3] = t[a]; }
  out->valid = 1;
}

void quatro_align(const float* src, int ns, const float* dst, int nt, const QuatroParams& p, QuatroResult* out) {
  std::vector<float> n1, s1, f1, n2, s2, f2;
  compute_fpfh(src, ns, p.fpfh_normal_radius, p.fpfh_radius, n1, s1, f1);
  compute_fpfh(dst, nt, p.fpfh_normal_radius, p.fpfh_radius, n2, s2, f2);
  std::vector<std::pair<int, int>> mutual;
  calculate_correspondences(src, ns, dst, nt, f1.data(), f2.data(), p, mutual, out->corres);
  solve(src, dst, out->corres, p, out);
}

}  // namespace orc
