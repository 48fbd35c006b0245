// ORACLE - TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).  PARITY UNPINNED.
//
// CPU restatement of the Quatro coarse registration that LoopClosure::coarseToFineAlignment drives
// (fast_lio_sam_qn/src/loop_closure.cpp:138-159; ctor arguments at :18-27):
//   quatro<PointType>::align(src, dst, is_converged)  ->  4x4 f64 transform.
// third_party/Quatro is an empty submodule, so the bodies follow the published algorithms as written
// down in SURVEY.md Appendix A.2: PCL NormalEstimation / FPFHEstimation (Rusu 2009), the
// FGR/TEASER++ matcher with Quatro's "optimizedMatching" (cross-checked feature NN, distance gate,
// tuple test, correspondence cap), and the TEASER++ solver with Quatro's yaw-only GNC-TLS rotation.
//
// Decisions the unpinned spec forced (shared with the GPU path, stated in DESIGN.md):
//  * radius search keeps points with f32 d2 < (float)(r*r), the query itself included;
//  * normals: covariance and eigenvectors in f64 (PCL uses f32), smallest eigenvector, flipped
//    towards the viewpoint (0,0,0), then rounded to f32;
//  * pair features in f32 with a fixed operation order and a fixed polynomial atan2 (qn_atan2f) so
//    that histogram bins are reproducible bit for bit; SPFH bin = count * (100 / (n_nbrs - 1));
//  * FPFH weighted sums accumulate in f64 and are rounded to f32 after the per-group normalisation;
//  * feature NN: exact, f32 sequential sum over the 33 dimensions, ties to the lowest index;
//  * the tuple test's rand() is replaced by a seeded LCG (the reference seeds from wall-clock and does
//    not reproduce itself); among maximum cliques the lexicographically smallest is taken.
#pragma once
#include <cstdint>
#include <vector>
#include <array>
#include <utility>

namespace orc {

struct QuatroParams {            // the 10 ctor arguments, in the order of loop_closure.cpp:18-27, + seed
  double fpfh_normal_radius = 0.9, fpfh_radius = 1.5, noise_bound = 0.3, rot_gnc_factor = 1.4, rot_cost_diff_thr = 1e-4;
  int rot_max_iter = 50;
  bool estimate_scale = false, use_optimized_matching = true;
  double distance_threshold = 35.0;
  int max_num_corres = 200;
  uint32_t rng_seed = 1;
  double tuple_scale = 0.95;
};

struct QuatroResult {
  double T[16];
  int valid;
  std::vector<std::pair<int, int>> corres;      // after the tuple test, sorted unique (src idx, dst idx)
  std::vector<int> clique;                      // indices into corres
  int rot_iterations;
  double scale = 1.0;                           // TEASER++ TLSScaleSolver's estimate (1 unless estimate_scale)
};

float qn_atan2f(float y, float x);

// normals[n][3] f32 (NaN when fewer than 3 neighbours)
void compute_normals(const float* xyz, int n, double radius, std::vector<float>& normals);
// spfh[n][33], fpfh[n][33] f32 (fpfh NaN for points without a usable neighbourhood)
void compute_fpfh(const float* xyz, int n, double normal_radius, double fpfh_radius, std::vector<float>& normals,
                  std::vector<float>& spfh, std::vector<float>& fpfh);
// exact NN of every query descriptor among the candidates (-1 when the query descriptor is not finite)
void feature_nn(const float* q, int nq, const float* c, int nc, std::vector<int>& nn);
// mutual matches + distance gate (before the tuple test), then the seeded tuple test + cap, sorted unique
void optimized_matching(const float* src, int ns, const float* dst, int nt, const float* fs, const float* ft,
                        const QuatroParams& p, std::vector<std::pair<int, int>>& mutual, std::vector<std::pair<int, int>>& corres);
// TEASER++ Matcher::advancedMatching (use_optimized_matching == false): no distance gate, no cap
void advanced_matching(const float* src, int ns, const float* dst, int nt, const float* fs, const float* ft,
                       const QuatroParams& p, std::vector<std::pair<int, int>>& mutual, std::vector<std::pair<int, int>>& corres);
// Matcher::calculateCorrespondences: one of the two, by p.use_optimized_matching
void calculate_correspondences(const float* src, int ns, const float* dst, int nt, const float* fs, const float* ft,
                               const QuatroParams& p, std::vector<std::pair<int, int>>& mutual, std::vector<std::pair<int, int>>& corres);
// TEASER++ solve with Quatro rotation on matched points
void solve(const float* src, const float* dst, const std::vector<std::pair<int, int>>& corres, const QuatroParams& p, QuatroResult* out);
void quatro_align(const float* src, int ns, const float* dst, int nt, const QuatroParams& p, QuatroResult* out);

// lexicographically smallest maximum clique of an undirected graph (adjacency as n x n bytes)
std::vector<int> max_clique_lex(const std::vector<uint8_t>& adj, int n);

}  // namespace orc
