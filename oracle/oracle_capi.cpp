// ORACLE - TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).  PARITY UNPINNED.
// Plain-C entry points so tests/ and bench.py's cpu_baseline leg can drive the restatement
// through ctypes.  Not the product boundary: that is include/qn_engine.h.
#include "gicp_oracle.hpp"
#include "quatro_oracle.hpp"
#include <cstring>
#include <chrono>
#include <omp.h>

using namespace orc;

extern "C" {

void* orc_gicp_create() { return new NanoGicpOracle(); }
void orc_gicp_destroy(void* h) { delete (NanoGicpOracle*)h; }

// params: k, max_iter, optimizer(0 LM/1 GN), lm_max_iter, force_iterations, num_threads ; doubles: max_corr_dist, trans_eps, rot_eps, lm_init_lambda_factor
void orc_gicp_set_params(void* h, const int* ip, const double* dp) {
  auto& p = ((NanoGicpOracle*)h)->params;
  p.k_correspondences = ip[0]; p.max_iterations = ip[1]; p.optimizer = ip[2]; p.lm_max_iterations = ip[3];
  p.force_iterations = ip[4]; p.num_threads = ip[5];
  p.max_corr_dist = dp[0]; p.transformation_epsilon = dp[1]; p.rotation_epsilon = dp[2]; p.lm_init_lambda_factor = dp[3];
}
void orc_gicp_set_source(void* h, const float* xyz, int n) { ((NanoGicpOracle*)h)->setInputSource(xyz, n); }
void orc_gicp_set_target(void* h, const float* xyz, int n) { ((NanoGicpOracle*)h)->setInputTarget(xyz, n); }
void orc_gicp_cov(void* h, int which) { auto* g = (NanoGicpOracle*)h; which ? g->calculateTargetCovariances() : g->calculateSourceCovariances(); }
void orc_gicp_get_cov(void* h, int which, double* out9) {
  auto* g = (NanoGicpOracle*)h; const auto& c = which ? g->targetCovs() : g->sourceCovs();
  if (!c.empty()) std::memcpy(out9, c.data(), c.size() * 9 * sizeof(double));
}
// k-NN on the source (which=0) or target (which=1) tree for nq query points
void orc_gicp_knn(void* h, int which, const float* q, int nq, int k, int* idx, float* d2) {
  auto* g = (NanoGicpOracle*)h; const KdTree& t = which ? g->targetTree() : g->sourceTree();
#pragma omp parallel for schedule(guided, 8)
  for (int i = 0; i < nq; i++) {
    int f = t.knn(q + 3 * i, k, idx + (size_t)i * k, d2 + (size_t)i * k);
    for (int j = f; j < k; j++) { idx[(size_t)i * k + j] = -1; d2[(size_t)i * k + j] = 0.f; }
  }
}
double orc_gicp_linearize(void* h, const double* T, double* H, double* b, int* corr, float* sqd) {
  auto* g = (NanoGicpOracle*)h; double e = g->linearize(T, H, b);
  if (corr) std::memcpy(corr, g->correspondences().data(), g->correspondences().size() * sizeof(int));
  if (sqd) std::memcpy(sqd, g->sqDistances().data(), g->sqDistances().size() * sizeof(float));
  return e;
}
double orc_gicp_compute_error(void* h, const double* T) { return ((NanoGicpOracle*)h)->compute_error(T); }

// out_d: T[16], H[36], fitness ; out_f: Tf[16] ; out_i: iterations, converged, trace_len ; trace: 7 doubles per outer iteration
void orc_gicp_align(void* h, const double* guess, double* out_d, float* out_f, int* out_i, double* trace, int trace_cap) {
  GicpResult r; ((NanoGicpOracle*)h)->align(guess, &r);
  std::memcpy(out_d, r.T, 16 * sizeof(double)); std::memcpy(out_d + 16, r.H, 36 * sizeof(double)); out_d[52] = r.fitness;
  std::memcpy(out_f, r.Tf, 16 * sizeof(float));
  out_i[0] = r.iterations; out_i[1] = r.converged; out_i[2] = (int)r.trace.size();
  for (int i = 0; i < (int)r.trace.size() && i < trace_cap; i++) {
    const auto& t = r.trace[i]; double* o = trace + 7 * i;
    o[0] = t.y0; o[1] = t.lambda; o[2] = t.rho; o[3] = t.max_dR; o[4] = t.max_dt; o[5] = t.inner; o[6] = t.accepted;
  }
}
double orc_gicp_fitness(void* h, const float* Tf, double max_range) { return ((NanoGicpOracle*)h)->getFitnessScore(Tf, max_range); }
void orc_gicp_transformed_source(void* h, const float* Tf, float* out) { ((NanoGicpOracle*)h)->transformedSource(Tf, out); }

// small-math hooks for known-answer tests
void orc_so3_exp(const double* om, double* R9) { Mat3 R = so3_exp(om); std::memcpy(R9, R.m, sizeof(R.m)); }
void orc_sym_eig3(const double* A9, double* w3, double* V9) { Mat3 A, V; std::memcpy(A.m, A9, sizeof(A.m)); sym_eig3(A, w3, V); std::memcpy(V9, V.m, sizeof(V.m)); }
void orc_ldlt_solve6(const double* A36, const double* rhs, double* x) { double A[6][6]; std::memcpy(A, A36, sizeof(A)); ldlt_solve6(A, rhs, x); }
int orc_num_threads() { return omp_get_max_threads(); }

// ---------------------------------------------------------------- Quatro
static QuatroParams qp_from(const double* dp, const int* ip) {
  QuatroParams p; p.fpfh_normal_radius = dp[0]; p.fpfh_radius = dp[1]; p.noise_bound = dp[2]; p.rot_gnc_factor = dp[3]; p.rot_cost_diff_thr = dp[4];
  p.distance_threshold = dp[5]; p.tuple_scale = dp[6]; p.rot_max_iter = ip[0]; p.estimate_scale = ip[1] != 0; p.use_optimized_matching = ip[2] != 0;
  p.max_num_corres = ip[3]; p.rng_seed = (uint32_t)ip[4]; return p;
}
void orc_quatro_fpfh(const float* xyz, int n, double rn, double rf, float* normals, float* spfh, float* fpfh) {
  std::vector<float> a, b, c; compute_fpfh(xyz, n, rn, rf, a, b, c);
  std::memcpy(normals, a.data(), a.size() * 4); std::memcpy(spfh, b.data(), b.size() * 4); std::memcpy(fpfh, c.data(), c.size() * 4);
}
void orc_quatro_feature_nn(const float* q, int nq, const float* c, int nc, int* nn) { std::vector<int> v; feature_nn(q, nq, c, nc, v); std::memcpy(nn, v.data(), v.size() * 4); }
// returns counts through n_out[0] (mutual), n_out[1] (corres); pair arrays sized min(ns, nt) x 2
void orc_quatro_match(const float* src, int ns, const float* dst, int nt, const float* fs, const float* ft, const double* dp, const int* ip,
                      int* mutual, int* corres, int* n_out) {
  std::vector<std::pair<int, int>> m, c; calculate_correspondences(src, ns, dst, nt, fs, ft, qp_from(dp, ip), m, c);
  for (size_t i = 0; i < m.size(); i++) { mutual[2 * i] = m[i].first; mutual[2 * i + 1] = m[i].second; }
  for (size_t i = 0; i < c.size(); i++) { corres[2 * i] = c[i].first; corres[2 * i + 1] = c[i].second; }
  n_out[0] = (int)m.size(); n_out[1] = (int)c.size();
}
// out_i: valid, clique_size, rot_iterations ; clique buffer sized ncorr
void orc_quatro_solve(const float* src, const float* dst, const int* corres, int ncorr, const double* dp, const int* ip, double* T, int* out_i, int* clique) {
  std::vector<std::pair<int, int>> c(ncorr); for (int i = 0; i < ncorr; i++) c[i] = {corres[2 * i], corres[2 * i + 1]};
  QuatroResult r; solve(src, dst, c, qp_from(dp, ip), &r);
  std::memcpy(T, r.T, sizeof(r.T)); out_i[0] = r.valid; out_i[1] = (int)r.clique.size(); out_i[2] = r.rot_iterations;
  for (size_t i = 0; i < r.clique.size(); i++) clique[i] = r.clique[i];
}
void orc_quatro_align(const float* src, int ns, const float* dst, int nt, const double* dp, const int* ip, double* T, int* out_i, int* corres, int corres_cap) {
  QuatroResult r; quatro_align(src, ns, dst, nt, qp_from(dp, ip), &r);
  std::memcpy(T, r.T, sizeof(r.T)); out_i[0] = r.valid; out_i[1] = (int)r.clique.size(); out_i[2] = r.rot_iterations; out_i[3] = (int)r.corres.size();
  for (size_t i = 0; i < r.corres.size() && (int)i < corres_cap; i++) { corres[2 * i] = r.corres[i].first; corres[2 * i + 1] = r.corres[i].second; }
}
// the solver with its scale estimate (estimate_scale): out_d[0] = scale
void orc_quatro_solve_scaled(const float* src, const float* dst, const int* corres, int ncorr, const double* dp, const int* ip, double* T, int* out_i, int* clique, double* out_d) {
  std::vector<std::pair<int, int>> c(ncorr); for (int i = 0; i < ncorr; i++) c[i] = {corres[2 * i], corres[2 * i + 1]};
  QuatroResult r; solve(src, dst, c, qp_from(dp, ip), &r);
  std::memcpy(T, r.T, sizeof(r.T)); out_i[0] = r.valid; out_i[1] = (int)r.clique.size(); out_i[2] = r.rot_iterations; out_d[0] = r.scale;
  for (size_t i = 0; i < r.clique.size(); i++) clique[i] = r.clique[i];
}
int orc_max_clique(const unsigned char* adj, int n, int* out) { std::vector<uint8_t> a(adj, adj + (size_t)n * n); auto c = max_clique_lex(a, n); for (size_t i = 0; i < c.size(); i++) out[i] = c[i]; return (int)c.size(); }
float orc_atan2f(float y, float x) { return qn_atan2f(y, x); }

}  // extern "C"
