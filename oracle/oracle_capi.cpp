// ORACLE - TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).  PARITY UNPINNED.
// Plain-C entry points so tests/ and bench.py's cpu_baseline leg can drive the restatement
// through ctypes.  Not the product boundary: that is include/qn_engine.h.
#include "gicp_oracle.hpp"
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

}  // extern "C"
