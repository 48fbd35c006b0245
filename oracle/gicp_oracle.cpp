// ORACLE - TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).  PARITY UNPINNED.
// Restatement of Nano-GICP per SURVEY.md Appendix A.1; call sequence and accept logic per
// fast_lio_sam_qn/src/loop_closure.cpp:110-136.  Compiled with -ffp-contract=off so the f32
// distance / transform arithmetic is plain mul+add in a fixed order (the reference is built
// -O3 without -march flags, fast_lio_sam_qn/CMakeLists.txt:6-16, i.e. no FMA contraction).
#include "gicp_oracle.hpp"
#include <omp.h>
#include <cfloat>
#include <cstdio>
#include <numeric>

namespace orc {

// ------------------------------------------------------------------ KD-tree
static inline float sqdist3(const float* a, const float* b) {
  float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return dx * dx + dy * dy + dz * dz;      // L2_Simple_Adaptor order: x, y, z
}

void KdTree::build(const float* xyz, int n) {
  pts_ = xyz; n_ = n;
  order_.resize(n); std::iota(order_.begin(), order_.end(), 0);
  nodes_.clear(); nodes_.reserve(n / 4 + 8);
  if (n > 0) build_rec(0, n);
}

int KdTree::build_rec(int begin, int end) {
  int id = (int)nodes_.size();
  nodes_.push_back(Node{-1, -1, begin, end, 0, 0.f});
  if (end - begin <= 10) return id;                       // leaf_max_size = 10 (nanoflann default in nano_gicp)
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = begin; i < end; i++) for (int d = 0; d < 3; d++) {
    float v = pts_[3 * order_[i] + d]; lo[d] = std::min(lo[d], v); hi[d] = std::max(hi[d], v); }
  int axis = 0; float ext = hi[0] - lo[0];
  for (int d = 1; d < 3; d++) if (hi[d] - lo[d] > ext) { ext = hi[d] - lo[d]; axis = d; }
  if (!(ext > 0.f)) return id;                            // all points identical: keep as a leaf
  int mid = (begin + end) / 2;
  std::nth_element(order_.begin() + begin, order_.begin() + mid, order_.begin() + end, [&](int a, int b) {
    float va = pts_[3 * a + axis], vb = pts_[3 * b + axis]; return va < vb || (va == vb && a < b); });
  float split = pts_[3 * order_[mid] + axis];
  int l = build_rec(begin, mid), r = build_rec(mid, end);
  nodes_[id].left = l; nodes_[id].right = r; nodes_[id].axis = axis; nodes_[id].split = split;
  return id;
}

namespace {
struct Cand { float d2; int idx; };
inline bool better(const Cand& a, const Cand& b) { return a.d2 < b.d2 || (a.d2 == b.d2 && a.idx < b.idx); }
}

int KdTree::knn(const float q[3], int k, int* idx, float* d2) const {
  if (n_ == 0 || k <= 0) return 0;
  k = std::min(k, n_);
  std::vector<Cand> heap; heap.reserve(k + 1);            // max-heap on `better` (worst on top)
  auto cmp = [](const Cand& a, const Cand& b) { return better(a, b); };
  // iterative DFS, near child first; far child pushed with its plane distance checked on pop
  struct Item { int node; float bound; };
  std::vector<Item> st; st.reserve(64); st.push_back({0, 0.f});
  while (!st.empty()) {
    Item it = st.back(); st.pop_back();
    if ((int)heap.size() == k && it.bound > heap.front().d2) continue;   // '>' keeps equal-distance ties reachable
    const Node& nd = nodes_[it.node];
    if (nd.left < 0) {
      for (int i = nd.begin; i < nd.end; i++) {
        int pi = order_[i]; Cand c{sqdist3(q, pts_ + 3 * pi), pi};
        if ((int)heap.size() < k) { heap.push_back(c); std::push_heap(heap.begin(), heap.end(), cmp); }
        else if (better(c, heap.front())) { std::pop_heap(heap.begin(), heap.end(), cmp); heap.back() = c; std::push_heap(heap.begin(), heap.end(), cmp); }
      }
      continue;
    }
    float diff = q[nd.axis] - nd.split;
    float pd = diff * diff;
    int nearc = diff < 0 ? nd.left : nd.right, farc = diff < 0 ? nd.right : nd.left;
    // points with coordinate == split may sit on either side after nth_element: bound 0 for diff==0 handles it
    st.push_back({farc, std::max(it.bound, pd)});
    st.push_back({nearc, it.bound});
  }
  std::sort(heap.begin(), heap.end(), better);
  for (size_t i = 0; i < heap.size(); i++) { idx[i] = heap[i].idx; d2[i] = heap[i].d2; }
  return (int)heap.size();
}

// ------------------------------------------------------------------ NanoGICP
int NanoGicpOracle::threads() const { return params.num_threads > 0 ? params.num_threads : omp_get_max_threads(); }

void NanoGicpOracle::setInputSource(const float* xyz, int n) {           // SURVEY A.1.2; loop_closure.cpp:120
  src_.assign(xyz, xyz + 3 * (size_t)n); src_tree_.build(src_.data(), n); src_cov_.clear();
}
void NanoGicpOracle::setInputTarget(const float* xyz, int n) {           // loop_closure.cpp:122
  tgt_.assign(xyz, xyz + 3 * (size_t)n); tgt_tree_.build(tgt_.data(), n); tgt_cov_.clear();
}
bool NanoGicpOracle::calculateSourceCovariances() { return calc_cov(src_, src_tree_, src_cov_); }   // loop_closure.cpp:121
bool NanoGicpOracle::calculateTargetCovariances() { return calc_cov(tgt_, tgt_tree_, tgt_cov_); }   // loop_closure.cpp:123

// k-NN incl. the point itself, centred 3xk block, cov = X X^T / k, PLANE regularisation
// (singular values replaced by (1, 1, 1e-3)).  cov is symmetric PSD so SVD == eigendecomposition.
bool NanoGicpOracle::calc_cov(const std::vector<float>& pts, const KdTree& tree, std::vector<std::array<double, 9>>& covs) {
  const int n = (int)(pts.size() / 3), k = params.k_correspondences;
  covs.assign(n, std::array<double, 9>{});
#pragma omp parallel for num_threads(threads()) schedule(guided, 8)
  for (int i = 0; i < n; i++) {
    std::vector<int> idx(k); std::vector<float> d2(k);
    int found = tree.knn(&pts[3 * i], k, idx.data(), d2.data());
    double mean[3] = {0, 0, 0};
    for (int j = 0; j < found; j++) for (int d = 0; d < 3; d++) mean[d] += (double)pts[3 * idx[j] + d];
    for (int d = 0; d < 3; d++) mean[d] /= found;
    Mat3 cov{};
    for (int j = 0; j < found; j++) {
      double c[3]; for (int d = 0; d < 3; d++) c[d] = (double)pts[3 * idx[j] + d] - mean[d];
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) cov.m[a][b] += c[a] * c[b];
    }
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) cov.m[a][b] /= found;
    double w[3]; Mat3 V; sym_eig3(cov, w, V);
    const double vals[3] = {1.0, 1.0, 1e-3};
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
      double s = 0; for (int e = 0; e < 3; e++) s += V.m[a][e] * vals[e] * V.m[b][e];
      covs[i][3 * a + b] = s;
    }
  }
  return true;
}

static inline void xform_f32(const float Tf[16], const float* p, float* out) {
  // Eigen (Isometry3f * Vector4f): ((c0*x + c1*y) + c2*z) + c3*w, w = 1   
  for (int r = 0; r < 3; r++) out[r] = ((Tf[4 * r + 0] * p[0] + Tf[4 * r + 1] * p[1]) + Tf[4 * r + 2] * p[2]) + Tf[4 * r + 3];
}

void NanoGicpOracle::update_correspondences(const double T[16]) {
  const int n = (int)(src_.size() / 3);
  float Tf[16]; for (int i = 0; i < 16; i++) Tf[i] = (float)T[i];
  corr_.assign(n, -1); sqd_.assign(n, 0.f); mahal_.assign(n, std::array<double, 9>{});
  Mat3 R; for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) R.m[a][b] = T[4 * a + b];
  const Mat3 Rt = mat3_T(R);
  const double thr = params.max_corr_dist * params.max_corr_dist;
#pragma omp parallel for num_threads(threads()) schedule(guided, 8)
  for (int i = 0; i < n; i++) {
    float pt[3]; xform_f32(Tf, &src_[3 * i], pt);
    int j; float d2;
    if (tgt_tree_.knn(pt, 1, &j, &d2) < 1) continue;
    sqd_[i] = d2;
    corr_[i] = ((double)d2 < thr) ? j : -1;
    if (corr_[i] < 0) continue;
    Mat3 CA, CB;
    std::memcpy(CA.m, src_cov_[i].data(), sizeof(CA.m)); std::memcpy(CB.m, tgt_cov_[j].data(), sizeof(CB.m));
    Mat3 RCR = mat3_add(CB, mat3_mul(mat3_mul(R, CA), Rt));
    Mat3 M = mat3_inverse(RCR);
    std::memcpy(mahal_[i].data(), M.m, sizeof(M.m));
  }
}

namespace {
struct Acc { double H[21]; double b[6]; double e; };
}

// SURVEY A.1.5 linearize: e = mu_B - T mu_A; J = [skew(T mu_A) | -I]; H += J^T M J; b += J^T M e; sum += e^T M e.
// Deterministic reduction: fixed chunks of 256 points summed in order, then chunk sums in order.
double NanoGicpOracle::linearize(const double T[16], double H[36], double b[6]) {
  update_correspondences(T);
  const int n = (int)(src_.size() / 3);
  const int CH = 256, nch = (n + CH - 1) / CH;
  std::vector<Acc> part(nch);
#pragma omp parallel for num_threads(threads()) schedule(dynamic, 4)
  for (int c = 0; c < nch; c++) {
    Acc a{};
    for (int i = c * CH; i < std::min(n, (c + 1) * CH); i++) {
      int j = corr_[i]; if (j < 0) continue;
      double mA[3] = {(double)src_[3*i], (double)src_[3*i+1], (double)src_[3*i+2]};
      double mB[3] = {(double)tgt_[3*j], (double)tgt_[3*j+1], (double)tgt_[3*j+2]};
      double tA[3], e[3];
      for (int r = 0; r < 3; r++) { tA[r] = T[4*r]*mA[0] + T[4*r+1]*mA[1] + T[4*r+2]*mA[2] + T[4*r+3]; e[r] = mB[r] - tA[r]; }
      const double* M = mahal_[i].data();
      double Me[3]; for (int r = 0; r < 3; r++) Me[r] = M[3*r]*e[0] + M[3*r+1]*e[1] + M[3*r+2]*e[2];
      a.e += e[0]*Me[0] + e[1]*Me[1] + e[2]*Me[2];
      // J (3x6) = [skew(tA) | -I]
      double J[3][6] = {{0, -tA[2], tA[1], -1, 0, 0}, {tA[2], 0, -tA[0], 0, -1, 0}, {-tA[1], tA[0], 0, 0, 0, -1}};
      double MJ[3][6];
      for (int r = 0; r < 3; r++) for (int cc = 0; cc < 6; cc++) MJ[r][cc] = M[3*r]*J[0][cc] + M[3*r+1]*J[1][cc] + M[3*r+2]*J[2][cc];
      int t = 0;
      for (int r = 0; r < 6; r++) for (int cc = r; cc < 6; cc++, t++) a.H[t] += J[0][r]*MJ[0][cc] + J[1][r]*MJ[1][cc] + J[2][r]*MJ[2][cc];
      for (int r = 0; r < 6; r++) a.b[r] += J[0][r]*Me[0] + J[1][r]*Me[1] + J[2][r]*Me[2];
    }
    part[c] = a;
  }
  Acc s{};
  for (int c = 0; c < nch; c++) { for (int t = 0; t < 21; t++) s.H[t] += part[c].H[t]; for (int t = 0; t < 6; t++) s.b[t] += part[c].b[t]; s.e += part[c].e; }
  int t = 0;
  for (int r = 0; r < 6; r++) for (int cc = r; cc < 6; cc++, t++) { H[6*r+cc] = s.H[t]; H[6*cc+r] = s.H[t]; }
  for (int r = 0; r < 6; r++) b[r] = s.b[r];
  return s.e;
}

double NanoGicpOracle::compute_error(const double T[16]) const {          // cached correspondences + mahalanobis
  const int n = (int)(src_.size() / 3);
  const int CH = 256, nch = (n + CH - 1) / CH;
  std::vector<double> part(nch, 0.0);
#pragma omp parallel for num_threads(threads()) schedule(dynamic, 4)
  for (int c = 0; c < nch; c++) {
    double a = 0;
    for (int i = c * CH; i < std::min(n, (c + 1) * CH); i++) {
      int j = corr_[i]; if (j < 0) continue;
      double e[3];
      for (int r = 0; r < 3; r++) {
        double tA = T[4*r]*(double)src_[3*i] + T[4*r+1]*(double)src_[3*i+1] + T[4*r+2]*(double)src_[3*i+2] + T[4*r+3];
        e[r] = (double)tgt_[3*j+r] - tA; }
      const double* M = mahal_[i].data();
      double Me[3]; for (int r = 0; r < 3; r++) Me[r] = M[3*r]*e[0] + M[3*r+1]*e[1] + M[3*r+2]*e[2];
      a += e[0]*Me[0] + e[1]*Me[1] + e[2]*Me[2];
    }
    part[c] = a;
  }
  double s = 0; for (int c = 0; c < nch; c++) s += part[c];
  return s;
}

static void iso_mul(const double A[16], const double B[16], double C[16]) {   // Isometry3d product
  double r[16] = {0};
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) r[4*i+j] = A[4*i]*B[j] + A[4*i+1]*B[4+j] + A[4*i+2]*B[8+j];
    r[4*i+3] = A[4*i]*B[3] + A[4*i+1]*B[7] + A[4*i+2]*B[11] + A[4*i+3];
  }
  r[15] = 1.0; std::memcpy(C, r, sizeof(r));
}

static void make_delta(const double d[6], double delta[16]) {
  Mat3 R = so3_exp(d);
  for (int i = 0; i < 16; i++) delta[i] = 0;
  for (int a = 0; a < 3; a++) { for (int b = 0; b < 3; b++) delta[4*a+b] = R.m[a][b]; delta[4*a+3] = d[3+a]; }
  delta[15] = 1.0;
}

bool NanoGicpOracle::is_converged(const double delta[16], IterTrace* tr) const {
  double mr = 0, mt = 0;
  for (int a = 0; a < 3; a++) { for (int b = 0; b < 3; b++) mr = std::max(mr, std::fabs(delta[4*a+b] - (a == b ? 1.0 : 0.0))); mt = std::max(mt, std::fabs(delta[4*a+3])); }
  if (tr) { tr->max_dR = mr; tr->max_dt = mt; }
  return std::max(mr / params.rotation_epsilon, mt / params.transformation_epsilon) < 1.0;
}

bool NanoGicpOracle::step_gn(double x0[16], double delta[16], IterTrace& tr) {
  double H[36], b[6]; tr.y0 = linearize(x0, H, b);
  double A[6][6], rhs[6], d[6];
  for (int i = 0; i < 6; i++) { for (int j = 0; j < 6; j++) A[i][j] = H[6*i+j]; rhs[i] = -b[i]; }
  ldlt_solve6(A, rhs, d);
  make_delta(d, delta);
  iso_mul(delta, x0, x0);
  std::memcpy(final_H_, H, sizeof(H));
  tr.inner = 1; tr.accepted = 1; tr.lambda = 0; tr.rho = 0;
  return true;
}

bool NanoGicpOracle::step_lm(double x0[16], double delta[16], IterTrace& tr) {
  double H[36], b[6]; double y0 = linearize(x0, H, b); tr.y0 = y0;
  if (lm_lambda_ < 0.0) {
    double mx = 0; for (int i = 0; i < 6; i++) mx = std::max(mx, std::fabs(H[7*i]));
    lm_lambda_ = params.lm_init_lambda_factor * mx;
  }
  double nu = 2.0;
  for (int it = 0; it < params.lm_max_iterations; it++) {
    double A[6][6], rhs[6], d[6];
    for (int i = 0; i < 6; i++) { for (int j = 0; j < 6; j++) A[i][j] = H[6*i+j] + (i == j ? lm_lambda_ : 0.0); rhs[i] = -b[i]; }
    ldlt_solve6(A, rhs, d);
    make_delta(d, delta);
    double xi[16]; iso_mul(delta, x0, xi);
    double yi = compute_error(xi);
    double den = 0; for (int i = 0; i < 6; i++) den += d[i] * (lm_lambda_ * d[i] - b[i]);
    double rho = (y0 - yi) / den;
    tr.inner = it + 1; tr.rho = rho; tr.lambda = lm_lambda_;
    if (rho < 0) {
      if (is_converged(delta, nullptr)) { tr.accepted = 0; return true; }
      lm_lambda_ = nu * lm_lambda_; nu = 2 * nu; continue;
    }
    std::memcpy(x0, xi, sizeof(xi));
    lm_lambda_ = lm_lambda_ * std::max(1.0 / 3.0, 1 - std::pow(2 * rho - 1, 3));
    std::memcpy(final_H_, H, sizeof(H));
    tr.accepted = 1;
    return true;
  }
  tr.accepted = 0;
  return false;
}

// LsqRegistration::computeTransformation behind pcl::Registration::align (A.1.6)
void NanoGicpOracle::align(const double guess[16], GicpResult* out) {
  double x0[16]; std::memcpy(x0, guess, sizeof(x0));
  lm_lambda_ = -1.0; bool converged = false; int iters = 0;
  for (int i = 0; i < 36; i++) final_H_[i] = (i % 7 == 0) ? 1.0 : 0.0;
  out->trace.clear();
  const int maxit = params.force_iterations > 0 ? params.force_iterations : params.max_iterations;
  for (int i = 0; i < maxit && !converged; i++) {
    iters = i + 1;
    double delta[16]; IterTrace tr{};
    bool ok = (params.optimizer == 0) ? step_lm(x0, delta, tr) : step_gn(x0, delta, tr);
    if (!ok) { out->trace.push_back(tr); break; }            // "lm not converged!!"
    converged = is_converged(delta, &tr);
    if (params.force_iterations > 0) converged = false;
    out->trace.push_back(tr);
  }
  std::memcpy(out->T, x0, sizeof(x0));
  for (int i = 0; i < 16; i++) out->Tf[i] = (float)x0[i];
  std::memcpy(out->H, final_H_, sizeof(final_H_));
  out->iterations = iters; out->converged = converged ? 1 : 0;
  out->fitness = getFitnessScore(out->Tf, DBL_MAX);
}

static inline void xform_pcl_f32(const float Tf[16], const float* p, float* out) {
  // pcl::transformPointCloud, Matrix4f, SSE2 Transformer (PCL >= 1.10): x*c0 + (y*c1 + (z*c2 + c3))
  for (int r = 0; r < 3; r++) out[r] = Tf[4*r] * p[0] + (Tf[4*r+1] * p[1] + (Tf[4*r+2] * p[2] + Tf[4*r+3]));
}

void NanoGicpOracle::transformedSource(const float Tf[16], float* out_xyz) const {
  const int n = (int)(src_.size() / 3);
  for (int i = 0; i < n; i++) xform_pcl_f32(Tf, &src_[3 * i], out_xyz + 3 * i);
}

// pcl::Registration::getFitnessScore: mean of f32 squared NN distances over all
// source points, summed in f64 in index order, no gating (max_range = DBL_MAX at loop_closure.cpp:127).
// (The reference runs this single-threaded; the per-point searches here are parallel, the sum is in order.)
double NanoGicpOracle::getFitnessScore(const float Tf[16], double max_range) const {
  const int n = (int)(src_.size() / 3);
  std::vector<float> d(n);
#pragma omp parallel for num_threads(threads()) schedule(guided, 8)
  for (int i = 0; i < n; i++) { float p[3]; xform_pcl_f32(Tf, &src_[3 * i], p); int j; float d2 = 0; tgt_tree_.knn(p, 1, &j, &d2); d[i] = d2; }
  double s = 0; int nr = 0;
  for (int i = 0; i < n; i++) if ((double)d[i] <= max_range) { s += (double)d[i]; nr++; }
  return nr > 0 ? s / nr : DBL_MAX;
}

}  // namespace orc
