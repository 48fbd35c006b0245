// ORACLE - TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// PARITY UNPINNED: the reference's registration arithmetic lives in un-vendored
// submodules (third_party/nano_gicp, third_party/Quatro are empty in /root/reference;
// .gitmodules:7-12) and the reference has no tests or golden vectors, so this restatement
// is anchored on the reference's call sites (fast_lio_sam_qn/src/loop_closure.cpp:9-27,
// 110-159) and on the published upstream algorithms.
//
// Small dense math used by the restatement (Eigen is not installed): 3x3 / 6x6 f64.
#pragma once
#include <cmath>
#include <cstring>
#include <algorithm>

namespace orc {

struct Mat3 { double m[3][3]; };
struct Vec3 { double v[3]; };

inline Mat3 mat3_identity() { Mat3 r{}; r.m[0][0] = r.m[1][1] = r.m[2][2] = 1.0; return r; }
inline Mat3 mat3_mul(const Mat3& a, const Mat3& b) {
  Mat3 r{};
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    double s = 0; for (int k = 0; k < 3; k++) s += a.m[i][k] * b.m[k][j]; r.m[i][j] = s; }
  return r;
}
inline Mat3 mat3_T(const Mat3& a) { Mat3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i]; return r; }
inline Mat3 mat3_add(const Mat3& a, const Mat3& b) { Mat3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] + b.m[i][j]; return r; }
inline Vec3 mat3_vec(const Mat3& a, const Vec3& x) { Vec3 r; for (int i = 0; i < 3; i++) r.v[i] = a.m[i][0]*x.v[0] + a.m[i][1]*x.v[1] + a.m[i][2]*x.v[2]; return r; }

// inverse by cofactors (the reference inverts the 4x4 [[RCR,0],[0,1]]: SURVEY A.1.4 - same 3x3 block)
inline Mat3 mat3_inverse(const Mat3& a) {
  const double (*m)[3] = a.m;
  double c00 = m[1][1]*m[2][2] - m[1][2]*m[2][1];
  double c01 = m[1][2]*m[2][0] - m[1][0]*m[2][2];
  double c02 = m[1][0]*m[2][1] - m[1][1]*m[2][0];
  double det = m[0][0]*c00 + m[0][1]*c01 + m[0][2]*c02;
  double id = 1.0 / det;
  Mat3 r;
  r.m[0][0] = c00*id; r.m[1][0] = c01*id; r.m[2][0] = c02*id;
  r.m[0][1] = (m[0][2]*m[2][1] - m[0][1]*m[2][2])*id;
  r.m[1][1] = (m[0][0]*m[2][2] - m[0][2]*m[2][0])*id;
  r.m[2][1] = (m[0][1]*m[2][0] - m[0][0]*m[2][1])*id;
  r.m[0][2] = (m[0][1]*m[1][2] - m[0][2]*m[1][1])*id;
  r.m[1][2] = (m[0][2]*m[1][0] - m[0][0]*m[1][2])*id;
  r.m[2][2] = (m[0][0]*m[1][1] - m[0][1]*m[1][0])*id;
  return r;
}

// Cyclic Jacobi eigen-decomposition of a symmetric 3x3.  On return A ~ V diag(w) V^T with
// w sorted DESCENDING (the order JacobiSVD gives its singular values).
inline void sym_eig3(const Mat3& A, double w[3], Mat3& V) {
  double a[3][3]; std::memcpy(a, A.m, sizeof(a));
  V = mat3_identity();
  for (int sweep = 0; sweep < 32; sweep++) {
    double off = std::fabs(a[0][1]) + std::fabs(a[0][2]) + std::fabs(a[1][2]);
    double diag = std::fabs(a[0][0]) + std::fabs(a[1][1]) + std::fabs(a[2][2]);
    if (off <= 1e-300 || off <= 1e-17 * diag) break;
    for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
      if (a[p][q] == 0.0) continue;
      double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
      double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta*theta + 1.0));
      double c = 1.0 / std::sqrt(t*t + 1.0), s = t * c;
      for (int k = 0; k < 3; k++) { double akp = a[k][p], akq = a[k][q]; a[k][p] = c*akp - s*akq; a[k][q] = s*akp + c*akq; }
      for (int k = 0; k < 3; k++) { double apk = a[p][k], aqk = a[q][k]; a[p][k] = c*apk - s*aqk; a[q][k] = s*apk + c*aqk; }
      for (int k = 0; k < 3; k++) { double vkp = V.m[k][p], vkq = V.m[k][q]; V.m[k][p] = c*vkp - s*vkq; V.m[k][q] = s*vkp + c*vkq; }
    }
  }
  w[0] = a[0][0]; w[1] = a[1][1]; w[2] = a[2][2];
  int idx[3] = {0, 1, 2};
  std::sort(idx, idx + 3, [&](int i, int j) { return w[i] > w[j]; });
  double ws[3]; Mat3 Vs;
  for (int j = 0; j < 3; j++) { ws[j] = w[idx[j]]; for (int i = 0; i < 3; i++) Vs.m[i][j] = V.m[i][idx[j]]; }
  std::memcpy(w, ws, sizeof(ws)); V = Vs;
}

// so3_exp of fast_gicp/nano_gicp: quaternion (real, imag*omega) -> rotation matrix.
inline Mat3 so3_exp(const double om[3]) {
  double theta_sq = om[0]*om[0] + om[1]*om[1] + om[2]*om[2];
  double imag, real;
  if (theta_sq < 1e-10) {
    double theta_quad = theta_sq * theta_sq;
    imag = 0.5 - 1.0/48.0 * theta_sq + 1.0/3840.0 * theta_quad;
    real = 1.0 - 1.0/8.0 * theta_sq + 1.0/384.0 * theta_quad;
  } else {
    double theta = std::sqrt(theta_sq), half = 0.5 * theta;
    imag = std::sin(half) / theta;
    real = std::cos(half);
  }
  double w = real, x = imag*om[0], y = imag*om[1], z = imag*om[2];
  // Eigen::Quaterniond::toRotationMatrix (the quaternion is used as constructed, not re-normalised)
  double tx = 2*x, ty = 2*y, tz = 2*z;
  double twx = tx*w, twy = ty*w, twz = tz*w, txx = tx*x, txy = ty*x, txz = tz*x, tyy = ty*y, tyz = tz*y, tzz = tz*z;
  Mat3 R;
  R.m[0][0] = 1 - (tyy + tzz); R.m[0][1] = txy - twz;       R.m[0][2] = txz + twy;
  R.m[1][0] = txy + twz;       R.m[1][1] = 1 - (txx + tzz); R.m[1][2] = tyz - twx;
  R.m[2][0] = txz - twy;       R.m[2][1] = tyz + twx;       R.m[2][2] = 1 - (txx + tyy);
  return R;
}

// Solve A x = rhs for symmetric 6x6 A by LDL^T with symmetric (diagonal) pivoting,
// the scheme Eigen::LDLT uses (SURVEY A.1.5 step_lm / step_gn).
inline void ldlt_solve6(const double Ain[6][6], const double rhs[6], double x[6]) {
  double A[6][6]; std::memcpy(A, Ain, sizeof(A));
  int perm[6]; for (int i = 0; i < 6; i++) perm[i] = i;
  for (int k = 0; k < 6; k++) {
    int piv = k; double best = std::fabs(A[k][k]);
    for (int i = k + 1; i < 6; i++) if (std::fabs(A[i][i]) > best) { best = std::fabs(A[i][i]); piv = i; }
    if (piv != k) {
      for (int j = 0; j < 6; j++) std::swap(A[k][j], A[piv][j]);
      for (int i = 0; i < 6; i++) std::swap(A[i][k], A[i][piv]);
      std::swap(perm[k], perm[piv]);
    }
    double d = A[k][k];
    if (d == 0.0) continue;
    double l[6];
    for (int i = k + 1; i < 6; i++) l[i] = A[i][k] / d;
    for (int i = k + 1; i < 6; i++)
      for (int j = k + 1; j <= i; j++) { A[i][j] -= l[i] * d * l[j]; A[j][i] = A[i][j]; }
    for (int i = k + 1; i < 6; i++) A[i][k] = l[i];   // L below the diagonal (row k right of it is unused)
  }
  double y[6];
  for (int i = 0; i < 6; i++) { double s = rhs[perm[i]]; for (int j = 0; j < i; j++) s -= A[i][j] * y[j]; y[i] = s; }
  for (int i = 0; i < 6; i++) y[i] = (A[i][i] != 0.0) ? y[i] / A[i][i] : 0.0;
  double z[6];
  for (int i = 5; i >= 0; i--) { double s = y[i]; for (int j = i + 1; j < 6; j++) s -= A[j][i] * z[j]; z[i] = s; }
  for (int i = 0; i < 6; i++) x[perm[i]] = z[i];
}

}  // namespace orc
