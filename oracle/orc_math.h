// ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's small math.
// Nothing under oracle/ is part of the shipped product; only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load it.
//
// Restates (reference paths relative to /root/reference):
//   thirdparty/Sophus/sophus/se3.hpp:131-140 (Adj), :406-428 (exp), :560-600 (log)
//   thirdparty/Sophus/sophus/so3.hpp:343-370 (expAndTheta), :491-531 (logAndTheta)
//   thirdparty/Sophus/sophus/sophus.hpp:45-59 (epsilon)
// Eigen (ldlt, inverse) is an external, un-pinned dependency of the reference; the
// LDLT below follows the published algorithm (symmetric pivoting on the largest |diagonal|).
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>

namespace orc {

// ---------------------------------------------------------------- fixed-size helpers (row-major)
template <typename T>
inline void mat3_mul(const T* A, const T* B, T* C) {  // C = A*B, all 3x3 row-major
  T r[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      T s = A[i * 3 + 0] * B[0 * 3 + j];
      s = s + A[i * 3 + 1] * B[1 * 3 + j];
      s = s + A[i * 3 + 2] * B[2 * 3 + j];
      r[i * 3 + j] = s;
    }
  std::memcpy(C, r, sizeof(r));
}
template <typename T>
inline void mat3_vec(const T* A, const T* x, T* y) {
  T r[3];
  for (int i = 0; i < 3; i++) {
    T s = A[i * 3 + 0] * x[0];
    s = s + A[i * 3 + 1] * x[1];
    s = s + A[i * 3 + 2] * x[2];
    r[i] = s;
  }
  y[0] = r[0]; y[1] = r[1]; y[2] = r[2];
}
// 3x3 inverse by cofactors * (1/det), the order Eigen's fixed-size 3x3 inverse uses.
template <typename T>
inline void mat3_inv(const T* m, T* inv) {
  T c00 = m[4] * m[8] - m[5] * m[7];
  T c10 = m[5] * m[6] - m[3] * m[8];
  T c20 = m[3] * m[7] - m[4] * m[6];
  T det = c00 * m[0] + c10 * m[1] + c20 * m[2];
  T id = T(1) / det;
  T r[9];
  r[0] = c00 * id;
  r[3] = c10 * id;
  r[6] = c20 * id;
  r[1] = (m[2] * m[7] - m[1] * m[8]) * id;
  r[4] = (m[0] * m[8] - m[2] * m[6]) * id;
  r[7] = (m[1] * m[6] - m[0] * m[7]) * id;
  r[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  r[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  r[8] = (m[0] * m[4] - m[1] * m[3]) * id;
  std::memcpy(inv, r, sizeof(r));
}

// ---------------------------------------------------------------- SE3 (double): R row-major + t
struct SE3 {
  double R[9];
  double t[3];
  SE3() { setIdentity(); }
  void setIdentity() {
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    t[0] = t[1] = t[2] = 0;
  }
};

inline void hat3(const double* w, double* O) {
  O[0] = 0;     O[1] = -w[2]; O[2] = w[1];
  O[3] = w[2];  O[4] = 0;     O[5] = -w[0];
  O[6] = -w[1]; O[7] = w[0];  O[8] = 0;
}

inline SE3 se3_mul(const SE3& A, const SE3& B) {
  SE3 C;
  mat3_mul(A.R, B.R, C.R);
  double Rt[3];
  mat3_vec(A.R, B.t, Rt);
  for (int i = 0; i < 3; i++) C.t[i] = Rt[i] + A.t[i];
  return C;
}
inline SE3 se3_inv(const SE3& A) {
  SE3 C;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C.R[i * 3 + j] = A.R[j * 3 + i];
  double v[3];
  mat3_vec(C.R, A.t, v);
  for (int i = 0; i < 3; i++) C.t[i] = -v[i];
  return C;
}

// quaternion (w,x,y,z) -> rotation matrix
inline void quat_to_R(double w, double x, double y, double z, double* R) {
  double n = std::sqrt(w * w + x * x + y * y + z * z);
  w /= n; x /= n; y /= n; z /= n;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}
// rotation matrix -> unit quaternion (w>=0 branchless-robust form)
inline void R_to_quat(const double* R, double* q) {
  double tr = R[0] + R[4] + R[8];
  double w, x, y, z;
  if (tr > 0) {
    double s = std::sqrt(tr + 1.0) * 2;
    w = 0.25 * s; x = (R[7] - R[5]) / s; y = (R[2] - R[6]) / s; z = (R[3] - R[1]) / s;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    double s = std::sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
    w = (R[7] - R[5]) / s; x = 0.25 * s; y = (R[1] + R[3]) / s; z = (R[2] + R[6]) / s;
  } else if (R[4] > R[8]) {
    double s = std::sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
    w = (R[2] - R[6]) / s; x = (R[1] + R[3]) / s; y = 0.25 * s; z = (R[5] + R[7]) / s;
  } else {
    double s = std::sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
    w = (R[3] - R[1]) / s; x = (R[2] + R[6]) / s; y = (R[5] + R[7]) / s; z = 0.25 * s;
  }
  q[0] = w; q[1] = x; q[2] = y; q[3] = z;
}

// so3.hpp:343-370
inline void so3_exp(const double* omega, double* R, double* theta_out) {
  const double eps = 1e-10;
  double theta_sq = omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2];
  double theta = std::sqrt(theta_sq);
  double half_theta = 0.5 * theta;
  double imag_factor, real_factor;
  if (theta < eps) {
    double theta_po4 = theta_sq * theta_sq;
    imag_factor = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
    real_factor = 1.0 - 0.5 * theta_sq + (1.0 / 384.0) * theta_po4;
  } else {
    double sin_half_theta = std::sin(half_theta);
    imag_factor = sin_half_theta / theta;
    real_factor = std::cos(half_theta);
  }
  quat_to_R(real_factor, imag_factor * omega[0], imag_factor * omega[1], imag_factor * omega[2], R);
  if (theta_out) *theta_out = theta;
}

// so3.hpp:491-531
inline void so3_log(const double* R, double* omega, double* theta_out) {
  const double eps = 1e-10;
  double q[4];
  R_to_quat(R, q);
  double squared_n = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  double n = std::sqrt(squared_n);
  double w = q[0];
  double two_atan_nbyw_by_n;
  if (n < eps) {
    double squared_w = w * w;
    two_atan_nbyw_by_n = 2.0 / w - 2.0 * (squared_n) / (w * squared_w);
  } else {
    if (std::fabs(w) < eps) {
      two_atan_nbyw_by_n = (w > 0 ? M_PI : -M_PI) / n;
    } else {
      two_atan_nbyw_by_n = 2.0 * std::atan(n / w) / n;
    }
  }
  if (theta_out) *theta_out = two_atan_nbyw_by_n * n;
  for (int i = 0; i < 3; i++) omega[i] = two_atan_nbyw_by_n * q[1 + i];
}

// se3.hpp:406-428 ; tangent = [upsilon(3) | omega(3)]
inline SE3 se3_exp(const double* a) {
  const double eps = 1e-10;
  const double* omega = a + 3;
  SE3 T;
  double theta;
  so3_exp(omega, T.R, &theta);
  double Om[9], Om2[9], V[9];
  hat3(omega, Om);
  mat3_mul(Om, Om, Om2);
  if (theta < eps) {
    std::memcpy(V, T.R, sizeof(V));
  } else {
    double theta_sq = theta * theta;
    double c1 = (1.0 - std::cos(theta)) / theta_sq;
    double c2 = (theta - std::sin(theta)) / (theta_sq * theta);
    for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c1 * Om[i] + c2 * Om2[i];
  }
  mat3_vec(V, a, T.t);
  return T;
}

// se3.hpp:560-600
inline void se3_log(const SE3& T, double* out) {
  const double eps = 1e-10;
  double theta;
  so3_log(T.R, out + 3, &theta);
  double Om[9], Om2[9], Vi[9];
  hat3(out + 3, Om);
  mat3_mul(Om, Om, Om2);
  if (std::fabs(theta) < eps) {
    for (int i = 0; i < 9; i++) Vi[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * Om[i] + (1. / 12.) * Om2[i];
  } else {
    double c = (1.0 - theta / (2.0 * std::tan(theta / 2.0))) / (theta * theta);
    for (int i = 0; i < 9; i++) Vi[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * Om[i] + c * Om2[i];
  }
  mat3_vec(Vi, T.t, out);
}

// se3.hpp:131-140 ; 6x6 row-major
inline void se3_adj(const SE3& T, double* A) {
  double th[9], thR[9];
  hat3(T.t, th);
  mat3_mul(th, T.R, thR);
  for (int i = 0; i < 36; i++) A[i] = 0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      A[i * 6 + j] = T.R[i * 3 + j];
      A[(i + 3) * 6 + (j + 3)] = T.R[i * 3 + j];
      A[i * 6 + (j + 3)] = thR[i * 3 + j];
    }
}

// ---------------------------------------------------------------- dynamic dense (double, row-major)
struct MatX {
  int r = 0, c = 0;
  std::vector<double> d;
  MatX() {}
  MatX(int r_, int c_) : r(r_), c(c_), d((size_t)r_ * c_, 0.0) {}
  double& operator()(int i, int j) { return d[(size_t)i * c + j]; }
  double operator()(int i, int j) const { return d[(size_t)i * c + j]; }
  void setZero() { std::fill(d.begin(), d.end(), 0.0); }
};
typedef std::vector<double> VecX;

// Solve A x = rhs for symmetric A (n x n) with Eigen's LDLT (`A.ldlt().solve(rhs)`, used at EnergyFunctional.cpp:976 and
// CoarseTracker.cpp:934).  Eigen is an external, un-pinned dependency of the reference (absent from /root/reference); what follows
// restates its published unblocked algorithm, `internal::ldlt_inplace<Lower>::unblocked` (Eigen/src/Cholesky/LDLT.h, 3.2 .. 3.4):
//   for k = 0 .. n-1
//     p = index of the FIRST largest |mat(i,i)|, i >= k            (`mat.diagonal().tail(size-k).cwiseAbs().maxCoeff(&idx)`)
//     exchange positions k and p symmetrically (only the lower triangle is ever read)
//     temp(q) = D(q) * L(k,q), q < k;   mat(k,k) -= L(k,:) . temp;   mat(i,k) -= L(i,:) . temp  for i > k      (left-looking)
//     if |mat(k,k)| > 0:  mat(i,k) /= mat(k,k)  for i > k         (a zero pivot leaves its column as it is)
// The trailing block is NOT updated before its turn, so the diagonal the pivot search sees at positions > k is the ORIGINAL one:
// the permutation depends on the input diagonal only.  The solve (`LDLT::_solve_impl`): P b, L^-1, D^-1 with 0 for a zero D
// (Eigen's tolerance there is the smallest positive double), L^-T, P^T.
// Returns false if a zero pivot was met.
inline bool ldlt_solve(const MatX& Ain, const VecX& rhs, VecX& x) {
  const int n = Ain.r;
  std::vector<int> perm(n);
  for (int i = 0; i < n; i++) perm[i] = i;
  MatX L(n, n);                        // unit lower factor by POSITION (strict lower part)
  std::vector<double> D(n, 0.0), temp(n, 0.0);
  auto a = [&](int pi, int pj) {       // element of the permuted input, read from the lower triangle of the original
    const int i = perm[pi], j = perm[pj];
    return i >= j ? Ain(i, j) : Ain(j, i);
  };
  bool ok = true;
  for (int k = 0; k < n; k++) {
    int p = k;
    double best = std::fabs(Ain(perm[k], perm[k]));
    for (int i = k + 1; i < n; i++) {
      const double v = std::fabs(Ain(perm[i], perm[i]));
      if (v > best) { best = v; p = i; }
    }
    if (p != k) {
      std::swap(perm[k], perm[p]);
      for (int q = 0; q < k; q++) std::swap(L(k, q), L(p, q));
    }
    for (int q = 0; q < k; q++) temp[q] = D[q] * L(k, q);
    double dk = a(k, k);
    { double s = 0; for (int q = 0; q < k; q++) s += L(k, q) * temp[q]; dk -= s; }
    D[k] = dk;
    const bool valid = std::fabs(dk) > 0.0;
    if (!valid) ok = false;
    for (int i = k + 1; i < n; i++) {
      double c = a(i, k);
      double s = 0;
      for (int q = 0; q < k; q++) s += L(i, q) * temp[q];
      c -= s;
      L(i, k) = valid ? c / dk : c;
    }
  }
  VecX y(n);
  for (int i = 0; i < n; i++) y[i] = rhs[perm[i]];
  for (int i = 0; i < n; i++) {
    double s = y[i];
    for (int j = 0; j < i; j++) s -= L(i, j) * y[j];
    y[i] = s;
  }
  for (int i = 0; i < n; i++) y[i] = (D[i] != 0.0) ? y[i] / D[i] : 0.0;
  for (int i = n - 1; i >= 0; i--) {
    double s = y[i];
    for (int j = i + 1; j < n; j++) s -= L(j, i) * y[j];
    y[i] = s;
  }
  x.assign(n, 0.0);
  for (int i = 0; i < n; i++) x[perm[i]] = y[i];
  return ok;
}

// inverse of a small symmetric-ish matrix via Gauss-Jordan with partial pivoting (Eigen's
// general inverse() for 8x8 is PartialPivLU; EnergyFunctional.cpp:614).
inline void mat_inverse(const MatX& Ain, MatX& inv) {
  int n = Ain.r;
  MatX A = Ain;
  inv = MatX(n, n);
  for (int i = 0; i < n; i++) inv(i, i) = 1;
  for (int k = 0; k < n; k++) {
    int p = k;
    for (int i = k + 1; i < n; i++)
      if (std::fabs(A(i, k)) > std::fabs(A(p, k))) p = i;
    if (p != k)
      for (int j = 0; j < n; j++) { std::swap(A(k, j), A(p, j)); std::swap(inv(k, j), inv(p, j)); }
    double d = A(k, k);
    for (int j = 0; j < n; j++) { A(k, j) /= d; inv(k, j) /= d; }
    for (int i = 0; i < n; i++) {
      if (i == k) continue;
      double f = A(i, k);
      if (f == 0) continue;
      for (int j = 0; j < n; j++) { A(i, j) -= f * A(k, j); inv(i, j) -= f * inv(k, j); }
    }
  }
}

// Orthogonal projector onto span(N) (N: dim x m, columns already normalised), computed through the
// eigen-decomposition of N^T N (cyclic Jacobi), dropping directions whose singular value is
// <= delta * max singular value.  Equals N * pinv(N) as built by EnergyFunctional.cpp:791-820
// (Eigen::JacobiSVD is external/un-pinned there).
inline void span_projector(const MatX& N, double delta, MatX& P) {
  int dim = N.r, m = N.c;
  MatX G(m, m), V(m, m);
  for (int i = 0; i < m; i++)
    for (int j = 0; j < m; j++) {
      double s = 0;
      for (int k = 0; k < dim; k++) s += N(k, i) * N(k, j);
      G(i, j) = s;
    }
  for (int i = 0; i < m; i++) V(i, i) = 1;
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0;
    for (int i = 0; i < m; i++)
      for (int j = i + 1; j < m; j++) off += G(i, j) * G(i, j);
    if (off < 1e-300) break;
    for (int p = 0; p < m; p++)
      for (int q = p + 1; q < m; q++) {
        if (std::fabs(G(p, q)) < 1e-300) continue;
        double tau = (G(q, q) - G(p, p)) / (2 * G(p, q));
        double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1 + tau * tau));
        double c = 1 / std::sqrt(1 + t * t), s = t * c;
        for (int k = 0; k < m; k++) {
          double gkp = G(k, p), gkq = G(k, q);
          G(k, p) = c * gkp - s * gkq;
          G(k, q) = s * gkp + c * gkq;
        }
        for (int k = 0; k < m; k++) {
          double gpk = G(p, k), gqk = G(q, k);
          G(p, k) = c * gpk - s * gqk;
          G(q, k) = s * gpk + c * gqk;
        }
        for (int k = 0; k < m; k++) {
          double vkp = V(k, p), vkq = V(k, q);
          V(k, p) = c * vkp - s * vkq;
          V(k, q) = s * vkp + c * vkq;
        }
      }
  }
  double maxSv = 0;
  std::vector<double> sv(m);
  for (int i = 0; i < m; i++) { sv[i] = std::sqrt(std::max(G(i, i), 0.0)); maxSv = std::max(maxSv, sv[i]); }
  // U_i = N V_i / sv_i ; P = sum_i U_i U_i^T over kept i
  P = MatX(dim, dim);
  std::vector<double> u(dim);
  for (int i = 0; i < m; i++) {
    if (!(sv[i] > delta * maxSv)) continue;
    for (int k = 0; k < dim; k++) {
      double s = 0;
      for (int j = 0; j < m; j++) s += N(k, j) * V(j, i);
      u[k] = s / sv[i];
    }
    for (int a = 0; a < dim; a++)
      for (int b = 0; b < dim; b++) P(a, b) += u[a] * u[b];
  }
}

// Eigen-decomposition of a symmetric matrix by cyclic Jacobi rotations: A = V diag(w) V^T (columns of V).  Stands in for
// Eigen::JacobiSVD on the symmetric system matrix of EnergyFunctional.cpp:931 (singular values = |w|, U = V sign(w)).
inline void sym_eigen(const MatX& Ain, VecX& w, MatX& V) {
  const int n = Ain.r;
  MatX G = Ain;
  V = MatX(n, n);
  for (int i = 0; i < n; i++) V(i, i) = 1;
  double nrm = 0;
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) nrm += G(i, j) * G(i, j);
  for (int sweep = 0; sweep < 100; sweep++) {
    double off = 0;
    for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) off += G(i, j) * G(i, j);
    if (off <= 1e-30 * nrm) break;
    for (int p = 0; p < n; p++)
      for (int q = p + 1; q < n; q++) {
        if (G(p, q) == 0.0) continue;
        const double tau = (G(q, q) - G(p, p)) / (2 * G(p, q));
        const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1 + tau * tau));
        const double c = 1 / std::sqrt(1 + t * t), sn = t * c;
        for (int k = 0; k < n; k++) { const double a = G(k, p), b = G(k, q); G(k, p) = c * a - sn * b; G(k, q) = sn * a + c * b; }
        for (int k = 0; k < n; k++) { const double a = G(p, k), b = G(q, k); G(p, k) = c * a - sn * b; G(q, k) = sn * a + c * b; }
        for (int k = 0; k < n; k++) { const double a = V(k, p), b = V(k, q); V(k, p) = c * a - sn * b; V(k, q) = sn * a + c * b; }
      }
  }
  w.assign(n, 0.0);
  for (int i = 0; i < n; i++) w[i] = G(i, i);
}

}  // namespace orc
