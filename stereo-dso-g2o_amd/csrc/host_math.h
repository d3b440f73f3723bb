// Small double-precision math of libsdso_hip.so: SE(3), tiny dense algebra, pivoted LDL^T and the nullspace projector.  Host code
// uses all of it; the functions marked SDSO_HD also run on the device (the Levenberg-Marquardt step of the resident tracker).  This is the product's own implementation of what
// the reference gets from Sophus / Eigen:
//   thirdparty/Sophus/sophus/se3.hpp:131-140 (Adj), :406-428 (exp), :560-600 (log)
//   thirdparty/Sophus/sophus/so3.hpp:343-370, :491-531
//   Eigen ldlt()/inverse() call sites: CoarseTracker.cpp:934, EnergyFunctional.cpp:614, :976
#pragma once
#include <array>
#include <cmath>
#include <cstring>
#include <vector>

#define SDSO_HD __host__ __device__

namespace sdso {

using M3 = std::array<double, 9>;
using V3 = std::array<double, 3>;

struct Se3 {
  M3 R{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
  V3 t{{0, 0, 0}};
};

SDSO_HD inline M3 mul(const M3& a, const M3& b) {
  M3 c;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
  return c;
}
SDSO_HD inline V3 mul(const M3& a, const V3& x) {
  return V3{{a[0] * x[0] + a[1] * x[1] + a[2] * x[2], a[3] * x[0] + a[4] * x[1] + a[5] * x[2],
             a[6] * x[0] + a[7] * x[1] + a[8] * x[2]}};
}
SDSO_HD inline M3 transpose(const M3& a) { return M3{{a[0], a[3], a[6], a[1], a[4], a[7], a[2], a[5], a[8]}}; }
SDSO_HD inline M3 skew(const V3& w) { return M3{{0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0}}; }

SDSO_HD inline Se3 operator*(const Se3& A, const Se3& B) {
  Se3 C;
  C.R = mul(A.R, B.R);
  V3 rt = mul(A.R, B.t);
  for (int i = 0; i < 3; ++i) C.t[i] = rt[i] + A.t[i];
  return C;
}
SDSO_HD inline Se3 inverse(const Se3& A) {
  Se3 C;
  C.R = transpose(A.R);
  V3 v = mul(C.R, A.t);
  for (int i = 0; i < 3; ++i) C.t[i] = -v[i];
  return C;
}

// unit quaternion {w,x,y,z} <-> rotation
SDSO_HD inline M3 rotationFromQuat(double w, double x, double y, double z) {
  const double n = std::sqrt(w * w + x * x + y * y + z * z);
  w /= n; x /= n; y /= n; z /= n;
  return M3{{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
             2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
             2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
}
inline std::array<double, 4> quatFromRotation(const M3& R) {
  const double tr = R[0] + R[4] + R[8];
  double w, x, y, z;
  if (tr > 0) {
    const double s = 2 * std::sqrt(tr + 1.0);
    w = 0.25 * s; x = (R[7] - R[5]) / s; y = (R[2] - R[6]) / s; z = (R[3] - R[1]) / s;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    const double s = 2 * std::sqrt(1.0 + R[0] - R[4] - R[8]);
    w = (R[7] - R[5]) / s; x = 0.25 * s; y = (R[1] + R[3]) / s; z = (R[2] + R[6]) / s;
  } else if (R[4] > R[8]) {
    const double s = 2 * std::sqrt(1.0 + R[4] - R[0] - R[8]);
    w = (R[2] - R[6]) / s; x = (R[1] + R[3]) / s; y = 0.25 * s; z = (R[5] + R[7]) / s;
  } else {
    const double s = 2 * std::sqrt(1.0 + R[8] - R[0] - R[4]);
    w = (R[3] - R[1]) / s; x = (R[2] + R[6]) / s; y = (R[5] + R[7]) / s; z = 0.25 * s;
  }
  return {{w, x, y, z}};
}

constexpr double kSophusEps = 1e-10;

SDSO_HD inline M3 expSo3(const V3& om, double* theta_out) {
  const double th2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
  const double th = std::sqrt(th2);
  double im, re;
  if (th < kSophusEps) {
    const double th4 = th2 * th2;
    im = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
    re = 1.0 - 0.5 * th2 + (1.0 / 384.0) * th4;
  } else {
    im = std::sin(0.5 * th) / th;
    re = std::cos(0.5 * th);
  }
  if (theta_out) *theta_out = th;
  return rotationFromQuat(re, im * om[0], im * om[1], im * om[2]);
}
inline V3 logSo3(const M3& R, double* theta_out) {
  const auto q = quatFromRotation(R);
  const double n2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const double n = std::sqrt(n2);
  const double w = q[0];
  double f;
  if (n < kSophusEps) f = 2.0 / w - 2.0 * n2 / (w * w * w);
  else if (std::fabs(w) < kSophusEps) f = (w > 0 ? M_PI : -M_PI) / n;
  else f = 2.0 * std::atan(n / w) / n;
  if (theta_out) *theta_out = f * n;
  return V3{{f * q[1], f * q[2], f * q[3]}};
}
// tangent = [upsilon | omega] (translation first, Sophus order)
SDSO_HD inline Se3 expSe3(const double* xi) {
  const V3 om{{xi[3], xi[4], xi[5]}};
  Se3 T;
  double th;
  T.R = expSo3(om, &th);
  const M3 Om = skew(om);
  const M3 Om2 = mul(Om, Om);
  M3 V;
  if (th < kSophusEps) {
    V = T.R;
  } else {
    const double a = (1.0 - std::cos(th)) / (th * th);
    const double b = (th - std::sin(th)) / (th * th * th);
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * Om[i] + b * Om2[i];
  }
  T.t = mul(V, V3{{xi[0], xi[1], xi[2]}});
  return T;
}
inline void logSe3(const Se3& T, double* xi) {
  double th;
  const V3 om = logSo3(T.R, &th);
  const M3 Om = skew(om);
  const M3 Om2 = mul(Om, Om);
  double c;
  if (std::fabs(th) < kSophusEps) c = 1.0 / 12.0;
  else c = (1.0 - th / (2.0 * std::tan(th / 2.0))) / (th * th);
  M3 Vi;
  for (int i = 0; i < 9; ++i) Vi[i] = (i % 4 == 0 ? 1.0 : 0.0) - 0.5 * Om[i] + c * Om2[i];
  const V3 u = mul(Vi, T.t);
  xi[0] = u[0]; xi[1] = u[1]; xi[2] = u[2];
  xi[3] = om[0]; xi[4] = om[1]; xi[5] = om[2];
}
// 6x6 row-major adjoint
inline void adjoint(const Se3& T, double* A) {
  const M3 tR = mul(skew(T.t), T.R);
  for (int i = 0; i < 36; ++i) A[i] = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      A[i * 6 + j] = T.R[i * 3 + j];
      A[(i + 3) * 6 + j + 3] = T.R[i * 3 + j];
      A[i * 6 + j + 3] = tR[i * 3 + j];
    }
}

// float 3x3: cofactor inverse (Eigen's fixed 3x3 inverse order), product, mat-vec
SDSO_HD inline void inv3f(const float* m, float* o) {
  const float c00 = m[4] * m[8] - m[5] * m[7];
  const float c10 = m[5] * m[6] - m[3] * m[8];
  const float c20 = m[3] * m[7] - m[4] * m[6];
  const float det = c00 * m[0] + c10 * m[1] + c20 * m[2];
  const float id = 1.0f / det;
  float r[9];
  r[0] = c00 * id; r[3] = c10 * id; r[6] = c20 * id;
  r[1] = (m[2] * m[7] - m[1] * m[8]) * id;
  r[4] = (m[0] * m[8] - m[2] * m[6]) * id;
  r[7] = (m[1] * m[6] - m[0] * m[7]) * id;
  r[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  r[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  r[8] = (m[0] * m[4] - m[1] * m[3]) * id;
  for (int i = 0; i < 9; ++i) o[i] = r[i];
}
SDSO_HD inline void mul3f(const float* a, const float* b, float* c) {
  float r[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float s = a[i * 3] * b[j];
      s = s + a[i * 3 + 1] * b[3 + j];
      s = s + a[i * 3 + 2] * b[6 + j];
      r[i * 3 + j] = s;
    }
  for (int i = 0; i < 9; ++i) c[i] = r[i];
}
SDSO_HD inline void mulv3f(const float* a, const float* x, float* y) {
  float r[3];
  for (int i = 0; i < 3; ++i) {
    float s = a[i * 3] * x[0];
    s = s + a[i * 3 + 1] * x[1];
    s = s + a[i * 3 + 2] * x[2];
    r[i] = s;
  }
  y[0] = r[0]; y[1] = r[1]; y[2] = r[2];
}

// AffLight::fromToVecExposure (src/util/NumType.h:159-170)
SDSO_HD inline void affFromTo(float expF, float expT, double aF, double bF, double aT, double bT, double* out) {
  if (expF == 0 || expT == 0) expT = expF = 1;
  const double a = std::exp(aT - aF) * expT / expF;
  out[0] = a;
  out[1] = bT - a * bF;
}

// dense row-major n x n
struct Dense {
  int n = 0;
  std::vector<double> a;
  Dense() {}
  explicit Dense(int n_) : n(n_), a((size_t)n_ * n_, 0.0) {}
  double& operator()(int i, int j) { return a[(size_t)i * n + j]; }
  double operator()(int i, int j) const { return a[(size_t)i * n + j]; }
};

// x = A^-1 rhs, A symmetric; LDL^T with Eigen's symmetric pivoting: the first largest |diagonal| among the positions not yet
// eliminated, where that diagonal is the ORIGINAL one — Eigen's unblocked LDLT is left-looking and touches a diagonal element only
// at its own step (ba_ldlt.h) — so `d0` carries the input diagonal through the exchanges.  A zero pivot leaves its column undivided.
inline bool solveLdlt(Dense A, const std::vector<double>& rhs, std::vector<double>& x) {
  const int n = A.n;
  std::vector<int> p(n);
  for (int i = 0; i < n; ++i) p[i] = i;
  std::vector<double> D(n, 0.0), d0(n);
  for (int i = 0; i < n; ++i) d0[i] = A(i, i);
  bool ok = true;
  for (int k = 0; k < n; ++k) {
    int piv = k;
    double best = std::fabs(d0[k]);
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(d0[i]) > best) { best = std::fabs(d0[i]); piv = i; }
    if (piv != k) {
      for (int j = 0; j < n; ++j) std::swap(A(k, j), A(piv, j));
      for (int i = 0; i < n; ++i) std::swap(A(i, k), A(i, piv));
      std::swap(p[k], p[piv]);
      std::swap(d0[k], d0[piv]);
    }
    const double d = A(k, k);
    D[k] = d;
    if (d == 0.0) { ok = false; continue; }
    for (int i = k + 1; i < n; ++i) A(i, k) /= d;
    for (int i = k + 1; i < n; ++i) {
      const double l = A(i, k);
      if (l == 0.0) continue;
      for (int j = k + 1; j <= i; ++j) A(i, j) -= l * d * A(j, k);
    }
    for (int i = k + 1; i < n; ++i)
      for (int j = i + 1; j < n; ++j) A(i, j) = A(j, i);
  }
  std::vector<double> y(n);
  for (int i = 0; i < n; ++i) y[i] = rhs[p[i]];
  for (int i = 0; i < n; ++i) { double s = y[i]; for (int j = 0; j < i; ++j) s -= A(i, j) * y[j]; y[i] = s; }
  for (int i = 0; i < n; ++i) y[i] = D[i] != 0.0 ? y[i] / D[i] : 0.0;
  for (int i = n - 1; i >= 0; --i) { double s = y[i]; for (int j = i + 1; j < n; ++j) s -= A(j, i) * y[j]; y[i] = s; }
  x.assign(n, 0.0);
  for (int i = 0; i < n; ++i) x[p[i]] = y[i];
  return ok;
}

// The same algorithm on plain arrays for n <= 8 (row-major A with row stride lda; A is copied): host and device.
// `work` = 80 doubles (the copy of A, D, y), `p` = 8 ints: the caller's storage, so that a device caller can keep them in LDS (dynamically
// indexed locals would live in scratch memory).
SDSO_HD inline bool solveLdltSmall(const double* Ain, int lda, int n, const double* rhs, double* x, double* work, int* p) {
  double* A = work;
  double* D = work + 64;
  double* y = work + 72;          // (also the input diagonal during the factorisation: the pivot search reads it, see solveLdlt)
  for (int i = 0; i < n; ++i) { p[i] = i; D[i] = 0.0; for (int j = 0; j < n; ++j) A[i * 8 + j] = Ain[i * lda + j]; y[i] = Ain[i * lda + i]; }
  bool ok = true;
  for (int k = 0; k < n; ++k) {
    int piv = k;
    double best = fabs(y[k]);
    for (int i = k + 1; i < n; ++i)
      if (fabs(y[i]) > best) { best = fabs(y[i]); piv = i; }
    if (piv != k) {
      for (int j = 0; j < n; ++j) { const double t = A[k * 8 + j]; A[k * 8 + j] = A[piv * 8 + j]; A[piv * 8 + j] = t; }
      for (int i = 0; i < n; ++i) { const double t = A[i * 8 + k]; A[i * 8 + k] = A[i * 8 + piv]; A[i * 8 + piv] = t; }
      const int t = p[k]; p[k] = p[piv]; p[piv] = t;
      const double td = y[k]; y[k] = y[piv]; y[piv] = td;
    }
    const double d = A[k * 8 + k];
    D[k] = d;
    if (d == 0.0) { ok = false; continue; }
    for (int i = k + 1; i < n; ++i) A[i * 8 + k] /= d;
    for (int i = k + 1; i < n; ++i) {
      const double l = A[i * 8 + k];
      if (l == 0.0) continue;
      for (int j = k + 1; j <= i; ++j) A[i * 8 + j] -= l * d * A[j * 8 + k];
    }
    for (int i = k + 1; i < n; ++i)
      for (int j = i + 1; j < n; ++j) A[i * 8 + j] = A[j * 8 + i];
  }
  for (int i = 0; i < n; ++i) y[i] = rhs[p[i]];
  for (int i = 0; i < n; ++i) { double s = y[i]; for (int j = 0; j < i; ++j) s -= A[i * 8 + j] * y[j]; y[i] = s; }
  for (int i = 0; i < n; ++i) y[i] = D[i] != 0.0 ? y[i] / D[i] : 0.0;
  for (int i = n - 1; i >= 0; --i) { double s = y[i]; for (int j = i + 1; j < n; ++j) s -= A[j * 8 + i] * y[j]; y[i] = s; }
  for (int i = 0; i < n; ++i) x[p[i]] = y[i];
  return ok;
}

// A = V diag(w) V^T for symmetric A by cyclic Jacobi rotations (columns of V).  Used where the reference calls
// Eigen::JacobiSVD on the symmetric system matrix (EnergyFunctional.cpp:931): singular values |w|, U = V sign(w).
inline void symEigen(Dense G, std::vector<double>& w, Dense& V) {
  const int n = G.n;
  V = Dense(n);
  for (int i = 0; i < n; ++i) V(i, i) = 1;
  double nrm = 0;
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) nrm += G(i, j) * G(i, j);
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0;
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) off += G(i, j) * G(i, j);
    if (off <= 1e-30 * nrm) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        if (G(p, q) == 0.0) continue;
        const double tau = (G(q, q) - G(p, p)) / (2 * G(p, q));
        const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1 + tau * tau));
        const double c = 1 / std::sqrt(1 + t * t), sn = t * c;
        for (int k = 0; k < n; ++k) { const double a = G(k, p), b = G(k, q); G(k, p) = c * a - sn * b; G(k, q) = sn * a + c * b; }
        for (int k = 0; k < n; ++k) { const double a = G(p, k), b = G(q, k); G(p, k) = c * a - sn * b; G(q, k) = sn * a + c * b; }
        for (int k = 0; k < n; ++k) { const double a = V(k, p), b = V(k, q); V(k, p) = c * a - sn * b; V(k, q) = sn * a + c * b; }
      }
  }
  w.assign(n, 0.0);
  for (int i = 0; i < n; ++i) w[i] = G(i, i);
}

// Projector onto span of the (normalised) columns of N (dim x m, row-major), directions with
// singular value <= delta*max dropped.  = N*pinv(N) (EnergyFunctional.cpp:791-820).
inline Dense spanProjector(const std::vector<double>& N, int dim, int m, double delta) {
  std::vector<double> G((size_t)m * m, 0.0), V((size_t)m * m, 0.0);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j) {
      double s = 0;
      for (int k = 0; k < dim; ++k) s += N[(size_t)k * m + i] * N[(size_t)k * m + j];
      G[i * m + j] = s;
    }
  for (int i = 0; i < m; ++i) V[i * m + i] = 1;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (int i = 0; i < m; ++i) for (int j = i + 1; j < m; ++j) off += G[i * m + j] * G[i * m + j];
    if (off < 1e-300) break;
    for (int p = 0; p < m; ++p)
      for (int q = p + 1; q < m; ++q) {
        if (std::fabs(G[p * m + q]) < 1e-300) continue;
        const double tau = (G[q * m + q] - G[p * m + p]) / (2 * G[p * m + q]);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1 + tau * tau));
        const double c = 1 / std::sqrt(1 + t * t), s = t * c;
        for (int k = 0; k < m; ++k) { const double a = G[k * m + p], b = G[k * m + q]; G[k * m + p] = c * a - s * b; G[k * m + q] = s * a + c * b; }
        for (int k = 0; k < m; ++k) { const double a = G[p * m + k], b = G[q * m + k]; G[p * m + k] = c * a - s * b; G[q * m + k] = s * a + c * b; }
        for (int k = 0; k < m; ++k) { const double a = V[k * m + p], b = V[k * m + q]; V[k * m + p] = c * a - s * b; V[k * m + q] = s * a + c * b; }
      }
  }
  std::vector<double> sv(m);
  double mx = 0;
  for (int i = 0; i < m; ++i) { sv[i] = std::sqrt(std::max(G[i * m + i], 0.0)); mx = std::max(mx, sv[i]); }
  Dense P(dim);
  std::vector<double> u(dim);
  for (int i = 0; i < m; ++i) {
    if (!(sv[i] > delta * mx)) continue;
    for (int k = 0; k < dim; ++k) { double s = 0; for (int j = 0; j < m; ++j) s += N[(size_t)k * m + j] * V[j * m + i]; u[k] = s / sv[i]; }
    for (int a = 0; a < dim; ++a) for (int b = 0; b < dim; ++b) P(a, b) += u[a] * u[b];
  }
  return P;
}

}  // namespace sdso
