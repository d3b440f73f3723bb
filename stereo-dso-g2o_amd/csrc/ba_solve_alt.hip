// EnergyFunctional::solveSystemF, the branches the default kernels do not take (paths under /root/reference/src/OptimizationBackend):
//   SOLVER_ORTHOGONALIZE_SYSTEM   EnergyFunctional.cpp:876-900   HT = HL + HA - Hsc, bT = bL + bA - bsc, both projected off the gauge
//                                                                (orthogonalize(&bT, &HT), :775-835) unless the window holds frame 0,
//                                                                then + HM / bM_top and the (1 + lambda) diagonal
//   SOLVER_SVD [| SOLVER_SVD_CUT7] EnergyFunctional.cpp:924-965   Jacobi-scaled system, Eigen::JacobiSVD of the symmetric matrix, singular
//                                                                values below setting_solverModeDelta * max (and, with CUT7, the last
//                                                                seven) dropped
//   and the closing orthogonalize(&x, 0) of SOLVER_ORTHOGONALIZE_X[_LATER]  :980-984
// Rounds 1-3 took the stitched blocks back to the host for these (one round trip per window and iteration, also behind the batch entry
// points): solve_system_host in ba.hip, kept behind SDSO_BA_SOLVE_HOST=1 as the A/B and as the statement of the arithmetic this kernel
// follows line by line.  Here: ONE 512-thread workgroup per window behind k_ba_stitch, everything in LDS (f64).
//   * the symmetric eigen-decomposition that stands in for JacobiSVD (singular values |w|, U = V sign(w)) is a cyclic Jacobi iteration
//     like host_math.h::symEigen, in the PARALLEL ordering: a sweep is n - 1 rounds of n / 2 disjoint index pairs (round-robin
//     tournament), the rotations of a round commute, so a round is three barrier-separated phases — 34 threads compute (c, s) from
//     the untouched 2 x 2 blocks, all threads apply the column rotations to G and V, all threads apply the row rotations to G.  The
//     eigenvectors of well-separated eigenvalues agree with the sequential ordering to rounding; x is a sum over the kept eigenpairs and
//     does not depend on the basis chosen inside an eigenspace.
//   * the LDL^T of the orthogonalised system without SVD is the tail kernel's: pivot rank from the scaled diagonal, register-resident
//     factorisation on wave 0 (ba_ldlt.h).
#include "ba_ldlt.h"

namespace sdso {

constexpr int ALT_NT = 512;
constexpr int ALT_LD = LDLT_LD;   // 70 doubles per row
struct AltLds {
  static constexpr int kMat = LDLT_NMAX * ALT_LD * 8;       // one 68 x 70 f64 matrix
  static constexpr int kHf = 0;                             // the assembled system; then G of the Jacobi iteration / As of the LDL^T
  static constexpr int kV = kHf + kMat;                     // eigenvectors / L^T / P*H of the projection
  static constexpr int kT = kV + kMat;                      // HT of the projection; then the permuted system As
  static constexpr int kVec = kT + kMat;                    // 12 vectors of 72 doubles
  static constexpr int kRot = kVec + 12 * 72 * 8;           // 36 x {c, s} of a round
  static constexpr int kInt = kRot + 36 * 2 * 8;            // pos 72, rank 72, 8 flags
  static constexpr int kKeys = kInt + (72 + 72 + 8) * 4;    // 72 u64
  static constexpr int kCol = kKeys + 72 * 8;               // 64 d (16-byte aligned)
  static constexpr int kBytes = kCol + 64 * 8;
};

// partner of index i in round r of the round-robin tournament over m (even) players: player m - 1 stays, the others rotate
__device__ __forceinline__ void alt_pair(int r, int k, int m, int& p, int& q) {
  const int mm = m - 1;
  if (k == 0) { p = mm; q = r; }
  else { p = (r + k) % mm; q = (r - k + mm) % mm; }
  if (p > q) { const int t = p; p = q; q = t; }
}

__global__ __launch_bounds__(ALT_NT) void k_ba_solve_alt(const BaDev* __restrict__ wins, double lambda, int orthogonalize_x) {
  const BaDev& B = wins[blockIdx.x];
  if (ba_finished(B)) return;
  if (orthogonalize_x & 2) lambda = B.opt->lambda;
  orthogonalize_x &= 1;
  __shared__ __attribute__((aligned(16))) char alt_smem[AltLds::kBytes];
  double* Hf = (double*)(alt_smem + AltLds::kHf);
  double* V = (double*)(alt_smem + AltLds::kV);
  double* T = (double*)(alt_smem + AltLds::kT);
  double* vec = (double*)(alt_smem + AltLds::kVec);
  double *bf = vec, *bMtop = vec + 72, *sv = vec + 144, *bs = vec + 216, *xv = vec + 288, *wv_ = vec + 360, *ub = vec + 432, *tmp = vec + 504, *dg = vec + 576,
         *bp = vec + 648, *xp = vec + 720, *red = vec + 792;
  double* rot = (double*)(alt_smem + AltLds::kRot);
  int* pos = (int*)(alt_smem + AltLds::kInt);
  int* rank = pos + 72;
  unsigned long long* keys = (unsigned long long*)(alt_smem + AltLds::kKeys);
  double* col = (double*)(alt_smem + AltLds::kCol);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = B.n, nf = B.nf, mode = B.solver_mode;
  const size_t blk = (size_t)n * n + n;
  const double* HA = B.sol; const double* bA = HA + (size_t)n * n;
  const double* HL = B.sol + blk; const double* bL = HL + (size_t)n * n;
  const double* HS = B.sol + 2 * blk; const double* bS = HS + (size_t)n * n;
  double* xout = B.sol + 3 * blk;
  double* lastHS = xout + n;
  double* lastbS = lastHS + (size_t)n * n;
  const double* delta = B.t_prior + nf * 16 + 4;
  const double* P = B.t_P;
  const bool orth_sys = (mode & SOLVER_ORTHOGONALIZE_SYSTEM) != 0, svd = (mode & SOLVER_SVD) != 0;

  // bM_top = bM + HM * delta   (:870)
  for (int i = tid; i < 72; i += ALT_NT) {
    double s = 0;
    if (i < n) { for (int k = 0; k < n; k++) s += B.t_HM[(size_t)i * n + k] * delta[k]; s = B.t_bM[i] + s; }
    bMtop[i] = s; bp[i] = 0.0; xp[i] = 0.0;
  }
  for (int e = tid; e < LDLT_NMAX * ALT_LD; e += ALT_NT) { Hf[e] = 0.0; V[e] = 0.0; T[e] = 0.0; }
  __syncthreads();
  if (orth_sys) {
    // HT = HL + HA - Hsc ; bT = bL + bA - bsc   (:878-879)
    for (int e = tid; e < n * n; e += ALT_NT) { const int i = e / n, j = e - i * n; T[i * ALT_LD + j] = HL[e] + HA[e] - HS[e]; }
    for (int i = tid; i < n; i += ALT_NT) tmp[i] = bL[i] + bA[i] - bS[i];
    __syncthreads();
    if (!B.have_first_frame) {   // orthogonalize(&bT, &HT): b -= P b ; H -= P H P   (:822-833 with the window's projector)
      for (int i = tid; i < n; i += ALT_NT) { double s = 0; for (int k = 0; k < n; k++) s += P[(size_t)i * n + k] * tmp[k]; xv[i] = s; }
      for (int e = tid; e < n * n; e += ALT_NT) {
        const int i = e / n, j = e - i * n;
        double s = 0;
        for (int k = 0; k < n; k++) s += P[(size_t)i * n + k] * T[k * ALT_LD + j];
        V[i * ALT_LD + j] = s;                         // P H
      }
      __syncthreads();
      for (int i = tid; i < n; i += ALT_NT) tmp[i] -= xv[i];
      for (int e = tid; e < n * n; e += ALT_NT) {
        const int i = e / n, j = e - i * n;
        double s = 0;
        for (int k = 0; k < n; k++) s += V[i * ALT_LD + k] * P[(size_t)k * n + j];
        Hf[i * ALT_LD + j] = s;                        // P H P (Hf as scratch)
      }
      __syncthreads();
      for (int e = tid; e < n * n; e += ALT_NT) { const int i = e / n, j = e - i * n; T[i * ALT_LD + j] -= Hf[i * ALT_LD + j]; }
      __syncthreads();
    }
    // HFinal_top = HT + HM ; bFinal_top = bT + bM_top ; lastHS / lastbS ; the (1 + lambda) diagonal   (:893-899)
    for (int e = tid; e < n * n; e += ALT_NT) {
      const int i = e / n, j = e - i * n;
      double v = T[i * ALT_LD + j] + B.t_HM[e];
      lastHS[e] = v;
      if (i == j) v *= (1 + lambda);
      Hf[i * ALT_LD + j] = v;
    }
    for (int i = tid; i < n; i += ALT_NT) { const double v = tmp[i] + bMtop[i]; bf[i] = v; lastbS[i] = v; }
  } else {
    // HFinal_top = HL + HM + HA ; bFinal_top = bL + bM_top + bA - b_sc ; lastHS = HFinal_top - H_sc ; diagonal ; -= H_sc / (1 + lambda)   (:906-918)
    const double f = (double)(1.0f / (1 + lambda));
    for (int e = tid; e < n * n; e += ALT_NT) {
      const int i = e / n, j = e - i * n;
      double v = HL[e] + B.t_HM[e] + HA[e];
      lastHS[e] = v - HS[e];
      if (i == j) v *= (1 + lambda);
      v -= HS[e] * f;
      Hf[i * ALT_LD + j] = v;
    }
    for (int i = tid; i < n; i += ALT_NT) { const double v = bL[i] + bMtop[i] + bA[i] - bS[i]; bf[i] = v; lastbS[i] = v; }
  }
  __syncthreads();

  if (svd) {
    // SVecI = (diag)^-1/2, HFinalScaled = SVecI H SVecI, bFinalScaled   (:926-929)
    for (int i = tid; i < 72; i += ALT_NT) { sv[i] = i < n ? 1.0 / sqrt(Hf[i * ALT_LD + i]) : 0.0; xv[i] = 0.0; }
    __syncthreads();
    for (int e = tid; e < n * n; e += ALT_NT) { const int i = e / n, j = e - i * n; T[i * ALT_LD + j] = sv[i] * Hf[i * ALT_LD + j] * sv[j]; }
    for (int i = tid; i < n; i += ALT_NT) bs[i] = sv[i] * bf[i];
    for (int e = tid; e < LDLT_NMAX * ALT_LD; e += ALT_NT) V[e] = 0.0;
    __syncthreads();
    double* G = T;
    for (int i = tid; i < n; i += ALT_NT) V[i * ALT_LD + i] = 1.0;
    // ||G||_F^2 once, the off-diagonal part before every sweep (host_math.h::symEigen: stop at off <= 1e-30 * nrm, 100 sweeps at most)
    auto block_sum = [&](double v) -> double {
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      __syncthreads();
      if (lane == 0) red[wave] = v;
      __syncthreads();
      double s = 0;
      for (int w = 0; w < ALT_NT / 64; w++) s += red[w];
      return s;
    };
    double part = 0;
    for (int e = tid; e < n * n; e += ALT_NT) { const int i = e / n, j = e - i * n; const double g = G[i * ALT_LD + j]; part += g * g; }
    const double nrm = block_sum(part);
    const int m = n;                                   // n = 8 nf + 4 is even
    const int half = m / 2;
    for (int sweep = 0; sweep < 100; ++sweep) {
      part = 0;
      for (int e = tid; e < n * n; e += ALT_NT) { const int i = e / n, j = e - i * n; if (j > i) { const double g = G[i * ALT_LD + j]; part += g * g; } }
      const double off = block_sum(part);
      if (off <= 1e-30 * nrm) break;                   // (uniform: every thread holds the same sum)
      for (int r = 0; r < m - 1; ++r) {
        if (tid < half) {
          int p, q;
          alt_pair(r, tid, m, p, q);
          const double gpq = G[p * ALT_LD + q];
          double c = 1.0, sn = 0.0;
          if (gpq != 0.0) {
            const double tau = (G[q * ALT_LD + q] - G[p * ALT_LD + p]) / (2 * gpq);
            const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1 + tau * tau));
            c = 1 / sqrt(1 + t * t); sn = t * c;
          }
          rot[2 * tid] = c; rot[2 * tid + 1] = sn;
        }
        __syncthreads();
        // columns p, q of G and of V:  (a, b) -> (c a - s b, s a + c b)
        for (int e = tid; e < half * n * 2; e += ALT_NT) {
          const int which = e / (half * n), ee = e - which * half * n, k = ee / n, row = ee - k * n;
          int p, q;
          alt_pair(r, k, m, p, q);
          const double c = rot[2 * k], sn = rot[2 * k + 1];
          double* M = which ? V : G;
          const double a = M[row * ALT_LD + p], b = M[row * ALT_LD + q];
          M[row * ALT_LD + p] = c * a - sn * b;
          M[row * ALT_LD + q] = sn * a + c * b;
        }
        __syncthreads();
        // rows p, q of G
        for (int e = tid; e < half * n; e += ALT_NT) {
          const int k = e / n, cc = e - k * n;
          int p, q;
          alt_pair(r, k, m, p, q);
          const double c = rot[2 * k], sn = rot[2 * k + 1];
          const double a = G[p * ALT_LD + cc], b = G[q * ALT_LD + cc];
          G[p * ALT_LD + cc] = c * a - sn * b;
          G[q * ALT_LD + cc] = sn * a + c * b;
        }
        __syncthreads();
      }
    }
    // singular values |w| in descending order (stable), Ub = U^T b with U = V sign(w), the cuts of :944-956, x = V (Ub / S), x *= SVecI
    if (tid < 72) wv_[tid] = tid < n ? G[tid * ALT_LD + tid] : 0.0;
    __syncthreads();
    if (tid < n) {
      const double wi = fabs(wv_[tid]);
      int rk = 0;
      double mx = 0;
      for (int j = 0; j < n; j++) { const double wj = fabs(wv_[j]); rk += (wj > wi || (wj == wi && j < tid)) ? 1 : 0; mx = wj > mx ? wj : mx; }
      rank[tid] = rk;
      double u = 0;
      for (int k = 0; k < n; k++) u += V[k * ALT_LD + tid] * bs[k];
      if (wv_[tid] < 0) u = -u;
      if (wi < kSolverModeDelta * mx) u = 0;
      if ((mode & SOLVER_SVD_CUT7) && rk >= n - 7) u = 0;
      else u /= wi;
      ub[tid] = u;
      pos[rk] = tid;                                   // pos[i] = the eigenpair at sorted place i
    }
    __syncthreads();
    if (tid < n) {
      double s = 0;
      for (int i = 0; i < n; i++) { const int c = pos[i]; s += V[tid * ALT_LD + c] * ub[c]; }
      xv[tid] = s * sv[tid];
    }
    __syncthreads();
  } else {
    // the LDL^T branch on the orthogonalised system: SVecI = (diag + 10)^-1/2 (:967), Eigen's pivot order, the permuted scaled system
    if (tid < 72) {
      double s = 0, d = 0;
      if (tid < n) { const double mii = Hf[tid * ALT_LD + tid]; s = 1.0 / sqrt(mii + 10); d = s * mii * s; }
      sv[tid] = s; dg[tid] = d;
    }
    for (int e = tid; e < LDLT_NMAX * ALT_LD; e += ALT_NT) T[e] = 0.0;
    __syncthreads();
    ldlt_pivot_rank(dg, n, pos, keys);
    for (int e = tid; e < n * n; e += ALT_NT) {
      const int i = e / n, j = e - i * n;
      if (i >= j) {
        const double v = sv[i] * Hf[i * ALT_LD + j] * sv[j];
        const int pi = pos[i], pj = pos[j];
        T[pi * ALT_LD + pj] = v;
        T[pj * ALT_LD + pi] = v;
      }
    }
    if (tid < n) bp[pos[tid]] = sv[tid] * bf[tid];
    __syncthreads();
    if (wave == 0) ldlt_solve_regs(T, bp, V /* L^T */, col, xp, n);
    __syncthreads();
    if (tid < 72) xv[tid] = tid < n ? sv[tid] * xp[pos[tid]] : 0.0;
    __syncthreads();
  }
  // x -= P x   (:980-984)
  if (orthogonalize_x) {
    if (tid < n) { double s = 0; for (int k = 0; k < n; k++) s += P[(size_t)tid * n + k] * xv[k]; tmp[tid] = xv[tid] - s; }
    __syncthreads();
    if (tid < n) xv[tid] = tmp[tid];
    __syncthreads();
  }
  if (tid < n) xout[tid] = xv[tid];
  // xAd[nf*h+t] = xF(h)^T adHostF[h+nf*t] + xF(t)^T adTargetF[h+nf*t]   (:289-291), float arithmetic, as k_ba_solve leaves it
  float* xAd = const_cast<float*>(B.t_xAd);
  for (int e = tid; e < nf * nf * 8; e += ALT_NT) {
    const int j = e & 7, ht = e >> 3, h = ht / nf, t = ht % nf;
    const double* AH = B.t_adHost + (size_t)(h + nf * t) * 64;
    const double* AT = B.t_adTarget + (size_t)(h + nf * t) * 64;
    float sh = 0, st = 0;
    for (int i = 0; i < 8; i++) { sh += (float)xv[4 + 8 * h + i] * (float)AH[i * 8 + j]; st += (float)xv[4 + 8 * t + i] * (float)AT[i * 8 + j]; }
    xAd[e] = sh + st;
  }
}

}  // namespace sdso
