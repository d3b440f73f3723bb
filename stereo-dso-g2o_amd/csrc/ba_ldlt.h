// Register-resident LDL^T of the (8 nf + 4)^2 window system on ONE wave of gfx950 (n <= 68), with the right-hand side carried along.
//
// Reference: EnergyFunctional::solveSystemF, `HFinalScaled.ldlt().solve(SVecI.asDiagonal() * bFinal_top)`
// (src/OptimizationBackend/EnergyFunctional.cpp:976).  Eigen is not vendored in the reference tree; what is restated is Eigen's
// published unblocked algorithm `ldlt_inplace<Lower>::unblocked`: at step k the pivot is the first largest |diagonal| among positions
// k.., where the diagonal of the not yet eliminated part is the ORIGINAL one (Eigen updates column k and element (k,k) only when
// it reaches step k) — so the symmetric permutation is a function of the input diagonal alone and is computed BEFORE the
// factorisation (ldlt_pivot_rank below), and the factorisation itself runs on the permuted matrix without any search.
//
// Mapping: lane i owns position i — its whole row (both triangles: 68 doubles = 136 VGPRs) sits in registers; the loops over the
// pivot k and the column j are fully unrolled, so every register index is static.  Step k (right-looking):
//     d_k        = A_k[k]                       two v_readlane from lane k
//     l_i        = A_i[k] / d_k   (i > k)       one division per step for all rows at once; 0 on the lanes i <= k
//     A_i[j]    -= l_i * A_j[k]   (j > k)       A_j[k] = d_k L_jk is lane j's own element of column k: the lanes park column k in LDS
//                                               (one ds_write per step) and read it back at uniform addresses, two values per
//                                               ds_read_b128, under the division; no barrier, no exchange of rows
//     y_i       -= l_i * y_k                    forward substitution rides along
// Rows 64..67 (n = 68) have no lane: by symmetry their entries are columns 64..67 of the lanes' rows, so L(64+r, i) = A_i[64+r] / d_i
// falls out of lane i's registers after the loop; the trailing 4x4 block and the last four y are then four-plus-ten wave sums and
// a scalar 4x4 factorisation.  L^T is written to LDS on the way (one ds_write per step) for the backward substitution, where lane
// i reads its own row back and the x_k are broadcast with v_readlane again.
// One wave, ~2300 f64 FMAs + ~4600 readlanes + 64 divisions per lane: measured in profiles/ (tools/ldlt_bench.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <utility>

namespace sdso {

constexpr int LDLT_NMAX = 68;         // 8 * 8 keyframes + 4 calibration parameters
constexpr int LDLT_BATCH = 16;       // pivot-row values in flight (SGPR pairs)
constexpr int LDLT_LD = 70;           // row stride (doubles) of the LDS matrices: 16-byte aligned rows, conflict-free b128 row reads

__device__ __forceinline__ double ldlt_rl(double v, int src) {   // value of lane `src` (compile-time or wave-uniform) in every lane
  const unsigned long long u = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, src), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), src);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double ldlt_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Eigen's pivot order for a symmetric n x n matrix with diagonal `diag` (see the header): pos[i] = position of original index i.
// The order is the descending sort of |diag|; equal values keep their index order.  That is exactly Eigen's order when the ties sit
// in the LARGEST value (nothing has been exchanged before their turn: the window systems scale every diagonal to v / (v + 10), which
// rounds to exactly 1 for the parameters a prior pins) and immaterial when they sit at zero (zero rows: their unknowns are 0 whatever
// the order); ties elsewhere would be ordered by Eigen's earlier exchanges — the factorisations then differ by a permutation among
// equal pivots, i.e. by rounding.  Called by all threads of the workgroup (>= n of them), keys: LDS scratch of 72 u64.
__device__ inline void ldlt_pivot_rank(const double* diag, int n, int* pos, unsigned long long* keys) {
  const int tid = threadIdx.x;
  if (tid < 72) keys[tid] = tid < n ? (unsigned long long)__double_as_longlong(fabs(diag[tid])) : 0ull;   // non-negative doubles order like integers
  __syncthreads();
  if (tid < n) {
    const unsigned long long ki = keys[tid];
    int r = 0;
#pragma unroll 4
    for (int j = 0; j < n; j++) {
      const unsigned long long kj = keys[j];
      r += (kj > ki || (kj == ki && j < tid)) ? 1 : 0;
    }
    pos[tid] = r;
  }
  __syncthreads();
}

// one pivot step with a compile-time k (every register index static; see the header)
template <int K>
__device__ __forceinline__ void ldlt_step(double (&A)[LDLT_NMAX], double& y, double& dmine, double* __restrict__ Lt, double* __restrict__ col, int lane, int n) {
  if (K >= n) return;                              // (wave-uniform)
  col[lane] = A[K];
  const double dk = ldlt_rl(A[K], K);
  const double yk = ldlt_rl(y, K);
  dmine = lane == K ? dk : dmine;
  double l = A[K];
  if (dk != 0.0) l = l / dk;
  l = lane > K ? l : 0.0;                          // rows at or before the pivot are final: a zero multiplier leaves them untouched
  Lt[K * LDLT_LD + lane] = l;
  y = __builtin_fma(-l, yk, y);
  // d_k L_jk for the columns j > k: every lane parked its own element of column k (= row k, by symmetry) in LDS before the
  // division, so the values come back as uniform-address reads while the division runs — half the cross-lane instructions of a
  // v_readlane pair per term.  Columns 64..67 have no lane: lane k's registers, by v_readlane.
#pragma unroll
  for (int j0 = K + 1; j0 < 64; j0 += LDLT_BATCH) {
    double sj[LDLT_BATCH];
#pragma unroll
    for (int u = 0; u < LDLT_BATCH; u++) if (j0 + u < 64) sj[u] = col[j0 + u];
#pragma unroll
    for (int u = 0; u < LDLT_BATCH; u++) if (j0 + u < 64) A[j0 + u] = __builtin_fma(-l, sj[u], A[j0 + u]);
  }
  double st[4];
#pragma unroll
  for (int u = 0; u < 4; u++) st[u] = ldlt_rl(A[64 + u], K);
#pragma unroll
  for (int u = 0; u < 4; u++) A[64 + u] = __builtin_fma(-l, st[u], A[64 + u]);
}
template <int... Ks>
__device__ __forceinline__ void ldlt_steps(std::integer_sequence<int, Ks...>, double (&A)[LDLT_NMAX], double& y, double& dmine, double* __restrict__ Lt, double* __restrict__ col, int lane, int n) {
  (ldlt_step<Ks>(A, y, dmine, Lt, col, lane, n), ...);
}
template <int K>
__device__ __forceinline__ void ldlt_back_step(const double (&A)[LDLT_NMAX], double& xv, int n) {
  if (K < n) xv = __builtin_fma(-A[K], ldlt_rl(xv, K), xv);
}
template <int... Ks>
__device__ __forceinline__ void ldlt_back_steps(std::integer_sequence<int, Ks...>, const double (&A)[LDLT_NMAX], double& xv, int n) {
  (ldlt_back_step<63 - Ks>(A, xv, n), ...);      // k = 63 .. 1
}

// packed layouts of the slim variant (ldlt_solve_regs_packed): the unpermuted, UNSCALED lower triangle with diagonal, M(i,j), i >= j, at
// i (i+1)/2 + j; the strict lower triangle of the 64 x 64 part of L by rows, L(i,k), i > k, at i (i-1)/2 + k
__host__ __device__ constexpr int ldlt_mtri(int i, int j) { return i * (i + 1) / 2 + j; }
__host__ __device__ constexpr int ldlt_ltri(int i, int k) { return i * (i - 1) / 2 + k; }
constexpr int LDLT_M_PACKED = LDLT_NMAX * (LDLT_NMAX + 1) / 2;   // 2346 doubles
constexpr int LDLT_L_PACKED = 64 * 63 / 2;                        // 2016 doubles

// (the pivot row comes from lane K's registers by v_readlane pairs: SGPRs instead of a VGPR batch and no LDS round trip to wait for —
//  the slim variant has 160 registers, and a batch small enough to fit exposes the read latency every four terms: 91 k against 60 k cycles)
template <int K, int BATCH>
__device__ __forceinline__ void ldlt_step_p(double (&A)[LDLT_NMAX], double& y, double& dmine, double* __restrict__ Lp, double* __restrict__ col, int lane, int n) {
  if (K >= n) return;                              // (wave-uniform)
  const double dk = ldlt_rl(A[K], K);
  const double yk = ldlt_rl(y, K);
  dmine = lane == K ? dk : dmine;
  double l = A[K];
  if (dk != 0.0) l = l / dk;
  l = lane > K ? l : 0.0;
  if (lane > K) Lp[ldlt_ltri(lane, K)] = l;
  y = __builtin_fma(-l, yk, y);
#pragma unroll
  for (int j0 = K + 1; j0 < LDLT_NMAX; j0 += BATCH) {
    double sj[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; u++) if (j0 + u < LDLT_NMAX) sj[u] = ldlt_rl(A[j0 + u], K);
#pragma unroll
    for (int u = 0; u < BATCH; u++) if (j0 + u < LDLT_NMAX) A[j0 + u] = __builtin_fma(-l, sj[u], A[j0 + u]);
  }
}
template <int BATCH, int... Ks>
__device__ __forceinline__ void ldlt_steps_p(std::integer_sequence<int, Ks...>, double (&A)[LDLT_NMAX], double& y, double& dmine, double* __restrict__ Lp, double* __restrict__ col, int lane, int n) {
  (ldlt_step_p<Ks, BATCH>(A, y, dmine, Lp, col, lane, n), ...);
}
// The same solve with a third of the LDS and fewer registers, for a kernel that has to fit beside the linearisation's workgroups
// (ba_tail.hip, slim variant): the lane gathers its permuted, scaled row straight from the packed lower triangle Mp of the unscaled
// system (perm[p] = original index at position p, sv = SVecI), the factor goes to the packed array Lp, which ALIASES Mp (every row is
// in registers before the first element of L is written).  bs = SVecI * b by ORIGINAL index; x by position; col: 80 doubles of scratch.
template <int BATCH>
__device__ __forceinline__ void ldlt_solve_regs_packed(double* __restrict__ MpLp, const double* __restrict__ sv, const int* __restrict__ perm, const double* __restrict__ bs,
                                                       double* __restrict__ col, double* __restrict__ x, int n) {
  const int lane = threadIdx.x & 63;
  double A[LDLT_NMAX];
  const int pi = lane < n ? perm[lane] : 0;
#pragma unroll
  for (int j = 0; j < LDLT_NMAX; j++) {
    double v = 0.0;
    if (j < n) {                                   // (wave-uniform)
      const int pj = perm[j];
      const int hi = pi > pj ? pi : pj, lo = pi > pj ? pj : pi;
      v = sv[hi] * MpLp[ldlt_mtri(hi, lo)] * sv[lo];     // the LOWER element (Eigen reads the lower triangle): both copies of a pair are the same double
    }
    A[j] = lane < n ? v : 0.0;
  }
  const bool tail = n > 64;
  // the trailing 4 x 4 block and the last four right-hand sides wait in `col[64..77]` (not in registers) until the lanes are through
  if (tail && lane < 14) {
    int r = 0, c = 0;                              // lane -> (r, c <= r) of the block, lanes 10..13 -> rhs r
    if (lane < 10) { r = lane < 1 ? 0 : lane < 3 ? 1 : lane < 6 ? 2 : 3; c = lane - (r * (r + 1)) / 2; }
    else r = lane - 10;
    const int pr = perm[64 + r], pc = perm[64 + c];
    const int hi = pr > pc ? pr : pc, lo = pr > pc ? pc : pr;
    col[64 + lane] = lane < 10 ? sv[hi] * MpLp[ldlt_mtri(hi, lo)] * sv[lo] : bs[pr];
  }
  double y = lane < n ? bs[pi] : 0.0;
  double dmine = 0.0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // Mp is dead from here on
  ldlt_steps_p<BATCH>(std::make_integer_sequence<int, 64>{}, A, y, dmine, MpLp, col, lane, n);
  double t[4] = {0, 0, 0, 0}, xt[4] = {0, 0, 0, 0}, dt[4] = {0, 0, 0, 0}, T[4][4], yt[4] = {0, 0, 0, 0};
  if (tail) {
#pragma unroll
    for (int r = 0; r < 4; r++) t[r] = dmine != 0.0 ? A[64 + r] / dmine : A[64 + r];       // L(64 + r, lane)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      yt[r] = col[64 + 10 + r] - ldlt_wave_sum(t[r] * y);
#pragma unroll
      for (int c = 0; c <= r; c++) T[r][c] = col[64 + (r * (r + 1)) / 2 + c] - ldlt_wave_sum(t[r] * A[64 + c]);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const double dk = T[k][k];
      dt[k] = dk;
      double s[4], l[4];
#pragma unroll
      for (int i = k + 1; i < 4; i++) { s[i] = T[i][k]; l[i] = dk != 0.0 ? s[i] / dk : s[i]; T[i][k] = l[i]; }
#pragma unroll
      for (int i = k + 1; i < 4; i++) {
        yt[i] = __builtin_fma(-l[i], yt[k], yt[i]);
#pragma unroll
        for (int j = k + 1; j <= i; j++) T[i][j] = __builtin_fma(-l[i], s[j], T[i][j]);
      }
    }
#pragma unroll
    for (int k = 3; k >= 0; k--) {
      double v = dt[k] != 0.0 ? yt[k] / dt[k] : 0.0;
#pragma unroll
      for (int i = k + 1; i < 4; i++) v = __builtin_fma(-T[i][k], xt[i], v);
      xt[k] = v;
    }
  }
  double xv = dmine != 0.0 ? y / dmine : 0.0;
  if (tail) {
#pragma unroll
    for (int r = 3; r >= 0; r--) xv = __builtin_fma(-t[r], xt[r], xv);
  }
  // L^T x = z: lane i needs L(k, i) for k > i — column i of L, one element per row k (the lanes' addresses are contiguous)
  constexpr int BT = 16;
#pragma unroll 1
  for (int k0 = 63; k0 >= 1; k0 -= BT) {
    double u[BT];
#pragma unroll
    for (int q = 0; q < BT; q++) {
      const int k = k0 - q;
      u[q] = (k >= 1 && k > lane) ? MpLp[ldlt_ltri(k, lane)] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < BT; q++) {
      const int k = k0 - q;
      if (k >= 1 && k < n) xv = __builtin_fma(-u[q], ldlt_rl(xv, k), xv);
    }
  }
  x[lane] = xv;
  if (lane < 4) x[64 + lane] = tail ? (lane == 0 ? xt[0] : lane == 1 ? xt[1] : lane == 2 ? xt[2] : xt[3]) : 0.0;
}

// Solve As x = b for the PERMUTED, scaled system in LDS.  Called by one full wave (all 64 lanes).
//   As   : LDLT_NMAX rows x LDLT_LD, symmetric (both triangles filled), zero outside n x n
//   b    : LDLT_NMAX, zero beyond n                x : LDLT_NMAX out (positions)
//   Lt   : 64 rows x LDLT_LD scratch (receives L^T; need not be initialised)
//   col  : 64 doubles scratch (16-byte aligned)
// A zero pivot leaves its column undivided and its unknown 0, as Eigen does (LDLT.h: `if (rs > 0 && pivot_is_valid) A21 /= realAkk`,
// and the solve's `dst.row(i).setZero()` for a zero D).
__device__ __forceinline__ void ldlt_solve_regs(const double* __restrict__ As, const double* __restrict__ b, double* __restrict__ Lt, double* __restrict__ col, double* __restrict__ x, int n) {
  const int lane = threadIdx.x & 63;
  double A[LDLT_NMAX];
  {
    const double* row = As + lane * LDLT_LD;
#pragma unroll
    for (int j = 0; j < LDLT_NMAX; j++) A[j] = row[j];
  }
  double y = b[lane];
  double dmine = 0.0;
  ldlt_steps(std::make_integer_sequence<int, 64>{}, A, y, dmine, Lt, col, lane, n);
  const bool tail = n > 64;                        // positions 64..67 (n = 68)
  double t[4] = {0, 0, 0, 0}, T[4][4], yt[4] = {0, 0, 0, 0}, xt[4] = {0, 0, 0, 0}, dt[4] = {0, 0, 0, 0};
  if (tail) {
#pragma unroll
    for (int r = 0; r < 4; r++) t[r] = dmine != 0.0 ? A[64 + r] / dmine : A[64 + r];       // L(64 + r, lane)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      yt[r] = b[64 + r] - ldlt_wave_sum(t[r] * y);
#pragma unroll
      for (int c = 0; c <= r; c++) T[r][c] = As[(64 + r) * LDLT_LD + 64 + c] - ldlt_wave_sum(t[r] * A[64 + c]);
    }
    // the last four pivots, every lane redundantly (uniform values); same update form as above: T_ij -= l_i * (d_k l_j)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const double dk = T[k][k];
      dt[k] = dk;
      double s[4], l[4];
#pragma unroll
      for (int i = k + 1; i < 4; i++) { s[i] = T[i][k]; l[i] = dk != 0.0 ? s[i] / dk : s[i]; T[i][k] = l[i]; }
#pragma unroll
      for (int i = k + 1; i < 4; i++) {
        yt[i] = __builtin_fma(-l[i], yt[k], yt[i]);
#pragma unroll
        for (int j = k + 1; j <= i; j++) T[i][j] = __builtin_fma(-l[i], s[j], T[i][j]);
      }
    }
#pragma unroll
    for (int k = 3; k >= 0; k--) {
      double v = dt[k] != 0.0 ? yt[k] / dt[k] : 0.0;
#pragma unroll
      for (int i = k + 1; i < 4; i++) v = __builtin_fma(-T[i][k], xt[i], v);
      xt[k] = v;
    }
  }
  // D^-1, then L^T x = z: lane i reads row i of L^T (zeros at and before its own column), the x_k are broadcast from high to low
  double xv = dmine != 0.0 ? y / dmine : 0.0;
  if (tail) {
#pragma unroll
    for (int r = 3; r >= 0; r--) xv = __builtin_fma(-t[r], xt[r], xv);
  }
  {
    const double* row = Lt + lane * LDLT_LD;
#pragma unroll
    for (int j = 0; j < 64; j++) A[j] = row[j];
  }
  ldlt_back_steps(std::make_integer_sequence<int, 63>{}, A, xv, n);
  x[lane] = xv;
  if (lane < 4) x[64 + lane] = tail ? (lane == 0 ? xt[0] : lane == 1 ? xt[1] : lane == 2 ? xt[2] : xt[3]) : 0.0;
}

}  // namespace sdso
