// LDL^T of the (8 nf + 4)^2 window system by FOUR waves of one workgroup (256 threads), columns split in four panels of 17.
//
// Same algorithm and pivot order as ba_ldlt.h (Eigen's unblocked LDLT<Lower>, permutation from the input diagonal, computed up
// front), organised so that it needs 34 registers per lane for the matrix instead of 136 — the fused tail kernel has to fit the
// register / LDS footprint of ONE workgroup of the linearisation kernel it runs beside (ba_tail.hip) — and so that the rank-17
// trailing updates of different panels run on different SIMDs:
//   * lane i = row (position) i, wave w holds columns 17 w .. 17 w + 16 of every row: A[17] doubles per lane;
//   * panel b, by wave b: 17 right-looking steps inside its own columns.  Step k parks the column (d_k L_jk, every lane its own element)
//     in the LDS panel U, broadcasts d_k / y_k with v_readlane, writes L_ik into the packed factor and updates the panel's later columns;
//   * trailing update by the waves w > b after a workgroup barrier: A_i[j] -= L_ik * U_k[j] for the 17 steps of the panel and the
//     wave's 17 columns — L_ik is a per-lane LDS read, U_k[j] a uniform one; the forward substitution rides along on every wave
//     (wave 3 ends up with the final y);
//   * rows 64..67 (n = 68) have no lane: by symmetry they are columns 64..67 of the lanes' rows — registers 13..16 of wave 3, whose
//     pivot-row values come from its own lane k by v_readlane; the trailing 4x4 block is finished by wave 3 as in ba_ldlt.h;
//   * backward substitution by wave 3 from the packed factor (lane i reads column i of L: contiguous over the lanes).
// Critical path: 4 panels + 3 updates instead of 64 full-width steps (measured: tools/ldlt_bench.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include "../stereo-dso-g2o_amd/csrc/ba_ldlt.h"

namespace sdso {

constexpr int LP_NB = 17;                          // columns per panel / wave
constexpr int LP_U_DOUBLES = LP_NB * 64;           // the parked columns of one panel
constexpr int LP_L_DOUBLES = 64 * 63 / 2;          // strict lower triangle of the 64 x 64 part of L, packed by rows: L(i,k) at i (i-1)/2 + k
__host__ __device__ constexpr int lp_tri(int i, int k) { return i * (i - 1) / 2 + k; }
// packed LOWER triangle (with diagonal) of the unpermuted, unscaled system: M(i,j), i >= j, at i (i+1)/2 + j
__host__ __device__ constexpr int lp_mtri(int i, int j) { return i * (i + 1) / 2 + j; }
constexpr int LP_M_DOUBLES = LDLT_NMAX * (LDLT_NMAX + 1) / 2;

struct LpShared {                                  // LDS the four waves share besides U / Lp
  double yk[64], dk[64];
};

// one step of panel `b` by its owner wave; KK = step inside the panel (compile time: register index); W3: the wave that owns columns 51..67
template <int KK, bool W3>
__device__ __forceinline__ void lp_panel_step(double (&A)[LP_NB], double& y, int b, int lane, int n, double* __restrict__ U, double* __restrict__ Lp, LpShared& S) {
  const int k = LP_NB * b + KK;                    // (wave-uniform)
  if (k >= n || k >= 64) return;
  const double a = A[KK];
  U[KK * 64 + lane] = a;
  const double dk = ldlt_rl(a, k);
  const double yk = ldlt_rl(y, k);
  if (lane == 0) { S.dk[k] = dk; S.yk[k] = yk; }
  // d_k L_jk of the panel's later columns: lane j's element of column k, just parked in U (uniform reads, all requested before the
  // division); columns 64.. have no lane: lane k's own registers
  double sj[LP_NB];
#pragma unroll
  for (int JJ = KK + 1; JJ < LP_NB; JJ++) sj[JJ] = (W3 && JJ >= 13) ? ldlt_rl(A[JJ], k) : U[KK * 64 + LP_NB * b + JJ];
  double l = a;
  if (dk != 0.0) l = l / dk;
  l = lane > k ? l : 0.0;
  if (lane > k) Lp[lp_tri(lane, k)] = l;
  y = __builtin_fma(-l, yk, y);
#pragma unroll
  for (int JJ = KK + 1; JJ < LP_NB; JJ++) A[JJ] = __builtin_fma(-l, sj[JJ], A[JJ]);
}
template <bool W3, int... KKs>
__device__ __forceinline__ void lp_panel(std::integer_sequence<int, KKs...>, double (&A)[LP_NB], double& y, int b, int lane, int n, double* __restrict__ U, double* __restrict__ Lp, LpShared& S) {
  (lp_panel_step<KKs, W3>(A, y, b, lane, n, U, Lp, S), ...);
}
// the 17 steps of panel `b` applied to the columns of wave w > b: the 17 multipliers of the lane's row first (one round trip), then per
// step the pivot column's 17 values (uniform reads, requested a step ahead by the unrolled code) and 17 FMAs
template <bool W3>
__device__ __forceinline__ void lp_update(double (&A)[LP_NB], double& y, int b, int w, int lane, int n, const double* __restrict__ U, const double* __restrict__ Lp, const LpShared& S) {
  double l[LP_NB], yk[LP_NB];
#pragma unroll
  for (int KK = 0; KK < LP_NB; KK++) {
    const int k = LP_NB * b + KK;
    const bool use = k < n && k < 63 && lane > k;   // (step 63 has no row below it)
    const double lr = Lp[lp_tri(lane > k ? lane : k + 1, k < 63 ? k : 62)];   // (clamped address)
    l[KK] = use ? lr : 0.0;
    yk[KK] = S.yk[k < 64 ? k : 63];
  }
#pragma unroll
  for (int KK = 0; KK < LP_NB; KK++) {
    const int k = LP_NB * b + KK;
    if (k < n && k < 63) {                          // (wave-uniform; the panel's later columns were never parked)
      double sj[LP_NB];
#pragma unroll
      for (int JJ = 0; JJ < LP_NB; JJ++) sj[JJ] = (W3 && JJ >= 13) ? ldlt_rl(A[JJ], k) : U[KK * 64 + LP_NB * w + JJ];
      y = __builtin_fma(-l[KK], yk[KK], y);
#pragma unroll
      for (int JJ = 0; JJ < LP_NB; JJ++) A[JJ] = __builtin_fma(-l[KK], sj[JJ], A[JJ]);
    }
  }
}

// Solve (SVecI M SVecI) x' = SVecI b, permuted by `perm` (perm[p] = original index at position p), for the packed lower triangle Mp of
// the UNSCALED system.  Called by all 256 threads of the workgroup; returns x' by POSITION in xp (LDLT_NMAX doubles, LDS).
//   Mp   : LP_M_DOUBLES (LDS, read only)        sv : SVecI (72)        bs : SVecI * b by ORIGINAL index (72)
//   U    : LP_U_DOUBLES scratch                  Lp : LP_L_DOUBLES scratch (may NOT alias Mp)
// Zero pivots as in ba_ldlt.h.
__device__ __forceinline__ void ldlt_solve_panels(const double* __restrict__ Mp, const double* __restrict__ sv, const int* __restrict__ perm, const double* __restrict__ bs,
                                                  double* __restrict__ U, double* __restrict__ Lp, LpShared& S, double* __restrict__ xp, int n) {
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  double A[LP_NB];
  const int pi = lane < n ? perm[lane] : 0;
#pragma unroll
  for (int JJ = 0; JJ < LP_NB; JJ++) {
    const int j = LP_NB * w + JJ;
    double v = 0.0;
    if (j < n) {
      const int pj = perm[j];
      const int hi = pi > pj ? pi : pj, lo = pi > pj ? pj : pi;
      v = sv[hi] * Mp[lp_mtri(hi, lo)] * sv[lo];   // (sv_i M_ij) sv_j of the LOWER element (Eigen reads the lower triangle): both copies of a pair are the same double
    }
    A[JJ] = lane < n ? v : 0.0;
  }
  double y = lane < n ? bs[pi] : 0.0;
#pragma unroll 1
  for (int b = 0; b < 4; b++) {
    if (w == b) { if (b == 3) lp_panel<true>(std::make_integer_sequence<int, LP_NB>{}, A, y, b, lane, n, U, Lp, S); else lp_panel<false>(std::make_integer_sequence<int, LP_NB>{}, A, y, b, lane, n, U, Lp, S); }
    __syncthreads();
    if (w > b) { if (w == 3) lp_update<true>(A, y, b, w, lane, n, U, Lp, S); else lp_update<false>(A, y, b, w, lane, n, U, Lp, S); }
    __syncthreads();
  }
  if (w != 3) return;
  // ---- wave 3: rows 64..67, D^-1, backward substitution
  const double dmine = lane < n && lane < 64 ? S.dk[lane] : 0.0;
  const bool tail = n > 64;
  double t[4] = {0, 0, 0, 0}, T[4][4], yt[4] = {0, 0, 0, 0}, xt[4] = {0, 0, 0, 0}, dt[4] = {0, 0, 0, 0};
  if (tail) {
#pragma unroll
    for (int r = 0; r < 4; r++) t[r] = dmine != 0.0 ? A[13 + r] / dmine : A[13 + r];       // L(64 + r, lane)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int pr = perm[64 + r];
      yt[r] = bs[pr] - ldlt_wave_sum(t[r] * y);
#pragma unroll
      for (int c = 0; c <= r; c++) {
        const int pc = perm[64 + c];
        const int hi = pr > pc ? pr : pc, lo = pr > pc ? pc : pr;
        const double a0 = sv[hi] * Mp[lp_mtri(hi, lo)] * sv[lo];
        T[r][c] = a0 - ldlt_wave_sum(t[r] * A[13 + c]);
      }
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
  // L^T x = z: lane i needs L(k, i) for k > i — column i of L, one element per row k: the lanes' addresses are contiguous
  constexpr int BT = 16;
#pragma unroll 1
  for (int k0 = 63; k0 >= 1; k0 -= BT) {
    double u[BT];
#pragma unroll
    for (int q = 0; q < BT; q++) {
      const int k = k0 - q;
      u[q] = (k >= 1 && k > lane) ? Lp[lp_tri(k, lane)] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < BT; q++) {
      const int k = k0 - q;
      if (k >= 1 && k < n) xv = __builtin_fma(-u[q], ldlt_rl(xv, k), xv);
    }
  }
  xp[lane] = xv;
  if (lane < 4) xp[64 + lane] = tail ? (lane == 0 ? xt[0] : lane == 1 ? xt[1] : lane == 2 ? xt[2] : xt[3]) : 0.0;
}

}  // namespace sdso
