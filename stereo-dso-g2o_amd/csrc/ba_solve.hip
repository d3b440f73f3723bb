// Second half of the windowed-BA iteration on gfx950: fold of the Schur partials, the double
// precision stitch into the (8nf+4)^2 systems, the damped pivoted-LDLT solve and the
// back-substitution.  Reference (paths under /root/reference):
//   AccumulatedTopHessianSSE::stitchDoubleInternal / stitchDoubleMT   AccumulatedTopHessian.cpp:265-337, .h:95-148
//   AccumulatedSCHessianSSE::stitchDoubleInternal / stitchDoubleMT    AccumulatedSCHessian.cpp:106-195, .h:96-135
//   EnergyFunctional::solveSystemF (default LDLT branch)              EnergyFunctional.cpp:838-995
//   EnergyFunctional::resubstituteF_MT / resubstituteFPt             EnergyFunctional.cpp:272-341
// The dense work here is tiny (<= 68x68): it is organised "owner computes" — one wave per 8x8
// output tile, fixed summation order, no atomics — so results are run-to-run reproducible.
#include "ba_kernels.h"

namespace sdso {

// ------------------------------------------------------------------ folds
// Hcc / bc after k_ba_sc_host: nf per-host partials of 20 floats
__device__ __forceinline__ void fold_hcc_hosts(const BaDev& B, int lane) {
  if (lane < 20) {
    float s = 0;
    for (int h = 0; h < B.nf; h++) s += B.sc_part[(size_t)h * 20 + lane];
    B.accum[acc_off_Hcc(B.nf) + lane] = s;
  }
}
__global__ __launch_bounds__(64) void k_ba_fold_hcc(const BaDev* __restrict__ wins) { if (ba_finished(wins[blockIdx.y])) return; fold_hcc_hosts(wins[blockIdx.y], threadIdx.x); }
// every fold of one accumulate phase in ONE launch (the usual case: no linearized residuals, topL is just cleared):
// grid.x = [1: Hcc / bc of the hosts' Schur workgroups | nf^2 top-A pairs | nf^2 top-L pairs], 128 threads
__global__ __launch_bounds__(128) void k_ba_fold_all(const BaDev* __restrict__ wins) {
  const BaDev& B = wins[blockIdx.y];
  if (ba_finished(B)) return;
  const int nf2 = B.nf * B.nf;
  const int b = blockIdx.x;
  if (b < 1) fold_hcc_hosts(B, threadIdx.x);
  else if (b < 1 + nf2) fold_top_body(B, b - 1, 0, threadIdx.x);
  else zero_topL_body(B, b - 1 - nf2, threadIdx.x);
}

// ------------------------------------------------------------------ stitch
// element (r,c) of the 13x13 AccumulatorApprox matrix from its 91 packed sums (MatrixAccumulators.h:589-618)
__device__ __forceinline__ double acc13(const float* __restrict__ p, int r, int c) {
  if (r > c) { const int t = r; r = c; c = t; }
  if (c < 10) return (double)p[r * 10 - r * (r - 1) / 2 + (c - r)];
  if (r < 10) return (double)p[55 + 3 * r + (c - 10)];
  const int k = (r == 10) ? (c - 10) : (r == 11 ? 3 + (c - 11) : 5);
  return (double)p[85 + k];
}

// index of element (r,c) in the 91 packed sums (same rule as acc13), usable in constant expressions
__host__ __device__ constexpr int acc13_index(int r, int c) {
  if (r > c) { const int t = r; r = c; c = t; }
  if (c < 10) return r * 10 - r * (r - 1) / 2 + (c - r);
  if (r < 10) return 55 + 3 * r + (c - 10);
  return 85 + ((r == 10) ? (c - 10) : (r == 11 ? 3 + (c - 11) : 5));
}
// LDS hand-off between the lanes of ONE wave
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The stitch is "owner computes": one wave per 8x8 output tile, lane (a, c) = element, fixed summation order, no atomics, no LDS —
// results are reproducible run to run.  Two facts keep every tile cheap:
//  (1) adTarget is DIAGONAL: setAdjointsF builds AT = I with AT(6,6) = -aff, AT(7,7) = -1 and scales its rows
//      (EnergyFunctional.cpp:62-92; ba_host.h buildAdjoints), so AT X and X AT^T are row / column scalings (the products with the
//      exact zeros of AT add exact zeros: same doubles as the full product).
//  (2) the Schur sums factor.  With S1(x,y) = sum_j AH(x,j) D(x,j,y) and S2(y,x) = sum_k D(y,x,k) AH(y,k)^T the four updates of
//      AccumulatedSCHessian.cpp:151-171 collapse to
//        H[x,y] = S1(x,y) colscale at(x,y)  +  rowscale at(y,x) S2(y,x)  +  sum_i at(i,x) (x) at(i,y) .* D(i,x,y)
//                 + [x == y]  sum_k S1(x,k) AH(x,k)^T
//      (nf^2 + nf triple products of 8x8 per diagonal tile became 64-term dot products computed once per tile by k_ba_stitch_pre).
// sol layout: [H_A n*n | b_A n | H_L n*n | b_L n | H_sc n*n | b_sc n | x n | lastHS n*n | lastbS n | S1 nf^2*64 | S2 nf^2*64]
__host__ __device__ inline size_t sol_off_pre(int n) { return 4 * ((size_t)n * n + n) + n; }
__host__ __device__ inline size_t sol_doubles(int n, int nf) { return sol_off_pre(n) + 2 * (size_t)nf * nf * 64; }

// grid.x = ceil(2 nf^2 / 4) blocks of 4 waves, one wave per S1 / S2 tile
__global__ __launch_bounds__(256) void k_ba_stitch_pre(const BaDev* __restrict__ wins) {
  const BaDev& B = wins[blockIdx.y];
  if (ba_finished(B)) return;
  const int nf = B.nf, nf2 = nf * nf;
  const int job = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (job >= 2 * nf2) return;
  const int lane = threadIdx.x & 63, a = lane >> 3, c = lane & 7;
  const float* accD = B.accum + acc_off_D(nf);
  const double* adH = B.t_adHost;
  double* pre = B.sol + sol_off_pre(B.n);
  double s = 0;
  if (job < nf2) {                   // S1(x,y)[a][c] = sum_j sum_m AH(x,j)[a][m] D(x,j,y)[m][c]
    const int x = job % nf, y = job / nf;
    for (int j = 0; j < nf; j++) {
      const double* L = adH + (size_t)(x + nf * j) * 64 + a * 8;
      const float* D = accD + (size_t)(x + nf * j + nf2 * y) * 64 + c;
#pragma unroll
      for (int m = 0; m < 8; m++) s += L[m] * (double)D[m * 8];
    }
    pre[(size_t)(x + nf * y) * 64 + lane] = s;
  } else {                           // S2(y,x)[a][c] = sum_k sum_n D(y,x,k)[a][n] AH(y,k)[c][n]
    const int q = job - nf2, y = q % nf, x = q / nf;
    for (int k = 0; k < nf; k++) {
      const float* D = accD + (size_t)(y + nf * x + nf2 * k) * 64 + a * 8;
      const double* R = adH + (size_t)(y + nf * k) * 64 + c * 8;
#pragma unroll
      for (int nn = 0; nn < 8; nn++) s += (double)D[nn] * R[nn];
    }
    pre[(size_t)nf2 * 64 + (size_t)(y + nf * x) * 64 + lane] = s;
  }
}

// grid.x = ceil(3 (nf^2 + nf + 1) / 4) blocks of 4 waves; job -> (matrix m in {0: top A, 1: top L with priors, 2: Schur}, tile);
// tile kinds: frame-frame (x,y) 8x8; frame-calib x: 8x4 + b(8); calib: 4x4 + b(4).
constexpr int ST_WAVES = 4;
// The loops over hosts / targets are partially unrolled so that the loads of several iterations are in flight together: the kernel is
// a chain of memory round trips otherwise, which stretches 2-3x when the linearisation of another batch saturates HBM next to it.
// NF > 0 fixes the keyframe count at compile time (full unrolling at NF = 8 was measured 2x slower: register pressure); NF = 0: runtime nf.
template <int NF>
__global__ __launch_bounds__(64 * ST_WAVES) void k_ba_stitch(const BaDev* __restrict__ wins) {
  const BaDev& B = wins[blockIdx.y];
  if (ba_finished(B)) return;
  const int nf = NF ? NF : B.nf, nf2 = nf * nf, n = B.n;
  const int per = nf2 + nf + 1;
  const int job = blockIdx.x * ST_WAVES + (threadIdx.x >> 6);
  if (job >= 3 * per) return;
  const int m = job / per;
  int tile = job % per;
  double* H = B.sol + (size_t)m * ((size_t)n * n + n);
  double* bvec = H + (size_t)n * n;
  const int lane = threadIdx.x & 63, a = lane >> 3, c = lane & 7;
  const double* adH = B.t_adHost;
  const double* adT = B.t_adTarget;
  auto at = [&](int p, int q, int d) { return adT[(size_t)(p + nf * q) * 64 + d * 9]; };   // diagonal of AT(p,q)

  if (m < 2) {
    const float* acc = B.accum + (m ? acc_off_topL(nf) : acc_off_topA(nf));
    if (tile < nf2) {
      const int x = tile % nf, y = tile / nf;  // block row x, block col y
      double out = 0;
      if (x == y) {
#pragma unroll 2
        for (int t = 0; t < nf; t++) {                     // H[h,h] += AH A AH^T over the targets of host x
          const int aidx = x + nf * t;
          const double* AH = adH + (size_t)aidx * 64;
          const float* ap = acc + (size_t)aidx * 91;
          // the 91 packed sums of the pair in two registers of the wave (lane l: sums l and 64 + l); element (4+mm, 4+nn) is then a
          // v_readlane at a compile-time lane instead of a per-lane index computation and a gather
          const unsigned pk0 = __float_as_uint(ap[lane]), pk1 = __float_as_uint(lane < 27 ? ap[64 + lane] : 0.f);
          double ra[8], rc[8];
#pragma unroll
          for (int q = 0; q < 8; q++) { ra[q] = AH[a * 8 + q]; rc[q] = AH[c * 8 + q]; }
          double s = 0;
#pragma unroll
          for (int mm = 0; mm < 8; mm++) {
            double tv = 0;
#pragma unroll
            for (int nn = 0; nn < 8; nn++) {
              constexpr int dummy = 0; (void)dummy;
              const int idx = acc13_index(4 + mm, 4 + nn);
              const float v = __uint_as_float(__builtin_amdgcn_readlane(idx < 64 ? pk0 : pk1, idx & 63));
              tv += (double)v * rc[nn];
            }
            s += ra[mm] * tv;
          }
          out += s;
        }
#pragma unroll 4
        for (int h = 0; h < nf; h++)                       // H[t,t] += AT A AT^T over the hosts of target x
          out += at(h, x, a) * acc13(acc + (size_t)(h + nf * x) * 91, 4 + a, 4 + c) * at(h, x, c);
        {                                                  // H[h,t] += AH A AT^T of the pair (x,x)
          const double* AH = adH + (size_t)(x + nf * x) * 64;
          const float* ap = acc + (size_t)(x + nf * x) * 91;
          double s = 0;
#pragma unroll
          for (int mm = 0; mm < 8; mm++) s += AH[a * 8 + mm] * acc13(ap, 4 + mm, 4 + c);
          out += s * at(x, x, c);
        }
        if (m == 1 && a == c) out += B.t_prior[x * 8 + a];
      } else {
        // after the symmetrisation of AccumulatedTopHessian.h:133-147 both (x,y) and (y,x) hold M(lo,hi) + M(hi,lo)^T with
        // M(h,t) = AH(h,t) A(h,t) AT(h,t)^T: element (a,c) of tile (x,y) is M(x,y)[a][c] + M(y,x)[c][a] either way
        const double* AH1 = adH + (size_t)(x + nf * y) * 64;
        const float* ap1 = acc + (size_t)(x + nf * y) * 91;
        const double* AH2 = adH + (size_t)(y + nf * x) * 64;
        const float* ap2 = acc + (size_t)(y + nf * x) * 91;
        double s1 = 0, s2 = 0;
#pragma unroll
        for (int mm = 0; mm < 8; mm++) { s1 += AH1[a * 8 + mm] * acc13(ap1, 4 + mm, 4 + c); s2 += AH2[c * 8 + mm] * acc13(ap2, 4 + mm, 4 + a); }
        const double m_lo = x < y ? s1 * at(x, y, c) : s2 * at(y, x, a);      // M(lo,hi) first, then M(hi,lo)^T: the order the CPU adds them in
        const double m_hi = x < y ? s2 * at(y, x, a) : s1 * at(x, y, c);
        out = m_lo + m_hi;
      }
      H[(size_t)(4 + x * 8 + a) * n + (4 + y * 8 + c)] = out;
      return;
    }
    tile -= nf2;
    if (tile < nf) {
      // frame-calib column block and b segment of frame x
      const int x = tile;
      // lane -> (a, c4) for c4 < 4 : H[x-rows, calib cols]; lanes with c in 4..7: c==4 -> b
      double hv = 0;
#pragma unroll 4
      for (int k = 0; k < 2 * nf; k++) {
        const int h = k < nf ? x : k - nf, t = k < nf ? k : x;
        const double* Am = (k < nf ? adH : adT) + (size_t)(h + nf * t) * 64;
        const float* ap = acc + (size_t)(h + nf * t) * 91;
        if (c < 5) {
          const int col = c < 4 ? c : 12;
          double s = 0;
#pragma unroll
          for (int kk = 0; kk < 8; kk++) s += Am[a * 8 + kk] * acc13(ap, 4 + kk, col);
          hv += s;
        }
      }
      if (c < 4) {
        H[(size_t)(4 + x * 8 + a) * n + c] = hv;
        H[(size_t)c * n + (4 + x * 8 + a)] = hv;
      } else if (c == 4) {
        if (m == 1) hv += B.t_prior[x * 8 + a] * B.t_prior[nf * 8 + x * 8 + a];
        bvec[4 + x * 8 + a] = hv;
      }
      return;
    }
    // calib-calib
    if (lane < 20) {
      const int r = lane < 16 ? lane >> 2 : lane - 16, col = lane < 16 ? (lane & 3) : 12;
      double s = 0;
#pragma unroll 16
      for (int p = 0; p < nf2; p++) s += acc13(acc + (size_t)p * 91, r, col);      // (64 dependent round trips when not unrolled: the longest job of the kernel)
      if (lane < 16) {
        if (m == 1 && r == col) s += B.t_prior[nf * 16 + r];
        H[(size_t)r * n + col] = s;
      } else {
        if (m == 1) s += B.t_prior[nf * 16 + r] * (double)B.t_cdelta[r];
        bvec[r] = s;
      }
    }
    return;
  }

  // ---- Schur complement part (AccumulatedSCHessian.cpp:124-186)
  const float* accD = B.accum + acc_off_D(nf);
  const float* accE = B.accum + acc_off_E(nf);
  const float* accEB = B.accum + acc_off_EB(nf);
  const double* S1 = B.sol + sol_off_pre(n);
  const double* S2 = S1 + (size_t)nf2 * 64;
  if (tile < nf2) {
    const int x = tile % nf, y = tile / nf;
    double out = S1[(size_t)(x + nf * y) * 64 + lane] * at(x, y, c);          // H[i,k] += AH(i,j) D AT(i,k)^T     (i = x, k = y)
    out += at(y, x, a) * S2[(size_t)(y + nf * x) * 64 + lane];               // H[j,i] += AT(i,j) D AH(i,k)^T     (j = x, i = y)
#pragma unroll 4
    for (int i = 0; i < nf; i++)                                              // H[j,k] += AT(i,j) D AT(i,k)^T     (j = x, k = y)
      out += at(i, x, a) * (double)accD[(size_t)(i + nf * x + nf2 * y) * 64 + lane] * at(i, y, c);
    if (x == y)                                                               // H[i,i] += AH(i,j) D AH(i,k)^T     (i = x)
#pragma unroll 4
      for (int k = 0; k < nf; k++) {
        const double* s1 = S1 + (size_t)(x + nf * k) * 64 + a * 8;
        const double* R = adH + (size_t)(x + nf * k) * 64 + c * 8;
        double s = 0;
#pragma unroll
        for (int nn = 0; nn < 8; nn++) s += s1[nn] * R[nn];
        out += s;
      }
    H[(size_t)(4 + x * 8 + a) * n + (4 + y * 8 + c)] = out;
    return;
  }
  tile -= nf2;
  if (tile < nf) {
    const int x = tile;
    double hv = 0;
#pragma unroll 4
    for (int k = 0; k < 2 * nf; k++) {
      const int i = k < nf ? x : k - nf, j = k < nf ? k : x;   // pair (i host, j target); frame x is host (AH) or target (AT)
      const int ij = i + nf * j;
      const double* Am = (k < nf ? adH : adT) + (size_t)ij * 64;
      if (c < 5) {
        double s = 0;
#pragma unroll
        for (int kk = 0; kk < 8; kk++) s += Am[a * 8 + kk] * (double)(c < 4 ? accE[(size_t)ij * 32 + kk * 4 + c] : accEB[(size_t)ij * 8 + kk]);
        hv += s;
      }
    }
    if (c < 4) {
      H[(size_t)(4 + x * 8 + a) * n + c] = hv;
      H[(size_t)c * n + (4 + x * 8 + a)] = hv;
    } else if (c == 4) bvec[4 + x * 8 + a] = hv;
    return;
  }
  if (lane < 20) {
    const float v = B.accum[acc_off_Hcc(nf) + lane];
    if (lane < 16) H[(size_t)(lane >> 2) * n + (lane & 3)] = (double)v;
    else bvec[lane - 16] = (double)v;
  }
}

// ------------------------------------------------------------------ solve
// One workgroup per window.  All 256 threads assemble and scale the (8nf+4)^2 system in LDS — memory latency, wants threads.  The
// factorisation, LDL^T with symmetric pivoting on the largest remaining |diagonal| (the strategy of Eigen::LDLT,
// EnergyFunctional.cpp:976; first index wins ties), is run by ONE wave, LEFT-looking, with no workgroup barrier inside:
//   * lane i = matrix position i (positions 64.. on lanes 0..); the running diagonal, the permutation and the pivots live in
//     registers, the pivot is found with DPP row rotations on the 64-bit patterns of |d| (non-negative doubles order like unsigned
//     integers) and a ballot for the first index;
//   * the scaled matrix As is never modified: column k of the factor is  L_ik = (As[perm_i][perm_k] - sum_{q<k} L_iq (d_q L_kq)) / d_k,
//     a dot product over the lane's own row of L with a broadcast vector — 3 instructions per term, no read-modify-write of the
//     matrix, and a pivot exchange swaps two rows of L and four registers instead of rows and columns of the matrix;
//   * one n x n array holds both: As (symmetric, addressed by ORIGINAL index) is read from its strict upper triangle, L (by position)
//     is written into the strict lower triangle, whose initial content — the mirror image of As — is finite and only ever meets zeros
//     of w before it is overwritten.
// The previous right-looking version spent ~8 instructions per element and pivot on one wave (or 3-5 workgroup barriers per pivot
// on four); measured per window on MI355X: 170 us (four waves, barriers) -> 245 us (one wave, right-looking) -> see profiles/.
// Terms are subtracted in pivot order q = 0..k-1 like the right-looking reference; (d_q L_kq) is formed first (one rounding differs).
// Waves 1-3 wait at the closing barrier and help with the adjoint products of the step at the end.
#ifdef SDSO_SOLVE_STAMPS   // diagnostic build only (make EXTRA=-DSDSO_SOLVE_STAMPS, tools/dbg_stamps.py): x[0..8] carry cycle counts
#define STAMP(i) do { if (threadIdx.x == 0) stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define ACCUM(i, t0v) do { if (threadIdx.x == 0) acc_t[i] += __builtin_amdgcn_s_memtime() - (t0v); } while (0)
#define TNOW() __builtin_amdgcn_s_memtime()
#else
#define STAMP(i) do { } while (0)
#define ACCUM(i, t0v) do { } while (0)
#define TNOW() 0ull
#endif
// orthogonalize_x bit 1: lambda of the window's resident GN loop (BaOptDev::lambda, energy-gated flow) instead of the argument
// lane = POSITION (rows of L exchanged physically when a pivot is chosen).
__global__ __launch_bounds__(BA_BLOCK) void k_ba_solve(const BaDev* __restrict__ wins, double lambda, int orthogonalize_x) {
  const BaDev& B = wins[blockIdx.y];
  if (ba_finished(B)) return;
  if (orthogonalize_x & 2) lambda = B.opt->lambda;
  orthogonalize_x &= 1;
#ifdef SDSO_SOLVE_STAMPS
  unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0}, acc_t[6] = {0, 0, 0, 0, 0, 0};
#endif
  STAMP(0);
  const int n = B.n, nf = B.nf, ld = (n + 2) & ~1;   // even row stride: rows are 16-byte aligned
  extern __shared__ double sm[];
  double* A = sm;                 // n*ld  scaled system; the factorisation reads As from the strict upper triangle only
  double* Lm = A;                 //       and builds the unit-lower factor (by position) in the strict lower triangle
  double* bF = A + n * ld;        // n
  double* sv = bF + n;            // n  SVecI
  double* yv = sv + n;            // n
  double* Dg = yv + n;            // n
  double* xv = Dg + n;            // n
  double* wq = xv + n;            // n  d_q L_kq of the current pivot row (+ 16 zeros: the dot products run in trips of 16)
  int* perm = (int*)(wq + n + 16); // n  original index at a position
  const size_t blk = (size_t)n * n + n;
  const double* HA = B.sol; const double* bA = HA + (size_t)n * n;
  const double* HL = B.sol + blk; const double* bL = HL + (size_t)n * n;
  const double* HS = B.sol + 2 * blk; const double* bS = HS + (size_t)n * n;
  double* xout = B.sol + 3 * blk;
  double* lastHS = xout + n;
  double* lastbS = lastHS + (size_t)n * n;
  const double* delta = B.t_prior + nf * 16 + 4;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const double f = (double)1.0f / (1 + lambda);
  // bM_top = bM + HM*delta ; HFinal_top = HL + HM + HA ; bFinal = bL + bM_top + bA - b_sc   (:870, :906-907)
  for (int i = tid; i < n; i += BA_BLOCK) {
    double s = 0;
    for (int k = 0; k < n; k++) s += B.t_HM[(size_t)i * n + k] * delta[k];
    const double bMtop = B.t_bM[i] + s;
    const double v = bL[i] + bMtop + bA[i] - bS[i];
    bF[i] = v;
    lastbS[i] = v;
  }
  {
  for (int e0 = tid; e0 < n * n; e0 += 4 * BA_BLOCK) {       // four elements per trip, loads first
    double hl[4], hm[4], ha[4], hs[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int e = e0 + u * BA_BLOCK, ec = e < n * n ? e : 0;
      hl[u] = HL[ec]; hm[u] = B.t_HM[ec]; ha[u] = HA[ec]; hs[u] = HS[ec];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int e = e0 + u * BA_BLOCK;
      if (e < n * n) {
        const int i = e / n, j = e - i * n;
        double v = hl[u] + hm[u] + ha[u];
        lastHS[e] = v - hs[u];                 // :909
        if (i == j) v *= (1 + lambda);         // :914-916
        v -= hs[u] * f;                        // :918
        A[i * ld + j] = v;
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += BA_BLOCK) sv[i] = 1.0 / sqrt(A[i * ld + i] + 10);   // :967
  __syncthreads();
  for (int e = tid; e < n * n; e += BA_BLOCK) { const int i = e / n, j = e - i * n; A[i * ld + j] = sv[i] * A[i * ld + j] * sv[j]; }
  for (int i = tid; i < n; i += BA_BLOCK) { bF[i] = sv[i] * bF[i]; perm[i] = i; }
  for (int i = tid; i < n + 16; i += BA_BLOCK) wq[i] = 0.0;
  // the dot products of the factorisation run in trips of 16 and read past column n-1 of a row (times an exact zero of w): the
  // padding columns of every row and the vectors behind the matrix must hold finite values, not whatever the last kernel left in LDS
  for (int i = tid; i < n * (ld - n); i += BA_BLOCK) A[(i / (ld - n)) * ld + n + i % (ld - n)] = 0.0;
  for (int i = tid; i < n; i += BA_BLOCK) { yv[i] = 0.0; Dg[i] = 0.0; xv[i] = 0.0; }
  }
  __syncthreads();
  STAMP(1);

  if (wv == 0) {
    const int r0 = lane, r1 = lane + 64;          // positions of this lane (r1 only while r1 < n)
    const bool has0 = r0 < n, has1 = r1 < n;
    const int q0c = has0 ? r0 : 0, q1c = has1 ? r1 : 0;          // clamped: loads stay unconditional and in bounds
    double d0 = has0 ? A[r0 * ld + r0] : 0.0, d1 = has1 ? A[r1 * ld + r1] : 0.0;   // running diagonal of the positions
    int pr0 = q0c, pr1 = q1c;                                     // original index sitting at the position
    double pv0 = 0.0, pv1 = 0.0;                                  // d_q of the pivots this lane's positions became
    auto rl = [](double v, int src) {             // value of lane `src` (wave-uniform index)
      const unsigned long long u = __double_as_longlong(v);
      const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, src), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), src);
      return __longlong_as_double(((unsigned long long)hi << 32) | lo);
    };
    for (int k = 0; k < n; k++) {
      unsigned long long tq = TNOW();
      (void)tq;
      // ---- pivot: first position with the largest |d|, position >= k
      const unsigned long long k0 = (has0 && r0 >= k) ? (unsigned long long)__double_as_longlong(fabs(d0)) : 0ull;
      const unsigned long long k1 = (has1 && r1 >= k) ? (unsigned long long)__double_as_longlong(fabs(d1)) : 0ull;
      unsigned long long m = k0 > k1 ? k0 : k1;   // (lanes without a candidate hold 0 = |+0.0|, the smallest pattern)
#define ROR_MAX(CTRL)                                                                                   \
      {                                                                                                 \
        const unsigned lo = __builtin_amdgcn_update_dpp(0, (unsigned)m, CTRL, 0xf, 0xf, false);         \
        const unsigned hi = __builtin_amdgcn_update_dpp(0, (unsigned)(m >> 32), CTRL, 0xf, 0xf, false); \
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;                               \
        m = o > m ? o : m;                                                                              \
      }
      ROR_MAX(0x128) ROR_MAX(0x124) ROR_MAX(0x122) ROR_MAX(0x121)   // row_ror:8,4,2,1 -> every lane holds the maximum of its 16-lane row
#undef ROR_MAX
      unsigned long long mx;
      {   // the four row maxima sit in lanes 0, 16, 32, 48: three scalar selects
        unsigned long long rm[4];
#pragma unroll
        for (int row = 0; row < 4; row++) {
          const unsigned lo = __builtin_amdgcn_readlane((unsigned)m, 16 * row), hi = __builtin_amdgcn_readlane((unsigned)(m >> 32), 16 * row);
          rm[row] = ((unsigned long long)hi << 32) | lo;
        }
        const unsigned long long m01 = rm[0] > rm[1] ? rm[0] : rm[1], m23 = rm[2] > rm[3] ? rm[2] : rm[3];
        mx = m01 > m23 ? m01 : m23;
      }
      const unsigned long long b0 = __ballot(has0 && r0 >= k && k0 == mx), b1 = __ballot(has1 && r1 >= k && k1 == mx);
      const int p = b0 ? __ffsll((long long)b0) - 1 : 64 + __ffsll((long long)b1) - 1;
      ACCUM(0, tq); tq = TNOW();
      if (p != k) {   // exchange positions k and p: two rows of L (columns < k), the registers of their lanes, the permutation
        const double lk0 = Lm[k * ld + q0c], lp0 = Lm[p * ld + q0c], lk1 = Lm[k * ld + q1c], lp1 = Lm[p * ld + q1c];
        if (has0 && r0 < k) { Lm[k * ld + r0] = lp0; Lm[p * ld + r0] = lk0; }
        if (has1 && r1 < k) { Lm[k * ld + r1] = lp1; Lm[p * ld + r1] = lk1; }
        const double dkk = k < 64 ? rl(d0, k) : rl(d1, k - 64), dpp = p < 64 ? rl(d0, p) : rl(d1, p - 64);
        const int pkk = k < 64 ? __builtin_amdgcn_readlane(pr0, k) : __builtin_amdgcn_readlane(pr1, k - 64);
        const int ppp = p < 64 ? __builtin_amdgcn_readlane(pr0, p) : __builtin_amdgcn_readlane(pr1, p - 64);
        if (k < 64) { if (lane == k) { d0 = dpp; pr0 = ppp; } } else if (lane == k - 64) { d1 = dpp; pr1 = ppp; }
        if (p < 64) { if (lane == p) { d0 = dkk; pr0 = pkk; } } else if (lane == p - 64) { d1 = dkk; pr1 = pkk; }
      }
      ACCUM(1, tq); tq = TNOW();
      const double dk = k < 64 ? rl(d0, k) : rl(d1, k - 64);
      const int pk = k < 64 ? __builtin_amdgcn_readlane(pr0, k) : __builtin_amdgcn_readlane(pr1, k - 64);
      if (k < 64) { if (lane == k) pv0 = dk; } else if (lane == k - 64) pv1 = dk;
      // ---- w_q = d_q L_kq for q < k (lane q), then the column: c_i = As[perm_i][perm_k] - sum_q L_iq w_q
      {
        const double a0 = Lm[k * ld + q0c], a1 = Lm[k * ld + q1c];
        if (has0 && r0 < k) wq[r0] = pv0 * a0;
        if (has1 && r1 < k) wq[r1] = pv1 * a1;
      }
      wave_sync();
      ACCUM(2, tq); tq = TNOW();
      double c0 = A[(pr0 < pk ? pr0 : pk) * ld + (pr0 < pk ? pk : pr0)], c1 = A[(pr1 < pk ? pr1 : pk) * ld + (pr1 < pk ? pk : pr1)];   // As[min][max]
      const double* row0 = Lm + q0c * ld;
      const double* row1 = Lm + q1c * ld;
      const bool any1 = n > 64;                    // (wave-uniform) rows past the 64th exist
      for (int q = 0; q < k; q += 16) {            // sixteen terms per trip (one exposed LDS latency per 16); beyond k the products are exact zeros (w is zero there)
        double lv[16], wv8[16];
#pragma unroll
        for (int u = 0; u < 16; u++) { lv[u] = row0[q + u]; wv8[u] = wq[q + u]; }
        if (any1) {
          double l2[16];
#pragma unroll
          for (int u = 0; u < 16; u++) l2[u] = row1[q + u];
#pragma unroll
          for (int u = 0; u < 16; u++) { c0 = __builtin_fma(-lv[u], wv8[u], c0); c1 = __builtin_fma(-l2[u], wv8[u], c1); }
        } else {
#pragma unroll
          for (int u = 0; u < 16; u++) c0 = __builtin_fma(-lv[u], wv8[u], c0);
        }
      }
      ACCUM(3, tq); tq = TNOW();
      // ---- L_ik, running diagonal (A(i,i) -= (L_ik d_k) L_ik)
      if (dk != 0.0) {
        // one division stream per pivot: from pivot 4 on, lanes 0..3 (whose first positions are factored) divide for their second
        // positions 64.. in the same instructions; before that the few second positions take a pass of their own
        const bool fold = k >= 4 || !any1;
        const bool second = fold && has1 && r1 > k;          // this lane divides c1 in the common stream
        const double l = (second ? c1 : c0) / dk;
        if (second) { Lm[r1 * ld + k] = l; d1 = d1 - (l * dk) * l; }
        else if (has0 && r0 > k) { Lm[r0 * ld + k] = l; d0 = d0 - (l * dk) * l; }
        if (!fold && has1 && r1 > k) { const double l2 = c1 / dk; Lm[r1 * ld + k] = l2; d1 = d1 - (l2 * dk) * l2; }
      } else {
        if (has0 && r0 > k) Lm[r0 * ld + k] = 0;
        if (has1 && r1 > k) Lm[r1 * ld + k] = 0;
      }
      wave_sync();
      ACCUM(4, tq);
    }
    STAMP(2);
    if (has0) { Dg[r0] = pv0; perm[r0] = pr0; }
    if (has1) { Dg[r1] = pv1; perm[r1] = pr1; }
    wave_sync();
    // ---- triangular solves: lane owns unknowns lane and lane+64
    const int i0 = lane, i1 = lane + 64;
    double y0 = i0 < n ? bF[perm[i0]] : 0.0, y1 = i1 < n ? bF[perm[i1]] : 0.0;
    for (int j0 = 0; j0 < n; j0 += 8) {  // forward, column sweep: y[i] -= L(i,j) y[j], i > j  (j ascending like the reference); eight columns of the lane's rows per trip
      double a0[8], a1[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { a0[u] = Lm[q0c * ld + j0 + u]; a1[u] = Lm[q1c * ld + j0 + u]; }    // (reads past column n-1 land in the row below / the vectors: unused)
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int j = j0 + u;
        if (j < n) {
          const double yj = rl(j < 64 ? y0 : y1, j & 63);
          if (i0 > j && i0 < n) y0 -= a0[u] * yj;
          if (i1 > j && i1 < n) y1 -= a1[u] * yj;
        }
      }
    }
    if (i0 < n) y0 = pv0 != 0.0 ? y0 / pv0 : 0.0;
    if (i1 < n) y1 = pv1 != 0.0 ? y1 / pv1 : 0.0;
    for (int j0 = n - 1; j0 >= 0; j0 -= 8) {  // backward, column sweep: y[i] -= L(j,i) y[j], i < j
      double a0[8], a1[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int j = j0 - u < 0 ? 0 : j0 - u; a0[u] = Lm[j * ld + q0c]; a1[u] = Lm[j * ld + q1c]; }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int j = j0 - u;
        if (j >= 0) {
          const double yj = rl(j < 64 ? y0 : y1, j & 63);
          if (i0 < j && i0 < n) y0 -= a0[u] * yj;
          if (i1 < j && i1 < n) y1 -= a1[u] * yj;
        }
      }
    }
    if (i0 < n) xv[pr0] = y0;
    if (i1 < n) xv[pr1] = y1;
    STAMP(3);
  }
  __syncthreads();
  for (int i = tid; i < n; i += BA_BLOCK) xv[i] = sv[i] * xv[i];   // x = SVecI * solve(...)  (:976)
  __syncthreads();
  if (orthogonalize_x) {  // x -= P x   (:980-984, :824-826)
    for (int i = tid; i < n; i += BA_BLOCK) {
      double s = 0;
      for (int k = 0; k < n; k++) s += B.t_P[(size_t)i * n + k] * xv[k];
      yv[i] = xv[i] - s;
    }
    __syncthreads();
    for (int i = tid; i < n; i += BA_BLOCK) xv[i] = yv[i];
    __syncthreads();
  }
  for (int i = tid; i < n; i += BA_BLOCK) xout[i] = xv[i];
  // xAd[nf*h+t] = xF(h)^T adHostF[h+nf*t] + xF(t)^T adTargetF[h+nf*t]   (:289-291), float arithmetic
  float* xAd = const_cast<float*>(B.t_xAd);
  for (int e = tid; e < nf * nf * 8; e += BA_BLOCK) {
    const int j = e & 7, ht = e >> 3, h = ht / nf, t = ht % nf;
    const double* AH = B.t_adHost + (size_t)(h + nf * t) * 64;
    const double* AT = B.t_adTarget + (size_t)(h + nf * t) * 64;
    float sh = 0, st = 0;
    for (int i = 0; i < 8; i++) { sh += (float)xv[4 + 8 * h + i] * (float)AH[i * 8 + j]; st += (float)xv[4 + 8 * t + i] * (float)AT[i * 8 + j]; }
    xAd[e] = sh + st;
  }
#ifdef SDSO_SOLVE_STAMPS
  __syncthreads();
  STAMP(4);
  if (threadIdx.x == 0) {
    for (int i = 0; i < 4; i++) xout[i] = (double)(stamps[i + 1] - stamps[i]);   // assemble+scale | factorise | triangular | tail
    for (int i = 0; i < 5; i++) xout[4 + i] = (double)acc_t[i];                  // pivot search | exchange | w | dot products | division
  }
#endif
}
#undef STAMP
#undef ACCUM
#undef TNOW

// ------------------------------------------------------------------ back-substitution, one lane per point
// EnergyFunctional::resubstituteFPt (EnergyFunctional.cpp:305-341) for point p: b = bdSumF - xc . Hcd - sum_r xAd[r] . JpJdF[r] over the active
// residuals in EFPoint::residualsAll order — the order the point's records lie in BaDev::r_cj, their targets in the nibbles of p_order — the
// subtractions of the reference one after the other; step = -b HdiF.  The active records are the bits the Schur kernel left in p_track.
// Every load of the point is issued before the first one is consumed (the loop is unrolled to the 8 residuals a point can hold and
// predicated): one memory round trip instead of one per residual.  WITH_L: the point's L sums (linearised / marginalised residuals) exist.
// x_cal / xAd: the solution's calibration part and the adjoint products where the caller has them (the fused tail kernel: in LDS);
// nullptr: where the solve left them in global memory
template <bool WITH_L>
__device__ __forceinline__ float resub_point(const BaDev& B, int p, const float* po, const double* x_cal = nullptr, const float* xAd = nullptr) {
  const double* x = x_cal ? x_cal : B.sol + 3 * ((size_t)B.n * B.n + B.n);
  const float* xAd_tab = xAd ? xAd : B.t_xAd;
  const int nf = B.nf;
  const float* recs = B.r_cj + (size_t)B.p_rbeg[p] * 8;
  const int h = B.p_host[p];
  const unsigned ord = B.p_order[p];
  const int cnt = 8 - (__clz((int)~ord) >> 2);    // the nibbles of `ord` that are not 0xF (targets are < 8: the top bit of a used nibble is clear)
  float4 j0[8], j1[8], xa0[8], xa1[8];
  const unsigned good = (unsigned)__float_as_int(((const float*)(B.p_track + p))[3]);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (k < cnt) {
      const float* rec = recs + k * 8;
      j0[k] = *reinterpret_cast<const float4*>(rec); j1[k] = *reinterpret_cast<const float4*>(rec + 4);
      const float* xa = xAd_tab + (size_t)(h * nf + (int)((ord >> (4 * k)) & 15u)) * 8;
      xa0[k] = *reinterpret_cast<const float4*>(xa); xa1[k] = *reinterpret_cast<const float4*>(xa + 4);
    }
  }
  const float4 hA = *reinterpret_cast<const float4*>(po + PO_HCD_A);
  float4 hL = make_float4(0.f, 0.f, 0.f, 0.f);
  if (WITH_L) hL = *reinterpret_cast<const float4*>(po + PO_HCD_L);
  const float bsum = po[PO_BDSUM], hdi = po[PO_HDI];
  const double x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
  if (good == 0) return 0.f;
  float b = bsum;
  float d = 0;
  d += (float)x0 * (hA.x + hL.x);
  d += (float)x1 * (hA.y + hL.y);
  d += (float)x2 * (hA.z + hL.z);
  d += (float)x3 * (hA.w + hL.w);
  b -= d;
#pragma unroll
  for (int k = 0; k < 8; k++) {            // xAd . JpJdF of every record, subtracted in residualsAll order
    if (k < cnt) {
      float a = 0;
      a += xa0[k].x * j0[k].x; a += xa0[k].y * j0[k].y; a += xa0[k].z * j0[k].z; a += xa0[k].w * j0[k].w;
      a += xa1[k].x * j1[k].x; a += xa1[k].y * j1[k].y; a += xa1[k].z * j1[k].z; a += xa1[k].w * j1[k].w;
      if ((good >> k) & 1u) b -= a;
    }
  }
  return -b * hdi;
}
template <bool WITH_L>
__global__ __launch_bounds__(BA_BLOCK) void k_ba_resub(const BaDev* __restrict__ wins) {
  const BaDev& B = wins[blockIdx.y];
  if (ba_finished(B)) return;
  const int p = blockIdx.x * BA_BLOCK + threadIdx.x;
  if (p >= B.np) return;
  float* po = B.p_out + (size_t)p * 16;
  // SOLVER_MOMENTUM: backupState keeps the step this back-substitution replaces (ph->step_backup = ph->step, zero before a loop's first
  // iteration: FullSystemOptimize.cpp:311-345).  Kept HERE, where the old step is overwritten, so that every flow that solves has it.
  // (Windows with that bit never take the fused k_ba_resub_step / TAIL_RESUB forms: opt_solve_step.)
  if (B.solver_mode & SOLVER_MOMENTUM) B.p_stepbk[p] = B.opt->iterations != 0 ? po[PO_STEP] : 0.f;
  po[PO_STEP] = resub_point<WITH_L>(B, p, po);
}

// one point of k_ba_resub_step: resubstituteFPt, backupState, doStepFromBackup (stepfacD = 1); sID / sNID: its terms of the break test's sums
template <bool WITH_L>
__device__ __forceinline__ void resub_step_point(const BaDev& B, int p, const double* x_cal, const float* xAd, float& sID, float& sNID) {
  float* po = B.p_out + (size_t)p * 16;
  float4 g = B.p_geo[p];
  const float st = resub_point<WITH_L>(B, p, po, x_cal, xAd);
  po[PO_STEP] = st;
  po[PO_BACKUP] = g.z;
  const float bk = g.z, nid = bk + 1.0f * st;
  g.z = nid; g.w = nid;     // setIdepth + setIdepthZero
  B.p_geo[p] = g;
  B.p_delta[p] = nid - nid;
  sID = st * st; sNID = fabsf(bk);
}

// resubstituteFPt + backupState + doStepFromBackup (stepfacD = 1) of the points in ONE pass over the point data (k_ba_resub followed
// by k_ba_points_op op 3), after the fused tail kernel of the resident loop.  expect_iterations >= 0: the window takes part iff its
// loop has taken exactly that many steps — the tail kernel that just ran may have set `finished` for the NEXT iteration (the break
// test fires after the step it belongs to); < 0: the plain `finished` test (sharded windows: k_ba_opt_step has not run yet).
template <bool WITH_L>
__global__ __launch_bounds__(BA_BLOCK) void k_ba_resub_step(const BaDev* __restrict__ wins, int expect_iterations, float* __restrict__ sums, int sums_stride) {
  const BaDev& B = wins[blockIdx.y];
  if (expect_iterations >= 0 ? (ba_finished_lin(B) || B.opt->iterations != expect_iterations) : ba_finished(B)) return;
  const int p = blockIdx.x * BA_BLOCK + threadIdx.x;
  float sID = 0, sNID = 0;
  if (p < B.np) resub_step_point<WITH_L>(B, p, nullptr, nullptr, sID, sNID);
  if (sums) {
    __shared__ float r0[BA_BLOCK / 64], r1[BA_BLOCK / 64];
    const float a = wave_sum(sID), b = wave_sum(sNID);
    if ((threadIdx.x & 63) == 0) { r0[threadIdx.x >> 6] = a; r1[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float s0 = 0, s1 = 0;
      for (int w = 0; w < BA_BLOCK / 64; w++) { s0 += r0[w]; s1 += r1[w]; }
      float* so = sums + (size_t)blockIdx.y * sums_stride;
      so[blockIdx.x * 2] = s0; so[blockIdx.x * 2 + 1] = s1;
    }
  }
}

// The solution of a window as one record of floats: x (n doubles), the adjoint products xAd (nf^2 * 8) and the residual count nres of the
// accumulate — everything the kernels after the solve read of it (k_ba_resub_step: x, xAd; k_ba_opt_step: x, nres).  Sharded windows
// under the reduce-scatter exchange are solved on one rank each: that rank packs (unpack = 0, its windows [first, first + count)), the
// records travel by all-gather, every other rank unpacks (unpack = 1, the windows outside that range).  grid.x = windows of the batch.
__host__ __device__ inline int sol_rec_floats(int n, int nf) { return 2 * n + nf * nf * 8 + 2; }
__global__ __launch_bounds__(256) void k_ba_sol_record(const BaDev* __restrict__ wins, int first, int count, float* __restrict__ recs, int unpack) {
  const int w = blockIdx.x;
  const bool own = w >= first && w < first + count;
  if (unpack ? own : !own) return;
  const BaDev& B = wins[w];
  const int n = B.n, nf = B.nf, tid = threadIdx.x;
  double* x = B.sol + 3 * ((size_t)n * n + n);
  float* xAd = const_cast<float*>(B.t_xAd);
  float* nres = B.accum + acc_off_nres(nf);
  float* rec = recs + (size_t)w * sol_rec_floats(n, nf);
  double* rx = (double*)rec;                 // (records are an even number of floats: 8-byte aligned)
  float* ra = rec + 2 * n;
  float* rn = ra + nf * nf * 8;
  if (unpack) {
    if (tid < n) x[tid] = rx[tid];
    for (int e = tid; e < nf * nf * 8; e += 256) xAd[e] = ra[e];
    if (tid < 2) nres[tid] = rn[tid];
  } else {
    if (tid < n) rx[tid] = x[tid];
    for (int e = tid; e < nf * nf * 8; e += 256) ra[e] = xAd[e];
    if (tid < 2) rn[tid] = nres[tid];
  }
}

// ------------------------------------------------------------------ the marginalisation prior, resident
// EnergyFunctional::marginalizePointsF's last lines (EnergyFunctional.cpp:727-728): HM += setting_margWeightFac * (M - Msc), bM likewise, on the
// DEVICE copy of the prior — M / Mb and Msc / Mbsc are blocks 0 and 2 of BaDev::sol, where the stitch kernels just put them.
__global__ __launch_bounds__(256) void k_ba_prior_add(const BaDev* __restrict__ wins, double fac) {
  const BaDev& B = wins[blockIdx.y];
  const int n = B.n;
  const size_t blk = (size_t)n * n + n;
  const double* MA = B.sol; const double* MS = B.sol + 2 * blk;
  double* HM = const_cast<double*>(B.t_HM); double* bM = const_cast<double*>(B.t_bM);
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n * n + n; e += gridDim.x * 256) {
    if (e < n * n) HM[e] += fac * (MA[e] - MS[e]);
    else bM[e - n * n] += fac * (MA[e] - MS[e]);
  }
}

// The same with setting_solverMode's nullspace bits (EnergyFunctional.cpp:707-731): H = M - Msc, b = Mb - Mbsc; SOLVER_ORTHOGONALIZE_POINTMARG
// (a window without frame 0 only): orthogonalize(&b, &H) — b -= P b, H -= (P H) P with the gauge projector P = NNpiTS (:775-835; BaDev::t_P, built
// from the nullspaces of the window's current evaluation points, as FullSystem.cpp:1453 refreshes them right before marginalizePointsF) —;
// HM += fac H, bM += fac b; SOLVER_ORTHOGONALIZE_FULL: orthogonalize(&bM, &HM).  One 256-thread workgroup per window, f64 in LDS.
__global__ __launch_bounds__(256) void k_ba_prior_orth(const BaDev* __restrict__ wins, double fac, int orth_marg, int orth_full) {
  const BaDev& B = wins[blockIdx.y];
  const int n = B.n, tid = threadIdx.x;
  const size_t blk = (size_t)n * n + n;
  const double* MA = B.sol; const double* MS = B.sol + 2 * blk;
  const double* P = B.t_P;
  double* HM = const_cast<double*>(B.t_HM); double* bM = const_cast<double*>(B.t_bM);
  constexpr int NM = 68;
  __shared__ double H[NM * NM], T[NM * NM], b[NM], pb[NM];
  auto project = [&]() {   // b -= P b;  H -= (P H) P
    for (int e = tid; e < n * n; e += 256) { const int i = e / n, j = e - i * n; double s = 0; for (int k = 0; k < n; k++) s += P[(size_t)i * n + k] * H[k * n + j]; T[e] = s; }
    if (tid < n) { double s = 0; for (int k = 0; k < n; k++) s += P[(size_t)tid * n + k] * b[k]; pb[tid] = s; }
    __syncthreads();
    for (int e = tid; e < n * n; e += 256) { const int i = e / n, j = e - i * n; double s = 0; for (int k = 0; k < n; k++) s += T[i * n + k] * P[(size_t)k * n + j]; H[e] -= s; }
    if (tid < n) b[tid] -= pb[tid];
    __syncthreads();
  };
  for (int e = tid; e < n * n; e += 256) H[e] = MA[e] - MS[e];
  if (tid < n) b[tid] = MA[(size_t)n * n + tid] - MS[(size_t)n * n + tid];
  __syncthreads();
  if (orth_marg) project();
  for (int e = tid; e < n * n; e += 256) H[e] = HM[e] + fac * H[e];
  if (tid < n) b[tid] = bM[tid] + fac * b[tid];
  __syncthreads();
  if (orth_full) project();
  for (int e = tid; e < n * n; e += 256) HM[e] = H[e];
  if (tid < n) bM[tid] = b[tid];
}

// EnergyFunctional::marginalizeFrame (EnergyFunctional.cpp:554-660) on the device-resident prior of one window: the frame's 8 rows / columns
// go to the end (step 1), its prior is added (step 2), the system is scaled by 1 / sqrt(|diag| + 10), the 8 x 8 corner inverted, the Schur
// complement taken and the scaling undone (step 3), the result symmetrised.  One 256-thread workgroup, everything f64 in LDS; every sum
// runs over its 8 terms in index order, the inverse is the partial-pivoting elimination of sdso_ba_marginalize_frame (the host statement
// of the same function: bit-identical results).  HMs / bMs: the prior it starts from, odim x odim row-major and odim — the window's resident
// prior, or the result of the previous call when several frames leave at one keyframe (FullSystem.cpp:1470-1476 marginalises every
// flagged frame in a loop).  idx: the frame's position among the frames that prior covers.  pr8: EFFrame::prior (8) then delta_prior (8).
// out: (odim - 8)^2 row-major, then odim - 8.
constexpr int MF_MAXN = 68;
__global__ __launch_bounds__(256) void k_ba_marg_frame(const double* __restrict__ HMs, const double* __restrict__ bMs, int odim, int idx, const double* __restrict__ pr8, double* __restrict__ out) {
  const int ndim = odim - 8, tid = threadIdx.x;
  __shared__ double H[MF_MAXN * MF_MAXN], b[MF_MAXN], S[MF_MAXN], Si[MF_MAXN], bli[(MF_MAXN - 8) * 8], inv[64];
  __shared__ int ord[MF_MAXN];
  if (tid < odim) {
    const int lo = idx * 8 + 4;
    ord[tid] = tid < lo ? tid : tid < ndim ? tid + 8 : lo + (tid - ndim);     // the others keep their order, the frame's 8 follow
  }
  __syncthreads();
  for (int e = tid; e < odim * odim; e += 256) { const int i = e / odim, j = e - i * odim; H[e] = HMs[(size_t)ord[i] * odim + ord[j]]; }
  if (tid < odim) b[tid] = bMs[ord[tid]];
  __syncthreads();
  if (tid < 8) { H[(ndim + tid) * odim + ndim + tid] += pr8[tid]; b[ndim + tid] += pr8[tid] * pr8[8 + tid]; }
  __syncthreads();
  if (tid < odim) { S[tid] = sqrt(fabs(H[tid * odim + tid]) + 10); Si[tid] = 1.0 / S[tid]; }
  __syncthreads();
  for (int e = tid; e < odim * odim; e += 256) { const int i = e / odim, j = e - i * odim; H[e] = Si[i] * H[e] * Si[j]; }
  if (tid < odim) b[tid] = Si[tid] * b[tid];
  __syncthreads();
  if (tid == 0) {   // hpi = 0.5f * (hpi + hpi); hpi = hpi.inverse(); hpi = 0.5f * (hpi + hpi)   (:617-620)
    double A[8][8], iv[8][8];
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) { const double v = H[(ndim + i) * odim + ndim + j]; A[i][j] = 0.5f * (v + v); iv[i][j] = i == j; }
    for (int k = 0; k < 8; k++) {
      int pv = k;
      for (int i = k + 1; i < 8; i++) if (fabs(A[i][k]) > fabs(A[pv][k])) pv = i;
      if (pv != k) for (int j = 0; j < 8; j++) { const double t1 = A[k][j]; A[k][j] = A[pv][j]; A[pv][j] = t1; const double t2 = iv[k][j]; iv[k][j] = iv[pv][j]; iv[pv][j] = t2; }
      const double d = A[k][k];
      for (int j = 0; j < 8; j++) { A[k][j] /= d; iv[k][j] /= d; }
      for (int i = 0; i < 8; i++) {
        if (i == k) continue;
        const double f = A[i][k];
        if (f == 0) continue;
        for (int j = 0; j < 8; j++) { A[i][j] -= f * A[k][j]; iv[i][j] -= f * iv[k][j]; }
      }
    }
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) inv[i * 8 + j] = 0.5f * (iv[i][j] + iv[i][j]);
  }
  __syncthreads();
  for (int e = tid; e < ndim * 8; e += 256) {          // bli = bottomLeft^T * hpi
    const int r = e >> 3, c = e & 7;
    double s = 0;
    for (int k = 0; k < 8; k++) s += H[(ndim + k) * odim + r] * inv[k * 8 + c];
    bli[e] = s;
  }
  __syncthreads();
  for (int e = tid; e < ndim * ndim; e += 256) {
    const int r = e / ndim, c = e - r * ndim;
    double s = 0;
    for (int k = 0; k < 8; k++) s += bli[r * 8 + k] * H[(ndim + k) * odim + c];
    H[r * odim + c] -= s;
  }
  if (tid < ndim) {
    double s = 0;
    for (int k = 0; k < 8; k++) s += bli[tid * 8 + k] * b[ndim + k];
    b[tid] -= s;
  }
  __syncthreads();
  for (int e = tid; e < ndim * ndim; e += 256) { const int r = e / ndim, c = e - r * ndim; H[r * odim + c] = S[r] * H[r * odim + c] * S[c]; }
  __syncthreads();
  for (int e = tid; e < ndim * ndim; e += 256) { const int r = e / ndim, c = e - r * ndim; out[e] = 0.5 * (H[r * odim + c] + H[c * odim + r]); }
  if (tid < ndim) out[(size_t)ndim * ndim + tid] = S[tid] * b[tid];
}

// FullSystem::backupState / doStepFromBackup / loadSateBackup for the points.  op: 0 backup, 1 step, 2 restore
// op 3 = backup + step in one pass (the resident loop never restores).  stepfacD < 0: the window's own stepsize (BaOptDev::stepsize,
// SOLVER_STEPMOMENTUM).  SOLVER_MOMENTUM (FullSystemOptimize.cpp:238-250): step + 0.5f * step_backup, no step factor.
__global__ __launch_bounds__(BA_BLOCK) void k_ba_points_op(const BaDev* __restrict__ wins, int op, float stepfacD, float* __restrict__ sums /* per block: sumID, sumNID */,
                                                           int sums_stride = 0 /* floats between the windows' sums */, int cond = 0) {
  const BaDev& B = wins[blockIdx.y];
  if (ba_finished(B)) return;
  if (ba_gate_skip(B, cond)) return;
  const int p = blockIdx.x * BA_BLOCK + threadIdx.x;
  float sID = 0, sNID = 0;
  if (p < B.np) {
    float* po = B.p_out + (size_t)p * 16;
    float4 g = B.p_geo[p];
    if (op == 0) po[PO_BACKUP] = g.z;
    else if (op == 1 || op == 3) {
      if (op == 3) po[PO_BACKUP] = g.z;
      float st = po[PO_STEP];
      const float bk = op == 3 ? g.z : po[PO_BACKUP];
      float fac = stepfacD < 0 ? B.opt->stepsize : stepfacD;
      if (B.solver_mode & SOLVER_MOMENTUM) { st = st + 0.5f * (B.p_stepbk[p]); fac = 1.0f; }
      const float nid = bk + fac * st;
      g.z = nid; g.w = nid;     // setIdepth + setIdepthZero
      B.p_geo[p] = g;
      B.p_delta[p] = nid - nid;
      sID = st * st; sNID = fabsf(bk);
    } else {
      const float bk = po[PO_BACKUP];
      g.z = bk; g.w = bk;
      B.p_geo[p] = g;
      B.p_delta[p] = 0;
    }
  }
  if (sums) {
    __shared__ float r0[BA_BLOCK / 64], r1[BA_BLOCK / 64];
    const float a = wave_sum(sID), b = wave_sum(sNID);
    if ((threadIdx.x & 63) == 0) { r0[threadIdx.x >> 6] = a; r1[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float s0 = 0, s1 = 0;
      for (int w = 0; w < BA_BLOCK / 64; w++) { s0 += r0[w]; s1 += r1[w]; }
      float* so = sums + (size_t)blockIdx.y * sums_stride;
      so[blockIdx.x * 2] = s0; so[blockIdx.x * 2 + 1] = s1;
    }
  }
}

}  // namespace sdso
