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

// ------------------------------------------------------------------ fold of the Schur partials
// grid.x = nf^3 (D bins, 64 threads each) + nf^2 (E/EB per pair) + 1 (Hcc, bc)
__device__ __forceinline__ void fold_sc_body(const BaDev& B, int b, int lane) {
  const int nf = B.nf, nf2 = nf * nf, nf3 = nf2 * nf;
  const int pf = sc_part_floats(nf);
  if (b < nf3) {
    // accD[h + t1*nf + t2*nf^2] <- items of host h, tile (t1,t2)
    const int h = b % nf, t1 = (b / nf) % nf, t2 = b / nf2;
    float s = 0;
    for (int it = B.host_item_beg[h]; it < B.host_item_beg[h + 1]; it++) s += B.sc_part[(size_t)it * pf + (t1 * nf + t2) * 64 + lane];
    B.accum[acc_off_D(nf) + (size_t)b * 64 + lane] = s;
    return;
  }
  b -= nf3;
  if (b < nf2) {
    const int h = b % nf, t1 = b / nf;
    if (lane < 40) {
      const int off = lane < 32 ? nf2 * 64 + t1 * 32 + lane : nf2 * 64 + nf * 32 + t1 * 8 + (lane - 32);
      float s = 0;
      for (int it = B.host_item_beg[h]; it < B.host_item_beg[h + 1]; it++) s += B.sc_part[(size_t)it * pf + off];
      if (lane < 32) B.accum[acc_off_E(nf) + (size_t)b * 32 + lane] = s;
      else B.accum[acc_off_EB(nf) + (size_t)b * 8 + (lane - 32)] = s;
    }
    return;
  }
  if (lane < 20) {
    const int off = nf2 * 64 + nf * 32 + nf * 8 + lane;
    float s = 0;
    for (int it = 0; it < B.nitems; it++) s += B.sc_part[(size_t)it * pf + off];
    B.accum[acc_off_Hcc(nf) + lane] = s;  // Hcc 16 then bc 4 are contiguous
  }
}

__global__ __launch_bounds__(64) void k_ba_fold_sc(const BaDev* __restrict__ wins) { fold_sc_body(wins[blockIdx.y], blockIdx.x, threadIdx.x); }
// Hcc / bc after k_ba_sc_host: nf per-host partials of 20 floats
__device__ __forceinline__ void fold_hcc_hosts(const BaDev& B, int lane) {
  if (lane < 20) {
    float s = 0;
    for (int h = 0; h < B.nf; h++) s += B.sc_part[(size_t)h * 20 + lane];
    B.accum[acc_off_Hcc(B.nf) + lane] = s;
  }
}
__global__ __launch_bounds__(64) void k_ba_fold_hcc(const BaDev* __restrict__ wins) { fold_hcc_hosts(wins[blockIdx.y], threadIdx.x); }
// every fold of one accumulate phase in ONE launch (the usual case: no linearized residuals, topL is just cleared):
// grid.x = [nf^3 + nf^2 + 1 Schur bins | nf^2 top-A pairs | nf^2 top-L pairs], 128 threads
// host_sc != 0: the Schur bins were written by k_ba_sc_host, only Hcc / bc remain (grid.x = 1 + 2 nf^2)
__global__ __launch_bounds__(128) void k_ba_fold_all(const BaDev* __restrict__ wins, int host_sc) {
  const BaDev& B = wins[blockIdx.y];
  const int nf = B.nf, nf2 = nf * nf, nsc = host_sc ? 1 : nf2 * nf + nf2 + 1;
  const int b = blockIdx.x;
  if (b < nsc) { if (host_sc) fold_hcc_hosts(B, threadIdx.x); else if (threadIdx.x < 64) fold_sc_body(B, b, threadIdx.x); }
  else if (b < nsc + nf2) fold_top_body(B, b - nsc, 0, threadIdx.x);
  else zero_topL_body(B, b - nsc - nf2, threadIdx.x);
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

// One wave computes out[a][c] += sum_{m,n} L[a][m] X[m][n] R[c][n] for 8x8 row-major L, X, R held in LDS.
struct TileWork {
  double* Ls; double* Xs; double* Rs; double* Ts;  // 64 doubles each (LDS)
};
// LDS hand-off between the lanes of ONE wave (each wave of the stitch kernel owns its tile buffers)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double lxr(const TileWork& W, int a, int c) {
  double t = 0;
#pragma unroll
  for (int m = 0; m < 8; m++) t += W.Ls[a * 8 + m] * W.Xs[m * 8 + c];
  W.Ts[a * 8 + c] = t;
  wave_sync();
  double o = 0;
#pragma unroll
  for (int n = 0; n < 8; n++) o += W.Ts[a * 8 + n] * W.Rs[c * 8 + n];
  wave_sync();
  return o;
}

// grid.x = 3 * (nf*nf + nf + 1) tiles: matrix m in {0: top A, 1: top L (with priors), 2: SC};
// tile kinds: frame-frame (x,y) 8x8; frame-calib x: 8x4 + b(8); calib: 4x4 + b(4).
// sol layout: [H_A n*n | b_A n | H_L n*n | b_L n | H_sc n*n | b_sc n | ...]
// A workgroup = 4 waves per output tile: the 8x8x8 triple products of a tile (up to nf^2 + 3nf of them for a diagonal
// Schur tile) are dealt round-robin to the waves, each with its own LDS operands, and the four partial tiles are added
// in wave order at the end (fixed order: reproducible).  Tile kinds without a product chain run on wave 0 alone.
constexpr int ST_WAVES = 4;
__global__ __launch_bounds__(64 * ST_WAVES) void k_ba_stitch(const BaDev* __restrict__ wins) {
  const BaDev& B = wins[blockIdx.y];
  const int nf = B.nf, nf2 = nf * nf, n = B.n;
  const int per = nf2 + nf + 1;
  const int m = blockIdx.x / per;
  int tile = blockIdx.x % per;
  if (m >= 3) return;
  double* H = B.sol + (size_t)m * ((size_t)n * n + n);
  double* bvec = H + (size_t)n * n;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, a = lane >> 3, c = lane & 7;
  __shared__ double sLb[ST_WAVES][64], sXb[ST_WAVES][64], sRb[ST_WAVES][64], sTb[ST_WAVES][64];
  double* sL = sLb[wv]; double* sX = sXb[wv]; double* sR = sRb[wv]; double* sT = sTb[wv];
  TileWork W{sL, sX, sR, sT};
  int turn = 0;
  auto mine = [&]() { return ((turn++) % ST_WAVES) == wv; };
  // sum of the waves' partial tiles, in wave order; returns the total on wave 0
  auto fold = [&](double part) {
    sTb[wv][lane] = part;
    __syncthreads();
    double s = sTb[0][lane];
#pragma unroll
    for (int w = 1; w < ST_WAVES; w++) s += sTb[w][lane];
    return s;
  };
  const double* adH = B.t_adHost;
  const double* adT = B.t_adTarget;

  if (m < 2) {
    const float* acc = B.accum + (m ? acc_off_topL(nf) : acc_off_topA(nf));
    if (tile < nf2) {
      const int x = tile % nf, y = tile / nf;  // block row x, block col y
      double out = 0;
      auto pairprod = [&](int h, int t, const double* Lm, const double* Rm) {
        const int aidx = h + nf * t;
        sL[lane] = Lm[(size_t)aidx * 64 + lane];
        sR[lane] = Rm[(size_t)aidx * 64 + lane];
        sX[lane] = acc13(acc + (size_t)aidx * 91, 4 + a, 4 + c);
        wave_sync();
        return lxr(W, a, c);
      };
      if (x == y) {
        for (int t = 0; t < nf; t++) if (mine()) out += pairprod(x, t, adH, adH);   // H[h,h] += AH A AH^T
        for (int h = 0; h < nf; h++) if (mine()) out += pairprod(h, x, adT, adT);   // H[t,t] += AT A AT^T
        if (mine()) out += pairprod(x, x, adH, adT);                                // H[h,t] with h==t
        out = fold(out);
        if (wv != 0) return;
        if (m == 1 && a == c) out += B.t_prior[x * 8 + a];
      } else {
        if (wv != 0) return;
        // after the symmetrisation of AccumulatedTopHessian.h:133-147: for lo<hi
        //   H[lo,hi] = M(lo,hi) + M(hi,lo)^T ;  H[hi,lo] = H[lo,hi]^T,  M(h,t) = AH_ht A_ht AT_ht^T
        const int lo = x < y ? x : y, hi = x < y ? y : x;
        const double m1 = pairprod(lo, hi, adH, adT);   // element (a,c) of M(lo,hi)
        const double m2 = pairprod(hi, lo, adH, adT);   // element (a,c) of M(hi,lo)
        // need m1[a][c] + m2[c][a] for tile (lo,hi); transpose through LDS
        sT[a * 8 + c] = m2;
        wave_sync();
        const double up = m1 + sT[c * 8 + a];           // (lo,hi)[a][c]
        wave_sync();
        if (x < y) out = up;
        else { sT[a * 8 + c] = up; wave_sync(); out = sT[c * 8 + a]; wave_sync(); }
      }
      H[(size_t)(4 + x * 8 + a) * n + (4 + y * 8 + c)] = out;
      return;
    }
    tile -= nf2;
    if (wv != 0) return;
    if (tile < nf) {
      // frame-calib column block and b segment of frame x
      const int x = tile;
      // lane -> (a, c4) for c4 < 4 : H[x-rows, calib cols]; lanes with c in 4..7: c==4 -> b
      double hv = 0;
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
      for (int p = 0; p < nf2; p++) s += acc13(acc + (size_t)p * 91, r, col);
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
  if (tile < nf2) {
    const int x = tile % nf, y = tile / nf;
    double out = 0;
    auto prod = [&](const double* Lm, int li, int lj, int di, int dj, int dk, const double* Rm, int ri, int rj) {
      sL[lane] = Lm[(size_t)(li + nf * lj) * 64 + lane];
      sR[lane] = Rm[(size_t)(ri + nf * rj) * 64 + lane];
      sX[lane] = (double)accD[(size_t)(di + nf * dj + nf2 * dk) * 64 + lane];
      wave_sync();
      return lxr(W, a, c);
    };
    if (x == y)
      for (int j = 0; j < nf; j++)
        for (int k = 0; k < nf; k++) if (mine()) out += prod(adH, x, j, x, j, k, adH, x, k);      // H[i,i] += AH_ij D_ijk AH_ik^T
    for (int i = 0; i < nf; i++) if (mine()) out += prod(adT, i, x, i, x, y, adT, i, y);          // H[j,k] += AT_ij D_ijk AT_ik^T  (j=x,k=y)
    for (int k = 0; k < nf; k++) if (mine()) out += prod(adT, y, x, y, x, k, adH, y, k);          // H[j,i] += AT_ij D_ijk AH_ik^T  (j=x,i=y)
    for (int j = 0; j < nf; j++) if (mine()) out += prod(adH, x, j, x, j, y, adT, x, y);          // H[i,k] += AH_ij D_ijk AT_ik^T  (i=x,k=y)
    out = fold(out);
    if (wv == 0) H[(size_t)(4 + x * 8 + a) * n + (4 + y * 8 + c)] = out;
    return;
  }
  tile -= nf2;
  if (wv != 0) return;
  if (tile < nf) {
    const int x = tile;
    double hv = 0;
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
// One 256-thread workgroup per window; the (8nf+4)^2 system lives in LDS (row stride n+1: conflict-free
// column walks).  LDL^T with symmetric pivoting on the largest remaining |diagonal| (the strategy of
// Eigen::LDLT, EnergyFunctional.cpp:976), same arithmetic per element as the host reference; the
// rank-1 updates are spread over a 16x16 thread grid and keep the matrix symmetric in place, the
// triangular solves run on wave 0 with the unknowns in registers.
__global__ __launch_bounds__(BA_BLOCK) void k_ba_solve(const BaDev* __restrict__ wins, double lambda, int orthogonalize_x) {
  const BaDev& B = wins[blockIdx.y];
  const int n = B.n, nf = B.nf, ld = n + 1;
  extern __shared__ double sm[];
  double* A = sm;                 // n*(n+1)
  double* bF = A + n * ld;        // n
  double* sv = bF + n;            // n  SVecI
  double* yv = sv + n;            // n
  double* Dg = yv + n;            // n
  double* xv = Dg + n;            // n
  int* perm = (int*)(xv + n);     // n
  __shared__ int s_p;
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
  for (int e = tid; e < n * n; e += BA_BLOCK) {
    const int i = e / n, j = e - i * n;
    double v = HL[e] + B.t_HM[e] + HA[e];
    lastHS[e] = v - HS[e];                   // :909
    if (i == j) v *= (1 + lambda);           // :914-916
    v -= HS[e] * f;                          // :918
    A[i * ld + j] = v;
  }
  __syncthreads();
  for (int i = tid; i < n; i += BA_BLOCK) sv[i] = 1.0 / sqrt(A[i * ld + i] + 10);   // :967
  __syncthreads();
  for (int e = tid; e < n * n; e += BA_BLOCK) { const int i = e / n, j = e - i * n; A[i * ld + j] = sv[i] * A[i * ld + j] * sv[j]; }
  for (int i = tid; i < n; i += BA_BLOCK) { bF[i] = sv[i] * bF[i]; perm[i] = i; }
  __syncthreads();

  const int ti = tid >> 4, tj = tid & 15;
  for (int k = 0; k < n; k++) {
    if (wv == 0) {  // pivot: first index with the largest |A(i,i)|, i >= k
      double best = -1.0; int p = k;
      for (int i = k + lane; i < n; i += 64) { const double v = fabs(A[i * ld + i]); if (v > best) { best = v; p = i; } }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o, 64); const int op = __shfl_xor(p, o, 64);
        if (ob > best || (ob == best && op < p)) { best = ob; p = op; }
      }
      if (lane == 0) s_p = p;
    }
    __syncthreads();
    const int p = s_p;
    if (p != k) {
      if (tid < n) { const double t = A[k * ld + tid]; A[k * ld + tid] = A[p * ld + tid]; A[p * ld + tid] = t; }
      __syncthreads();
      if (tid < n) { const double t = A[tid * ld + k]; A[tid * ld + k] = A[tid * ld + p]; A[tid * ld + p] = t; }
      if (tid == 0) { const int t = perm[k]; perm[k] = perm[p]; perm[p] = t; }
      __syncthreads();
    }
    const double dk = A[k * ld + k];
    if (tid == 0) Dg[k] = dk;
    if (dk == 0.0) {
      for (int i = k + 1 + tid; i < n; i += BA_BLOCK) A[i * ld + k] = 0;
      __syncthreads();
      continue;
    }
    for (int i = k + 1 + tid; i < n; i += BA_BLOCK) A[i * ld + k] = A[i * ld + k] / dk;
    __syncthreads();
    for (int i = k + 1 + ti; i < n; i += 16) {
      const double lik = A[i * ld + k];
      if (lik == 0.0) continue;
      const double ld_ = lik * dk;
      for (int j = k + 1 + tj; j <= i; j += 16) {
        const double v = A[i * ld + j] - ld_ * A[j * ld + k];
        A[i * ld + j] = v;
        A[j * ld + i] = v;   // keep the trailing block symmetric (the reference mirrors after every step)
      }
    }
    __syncthreads();
  }
  if (wv == 0) {  // triangular solves: lane owns unknowns lane and lane+64
    const int i0 = lane, i1 = lane + 64;
    double y0 = i0 < n ? bF[perm[i0]] : 0.0, y1 = i1 < n ? bF[perm[i1]] : 0.0;
    for (int j = 0; j < n; j++) {  // forward, column sweep: y[i] -= L(i,j) y[j], i > j  (j ascending like the reference)
      const double yj = __shfl(j < 64 ? y0 : y1, j & 63, 64);
      if (i0 > j && i0 < n) y0 -= A[i0 * ld + j] * yj;
      if (i1 > j && i1 < n) y1 -= A[i1 * ld + j] * yj;
    }
    if (i0 < n) y0 = Dg[i0] != 0.0 ? y0 / Dg[i0] : 0.0;
    if (i1 < n) y1 = Dg[i1] != 0.0 ? y1 / Dg[i1] : 0.0;
    for (int j = n - 1; j >= 0; j--) {  // backward, column sweep: y[i] -= L(j,i) y[j], i < j
      const double yj = __shfl(j < 64 ? y0 : y1, j & 63, 64);
      if (i0 < j) y0 -= A[j * ld + i0] * yj;
      if (i1 < j && i1 < n) y1 -= A[j * ld + i1] * yj;
    }
    if (i0 < n) xv[perm[i0]] = y0;
    if (i1 < n) xv[perm[i1]] = y1;
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
}

// ------------------------------------------------------------------ back-substitution, one lane per point
__global__ __launch_bounds__(BA_BLOCK) void k_ba_resub(const BaDev* __restrict__ wins) {
  const BaDev& B = wins[blockIdx.y];
  const int p = blockIdx.x * BA_BLOCK + threadIdx.x;
  if (p >= B.np) return;
  float* po = B.p_out + (size_t)p * 16;
  const double* x = B.sol + 3 * ((size_t)B.n * B.n + B.n);
  const int nf = B.nf;
  const float* recs = B.r_rec + (size_t)p * nf * 16;
  int ngood = 0;
  for (int t = 0; t < nf; t++) if (((int)recs[t * 16 + RR_FLAGS]) & 1) ngood++;
  if (ngood == 0) { po[PO_STEP] = 0; return; }
  float b = po[PO_BDSUM];
  float d = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) d += (float)x[k] * (po[PO_HCD_A + k] + po[PO_HCD_L + k]);
  b -= d;
  const int h = B.p_host[p];
  for (int t = 0; t < nf; t++) {       // residuals in target order
    const float* rec = recs + t * 16;
    if (!(((int)rec[RR_FLAGS]) & 1)) continue;
    const float* xa = B.t_xAd + (size_t)(h * nf + t) * 8;
    float sacc = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) sacc += xa[k] * rec[k];
    b -= sacc;
  }
  po[PO_STEP] = -b * po[PO_HDI];
}

// FullSystem::backupState / doStepFromBackup / loadSateBackup for the points.  op: 0 backup, 1 step, 2 restore
__global__ __launch_bounds__(BA_BLOCK) void k_ba_points_op(const BaDev* __restrict__ wins, int op, float stepfacD, float* __restrict__ sums /* per block: sumID, sumNID */) {
  const BaDev& B = wins[blockIdx.y];
  const int p = blockIdx.x * BA_BLOCK + threadIdx.x;
  float sID = 0, sNID = 0;
  if (p < B.np) {
    float* po = B.p_out + (size_t)p * 16;
    float4 g = B.p_geo[p];
    if (op == 0) po[PO_BACKUP] = g.z;
    else if (op == 1) {
      const float st = po[PO_STEP], bk = po[PO_BACKUP];
      const float nid = bk + stepfacD * st;
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
      sums[blockIdx.x * 2] = s0; sums[blockIdx.x * 2 + 1] = s1;
    }
  }
}

}  // namespace sdso
