// Stand-alone check + timing of the register-resident LDL^T (stereo-dso-g2o_amd/csrc/ba_ldlt.h) on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ldlt_bench.hip -o tools/ldlt_bench.bin && tools/ldlt_bench.bin
// Random SPD systems of size n (default 68, 60, 36, 12): the device solution against a host long-double Cholesky; then the time of a
// launch of W independent systems (one wave each).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../stereo-dso-g2o_amd/csrc/ba_ldlt.h"
#include "ba_ldlt4.h"
using namespace sdso;

__global__ __launch_bounds__(64) void k_ldlt(const double* __restrict__ Ms, const double* __restrict__ bs, double* __restrict__ xs, int n, long long* ticks) {
  __shared__ double As[LDLT_NMAX * LDLT_LD], Lt[64 * LDLT_LD], bp[LDLT_NMAX], xp[LDLT_NMAX], dg[LDLT_NMAX];
  __shared__ __attribute__((aligned(16))) double colb[64];
  __shared__ int pos[LDLT_NMAX], perm[LDLT_NMAX];
  __shared__ unsigned long long keys[LDLT_NMAX];
  const double* M = Ms + (size_t)blockIdx.x * n * n;
  const double* b = bs + (size_t)blockIdx.x * n;
  const int lane = threadIdx.x;
  for (int i = lane; i < LDLT_NMAX * LDLT_LD; i += 64) As[i] = 0.0;
  for (int i = lane; i < LDLT_NMAX; i += 64) { bp[i] = 0.0; dg[i] = i < n ? M[i * n + i] : 0.0; }
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = lane; i < n; i += 64) {     // (ldlt_pivot_rank wants n threads; this kernel has one wave)
    int r = 0;
    for (int j = 0; j < n; j++) r += (fabs(dg[j]) > fabs(dg[i]) || (fabs(dg[j]) == fabs(dg[i]) && j < i)) ? 1 : 0;
    pos[i] = r;
  }
  __syncthreads();
  const long long t1 = __builtin_amdgcn_s_memtime();
  for (int e = lane; e < n * n; e += 64) {
    const int i = e / n, j = e % n;
    if (i >= j) { const double v = M[i * n + j]; As[pos[i] * LDLT_LD + pos[j]] = v; As[pos[j] * LDLT_LD + pos[i]] = v; }   // Eigen reads the lower triangle
  }
  for (int i = lane; i < n; i += 64) bp[pos[i]] = b[i];
  __syncthreads();
  const long long t2 = __builtin_amdgcn_s_memtime();
  ldlt_solve_regs(As, bp, Lt, colb, xp, n);
  __syncthreads();
  const long long t3 = __builtin_amdgcn_s_memtime();
  for (int i = lane; i < n; i += 64) xs[(size_t)blockIdx.x * n + i] = xp[pos[i]];
  if (ticks && lane == 0) { ticks[blockIdx.x * 3] = t1 - t0; ticks[blockIdx.x * 3 + 1] = t2 - t1; ticks[blockIdx.x * 3 + 2] = t3 - t2; }
}

// the four-wave panel variant (ba_ldlt4.h): 256 threads, packed lower triangle in, SVecI = 1
__global__ __launch_bounds__(256, 3) void k_ldlt4(const double* __restrict__ Ms, const double* __restrict__ bs, double* __restrict__ xs, int n, long long* ticks) {
  __shared__ double Mp[LP_M_DOUBLES], U[LP_U_DOUBLES], Lp[LP_L_DOUBLES + 8], svv[72], bsv[72], xp[72], dg[72];
  __shared__ LpShared S;
  __shared__ int pos[72], perm[72];
  __shared__ unsigned long long keys[72];
  const double* M = Ms + (size_t)blockIdx.x * n * n;
  const double* b = bs + (size_t)blockIdx.x * n;
  const int tid = threadIdx.x;
  for (int e = tid; e < n * n; e += 256) { const int i = e / n, j = e % n; if (i >= j) Mp[lp_mtri(i, j)] = M[e]; }
  if (tid < 72) { svv[tid] = 1.0; bsv[tid] = tid < n ? b[tid] : 0.0; dg[tid] = tid < n ? M[tid * n + tid] : 0.0; }
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  ldlt_pivot_rank(dg, n, pos, keys);
  if (tid < n) perm[pos[tid]] = tid;
  __syncthreads();
  const long long t1 = __builtin_amdgcn_s_memtime();
  ldlt_solve_panels(Mp, svv, perm, bsv, U, Lp, S, xp, n);
  __syncthreads();
  const long long t2 = __builtin_amdgcn_s_memtime();
  for (int i = tid; i < n; i += 256) xs[(size_t)blockIdx.x * n + i] = xp[pos[i]];
  if (ticks && tid == 0) { ticks[blockIdx.x * 3] = t1 - t0; ticks[blockIdx.x * 3 + 1] = 0; ticks[blockIdx.x * 3 + 2] = t2 - t1; }
}

// the slim one-wave variant (ldlt_solve_regs_packed): packed lower triangle in, factor over it, SVecI = 1; 256-thread workgroup
__global__ __launch_bounds__(256, 3) void k_ldltp(const double* __restrict__ Ms, const double* __restrict__ bs, double* __restrict__ xs, int n, long long* ticks) {
  __shared__ double Mp[LDLT_M_PACKED], svv[72], bsv[72], xp[72], dg[72];
  __shared__ __attribute__((aligned(16))) double colb[80];
  __shared__ int pos[72], perm[72];
  __shared__ unsigned long long keys[72];
  const double* M = Ms + (size_t)blockIdx.x * n * n;
  const double* b = bs + (size_t)blockIdx.x * n;
  const int tid = threadIdx.x;
  for (int e = tid; e < n * n; e += 256) { const int i = e / n, j = e % n; if (i >= j) Mp[ldlt_mtri(i, j)] = M[e]; }
  if (tid < 72) { svv[tid] = 1.0; bsv[tid] = tid < n ? b[tid] : 0.0; dg[tid] = tid < n ? M[tid * n + tid] : 0.0; }
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  ldlt_pivot_rank(dg, n, pos, keys);
  if (tid < n) perm[pos[tid]] = tid;
  __syncthreads();
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (tid < 64) ldlt_solve_regs_packed<8>(Mp, svv, perm, bsv, colb, xp, n);
  __syncthreads();
  const long long t2 = __builtin_amdgcn_s_memtime();
  for (int i = tid; i < n; i += 256) xs[(size_t)blockIdx.x * n + i] = xp[pos[i]];
  if (ticks && tid == 0) { ticks[blockIdx.x * 3] = t1 - t0; ticks[blockIdx.x * 3 + 1] = 0; ticks[blockIdx.x * 3 + 2] = t2 - t1; }
}

static void host_solve(const std::vector<double>& M, const std::vector<double>& b, int n, std::vector<long double>& x) {
  std::vector<long double> L((size_t)n * n, 0.0L);
  for (int j = 0; j < n; j++) {
    long double s = M[j * n + j];
    for (int k = 0; k < j; k++) s -= L[j * n + k] * L[j * n + k];
    L[j * n + j] = sqrtl(s);
    for (int i = j + 1; i < n; i++) {
      long double t = M[i * n + j];
      for (int k = 0; k < j; k++) t -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = t / L[j * n + j];
    }
  }
  x.assign(n, 0.0L);
  for (int i = 0; i < n; i++) { long double s = b[i]; for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k]; x[i] = s / L[i * n + i]; }
  for (int i = n - 1; i >= 0; i--) { long double s = x[i]; for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k]; x[i] = s / L[i * n + i]; }
}

int main(int argc, char** argv) {
  const int W = argc > 1 ? atoi(argv[1]) : 128;
  int rc = 0;
  for (int n : {68, 60, 36, 12}) {
    std::mt19937_64 rng(1234 + n);
    std::normal_distribution<double> nd;
    std::vector<double> M((size_t)W * n * n), b((size_t)W * n);
    for (int w = 0; w < W; w++) {
      std::vector<double> G((size_t)n * n);
      for (double& v : G) v = nd(rng);
      for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) {
          double s = 0;
          for (int k = 0; k < n; k++) s += G[i * n + k] * G[j * n + k];
          if (i == j) s += 1e-3 * n;
          // ties of the diagonal in some systems (w % 4 == 1): the exact replay of Eigen's exchanges
          M[(size_t)w * n * n + i * n + j] = M[(size_t)w * n * n + j * n + i] = s;
        }
      if (w % 4 == 1) for (int i = 0; i < n; i += 3) M[(size_t)w * n * n + i * n + i] = 2.0 * n;
      if (w % 4 == 2) { const int z = n / 2; for (int j = 0; j < n; j++) M[(size_t)w * n * n + z * n + j] = M[(size_t)w * n * n + j * n + z] = 0.0; }   // a zero row: zero pivot, x = 0 there
      for (int i = 0; i < n; i++) b[(size_t)w * n + i] = nd(rng);
      if (w % 4 == 2) b[(size_t)w * n + n / 2] = 0.0;
    }
    double *dM, *db, *dx; long long* dt;
    hipMalloc(&dM, M.size() * 8); hipMalloc(&db, b.size() * 8); hipMalloc(&dx, b.size() * 8); hipMalloc(&dt, W * 3 * 8);
    hipMemcpy(dM, M.data(), M.size() * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 8, hipMemcpyHostToDevice);
    for (int variant = 0; variant < 3; variant++) {
    if (variant == 0) hipLaunchKernelGGL(k_ldlt, dim3(W), dim3(64), 0, 0, dM, db, dx, n, dt);
    else if (variant == 1) hipLaunchKernelGGL(k_ldlt4, dim3(W), dim3(256), 0, 0, dM, db, dx, n, dt);
    else hipLaunchKernelGGL(k_ldltp, dim3(W), dim3(256), 0, 0, dM, db, dx, n, dt);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
    std::vector<double> x(b.size()); std::vector<long long> tk(W * 3);
    hipMemcpy(x.data(), dx, x.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(tk.data(), dt, tk.size() * 8, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int w = 0; w < W; w++) {
      std::vector<double> Mw(M.begin() + (size_t)w * n * n, M.begin() + (size_t)(w + 1) * n * n), bw(b.begin() + (size_t)w * n, b.begin() + (size_t)(w + 1) * n);
      if (w % 4 == 2) {   // reduced system without the zero row
        const int z = n / 2, m = n - 1;
        std::vector<double> Mr((size_t)m * m), br(m);
        for (int i = 0, ii = 0; i < n; i++) { if (i == z) continue; br[ii] = bw[i]; for (int j = 0, jj = 0; j < n; j++) { if (j == z) continue; Mr[ii * m + jj] = Mw[i * n + j]; jj++; } ii++; }
        std::vector<long double> xr; host_solve(Mr, br, m, xr);
        double xn = 0, en = 0;
        for (int i = 0, ii = 0; i < n; i++) { const long double ref = i == z ? 0.0L : xr[ii++]; xn = fmax(xn, fabs((double)ref)); en = fmax(en, fabs((double)(x[(size_t)w * n + i] - ref))); }
        worst = fmax(worst, en / xn);
      } else {
        std::vector<long double> xr; host_solve(Mw, bw, n, xr);
        double xn = 0, en = 0;
        for (int i = 0; i < n; i++) { xn = fmax(xn, fabs((double)xr[i])); en = fmax(en, fabs((double)(x[(size_t)w * n + i] - xr[i]))); }
        worst = fmax(worst, en / xn);
      }
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 50;
    for (int i = 0; i < 5 + reps; i++) {
      if (i == 5) hipEventRecord(e0, 0);
      if (variant == 0) hipLaunchKernelGGL(k_ldlt, dim3(W), dim3(64), 0, 0, dM, db, dx, n, (long long*)nullptr);
      else if (variant == 1) hipLaunchKernelGGL(k_ldlt4, dim3(W), dim3(256), 0, 0, dM, db, dx, n, (long long*)nullptr);
      else hipLaunchKernelGGL(k_ldltp, dim3(W), dim3(256), 0, 0, dM, db, dx, n, (long long*)nullptr);
    }
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%s n %2d: max rel err %.3e %s | launch of %d systems %.1f us | ticks(shader clock) order %lld assemble %lld solve %lld\n", variant == 0 ? "1 wave, registers" : variant == 1 ? "4 waves, panels " : "1 wave, packed   ", n, worst, worst < 1e-9 ? "ok" : "FAIL", W, ms * 1e3 / reps,
           tk[0], tk[1], tk[2]);
    if (!(worst < 1e-9)) rc = 1;
    }
    hipFree(dM); hipFree(db); hipFree(dx); hipFree(dt);
  }
  return rc;
}
