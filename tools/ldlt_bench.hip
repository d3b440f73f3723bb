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
  ldlt_pivot_order(dg, n, pos, perm, keys);
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
    hipLaunchKernelGGL(k_ldlt, dim3(W), dim3(64), 0, 0, dM, db, dx, n, dt);
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
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k_ldlt, dim3(W), dim3(64), 0, 0, dM, db, dx, n, (long long*)nullptr);
    hipEventRecord(e0, 0);
    const int reps = 50;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_ldlt, dim3(W), dim3(64), 0, 0, dM, db, dx, n, (long long*)nullptr);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("n %2d: max rel err %.3e %s | launch of %d systems %.1f us | ticks(shader clock) order %lld assemble %lld solve %lld\n", n, worst, worst < 1e-9 ? "ok" : "FAIL", W, ms * 1e3 / reps,
           tk[0], tk[1], tk[2]);
    if (!(worst < 1e-9)) rc = 1;
    hipFree(dM); hipFree(db); hipFree(dx); hipFree(dt);
  }
  return rc;
}
