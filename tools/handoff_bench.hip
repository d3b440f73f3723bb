// One-way latency of a flag hand-off between two workgroups on different CUs (agent-scope release store -> acquire load), same XCD
// (workgroups 0 and 8 of a 16-workgroup grid) and different XCDs (0 and 1): hipcc --offload-arch=gfx950 -O3 tools/handoff_bench.hip -o tools/handoff_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_pingpong(int* flags, int partner_a, int partner_b, int rounds, unsigned long long* ticks, float* payload) {
  const int b = blockIdx.x;
  if (b != partner_a && b != partner_b) return;
  if (threadIdx.x != 0) return;
  const bool first = b == partner_a;
  int* mine = flags + (first ? 0 : 64);
  int* theirs = flags + (first ? 64 : 0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 1; r <= rounds; r++) {
    if (first) {
      payload[r & 63] = (float)r;                                  // data the flag publishes
      __hip_atomic_store(theirs, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < r && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
    } else {
      int spins = 0;
      while (__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < r && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
      payload[64 + (r & 63)] = payload[r & 63] + 1.f;
      __hip_atomic_store(theirs, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (first) ticks[0] = __builtin_amdgcn_s_memtime() - t0;
}
int main() {
  int* flags; unsigned long long* ticks; float* payload;
  hipMalloc(&flags, 1024); hipMalloc(&ticks, 64); hipMalloc(&payload, 1024);
  const int rounds = 2000;
  for (int pass = 0; pass < 2; pass++)
    for (int pb : {8, 1, 16, 4}) {
      hipMemset(flags, 0, 1024);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_pingpong, dim3(32), dim3(64), 0, 0, flags, 0, pb, rounds, ticks, payload);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
      printf("workgroups 0 <-> %2d: %d round trips in %.3f ms = %.0f ns one way (%.0f shader ticks per round trip)\n", pb, rounds, ms, ms * 1e6 / rounds / 2, (double)t / rounds);
    }
  return 0;
}
