// Calibration of rocprofv3 FETCH_SIZE on gfx950 for the access patterns of this repository
// (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern"):
//   k_stream   : coalesced 16 B/lane streaming read of the whole buffer        (known: FETCH_SIZE reads 1/2)
//   k_gather1  : every lane one random 16-B element                            (one line/sector per lane)
//   k_gather4  : every lane the 4 bilinear taps (x,y),(x+1,y),(x,y+1),(x+1,y+1) of a random pixel of a 1232-wide float4 image
// Build: hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o gpurun_out/calib_fetch ; run under rocprofv3 --pmc FETCH_SIZE
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__global__ void k_stream(const float4* __restrict__ a, size_t n, float* out) {
  float s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 q = a[i]; s += q.x + q.w; }
  if (s == 12345.f) out[0] = s;
}
__global__ void k_gather1(const float4* __restrict__ a, uint32_t nelem, float* out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const float4 q = a[hash32(t * 2654435761u + 17) % nelem];
  if (q.x == 12345.f) out[0] = q.y;
}
__global__ void k_gather4(const float4* __restrict__ a, uint32_t nimg, float* out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const int W = 1232, H = 368;
  const uint32_t h = hash32(t * 2654435761u + 99);
  const size_t img = (size_t)(h % nimg) * W * H;
  const uint32_t h2 = hash32(h + 7);
  const int x = 2 + (h2 % (W - 5)), y = 2 + ((h2 >> 12) % (H - 5));
  const float4* p = a + img + (size_t)y * W + x;
  const float4 q0 = p[0], q1 = p[1], q2 = p[W], q3 = p[W + 1];
  if (q0.x + q1.x + q2.x + q3.x == 12345.f) out[0] = q0.y;
}
int main() {
  const size_t bytes = 4ull << 30;  // 4 GiB >> 256 MiB MALL
  float4* a; float* out;
  hipMalloc(&a, bytes); hipMalloc(&out, 64);
  hipMemset(a, 0, bytes);
  const size_t n = bytes / 16;
  const uint32_t lanes = 16u << 20;
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_stream, dim3(8192), dim3(256), 0, 0, a, n, out);
    hipLaunchKernelGGL(k_gather1, dim3(lanes / 256), dim3(256), 0, 0, a, (uint32_t)n, out);
    hipLaunchKernelGGL(k_gather4, dim3(lanes / 256), dim3(256), 0, 0, a, (uint32_t)(n / (1232 * 368)), out);
  }
  hipDeviceSynchronize();
  printf("stream bytes %zu ; gather1 lanes %u x 16 B = %zu B useful ; gather4 lanes %u x 64 B = %zu B useful\n", bytes, lanes, (size_t)lanes * 16, lanes, (size_t)lanes * 64);
  return 0;
}
