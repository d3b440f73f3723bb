// What does the memory system deliver for the access MIX of k_ba_lin_fused, with no arithmetic in the way?
// Per lane: L random 128-B lines of a 4 GiB buffer with T 16-byte taps from each (all issued before the first use),
// then G coalesced non-temporal float4 stores (SoA over lanes, like the Jacobian record).  Timed with HIP events.
//   hipcc --offload-arch=gfx950 -O3 tools/mix_bw.hip -o gpurun_out/mix_bw && gpurun_out/mix_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int L, int T, int G>
__global__ __launch_bounds__(256) void k_mix(const float4* __restrict__ a, uint32_t nlines, f4* __restrict__ rec, uint32_t nl, float* out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  float4 q[L * T + 1];
#pragma unroll
  for (int l = 0; l < L; l++) {
    const uint32_t line = hash32(t * 2654435761u + 17 * l + 3) % nlines;
#pragma unroll
    for (int k = 0; k < T; k++) q[l * T + k] = a[(size_t)line * 8 + ((k * 3 + l) & 7)];
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < L * T; i++) s += q[i].x + q[i].y;
#pragma unroll
  for (int g = 0; g < G; g++) __builtin_nontemporal_store((f4){s, s + g, s, s}, rec + (size_t)g * nl + t);
  if (s == 12345.f) out[0] = s;
}
// the same taps with the T taps of a line spread over T neighbouring LANES instead of T instructions: every load instruction
// then touches 64 / T distinct lines.  A "unit" (the work of one lane of k_mix) is done by T lanes; L instructions per lane.
template <int L, int T>
__global__ __launch_bounds__(256) void k_mix_lanes(const float4* __restrict__ a, uint32_t nlines, float* out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t unit = t / T, sub = t % T;
  float4 q[L + 1];
#pragma unroll
  for (int l = 0; l < L; l++) {
    const uint32_t line = hash32(unit * 2654435761u + 17 * l + 3) % nlines;
    q[l] = a[(size_t)line * 8 + ((sub * 3 + l) & 7)];
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < L; i++) s += q[i].x + q[i].y;
  if (s == 12345.f) out[0] = s;
}
// hybrid: the unit's L*T taps are FETCHED with the taps-on-lanes mapping (T neighbouring lanes share a line), parked in LDS, and
// then consumed by the unit's own lane — what k_ba_lin_fused would do to keep its per-residual arithmetic on one lane.
// One wave per workgroup; the wave's 64 units x L*T taps go through LDS in rounds of 16 loads per lane.
template <int L, int T, int G>
__global__ __launch_bounds__(64) void k_mix_lds(const float4* __restrict__ a, uint32_t nlines, f4* __restrict__ rec, uint32_t nl, float* out) {
  constexpr int NT = L * T;                       // taps per unit
  __shared__ float stage[64 * NT * 3];            // 12 B per tap kept (x, y, z)
  const uint32_t lane = threadIdx.x, ubase = blockIdx.x * 64;
  constexpr int ROUNDS = NT;                      // 64 * NT taps / 64 lanes
#pragma unroll
  for (int r0 = 0; r0 < ROUNDS; r0 += 16) {
    float4 q[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const uint32_t g = (r0 + r) * 64 + lane, unit = g / NT, tap = g % NT, l = tap / T, seg = tap % T;
      const uint32_t line = hash32((ubase + unit) * 2654435761u + 17 * l + 3) % nlines;
      q[r] = a[(size_t)line * 8 + ((seg * 3 + l) & 7)];
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const uint32_t g = (r0 + r) * 64 + lane;
      stage[g * 3] = q[r].x; stage[g * 3 + 1] = q[r].y; stage[g * 3 + 2] = q[r].z;
    }
  }
  __syncthreads();
  float s = 0;
#pragma unroll
  for (int i = 0; i < NT; i++) s += stage[(lane * NT + i) * 3] + stage[(lane * NT + i) * 3 + 1];
  const uint32_t t = ubase + lane;
#pragma unroll
  for (int g = 0; g < G; g++) __builtin_nontemporal_store((f4){s, s + g, s, s}, rec + (size_t)g * nl + t);
  if (s == 12345.f) out[0] = s;
}
template <int L, int T, int G>
static void run_lds(const char* name, const float4* a, uint32_t nlines, f4* rec, uint32_t lanes, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_mix_lds<L, T, G>), dim3(lanes / 64), dim3(64), 0, 0, a, nlines, rec, lanes, out);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k_mix_lds<L, T, G>), dim3(lanes / 64), dim3(64), 0, 0, a, nlines, rec, lanes, out);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double rd = (double)lanes * L * 128, wr = (double)lanes * G * 16;
  printf("%-34s lines/unit %d taps/line %d stores %d (via LDS): %.3f ms  %.2f TB/s (lines), %.1f M units/s\n", name, L, T, G, ms, (rd + wr) / (ms * 1e-3) / 1e12,
         lanes / (ms * 1e-3) / 1e6);
}
template <int L, int T>
static void run_lanes(const char* name, const float4* a, uint32_t nlines, uint32_t units, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const uint32_t lanes = units * T;
  hipLaunchKernelGGL((k_mix_lanes<L, T>), dim3(lanes / 256), dim3(256), 0, 0, a, nlines, out);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k_mix_lanes<L, T>), dim3(lanes / 256), dim3(256), 0, 0, a, nlines, out);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double rd = (double)units * L * 128;
  printf("%-34s lines/unit %d taps/line %d (by lanes) : %.3f ms  read %.2f GB = %.2f TB/s (lines), %.1f M units/s\n", name, L, T, ms, rd / 1e9,
         rd / (ms * 1e-3) / 1e12, units / (ms * 1e-3) / 1e6);
}
template <int L, int T, int G>
static void run(const char* name, const float4* a, uint32_t nlines, f4* rec, uint32_t lanes, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_mix<L, T, G>), dim3(lanes / 256), dim3(256), 0, 0, a, nlines, rec, lanes, out);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k_mix<L, T, G>), dim3(lanes / 256), dim3(256), 0, 0, a, nlines, rec, lanes, out);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double rd = (double)lanes * L * 128, wr = (double)lanes * G * 16;
  printf("%-34s lines/lane %d taps/line %d stores %d : %.3f ms  read %.2f GB + write %.2f GB = %.2f TB/s (lines), %.1f M lanes/s\n", name, L, T, G, ms, rd / 1e9,
         wr / 1e9, (rd + wr) / (ms * 1e-3) / 1e12, lanes / (ms * 1e-3) / 1e6);
}
int main() {
  const size_t bytes = 4ull << 30;
  float4* a; f4* rec; float* out;
  const uint32_t lanes = 4u << 20;
  hipMalloc(&a, bytes); hipMalloc(&rec, (size_t)lanes * 19 * 16); hipMalloc(&out, 64);
  hipMemset(a, 0, bytes);
  const uint32_t nlines = (uint32_t)(bytes / 128);
  run<8, 1, 0>("8 lines, 1 tap each, no stores", a, nlines, rec, lanes, out);
  run<8, 4, 0>("8 lines, 4 taps each, no stores", a, nlines, rec, lanes, out);
  run<8, 4, 19>("8 lines, 4 taps each, 19 stores", a, nlines, rec, lanes, out);
  run<8, 2, 19>("8 lines, 2 taps each, 19 stores", a, nlines, rec, lanes, out);
  run<0, 1, 19>("stores only", a, nlines, rec, lanes, out);
  run<4, 4, 19>("4 lines, 4 taps each, 19 stores", a, nlines, rec, lanes, out);
  run_lds<8, 4, 0>("8x4 gathered by lanes, used by unit", a, nlines, rec, lanes, out);
  run_lds<8, 4, 19>("the same + 19 stores", a, nlines, rec, lanes, out);
  run_lanes<8, 4>("8 lines x 4 taps, taps on lanes", a, nlines, lanes, out);
  run_lanes<8, 2>("8 lines x 2 taps, taps on lanes", a, nlines, lanes, out);
  run_lanes<16, 2>("16 lines x 2 taps, taps on lanes", a, nlines, lanes, out);
  return 0;
}
