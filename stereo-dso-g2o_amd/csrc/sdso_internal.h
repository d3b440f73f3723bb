// Internal definitions of libsdso_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/sdso_abi.h"

namespace sdso {

// ---- constants of the reference (values @ file:line in SURVEY.md Appendix A)
#define SCALE_IDEPTH 1.0f
#define SCALE_XI_ROT 1.0f
#define SCALE_XI_TRANS 0.5f
#define SCALE_F 50.0f
#define SCALE_C 50.0f
#define SCALE_A 10.0f
#define SCALE_B 1000.0f
constexpr float kHuberTH = 9.0f;                 // settings.cpp:95
constexpr float kOutlierTHSumComponent = 2500.f; // settings.cpp:73

// A pyramid level on the device: one float4 {I, dx, dy, 0} per pixel (16-byte taps).
struct PyramidDev {
  int levels = 0;
  int w[SDSO_PYR_LEVELS] = {0}, h[SDSO_PYR_LEVELS] = {0};
  float4* d[SDSO_PYR_LEVELS] = {nullptr};
  // level 0 once more in 4x2-pixel tiles (one 128-B line per tile) for the BA linearisation, built on first use
  float4* tiled0 = nullptr;
  bool tiled_ok = false;
  // level-0 intensities alone (4 B per pixel) for the discrete epipolar search, which samples I only; built on first use
  float* plane0 = nullptr;
  bool plane_ok = false;
};
// pc_* of one reference keyframe: one float4 {u, v, idepth, color} per template point.
struct RefDev {
  int n[SDSO_PYR_LEVELS] = {0};
  float4* pc[SDSO_PYR_LEVELS] = {nullptr};
};

struct ProfEntry {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;  // pending (start, stop) pairs
  double total_ms = 0;
  long launches = 0;
};

struct BaWindowDev;  // ba.hip
struct TrackBatch;   // tracker.hip

}  // namespace sdso

struct sdso_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  std::map<int, sdso::PyramidDev> pyr;
  std::map<int, sdso::RefDev> refs;
  std::map<int, sdso::BaWindowDev*> wins;
  sdso::TrackBatch* tb = nullptr;
  // generic scratch
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  // k_track_lm's cluster records: written by that kernel only, with tags that grow from call to call (no clearing between calls)
  void* lm_clusters = nullptr;
  size_t lm_clusters_bytes = 0;
  int lm_epoch = 0;
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
  int n_cu = 256;
  // CU partition (sdso_ctx_partition_cus): `stream` is masked to the large share of the CUs, `aux` to the small one; inside a batch's
  // resident GN loop the linearisation runs on `stream`, the Schur accumulation and the fused tail kernel on `aux` — next to the
  // linearisation of ANOTHER ctx's batch whose stream carries the same large mask.  aux == nullptr: no partition (everything on `stream`).
  hipStream_t aux = nullptr;
  hipEvent_t ev_main = nullptr, ev_aux = nullptr;
  int aux_cus = 0;
  int gn_mode = 0;   // traceStereo refinement: 0 DSO-native, 1 fork-live g2o GN (sdso_trace_set_gn_mode)
  float* gammaB = nullptr;   // device copy of CalibHessian::B (256 floats) for the gamma-weighted absSquaredGrad; null = identity response
  // device buffers of released BA windows, kept for the next upload (a window is re-uploaded for every keyframe)
  std::vector<std::pair<void*, size_t>> ba_pool;
  // optional in-library kernel timing (HIP events on ctx->stream), see sdso_prof_*
  int prof_on = 0;           // 0 off; 1 the dominant kernel of each workload; 2 every bracketed kernel (each bracket is two hipEventRecord on the
                             //   stream: ~5 us of queue time apiece, which a timed loop should not pay for kernels it does not report)
  std::map<std::string, sdso::ProfEntry> prof;
};

namespace sdso {

// Per-context state kept outside sdso_ctx lives in file-local registries keyed by the ctx.  A ctx is used by one thread at a
// time, but different contexts may be driven from different threads (tracking / mapping), so the registries themselves are
// guarded; the mapped objects are only touched by their ctx's thread (std::map nodes are stable).
inline std::mutex& registry_mutex() { static std::mutex m; return m; }
template <class V> inline V& reg_get(std::map<sdso_ctx*, V>& m, sdso_ctx* ctx) { std::lock_guard<std::mutex> g(registry_mutex()); return m[ctx]; }
template <class V> inline bool reg_has(std::map<sdso_ctx*, V>& m, sdso_ctx* ctx) { std::lock_guard<std::mutex> g(registry_mutex()); return m.count(ctx) != 0; }
template <class V> inline bool reg_take(std::map<sdso_ctx*, V>& m, sdso_ctx* ctx, V& out) {
  std::lock_guard<std::mutex> g(registry_mutex());
  auto it = m.find(ctx);
  if (it == m.end()) return false;
  out = it->second;
  m.erase(it);
  return true;
}

inline int fail(sdso_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  return code;
}

#define SDSO_HIP(ctx, expr)                                                                  \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return sdso::fail(ctx, SDSO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

#define SDSO_REQUIRE(ctx, cond, msg) \
  do {                               \
    if (!(cond)) return sdso::fail(ctx, SDSO_ERR_ARG, msg); \
  } while (0)

int ensure_scratch(sdso_ctx* ctx, size_t bytes);
// Bracket the launches of one named kernel with HIP events when profiling is enabled.
// The A/B and diagnostic switches of the library (SDSO_BA_*, SDSO_TRK_*, SDSO_OPT_*, SDSO_PROF_BRACKET) exist only under SDSO_DEBUG_ENV=1,
// which is read ONCE per process: without it no environment variable changes what the library computes or launches, and no call reads
// the environment.  (tests/conftest.py sets the gate: the variant tests flip switches per call; bench.py does not.)
inline const char* dbg_env(const char* name) {
  static const bool on = [] { const char* g = getenv("SDSO_DEBUG_ENV"); return g && atoi(g) != 0; }();
  return on ? getenv(name) : nullptr;
}
struct ProfScope {
  sdso_ctx* ctx;
  hipEvent_t a = nullptr, b = nullptr;
  const char* name;
  ProfScope(sdso_ctx* c, const char* n, int level = 1) : ctx(c), name(n) {
    if (ctx->prof_on < level) return;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, ctx->stream);
  }
  ~ProfScope() {
    if (!a) return;
    hipEventRecord(b, ctx->stream);
    ctx->prof[name].ev.emplace_back(a, b);
  }
};
// ONE kernel launch, timed exactly when profiling is enabled: hipExtLaunchKernelGGL ties the two events to the start and the end of
// the dispatch itself (the timestamps rocprofv3 reports), where a pair of hipEventRecord calls around the launch also counts the
// dispatch latency after the first record (≈ 10 us on a 390-us kernel: bench.py's figure sat 2.5 % above rocprofv3's).
// SDSO_PROF_BRACKET=1: the old record / launch / record bracket.
template <typename K, typename... A>
inline void launch_timed(sdso_ctx* ctx, const char* name, int level, K kernel, const dim3& grid, const dim3& block, A... args) {
  static const bool bracket = dbg_env("SDSO_PROF_BRACKET") != nullptr;
  if (ctx->prof_on >= level && !bracket) {
    hipEvent_t ea = nullptr, eb = nullptr;
    hipEventCreate(&ea); hipEventCreate(&eb);
    hipExtLaunchKernelGGL(kernel, grid, block, 0, ctx->stream, ea, eb, 0, args...);
    ctx->prof[name].ev.emplace_back(ea, eb);
    return;
  }
  ProfScope ps(ctx, name, level);
  hipLaunchKernelGGL(kernel, grid, block, 0, ctx->stream, args...);
}
int ensure_pinned(sdso_ctx* ctx, size_t bytes);
int ensure_tiled0(sdso_ctx* ctx, PyramidDev& P);   // ctx.hip
int ensure_plane0(sdso_ctx* ctx, PyramidDev& P);   // ctx.hip
// pixel (x, y) of a 4x2-tiled level-0 image with T tiles per row
__host__ __device__ inline int tiled_index(int x, int y, int T) { return (((y >> 1) * T + (x >> 2)) << 3) + ((y & 1) << 2) + (x & 3); }

// ------------------------------------------------------------------ device helpers
// getInterpolatedElement33 (src/util/globalFuncs.h:73-86) on the float4 image.
// Same operation order as the reference so that per-point results are bit-identical to the CPU.
__device__ __forceinline__ float3 interp33(const float4* __restrict__ img, float x, float y, int width) {
  const int ix = (int)x;
  const int iy = (int)y;
  const float dx = x - ix;
  const float dy = y - iy;
  const float dxdy = dx * dy;
  const float4* bp = img + ix + iy * width;
  const float4 p00 = bp[0], p10 = bp[1], p01 = bp[width], p11 = bp[1 + width];
  const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
  float3 r;
  r.x = w11 * p11.x + w01 * p01.x + w10 * p10.x + w00 * p00.x;
  r.y = w11 * p11.y + w01 * p01.y + w10 * p10.y + w00 * p00.y;
  r.z = w11 * p11.z + w01 * p01.z + w10 * p10.z + w00 * p00.z;
  return r;
}
// getInterpolatedElement31 (globalFuncs.h:122-135)
__device__ __forceinline__ float interp31(const float4* __restrict__ img, float x, float y, int width) {
  const int ix = (int)x;
  const int iy = (int)y;
  const float dx = x - ix;
  const float dy = y - iy;
  const float dxdy = dx * dy;
  const float4* bp = img + ix + iy * width;
  return dxdy * bp[1 + width].x + (dy - dxdy) * bp[width].x + (dx - dxdy) * bp[1].x + (1 - dx - dy + dxdy) * bp[0].x;
}

// getInterpolatedElement31 on the intensity plane (same expression, a quarter of the bytes per tap)
__device__ __forceinline__ float interp31_plane(const float* __restrict__ I, float x, float y, int width) {
  const int ix = (int)x;
  const int iy = (int)y;
  const float dx = x - ix;
  const float dy = y - iy;
  const float dxdy = dx * dy;
  const float* bp = I + ix + iy * width;
  return dxdy * bp[1 + width] + (dy - dxdy) * bp[width] + (dx - dxdy) * bp[1] + (1 - dx - dy + dxdy) * bp[0];
}

// value of a compile-time-constant lane in every lane (v_readlane_b32: scalar broadcast, no LDS crossbar trip)
__device__ __forceinline__ float lane_bcast(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ int lane_bcast(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// 64-lane butterfly sum
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}


// Sum N (multiple of 4) per-lane values over the 64 lanes of a wave with gfx950's half/row swaps instead of one
// 6-step butterfly per value: v_permlane32_swap exchanges the upper half of one register with the lower half of
// another, so one swap + one add finishes the 32-lane step for TWO values; v_permlane16_swap does the same for the
// 16-lane rows.  N values -> N/4 registers, each then folded inside its rows by four DPP row rotations:
// 2.5 instructions per value instead of 12.  emit(k, sum) is called by one lane per value.
template <int N, class Emit>
__device__ __forceinline__ void wave_reduce_rows(const float* v, Emit emit) {
  static_assert(N % 4 == 0, "N must be a multiple of 4");
  const int lane = threadIdx.x & 63;
  float z[N / 4];
#pragma unroll
  for (int m = 0; m < N / 4; m++) {
    float x[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[4 * m + 2 * h]), __float_as_uint(v[4 * m + 2 * h + 1]), false, false);
      x[h] = __uint_as_float(r[0]) + __uint_as_float(r[1]);   // rows 0,1: v[4m+2h] (32-lane partials), rows 2,3: v[4m+2h+1]
    }
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[0]), __float_as_uint(x[1]), false, false);
    float t = __uint_as_float(r[0]) + __uint_as_float(r[1]);  // rows: v[4m], v[4m+2], v[4m+1], v[4m+3] (16-lane partials)
    t += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(t), 0x128, 0xf, 0xf, false));   // row_ror:8
    t += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(t), 0x124, 0xf, 0xf, false));   // row_ror:4
    t += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(t), 0x122, 0xf, 0xf, false));   // row_ror:2
    t += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(t), 0x121, 0xf, 0xf, false));   // row_ror:1
    z[m] = t;
  }
  if ((lane & 15) == 0) {
    const int row = lane >> 4;
    const int sub = row == 0 ? 0 : row == 1 ? 2 : row == 2 ? 1 : 3;
#pragma unroll
    for (int m = 0; m < N / 4; m++) emit(4 * m + sub, z[m]);
  }
}

}  // namespace sdso
