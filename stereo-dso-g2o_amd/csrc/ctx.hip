// Context, device pyramids (FrameHessian::dIp mirrors) and FrameHessian::makeImages on the device.
#include "sdso_internal.h"
#include <cmath>
#include <cstring>

using namespace sdso;

namespace sdso {
int ensure_scratch(sdso_ctx* ctx, size_t bytes) {
  if (ctx->scratch_bytes >= bytes) return SDSO_OK;
  if (ctx->scratch) { SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream)); SDSO_HIP(ctx, hipFree(ctx->scratch)); ctx->scratch = nullptr; ctx->scratch_bytes = 0; }
  size_t want = bytes + bytes / 2 + 4096;
  SDSO_HIP(ctx, hipMalloc(&ctx->scratch, want));
  ctx->scratch_bytes = want;
  return SDSO_OK;
}
int ensure_pinned(sdso_ctx* ctx, size_t bytes) {
  if (ctx->pinned_bytes >= bytes) return SDSO_OK;
  if (ctx->pinned) { SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream)); SDSO_HIP(ctx, hipHostFree(ctx->pinned)); ctx->pinned = nullptr; ctx->pinned_bytes = 0; }
  size_t want = bytes + bytes / 2 + 4096;
  SDSO_HIP(ctx, hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault));
  ctx->pinned_bytes = want;
  return SDSO_OK;
}
}  // namespace sdso

// ------------------------------------------------------------------ kernels
// CalibHessian::getBGradOnly (HessianBlocks.h:356-362) squared: the weight of absSquaredGrad when setting_gammaWeightsPixelSelect == 1
// (HessianBlocks.cpp:194-198).  B == nullptr: identity response, weight 1.
__device__ __forceinline__ float abs_grad(float color, float dx, float dy, const float* __restrict__ B) {
  float a = dx * dx + dy * dy;
  if (B) {
    int c = color + 0.5f;
    if (c < 5) c = 5;
    if (c > 250) c = 250;
    const float gw = B[c + 1] - B[c];
    a *= gw * gw;
  }
  return a;
}
// AoS float3 {I,dx,dy} -> float4 {I,dx,dy,absSquaredGrad}  (HessianBlocks.cpp:192-198)
__global__ void k_expand3to4(const float* __restrict__ src, float4* __restrict__ dst, int npix, const float* __restrict__ B) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npix) { const float dx = src[3 * i + 1], dy = src[3 * i + 2]; dst[i] = make_float4(src[3 * i], dx, dy, abs_grad(src[3 * i], dx, dy, B)); }
}
__global__ void k_pack4to3(const float4* __restrict__ src, float* __restrict__ dst, int npix) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npix) { float4 p = src[i]; dst[3 * i] = p.x; dst[3 * i + 1] = p.y; dst[3 * i + 2] = p.z; }
}
// HessianBlocks.cpp:156-157 — level-0 intensities
__global__ void k_set_level0(const float* __restrict__ color, float4* __restrict__ dst, int npix) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npix) dst[i] = make_float4(color[i], 0.f, 0.f, 0.f);
}
// HessianBlocks.cpp:172-178 — 2x2 box mean of the level below
__global__ void k_downsample(const float4* __restrict__ src, int wsrc, float4* __restrict__ dst, int wl, int hl) {
  int x = blockIdx.x * blockDim.x + threadIdx.x;
  int y = blockIdx.y;
  if (x >= wl || y >= hl) return;
  const float4* r0 = src + 2 * x + 2 * y * wsrc;
  float v = 0.25f * (((r0[0].x + r0[1].x) + r0[wsrc].x) + r0[wsrc + 1].x);
  dst[x + y * wl] = make_float4(v, 0.f, 0.f, 0.f);
}
// HessianBlocks.cpp:182-192 — central differences on rows 1..h-2 (linear index wl .. wl*(hl-1)-1)
__global__ void k_gradients(float4* __restrict__ img, int wl, int hl, const float* __restrict__ B) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x + wl;
  if (idx >= wl * (hl - 1)) return;
  float dx = 0.5f * (img[idx + 1].x - img[idx - 1].x);
  float dy = 0.5f * (img[idx + wl].x - img[idx - wl].x);
  if (!isfinite(dx)) dx = 0;
  if (!isfinite(dy)) dy = 0;
  img[idx].y = dx;
  img[idx].z = dy;
  img[idx].w = abs_grad(img[idx].x, dx, dy, B);   // absSquaredGrad (:192) times the squared response gradient (:194-198)
}

__global__ void k_tile0(const float4* __restrict__ src, float4* __restrict__ dst, int w, int h, int T) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x < w && y < h) dst[tiled_index(x, y, T)] = src[x + y * w];
}
__global__ void k_plane0(const float4* __restrict__ src, float* __restrict__ dst, int npix) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npix) dst[i] = src[i].x;
}
namespace sdso {
int ensure_plane0(sdso_ctx* ctx, PyramidDev& P) {
  if (P.plane_ok) return SDSO_OK;
  const int npix = P.w[0] * P.h[0];
  if (!P.plane0) SDSO_HIP(ctx, hipMalloc(&P.plane0, sizeof(float) * (size_t)npix));
  hipLaunchKernelGGL(k_plane0, dim3((npix + 255) / 256), dim3(256), 0, ctx->stream, (const float4*)P.d[0], P.plane0, npix);
  SDSO_HIP(ctx, hipGetLastError());
  P.plane_ok = true;
  return SDSO_OK;
}
int ensure_tiled0(sdso_ctx* ctx, PyramidDev& P) {
  if (P.tiled_ok) return SDSO_OK;
  const int w = P.w[0], h = P.h[0], T = (w + 3) / 4, Th = (h + 1) / 2;
  if (!P.tiled0) SDSO_HIP(ctx, hipMalloc(&P.tiled0, sizeof(float4) * 8 * (size_t)T * Th));
  hipLaunchKernelGGL(k_tile0, dim3((w + 255) / 256, h), dim3(256), 0, ctx->stream, (const float4*)P.d[0], P.tiled0, w, h, T);
  SDSO_HIP(ctx, hipGetLastError());
  P.tiled_ok = true;
  return SDSO_OK;
}
}  // namespace sdso

// ------------------------------------------------------------------ API
extern "C" int sdso_ctx_create(int device_ordinal, sdso_ctx** out) {
  if (!out) return SDSO_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SDSO_ERR_NODEV;
  if (device_ordinal < 0 || device_ordinal >= ndev) return SDSO_ERR_NODEV;
  if (hipSetDevice(device_ordinal) != hipSuccess) return SDSO_ERR_NODEV;
  sdso_ctx* ctx = new sdso_ctx();
  ctx->device = device_ordinal;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return SDSO_ERR_HIP; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
  *out = ctx;
  return SDSO_OK;
}

namespace sdso {
void release_all_windows(sdso_ctx* ctx);
void release_track_batch(sdso_ctx* ctx);
void release_trace(sdso_ctx* ctx);
void release_match(sdso_ctx* ctx);
void release_selector(sdso_ctx* ctx);
void release_g2o(sdso_ctx* ctx);
void release_comm(sdso_ctx* ctx);
}

extern "C" void sdso_ctx_destroy(sdso_ctx* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  if (ctx->aux) { hipStreamSynchronize(ctx->aux); hipStreamDestroy(ctx->aux); ctx->aux = nullptr; }
  if (ctx->ev_main) hipEventDestroy(ctx->ev_main);
  if (ctx->ev_aux) hipEventDestroy(ctx->ev_aux);
  for (auto& kv : ctx->pyr)
    { for (int l = 0; l < kv.second.levels; l++) hipFree(kv.second.d[l]); if (kv.second.tiled0) hipFree(kv.second.tiled0); if (kv.second.plane0) hipFree(kv.second.plane0); }
  for (auto& kv : ctx->refs)
    for (int l = 0; l < SDSO_PYR_LEVELS; l++) if (kv.second.pc[l]) hipFree(kv.second.pc[l]);
  release_all_windows(ctx);
  for (auto& b : ctx->ba_pool) hipFree(b.first);
  release_track_batch(ctx);
  release_trace(ctx);
  release_match(ctx);
  release_selector(ctx);
  release_g2o(ctx);
  release_comm(ctx);
  if (ctx->gammaB) hipFree(ctx->gammaB);
  if (ctx->scratch) hipFree(ctx->scratch);
  if (ctx->lm_clusters) hipFree(ctx->lm_clusters);
  if (ctx->pinned) hipHostFree(ctx->pinned);
  hipStreamDestroy(ctx->stream);
  delete ctx;
}
extern "C" const char* sdso_last_error(const sdso_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }
extern "C" void* sdso_ctx_stream(sdso_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

// Split the device's CUs between the ctx's two streams (hipExtStreamCreateWithCUMask): the first `aux_cus` of every `stride` CU indices
// belong to `aux`, the rest to `stream`.  Both streams are created anew (call it before sdso_ctx_stream is handed to anyone); aux_cus = 0
// removes the partition.  The 234 us that follow the linearisation in a GN iteration (Schur accumulation: 2 waves per SIMD; fused tail:
// one 150-KB-LDS workgroup per window) cannot co-reside with linearisation workgroups on a CU, so under plain streams two batches
// only take turns; with a partition the tail of one batch owns its CUs while the other batch's linearisation streams on the rest.
extern "C" int sdso_ctx_partition_cus(sdso_ctx* ctx, int aux_cus, int stride) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  SDSO_REQUIRE(ctx, aux_cus >= 0 && aux_cus < ctx->n_cu && stride >= 0, "bad CU partition");
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->aux) { SDSO_HIP(ctx, hipStreamSynchronize(ctx->aux)); hipStreamDestroy(ctx->aux); ctx->aux = nullptr; }
  if (ctx->ev_main) { hipEventDestroy(ctx->ev_main); ctx->ev_main = nullptr; }
  if (ctx->ev_aux) { hipEventDestroy(ctx->ev_aux); ctx->ev_aux = nullptr; }
  hipStreamDestroy(ctx->stream); ctx->stream = nullptr;
  ctx->aux_cus = aux_cus;
  if (aux_cus == 0) {
    SDSO_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    return SDSO_OK;
  }
  const int ncu = ctx->n_cu, words = (ncu + 31) / 32;
  std::vector<uint32_t> ma(words, 0u), mm(words, 0u);
  // stride = 0: the lowest aux_cus indices; otherwise the first (aux_cus * stride / ncu) indices of every `stride` (e.g. stride 32 with
  // aux_cus 32 on 256 CUs: 4 CUs out of every 32 — an equal share of every XCD whichever way the indices run over the XCDs)
  const int per = stride ? std::max(1, aux_cus * stride / ncu) : 0;
  int given = 0;
  for (int c = 0; c < ncu; c++) {
    const bool a = stride ? ((c % stride) < per && given < aux_cus) : c < aux_cus;
    if (a) { ma[c >> 5] |= 1u << (c & 31); given++; } else mm[c >> 5] |= 1u << (c & 31);
  }
  SDSO_HIP(ctx, hipExtStreamCreateWithCUMask(&ctx->stream, words, mm.data()));
  SDSO_HIP(ctx, hipExtStreamCreateWithCUMask(&ctx->aux, words, ma.data()));
  SDSO_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_main, hipEventDisableTiming));
  SDSO_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_aux, hipEventDisableTiming));
  ctx->aux_cus = given;
  return SDSO_OK;
}
extern "C" int sdso_ctx_sync(sdso_ctx* ctx) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SDSO_OK;
}

extern "C" int sdso_prof_enable(sdso_ctx* ctx, int on) {
  if (!ctx) return SDSO_ERR_STATE;
  ctx->prof_on = on < 0 ? 0 : on;
  return SDSO_OK;
}
extern "C" int sdso_prof_reset(sdso_ctx* ctx) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (auto& kv : ctx->prof) {
    for (auto& e : kv.second.ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
  }
  ctx->prof.clear();
  return SDSO_OK;
}
extern "C" int sdso_prof_read(sdso_ctx* ctx, const char* kernel, double* total_ms, long* launches) {
  if (!ctx || !kernel) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  auto it = ctx->prof.find(kernel);
  if (it == ctx->prof.end()) { if (total_ms) *total_ms = 0; if (launches) *launches = 0; return SDSO_OK; }
  for (auto& e : it->second.ev) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) { it->second.total_ms += ms; it->second.launches++; }
    hipEventDestroy(e.first); hipEventDestroy(e.second);
  }
  it->second.ev.clear();
  if (total_ms) *total_ms = it->second.total_ms;
  if (launches) *launches = it->second.launches;
  return SDSO_OK;
}

extern "C" int sdso_pyramid_levels(int w, int h) {  // globalCalib.cpp:52-58
  int wl = w, hl = h, n = 1;
  while (wl % 2 == 0 && hl % 2 == 0 && wl * hl > 5000 && n < SDSO_PYR_LEVELS) { wl /= 2; hl /= 2; n++; }
  return n;
}

extern "C" int sdso_release_pyramid(sdso_ctx* ctx, int frame_slot) {
  if (!ctx) return SDSO_ERR_STATE;
  auto it = ctx->pyr.find(frame_slot);
  if (it == ctx->pyr.end()) return SDSO_OK;
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int l = 0; l < it->second.levels; l++) hipFree(it->second.d[l]);
  if (it->second.tiled0) hipFree(it->second.tiled0);
  if (it->second.plane0) hipFree(it->second.plane0);
  ctx->pyr.erase(it);
  return SDSO_OK;
}

static int alloc_pyramid(sdso_ctx* ctx, int frame_slot, int levels, const int* w, const int* h) {
  SDSO_REQUIRE(ctx, levels >= 1 && levels <= SDSO_PYR_LEVELS, "levels out of range");
  auto it = ctx->pyr.find(frame_slot);
  if (it != ctx->pyr.end()) {
    bool same = it->second.levels == levels;
    for (int l = 0; same && l < levels; l++) same = it->second.w[l] == w[l] && it->second.h[l] == h[l];
    if (same) { it->second.tiled_ok = false; it->second.plane_ok = false; return SDSO_OK; }   // new content arrives in the same buffers
    int rc = sdso_release_pyramid(ctx, frame_slot);
    if (rc) return rc;
  }
  PyramidDev P;
  P.levels = levels;
  for (int l = 0; l < levels; l++) {
    SDSO_REQUIRE(ctx, w[l] >= 8 && h[l] >= 8, "pyramid level too small");
    P.w[l] = w[l]; P.h[l] = h[l];
    SDSO_HIP(ctx, hipMalloc(&P.d[l], sizeof(float4) * (size_t)w[l] * h[l]));
  }
  ctx->pyr[frame_slot] = P;
  return SDSO_OK;
}

// FullSystem::setGammaFunction (FullSystem.cpp:210-234): B from the inverse response Binv (host, 256 entries each)
extern "C" int sdso_gamma_from_binv(const float* BInv, float* B) {
  if (!BInv || !B) return SDSO_ERR_ARG;
  for (int i = 0; i < 256; i++) B[i] = 0;
  for (int i = 1; i < 255; i++)
    for (int s = 1; s < 255; s++)
      if (BInv[s] <= i && BInv[s + 1] >= i) { B[i] = s + (i - BInv[s]) / (BInv[s + 1] - BInv[s]); break; }
  B[0] = 0;
  B[255] = 255;
  return SDSO_OK;
}
// CalibHessian::B for the pyramids built or uploaded from now on (setting_gammaWeightsPixelSelect == 1, HessianBlocks.cpp:194-198);
// B == NULL: identity response (the weight is exactly 1)
extern "C" int sdso_set_gamma(sdso_ctx* ctx, const float* B) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (!B) { if (ctx->gammaB) hipFree(ctx->gammaB); ctx->gammaB = nullptr; return SDSO_OK; }
  if (!ctx->gammaB) SDSO_HIP(ctx, hipMalloc(&ctx->gammaB, 256 * sizeof(float)));
  SDSO_HIP(ctx, hipMemcpy(ctx->gammaB, B, 256 * sizeof(float), hipMemcpyHostToDevice));
  return SDSO_OK;
}

extern "C" int sdso_upload_pyramid(sdso_ctx* ctx, int frame_slot, int levels, const int* w, const int* h, const float* const* dIp) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  SDSO_REQUIRE(ctx, w && h && dIp, "null argument");
  int rc = alloc_pyramid(ctx, frame_slot, levels, w, h);
  if (rc) return rc;
  PyramidDev& P = ctx->pyr[frame_slot];
  size_t maxb = 0;
  for (int l = 0; l < levels; l++) maxb = std::max(maxb, (size_t)w[l] * h[l] * 3 * sizeof(float));
  rc = ensure_scratch(ctx, maxb);
  if (rc) return rc;
  for (int l = 0; l < levels; l++) {
    int npix = w[l] * h[l];
    SDSO_HIP(ctx, hipMemcpyAsync(ctx->scratch, dIp[l], (size_t)npix * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_expand3to4, dim3((npix + 255) / 256), dim3(256), 0, ctx->stream, (const float*)ctx->scratch, P.d[l], npix, (const float*)ctx->gammaB);
    SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));  // scratch is reused by the next level / caller buffer may go away
  }
  SDSO_HIP(ctx, hipGetLastError());
  return SDSO_OK;
}

__global__ void k_pack_w(const float4* __restrict__ src, float* __restrict__ dst, int npix) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npix) dst[i] = src[i].w;
}
// FrameHessian::absSquaredGrad[lvl] as the device holds it (w_l * h_l floats)
extern "C" int sdso_download_abs_grad(sdso_ctx* ctx, int frame_slot, int lvl, float* out) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  auto it = ctx->pyr.find(frame_slot);
  SDSO_REQUIRE(ctx, it != ctx->pyr.end() && out && lvl >= 0 && lvl < it->second.levels, "unknown pyramid slot / level");
  const int npix = it->second.w[lvl] * it->second.h[lvl];
  int rc = ensure_scratch(ctx, (size_t)npix * sizeof(float));
  if (rc) return rc;
  hipLaunchKernelGGL(k_pack_w, dim3((npix + 255) / 256), dim3(256), 0, ctx->stream, (const float4*)it->second.d[lvl], (float*)ctx->scratch, npix);
  SDSO_HIP(ctx, hipMemcpyAsync(out, ctx->scratch, (size_t)npix * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SDSO_OK;
}

extern "C" int sdso_make_pyramid(sdso_ctx* ctx, int frame_slot, int w, int h, const float* color) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  SDSO_REQUIRE(ctx, color, "null argument");
  int levels = sdso_pyramid_levels(w, h);
  int ws[SDSO_PYR_LEVELS], hs[SDSO_PYR_LEVELS];
  for (int l = 0; l < levels; l++) { ws[l] = w >> l; hs[l] = h >> l; }
  int rc = alloc_pyramid(ctx, frame_slot, levels, ws, hs);
  if (rc) return rc;
  PyramidDev& P = ctx->pyr[frame_slot];
  rc = ensure_scratch(ctx, (size_t)w * h * sizeof(float));
  if (rc) return rc;
  SDSO_HIP(ctx, hipMemcpyAsync(ctx->scratch, color, (size_t)w * h * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_set_level0, dim3((w * h + 255) / 256), dim3(256), 0, ctx->stream, (const float*)ctx->scratch, P.d[0], w * h);
  for (int l = 0; l < levels; l++) {
    if (l > 0) hipLaunchKernelGGL(k_downsample, dim3((ws[l] + 255) / 256, hs[l]), dim3(256), 0, ctx->stream, P.d[l - 1], ws[l - 1], P.d[l], ws[l], hs[l]);
    int ng = ws[l] * (hs[l] - 2);
    hipLaunchKernelGGL(k_gradients, dim3((ng + 255) / 256), dim3(256), 0, ctx->stream, P.d[l], ws[l], hs[l], (const float*)ctx->gammaB);
  }
  SDSO_HIP(ctx, hipGetLastError());
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SDSO_OK;
}

extern "C" int sdso_download_pyramid_level(sdso_ctx* ctx, int frame_slot, int lvl, float* dI_out) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  auto it = ctx->pyr.find(frame_slot);
  SDSO_REQUIRE(ctx, it != ctx->pyr.end(), "unknown frame slot");
  SDSO_REQUIRE(ctx, lvl >= 0 && lvl < it->second.levels && dI_out, "bad level");
  int npix = it->second.w[lvl] * it->second.h[lvl];
  int rc = ensure_scratch(ctx, (size_t)npix * 3 * sizeof(float));
  if (rc) return rc;
  hipLaunchKernelGGL(k_pack4to3, dim3((npix + 255) / 256), dim3(256), 0, ctx->stream, it->second.d[lvl], (float*)ctx->scratch, npix);
  SDSO_HIP(ctx, hipMemcpyAsync(dI_out, ctx->scratch, (size_t)npix * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SDSO_OK;
}

// ------------------------------------------------------------------ self-test of the device Lie-group arithmetic
#include "host_math.h"
// k_track_lm and opt_step_body (the resident LM / GN loops) run host_math.h's SE3::exp, product and inverse ON THE DEVICE; this entry
// point runs exactly those functions in a kernel so that the reference's Sophus fixtures (thirdparty/Sophus/sophus/test_se3.cpp:40-82,
// tests.hpp:43-200) can be put through the device code as well as through the oracle (tests/test_oracle_se3.py).
namespace sdso {
__global__ void k_selftest_se3(int n, const double* __restrict__ xi, double* __restrict__ T, double* __restrict__ Tinv, double* __restrict__ Tmul) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Se3 E = expSe3(xi + 6 * i);
  const Se3 I = inverse(E);
  const Se3 P = E * expSe3(xi + 6 * ((i + 1) % n));
  for (int k = 0; k < 9; k++) { T[12 * i + k] = E.R[k]; Tinv[12 * i + k] = I.R[k]; Tmul[12 * i + k] = P.R[k]; }
  for (int k = 0; k < 3; k++) { T[12 * i + 9 + k] = E.t[k]; Tinv[12 * i + 9 + k] = I.t[k]; Tmul[12 * i + 9 + k] = P.t[k]; }
}
}  // namespace sdso
extern "C" int sdso_selftest_se3(sdso_ctx* ctx, int n, const double* xi, double* T_exp, double* T_inv, double* T_mul_next) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_REQUIRE(ctx, n > 0 && xi && T_exp && T_inv && T_mul_next, "bad arguments");
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  double *d_xi = nullptr, *d_out = nullptr;
  SDSO_HIP(ctx, hipMalloc(&d_xi, sizeof(double) * 6 * n));
  if (hipMalloc(&d_out, sizeof(double) * 36 * n) != hipSuccess) { hipFree(d_xi); return sdso::fail(ctx, SDSO_ERR_HIP, "hipMalloc"); }
  hipMemcpyAsync(d_xi, xi, sizeof(double) * 6 * n, hipMemcpyHostToDevice, ctx->stream);
  hipLaunchKernelGGL(sdso::k_selftest_se3, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, n, (const double*)d_xi, d_out, d_out + 12 * n, d_out + 24 * n);
  hipMemcpyAsync(T_exp, d_out, sizeof(double) * 12 * n, hipMemcpyDeviceToHost, ctx->stream);
  hipMemcpyAsync(T_inv, d_out + 12 * n, sizeof(double) * 12 * n, hipMemcpyDeviceToHost, ctx->stream);
  hipMemcpyAsync(T_mul_next, d_out + 24 * n, sizeof(double) * 12 * n, hipMemcpyDeviceToHost, ctx->stream);
  const hipError_t e = hipStreamSynchronize(ctx->stream);
  hipFree(d_xi); hipFree(d_out);
  if (e != hipSuccess) return sdso::fail(ctx, SDSO_ERR_HIP, hipGetErrorString(e));
  return SDSO_OK;
}
