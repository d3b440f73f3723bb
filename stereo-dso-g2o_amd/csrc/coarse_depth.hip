// CoarseTracker::makeCoarseDepthL0 on gfx950: from the weighted inverse depths of the active points to the
// tracking template pc_u / pc_v / pc_idepth / pc_color of every pyramid level.
//
// Reference (paths under /root/reference): src/FullSystem/CoarseTracker.cpp
//   :352-354  STEP1 splat     idepth[0][u+w*v] += new_idepth*weight; weightSums[0][u+w*v] += weight   (in point order)
//   :360-386  STEP2 pyramid   2x2 sums, ((a+b)+c)+d
//   :390-441  STEP3 dilation  levels 0,1: the four diagonal neighbours
//   :445-488  STEP4 dilation  levels >= 2: the four axis neighbours
//   :491-533  STEP5 normalise + compaction in the scan order y in [2,h-2), x in [2,w-2)
// (the static-stereo refinement at the top of STEP1, :295-347, is sdso_stereo_match_batch; the caller applies the accept
// rule and passes the resulting new_idepth / weight here.)
//
// This is the "bit-exact point-index bookkeeping" part of the tracker: pc_n[lvl] and the ORDER of the template
// points must be those of the CPU path, and every float is produced by the same expression.
//   * splat: two points on one pixel must be added in point order.  A point whose pixel no earlier point shares owns
//     the pixel and adds all later points of that pixel in order (n <= ~16k: an LDS-tiled all-pairs scan, ~0.1 ms,
//     instead of float atomics whose order is not defined).
//   * compaction: per-row counts -> exclusive scan over rows -> in-row ranks by ballot prefix: raster order, no atomics.
#include "sdso_internal.h"
#include <vector>

using namespace sdso;

namespace {

constexpr int CD_TILE = 2048;

__global__ __launch_bounds__(256) void k_cd_splat(int n, int w, const int* __restrict__ u, const int* __restrict__ v, const float* __restrict__ idp,
                                                  const float* __restrict__ wgt, float* __restrict__ idepth0, float* __restrict__ wsum0) {
  __shared__ int s_pix[CD_TILE];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int mine = i < n ? u[i] + w * v[i] : -1;
  bool first = i < n;
  for (int base = 0; base < n && base <= (int)(blockIdx.x * blockDim.x + blockDim.x - 1); base += CD_TILE) {   // earlier points: j < i
    __syncthreads();
    for (int k = threadIdx.x; k < CD_TILE; k += blockDim.x) { const int j = base + k; s_pix[k] = j < n ? u[j] + w * v[j] : -2; }
    __syncthreads();
    const int lim = min(CD_TILE, i - base);
    for (int k = 0; k < lim; k++) if (s_pix[k] == mine) { first = false; break; }
  }
  // the owner adds itself and every later point on its pixel, in point order (0 + x is exact, so starting from the
  // zeroed map is the same as the reference's +=)
  float sid = 0.f, sw = 0.f;
  const int blockFirst = blockIdx.x * blockDim.x;
  for (int base = (blockFirst / CD_TILE) * CD_TILE; base < n; base += CD_TILE) {
    __syncthreads();
    for (int k = threadIdx.x; k < CD_TILE; k += blockDim.x) { const int j = base + k; s_pix[k] = j < n ? u[j] + w * v[j] : -2; }
    __syncthreads();
    if (first) {
      for (int k = max(0, i - base); k < CD_TILE; k++)
        if (s_pix[k] == mine) { const int j = base + k; sid += idp[j] * wgt[j]; sw += wgt[j]; }
    }
  }
  if (first) { idepth0[mine] += sid; wsum0[mine] += sw; }
}

__global__ __launch_bounds__(256) void k_cd_down(const float* __restrict__ idm, const float* __restrict__ wsm, int wlm1, float* __restrict__ idl,
                                                 float* __restrict__ wsl, int wl, int hl) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= wl || y >= hl) return;
  const int bidx = 2 * x + 2 * y * wlm1;
  idl[x + y * wl] = idm[bidx] + idm[bidx + 1] + idm[bidx + wlm1] + idm[bidx + wlm1 + 1];
  wsl[x + y * wl] = wsm[bidx] + wsm[bidx + 1] + wsm[bidx + wlm1] + wsm[bidx + wlm1 + 1];
}

// reads idepth only where bak > 0 and writes only where bak <= 0: in place like the reference
__global__ __launch_bounds__(256) void k_cd_dilate(float* __restrict__ idl, float* __restrict__ wsl, const float* __restrict__ bak, int wl, int hl, int diag) {
  const int i = wl + blockIdx.x * blockDim.x + threadIdx.x;
  const int wh = wl * hl - wl, tot = wl * hl;
  if (i >= wh) return;
  if (!(bak[i] <= 0)) return;
  const int o0 = diag ? 1 + wl : 1, o1 = diag ? -1 - wl : -1, o2 = diag ? wl - 1 : wl, o3 = diag ? -wl + 1 : -wl;
  float sum = 0, num = 0, numn = 0;
  const int offs[4] = {o0, o1, o2, o3};
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int j = i + offs[k];
    if (j >= 0 && j < tot && bak[j] > 0) { sum += idl[j]; num += bak[j]; numn++; }   // (the reference reads one element past the map at the last pixel; skipped)
  }
  if (numn > 0) { idl[i] = sum / numn; wsl[i] = num / numn; }
}

// STEP5, pass 1: normalise and flag; one workgroup per row y in [2, hl-2)
__global__ __launch_bounds__(256) void k_cd_flag(float* __restrict__ idl, float* __restrict__ wsl, const float4* __restrict__ ref, int wl, int hl,
                                                 int* __restrict__ rowcnt) {
  const int y = 2 + blockIdx.x;
  __shared__ int s_cnt[4];
  int cnt = 0;
  for (int x0 = 2; x0 < wl - 2; x0 += blockDim.x) {
    const int x = x0 + threadIdx.x;
    bool keep = false;
    if (x < wl - 2) {
      const int i = x + y * wl;
      if (wsl[i] > 0) {
        const float v = idl[i] / wsl[i];
        const float c = ref[i].x;
        if (!isfinite(c) || !(v > 0)) idl[i] = -1;
        else { idl[i] = v; keep = true; wsl[i] = 1; }
        // (a rejected pixel keeps its weight: the reference `continue`s before weightSumsl[i] = 1)
      } else { idl[i] = -1; wsl[i] = 1; }
    }
    cnt += __popcll(__ballot(keep));
  }
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) rowcnt[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}
// exclusive scan of the row counts (hl <= a few hundred rows: one workgroup)
__global__ __launch_bounds__(256) void k_cd_scan(int* __restrict__ rowcnt, int nrows, int* __restrict__ total) {
  __shared__ int s[256];
  int carry = 0;
  for (int base = 0; base < nrows; base += 256) {
    const int r = base + threadIdx.x;
    const int v = r < nrows ? rowcnt[r] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      const int t = threadIdx.x >= o ? s[threadIdx.x - o] : 0;
      __syncthreads();
      s[threadIdx.x] += t;
      __syncthreads();
    }
    if (r < nrows) rowcnt[r] = carry + s[threadIdx.x] - v;
    const int blocksum = s[255];
    __syncthreads();
    carry += blocksum;
  }
  if (threadIdx.x == 0) *total = carry;
}
// pass 2: scatter in raster order (kept pixels are those with idepth > 0 after pass 1)
__global__ __launch_bounds__(256) void k_cd_scatter(const float* __restrict__ idl, const float4* __restrict__ ref, int wl, int hl, const int* __restrict__ rowoff,
                                                    float4* __restrict__ pc) {
  const int y = 2 + blockIdx.x;
  __shared__ int s_cnt[4];
  int base = rowoff[blockIdx.x];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int x0 = 2; x0 < wl - 2; x0 += blockDim.x) {
    const int x = x0 + threadIdx.x;
    bool keep = false;
    float v = 0, c = 0;
    if (x < wl - 2) { const int i = x + y * wl; v = idl[i]; keep = v > 0; c = ref[i].x; }
    const unsigned long long m = __ballot(keep);
    if (lane == 0) s_cnt[wv] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int k = 0; k < wv; k++) off += s_cnt[k];
    off += __popcll(m & ((1ull << lane) - 1ull));
    if (keep) pc[off] = make_float4((float)x, (float)y, v, c);
    base += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    __syncthreads();
  }
}
__global__ void k_cd_unpack(const float4* __restrict__ pc, int n, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 q = pc[i];
  out[i] = q.x; out[n + i] = q.y; out[2 * (size_t)n + i] = q.z; out[3 * (size_t)n + i] = q.w;
}

}  // namespace

extern "C" int sdso_track_make_ref(sdso_ctx* ctx, int ref_slot, int frame_slot, int n, const int* u, const int* v, const float* new_idepth,
                                   const float* weight, int* pc_n_out) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  auto ip = ctx->pyr.find(frame_slot);
  SDSO_REQUIRE(ctx, ip != ctx->pyr.end(), "unknown frame slot");
  SDSO_REQUIRE(ctx, n >= 0 && (n == 0 || (u && v && new_idepth && weight)), "null point arrays");
  const PyramidDev& P = ip->second;
  const int L = P.levels, w0 = P.w[0], h0 = P.h[0];
  for (int i = 0; i < n; i++) SDSO_REQUIRE(ctx, u[i] >= 0 && u[i] < w0 && v[i] >= 0 && v[i] < h0, "point outside the image");
  // maps of all levels in one allocation: idepth, weightSums, weightSums_bak
  size_t tot = 0, off[SDSO_PYR_LEVELS];
  for (int l = 0; l < L; l++) { off[l] = tot; tot += (size_t)P.w[l] * P.h[l]; }
  const size_t need = sizeof(float) * 3 * tot + sizeof(int) * ((size_t)h0 + 8) + sizeof(float) * 2 * (size_t)std::max(n, 1) + sizeof(int) * 2 * (size_t)std::max(n, 1);
  int rc = ensure_scratch(ctx, need);
  if (rc) return rc;
  float* idm = (float*)ctx->scratch;
  float* wsm = idm + tot;
  float* bak = wsm + tot;
  int* rowcnt = (int*)(bak + tot);
  int* d_total = rowcnt + h0 + 4;
  float* d_idp = (float*)(rowcnt + h0 + 8);
  float* d_wgt = d_idp + std::max(n, 1);
  int* d_u = (int*)(d_wgt + std::max(n, 1));
  int* d_v = d_u + std::max(n, 1);
  SDSO_HIP(ctx, hipMemsetAsync(idm, 0, sizeof(float) * 2 * tot, ctx->stream));
  if (n) {
    SDSO_HIP(ctx, hipMemcpyAsync(d_idp, new_idepth, sizeof(float) * n, hipMemcpyHostToDevice, ctx->stream));
    SDSO_HIP(ctx, hipMemcpyAsync(d_wgt, weight, sizeof(float) * n, hipMemcpyHostToDevice, ctx->stream));
    SDSO_HIP(ctx, hipMemcpyAsync(d_u, u, sizeof(int) * n, hipMemcpyHostToDevice, ctx->stream));
    SDSO_HIP(ctx, hipMemcpyAsync(d_v, v, sizeof(int) * n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_cd_splat, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, w0, d_u, d_v, d_idp, d_wgt, idm, wsm);
  }
  for (int l = 1; l < L; l++)
    hipLaunchKernelGGL(k_cd_down, dim3((P.w[l] + 255) / 256, P.h[l]), dim3(256), 0, ctx->stream, idm + off[l - 1], wsm + off[l - 1], P.w[l - 1], idm + off[l],
                       wsm + off[l], P.w[l], P.h[l]);
  for (int l = 0; l < L; l++) {
    const size_t px = (size_t)P.w[l] * P.h[l];
    SDSO_HIP(ctx, hipMemcpyAsync(bak + off[l], wsm + off[l], sizeof(float) * px, hipMemcpyDeviceToDevice, ctx->stream));
    const int cnt = P.w[l] * P.h[l] - 2 * P.w[l];
    if (cnt > 0)
      hipLaunchKernelGGL(k_cd_dilate, dim3((cnt + 255) / 256), dim3(256), 0, ctx->stream, idm + off[l], wsm + off[l], bak + off[l], P.w[l], P.h[l], l < 2 ? 1 : 0);
  }
  RefDev& R = ctx->refs[ref_slot];
  SDSO_HIP(ctx, hipGetLastError());
  for (int l = 0; l < SDSO_PYR_LEVELS; l++) {
    if (R.pc[l]) { SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream)); hipFree(R.pc[l]); R.pc[l] = nullptr; }
    R.n[l] = 0;
  }
  for (int l = 0; l < L; l++) {
    const int nrows = P.h[l] - 4;
    int total = 0;
    if (nrows > 0 && P.w[l] > 4) {
      hipLaunchKernelGGL(k_cd_flag, dim3(nrows), dim3(256), 0, ctx->stream, idm + off[l], wsm + off[l], P.d[l], P.w[l], P.h[l], rowcnt);
      hipLaunchKernelGGL(k_cd_scan, dim3(1), dim3(256), 0, ctx->stream, rowcnt, nrows, d_total);
      SDSO_HIP(ctx, hipMemcpyAsync(&total, d_total, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (total > 0) {
        SDSO_HIP(ctx, hipMalloc(&R.pc[l], sizeof(float4) * (size_t)total));
        hipLaunchKernelGGL(k_cd_scatter, dim3(nrows), dim3(256), 0, ctx->stream, idm + off[l], P.d[l], P.w[l], P.h[l], rowcnt, R.pc[l]);
      }
    }
    R.n[l] = total;
    if (pc_n_out) pc_n_out[l] = total;
  }
  SDSO_HIP(ctx, hipGetLastError());
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SDSO_OK;
}

extern "C" int sdso_track_get_ref(sdso_ctx* ctx, int ref_slot, int lvl, int* n_out, float* pc_u, float* pc_v, float* pc_idepth, float* pc_color) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  auto ir = ctx->refs.find(ref_slot);
  SDSO_REQUIRE(ctx, ir != ctx->refs.end(), "unknown ref slot");
  SDSO_REQUIRE(ctx, lvl >= 0 && lvl < SDSO_PYR_LEVELS, "bad level");
  const int n = ir->second.n[lvl];
  if (n_out) *n_out = n;
  if (n == 0 || !(pc_u || pc_v || pc_idepth || pc_color)) return SDSO_OK;
  int rc = ensure_scratch(ctx, sizeof(float) * 4 * (size_t)n);
  if (rc) return rc;
  float* d = (float*)ctx->scratch;
  hipLaunchKernelGGL(k_cd_unpack, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ir->second.pc[lvl], n, d);
  float* dst[4] = {pc_u, pc_v, pc_idepth, pc_color};
  for (int k = 0; k < 4; k++)
    if (dst[k]) SDSO_HIP(ctx, hipMemcpyAsync(dst[k], d + (size_t)k * n, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SDSO_OK;
}
