// PixelSelector::makeMaps on gfx950 (candidate pixels for new immature points).
//
// Reference (paths under /root/reference): src/FullSystem/PixelSelector2.cpp
//   :40-56    constructor: randomPattern[i] = rand() & 0xFF after srand(3141592)
//   :67-81    computeHistQuantil;  :84-189 makeHists (32x32-cell gradient histograms, 3x3 smoothing)
//   :193-300  makeMaps (potential recursion, random thinning);  :330-540 select (three nested block levels)
//
// select() is a raster scan with ONE running counter: the direction tested inside a block is
// directions[randomPattern[n2] & 0xF] with n2 = number of level-0 picks made so far in scan order.  Whether a pot x pot
// block makes a level-0 pick depends on its direction only through "is |grad . dir| > 0 for some pixel above the
// threshold", so the scan is split in three:
//   1. k_ps_masks   (one lane per pot-block): 16-bit mask, bit d = the block picks a pixel if its direction is d
//   2. host         : the reference's nested block order with those masks -> n2 at the entry of every pot-block
//                     (~w*h/pot^2 table look-ups; the only sequential part)
//   3. k_ps_select  (one lane per 4pot-block): the reference's loops for that block with the known counters
// Everything is decided by comparisons of the same floats as on the CPU: the map is identical, not approximately equal.
#include "sdso_internal.h"
#include <algorithm>
#include <cmath>
#include <vector>

using namespace sdso;

namespace {

__constant__ float c_dirs[16][2] = {{0, 1.0000f}, {0.3827f, 0.9239f}, {0.1951f, 0.9808f}, {0.9239f, 0.3827f}, {0.7071f, 0.7071f}, {0.3827f, -0.9239f},
                                    {0.8315f, 0.5556f}, {0.8315f, -0.5556f}, {0.5556f, -0.8315f}, {0.9808f, 0.1951f}, {0.9239f, -0.3827f},
                                    {0.7071f, -0.7071f}, {0.5556f, 0.8315f}, {0.9808f, -0.1951f}, {1.0000f, 0.0000f}, {0.1951f, -0.9808f}};
constexpr float kMinGradHistCut = 0.5f, kMinGradHistAdd = 7.f, kGradDownweightPerLevel = 0.75f;   // settings.cpp:105-107

// glibc's rand() (TYPE_3 additive feedback, r[i] = r[i-3] + r[i-31]) so that the pattern does not depend on — or disturb —
// the process-wide generator the reference reseeds in its constructor.  Checked against srand/rand in tests.
static void glibc_rand_bytes(unsigned seed, size_t n, std::vector<unsigned char>& out) {
  std::vector<int32_t> r(344 + n);
  r[0] = (int32_t)seed;
  for (int i = 1; i < 31; i++) {
    const int64_t hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
    int64_t word = 16807 * lo - 2836 * hi;
    if (word < 0) word += 2147483647;
    r[i] = (int32_t)word;
  }
  for (int i = 31; i < 34; i++) r[i] = r[i - 31];
  for (size_t i = 34; i < 344 + n; i++) r[i] = (int32_t)((uint32_t)r[i - 31] + (uint32_t)r[i - 3]);
  out.resize(n);
  for (size_t k = 0; k < n; k++) out[k] = (unsigned char)((((uint32_t)r[k + 344]) >> 1) & 0xFF);
}

// makeHists: one workgroup per 32x32 cell
__global__ __launch_bounds__(256) void k_ps_hist(const float4* __restrict__ img0, int w, int h, int w32, float* __restrict__ ths) {
  __shared__ int hist[52];
  const int cx = blockIdx.x % w32, cy = blockIdx.x / w32;
  if (threadIdx.x < 52) hist[threadIdx.x] = 0;
  __syncthreads();
  for (int k = threadIdx.x; k < 1024; k += blockDim.x) {
    const int i = k & 31, j = k >> 5;
    const int it = i + 32 * cx, jt = j + 32 * cy;
    if (it > w - 2 || jt > h - 2 || it < 1 || jt < 1) continue;
    int g = sqrtf(img0[it + jt * w].w);
    if (g > 48) g = 48;
    atomicAdd(&hist[g + 1], 1);
    atomicAdd(&hist[0], 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int th = hist[0] * kMinGradHistCut + 0.5f;
    int q = 90;
    for (int i = 0; i < 90; i++) {
      th -= (i + 1 < 52 ? hist[i + 1] : 0);
      if (th < 0) { q = i; break; }
    }
    ths[blockIdx.x] = q + kMinGradHistAdd;
  }
}
__global__ void k_ps_smooth(const float* __restrict__ ths, int w32, int h32, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w32 * h32) return;
  const int x = i % w32, y = i / w32;
  float sum = 0, num = 0;
  if (x > 0) {
    if (y > 0) { num++; sum += ths[x - 1 + (y - 1) * w32]; }
    if (y < h32 - 1) { num++; sum += ths[x - 1 + (y + 1) * w32]; }
    num++; sum += ths[x - 1 + y * w32];
  }
  if (x < w32 - 1) {
    if (y > 0) { num++; sum += ths[x + 1 + (y - 1) * w32]; }
    if (y < h32 - 1) { num++; sum += ths[x + 1 + (y + 1) * w32]; }
    num++; sum += ths[x + 1 + y * w32];
  }
  if (y > 0) { num++; sum += ths[x + (y - 1) * w32]; }
  if (y < h32 - 1) { num++; sum += ths[x + (y + 1) * w32]; }
  num++; sum += ths[x + y * w32];
  out[i] = (sum / num) * (sum / num);
}

// pass 1: bit d of mask[pot-block] = "with direction d this block picks a level-0 pixel"
__global__ __launch_bounds__(256) void k_ps_masks(const float4* __restrict__ img0, int w, int h, int pot, int nbx, int nby, const float* __restrict__ thsS,
                                                  int thsStep, float thFactor, unsigned short* __restrict__ mask) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbx * nby) return;
  const int x0 = (b % nbx) * pot, y0 = (b / nbx) * pot;
  unsigned m = 0;
  for (int y1 = 0; y1 < pot && y0 + y1 < h; y1++)
    for (int x1 = 0; x1 < pot && x0 + x1 < w; x1++) {
      const int xf = x0 + x1, yf = y0 + y1;
      if (xf < 4 || xf >= w - 5 || yf < 4 || yf > h - 4) continue;
      const float4 px = img0[xf + w * yf];
      const float pixelTH0 = thsS[(xf >> 5) + (yf >> 5) * thsStep];
      if (px.w > pixelTH0 * thFactor) {
#pragma unroll
        for (int d = 0; d < 16; d++) if (fabsf((float)(px.y * c_dirs[d][0] + px.z * c_dirs[d][1])) > 0.f) m |= 1u << d;
      }
    }
  mask[b] = (unsigned short)m;
}

// pass 3: one lane per 4pot x 4pot block, the reference's loops with the counters resolved on the host
__global__ __launch_bounds__(128) void k_ps_select(const float4* __restrict__ img0, const float4* __restrict__ img1, const float4* __restrict__ img2, int w, int h,
                                                   int w1, int w2, int pot, int nb4x, int nb4y, const float* __restrict__ thsS, int thsStep, float thFactor,
                                                   const unsigned char* __restrict__ rnd, const int* __restrict__ n2_at4, float* __restrict__ map_out,
                                                   int* __restrict__ counts) {
  const int b4 = blockIdx.x * blockDim.x + threadIdx.x;
  if (b4 >= nb4x * nb4y) return;
  const int x4 = (b4 % nb4x) * 4 * pot, y4 = (b4 / nb4x) * 4 * pot;
  const float dw1 = kGradDownweightPerLevel, dw2 = dw1 * dw1;
  int n2 = n2_at4[b4], n3 = 0, n4 = 0;
  const int n2_start = n2;
  const int my3 = min(4 * pot, h - y4), mx3 = min(4 * pot, w - x4);
  int bestIdx4 = -1; float bestVal4 = 0;
  const int d4 = rnd[n2] & 0xF;
  for (int y3 = 0; y3 < my3; y3 += 2 * pot)
    for (int x3 = 0; x3 < mx3; x3 += 2 * pot) {
      const int x34 = x3 + x4, y34 = y3 + y4;
      const int my2 = min(2 * pot, h - y34), mx2 = min(2 * pot, w - x34);
      int bestIdx3 = -1; float bestVal3 = 0;
      const int d3 = rnd[n2] & 0xF;
      for (int y2 = 0; y2 < my2; y2 += pot)
        for (int x2 = 0; x2 < mx2; x2 += pot) {
          const int x234 = x2 + x34, y234 = y2 + y34;
          const int my1 = min(pot, h - y234), mx1 = min(pot, w - x234);
          int bestIdx2 = -1; float bestVal2 = 0;
          const int d2 = rnd[n2] & 0xF;
          for (int y1 = 0; y1 < my1; y1++)
            for (int x1 = 0; x1 < mx1; x1++) {
              const int xf = x1 + x234, yf = y1 + y234;
              const int idx = xf + w * yf;
              if (xf < 4 || xf >= w - 5 || yf < 4 || yf > h - 4) continue;
              const float pixelTH0 = thsS[(xf >> 5) + (yf >> 5) * thsStep];
              const float pixelTH1 = pixelTH0 * dw1;
              const float pixelTH2 = pixelTH1 * dw2;
              const float4 px = img0[idx];
              if (px.w > pixelTH0 * thFactor) {
                const float dirNorm = fabsf((float)(px.y * c_dirs[d2][0] + px.z * c_dirs[d2][1]));
                if (dirNorm > bestVal2) { bestVal2 = dirNorm; bestIdx2 = idx; bestIdx3 = -2; bestIdx4 = -2; }
              }
              if (bestIdx3 == -2) continue;
              const float ag1 = img1[(int)(xf * 0.5f + 0.25f) + (int)(yf * 0.5f + 0.25f) * w1].w;
              if (ag1 > pixelTH1 * thFactor) {
                const float dirNorm = fabsf((float)(px.y * c_dirs[d3][0] + px.z * c_dirs[d3][1]));
                if (dirNorm > bestVal3) { bestVal3 = dirNorm; bestIdx3 = idx; bestIdx4 = -2; }
              }
              if (bestIdx4 == -2) continue;
              const float ag2 = img2[(int)(xf * 0.25f + 0.125) + (int)(yf * 0.25f + 0.125) * w2].w;
              if (ag2 > pixelTH2 * thFactor) {
                const float dirNorm = fabsf((float)(px.y * c_dirs[d4][0] + px.z * c_dirs[d4][1]));
                if (dirNorm > bestVal4) { bestVal4 = dirNorm; bestIdx4 = idx; }
              }
            }
          if (bestIdx2 > 0) { map_out[bestIdx2] = 1; bestVal3 = 1e10; n2++; }
        }
      if (bestIdx3 > 0) { map_out[bestIdx3] = 2; bestVal4 = 1e10; n3++; }
    }
  if (bestIdx4 > 0) { map_out[bestIdx4] = 4; n4++; }
  if (n2 != n2_start) atomicAdd(&counts[0], n2 - n2_start);
  if (n3) atomicAdd(&counts[1], n3);
  if (n4) atomicAdd(&counts[2], n4);
}

// random thinning (:246-262): rn = rank of the pixel among the non-zero map entries in raster order
__global__ __launch_bounds__(256) void k_ps_rowcount(const float* __restrict__ map, int w, int* __restrict__ rowcnt) {
  __shared__ int s[4];
  int cnt = 0;
  for (int x0 = 0; x0 < w; x0 += blockDim.x) {
    const int x = x0 + threadIdx.x;
    const bool nz = x < w && map[x + blockIdx.x * w] != 0;
    cnt += __popcll(__ballot(nz));
  }
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) rowcnt[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ __launch_bounds__(256) void k_ps_thin(float* __restrict__ map, int w, const int* __restrict__ rowoff, const unsigned char* __restrict__ rnd, int charTH,
                                                 int* __restrict__ killed) {
  __shared__ int s[4];
  int base = rowoff[blockIdx.x];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int nk = 0;
  for (int x0 = 0; x0 < w; x0 += blockDim.x) {
    const int x = x0 + threadIdx.x;
    const bool nz = x < w && map[x + blockIdx.x * w] != 0;
    const unsigned long long m = __ballot(nz);
    if (lane == 0) s[wv] = __popcll(m);
    __syncthreads();
    int rn = base;
    for (int k = 0; k < wv; k++) rn += s[k];
    rn += __popcll(m & ((1ull << lane) - 1ull));
    if (nz && (int)rnd[rn] > charTH) { map[x + blockIdx.x * w] = 0; nk++; }
    base += s[0] + s[1] + s[2] + s[3];
    __syncthreads();
  }
  if (nk) atomicAdd(killed, nk);
}

struct SelState {
  unsigned char* d_rnd = nullptr;
  std::vector<unsigned char> h_rnd;
  int w = 0, h = 0;
};
static std::map<sdso_ctx*, SelState> g_sel;

}  // namespace

namespace sdso {
void release_selector(sdso_ctx* ctx) {
  SelState st;
  if (!reg_take(g_sel, ctx, st)) return;
  if (st.d_rnd) hipFree(st.d_rnd);
}
}  // namespace sdso

// the first n bytes of the selector's random pattern (tests: must equal srand(3141592); rand() & 0xFF)
extern "C" int sdso_pixel_selector_pattern(int n, unsigned char* out) {
  if (n < 0 || (n && !out)) return SDSO_ERR_ARG;
  std::vector<unsigned char> v;
  glibc_rand_bytes(3141592u, (size_t)n, v);
  std::copy(v.begin(), v.end(), out);
  return SDSO_OK;
}

extern "C" int sdso_pixel_select(sdso_ctx* ctx, int frame_slot, float density, int recursionsLeft, float thFactor, int* potential, float* map_out,
                                 int* num_out) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  SDSO_REQUIRE(ctx, potential && *potential >= 1 && num_out, "null / bad potential");
  auto ip = ctx->pyr.find(frame_slot);
  SDSO_REQUIRE(ctx, ip != ctx->pyr.end(), "unknown frame slot");
  const PyramidDev& P = ip->second;
  SDSO_REQUIRE(ctx, P.levels >= 3, "the selector reads absSquaredGrad of levels 0..2");
  const int w = P.w[0], h = P.h[0], w32 = w / 32, h32 = h / 32;
  SDSO_REQUIRE(ctx, w32 > 0 && h32 > 0, "image smaller than one 32x32 cell");
  SelState& S = reg_get(g_sel, ctx);
  if (S.w != w || S.h != h) {
    if (S.d_rnd) { SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream)); hipFree(S.d_rnd); S.d_rnd = nullptr; }
    glibc_rand_bytes(3141592u, (size_t)w * h, S.h_rnd);
    SDSO_HIP(ctx, hipMalloc(&S.d_rnd, (size_t)w * h));
    SDSO_HIP(ctx, hipMemcpy(S.d_rnd, S.h_rnd.data(), (size_t)w * h, hipMemcpyHostToDevice));
    S.w = w; S.h = h;
  }
  // scratch: map (w*h floats) | ths, thsSmoothed | masks (pot = 1 worst case: w*h) | n2 table | row counts | counters
  const size_t npx = (size_t)w * h;
  const size_t bytes = sizeof(float) * npx + sizeof(float) * 2 * ((size_t)w32 * h32 + w32 + 8) + sizeof(unsigned short) * npx + sizeof(int) * npx + sizeof(int) * ((size_t)h + 16);
  int rc = ensure_scratch(ctx, bytes);
  if (rc) return rc;
  float* d_map = (float*)ctx->scratch;
  float* d_ths = d_map + npx;
  float* d_thsS = d_ths + (size_t)w32 * h32 + w32 + 8;
  unsigned short* d_mask = (unsigned short*)(d_thsS + (size_t)w32 * h32 + w32 + 8);
  int* d_n2 = (int*)(d_mask + npx);
  int* d_row = d_n2 + npx;
  int* d_cnt = d_row + h + 4;    // n2, n3, n4, killed
  // pixels right of / below the last full 32x32 cell index thsSmoothed past the w32*h32 cells that makeHists fills (the reference
  // reads uninitialised floats of its over-allocated array there); those entries are defined as 0 here and in the oracle
  SDSO_HIP(ctx, hipMemsetAsync(d_thsS, 0, sizeof(float) * ((size_t)w32 * h32 + w32 + 8), ctx->stream));
  hipLaunchKernelGGL(k_ps_hist, dim3(w32 * h32), dim3(256), 0, ctx->stream, P.d[0], w, h, w32, d_ths);
  hipLaunchKernelGGL(k_ps_smooth, dim3((w32 * h32 + 255) / 256), dim3(256), 0, ctx->stream, d_ths, w32, h32, d_thsS);
  SDSO_HIP(ctx, hipGetLastError());

  int currentPotential = *potential;
  float quotia = 0, numHave = 0;
  int idealPotential = currentPotential;
  std::vector<unsigned short> h_mask;
  std::vector<int> h_n2;
  for (;;) {   // makeMaps' tail recursion (:193-243)
    const int pot = currentPotential;
    const int nbx = (w + pot - 1) / pot, nby = (h + pot - 1) / pot, nb4x = (w + 4 * pot - 1) / (4 * pot), nb4y = (h + 4 * pot - 1) / (4 * pot);
    hipLaunchKernelGGL(k_ps_masks, dim3((nbx * nby + 255) / 256), dim3(256), 0, ctx->stream, P.d[0], w, h, pot, nbx, nby, d_thsS, w32, thFactor, d_mask);
    h_mask.resize((size_t)nbx * nby);
    SDSO_HIP(ctx, hipMemcpyAsync(h_mask.data(), d_mask, sizeof(unsigned short) * h_mask.size(), hipMemcpyDeviceToHost, ctx->stream));
    SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // the reference's block order; only the running count of level-0 picks is tracked
    h_n2.assign((size_t)nb4x * nb4y, 0);
    int n2 = 0;
    for (int y4 = 0, b4 = 0; y4 < h; y4 += 4 * pot)
      for (int x4 = 0; x4 < w; x4 += 4 * pot, b4++) {
        h_n2[b4] = n2;
        const int my3 = std::min(4 * pot, h - y4), mx3 = std::min(4 * pot, w - x4);
        for (int y3 = 0; y3 < my3; y3 += 2 * pot)
          for (int x3 = 0; x3 < mx3; x3 += 2 * pot) {
            const int my2 = std::min(2 * pot, h - (y3 + y4)), mx2 = std::min(2 * pot, w - (x3 + x4));
            for (int y2 = 0; y2 < my2; y2 += pot)
              for (int x2 = 0; x2 < mx2; x2 += pot) {
                const int bx = (x2 + x3 + x4) / pot, by = (y2 + y3 + y4) / pot;
                if ((h_mask[(size_t)by * nbx + bx] >> (S.h_rnd[n2] & 0xF)) & 1) n2++;
              }
          }
      }
    SDSO_HIP(ctx, hipMemcpyAsync(d_n2, h_n2.data(), sizeof(int) * h_n2.size(), hipMemcpyHostToDevice, ctx->stream));
    SDSO_HIP(ctx, hipMemsetAsync(d_map, 0, sizeof(float) * npx, ctx->stream));
    SDSO_HIP(ctx, hipMemsetAsync(d_cnt, 0, sizeof(int) * 4, ctx->stream));
    hipLaunchKernelGGL(k_ps_select, dim3((nb4x * nb4y + 127) / 128), dim3(128), 0, ctx->stream, P.d[0], P.d[1], P.d[2], w, h, P.w[1], P.w[2], pot, nb4x, nb4y,
                       d_thsS, w32, thFactor, (const unsigned char*)S.d_rnd, (const int*)d_n2, d_map, d_cnt);
    SDSO_HIP(ctx, hipGetLastError());
    int cnt[4];
    SDSO_HIP(ctx, hipMemcpyAsync(cnt, d_cnt, sizeof(int) * 4, hipMemcpyDeviceToHost, ctx->stream));
    SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SDSO_REQUIRE(ctx, cnt[0] == n2, "selector: the level-0 count of the select pass differs from the resolved scan");
    numHave = cnt[0] + cnt[1] + cnt[2];
    const float numWant = density;
    quotia = numWant / numHave;
    const float K = numHave * (currentPotential + 1) * (currentPotential + 1);
    idealPotential = sqrtf(K / numWant) - 1;
    if (idealPotential < 1) idealPotential = 1;
    if (recursionsLeft > 0 && quotia > 1.25 && currentPotential > 1) {
      if (idealPotential >= currentPotential) idealPotential = currentPotential - 1;
      currentPotential = idealPotential; recursionsLeft--;
      continue;
    } else if (recursionsLeft > 0 && quotia < 0.25) {
      if (idealPotential <= currentPotential) idealPotential = currentPotential + 1;
      currentPotential = idealPotential; recursionsLeft--;
      continue;
    }
    break;
  }
  int numHaveSub = numHave;
  if (quotia < 0.95) {
    const unsigned char charTH = 255 * quotia;
    hipLaunchKernelGGL(k_ps_rowcount, dim3(h), dim3(256), 0, ctx->stream, (const float*)d_map, w, d_row);
    std::vector<int> rows(h);
    SDSO_HIP(ctx, hipMemcpyAsync(rows.data(), d_row, sizeof(int) * h, hipMemcpyDeviceToHost, ctx->stream));
    SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int run = 0;
    for (int y = 0; y < h; y++) { const int c = rows[y]; rows[y] = run; run += c; }
    SDSO_HIP(ctx, hipMemcpyAsync(d_row, rows.data(), sizeof(int) * h, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_ps_thin, dim3(h), dim3(256), 0, ctx->stream, d_map, w, (const int*)d_row, (const unsigned char*)S.d_rnd, (int)charTH, d_cnt + 3);
    int killed = 0;
    SDSO_HIP(ctx, hipMemcpyAsync(&killed, d_cnt + 3, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    numHaveSub -= killed;
  }
  *potential = idealPotential;
  *num_out = numHaveSub;
  if (map_out) SDSO_HIP(ctx, hipMemcpyAsync(map_out, d_map, sizeof(float) * npx, hipMemcpyDeviceToHost, ctx->stream));
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SDSO_OK;
}
