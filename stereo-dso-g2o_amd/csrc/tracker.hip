// Coarse tracker on gfx950: fused CoarseTracker::calcRes + calcGSSSE (one launch per evaluation,
// any number of independent problems per launch) and the DSO-native LM driver of
// CoarseTracker::trackNewestCoarse.
//
// Reference (paths under /root/reference):
//   src/FullSystem/CoarseTracker.cpp:600-792  calcRes   (native body preserved as comments :699-775)
//   src/FullSystem/CoarseTracker.cpp:537-596  calcGSSSE (Accumulator9: MatrixAccumulators.h:907-1277)
//   src/FullSystem/CoarseTracker.cpp:827-1069 trackNewestCoarse (native LM preserved as comments)
//
// Kernel design (HBM/latency bound, ~160 flop for 64 algorithmic bytes per point):
//   * template points are float4 {u,v,idepth,color}: one 16-B coalesced load per lane;
//   * images are float4 {I,dx,dy,0}: the four bilinear taps are two 32-B row segments;
//   * every lane keeps the 45 upper-triangle sums of the 9x9 system + energy/statistics in
//     registers over its grid-stride loop, then a 64-lane butterfly + LDS cross-wave reduce writes
//     ONE partial record per workgroup (no atomics, deterministic);
//   * a second tiny kernel folds the partials of each problem in fixed order and applies the
//     1/n + SCALE_* scaling in double;
//   * workgroup -> (problem, chunk) mapping keeps all chunks of a problem on one XCD
//     (linear id % 8), so the problem's image rows are fetched into one L2 only.
// Per-point arithmetic is written in the reference's operation order and compiled without FP
// contraction, so residuals/Jacobian rows/inlier decisions are bit-identical to the CPU path;
// only the order of the cross-point sums differs (float tolerance, tests/test_tracker_gpu.py).
#include "sdso_internal.h"
#include "host_math.h"
#include <cmath>
#include <cstring>

using namespace sdso;

namespace sdso {

constexpr int TRK_BLOCK = 256;
constexpr int TRK_UNROLL = 4;  // template points per lane and loop trip (16 gathers in flight)
constexpr int TRK_NF = 48;  // float partials: 45 H + E + shiftT + shiftRT
constexpr int TRK_NI = 4;   // int partials: numTermsInE, numSaturated, numWarped, shiftNum

struct TrackProb {
  sdso_track_eval_t ev;
  const float4* pc;
  const float4* img;
  int n;
  int pad;
};

struct TrackOut {
  double H[64];
  double b[8];
  double res[6];
  int n_warped;
  int pad;
};

struct TrackBatch {
  int cap = 0, nprob = 0, gx = 0;
  TrackProb* d_probs = nullptr;
  float* d_partF = nullptr;
  int* d_partI = nullptr;
  TrackOut* d_out = nullptr;
  size_t part_cap = 0;
};

void release_track_batch(sdso_ctx* ctx) {
  if (!ctx->tb) return;
  TrackBatch* tb = ctx->tb;
  if (tb->d_probs) hipFree(tb->d_probs);
  if (tb->d_partF) hipFree(tb->d_partF);
  if (tb->d_partI) hipFree(tb->d_partI);
  if (tb->d_out) hipFree(tb->d_out);
  delete tb;
  ctx->tb = nullptr;
}

}  // namespace sdso

// ------------------------------------------------------------------ kernels
template <bool MASK>
__global__ __launch_bounds__(TRK_BLOCK) void k_track_eval(const TrackProb* __restrict__ probs, int nprob, int gx,
                                                          float* __restrict__ partF, int* __restrict__ partI,
                                                          uint8_t* __restrict__ mask) {
  // XCD-aware mapping: linear workgroup id L runs on XCD (L % 8); give every chunk of problem p
  // the same residue so one L2 serves the problem's image.  Speed only; any placement is correct.
  const int L = blockIdx.x;
  const int xcd = L & 7;
  const int j = L >> 3;
  const int p = (j / gx) * 8 + xcd;
  const int bx = j % gx;
  if (p >= nprob) return;
  const TrackProb& P = probs[p];
  const int n = P.n;
  if (bx > 0 && bx * TRK_BLOCK >= n) return;   // no points for this workgroup (k_track_finalize skips its partial)
  const int lvl = P.ev.lvl, wl = P.ev.w, hl = P.ev.h;
  const float fxl = P.ev.fx, fyl = P.ev.fy, cxl = P.ev.cx, cyl = P.ev.cy;
  const float affLL0 = P.ev.affLL[0], affLL1 = P.ev.affLL[1];
  const float b0 = P.ev.ref_b0, cutoffTH = P.ev.cutoffTH, huberTH = P.ev.huberTH;
  const float maxEnergy = 2 * huberTH * cutoffTH - huberTH * huberTH;
  const float4* __restrict__ pc = P.pc;
  const float4* __restrict__ img = P.img;
  float RKi[9], Ki[9], t[3];
#pragma unroll
  for (int k = 0; k < 9; k++) { RKi[k] = P.ev.RKi[k]; Ki[k] = P.ev.Ki[k]; }
#pragma unroll
  for (int k = 0; k < 3; k++) t[k] = P.ev.t[k];
  const float wlm3 = (float)(wl - 3), hlm3 = (float)(hl - 3);

  float acc[45];
#pragma unroll
  for (int k = 0; k < 45; k++) acc[k] = 0.f;
  float E = 0.f, sT = 0.f, sRT = 0.f;
  int nE = 0, nSat = 0, nWarp = 0, nShift = 0;

  // TRK_UNROLL template points per lane and trip, in three straight-line stages so that the memory system sees
  // all of a trip's requests at once: (1) the pc loads, (2) projection + bounds test + the 4 bilinear taps of every
  // point (an out-of-bounds point reads pixel (2,2) instead of branching around its loads), (3) residual, Huber,
  // the 45 products — in point order, so the per-lane sums are those of the one-point-per-trip loop.
  const int stride = gx * TRK_BLOCK;
  for (int i0 = bx * TRK_BLOCK + threadIdx.x; i0 < n; i0 += TRK_UNROLL * stride) {
    float4 q[TRK_UNROLL];
#pragma unroll
    for (int s = 0; s < TRK_UNROLL; s++) {
      const int i = i0 + s * stride;
      q[s] = pc[i < n ? i : i0];
    }
    float us[TRK_UNROLL], vs[TRK_UNROLL], nid[TRK_UNROLL];
    bool ok[TRK_UNROLL];
    float3 hits[TRK_UNROLL];
#pragma unroll
    for (int s = 0; s < TRK_UNROLL; s++) {
      const int i = i0 + s * stride;
      const float x = q[s].x, y = q[s].y, id = q[s].z;
      float pt[3];
#pragma unroll
      for (int r = 0; r < 3; r++) pt[r] = ((RKi[r * 3 + 0] * x + RKi[r * 3 + 1] * y) + RKi[r * 3 + 2]) + t[r] * id;
      const float u = pt[0] / pt[2];
      const float v = pt[1] / pt[2];
      const float Ku = fxl * u + cxl;
      const float Kv = fyl * v + cyl;
      const float new_idepth = id / pt[2];
      us[s] = u; vs[s] = v; nid[s] = new_idepth;

      if (lvl == 0 && (i & 31) == 0 && i < n) {  // CoarseTracker.cpp:662-693 flow indicators
        float ptT[3], ptT2[3], pt3[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
          const float kp = (Ki[r * 3 + 0] * x + Ki[r * 3 + 1] * y) + Ki[r * 3 + 2];
          const float rp = (RKi[r * 3 + 0] * x + RKi[r * 3 + 1] * y) + RKi[r * 3 + 2];
          ptT[r] = kp + t[r] * id;
          ptT2[r] = kp - t[r] * id;
          pt3[r] = rp - t[r] * id;
        }
        const float KuT = fxl * (ptT[0] / ptT[2]) + cxl, KvT = fyl * (ptT[1] / ptT[2]) + cyl;
        const float KuT2 = fxl * (ptT2[0] / ptT2[2]) + cxl, KvT2 = fyl * (ptT2[1] / ptT2[2]) + cyl;
        const float Ku3 = fxl * (pt3[0] / pt3[2]) + cxl, Kv3 = fyl * (pt3[1] / pt3[2]) + cyl;
        sT += (KuT - x) * (KuT - x) + (KvT - y) * (KvT - y);
        sT += (KuT2 - x) * (KuT2 - x) + (KvT2 - y) * (KvT2 - y);
        sRT += (Ku - x) * (Ku - x) + (Kv - y) * (Kv - y);
        sRT += (Ku3 - x) * (Ku3 - x) + (Kv3 - y) * (Kv3 - y);
        nShift += 2;
      }
      ok[s] = i < n && Ku > 2 && Kv > 2 && Ku < wlm3 && Kv < hlm3 && new_idepth > 0;  // :696
      hits[s] = interp33(img, ok[s] ? Ku : 2.5f, ok[s] ? Kv : 2.5f, wl);
    }
#pragma unroll
    for (int s = 0; s < TRK_UNROLL; s++) {
      const int i = i0 + s * stride;
      const float u = us[s], v = vs[s], new_idepth = nid[s], refColor = q[s].w;
      const float3 hit = hits[s];
      bool inl = false;
      if (ok[s] && isfinite(hit.x)) {
        const float residual = hit.x - (affLL0 * refColor + affLL1);
        const float ar = fabsf(residual);
        const float hw = ar < huberTH ? 1.f : huberTH / ar;
        nE++;
        if (ar > cutoffTH) {
          E += maxEnergy;
          nSat++;
        } else {
          E += hw * residual * residual * (2 - hw);
          nWarp++;
          inl = true;
          // calcGSSSE rows (:555-577), same nesting as the SSE expressions
          const float dx = hit.y * fxl;
          const float dy = hit.z * fyl;
          float J[9];
          J[0] = new_idepth * dx;
          J[1] = new_idepth * dy;
          J[2] = 0.0f - new_idepth * (u * dx + v * dy);
          J[3] = 0.0f - ((u * v) * dx + dy * (1.0f + v * v));
          J[4] = (u * v) * dy + dx * (1.0f + u * u);
          J[5] = u * dy - v * dx;
          J[6] = affLL0 * (b0 - refColor);
          J[7] = -1.0f;
          J[8] = residual;
          int k = 0;
#pragma unroll
          for (int r = 0; r < 9; r++) {
            const float Jw = J[r] * hw;
#pragma unroll
            for (int c = r; c < 9; c++) { acc[k] = __builtin_fmaf(Jw, J[c], acc[k]); k++; }
          }
        }
      }
      if (MASK && i < n) mask[i] = inl ? 1 : 0;
    }
  }

  // ---- workgroup reduction: 64-lane butterfly, then 4 waves through LDS
  __shared__ float sF[TRK_BLOCK / 64][TRK_NF];
  __shared__ int sI[TRK_BLOCK / 64][TRK_NI];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  {
    float v48[TRK_NF];
#pragma unroll
    for (int k = 0; k < 45; k++) v48[k] = acc[k];
    v48[45] = E; v48[46] = sT; v48[47] = sRT;
    wave_reduce_rows<TRK_NF>(v48, [&](int k, float s) { sF[wv][k] = s; });
  }
  {
    int i0 = nE, i1 = nSat, i2 = nWarp, i3 = nShift;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      i0 += __shfl_xor(i0, o, 64); i1 += __shfl_xor(i1, o, 64); i2 += __shfl_xor(i2, o, 64); i3 += __shfl_xor(i3, o, 64);
    }
    if (lane == 0) { sI[wv][0] = i0; sI[wv][1] = i1; sI[wv][2] = i2; sI[wv][3] = i3; }
  }
  __syncthreads();
  const size_t rec = (size_t)p * gx + bx;
  if (threadIdx.x < TRK_NF) {
    float s = sF[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < TRK_BLOCK / 64; w++) s += sF[w][threadIdx.x];
    partF[rec * TRK_NF + threadIdx.x] = s;
  } else if (threadIdx.x < TRK_NF + TRK_NI) {
    const int k = threadIdx.x - TRK_NF;
    int s = sI[0][k];
#pragma unroll
    for (int w = 1; w < TRK_BLOCK / 64; w++) s += sI[w][k];
    partI[rec * TRK_NI + k] = s;
  }
}

// Fold the per-workgroup partials of each problem (fixed order) and finish like calcGSSSE :580-595
// and calcRes :783-789.
__global__ __launch_bounds__(64) void k_track_finalize(const TrackProb* __restrict__ probs, const float* __restrict__ partF,
                                                       const int* __restrict__ partI, int gx, TrackOut* __restrict__ out) {
  const int p = blockIdx.x;
  const int nb = min(gx, max(1, (probs[p].n + TRK_BLOCK - 1) / TRK_BLOCK));   // workgroups that had points
  __shared__ float F[TRK_NF];
  __shared__ int I[TRK_NI];
  const int tid = threadIdx.x;
  if (tid < TRK_NF) {
    float s = 0.f;
    for (int b = 0; b < nb; b++) s += partF[((size_t)p * gx + b) * TRK_NF + tid];
    F[tid] = s;
  } else if (tid < TRK_NF + TRK_NI) {
    int s = 0;
    for (int b = 0; b < nb; b++) s += partI[((size_t)p * gx + b) * TRK_NI + tid - TRK_NF];
    I[tid - TRK_NF] = s;
  }
  __syncthreads();
  const int nE = I[0], nSat = I[1], nWarp = I[2], nShift = I[3];
  const int npad = (nWarp + 3) & ~3;  // buf_warped_n with its zero padding (:763-775)
  TrackOut& O = out[p];
  const double SC[8] = {SCALE_XI_ROT, SCALE_XI_ROT, SCALE_XI_ROT, SCALE_XI_TRANS, SCALE_XI_TRANS, SCALE_XI_TRANS, SCALE_A, SCALE_B};
  const float inv_n = 1.0f / npad;
  // upper-triangle index of (r,c), r<=c, 9 columns
  for (int e = tid; e < 72; e += 64) {
    const int r = e / 9, c = e % 9;  // r in 0..7, c in 0..8
    const int lo = r < c ? r : c, hi = r < c ? c : r;
    const int idx = lo * 9 - lo * (lo - 1) / 2 + (hi - lo);
    double v = npad > 0 ? (double)F[idx] * (double)inv_n : 0.0;
    if (c < 8) { v *= SC[c]; v *= SC[r]; O.H[r * 8 + c] = v; }
    else { v *= SC[r]; O.b[r] = v; }
  }
  if (tid == 0) {
    O.res[0] = (double)F[45];
    O.res[1] = (double)nE;
    O.res[2] = (double)F[46] / ((double)(float)nShift + 0.1);
    O.res[3] = 0;
    O.res[4] = (double)F[47] / ((double)(float)nShift + 0.1);
    O.res[5] = (double)((float)nSat / (float)nE);
    O.n_warped = npad;
  }
}

// ------------------------------------------------------------------ host side
static void fill_eval(const sdso_track_params_t& p, int lvl, const Se3& T, const sdso_aff_t& aff, float cutoff, sdso_track_eval_t& ev) {
  ev.lvl = lvl; ev.w = p.w[lvl]; ev.h = p.h[lvl];
  ev.fx = p.fx[lvl]; ev.fy = p.fy[lvl]; ev.cx = p.cx[lvl]; ev.cy = p.cy[lvl];
  const float K[9] = {ev.fx, 0, ev.cx, 0, ev.fy, ev.cy, 0, 0, 1};
  inv3f(K, ev.Ki);                                   // CoarseTracker.cpp:129-130
  float Rf[9];
  for (int i = 0; i < 9; i++) Rf[i] = (float)T.R[i];
  mul3f(Rf, ev.Ki, ev.RKi);                          // :617
  for (int i = 0; i < 3; i++) ev.t[i] = (float)T.t[i];
  double a2[2];
  affFromTo(p.ref_exposure, p.new_exposure, p.ref_aff_g2l.a, p.ref_aff_g2l.b, aff.a, aff.b, a2);
  ev.affLL[0] = (float)a2[0]; ev.affLL[1] = (float)a2[1];
  ev.ref_b0 = (float)p.ref_aff_g2l.b;
  ev.cutoffTH = cutoff;
  ev.huberTH = p.huberTH;
}

extern "C" void sdso_track_make_eval(const sdso_track_params_t* prm, int lvl, const sdso_se3_t* refToNew, const sdso_aff_t* aff_g2l,
                                     float levelCutoffRepeat, sdso_track_eval_t* ev) {
  Se3 T;
  std::memcpy(T.R.data(), refToNew->R, 72);
  std::memcpy(T.t.data(), refToNew->t, 24);
  fill_eval(*prm, lvl, T, *aff_g2l, prm->coarseCutoffTH * levelCutoffRepeat, *ev);
}

extern "C" int sdso_track_set_ref(sdso_ctx* ctx, int ref_slot, int lvl, int n, const float* pc_u, const float* pc_v,
                                  const float* pc_idepth, const float* pc_color) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  SDSO_REQUIRE(ctx, lvl >= 0 && lvl < SDSO_PYR_LEVELS && n >= 0, "bad level / n");
  SDSO_REQUIRE(ctx, n == 0 || (pc_u && pc_v && pc_idepth && pc_color), "null pc arrays");
  RefDev& R = ctx->refs[ref_slot];
  if (R.pc[lvl]) { SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream)); hipFree(R.pc[lvl]); R.pc[lvl] = nullptr; }
  R.n[lvl] = n;
  if (n == 0) return SDSO_OK;
  std::vector<float4> h(n);
  for (int i = 0; i < n; i++) h[i] = make_float4(pc_u[i], pc_v[i], pc_idepth[i], pc_color[i]);
  SDSO_HIP(ctx, hipMalloc(&R.pc[lvl], sizeof(float4) * (size_t)n));
  SDSO_HIP(ctx, hipMemcpy(R.pc[lvl], h.data(), sizeof(float4) * (size_t)n, hipMemcpyHostToDevice));
  return SDSO_OK;
}

namespace sdso { void release_g2o_ref(sdso_ctx* ctx, int ref_slot); }   // g2o_factors.hip
extern "C" int sdso_track_release_ref(sdso_ctx* ctx, int ref_slot) {
  if (!ctx) return SDSO_ERR_STATE;
  auto it = ctx->refs.find(ref_slot);
  if (it == ctx->refs.end()) return SDSO_OK;
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  sdso::release_g2o_ref(ctx, ref_slot);
  for (int l = 0; l < SDSO_PYR_LEVELS; l++) if (it->second.pc[l]) hipFree(it->second.pc[l]);
  ctx->refs.erase(it);
  return SDSO_OK;
}

static int resolve_prob(sdso_ctx* ctx, int ref_slot, int frame_slot, const sdso_track_eval_t& ev, TrackProb& P) {
  auto ir = ctx->refs.find(ref_slot);
  SDSO_REQUIRE(ctx, ir != ctx->refs.end(), "unknown ref slot");
  auto ip = ctx->pyr.find(frame_slot);
  SDSO_REQUIRE(ctx, ip != ctx->pyr.end(), "unknown frame slot");
  const int lvl = ev.lvl;
  SDSO_REQUIRE(ctx, lvl >= 0 && lvl < ip->second.levels, "level not in pyramid");
  // the kernel indexes the image with (w,h) from the eval: they must be the uploaded level's size
  SDSO_REQUIRE(ctx, ev.w == ip->second.w[lvl] && ev.h == ip->second.h[lvl], "eval w/h do not match the uploaded pyramid level");
  P.ev = ev;
  P.pc = ir->second.pc[lvl];
  P.img = ip->second.d[lvl];
  P.n = ir->second.n[lvl];
  P.pad = 0;
  return SDSO_OK;
}

static int batch_reserve(sdso_ctx* ctx, int nprob, int gx) {
  if (!ctx->tb) ctx->tb = new TrackBatch();
  TrackBatch* tb = ctx->tb;
  if (tb->cap < nprob) {
    SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (tb->d_probs) hipFree(tb->d_probs);
    if (tb->d_out) hipFree(tb->d_out);
    tb->cap = nprob + nprob / 2 + 8;
    SDSO_HIP(ctx, hipMalloc(&tb->d_probs, sizeof(TrackProb) * tb->cap));
    SDSO_HIP(ctx, hipMalloc(&tb->d_out, sizeof(TrackOut) * tb->cap));
  }
  size_t need = (size_t)nprob * gx;
  if (tb->part_cap < need) {
    SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (tb->d_partF) hipFree(tb->d_partF);
    if (tb->d_partI) hipFree(tb->d_partI);
    tb->part_cap = need + need / 2 + 64;
    SDSO_HIP(ctx, hipMalloc(&tb->d_partF, sizeof(float) * TRK_NF * tb->part_cap));
    SDSO_HIP(ctx, hipMalloc(&tb->d_partI, sizeof(int) * TRK_NI * tb->part_cap));
  }
  return SDSO_OK;
}

// workgroups per problem: enough to fill the chip when few problems are in flight.
static int choose_gx(const sdso_ctx* ctx, int nprob, int maxn) {
  if (maxn <= 0) return 1;
  int by_points = (maxn + TRK_BLOCK - 1) / TRK_BLOCK;          // 1 point / thread
  int target = (ctx->n_cu * 8 + nprob - 1) / nprob;            // ~8 workgroups per CU over the batch
  // few, fat workgroups: the per-workgroup epilogue (48-value reduction, partial stores) is amortised over several
  // loop trips (measured on 640 problems: gx 10 -> 65 us, 4 -> 62 us, 1 -> 71 us)
  int gx = std::max(1, std::min(by_points, target));
  if (const char* e = getenv("SDSO_TRK_GX")) gx = std::max(1, std::min(by_points, atoi(e)));   // experiment
  return gx;
}

extern "C" int sdso_track_batch_prepare(sdso_ctx* ctx, int nprob, const int* ref_slots, const int* frame_slots, const sdso_track_eval_t* evs) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  SDSO_REQUIRE(ctx, nprob > 0 && ref_slots && frame_slots && evs, "bad batch arguments");
  std::vector<TrackProb> h(nprob);
  int maxn = 0;
  for (int i = 0; i < nprob; i++) {
    int rc = resolve_prob(ctx, ref_slots[i], frame_slots[i], evs[i], h[i]);
    if (rc) return rc;
    maxn = std::max(maxn, h[i].n);
  }
  int gx = choose_gx(ctx, nprob, maxn);
  int rc = batch_reserve(ctx, nprob, gx);
  if (rc) return rc;
  ctx->tb->nprob = nprob;
  ctx->tb->gx = gx;
  SDSO_HIP(ctx, hipMemcpy(ctx->tb->d_probs, h.data(), sizeof(TrackProb) * nprob, hipMemcpyHostToDevice));
  return SDSO_OK;
}

extern "C" int sdso_track_batch_enqueue(sdso_ctx* ctx) {
  if (!ctx || !ctx->tb || ctx->tb->nprob <= 0) return sdso::fail(ctx, SDSO_ERR_STATE, "no prepared batch");
  TrackBatch* tb = ctx->tb;
  const int groups = (tb->nprob + 7) / 8;
  const int nblk = groups * 8 * tb->gx;
  {
    ProfScope ps(ctx, "k_track_eval");
    hipLaunchKernelGGL(k_track_eval<false>, dim3(nblk), dim3(TRK_BLOCK), 0, ctx->stream, tb->d_probs, tb->nprob, tb->gx, tb->d_partF, tb->d_partI, (uint8_t*)nullptr);
  }
  hipLaunchKernelGGL(k_track_finalize, dim3(tb->nprob), dim3(64), 0, ctx->stream, tb->d_probs, tb->d_partF, tb->d_partI, tb->gx, tb->d_out);
  SDSO_HIP(ctx, hipGetLastError());
  return SDSO_OK;
}

extern "C" int sdso_track_batch_fetch(sdso_ctx* ctx, double* H, double* b, double* res, int* n_warped) {
  if (!ctx || !ctx->tb || ctx->tb->nprob <= 0) return sdso::fail(ctx, SDSO_ERR_STATE, "no prepared batch");
  TrackBatch* tb = ctx->tb;
  int rc = ensure_pinned(ctx, sizeof(TrackOut) * tb->nprob);
  if (rc) return rc;
  SDSO_HIP(ctx, hipMemcpyAsync(ctx->pinned, tb->d_out, sizeof(TrackOut) * tb->nprob, hipMemcpyDeviceToHost, ctx->stream));
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const TrackOut* o = (const TrackOut*)ctx->pinned;
  for (int i = 0; i < tb->nprob; i++) {
    if (H) std::memcpy(H + (size_t)i * 64, o[i].H, sizeof(double) * 64);
    if (b) std::memcpy(b + (size_t)i * 8, o[i].b, sizeof(double) * 8);
    if (res) std::memcpy(res + (size_t)i * 6, o[i].res, sizeof(double) * 6);
    if (n_warped) n_warped[i] = o[i].n_warped;
  }
  return SDSO_OK;
}

extern "C" int sdso_track_calc_res_gs_batch(sdso_ctx* ctx, int nprob, const int* ref_slots, const int* frame_slots,
                                            const sdso_track_eval_t* evs, double* H, double* b, double* res, int* n_warped) {
  int rc = sdso_track_batch_prepare(ctx, nprob, ref_slots, frame_slots, evs);
  if (rc) return rc;
  rc = sdso_track_batch_enqueue(ctx);
  if (rc) return rc;
  return sdso_track_batch_fetch(ctx, H, b, res, n_warped);
}

// one evaluation (the LM loop's unit of work); results land in ctx->pinned as a TrackOut
static int eval_one(sdso_ctx* ctx, const TrackProb& P, uint8_t* mask_host) {
  int gx = choose_gx(ctx, 1, P.n);
  int rc = batch_reserve(ctx, 1, gx);
  if (rc) return rc;
  TrackBatch* tb = ctx->tb;
  tb->nprob = 0;  // invalidates any prepared batch
  rc = ensure_pinned(ctx, sizeof(TrackOut) + sizeof(TrackProb));
  if (rc) return rc;
  uint8_t* d_mask = nullptr;
  if (mask_host && P.n > 0) {
    rc = ensure_scratch(ctx, (size_t)P.n);
    if (rc) return rc;
    d_mask = (uint8_t*)ctx->scratch;
  }
  SDSO_HIP(ctx, hipMemcpyAsync(tb->d_probs, &P, sizeof(TrackProb), hipMemcpyHostToDevice, ctx->stream));
  if (d_mask)
    hipLaunchKernelGGL(k_track_eval<true>, dim3(8 * gx), dim3(TRK_BLOCK), 0, ctx->stream, tb->d_probs, 1, gx, tb->d_partF, tb->d_partI, d_mask);
  else
    hipLaunchKernelGGL(k_track_eval<false>, dim3(8 * gx), dim3(TRK_BLOCK), 0, ctx->stream, tb->d_probs, 1, gx, tb->d_partF, tb->d_partI, (uint8_t*)nullptr);
  hipLaunchKernelGGL(k_track_finalize, dim3(1), dim3(64), 0, ctx->stream, tb->d_probs, tb->d_partF, tb->d_partI, gx, tb->d_out);
  SDSO_HIP(ctx, hipGetLastError());
  SDSO_HIP(ctx, hipMemcpyAsync(ctx->pinned, tb->d_out, sizeof(TrackOut), hipMemcpyDeviceToHost, ctx->stream));
  if (d_mask) SDSO_HIP(ctx, hipMemcpyAsync(mask_host, d_mask, (size_t)P.n, hipMemcpyDeviceToHost, ctx->stream));
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SDSO_OK;
}

extern "C" int sdso_track_calc_res_gs(sdso_ctx* ctx, int ref_slot, int frame_slot, const sdso_track_eval_t* ev, double* H, double* b,
                                      double* res, int* n_warped, uint8_t* inlier_mask) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  SDSO_REQUIRE(ctx, ev, "null eval");
  TrackProb P;
  int rc = resolve_prob(ctx, ref_slot, frame_slot, *ev, P);
  if (rc) return rc;
  rc = eval_one(ctx, P, inlier_mask);
  if (rc) return rc;
  const TrackOut* o = (const TrackOut*)ctx->pinned;
  if (H) std::memcpy(H, o->H, sizeof(double) * 64);
  if (b) std::memcpy(b, o->b, sizeof(double) * 8);
  if (res) std::memcpy(res, o->res, sizeof(double) * 6);
  if (n_warped) *n_warped = o->n_warped;
  return SDSO_OK;
}

// CoarseTracker::trackNewestCoarse, DSO-native LM.
// ------------------------------------------------------------------ trackNewestCoarse for many hypotheses in lock-step
// CoarseTracker::trackNewestCoarse (CoarseTracker.cpp:827-1069, DSO-native LM :908-1024) is a chain of calcRes+calcGSSSE
// evaluations with a little 8x8 algebra in between.  Each hypothesis is a small state machine that always has exactly one
// evaluation pending (first evaluation of a level, repeat with a doubled cutoff, or the trial step of an LM iteration); all
// pending evaluations of a round go to the device in ONE launch.  The sequence of evaluations — and therefore every number —
// of a hypothesis is the one the sequential loop produces.
namespace sdso {
struct LmState {
  int ref_slot = 0, frame_slot = 0;
  sdso_track_params_t p;
  sdso_track_result_t* out = nullptr;
  sdso_se3_t* T_io = nullptr;
  sdso_aff_t* aff_io = nullptr;
  Se3 cur, Tnew;
  sdso_aff_t affCur, affNew;
  bool haveRepeated = false, done = false;
  int lvl = 0, iteration = 0, phase = 0;   // phase 0: first / repeated evaluation of a level, 1: trial step
  float levelCutoffRepeat = 1, lambda = 0.01f;
  TrackOut oldO;
  double H[64], b[8];
  std::vector<double> inc;
  // the evaluation this hypothesis waits for
  Se3 reqT; sdso_aff_t reqAff;

  void request(const Se3& T, const sdso_aff_t& a) { reqT = T; reqAff = a; }
  void start_level() { levelCutoffRepeat = 1; phase = 0; request(cur, affCur); }
  void finish() {   // :1044-1068
    done = true;
    std::memcpy(T_io->R, cur.R.data(), 72);
    std::memcpy(T_io->t, cur.t.data(), 24);
    *aff_io = affCur;
    if ((p.affineOptModeA != 0 && (fabsf((float)aff_io->a) > 1.2)) || (p.affineOptModeB != 0 && (fabsf((float)aff_io->b) > 200))) return;
    double rel[2];
    affFromTo(p.ref_exposure, p.new_exposure, p.ref_aff_g2l.a, p.ref_aff_g2l.b, aff_io->a, aff_io->b, rel);
    const float r0 = (float)rel[0], r1 = (float)rel[1];
    if ((p.affineOptModeA == 0 && (fabsf(logf(r0)) > 1.5)) || (p.affineOptModeB == 0 && (fabsf(r1) > 200))) return;
    if (p.affineOptModeA < 0) aff_io->a = 0;
    if (p.affineOptModeB < 0) aff_io->b = 0;
    out->good = 1;
  }
  void finish_level() {
    out->lastResiduals[lvl] = sqrtf((float)(oldO.res[0] / oldO.res[1]));
    out->lastFlowIndicators[0] = oldO.res[2]; out->lastFlowIndicators[1] = oldO.res[3]; out->lastFlowIndicators[2] = oldO.res[4];
    if (out->lastResiduals[lvl] > 1.5 * p.minResForAbort[lvl]) { done = true; return; }  // :1032 (good stays 0, pose untouched)
    if (levelCutoffRepeat > 1 && !haveRepeated) { lvl++; haveRepeated = true; }
    lvl--;
    if (lvl < 0) finish(); else start_level();
  }
  void propose() {   // one LM step from (H, b, lambda): :931-1000
    const float lambdaExtrapolationLimit = 0.001f;
    if (iteration >= p.maxIterations[lvl]) { finish_level(); return; }
    out->iterations[lvl]++;
    Dense Hl(8);
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) Hl(i, j) = H[i * 8 + j];
    for (int i = 0; i < 8; i++) Hl(i, i) *= (1 + lambda);
    std::vector<double> nb(8);
    for (int i = 0; i < 8; i++) nb[i] = -b[i];
    solveLdlt(Hl, nb, inc);
    auto sub = [&](const Dense& Hs, const double* bs, int m, std::vector<double>& xs) {
      Dense Hm(m);
      std::vector<double> bm(m);
      for (int i = 0; i < m; i++) { bm[i] = -bs[i]; for (int j = 0; j < m; j++) Hm(i, j) = Hs(i, j); }
      solveLdlt(Hm, bm, xs);
    };
    if (p.affineOptModeA < 0 && p.affineOptModeB < 0) {  // fix a, b (:937-940)
      std::vector<double> x6; sub(Hl, b, 6, x6);
      for (int i = 0; i < 6; i++) inc[i] = x6[i];
      inc[6] = inc[7] = 0;
    }
    if (!(p.affineOptModeA < 0) && p.affineOptModeB < 0) {  // fix b (:943-946)
      std::vector<double> x7; sub(Hl, b, 7, x7);
      for (int i = 0; i < 7; i++) inc[i] = x7[i];
      inc[7] = 0;
    }
    if (p.affineOptModeA < 0 && !(p.affineOptModeB < 0)) {  // fix a (:949-964)
      Dense Hs = Hl;
      double bs[8];
      std::memcpy(bs, b, sizeof(bs));
      for (int i = 0; i < 8; i++) Hs(i, 6) = Hs(i, 7);
      for (int j = 0; j < 8; j++) Hs(6, j) = Hs(7, j);
      bs[6] = bs[7];
      std::vector<double> x7; sub(Hs, bs, 7, x7);
      for (int i = 0; i < 8; i++) inc[i] = 0;
      for (int i = 0; i < 6; i++) inc[i] = x7[i];
      inc[7] = x7[6];
    }
    float extrapFac = 1;
    if (lambda < lambdaExtrapolationLimit) extrapFac = sqrt(sqrt(lambdaExtrapolationLimit / lambda));
    for (int i = 0; i < 8; i++) inc[i] *= extrapFac;
    double incScaled[8];
    for (int i = 0; i < 8; i++) incScaled[i] = inc[i];
    for (int i = 0; i < 3; i++) incScaled[i] *= SCALE_XI_ROT;
    for (int i = 3; i < 6; i++) incScaled[i] *= SCALE_XI_TRANS;
    incScaled[6] *= SCALE_A;
    incScaled[7] *= SCALE_B;
    double sum = 0;
    for (int i = 0; i < 8; i++) sum += incScaled[i];
    if (!std::isfinite(sum)) for (int i = 0; i < 8; i++) incScaled[i] = 0;
    Tnew = expSe3(incScaled) * cur;
    affNew = affCur;
    affNew.a += incScaled[6];
    affNew.b += incScaled[7];
    phase = 1;
    request(Tnew, affNew);
  }
  void consume(const TrackOut& O) {
    const float lambdaExtrapolationLimit = 0.001f;
    if (phase == 0) {
      oldO = O;
      if (oldO.res[5] > 0.6 && levelCutoffRepeat < 50) { levelCutoffRepeat *= 2; request(cur, affCur); return; }   // :897-904
      std::memcpy(H, oldO.H, sizeof(H));
      std::memcpy(b, oldO.b, sizeof(b));
      lambda = 0.01f;
      iteration = 0;
      propose();
      return;
    }
    const bool accept = (O.res[0] / O.res[1]) < (oldO.res[0] / oldO.res[1]);
    if (accept) {
      std::memcpy(H, O.H, sizeof(H));
      std::memcpy(b, O.b, sizeof(b));
      oldO = O;
      affCur = affNew;
      cur = Tnew;
      lambda *= 0.5;
    } else {
      lambda *= 4;
      if (lambda < lambdaExtrapolationLimit) lambda = lambdaExtrapolationLimit;
    }
    double nrm = 0;
    for (int i = 0; i < 8; i++) nrm += inc[i] * inc[i];
    if (!(std::sqrt(nrm) > 1e-3)) { finish_level(); return; }
    iteration++;
    propose();
  }
};
}  // namespace sdso

extern "C" int sdso_track_newest_coarse_batch(sdso_ctx* ctx, int nhyp, const int* ref_slots, const int* frame_slots, const sdso_track_params_t* prms,
                                              sdso_se3_t* lastToNew, sdso_aff_t* aff_g2l, sdso_track_result_t* outs) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  SDSO_REQUIRE(ctx, nhyp > 0 && ref_slots && frame_slots && prms && lastToNew && aff_g2l && outs, "null argument");
  std::vector<LmState> S(nhyp);
  for (int k = 0; k < nhyp; k++) {
    LmState& s = S[k];
    s.p = prms[k];
    SDSO_REQUIRE(ctx, s.p.coarsestLvl >= 0 && s.p.coarsestLvl < 5 && s.p.coarsestLvl < s.p.levels, "coarsestLvl out of range");  // assert :853
    s.ref_slot = ref_slots[k]; s.frame_slot = frame_slots[k];
    s.out = &outs[k]; s.T_io = &lastToNew[k]; s.aff_io = &aff_g2l[k];
    for (int i = 0; i < 5; i++) { s.out->lastResiduals[i] = NAN; s.out->iterations[i] = 0; }
    for (int i = 0; i < 3; i++) s.out->lastFlowIndicators[i] = 1000;
    s.out->evaluations = 0; s.out->point_evals = 0; s.out->good = 0;
    std::memcpy(s.cur.R.data(), lastToNew[k].R, 72);
    std::memcpy(s.cur.t.data(), lastToNew[k].t, 24);
    s.affCur = aff_g2l[k];
    s.lvl = s.p.coarsestLvl;
    s.start_level();
  }
  std::vector<TrackProb> probs;
  std::vector<int> who;
  for (;;) {
    probs.clear(); who.clear();
    for (int k = 0; k < nhyp; k++) {
      LmState& s = S[k];
      if (s.done) continue;
      sdso_track_eval_t ev;
      fill_eval(s.p, s.lvl, s.reqT, s.reqAff, s.p.coarseCutoffTH * s.levelCutoffRepeat, ev);
      TrackProb P;
      int rc = resolve_prob(ctx, s.ref_slot, s.frame_slot, ev, P);
      if (rc) return rc;
      probs.push_back(P); who.push_back(k);
      s.out->evaluations++;
      s.out->point_evals += P.n;
    }
    if (probs.empty()) break;
    const int np = (int)probs.size();
    int maxn = 0;
    for (const TrackProb& P : probs) maxn = std::max(maxn, P.n);
    const int gx = choose_gx(ctx, np, maxn);
    int rc = batch_reserve(ctx, np, gx);
    if (rc) return rc;
    TrackBatch* tb = ctx->tb;
    tb->nprob = 0;  // invalidates any prepared batch
    rc = ensure_pinned(ctx, (sizeof(TrackOut) + sizeof(TrackProb)) * (size_t)np);
    if (rc) return rc;
    TrackProb* hp = (TrackProb*)((char*)ctx->pinned + sizeof(TrackOut) * (size_t)np);
    std::memcpy(hp, probs.data(), sizeof(TrackProb) * np);
    SDSO_HIP(ctx, hipMemcpyAsync(tb->d_probs, hp, sizeof(TrackProb) * np, hipMemcpyHostToDevice, ctx->stream));
    const int groups = (np + 7) / 8;
    hipLaunchKernelGGL(k_track_eval<false>, dim3(groups * 8 * gx), dim3(TRK_BLOCK), 0, ctx->stream, tb->d_probs, np, gx, tb->d_partF, tb->d_partI, (uint8_t*)nullptr);
    hipLaunchKernelGGL(k_track_finalize, dim3(np), dim3(64), 0, ctx->stream, tb->d_probs, tb->d_partF, tb->d_partI, gx, tb->d_out);
    SDSO_HIP(ctx, hipGetLastError());
    SDSO_HIP(ctx, hipMemcpyAsync(ctx->pinned, tb->d_out, sizeof(TrackOut) * np, hipMemcpyDeviceToHost, ctx->stream));
    SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const TrackOut* O = (const TrackOut*)ctx->pinned;
    for (int j = 0; j < np; j++) S[who[j]].consume(O[j]);
  }
  return SDSO_OK;
}

extern "C" int sdso_track_newest_coarse(sdso_ctx* ctx, int ref_slot, int frame_slot, const sdso_track_params_t* prm,
                                        sdso_se3_t* lastToNew, sdso_aff_t* aff_g2l, sdso_track_result_t* out) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_REQUIRE(ctx, prm && lastToNew && aff_g2l && out, "null argument");
  return sdso_track_newest_coarse_batch(ctx, 1, &ref_slot, &frame_slot, prm, lastToNew, aff_g2l, out);
}
