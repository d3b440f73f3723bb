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

#ifdef SDSO_LM_STAMPS   // diagnostic build (make EXTRA=-DSDSO_LM_STAMPS, tools/dbg_lm_stamps.py): shader-clock cycles of thread 0 per phase of k_track_lm
__shared__ unsigned long long lm_st_acc[16];
__shared__ unsigned long long lm_st_last;
#ifdef SDSO_LM_STAMPS_NOWAIT
#define LMS_WAIT
#else
#define LMS_WAIT asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
#define LMS(i) do { if (threadIdx.x == 0) { LMS_WAIT const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); lm_st_acc[i] += tn_ - lm_st_last; lm_st_last = tn_; } } while (0)
#else
#define LMS(i) do { } while (0)
#endif
// ------------------------------------------------------------------ kernels
// Per-lane sums of calcRes + calcGSSSE over the points first, first + stride, ... of one problem (TRK_UNROLL points per trip):
// the 45 upper-triangle products, E, the flow-indicator sums and the four counters.
struct TrackLaneSums {
  float acc[45];
  float E, sT, sRT;
  int nE, nSat, nWarp, nShift;
};
// PRE: the caller hands over the lane's TRK_UNROLL points of the FIRST trip (k_track_lm keeps them in registers while it stays on a
// level: the template does not move between evaluations); later trips (n > TRK_UNROLL * stride) load theirs.
template <bool MASK, bool PRE = false, int UNR = TRK_UNROLL>
__device__ __forceinline__ void track_accumulate(const sdso_track_eval_t& EV, const float4* __restrict__ pc, const float4* __restrict__ img, int n,
                                                 int first, int stride, uint8_t* __restrict__ mask, TrackLaneSums& Sout, const float4* qpre = nullptr) {
  TrackLaneSums S;   // a local, copied out at the end: accumulating through the reference cost 40 VGPRs (196 instead of 154: 2 waves per SIMD instead of 3)
  const int lvl = EV.lvl, wl = EV.w, hl = EV.h;
  const float fxl = EV.fx, fyl = EV.fy, cxl = EV.cx, cyl = EV.cy;
  const float affLL0 = EV.affLL[0], affLL1 = EV.affLL[1];
  const float b0 = EV.ref_b0, cutoffTH = EV.cutoffTH, huberTH = EV.huberTH;
  const float maxEnergy = 2 * huberTH * cutoffTH - huberTH * huberTH;
  float RKi[9], Ki[9], t[3];
#pragma unroll
  for (int k = 0; k < 9; k++) { RKi[k] = EV.RKi[k]; Ki[k] = EV.Ki[k]; }
#pragma unroll
  for (int k = 0; k < 3; k++) t[k] = EV.t[k];
  const float wlm3 = (float)(wl - 3), hlm3 = (float)(hl - 3);
  float* acc = S.acc;
#pragma unroll
  for (int k = 0; k < 45; k++) acc[k] = 0.f;
  float E = 0.f, sT = 0.f, sRT = 0.f;
  int nE = 0, nSat = 0, nWarp = 0, nShift = 0;

  // UNR template points per lane and trip, in three straight-line stages so that the memory system sees
  // all of a trip's requests at once: (1) the pc loads, (2) projection + bounds test + the 4 bilinear taps of every
  // point (an out-of-bounds point reads pixel (2,2) instead of branching around its loads), (3) residual, Huber,
  // the 45 products — in point order, so the per-lane sums are those of the one-point-per-trip loop.
  for (int i0 = first; i0 < n; i0 += UNR * stride) {
    float4 q[UNR];
#pragma unroll
    for (int s = 0; s < UNR; s++) {
      const int i = i0 + s * stride;
      if (PRE && i0 == first) q[s] = qpre[s]; else q[s] = pc[i < n ? i : i0];
    }
    LMS(1);
    float us[UNR], vs[UNR], nid[UNR];
    bool ok[UNR];
    float3 hits[UNR];
#pragma unroll
    for (int s = 0; s < UNR; s++) {
      const int i = i0 + s * stride;
      ok[s] = false; us[s] = 0.f; vs[s] = 0.f; nid[s] = 0.f; hits[s] = make_float3(0.f, 0.f, 0.f);
      if (i0 - first + s * stride >= n) continue;            // (uniform over the launch's threads: the whole slot is past the end — the coarse levels have fewer points than threads)
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
    LMS(2);
#pragma unroll
    for (int s = 0; s < UNR; s++) {
      const int i = i0 + s * stride;
      if (i0 - first + s * stride >= n) continue;
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
  S.E = E; S.sT = sT; S.sRT = sRT; S.nE = nE; S.nSat = nSat; S.nWarp = nWarp; S.nShift = nShift;
  Sout = S;
}

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
  TrackLaneSums S;
  track_accumulate<MASK>(P.ev, P.pc, P.img, n, bx * TRK_BLOCK + threadIdx.x, gx * TRK_BLOCK, mask, S);
  float* acc = S.acc;
  const float E = S.E, sT = S.sT, sRT = S.sRT;
  const int nE = S.nE, nSat = S.nSat, nWarp = S.nWarp, nShift = S.nShift;

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
SDSO_HD static void fill_eval(const sdso_track_params_t& p, int lvl, const Se3& T, const sdso_aff_t& aff, float cutoff, sdso_track_eval_t& ev) {
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
  if (const char* e = dbg_env("SDSO_TRK_GX")) gx = std::max(1, std::min(by_points, atoi(e)));   // experiment
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
  launch_timed(ctx, "k_track_eval", 1, k_track_eval<false>, dim3(nblk), dim3(TRK_BLOCK), (const TrackProb*)tb->d_probs, tb->nprob, tb->gx, tb->d_partF, tb->d_partI, (uint8_t*)nullptr);
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

// ------------------------------------------------------------------ trackNewestCoarse
// CoarseTracker::trackNewestCoarse (CoarseTracker.cpp:827-1069, DSO-native LM :908-1024) is a chain of calcRes+calcGSSSE
// evaluations with a little 8x8 algebra in between: LmCore is that state machine, written once for host and device.  It always
// has exactly one evaluation pending (first evaluation of a level, repeat with a doubled cutoff, or the trial step of an LM
// iteration).  Two drivers:
//   * k_track_lm (default): ONE launch runs the whole call — a cluster of up to eight 512-thread workgroups per motion hypothesis
//     (FullSystem::trackNewCoarse tries up to 53 of them, FullSystem.cpp:305-441; the clusters run side by side), all threads evaluate the
//     pending calcRes+calcGSSSE over the level's points, the members exchange their partial sums once, and every member solves the 8x8
//     system on one wave, applies SE3::exp and takes the accept / reject and level decisions on the same sums (see the cluster notes at
//     the kernel).  No host round trip per evaluation (it cost 25 us of launch + synchronisation each, 28 times per call).
//   * the lock-step host loop (SDSO_TRK_HOST_LM=1): every round evaluates the pending requests of all hypotheses in one k_track_eval
//     launch.  Same LmCore, same sequence of evaluations.
namespace sdso {
struct LmCore {
  sdso_track_params_t p;
  sdso_track_result_t out;
  Se3 cur, Tnew;
  sdso_aff_t affCur, affNew;
  sdso_se3_t T_final; sdso_aff_t aff_final;     // what lastToNew / aff_g2l receive (only when the call reaches its end)
  bool wrote_final, haveRepeated, done;
  int lvl, iteration, phase;                    // phase 0: first / repeated evaluation of a level, 1: trial step
  float levelCutoffRepeat, lambda;
  double oldres[6];                             // calcRes' Vec6 of the accepted state
  double H[64], b[8], inc[8];
  Se3 reqT; sdso_aff_t reqAff;                  // the evaluation this hypothesis waits for
  double wHl[64], wHs[64], wnb[8], wbs[8], wx[8], wwork[80];   // work space of solve_inc() (members: in LDS on the device)
  int wperm[8];

  SDSO_HD void init(const sdso_track_params_t& prm, const sdso_se3_t& T0, const sdso_aff_t& aff0) {
    p = prm;
    for (int i = 0; i < 5; i++) { out.lastResiduals[i] = NAN; out.iterations[i] = 0; }
    for (int i = 0; i < 3; i++) out.lastFlowIndicators[i] = 1000;
    out.evaluations = 0; out.point_evals = 0; out.good = 0;
    for (int i = 0; i < 9; i++) cur.R[i] = T0.R[i];
    for (int i = 0; i < 3; i++) cur.t[i] = T0.t[i];
    affCur = aff0;
    T_final = T0; aff_final = aff0; wrote_final = false;
    haveRepeated = false; done = false;
    iteration = 0; lambda = 0.01f;
    lvl = p.coarsestLvl;
    for (int i = 0; i < 8; i++) inc[i] = 0;
    for (int i = 0; i < 6; i++) oldres[i] = 0;
    start_level();
  }
  SDSO_HD void request(const Se3& T, const sdso_aff_t& a) { reqT = T; reqAff = a; }
  SDSO_HD void start_level() { levelCutoffRepeat = 1; phase = 0; request(cur, affCur); }
  SDSO_HD void finish() {   // :1044-1068
    done = true;
    wrote_final = true;
    for (int i = 0; i < 9; i++) T_final.R[i] = cur.R[i];
    for (int i = 0; i < 3; i++) T_final.t[i] = cur.t[i];
    aff_final = affCur;
    if ((p.affineOptModeA != 0 && (fabsf((float)aff_final.a) > 1.2)) || (p.affineOptModeB != 0 && (fabsf((float)aff_final.b) > 200))) return;
    double rel[2];
    affFromTo(p.ref_exposure, p.new_exposure, p.ref_aff_g2l.a, p.ref_aff_g2l.b, aff_final.a, aff_final.b, rel);
    const float r0 = (float)rel[0], r1 = (float)rel[1];
    if ((p.affineOptModeA == 0 && (fabsf(logf(r0)) > 1.5)) || (p.affineOptModeB == 0 && (fabsf(r1) > 200))) return;
    if (p.affineOptModeA < 0) aff_final.a = 0;
    if (p.affineOptModeB < 0) aff_final.b = 0;
    out.good = 1;
  }
  SDSO_HD void finish_level() {
    out.lastResiduals[lvl] = sqrtf((float)(oldres[0] / oldres[1]));
    out.lastFlowIndicators[0] = oldres[2]; out.lastFlowIndicators[1] = oldres[3]; out.lastFlowIndicators[2] = oldres[4];
    if (out.lastResiduals[lvl] > 1.5 * p.minResForAbort[lvl]) { done = true; return; }  // :1032 (good stays 0, pose untouched)
    if (levelCutoffRepeat > 1 && !haveRepeated) { lvl++; haveRepeated = true; }
    lvl--;
    if (lvl < 0) finish(); else start_level();
  }
  // Stage 1 of consuming an evaluation: the scalar decisions (:897-904, :1004-1023).  Returns 1 when an LM step has to be proposed
  // (then: if take_Hb copy H, b from the evaluation, solve_inc, propose_post); 0 when the next request (or `done`) is already set.
  SDSO_HD int consume_pre(const double* res, bool& take_Hb) {
    const float lambdaExtrapolationLimit = 0.001f;
    take_Hb = false;
    if (phase == 0) {
      for (int i = 0; i < 6; i++) oldres[i] = res[i];
      if (oldres[5] > 0.6 && levelCutoffRepeat < 50) { levelCutoffRepeat *= 2; request(cur, affCur); return 0; }   // :897-904
      take_Hb = true;
      lambda = 0.01f;
      iteration = 0;
    } else {
      const bool accept = (res[0] / res[1]) < (oldres[0] / oldres[1]);
      if (accept) {
        take_Hb = true;
        for (int i = 0; i < 6; i++) oldres[i] = res[i];
        affCur = affNew;
        cur = Tnew;
        lambda *= 0.5;
      } else {
        lambda *= 4;
        if (lambda < lambdaExtrapolationLimit) lambda = lambdaExtrapolationLimit;
      }
      double nrm = 0;
      for (int i = 0; i < 8; i++) nrm += inc[i] * inc[i];
      if (!(std::sqrt(nrm) > 1e-3)) { finish_level(); return 0; }
      iteration++;
    }
    if (iteration >= p.maxIterations[lvl]) { finish_level(); return 0; }
    out.iterations[lvl]++;
    return 1;
  }
  // Stage 2 (host form): inc from (H, b, lambda) and the affine modes (:931-964)
  SDSO_HD void solve_inc() {
    double* Hl = wHl; double* nb = wnb;
    for (int i = 0; i < 64; i++) Hl[i] = H[i];
    for (int i = 0; i < 8; i++) Hl[i * 8 + i] *= (1 + lambda);
    for (int i = 0; i < 8; i++) nb[i] = -b[i];
    solveLdltSmall(Hl, 8, 8, nb, inc, wwork, wperm);
    if (p.affineOptModeA < 0 && p.affineOptModeB < 0) {  // fix a, b (:937-940)
      double* x6 = wx;
      solveLdltSmall(Hl, 8, 6, nb, x6, wwork, wperm);
      for (int i = 0; i < 6; i++) inc[i] = x6[i];
      inc[6] = inc[7] = 0;
    }
    if (!(p.affineOptModeA < 0) && p.affineOptModeB < 0) {  // fix b (:943-946)
      double* x7 = wx;
      solveLdltSmall(Hl, 8, 7, nb, x7, wwork, wperm);
      for (int i = 0; i < 7; i++) inc[i] = x7[i];
      inc[7] = 0;
    }
    if (p.affineOptModeA < 0 && !(p.affineOptModeB < 0)) {  // fix a (:949-964)
      double* Hs = wHs; double* bs = wbs; double* x7 = wx;
      for (int i = 0; i < 64; i++) Hs[i] = Hl[i];
      for (int i = 0; i < 8; i++) bs[i] = -b[i];
      for (int i = 0; i < 8; i++) Hs[i * 8 + 6] = Hs[i * 8 + 7];
      for (int j = 0; j < 8; j++) Hs[6 * 8 + j] = Hs[7 * 8 + j];
      bs[6] = bs[7];
      solveLdltSmall(Hs, 8, 7, bs, x7, wwork, wperm);
      for (int i = 0; i < 8; i++) inc[i] = 0;
      for (int i = 0; i < 6; i++) inc[i] = x7[i];
      inc[7] = x7[6];
    }
  }
  // Stage 3: extrapolation, scaling, SE3::exp and the request of the trial evaluation (:966-1000)
  SDSO_HD void propose_post() {
    const float lambdaExtrapolationLimit = 0.001f;
    float extrapFac = 1;
    if (lambda < lambdaExtrapolationLimit) extrapFac = sqrtf(sqrtf(lambdaExtrapolationLimit / lambda));
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
  // host form of the whole consumption of one evaluation
  void consume(const TrackOut& O) {
    bool take = false;
    if (!consume_pre(O.res, take)) return;
    if (take) { for (int i = 0; i < 64; i++) H[i] = O.H[i]; for (int i = 0; i < 8; i++) b[i] = O.b[i]; }
    solve_inc();
    propose_post();
  }
};

// one hypothesis of the resident driver
struct LmJob {
  sdso_track_params_t p;
  const float4* pc[SDSO_PYR_LEVELS];
  const float4* img[SDSO_PYR_LEVELS];
  int n[SDSO_PYR_LEVELS];
  sdso_se3_t T;            // in: initial lastToNew; out: the call's result (unchanged when the call aborts, like the reference's references)
  sdso_aff_t aff;
  sdso_track_result_t out;
};
#ifndef LM_BLOCK_THREADS
#define LM_BLOCK_THREADS 512
#endif
#ifndef LM_UNROLL
#define LM_UNROLL 4
#endif
constexpr int LM_BLOCK = LM_BLOCK_THREADS;    // 512: 8 waves, the evaluation body wants ~200 VGPRs at four points per trip (two waves per SIMD)
}  // namespace sdso

// ---- the LM step of the resident driver, by the 64 lanes of wave 0 ------------------------------------------------------------
__device__ __forceinline__ double lm_readlane(double v, int src) {
  const unsigned long long u = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, src), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), src);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
// value of lane 8 (lane / 8) + k: ds_swizzle_b32 in bit mode (and 0x18, or k) — a broadcast inside every group of eight lanes, no address register
template <int K>
__device__ __forceinline__ double lm_bcast8_c(double v) {
  const unsigned long long u = __double_as_longlong(v);
  constexpr int pat = 0x18 | (K << 5);
  const unsigned lo = (unsigned)__builtin_amdgcn_ds_swizzle((int)(unsigned)u, pat), hi = (unsigned)__builtin_amdgcn_ds_swizzle((int)(unsigned)(u >> 32), pat);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double lm_bcast8(double v, int k) {   // k is a constant after unrolling
  switch (k) {
    case 0: return lm_bcast8_c<0>(v); case 1: return lm_bcast8_c<1>(v); case 2: return lm_bcast8_c<2>(v); case 3: return lm_bcast8_c<3>(v);
    case 4: return lm_bcast8_c<4>(v); case 5: return lm_bcast8_c<5>(v); case 6: return lm_bcast8_c<6>(v); default: return lm_bcast8_c<7>(v);
  }
}
// x = A^-1 rhs for the leading n x n block (n <= 8) of a symmetric A: the algorithm of solveLdltSmall / Eigen::LDLT (symmetric pivoting
// on the first largest |diagonal| of the not yet eliminated positions, read from the INPUT matrix as Eigen's left-looking loop does), with
// the matrix spread over the wave — lane 8i + j holds A(i,j), every lane of row i holds rhs(i) — and every element updated by the
// expression the sequential code uses (the upper triangle mirrors the lower one: its lanes evaluate the lower element's expression with
// the roles swapped).  Because the pivot search only ever reads the input diagonal, the whole pivot order is known before the first
// elimination: every lane replays the eight selections on the eight diagonal values (wave-uniform arithmetic), the matrix is exchanged
// ONCE, and the eight elimination steps are readlane -> divide -> two broadcasts -> update (the forward substitution rides along: same
// terms, same order as the sequential loop).  The division by D runs on all rows at once.  All lanes return with the same x[0..7].
// (History: a single lane walking these 64 doubles through LDS took ~19 us per solve; exchanging per step 4.2 us; this form 2.3 us.)
__device__ __forceinline__ void lm_wave_ldlt(double a, double rhs, int n, double* __restrict__ xs /* LDS, 8 doubles: x by ORIGINAL index */) {
  const int lane = threadIdx.x & 63, i = lane >> 3, j = lane & 7;
  // everything outside the leading n x n block is zero: a zero pivot leaves its column alone and contributes nothing anywhere, so the
  // eight steps below run unconditionally — straight-line code, selects instead of branches (the branchy form was 2 500 instructions)
  a = (i < n && j < n) ? a : 0.0;
  rhs = i < n ? rhs : 0.0;
  double dg[8];
  int perm[8];
#pragma unroll
  for (int m = 0; m < 8; m++) { dg[m] = fabs(lm_readlane(a, m * 9)); perm[m] = m; }
#pragma unroll
  for (int k = 0; k < 7; k++) {
    double best = dg[k];
    int p = k;
#pragma unroll
    for (int m = k + 1; m < 8; m++) { const bool gt = dg[m] > best; best = gt ? dg[m] : best; p = gt ? m : p; }
    const int pk = perm[k];
    int pp = pk;
#pragma unroll
    for (int m = k + 1; m < 8; m++) { const bool is = m == p; pp = is ? perm[m] : pp; dg[m] = is ? dg[k] : dg[m]; perm[m] = is ? pk : perm[m]; }
    perm[k] = pp;
    dg[k] = best;
  }
  int si = perm[0], sj = perm[0];
#pragma unroll
  for (int m = 1; m < 8; m++) { si = i == m ? perm[m] : si; sj = j == m ? perm[m] : sj; }
  a = __shfl(a, si * 8 + sj, 64);
  double y = __shfl(rhs, si * 8, 64);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const double dk = lm_readlane(a, k * 9);
    const double yk = lm_readlane(y, k * 8);
    const bool nz = dk != 0.0;                      // a zero pivot leaves its column as it is
    const double l = nz ? a / dk : a;               // column k below the diagonal: L(i,k)
    const double lik = lm_bcast8(l, k), ljk = __shfl(l, j * 8 + k, 64);
    const bool lower = i >= j;                      // the upper triangle mirrors the lower element (j,i): the same expression with the roles swapped
    const double an = a - ((lower ? lik : ljk) * dk) * (lower ? ljk : lik);   // A(i,j) -= (l_ik d_k) A(j,k)
    a = (nz && i > k && j > k) ? an : a;
    a = (nz && j == k && i > k) ? l : a;
    y = i > k ? y - lik * yk : y;                   // L z = rhs, term k of row i
  }
  // D, on every row at once
  double dmine = lm_readlane(a, 0);
#pragma unroll
  for (int m = 1; m < 8; m++) { const double d = lm_readlane(a, m * 9); dmine = i == m ? d : dmine; }
  const double w = dmine != 0.0 ? y / dmine : 0.0;
  // L^T x = w, every lane redundantly (values by v_readlane at fixed lanes): same summation order as the sequential code (the terms past n are 0 * 0)
  double yv[8];
#pragma unroll
  for (int r = 0; r < 8; r++) yv[r] = lm_readlane(w, r * 8);
#pragma unroll
  for (int r = 6; r >= 0; r--) {
    double sacc = yv[r];
#pragma unroll
    for (int c = r + 1; c < 8; c++) sacc -= lm_readlane(a, c * 8 + r) * yv[c];
    yv[r] = sacc;
  }
  // position r holds the unknown of original index perm[r]
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < 8; r++) xs[perm[r]] = yv[r];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// SE3::exp of host_math.h (expSe3 / expSo3, the same expressions element by element) for a WAVE-UNIFORM tangent: the four
// trigonometric values it needs — sin, cos of theta / 2 and of theta — come from ONE sincos evaluated on two lanes (lane 0: theta / 2,
// lane 1: theta) instead of four calls in a row on one lane; everything else is evaluated by every lane on the same numbers.
__device__ __forceinline__ Se3 lm_exp_se3_wave(const double* xi) {
  const V3 om{{xi[3], xi[4], xi[5]}};
  const double th2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
  const double th = std::sqrt(th2);
  double sv, cv;
  sincos((threadIdx.x & 1) ? th : 0.5 * th, &sv, &cv);
  const double s_half = lm_readlane(sv, 0), c_half = lm_readlane(cv, 0), s_full = lm_readlane(sv, 1), c_full = lm_readlane(cv, 1);
  double im, re;
  if (th < kSophusEps) {
    const double th4 = th2 * th2;
    im = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
    re = 1.0 - 0.5 * th2 + (1.0 / 384.0) * th4;
  } else {
    im = s_half / th;
    re = c_half;
  }
  Se3 T;
  T.R = rotationFromQuat(re, im * om[0], im * om[1], im * om[2]);
  const M3 Om = skew(om);
  const M3 Om2 = mul(Om, Om);
  M3 V;
  if (th < kSophusEps) {
    V = T.R;
  } else {
    const double a = (1.0 - c_full) / (th * th);
    const double b = (th - s_full) / (th * th * th);
#pragma unroll
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * Om[i] + b * Om2[i];
  }
  T.t = mul(V, V3{{xi[0], xi[1], xi[2]}});
  return T;
}

// wave 0 of k_track_lm, between two evaluations: finalise the sums (calcGSSSE :580-595, calcRes :783-789, the expressions of
// k_track_finalize), take the LM decisions (LmCore::consume_pre), solve for the increment, propose the trial pose
// (LmCore::propose_post) and build the request of the next evaluation (fill_eval) — the serial part of a call, 28 times per call.
// Everything here is WAVE-UNIFORM arithmetic: all 64 lanes evaluate the same scalar expressions on the same numbers (LDS broadcast
// reads), so the loads of a stage are requested together, nothing waits for one lane's chain of LDS round trips, and only the
// stores are lane 0's.  (History: decisions, SE3::exp and fill_eval as scalar code of lane 0 / thread 0 with LmCore in LDS between
// them: 2 150 + 3 830 + 2 150 cycles per evaluation, profiles/r04_lm_stamps.txt.)  The common case — the evaluation is consumed and
// another LM step is proposed on the same level — runs in this form; what ends a level or repeats an evaluation with a doubled
// cut-off (five to ten times per call) goes through LmCore's own methods on lane 0, exactly as the host driver runs them.
// The accepted system lives in the wave's registers — lane 8i + j holds H(i,j) and b(i) — between the evaluations (Hacc / bacc).
// Returns the call's `done`; otherwise `ev`, `s_lvl` and the call's counters are those of the next evaluation.
__device__ __forceinline__ bool lm_wave_step(LmCore& core, const float* F, const int* I, double& Hacc, double& bacc, sdso_track_eval_t& ev,
                                             const float (*s_Ki)[9], const int* s_n, int& s_lvl) {
  const int lane = threadIdx.x & 63;
  const int i = lane >> 3, j = lane & 7;
  // ---- every LDS input of the decision, requested together
  const int nE = I[0], nSat = I[1], nWarp = I[2], nShift = I[3];
  const float f45 = F[45], f46 = F[46], f47 = F[47];
  const int phase = core.phase, lvl = core.lvl, it0 = core.iteration;
  const float lam0 = core.lambda, lcr = core.levelCutoffRepeat;
  const double old0 = core.oldres[0], old1 = core.oldres[1];
  const int maxIt = core.p.maxIterations[lvl];
  double nrm = 0;
#pragma unroll
  for (int r = 0; r < 8; r++) { const double v = core.inc[r]; nrm += v * v; }
  const int npad = (nWarp + 3) & ~3;
  double Hnew, bnew;
  {
    auto scale_of = [](int k) -> double { return k < 3 ? (double)SCALE_XI_ROT : k < 6 ? (double)SCALE_XI_TRANS : k == 6 ? (double)SCALE_A : (double)SCALE_B; };
    const float inv_n = 1.0f / npad;
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const float fh = F[lo * 9 - lo * (lo - 1) / 2 + (hi - lo)], fb = F[i * 9 - i * (i - 1) / 2 + (8 - i)];
    double v = npad > 0 ? (double)fh * (double)inv_n : 0.0;
    v *= scale_of(j); v *= scale_of(i);
    Hnew = v;
    double u = npad > 0 ? (double)fb * (double)inv_n : 0.0;
    u *= scale_of(i);
    bnew = u;
  }
  LMS(7);
  double res[6];
  res[0] = (double)f45;
  res[1] = (double)nE;
  res[2] = (double)f46 / ((double)(float)nShift + 0.1);
  res[3] = 0;
  res[4] = (double)f47 / ((double)(float)nShift + 0.1);
  res[5] = (double)((float)nSat / (float)nE);
  // ---- LmCore::consume_pre, the case that proposes another step on this level (uniform); anything else: lane 0, below
  const float lambdaExtrapolationLimit = 0.001f;
  bool fast, take, accept = false;
  float lambda = lam0;
  int iteration = it0;
  if (phase == 0) {
    fast = !(res[5] > 0.6 && lcr < 50) && 0 < maxIt;                                  // :897-904
    take = true; lambda = 0.01f; iteration = 0;
  } else {
    accept = (res[0] / res[1]) < (old0 / old1);                                       // :1004
    take = accept;
    if (accept) lambda *= 0.5;
    else { lambda *= 4; if (lambda < lambdaExtrapolationLimit) lambda = lambdaExtrapolationLimit; }
    iteration = it0 + 1;
    fast = std::sqrt(nrm) > 1e-3 && iteration < maxIt;                                // :1022, :927
  }
  int act = 1;
  if (fast) {
    if (lane == 0) {
      if (take) {
#pragma unroll
        for (int r = 0; r < 6; r++) core.oldres[r] = res[r];
      }
      if (phase == 1 && accept) { core.affCur = core.affNew; core.cur = core.Tnew; }
      core.lambda = lambda; core.iteration = iteration;
      core.out.iterations[lvl]++;
    }
  } else {
    int a = 0, t = 0;
    if (lane == 0) { bool tk = false; a = core.consume_pre(res, tk); t = tk ? 1 : 0; }
    act = __builtin_amdgcn_readfirstlane(a);
    take = __builtin_amdgcn_readfirstlane(t) != 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (act) lambda = core.lambda;               // (cannot happen with the predicates above; kept so that the two forms can never disagree silently)
  }
  LMS(8);
  if (act) {
    if (take) { Hacc = Hnew; bacc = bnew; }
    // LmCore::solve_inc on the wave
    const double lam1 = 1 + lambda;
    double a = Hacc;
    if (i == j) a *= lam1;
    const double nb = -bacc;
    const bool fixA = core.p.affineOptModeA < 0, fixB = core.p.affineOptModeB < 0;
    // the full solve, then (when an affine parameter is fixed) the reduced one of :937-964 — ONE copy of the factorisation in the code: the
    // kernel's loop has to stay inside the instruction cache
    double incv[8];
    const int npass = (fixA || fixB) ? 2 : 1;
#pragma unroll 1
    for (int pass = 0; pass < npass; pass++) {
      double am = a, bm = nb;
      int n = 8;
      if (pass == 1) {
        n = (fixA && fixB) ? 6 : 7;
        if (fixA && !fixB) {   // rows / columns 6 <- 7 of the damped matrix, b likewise (:949-964)
          const int si = i == 6 ? 7 : i, sj = j == 6 ? 7 : j;
          am = __shfl(a, si * 8 + sj, 64);
          bm = __shfl(nb, si * 8, 64);
        }
      }
      lm_wave_ldlt(am, bm, n, core.wx);
      double x[8];
#pragma unroll
      for (int r = 0; r < 8; r++) x[r] = core.wx[r];
      if (pass == 0) {
#pragma unroll
        for (int r = 0; r < 8; r++) incv[r] = x[r];
      } else if (fixA && fixB) {
#pragma unroll
        for (int r = 0; r < 6; r++) incv[r] = x[r];
        incv[6] = incv[7] = 0;
      } else if (fixB) {
#pragma unroll
        for (int r = 0; r < 7; r++) incv[r] = x[r];
        incv[7] = 0;
      } else {
#pragma unroll
        for (int r = 0; r < 8; r++) incv[r] = 0;
#pragma unroll
        for (int r = 0; r < 6; r++) incv[r] = x[r];
        incv[7] = x[6];
      }
    }
    LMS(9);
    // the state the proposal starts from and the level's constants of the next request, requested together (lane 0's stores above are
    // behind the wave barrier that ends lm_wave_ldlt)
    Se3 cur;
#pragma unroll
    for (int r = 0; r < 9; r++) cur.R[r] = core.cur.R[r];
#pragma unroll
    for (int r = 0; r < 3; r++) cur.t[r] = core.cur.t[r];
    const sdso_aff_t affCur = core.affCur;
    const float fxl = core.p.fx[lvl], fyl = core.p.fy[lvl], cxl = core.p.cx[lvl], cyl = core.p.cy[lvl];
    const int wl = core.p.w[lvl], hl = core.p.h[lvl];
    float Ki[9];
#pragma unroll
    for (int r = 0; r < 9; r++) Ki[r] = s_Ki[lvl][r];
    const float expR = core.p.ref_exposure, expN = core.p.new_exposure;
    const double refA = core.p.ref_aff_g2l.a, refB = core.p.ref_aff_g2l.b;
    const float cutoff = core.p.coarseCutoffTH * core.levelCutoffRepeat, huber = core.p.huberTH;
    // LmCore::propose_post (:966-1000), uniform
    float extrapFac = 1;
    if (lambda < lambdaExtrapolationLimit) extrapFac = sqrtf(sqrtf(lambdaExtrapolationLimit / lambda));
    double incScaled[8];
#pragma unroll
    for (int r = 0; r < 8; r++) { incv[r] *= extrapFac; incScaled[r] = incv[r]; }
#pragma unroll
    for (int r = 0; r < 3; r++) incScaled[r] *= SCALE_XI_ROT;
#pragma unroll
    for (int r = 3; r < 6; r++) incScaled[r] *= SCALE_XI_TRANS;
    incScaled[6] *= SCALE_A;
    incScaled[7] *= SCALE_B;
    double sum = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) sum += incScaled[r];
    if (!std::isfinite(sum)) {
#pragma unroll
      for (int r = 0; r < 8; r++) incScaled[r] = 0;
    }
    const Se3 Tnew = lm_exp_se3_wave(incScaled) * cur;
    sdso_aff_t affNew = affCur;
    affNew.a += incScaled[6];
    affNew.b += incScaled[7];
    // fill_eval for (Tnew, affNew)
    sdso_track_eval_t evl;
    evl.lvl = lvl; evl.w = wl; evl.h = hl;
    evl.fx = fxl; evl.fy = fyl; evl.cx = cxl; evl.cy = cyl;
    float Rf[9];
#pragma unroll
    for (int r = 0; r < 9; r++) { evl.Ki[r] = Ki[r]; Rf[r] = (float)Tnew.R[r]; }
    mul3f(Rf, Ki, evl.RKi);
#pragma unroll
    for (int r = 0; r < 3; r++) evl.t[r] = (float)Tnew.t[r];
    double a2[2];
    affFromTo(expR, expN, refA, refB, affNew.a, affNew.b, a2);
    evl.affLL[0] = (float)a2[0]; evl.affLL[1] = (float)a2[1];
    evl.ref_b0 = (float)refB;
    evl.cutoffTH = cutoff;
    evl.huberTH = huber;
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < 8; r++) core.inc[r] = incv[r];
      core.Tnew = Tnew; core.affNew = affNew;
      core.phase = 1;
      core.reqT = Tnew; core.reqAff = affNew;
      ev = evl;
      core.out.evaluations++;
      core.out.point_evals += s_n[lvl];
    }
    LMS(10);
    return false;
  }
  LMS(10);
  // lane 0 has set the next request itself (a new level, a repeated evaluation) or ended the call
  int done = 0;
  if (lane == 0) {
    done = core.done ? 1 : 0;
    if (!done) {
      fill_eval(core.p, core.lvl, core.reqT, core.reqAff, core.p.coarseCutoffTH * core.levelCutoffRepeat, ev);
      s_lvl = core.lvl;
      core.out.evaluations++;
      core.out.point_evals += s_n[core.lvl];
    }
  }
  return __builtin_amdgcn_readfirstlane(done) != 0;
}

// ---- a CLUSTER of G workgroups per hypothesis --------------------------------------------------------------------------------------
// One CU evaluates a 4 000-point level at the rate its L1 is filled (two 128-byte lines per template point at 64 bytes per clock: the
// point loop of a single workgroup was 55 % of the call).  With G > 1 the hypothesis' points are strided over G workgroups on G CUs (placed
// on ONE XCD: workgroup L runs on XCD L % 8).  EVERY member runs the LM state machine: per evaluation a member publishes its 52 partial
// sums in an LmCluster record in global memory, collects the other members', adds all of them in member order — so every member holds the
// same sums, bit for bit — and takes the same decisions with the same arithmetic: the next request never has to travel.  ONE hand-off per
// evaluation (a leader that gathers the partials and publishes the next request needs two: 0.355 against 0.31 ms per call).
// A partial travels as 64-bit {word, evaluation number} pairs written and polled by single relaxed agent-scope atomics: a reader that sees
// the tag of evaluation e has that evaluation's word — no flag, no fence, one round trip (≈ 0.7 us on one XCD, tools/handoff_bench.hip).
// Two buffers alternate: a member can be one evaluation ahead of a slow reader of its previous partial, never two.  The tags grow from call
// to call (e_base): the records are never cleared between calls — whatever an earlier call left carries a smaller tag.
// Every spin is bounded (a member that never became resident — the device was shared — ends the call with out.evaluations = -1 and
// the host repeats it with G = 1, which needs no co-residency).  Member 0 reports the result.
constexpr int LM_MAXG = 8;
constexpr int LM_SPIN_LIMIT = 1 << 21;
struct LmCluster {                                           // zeroed when allocated; a call's first evaluation is number e_base + 1
  unsigned long long part[2][LM_MAXG][64];
};
__device__ __forceinline__ void lm_put(unsigned long long* slot, unsigned word, int e) {
  __hip_atomic_store(slot, (unsigned long long)word | ((unsigned long long)(unsigned)e << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long lm_get(const unsigned long long* slot) {
  return __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(LM_BLOCK) void k_track_lm(LmJob* __restrict__ jobs, LmCluster* __restrict__ clusters, int nhyp, int G, int spin_limit, int drop_member /* test hook: member G - 1 of every cluster never answers */,
                                                       int solo_n /* levels of at most this many points are not shared */, int e_base /* this call's evaluations carry the tags e_base + 1 .. */) {
  // hypothesis c, member g: for G > 1 the members of a cluster share blockIdx % 8 (one XCD, one L2); speed only, any placement is correct
  int c = blockIdx.x, g = 0;
  if (G > 1) { const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3; g = j % G; c = (j / G) * 8 + xcd; }
  if (c >= nhyp) return;
  if (drop_member && G > 1 && g == G - 1) return;
  LmJob& J = jobs[c];
  LmCluster& C = clusters[c];
  __shared__ __align__(16) unsigned char core_raw[sizeof(LmCore)];     // (LmCore has member initialisers: raw storage, init() sets every field it reads)
  LmCore& core = *reinterpret_cast<LmCore*>(core_raw);
  __shared__ sdso_track_eval_t ev;
  __shared__ float sF[LM_BLOCK / 64][TRK_NF + TRK_NI];
  __shared__ float F[TRK_NF];
  __shared__ int I[TRK_NI];
  __shared__ int s_lvl, s_done, s_abort;
  __shared__ const float4* s_pc[SDSO_PYR_LEVELS];                      // the job's tables, read once (a global round trip per evaluation otherwise)
  __shared__ const float4* s_img[SDSO_PYR_LEVELS];
  __shared__ int s_n[SDSO_PYR_LEVELS];
  const int tid = threadIdx.x, wv = tid >> 6;
  const bool leader = g == 0;
  __shared__ float s_Ki[SDSO_PYR_LEVELS][9];                           // K[lvl]^-1 (fill_eval's inv3f, CoarseTracker.cpp:129-130): per level, not per evaluation
  if (tid < SDSO_PYR_LEVELS) {
    s_pc[tid] = J.pc[tid]; s_img[tid] = J.img[tid]; s_n[tid] = J.n[tid];
    const float K[9] = {J.p.fx[tid], 0, J.p.cx[tid], 0, J.p.fy[tid], J.p.cy[tid], 0, 0, 1};
    float Ki[9];
    inv3f(K, Ki);
    for (int k = 0; k < 9; k++) s_Ki[tid][k] = Ki[k];
  }
  if (tid == 0) {
    core.init(J.p, J.T, J.aff);                 // every member: the same state machine on the same inputs
    s_done = 0; s_abort = 0;
  }
  __syncthreads();
#ifdef SDSO_LM_STAMPS
  if (tid == 0) { for (int k = 0; k < 16; k++) lm_st_acc[k] = 0; lm_st_last = __builtin_amdgcn_s_memtime(); }
#define LMSL(i) do { if (leader) LMS(i); } while (0)
#else
#define LMSL(i) do { } while (0)
#endif
  double Hacc = 0.0, bacc = 0.0;                // wave 0: the accepted system (lm_wave_step)
  float4 qc[LM_UNROLL];                         // this thread's template points of level qlvl
  int qlvl = -1;
#pragma unroll
  for (int u = 0; u < LM_UNROLL; u++) qc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  // A level whose points fit ONE trip of one workgroup (LM_UNROLL points per thread: the coarse levels, more than half of a call's
  // evaluations) is evaluated by every member in full, in the single workgroup's order: the members hold the same sums without the
  // exchange — which costs more (5 k cycles at G = 8) than those points do.  solo_n == 0 (SDSO_TRK_LM_SOLO=0): every level is shared.
  int first = g * LM_BLOCK + tid, stride = G * LM_BLOCK;
  // every trip is one evaluation; the loop ends for all threads together (the flags are read behind a barrier)
  if (tid == 0) {                               // the first request; every later one is built by wave 0 at the end of lm_wave_step
    fill_eval(core.p, core.lvl, core.reqT, core.reqAff, core.p.coarseCutoffTH * core.levelCutoffRepeat, ev);
    s_lvl = core.lvl;
    core.out.evaluations++;
    core.out.point_evals += s_n[core.lvl];
  }
  __syncthreads();
  int shared_evals = 0;
  for (int e = 1; e <= 1024; e++) {
    LMSL(0);
    const int lvl = s_lvl, n = s_n[lvl];
    const bool shared_lvl = G > 1 && n > solo_n;
    if (lvl != qlvl) {                          // (uniform) first evaluation on this level: the points move into registers
      first = shared_lvl ? g * LM_BLOCK + tid : tid; stride = shared_lvl ? G * LM_BLOCK : LM_BLOCK;
      const float4* __restrict__ pc = s_pc[lvl];
      if (first < n) {
#pragma unroll
        for (int u = 0; u < LM_UNROLL; u++) { const int i = first + u * stride; qc[u] = pc[i < n ? i : first]; }
      }
      qlvl = lvl;
    }
    TrackLaneSums S;
    track_accumulate<false, true, LM_UNROLL>(ev, s_pc[lvl], s_img[lvl], n, first, stride, nullptr, S, qc);
    LMSL(3);
    {   // the four counters ride along as floats (exact: they stay far below 2^24), so one 52-value row reduction covers everything
      float v52[TRK_NF + TRK_NI];
#pragma unroll
      for (int k = 0; k < 45; k++) v52[k] = S.acc[k];
      v52[45] = S.E; v52[46] = S.sT; v52[47] = S.sRT;
      v52[48] = (float)S.nE; v52[49] = (float)S.nSat; v52[50] = (float)S.nWarp; v52[51] = (float)S.nShift;
      wave_reduce_rows<TRK_NF + TRK_NI>(v52, [&](int k, float sum) { sF[wv][k] = sum; });
    }
    LMSL(4);
    __syncthreads();
    LMSL(5);
    float mine = 0.f;
    if (tid < TRK_NF + TRK_NI) {                // fixed order over the waves: run-to-run reproducible
      mine = sF[0][tid];
#pragma unroll
      for (int w = 1; w < LM_BLOCK / 64; w++) mine += sF[w][tid];
    }
    if (shared_lvl && tid < TRK_NF + TRK_NI) {  // publish this member's partial, collect the others', add all of them in member order
      unsigned long long (*buf)[64] = C.part[shared_evals & 1];   // (alternating over the SHARED evaluations: between two uses of a buffer lies an exchange on the other one)
      const float own = mine;
      lm_put(&buf[g][tid], __float_as_uint(own), e_base + e);
      unsigned long long v[LM_MAXG];
      bool all = false;
      for (int spins = 0; spins < spin_limit && !all; spins++) {
        all = true;
#pragma unroll
        for (int m = 0; m < LM_MAXG; m++)
          if (m < G && m != g) { v[m] = lm_get(&buf[m][tid]); all = all && (int)(v[m] >> 32) == e_base + e; }
        if (!all) __builtin_amdgcn_s_sleep(1);
      }
      if (!all) s_abort = 1;
      float tot = 0.f;
#pragma unroll
      for (int m = 0; m < LM_MAXG; m++) if (m < G) tot = m == 0 ? (g == 0 ? own : __uint_as_float((unsigned)v[0])) : tot + (m == g ? own : __uint_as_float((unsigned)v[m]));
      mine = tot;
    }
    if (shared_lvl) shared_evals++;
    if (tid < TRK_NF) F[tid] = mine; else if (tid < TRK_NF + TRK_NI) I[tid - TRK_NF] = (int)mine;
    __syncthreads();
    if (s_abort) {                              // a member never answered (every member notices): member 0 gives the call back to the host
      if (leader && tid == 0) { J.out = core.out; J.out.evaluations = -1; }
      return;
    }
    LMSL(6);
    if (wv == 0) { const bool d = lm_wave_step(core, F, I, Hacc, bacc, ev, s_Ki, s_n, s_lvl); if (tid == 0) s_done = d ? 1 : 0; }
    __syncthreads();
    LMSL(11);
    if (s_done) break;                          // (every member reaches the same verdict)
  }
#undef LMSL
  if (leader && tid == 0) {
    J.out = core.out;
    if (core.wrote_final) { J.T = core.T_final; J.aff = core.aff_final; }
#ifdef SDSO_LM_STAMPS
    for (int k = 0; k < 9; k++) J.T.R[k] = (double)lm_st_acc[k];
    for (int k = 0; k < 3; k++) J.T.t[k] = (double)lm_st_acc[9 + k];
#endif
  }
}

namespace sdso {
static int resolve_job(sdso_ctx* ctx, int ref_slot, int frame_slot, const sdso_track_params_t& p, LmJob& J) {
  auto ir = ctx->refs.find(ref_slot);
  SDSO_REQUIRE(ctx, ir != ctx->refs.end(), "unknown ref slot");
  auto ip = ctx->pyr.find(frame_slot);
  SDSO_REQUIRE(ctx, ip != ctx->pyr.end(), "unknown frame slot");
  SDSO_REQUIRE(ctx, p.coarsestLvl < ip->second.levels, "level not in pyramid");
  for (int l = 0; l < SDSO_PYR_LEVELS; l++) { J.pc[l] = nullptr; J.img[l] = nullptr; J.n[l] = 0; }
  for (int l = 0; l <= p.coarsestLvl; l++) {
    // the kernel indexes the image with (w,h) of the params: they must be the uploaded level's size
    SDSO_REQUIRE(ctx, p.w[l] == ip->second.w[l] && p.h[l] == ip->second.h[l], "params w/h do not match the uploaded pyramid level");
    J.pc[l] = ir->second.pc[l]; J.img[l] = ip->second.d[l]; J.n[l] = ir->second.n[l];
  }
  J.p = p;
  return SDSO_OK;
}
}  // namespace sdso

// the lock-step host driver (A/B and fallback for SDSO_TRK_HOST_LM=1)
static int track_newest_coarse_host(sdso_ctx* ctx, int nhyp, const int* ref_slots, const int* frame_slots, const sdso_track_params_t* prms,
                                    sdso_se3_t* lastToNew, sdso_aff_t* aff_g2l, sdso_track_result_t* outs) {
  std::vector<LmCore> S(nhyp);
  for (int k = 0; k < nhyp; k++) S[k].init(prms[k], lastToNew[k], aff_g2l[k]);
  std::vector<TrackProb> probs;
  std::vector<int> who;
  for (;;) {
    probs.clear(); who.clear();
    for (int k = 0; k < nhyp; k++) {
      LmCore& s = S[k];
      if (s.done) continue;
      sdso_track_eval_t ev;
      fill_eval(s.p, s.lvl, s.reqT, s.reqAff, s.p.coarseCutoffTH * s.levelCutoffRepeat, ev);
      TrackProb P;
      int rc = resolve_prob(ctx, ref_slots[k], frame_slots[k], ev, P);
      if (rc) return rc;
      probs.push_back(P); who.push_back(k);
      s.out.evaluations++;
      s.out.point_evals += P.n;
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
  for (int k = 0; k < nhyp; k++) {
    outs[k] = S[k].out;
    if (S[k].wrote_final) { lastToNew[k] = S[k].T_final; aff_g2l[k] = S[k].aff_final; }
  }
  return SDSO_OK;
}

extern "C" int sdso_track_newest_coarse_batch(sdso_ctx* ctx, int nhyp, const int* ref_slots, const int* frame_slots, const sdso_track_params_t* prms,
                                              sdso_se3_t* lastToNew, sdso_aff_t* aff_g2l, sdso_track_result_t* outs) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  SDSO_REQUIRE(ctx, nhyp > 0 && ref_slots && frame_slots && prms && lastToNew && aff_g2l && outs, "null argument");
  for (int k = 0; k < nhyp; k++)
    SDSO_REQUIRE(ctx, prms[k].coarsestLvl >= 0 && prms[k].coarsestLvl < 5 && prms[k].coarsestLvl < prms[k].levels, "coarsestLvl out of range");  // assert :853
  static const bool host_lm = dbg_env("SDSO_TRK_HOST_LM") != nullptr;
  if (host_lm) return track_newest_coarse_host(ctx, nhyp, ref_slots, frame_slots, prms, lastToNew, aff_g2l, outs);
  // resident driver: jobs through pinned memory, one launch, one synchronisation.  The cluster records are the library's own allocation
  // and are never cleared between calls (tags, see LmCluster).  (The kernel reading the jobs in pinned host memory directly, without
  // the two copies, measured the same 0.305 ms per call: tools/time_track.py, round 5.)
  int rc = ensure_pinned(ctx, sizeof(LmJob) * (size_t)nhyp);
  if (rc) return rc;
  const size_t jobs_bytes = (sizeof(LmJob) * (size_t)nhyp + 255) & ~(size_t)255;
  rc = ensure_scratch(ctx, jobs_bytes);
  if (rc) return rc;
  if (ctx->lm_clusters_bytes < sizeof(LmCluster) * (size_t)nhyp || ctx->lm_epoch > (1 << 30)) {
    SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->lm_clusters_bytes < sizeof(LmCluster) * (size_t)nhyp) {
      if (ctx->lm_clusters) SDSO_HIP(ctx, hipFree(ctx->lm_clusters));
      ctx->lm_clusters = nullptr; ctx->lm_clusters_bytes = 0;
      const size_t want = sizeof(LmCluster) * (size_t)std::max(nhyp, 8);
      SDSO_HIP(ctx, hipMalloc(&ctx->lm_clusters, want));
      ctx->lm_clusters_bytes = want;
    }
    SDSO_HIP(ctx, hipMemsetAsync(ctx->lm_clusters, 0, ctx->lm_clusters_bytes, ctx->stream));
    ctx->lm_epoch = 0;
  }
  LmJob* hj = (LmJob*)ctx->pinned;
  for (int k = 0; k < nhyp; k++) {
    rc = resolve_job(ctx, ref_slots[k], frame_slots[k], prms[k], hj[k]);
    if (rc) return rc;
  }
  if (ctx->tb) ctx->tb->nprob = 0;   // (a prepared evaluation batch keeps its own buffers; nothing shared)
  LmJob* dj = (LmJob*)ctx->scratch;
  LmCluster* dc = (LmCluster*)ctx->lm_clusters;
  // workgroups per hypothesis: as many as keep the whole grid resident at once (the members of a cluster wait for each other; one
  // 512-thread workgroup of this kernel fills a CU), eight at most.  SDSO_TRK_LM_CLUSTER=1 forces single workgroups.
  const int g_env = dbg_env("SDSO_TRK_LM_CLUSTER") ? atoi(dbg_env("SDSO_TRK_LM_CLUSTER")) : 0;   // (read per call: the tests walk the cluster sizes)
  const int slots8 = 8 * ((nhyp + 7) / 8);
  int G = std::min(LM_MAXG, (ctx->n_cu * 7 / 8) / slots8);   // (an eighth of the CUs stays free: a grid that needs every CU waits on any straggler)
  if (g_env > 0) G = std::min(G, g_env);
  if (G < 2) G = 1;
  const int solo_n = dbg_env("SDSO_TRK_LM_SOLO") ? atoi(dbg_env("SDSO_TRK_LM_SOLO")) : LM_UNROLL * LM_BLOCK;
#ifdef SDSO_TEST_HOOKS
  // test hook, compiled into libsdso_hip_hooks.so only (csrc/Makefile; tests/test_variants_gpu.py): the first attempt loses one member of
  // every cluster, with a short spin limit — the call must come back through the single-workgroup repetition with its result
  const bool drop = dbg_env("SDSO_TRK_LM_TEST_DROP_MEMBER") != nullptr;
#else
  const bool drop = false;
#endif
  for (int attempt = 0; attempt < 2; attempt++) {
    for (int k = 0; k < nhyp; k++) { hj[k].T = lastToNew[k]; hj[k].aff = aff_g2l[k]; hj[k].out.evaluations = -1; }   // (-1 until member 0 reports)
    SDSO_HIP(ctx, hipMemcpyAsync(dj, hj, sizeof(LmJob) * nhyp, hipMemcpyHostToDevice, ctx->stream));
    const int e_base = ctx->lm_epoch;
    ctx->lm_epoch += 1040;             // (a call has at most 1024 evaluations)
    {
      ProfScope ps(ctx, "k_track_lm");
      hipLaunchKernelGGL(k_track_lm, dim3(G > 1 ? slots8 * G : nhyp), dim3(LM_BLOCK), 0, ctx->stream, dj, dc, nhyp, G, drop ? 1 << 12 : LM_SPIN_LIMIT, drop && G > 1 ? 1 : 0, solo_n, e_base);
    }
    SDSO_HIP(ctx, hipGetLastError());
    SDSO_HIP(ctx, hipMemcpyAsync(hj, dj, sizeof(LmJob) * nhyp, hipMemcpyDeviceToHost, ctx->stream));
    SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    bool gave_up = false;
    for (int k = 0; k < nhyp; k++) gave_up = gave_up || hj[k].out.evaluations < 0;
    if (!gave_up) break;
    SDSO_REQUIRE(ctx, G > 1, "k_track_lm gave up without a cluster");   // (cannot happen: single workgroups wait for nobody)
    G = 1;                             // a cluster was not co-resident (shared device): the single-workgroup form needs no co-residency
  }
  for (int k = 0; k < nhyp; k++) { outs[k] = hj[k].out; lastToNew[k] = hj[k].T; aff_g2l[k] = hj[k].aff; }
  return SDSO_OK;
}

extern "C" int sdso_track_newest_coarse(sdso_ctx* ctx, int ref_slot, int frame_slot, const sdso_track_params_t* prm,
                                        sdso_se3_t* lastToNew, sdso_aff_t* aff_g2l, sdso_track_result_t* out) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_REQUIRE(ctx, prm && lastToNew && aff_g2l && out, "null argument");
  return sdso_track_newest_coarse_batch(ctx, 1, &ref_slot, &frame_slot, prm, lastToNew, aff_g2l, out);
}
