// The fork's LIVE factors on the device (S3 lives in stereo.hip as k_trace_stereo_blk<1, .>).
//
//   EdgeSE3PosePhotoDSO              src/FullSystem/dso_g2o_edge.cpp:395-500, graph build CoarseTracker.cpp:600-792
//   EdgeLBASE3PosePhotoIdepthCamDSO  src/FullSystem/dso_g2o_edge.cpp:5-282,   graph build FullSystemOptimize.cpp:455-542
//
// The per-edge arithmetic (computeError / linearizeOplus, mixed float/double exactly as written there) is evaluated one
// edge per lane (tracker) or one pattern pixel per lane (LBA), bit-identical to oracle/orc_g2o.cpp.  What g2o does around
// the edges (Huber kernel, quadratic form, Levenberg-Marquardt control) is not part of the reference tree and not
// version-pinned (CMakeLists.txt:47-58); it is restated from g2o's published algorithm in sdso_g2o_track_newest_coarse,
// on the host, in double, around two kernels.  `SE3 * Vec3` is R*X + t with R = rotationMatrix() (Sophus goes through
// Eigen's quaternion product, so3.hpp:255-257; Eigen is not in the tree either).
#include "sdso_internal.h"
#include "host_math.h"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace sdso {

struct G2oEdgeSet {
  int cap = 0, n = 0, n_edges = 0;
  uint8_t* mask = nullptr;
  float4* xref = nullptr;    // {Xref, measurement}
};
struct G2oState {
  std::map<std::pair<int, int>, G2oEdgeSet> sets;   // (ref_slot, level)
  double* d_part = nullptr;                          // per-block partial systems
  int part_blocks = 0;
  int* d_cnt = nullptr;                              // {numTermsInE, numSaturated}
  float* d_flow = nullptr;                           // 4 floats per 32nd level-0 point
  int flow_cap = 0;
};
static std::map<sdso_ctx*, G2oState> g_g2o;

void release_g2o(sdso_ctx* ctx) {
  G2oState st;
  if (!reg_take(g_g2o, ctx, st)) return;
  for (auto& kv : st.sets) { if (kv.second.mask) hipFree(kv.second.mask); if (kv.second.xref) hipFree(kv.second.xref); }
  if (st.d_part) hipFree(st.d_part);
  if (st.d_cnt) hipFree(st.d_cnt);
  if (st.d_flow) hipFree(st.d_flow);
}

// sdso_track_release_ref: the edge sets built on a released template go with it
void release_g2o_ref(sdso_ctx* ctx, int ref_slot) {
  if (!reg_has(g_g2o, ctx)) return;
  G2oState& st = reg_get(g_g2o, ctx);
  for (auto it = st.sets.begin(); it != st.sets.end();) {
    if (it->first.first == ref_slot) {
      if (it->second.mask) hipFree(it->second.mask);
      if (it->second.xref) hipFree(it->second.xref);
      it = st.sets.erase(it);
    } else ++it;
  }
}

constexpr int kSysDoubles = 36 + 8 + 2;   // upper triangle of H, b, {chi2, robust chi2}
constexpr int kMaxLinBlocks = 128;

// dso_util.hpp:25-46
__device__ __forceinline__ bool check_boundary(double u, double v, int wl, int hl) { return (u - 2) < 0 || (u + 3) > wl || (v - 2) < 0 || (v + 3) > hl; }

struct TrackEdgeVal { double e; double J[8]; };

// EdgeSE3PosePhotoDSO::computeError + linearizeOplus for one edge (dso_g2o_edge.cpp:395-500)
template <bool JAC>
__device__ __forceinline__ TrackEdgeVal track_edge(const float4 xr, const float4* __restrict__ img, const sdso_g2o_track_eval_t& ev) {
  TrackEdgeVal r;
  r.e = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) r.J[k] = 0;
  const double X0 = xr.x, X1 = xr.y, X2 = xr.z;
  double Xc[3];
#pragma unroll
  for (int i = 0; i < 3; i++) Xc[i] = ((ev.R[i * 3 + 0] * X0 + ev.R[i * 3 + 1] * X1) + ev.R[i * 3 + 2] * X2) + ev.t[i];
  const double fx = ev.fx, fy = ev.fy, cx = ev.cx, cy = ev.cy;
  const double uvx = fx * (Xc[0] / Xc[2]) + cx;
  const double uvy = fy * (Xc[1] / Xc[2]) + cy;
  if (check_boundary(uvx, uvy, ev.w, ev.h)) return r;
  const float3 hit = interp33(img, (float)uvx, (float)uvy, ev.w);
  const double meas = xr.w;
  if (isfinite(hit.x)) r.e = (double)hit.x - ((double)ev.ab[0] * meas + (double)ev.ab[1]);
  if constexpr (JAC) {
    const double x = Xc[0], y = Xc[1], invz = 1.0 / Xc[2];
    const double u = x * invz, v = y * invz;
    const double dx = hit.y * fx, dy = hit.z * fy;
    r.J[0] = invz * dx;
    r.J[1] = invz * dy;
    r.J[2] = -invz * (u * dx + v * dy);
    r.J[3] = -(u * v * dx + (1 + v * v) * dy);
    r.J[4] = u * v * dy + (1 + u * u) * dx;
    r.J[5] = u * dy - v * dx;
    r.J[6] = (double)ev.ab[0] * (ev.b0 - meas);
    r.J[7] = -1;
  }
  return r;
}

// CoarseTracker::calcRes, fork-live body (CoarseTracker.cpp:600-792): one pc point per lane.
__global__ __launch_bounds__(256) void k_g2o_track_edges(int n, const float4* __restrict__ pc, const float4* __restrict__ img, sdso_g2o_track_eval_t ev,
                                                         uint8_t* __restrict__ mask, float4* __restrict__ xref, int* __restrict__ cnt,
                                                         float* __restrict__ flow) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int term = 0, sat = 0;
  if (i < n) {
    const float4 p = pc[i];
    const float x = p.x, y = p.y, id = p.z;
    float pt[3], kp[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
      kp[r] = (ev.Ki[r * 3 + 0] * x + ev.Ki[r * 3 + 1] * y) + ev.Ki[r * 3 + 2] * 1.0f;
      pt[r] = ((ev.RKi[r * 3 + 0] * x + ev.RKi[r * 3 + 1] * y) + ev.RKi[r * 3 + 2] * 1.0f) + ev.t_cull[r] * id;
    }
    const float u = pt[0] / pt[2], v = pt[1] / pt[2];
    const float Ku = ev.fx * u + ev.cx, Kv = ev.fy * v + ev.cy;
    const float new_idepth = id / pt[2];
    if (ev.lvl == 0 && i % 32 == 0) {   // :662-693, summed on the host in point order
      float ptT[3], ptT2[3], pt3[3];
#pragma unroll
      for (int r = 0; r < 3; r++) {
        const float rp = (ev.RKi[r * 3 + 0] * x + ev.RKi[r * 3 + 1] * y) + ev.RKi[r * 3 + 2] * 1.0f;
        ptT[r] = kp[r] + ev.t_cull[r] * id;
        ptT2[r] = kp[r] - ev.t_cull[r] * id;
        pt3[r] = rp - ev.t_cull[r] * id;
      }
      const float KuT = ev.fx * (ptT[0] / ptT[2]) + ev.cx, KvT = ev.fy * (ptT[1] / ptT[2]) + ev.cy;
      const float KuT2 = ev.fx * (ptT2[0] / ptT2[2]) + ev.cx, KvT2 = ev.fy * (ptT2[1] / ptT2[2]) + ev.cy;
      const float Ku3 = ev.fx * (pt3[0] / pt3[2]) + ev.cx, Kv3 = ev.fy * (pt3[1] / pt3[2]) + ev.cy;
      float* f = flow + (size_t)(i / 32) * 4;
      f[0] = (KuT - x) * (KuT - x) + (KvT - y) * (KvT - y);
      f[1] = (KuT2 - x) * (KuT2 - x) + (KvT2 - y) * (KvT2 - y);
      f[2] = (Ku - x) * (Ku - x) + (Kv - y) * (Kv - y);
      f[3] = (Ku3 - x) * (Ku3 - x) + (Kv3 - y) * (Kv3 - y);
    }
    uint8_t m = 0;
    float4 xr = make_float4(0, 0, 0, p.w);
    if (Ku > 2 && Kv > 2 && Ku < ev.w - 3 && Kv < ev.h - 3 && new_idepth > 0) {   // :696
      xr.x = kp[0] / id; xr.y = kp[1] / id; xr.z = kp[2] / id;                     // :707
      const TrackEdgeVal e1 = track_edge<false>(xr, img, ev);                      // :721
      if (e1.e > ev.cutoffTH * 10) sat = 1;                                        // :724-727
      else { m = 1; term = 1; }
    }
    mask[i] = m;
    xref[i] = m ? xr : make_float4(0, 0, 0, p.w);
  }
  const unsigned long long bt = __ballot(term), bs = __ballot(sat);
  if ((threadIdx.x & 63) == 0) {
    if (bt) atomicAdd(cnt, __popcll(bt));
    if (bs) atomicAdd(cnt + 1, __popcll(bs));
  }
}

// computeActiveErrors + linearizeOplus + constructQuadraticForm (g2o: H += J^T rho1 J, b -= rho1 J^T e with the Huber
// kernel of delta huberTH) over the edge set; one edge per lane and trip, 46 double sums per lane, wave butterfly,
// per-block partial written to part[block][46].  err / J (optional) are indexed like the pc arrays.
__global__ __launch_bounds__(256) void k_g2o_track_lin(int n, const uint8_t* __restrict__ mask, const float4* __restrict__ xref,
                                                       const float4* __restrict__ img, sdso_g2o_track_eval_t ev, double* __restrict__ part,
                                                       double* __restrict__ err, double* __restrict__ Jout) {
  double acc[kSysDoubles];
#pragma unroll
  for (int k = 0; k < kSysDoubles; k++) acc[k] = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const bool on = mask[i] != 0;
    TrackEdgeVal v;
    if (on) v = track_edge<true>(xref[i], img, ev);
    else { v.e = 0; for (int k = 0; k < 8; k++) v.J[k] = 0; }
    if (err) err[i] = v.e;
    if (Jout) { for (int k = 0; k < 8; k++) Jout[(size_t)i * 8 + k] = v.J[k]; }
    if (!on) continue;
    const double e2 = v.e * v.e, delta = ev.huberTH, dsqr = delta * delta;
    double rho0, rho1;
    if (e2 <= dsqr) { rho0 = e2; rho1 = 1.; }
    else { const double sqrte = sqrt(e2); rho0 = 2 * sqrte * delta - dsqr; rho1 = delta / sqrte; }
    int k = 0;
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
      for (int c = r; c < 8; c++) acc[k++] += v.J[r] * rho1 * v.J[c];
#pragma unroll
    for (int r = 0; r < 8; r++) acc[36 + r] -= rho1 * v.J[r] * v.e;
    acc[44] += e2;
    acc[45] += rho0;
  }
  __shared__ double sm[4][kSysDoubles];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < kSysDoubles; k++) {
    const double s = wave_sum(acc[k]);
    if (lane == 0) sm[wv][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < kSysDoubles) part[(size_t)blockIdx.x * kSysDoubles + threadIdx.x] = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
}

// ------------------------------------------------------------------------------------------------ LBA edge (B13)
struct LbaDev {
  int nf, nr, w, h;
  const float4* img[8];
  const float* pair_R; const float* pair_t; const float* pair_ab; const double* host_b0; const float* frameEnergyTH;
  double cam[4];
  const int* host; const int* target; const float* u; const float* v; const double* idepth; const float* color; const float* weights;
  double* error; double* J; uint8_t* state; float* energy; float* cpt; float* idepth_hessian; uint8_t* edge_level;
};

__constant__ int c_lba_pat[8][2] = {{0, -2}, {-1, -1}, {1, -1}, {-2, 0}, {0, 0}, {2, 0}, {-1, 1}, {0, 2}};

// EdgeLBASE3PosePhotoIdepthCamDSO::computeError + linearizeOplus (dso_g2o_edge.cpp:5-282): 8 lanes per residual, lane = pattern
// pixel; the running float sums (energy, wJI2_sum, H_idepth_idepth) are rebuilt in pattern order from the lane values.
__global__ __launch_bounds__(256) void k_g2o_lba_eval(LbaDev L) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int ri = gid >> 3, idx = gid & 7;
  const int lane = threadIdx.x & 63, base = lane & ~7;
  const bool live = ri < L.nr;
  const int rr = live ? ri : 0;
  double e = 0, X = 0, row[13];
  float wj = 0, Ku_f = 0, Kv_f = 0, nid_f = 0;
  bool f_neg = false, f_oob = false, f_nan = false;
#pragma unroll
  for (int k = 0; k < 13; k++) row[k] = 0;
  const int h = L.host[rr], tg = L.target[rr];
  const float* R = L.pair_R + (size_t)(h * L.nf + tg) * 9;
  const float* t = L.pair_t + (size_t)(h * L.nf + tg) * 3;
  const float ab0 = L.pair_ab[(size_t)(h * L.nf + tg) * 2], ab1 = L.pair_ab[(size_t)(h * L.nf + tg) * 2 + 1];
  const double fx = L.cam[0], fy = L.cam[1], cx = L.cam[2], cy = L.cam[3];
  const double idepth = L.idepth[rr];
  const float pu = L.u[rr], pv = L.v[rr];
  const float col = L.color[(size_t)rr * 8 + idx], wgt = L.weights[(size_t)rr * 8 + idx];
  const double u_host = pu + c_lba_pat[idx][0];
  const double v_host = pv + c_lba_pat[idx][1];
  const float Klip0 = (float)((u_host - cx) / fx), Klip1 = (float)((v_host - cy) / fy), Klip2 = 1;
  float ptp[3];
#pragma unroll
  for (int r = 0; r < 3; r++) ptp[r] = ((R[r * 3 + 0] * Klip0 + R[r * 3 + 1] * Klip1) + R[r * 3 + 2] * Klip2) + t[r] * (float)idepth;
  const double drescale = 1.0f / ptp[2];
  f_neg = drescale <= 0;
  const double new_idepth = idepth * drescale;
  const double _u = ptp[0] * drescale, _v = ptp[1] * drescale;
  const double _Ku = _u * fx + cx, _Kv = _v * fy + cy;
  f_oob = !f_neg && check_boundary(_Ku, _Kv, L.w - 3, L.h - 3);
  Ku_f = (float)_Ku; Kv_f = (float)_Kv; nid_f = (float)new_idepth;
  float3 hit = make_float3(0, 0, 0);
  if (!f_neg && !f_oob) {
    hit = interp33(L.img[tg], (float)_Ku, (float)_Kv, L.w);
    if (!isfinite(hit.x)) f_nan = true;
    else {
      e = hit.x - (ab0 * col + ab1);
      float w = sqrtf(kOutlierTHSumComponent / (kOutlierTHSumComponent + (hit.y * hit.y + hit.z * hit.z)));
      w = 0.5f * (w + wgt);
      const float hw = fabsf((float)e) < kHuberTH ? 1 : kHuberTH / fabsf((float)e);
      X = w * w * hw * e * e * (2 - hw);
      wj = hw * hw * (hit.y * hit.y + hit.z * hit.z);
      // linearizeOplus row
      const double fxi = 1 / fx, fyi = 1 / fy;
      const double p0 = hit.y, p1 = hit.z;
      double Cm[2][4];
      Cm[0][2] = drescale * (R[6] * _u - R[0]);
      Cm[0][3] = fx * fyi * drescale * (R[7] * _u - R[1]);
      Cm[0][0] = Klip0 * Cm[0][2];
      Cm[0][1] = Klip1 * Cm[0][3];
      Cm[1][2] = fy * fxi * drescale * (R[6] * _v - R[3]);
      Cm[1][3] = drescale * (R[7] * _v - R[4]);
      Cm[1][0] = Klip0 * Cm[1][2];
      Cm[1][1] = Klip1 * Cm[1][3];
#pragma unroll
      for (int c = 0; c < 4; c++) row[9 + c] = p0 * Cm[0][c] + p1 * Cm[1][c];
      const double dx = hit.y * fx, dy = hit.z * fy;
      row[0] = new_idepth * dx;
      row[1] = new_idepth * dy;
      row[2] = -new_idepth * (_u * dx + _v * dy);
      row[3] = -(_u * _v * dx + (1 + _v * _v) * dy);
      row[4] = _u * _v * dy + (1 + _u * _u) * dx;
      row[5] = _u * dy - _v * dx;
      row[6] = ab0 * (L.host_b0[h] - col);
      row[7] = -1;
      row[8] = dx * drescale * (t[0] - t[2] * _u) + dy * drescale * (t[1] - t[2] * _v);
    }
  }
  // group-wide control flow of the sequential loops
  const unsigned gm_early = (unsigned)((__ballot(f_neg || f_oob) >> base) & 0xffull);
  const unsigned gm_neg = (unsigned)((__ballot(f_neg) >> base) & 0xffull);
  const unsigned gm_nan = (unsigned)((__ballot(f_nan) >> base) & 0xffull);
  const float eTH = fmaxf(L.frameEnergyTH[h], L.frameEnergyTH[tg]);
  float energyLeft = 0, wJI2 = 0, Hii = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const double Xk = __shfl(X, base + k, 64);
    const float wk = __shfl(wj, base + k, 64);
    const double r8 = __shfl(row[8], base + k, 64);
    energyLeft += Xk;
    wJI2 += wk;
    Hii += r8 * r8;
  }
  const float c_Ku = __shfl(Ku_f, base + 4, 64), c_Kv = __shfl(Kv_f, base + 4, 64), c_id = __shfl(nid_f, base + 4, 64);
  if (!live) return;
  double* eo = L.error + (size_t)ri * 8;
  double* Jo = L.J + (size_t)ri * 104 + idx * 13;
  if (gm_early) {
    const int first = __ffs(gm_early) - 1;
    eo[idx] = 0;
#pragma unroll
    for (int k = 0; k < 13; k++) Jo[k] = 0;
    if (idx == 0) {
      L.state[ri] = 1;
      L.edge_level[ri] = ((gm_neg >> first) & 1u) ? 0 : 1;
      L.energy[ri * 2] = 0; L.energy[ri * 2 + 1] = 0;
      L.idepth_hessian[ri] = 0;
      const bool center = first > 4;
      L.cpt[ri * 3] = center ? c_Ku : 2.f; L.cpt[ri * 3 + 1] = center ? c_Kv : 2.f; L.cpt[ri * 3 + 2] = center ? c_id : 0.f;
    }
    return;
  }
  eo[idx] = e;
  int st;
  float eNew = energyLeft;
  if (energyLeft > eTH || wJI2 < 2) { eNew = eTH; st = 2; } else st = 0;
  const bool jearly = gm_nan != 0;   // linearizeOplus returns at the first non-finite sample, leaving the Jacobians untouched (zero here)
#pragma unroll
  for (int k = 0; k < 13; k++) Jo[k] = jearly ? 0. : row[k];
  if (idx == 0) {
    L.state[ri] = jearly ? 1 : (uint8_t)st;
    L.edge_level[ri] = 0;
    L.energy[ri * 2] = eNew; L.energy[ri * 2 + 1] = energyLeft;
    if (Hii < 1e-10) Hii = 1e-10;
    L.idepth_hessian[ri] = jearly ? 0.f : Hii;
    L.cpt[ri * 3] = c_Ku; L.cpt[ri * 3 + 1] = c_Kv; L.cpt[ri * 3 + 2] = c_id;
  }
}

static G2oEdgeSet* find_set(sdso_ctx* ctx, int ref_slot, int lvl) {
  if (!reg_has(g_g2o, ctx)) return nullptr;
  G2oState& st = reg_get(g_g2o, ctx);
  auto it = st.sets.find({ref_slot, lvl});
  return it == st.sets.end() ? nullptr : &it->second;
}

}  // namespace sdso

using namespace sdso;

extern "C" int sdso_g2o_track_add_edges(sdso_ctx* ctx, int ref_slot, int frame_slot, const sdso_g2o_track_eval_t* ev, double* res6, int* n_edges,
                                        uint8_t* edge_mask, float* Xref) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  SDSO_REQUIRE(ctx, ev && res6, "null argument");
  auto ir = ctx->refs.find(ref_slot);
  SDSO_REQUIRE(ctx, ir != ctx->refs.end(), "unknown reference slot");
  auto ip = ctx->pyr.find(frame_slot);
  SDSO_REQUIRE(ctx, ip != ctx->pyr.end(), "unknown frame slot");
  const int lvl = ev->lvl;
  SDSO_REQUIRE(ctx, lvl >= 0 && lvl < ip->second.levels, "level out of range");
  SDSO_REQUIRE(ctx, ev->w == ip->second.w[lvl] && ev->h == ip->second.h[lvl], "level size does not match the uploaded pyramid");
  const int n = ir->second.n[lvl];
  G2oState& st = reg_get(g_g2o, ctx);
  G2oEdgeSet& S = st.sets[{ref_slot, lvl}];
  if (S.cap < n) {
    SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (S.mask) hipFree(S.mask);
    if (S.xref) hipFree(S.xref);
    S.mask = nullptr; S.xref = nullptr;
    S.cap = n + n / 4 + 64;
    SDSO_HIP(ctx, hipMalloc(&S.mask, (size_t)S.cap));
    SDSO_HIP(ctx, hipMalloc(&S.xref, sizeof(float4) * (size_t)S.cap));
  }
  if (!st.d_cnt) SDSO_HIP(ctx, hipMalloc(&st.d_cnt, sizeof(int) * 2));
  const int nflow = (n + 31) / 32;
  if (st.flow_cap < nflow) {
    SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (st.d_flow) hipFree(st.d_flow);
    st.flow_cap = nflow + 64;
    SDSO_HIP(ctx, hipMalloc(&st.d_flow, sizeof(float) * 4 * (size_t)st.flow_cap));
  }
  S.n = n; S.n_edges = 0;
  int cnt[2] = {0, 0};
  std::vector<float> flow((size_t)nflow * 4, 0.f);
  if (n > 0) {
    SDSO_HIP(ctx, hipMemsetAsync(st.d_cnt, 0, sizeof(int) * 2, ctx->stream));
    {
      ProfScope ps(ctx, "k_g2o_track_edges");
      hipLaunchKernelGGL(k_g2o_track_edges, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, (const float4*)ir->second.pc[lvl],
                         (const float4*)ip->second.d[lvl], *ev, S.mask, S.xref, st.d_cnt, st.d_flow);
    }
    SDSO_HIP(ctx, hipGetLastError());
    SDSO_HIP(ctx, hipMemcpyAsync(cnt, st.d_cnt, sizeof(cnt), hipMemcpyDeviceToHost, ctx->stream));
    if (lvl == 0) SDSO_HIP(ctx, hipMemcpyAsync(flow.data(), st.d_flow, sizeof(float) * 4 * (size_t)nflow, hipMemcpyDeviceToHost, ctx->stream));
    if (edge_mask) SDSO_HIP(ctx, hipMemcpyAsync(edge_mask, S.mask, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    std::vector<float4> hx;
    if (Xref) { hx.resize(n); SDSO_HIP(ctx, hipMemcpyAsync(hx.data(), S.xref, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream)); }
    SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (Xref) for (int i = 0; i < n; i++) { Xref[i * 3] = hx[i].x; Xref[i * 3 + 1] = hx[i].y; Xref[i * 3 + 2] = hx[i].z; }
  }
  S.n_edges = cnt[0];
  float sT = 0, sRT = 0, sN = 0;
  if (lvl == 0)
    for (int k = 0; k < nflow; k++) { sT += flow[k * 4]; sT += flow[k * 4 + 1]; sRT += flow[k * 4 + 2]; sRT += flow[k * 4 + 3]; sN += 2; }
  res6[0] = 0;
  res6[1] = cnt[0];
  res6[2] = sT / (sN + 0.1);
  res6[3] = 0;
  res6[4] = sRT / (sN + 0.1);
  res6[5] = cnt[1] / (float)cnt[0];
  if (n_edges) *n_edges = cnt[0];
  return SDSO_OK;
}

extern "C" int sdso_g2o_track_linearize(sdso_ctx* ctx, int ref_slot, int frame_slot, const sdso_g2o_track_eval_t* ev, double* H, double* b, double* chi2,
                                        double* err, double* J) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  SDSO_REQUIRE(ctx, ev && H && b && chi2, "null argument");
  auto ip = ctx->pyr.find(frame_slot);
  SDSO_REQUIRE(ctx, ip != ctx->pyr.end(), "unknown frame slot");
  const int lvl = ev->lvl;
  SDSO_REQUIRE(ctx, lvl >= 0 && lvl < ip->second.levels, "level out of range");
  SDSO_REQUIRE(ctx, ev->w == ip->second.w[lvl] && ev->h == ip->second.h[lvl], "level size does not match the uploaded pyramid");
  G2oEdgeSet* S = find_set(ctx, ref_slot, lvl);
  SDSO_REQUIRE(ctx, S != nullptr, "no edge set for (reference, level): call sdso_g2o_track_add_edges first");
  for (int i = 0; i < 64; i++) H[i] = 0;
  for (int i = 0; i < 8; i++) b[i] = 0;
  chi2[0] = chi2[1] = 0;
  const int n = S->n;
  if (n == 0) return SDSO_OK;
  G2oState& st = reg_get(g_g2o, ctx);
  if (!st.d_part) { SDSO_HIP(ctx, hipMalloc(&st.d_part, sizeof(double) * kSysDoubles * kMaxLinBlocks)); st.part_blocks = kMaxLinBlocks; }
  const int blocks = std::min(kMaxLinBlocks, (n + 255) / 256);
  double *d_err = nullptr, *d_J = nullptr;
  if (err || J) {
    int rc = ensure_scratch(ctx, sizeof(double) * 9 * (size_t)n);
    if (rc) return rc;
    if (err) d_err = (double*)ctx->scratch;
    if (J) d_J = (double*)ctx->scratch + n;
  }
  {
    ProfScope ps(ctx, "k_g2o_track_lin");
    hipLaunchKernelGGL(k_g2o_track_lin, dim3(blocks), dim3(256), 0, ctx->stream, n, (const uint8_t*)S->mask, (const float4*)S->xref,
                       (const float4*)ip->second.d[lvl], *ev, st.d_part, d_err, d_J);
  }
  SDSO_HIP(ctx, hipGetLastError());
  std::vector<double> part((size_t)blocks * kSysDoubles);
  SDSO_HIP(ctx, hipMemcpyAsync(part.data(), st.d_part, sizeof(double) * part.size(), hipMemcpyDeviceToHost, ctx->stream));
  if (err) SDSO_HIP(ctx, hipMemcpyAsync(err, d_err, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  if (J) SDSO_HIP(ctx, hipMemcpyAsync(J, d_J, sizeof(double) * 8 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double sum[kSysDoubles] = {0};
  for (int bl = 0; bl < blocks; bl++)
    for (int k = 0; k < kSysDoubles; k++) sum[k] += part[(size_t)bl * kSysDoubles + k];
  int k = 0;
  for (int r = 0; r < 8; r++)
    for (int c = r; c < 8; c++) { H[r * 8 + c] = sum[k]; H[c * 8 + r] = sum[k]; k++; }
  for (int r = 0; r < 8; r++) b[r] = sum[36 + r];
  chi2[0] = sum[44];
  chi2[1] = sum[45];
  return SDSO_OK;
}

// CoarseTracker::trackNewestCoarse, fork-live (CoarseTracker.cpp:827-1069).  g2o's OptimizationAlgorithmLevenberg, its
// SparseOptimizerTerminateAction and the 8x8 solve are restated on the host (see the header of this file: unpinned).
extern "C" int sdso_g2o_track_newest_coarse(sdso_ctx* ctx, int ref_slot, int frame_slot, const sdso_track_params_t* prm, sdso_se3_t* lastToNew,
                                            sdso_aff_t* aff_g2l, sdso_track_result_t* out) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_REQUIRE(ctx, prm && lastToNew && aff_g2l && out, "null argument");
  SDSO_REQUIRE(ctx, prm->coarsestLvl >= 0 && prm->coarsestLvl < 5 && prm->coarsestLvl < prm->levels, "coarsestLvl out of range");
  for (int i = 0; i < 5; i++) { out->lastResiduals[i] = NAN; out->iterations[i] = 0; }
  for (int i = 0; i < 3; i++) out->lastFlowIndicators[i] = 1000;
  out->good = 0; out->evaluations = 0; out->point_evals = 0;
  auto ir = ctx->refs.find(ref_slot);
  SDSO_REQUIRE(ctx, ir != ctx->refs.end(), "unknown reference slot");
  Se3 pose;
  std::memcpy(pose.R.data(), lastToNew->R, 72);
  std::memcpy(pose.t.data(), lastToNew->t, 24);
  double aff[2] = {aff_g2l->a, aff_g2l->b};
  const sdso_se3_t refToNew_current = *lastToNew;
  const int maxIterations[5] = {2, 2, 2, 2, 2};   // :861
  size_t total_edges = 0;

  auto make_ev = [&](int lvl, const Se3& T, const double* a1b1) {
    sdso_g2o_track_eval_t ev;
    sdso_track_eval_t base;
    sdso_aff_t a = {a1b1[0], a1b1[1]};
    sdso_track_make_eval(prm, lvl, &refToNew_current, &a, 1.0f, &base);
    ev.lvl = lvl; ev.w = base.w; ev.h = base.h; ev.fx = base.fx; ev.fy = base.fy; ev.cx = base.cx; ev.cy = base.cy;
    std::memcpy(ev.Ki, base.Ki, sizeof(ev.Ki)); std::memcpy(ev.RKi, base.RKi, sizeof(ev.RKi)); std::memcpy(ev.t_cull, base.t, sizeof(ev.t_cull));
    std::memcpy(ev.R, T.R.data(), 72); std::memcpy(ev.t, T.t.data(), 24);
    ev.ab[0] = base.affLL[0]; ev.ab[1] = base.affLL[1];
    ev.b0 = prm->ref_aff_g2l.b;
    ev.cutoffTH = base.cutoffTH; ev.huberTH = base.huberTH;
    return ev;
  };

  for (int lvl = prm->coarsestLvl; lvl >= 0; lvl--) {
    const int n = ir->second.n[lvl];
    double resOld[6];
    int ne = 0;
    sdso_g2o_track_eval_t ev = make_ev(lvl, pose, aff);
    int rc = sdso_g2o_track_add_edges(ctx, ref_slot, frame_slot, &ev, resOld, &ne, nullptr, nullptr);
    if (rc) return rc;
    total_edges += ne;
    out->evaluations++; out->point_evals += n;
    double H[64], b[8], chi[2];
    auto linearize = [&](const Se3& T, const double* a1b1) -> int {
      sdso_g2o_track_eval_t e2 = make_ev(lvl, T, a1b1);
      out->evaluations++; out->point_evals += n;
      return sdso_g2o_track_linearize(ctx, ref_slot, frame_slot, &e2, H, b, chi, nullptr, nullptr);
    };
    double lambda = 0, ni = 2, lastChi = 0;
    bool stop = false, ok = true;
    for (int it = 0; it < maxIterations[lvl] && !stop && ok; it++) {
      if ((rc = linearize(pose, aff))) return rc;
      double currentChi = chi[1];
      if (it == 0) { lambda = 0.01; ni = 2; }   // setUserLambdaInit(0.01), :839-840
      double rho = 0;
      int qmax = 0;
      double Hs[64], bs[8];
      std::memcpy(Hs, H, sizeof(H)); std::memcpy(bs, b, sizeof(b));
      do {
        Dense A(8);
        std::vector<double> rhs(8), x(8, 0.0);
        for (int r = 0; r < 8; r++) { rhs[r] = bs[r]; for (int c = 0; c < 8; c++) A(r, c) = Hs[r * 8 + c]; A(r, r) += lambda; }
        const bool ok2 = solveLdlt(A, rhs, x);
        Se3 trial = pose;
        double affT[2] = {aff[0], aff[1]};
        if (ok2) {
          trial = expSe3(x.data()) * pose;   // VertexSE3PoseDSO::oplusImpl, dso_g2o_vertex.cpp:15-18
          affT[0] += x[6]; affT[1] += x[7];  // VertexPhotometricDSO::oplusImpl, :30-40
        }
        if ((rc = linearize(trial, affT))) return rc;
        const double tempChi = ok2 ? chi[1] : 1.7976931348623157e308;
        rho = currentChi - tempChi;
        double scale = 0;
        for (int k = 0; k < 8; k++) scale += x[k] * (lambda * x[k] + bs[k]);
        scale += 1e-3;
        rho /= scale;
        if (rho > 0 && std::isfinite(tempChi)) {
          double alpha = 1. - std::pow((2 * rho - 1), 3);
          alpha = std::min(alpha, 2. / 3.);
          lambda *= std::max(1. / 3., alpha);
          ni = 2;
          currentChi = tempChi;
          pose = trial; aff[0] = affT[0]; aff[1] = affT[1];
        } else {
          lambda *= ni;
          ni *= 2;
          if (!std::isfinite(lambda)) break;
        }
        qmax++;
      } while (rho < 0 && qmax < 10);
      out->iterations[lvl]++;
      if (qmax == 10 || rho == 0 || !std::isfinite(lambda)) ok = false;
      if ((rc = linearize(pose, aff))) return rc;   // terminate action: computeActiveErrors at the current estimate
      if (it == 0) lastChi = chi[1];
      else {
        const double gain = (lastChi - chi[1]) / chi[1];
        lastChi = chi[1];
        if (gain >= 0 && gain < 1e-3) stop = true;
      }
    }
    if (maxIterations[lvl] == 0 || n == 0) { if ((rc = linearize(pose, aff))) return rc; }
    out->lastResiduals[lvl] = sqrtf((float)chi[1] / total_edges);   // :1029
    for (int k = 0; k < 3; k++) out->lastFlowIndicators[k] = resOld[2 + k];
    if (out->lastResiduals[lvl] > 1.5 * prm->minResForAbort[lvl]) return SDSO_OK;   // :1032-1033
  }
  std::memcpy(lastToNew->R, pose.R.data(), 72);
  std::memcpy(lastToNew->t, pose.t.data(), 24);
  aff_g2l->a = aff[0]; aff_g2l->b = aff[1];
  if ((prm->affineOptModeA != 0 && (fabsf((float)aff_g2l->a) > 1.2)) || (prm->affineOptModeB != 0 && (fabsf((float)aff_g2l->b) > 200))) return SDSO_OK;
  double relAff[2];
  affFromTo(prm->ref_exposure, prm->new_exposure, prm->ref_aff_g2l.a, prm->ref_aff_g2l.b, aff_g2l->a, aff_g2l->b, relAff);
  if ((prm->affineOptModeA == 0 && (fabsf(logf((float)relAff[0])) > 1.5)) || (prm->affineOptModeB == 0 && (fabsf((float)relAff[1]) > 200))) return SDSO_OK;
  if (prm->affineOptModeA < 0) aff_g2l->a = 0;
  if (prm->affineOptModeB < 0) aff_g2l->b = 0;
  out->good = 1;
  return SDSO_OK;
}

extern "C" int sdso_g2o_lba_eval(sdso_ctx* ctx, const sdso_g2o_lba_t* Lh, double* error, double* J, uint8_t* state, float* energy,
                                 float* centerProjectedTo, float* idepth_hessian, uint8_t* edge_level) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  SDSO_REQUIRE(ctx, Lh && error && J && state && energy && centerProjectedTo && idepth_hessian && edge_level, "null argument");
  const int nf = Lh->nf, nr = Lh->nr;
  SDSO_REQUIRE(ctx, nf >= 1 && nf <= 8 && nr >= 0, "nf must be 1..8");
  if (nr == 0) return SDSO_OK;
  SDSO_REQUIRE(ctx, Lh->frame_slot && Lh->pair_R && Lh->pair_t && Lh->pair_ab && Lh->host_b0 && Lh->frameEnergyTH && Lh->host && Lh->target && Lh->u &&
                        Lh->v && Lh->idepth && Lh->color && Lh->weights, "null window array");
  LbaDev D;
  std::memset(&D, 0, sizeof(D));
  D.nf = nf; D.nr = nr; D.w = Lh->w; D.h = Lh->h;
  for (int f = 0; f < nf; f++) {
    auto ip = ctx->pyr.find(Lh->frame_slot[f]);
    SDSO_REQUIRE(ctx, ip != ctx->pyr.end(), "unknown frame slot");
    SDSO_REQUIRE(ctx, ip->second.w[0] == Lh->w && ip->second.h[0] == Lh->h, "frame size mismatch");
    D.img[f] = ip->second.d[0];
  }
  for (int r = 0; r < nr; r++) {
    SDSO_REQUIRE(ctx, Lh->host[r] >= 0 && Lh->host[r] < nf && Lh->target[r] >= 0 && Lh->target[r] < nf, "host / target out of range");
    // the reference samples unchecked inside the boundary test; the host pixel itself must be a valid pattern centre
    SDSO_REQUIRE(ctx, std::isfinite(Lh->u[r]) && std::isfinite(Lh->v[r]) && std::isfinite(Lh->idepth[r]), "non-finite point");
  }
  for (int k = 0; k < 4; k++) D.cam[k] = Lh->cam[k];
  // one device blob: inputs then outputs
  const size_t nn = (size_t)nf * nf;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t o_R = take(sizeof(float) * nn * 9), o_t = take(sizeof(float) * nn * 3), o_ab = take(sizeof(float) * nn * 2), o_b0 = take(sizeof(double) * nf),
               o_eth = take(sizeof(float) * nf), o_host = take(sizeof(int) * nr), o_tg = take(sizeof(int) * nr), o_u = take(sizeof(float) * nr),
               o_v = take(sizeof(float) * nr), o_id = take(sizeof(double) * nr), o_col = take(sizeof(float) * 8 * nr), o_w = take(sizeof(float) * 8 * nr),
               o_err = take(sizeof(double) * 8 * nr), o_J = take(sizeof(double) * 104 * (size_t)nr), o_st = take(nr), o_en = take(sizeof(float) * 2 * nr),
               o_cpt = take(sizeof(float) * 3 * nr), o_ih = take(sizeof(float) * nr), o_lv = take(nr);
  int rc = ensure_scratch(ctx, off);
  if (rc) return rc;
  char* base = (char*)ctx->scratch;
#define UPL(o, src, bytes) SDSO_HIP(ctx, hipMemcpyAsync(base + (o), (src), (bytes), hipMemcpyHostToDevice, ctx->stream))
  UPL(o_R, Lh->pair_R, sizeof(float) * nn * 9); UPL(o_t, Lh->pair_t, sizeof(float) * nn * 3); UPL(o_ab, Lh->pair_ab, sizeof(float) * nn * 2);
  UPL(o_b0, Lh->host_b0, sizeof(double) * nf); UPL(o_eth, Lh->frameEnergyTH, sizeof(float) * nf);
  UPL(o_host, Lh->host, sizeof(int) * nr); UPL(o_tg, Lh->target, sizeof(int) * nr); UPL(o_u, Lh->u, sizeof(float) * nr); UPL(o_v, Lh->v, sizeof(float) * nr);
  UPL(o_id, Lh->idepth, sizeof(double) * nr); UPL(o_col, Lh->color, sizeof(float) * 8 * nr); UPL(o_w, Lh->weights, sizeof(float) * 8 * nr);
#undef UPL
  D.pair_R = (const float*)(base + o_R); D.pair_t = (const float*)(base + o_t); D.pair_ab = (const float*)(base + o_ab);
  D.host_b0 = (const double*)(base + o_b0); D.frameEnergyTH = (const float*)(base + o_eth);
  D.host = (const int*)(base + o_host); D.target = (const int*)(base + o_tg); D.u = (const float*)(base + o_u); D.v = (const float*)(base + o_v);
  D.idepth = (const double*)(base + o_id); D.color = (const float*)(base + o_col); D.weights = (const float*)(base + o_w);
  D.error = (double*)(base + o_err); D.J = (double*)(base + o_J); D.state = (uint8_t*)(base + o_st); D.energy = (float*)(base + o_en);
  D.cpt = (float*)(base + o_cpt); D.idepth_hessian = (float*)(base + o_ih); D.edge_level = (uint8_t*)(base + o_lv);
  {
    ProfScope ps(ctx, "k_g2o_lba_eval");
    hipLaunchKernelGGL(k_g2o_lba_eval, dim3(((size_t)nr * 8 + 255) / 256), dim3(256), 0, ctx->stream, D);
  }
  SDSO_HIP(ctx, hipGetLastError());
#define DNL(dst, o, bytes) SDSO_HIP(ctx, hipMemcpyAsync((dst), base + (o), (bytes), hipMemcpyDeviceToHost, ctx->stream))
  DNL(error, o_err, sizeof(double) * 8 * nr); DNL(J, o_J, sizeof(double) * 104 * (size_t)nr); DNL(state, o_st, nr); DNL(energy, o_en, sizeof(float) * 2 * nr);
  DNL(centerProjectedTo, o_cpt, sizeof(float) * 3 * nr); DNL(idepth_hessian, o_ih, sizeof(float) * nr); DNL(edge_level, o_lv, nr);
#undef DNL
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SDSO_OK;
}
