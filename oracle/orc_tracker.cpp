// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_api.h).
// Coarse tracker, DSO-native arithmetic.  Follows (paths under /root/reference):
//   src/FullSystem/CoarseTracker.cpp:600-792  calcRes (native body kept as comments :699-775)
//   src/FullSystem/CoarseTracker.cpp:537-596  calcGSSSE
//   src/FullSystem/CoarseTracker.cpp:827-1069 trackNewestCoarse (native LM kept as comments :861-1047)
//   src/util/globalFuncs.h:73-86              getInterpolatedElement33
//   src/util/NumType.h:159-170                AffLight::fromToVecExposure
#include "orc_api.h"
#include "orc_math.h"
#include "orc_acc.h"
#include "orc_common.h"
#include <cmath>
#include <vector>

using namespace orc;

namespace {

struct WarpBuf {
  std::vector<float> idepth, u, v, dx, dy, residual, weight, refColor;
  int n = 0;
  void reserve(int cap) {
    idepth.assign(cap, 0); u.assign(cap, 0); v.assign(cap, 0); dx.assign(cap, 0); dy.assign(cap, 0);
    residual.assign(cap, 0); weight.assign(cap, 0); refColor.assign(cap, 0);
    n = 0;
  }
};

// CoarseTracker.cpp:600-792
void calcRes(int nl, const float* lpc_u, const float* lpc_v, const float* lpc_idepth,
             const float* lpc_color, const float* dINewl, const orc_track_eval_t& ev, double* rs,
             WarpBuf& wb, uint8_t* mask) {
  float E = 0;
  int numTermsInE = 0;
  int numTermsInWarped = 0;
  int numSaturated = 0;
  const int lvl = ev.lvl, wl = ev.w, hl = ev.h;
  const float fxl = ev.fx, fyl = ev.fy, cxl = ev.cx, cyl = ev.cy;
  const float* RKi = ev.RKi;
  const float* Ki = ev.Ki;
  const float* t = ev.t;
  const float affLL0 = ev.affLL[0], affLL1 = ev.affLL[1];
  float sumSquaredShiftT = 0, sumSquaredShiftRT = 0, sumSquaredShiftNum = 0;
  const float setting_huberTH = ev.huberTH;
  const float cutoffTH = ev.cutoffTH;
  float maxEnergy = 2 * setting_huberTH * cutoffTH - setting_huberTH * setting_huberTH;

  wb.reserve(nl + 4);
  for (int i = 0; i < nl; i++) {
    if (mask) mask[i] = 0;
    float id = lpc_idepth[i];
    float x = lpc_u[i];
    float y = lpc_v[i];
    float pt[3];
    for (int r = 0; r < 3; r++) pt[r] = ((RKi[r * 3 + 0] * x + RKi[r * 3 + 1] * y) + RKi[r * 3 + 2] * 1.0f) + t[r] * id;
    float u = pt[0] / pt[2];
    float v = pt[1] / pt[2];
    float Ku = fxl * u + cxl;
    float Kv = fyl * v + cyl;
    float new_idepth = id / pt[2];

    if (lvl == 0 && i % 32 == 0) {  // :662-693
      float ptT[3], ptT2[3], pt3[3];
      for (int r = 0; r < 3; r++) {
        float kp = (Ki[r * 3 + 0] * x + Ki[r * 3 + 1] * y) + Ki[r * 3 + 2] * 1.0f;
        float rp = (RKi[r * 3 + 0] * x + RKi[r * 3 + 1] * y) + RKi[r * 3 + 2] * 1.0f;
        ptT[r] = kp + t[r] * id;
        ptT2[r] = kp - t[r] * id;
        pt3[r] = rp - t[r] * id;
      }
      float uT = ptT[0] / ptT[2], vT = ptT[1] / ptT[2];
      float KuT = fxl * uT + cxl, KvT = fyl * vT + cyl;
      float uT2 = ptT2[0] / ptT2[2], vT2 = ptT2[1] / ptT2[2];
      float KuT2 = fxl * uT2 + cxl, KvT2 = fyl * vT2 + cyl;
      float u3 = pt3[0] / pt3[2], v3 = pt3[1] / pt3[2];
      float Ku3 = fxl * u3 + cxl, Kv3 = fyl * v3 + cyl;
      sumSquaredShiftT += (KuT - x) * (KuT - x) + (KvT - y) * (KvT - y);
      sumSquaredShiftT += (KuT2 - x) * (KuT2 - x) + (KvT2 - y) * (KvT2 - y);
      sumSquaredShiftRT += (Ku - x) * (Ku - x) + (Kv - y) * (Kv - y);
      sumSquaredShiftRT += (Ku3 - x) * (Ku3 - x) + (Kv3 - y) * (Kv3 - y);
      sumSquaredShiftNum += 2;
    }

    if (!(Ku > 2 && Kv > 2 && Ku < wl - 3 && Kv < hl - 3 && new_idepth > 0)) continue;  // :696

    float refColor = lpc_color[i];
    float hitColor[3];
    interp33(dINewl, Ku, Kv, wl, hitColor);
    if (!std::isfinite(hitColor[0])) continue;
    float residual = hitColor[0] - (float)(affLL0 * refColor + affLL1);
    float hw = fabsf(residual) < setting_huberTH ? 1 : setting_huberTH / fabsf(residual);

    if (fabsf(residual) > cutoffTH) {
      E += maxEnergy;
      numTermsInE++;
      numSaturated++;
    } else {
      E += hw * residual * residual * (2 - hw);
      numTermsInE++;
      wb.idepth[numTermsInWarped] = new_idepth;
      wb.u[numTermsInWarped] = u;
      wb.v[numTermsInWarped] = v;
      wb.dx[numTermsInWarped] = hitColor[1];
      wb.dy[numTermsInWarped] = hitColor[2];
      wb.residual[numTermsInWarped] = residual;
      wb.weight[numTermsInWarped] = hw;
      wb.refColor[numTermsInWarped] = lpc_color[i];
      numTermsInWarped++;
      if (mask) mask[i] = 1;
    }
  }
  while (numTermsInWarped % 4 != 0) {  // :763-773 zero padding
    wb.idepth[numTermsInWarped] = 0; wb.u[numTermsInWarped] = 0; wb.v[numTermsInWarped] = 0;
    wb.dx[numTermsInWarped] = 0; wb.dy[numTermsInWarped] = 0; wb.residual[numTermsInWarped] = 0;
    wb.weight[numTermsInWarped] = 0; wb.refColor[numTermsInWarped] = 0;
    numTermsInWarped++;
  }
  wb.n = numTermsInWarped;

  rs[0] = E;
  rs[1] = numTermsInE;
  rs[2] = sumSquaredShiftT / (sumSquaredShiftNum + 0.1);
  rs[3] = 0;
  rs[4] = sumSquaredShiftRT / (sumSquaredShiftNum + 0.1);
  rs[5] = numSaturated / (float)numTermsInE;
}

// CoarseTracker.cpp:537-596
void calcGSSSE(const WarpBuf& wb, const orc_track_eval_t& ev, double* H_out, double* b_out) {
  Accumulator9 acc;
  acc.initialize();
  const float fxl = ev.fx, fyl = ev.fy, b0 = ev.ref_b0, a = ev.affLL[0];
  int n = wb.n;
  for (int i = 0; i < n; i += 4) {
    float J[9][4], w[4];
    for (int l = 0; l < 4; l++) {
      float dx = wb.dx[i + l] * fxl;
      float dy = wb.dy[i + l] * fyl;
      float u = wb.u[i + l], v = wb.v[i + l], id = wb.idepth[i + l];
      J[0][l] = id * dx;
      J[1][l] = id * dy;
      J[2][l] = 0.0f - id * (u * dx + v * dy);
      J[3][l] = 0.0f - ((u * v) * dx + dy * (1.0f + v * v));
      J[4][l] = (u * v) * dy + dx * (1.0f + u * u);
      J[5][l] = u * dy - v * dx;
      J[6][l] = a * (b0 - wb.refColor[i + l]);
      J[7][l] = -1.0f;
      J[8][l] = wb.residual[i + l];
      w[l] = wb.weight[i + l];
    }
    acc.updateSSE_weighted(J, w);
  }
  acc.finish();
  float inv_n = 1.0f / n;
  double H[8][8], b[8];
  for (int r = 0; r < 8; r++) {
    for (int c = 0; c < 8; c++) H[r][c] = (double)acc.H[r][c] * inv_n;
    b[r] = (double)acc.H[r][8] * inv_n;
  }
  const double SC[8] = {SCALE_XI_ROT, SCALE_XI_ROT, SCALE_XI_ROT, SCALE_XI_TRANS, SCALE_XI_TRANS,
                        SCALE_XI_TRANS, SCALE_A, SCALE_B};  // :584-595 (ROT/TRANS order as in the reference)
  for (int r = 0; r < 8; r++)
    for (int c = 0; c < 8; c++) H[r][c] *= SC[c];
  for (int r = 0; r < 8; r++)
    for (int c = 0; c < 8; c++) H[r][c] *= SC[r];
  for (int r = 0; r < 8; r++) b[r] *= SC[r];
  for (int r = 0; r < 8; r++) {
    for (int c = 0; c < 8; c++) H_out[r * 8 + c] = H[r][c];
    b_out[r] = b[r];
  }
}

void make_eval(const orc_track_params_t& p, int lvl, const SE3& refToNew, const orc_aff_t& aff,
               float cutoff, orc_track_eval_t& ev) {
  ev.lvl = lvl; ev.w = p.w[lvl]; ev.h = p.h[lvl];
  ev.fx = p.fx[lvl]; ev.fy = p.fy[lvl]; ev.cx = p.cx[lvl]; ev.cy = p.cy[lvl];
  float K[9] = {ev.fx, 0, ev.cx, 0, ev.fy, ev.cy, 0, 0, 1};
  mat3_inv<float>(K, ev.Ki);
  float Rf[9];
  for (int i = 0; i < 9; i++) Rf[i] = (float)refToNew.R[i];
  mat3_mul<float>(Rf, ev.Ki, ev.RKi);
  for (int i = 0; i < 3; i++) ev.t[i] = (float)refToNew.t[i];
  double affd[2];
  fromToVecExposure(p.ref_exposure, p.new_exposure, p.ref_aff_g2l.a, p.ref_aff_g2l.b, aff.a, aff.b, affd);
  ev.affLL[0] = (float)affd[0]; ev.affLL[1] = (float)affd[1];
  ev.ref_b0 = (float)p.ref_aff_g2l.b;
  ev.cutoffTH = cutoff;
  ev.huberTH = p.huberTH;
}

}  // namespace

extern "C" int orc_track_calc_res_gs(int n, const float* pc_u, const float* pc_v,
                                     const float* pc_idepth, const float* pc_color, const float* dI,
                                     const orc_track_eval_t* ev, double* H, double* b, double* res,
                                     int* n_warped, uint8_t* inlier_mask, float* buf_warped, int n_cap) {
  WarpBuf wb;
  calcRes(n, pc_u, pc_v, pc_idepth, pc_color, dI, *ev, res, wb, inlier_mask);
  if (n_warped) *n_warped = wb.n;
  if (H && b) {
    if (wb.n > 0) calcGSSSE(wb, *ev, H, b);
    else { for (int i = 0; i < 64; i++) H[i] = 0; for (int i = 0; i < 8; i++) b[i] = 0; }
  }
  if (buf_warped) {
    const std::vector<float>* arrs[8] = {&wb.idepth, &wb.u, &wb.v, &wb.dx, &wb.dy, &wb.residual, &wb.weight, &wb.refColor};
    for (int a = 0; a < 8; a++)
      for (int i = 0; i < wb.n && i < n_cap; i++) buf_warped[(size_t)a * n_cap + i] = (*arrs[a])[i];
  }
  return 0;
}

extern "C" void orc_track_make_eval(const orc_track_params_t* prm, int lvl, const orc_se3_t* refToNew,
                                    const orc_aff_t* aff_g2l, float levelCutoffRepeat, orc_track_eval_t* ev) {
  SE3 T;
  std::memcpy(T.R, refToNew->R, sizeof(double) * 9);
  std::memcpy(T.t, refToNew->t, sizeof(double) * 3);
  make_eval(*prm, lvl, T, *aff_g2l, prm->coarseCutoffTH * levelCutoffRepeat, *ev);
}

// Test instrumentation (not in the reference): the smallest relative margin of the decisions the LM loop took on every level of the LAST
// call — accept (`resNew[0]/resNew[1] < resOld[0]/resOld[1]`, :1004), stop (`inc.norm() > 1e-3`, :1022) and the cut-off repeat
// (`resOld[5] > 0.6`, :897).  A path whose float sums differ in the last bits can only take another number of iterations where one of these
// margins is of the size of that difference (tests/test_tracker_gpu.py::test_cluster_sizes_reproduce_the_iteration_counts).
static double g_track_margin[5] = {1e300, 1e300, 1e300, 1e300, 1e300};
extern "C" void orc_track_last_margins(double* out5) { for (int i = 0; i < 5; i++) out5[i] = g_track_margin[i]; }
static void track_margin(int lvl, double a, double b) {
  const double m = std::fabs(a - b) / std::max(std::max(std::fabs(a), std::fabs(b)), 1e-300);
  if (m < g_track_margin[lvl]) g_track_margin[lvl] = m;
}

// CoarseTracker.cpp:827-1069 with the DSO-native LM of the commented block.
extern "C" int orc_track_newest_coarse(const int* pc_n, const float* const* pc_u,
                                       const float* const* pc_v, const float* const* pc_idepth,
                                       const float* const* pc_color, const float* const* dIp,
                                       const orc_track_params_t* prm, orc_se3_t* lastToNew,
                                       orc_aff_t* aff_g2l, orc_track_result_t* out) {
  const orc_track_params_t& p = *prm;
  for (int i = 0; i < 5; i++) { out->lastResiduals[i] = NAN; out->iterations[i] = 0; }
  for (int i = 0; i < 3; i++) out->lastFlowIndicators[i] = 1000;
  out->evaluations = 0; out->point_evals = 0; out->good = 0;
  const float lambdaExtrapolationLimit = 0.001f;
  for (int i = 0; i < 5; i++) g_track_margin[i] = 1e300;

  SE3 refToNew_current;
  std::memcpy(refToNew_current.R, lastToNew->R, sizeof(double) * 9);
  std::memcpy(refToNew_current.t, lastToNew->t, sizeof(double) * 3);
  orc_aff_t aff_g2l_current = *aff_g2l;
  bool haveRepeated = false;

  for (int lvl = p.coarsestLvl; lvl >= 0; lvl--) {
    double H[64], b[8];
    float levelCutoffRepeat = 1;
    orc_track_eval_t ev;
    WarpBuf wb;
    double resOld[6];
    auto eval = [&](const SE3& T, const orc_aff_t& a, double* rs, WarpBuf& w) {
      make_eval(p, lvl, T, a, p.coarseCutoffTH * levelCutoffRepeat, ev);
      calcRes(pc_n[lvl], pc_u[lvl], pc_v[lvl], pc_idepth[lvl], pc_color[lvl], dIp[lvl], ev, rs, w, nullptr);
      out->evaluations++; out->point_evals += pc_n[lvl];
    };
    eval(refToNew_current, aff_g2l_current, resOld, wb);
    track_margin(lvl, resOld[5], 0.6);
    while (resOld[5] > 0.6 && levelCutoffRepeat < 50) {  // :897-904
      levelCutoffRepeat *= 2;
      eval(refToNew_current, aff_g2l_current, resOld, wb);
      track_margin(lvl, resOld[5], 0.6);
    }
    calcGSSSE(wb, ev, H, b);
    float lambda = 0.01;

    for (int iteration = 0; iteration < p.maxIterations[lvl]; iteration++) {
      out->iterations[lvl]++;
      MatX Hl(8, 8);
      for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) Hl(i, j) = H[i * 8 + j];
      for (int i = 0; i < 8; i++) Hl(i, i) *= (1 + lambda);
      VecX nb(8), inc;
      for (int i = 0; i < 8; i++) nb[i] = -b[i];
      ldlt_solve(Hl, nb, inc);
      auto solve_sub = [&](const MatX& Hs, const VecX& bs, int m, VecX& xs) {
        MatX Hm(m, m); VecX bm(m);
        for (int i = 0; i < m; i++) { bm[i] = -bs[i]; for (int j = 0; j < m; j++) Hm(i, j) = Hs(i, j); }
        ldlt_solve(Hm, bm, xs);
      };
      VecX bv(b, b + 8);
      if (p.affineOptModeA < 0 && p.affineOptModeB < 0) {  // :937-940
        VecX x6; solve_sub(Hl, bv, 6, x6);
        for (int i = 0; i < 6; i++) inc[i] = x6[i];
        inc[6] = inc[7] = 0;
      }
      if (!(p.affineOptModeA < 0) && p.affineOptModeB < 0) {  // :943-946
        VecX x7; solve_sub(Hl, bv, 7, x7);
        for (int i = 0; i < 7; i++) inc[i] = x7[i];
        inc[7] = 0;
      }
      if (p.affineOptModeA < 0 && !(p.affineOptModeB < 0)) {  // :949-964
        MatX HlStitch = Hl; VecX bStitch = bv;
        for (int i = 0; i < 8; i++) HlStitch(i, 6) = HlStitch(i, 7);
        for (int j = 0; j < 8; j++) HlStitch(6, j) = HlStitch(7, j);
        bStitch[6] = bStitch[7];
        VecX x7; solve_sub(HlStitch, bStitch, 7, x7);
        for (int i = 0; i < 8; i++) inc[i] = 0;
        for (int i = 0; i < 6; i++) inc[i] = x7[i];
        inc[6] = 0; inc[7] = x7[6];
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
      double s = 0; for (int i = 0; i < 8; i++) s += incScaled[i];
      if (!std::isfinite(s)) for (int i = 0; i < 8; i++) incScaled[i] = 0;

      SE3 refToNew_new = se3_mul(se3_exp(incScaled), refToNew_current);
      orc_aff_t aff_g2l_new = aff_g2l_current;
      aff_g2l_new.a += incScaled[6];
      aff_g2l_new.b += incScaled[7];

      double resNew[6];
      WarpBuf wbNew;
      orc_track_eval_t evOld = ev;
      eval(refToNew_new, aff_g2l_new, resNew, wbNew);
      bool accept = (resNew[0] / resNew[1]) < (resOld[0] / resOld[1]);
      track_margin(lvl, resNew[0] / resNew[1], resOld[0] / resOld[1]);
      if (accept) {
        calcGSSSE(wbNew, ev, H, b);
        for (int i = 0; i < 6; i++) resOld[i] = resNew[i];
        aff_g2l_current = aff_g2l_new;
        refToNew_current = refToNew_new;
        lambda *= 0.5;
      } else {
        ev = evOld;
        lambda *= 4;
        if (lambda < lambdaExtrapolationLimit) lambda = lambdaExtrapolationLimit;
      }
      double nrm = 0; for (int i = 0; i < 8; i++) nrm += inc[i] * inc[i];
      nrm = std::sqrt(nrm);
      track_margin(lvl, nrm, 1e-3);
      if (!(nrm > 1e-3)) break;
    }

    out->lastResiduals[lvl] = sqrtf((float)(resOld[0] / resOld[1]));
    out->lastFlowIndicators[0] = resOld[2]; out->lastFlowIndicators[1] = resOld[3]; out->lastFlowIndicators[2] = resOld[4];
    if (out->lastResiduals[lvl] > 1.5 * p.minResForAbort[lvl]) return 0;  // :1032
    if (levelCutoffRepeat > 1 && !haveRepeated) { lvl++; haveRepeated = true; }
  }

  std::memcpy(lastToNew->R, refToNew_current.R, sizeof(double) * 9);
  std::memcpy(lastToNew->t, refToNew_current.t, sizeof(double) * 3);
  *aff_g2l = aff_g2l_current;

  // :1050-1066
  if ((p.affineOptModeA != 0 && (fabsf((float)aff_g2l->a) > 1.2)) ||
      (p.affineOptModeB != 0 && (fabsf((float)aff_g2l->b) > 200)))
    return 0;
  double rel[2];
  fromToVecExposure(p.ref_exposure, p.new_exposure, p.ref_aff_g2l.a, p.ref_aff_g2l.b, aff_g2l->a, aff_g2l->b, rel);
  float relAff0 = (float)rel[0], relAff1 = (float)rel[1];
  if ((p.affineOptModeA == 0 && (fabsf(logf(relAff0)) > 1.5)) || (p.affineOptModeB == 0 && (fabsf(relAff1) > 200)))
    return 0;
  if (p.affineOptModeA < 0) aff_g2l->a = 0;
  if (p.affineOptModeB < 0) aff_g2l->b = 0;
  out->good = 1;
  return 0;
}

// ---------------------------------------------------------------- CoarseTracker::makeCoarseDepthL0, STEP1's splat .. STEP5
// src/FullSystem/CoarseTracker.cpp:352-534.  Inputs: what STEP1 computed per active point — the integer pixel (u, v) on lastRef
// (:307-308), the (stereo-refined) new_idepth and weight = sqrtf(1e-3 / (HdiF + 1e-12)) (:350); dIp = lastRef->dIp (AoS {I, dx, dy}).
// Outputs: pc_n[lvl] and pc_u / pc_v / pc_idepth / pc_color[lvl] (caller-allocated, w[lvl] * h[lvl] entries each) in the reference's
// scan order.  The dilation reads idepthl in place: only entries with weightSumsl_bak > 0 are read and only entries with
// weightSumsl_bak <= 0 are written, so no copy is needed (the reference says so at :404-406).  The one element past the map that
// :410 / :459 can touch (i + 1 + wl, i + wl at the last scanned index) is treated as "no neighbour" (reading it is undefined in the
// reference; the product makes the same choice).
extern "C" int orc_make_coarse_depth(int levels, const int* w, const int* h, const float* const* dIp, int n, const int* u, const int* v,
                                     const float* new_idepth, const float* weight, int* pc_n, float* const* pc_u, float* const* pc_v,
                                     float* const* pc_idepth, float* const* pc_color) {
  std::vector<std::vector<float>> idepth(levels), weightSums(levels), weightSums_bak(levels);
  for (int l = 0; l < levels; l++) { idepth[l].assign((size_t)w[l] * h[l], 0.f); weightSums[l].assign((size_t)w[l] * h[l], 0.f); weightSums_bak[l].assign((size_t)w[l] * h[l], 0.f); }
  for (int i = 0; i < n; i++) {                                       // :352-354
    idepth[0][u[i] + w[0] * v[i]] += new_idepth[i] * weight[i];
    weightSums[0][u[i] + w[0] * v[i]] += weight[i];
  }
  for (int lvl = 1; lvl < levels; lvl++) {                             // STEP2 :360-386
    const int lvlm1 = lvl - 1, wl = w[lvl], hl = h[lvl], wlm1 = w[lvlm1];
    float* idepth_l = idepth[lvl].data(); float* weightSums_l = weightSums[lvl].data();
    const float* idepth_lm = idepth[lvlm1].data(); const float* weightSums_lm = weightSums[lvlm1].data();
    for (int y = 0; y < hl; y++)
      for (int x = 0; x < wl; x++) {
        const int bidx = 2 * x + 2 * y * wlm1;
        idepth_l[x + y * wl] = idepth_lm[bidx] + idepth_lm[bidx + 1] + idepth_lm[bidx + wlm1] + idepth_lm[bidx + wlm1 + 1];
        weightSums_l[x + y * wl] = weightSums_lm[bidx] + weightSums_lm[bidx + 1] + weightSums_lm[bidx + wlm1] + weightSums_lm[bidx + wlm1 + 1];
      }
  }
  auto dilate = [&](int lvl, const int offs[4]) {                      // STEP3 :390-441 (diagonal neighbours), STEP4 :445-488 (axis neighbours)
    const int wh = w[lvl] * h[lvl] - w[lvl], total = w[lvl] * h[lvl];
    float* weightSumsl = weightSums[lvl].data(); float* weightSumsl_bak = weightSums_bak[lvl].data();
    std::memcpy(weightSumsl_bak, weightSumsl, (size_t)total * sizeof(float));
    float* idepthl = idepth[lvl].data();
    for (int i = w[lvl]; i < wh; i++) {
      if (weightSumsl_bak[i] <= 0) {
        float sum = 0, num = 0, numn = 0;
        for (int k = 0; k < 4; k++) {
          const int j = i + offs[k];
          if (j >= 0 && j < total && weightSumsl_bak[j] > 0) { sum += idepthl[j]; num += weightSumsl_bak[j]; numn++; }
        }
        if (numn > 0) { idepthl[i] = sum / numn; weightSumsl[i] = num / numn; }
      }
    }
  };
  for (int lvl = 0; lvl < 2 && lvl < levels; lvl++) { const int wl = w[lvl]; const int offs[4] = {1 + wl, -1 - wl, wl - 1, -wl + 1}; dilate(lvl, offs); }
  for (int lvl = 2; lvl < levels; lvl++) { const int wl = w[lvl]; const int offs[4] = {1, -1, wl, -wl}; dilate(lvl, offs); }
  for (int lvl = 0; lvl < levels; lvl++) {                             // STEP5 :492-533
    float* weightSumsl = weightSums[lvl].data(); float* idepthl = idepth[lvl].data();
    const float* dIRefl = dIp[lvl];
    const int wl = w[lvl], hl = h[lvl];
    int lpc_n = 0;
    for (int y = 2; y < hl - 2; y++)
      for (int x = 2; x < wl - 2; x++) {
        const int i = x + y * wl;
        if (weightSumsl[i] > 0) {
          idepthl[i] /= weightSumsl[i];
          pc_u[lvl][lpc_n] = x; pc_v[lvl][lpc_n] = y; pc_idepth[lvl][lpc_n] = idepthl[i]; pc_color[lvl][lpc_n] = dIRefl[(size_t)i * 3];
          if (!std::isfinite(pc_color[lvl][lpc_n]) || !(idepthl[i] > 0)) { idepthl[i] = -1; continue; }
          lpc_n++;
        } else idepthl[i] = -1;
        weightSumsl[i] = 1;
      }
    pc_n[lvl] = lpc_n;
  }
  return 0;
}
