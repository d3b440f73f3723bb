// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_api.h).  parity unpinned.
//
// The fork's LIVE factors (rows T5, B13, S3 of SURVEY.md §8a): the g2o edges of src/FullSystem/dso_g2o_edge.cpp and the
// call sites that build them.  The per-edge arithmetic (computeError / linearizeOplus) is fully specified by the
// reference's own sources and is restated here line by line:
//   EdgeSE3PosePhotoDSO                  dso_g2o_edge.cpp:395-500, dso_util.hpp:10-46, graph build CoarseTracker.cpp:600-792
//   EdgeLBASE3PosePhotoIdepthCamDSO      dso_g2o_edge.cpp:5-282,   graph build FullSystemOptimize.cpp:455-542
//   EdgeTracePointUVDSO / VertexUVDSO    dso_g2o_edge.cpp:571-619, dso_g2o_vertex.cpp:65-88, loop ImmaturePoint.cpp:309-412
// What runs AROUND the edges lives in g2o, which is neither vendored nor version-pinned (CMakeLists.txt:47-58,
// `find_package(g2o REQUIRED)`; the API used — g2o::make_unique, number_t, g2o::cst — is that of the 2018-2020 master).
// Restated from g2o's published algorithm, and marked so in the code below:
//   RobustKernelHuber::robustify            rho = (e, 1) for e <= d^2, else (2 sqrt(e) d - d^2, d / sqrt(e))
//   Base{Unary,Binary,Multi}Edge::constructQuadraticForm   H += J^T (rho1 Omega) J,  b -= rho1 J^T Omega e
//   OptimizationAlgorithmLevenberg::solve   lambda0 = userLambdaInit, rho = (chi - chi') / (x.(lambda x + b) + 1e-3),
//                                           accept: lambda *= max(1/3, min(2/3, 1 - (2 rho - 1)^3)), ni = 2;
//                                           reject: lambda *= ni, ni *= 2, at most 10 trials
//   OptimizationAlgorithmGaussNewton::solve one undamped step; a failed factorisation (H = 0) leaves the estimate alone
//   SparseOptimizerTerminateAction          reset in initializeOptimization, stop when 0 <= gain < threshold after iteration >= 1
// Rotating a point: Sophus routes `SE3 * Vec3` through Eigen's quaternion product (so3.hpp:255-257; Eigen is absent too).
// Here, and in the device code, it is R*X + t with R = rotationMatrix(), evaluated row by row.
#include "orc_api.h"
#include "orc_common.h"
#include "orc_math.h"
#include <cmath>
#include <cstring>
#include <vector>

using namespace orc;

namespace {

// dso_util.hpp:25-46
inline bool CheckBoundary(double u, double v, int wl, int hl) { return (u - 2) < 0 || (u + 3) > wl || (v - 2) < 0 || (v + 3) > hl; }

inline void huber(double e2, double delta, double* rho) {   // g2o RobustKernelHuber::robustify
  const double dsqr = delta * delta;
  if (e2 <= dsqr) { rho[0] = e2; rho[1] = 1.; }
  else { const double sqrte = std::sqrt(e2); rho[0] = 2 * sqrte * delta - dsqr; rho[1] = delta / sqrte; }
}

struct TrackEdgeVal { double e; double J[8]; bool inside; };

// EdgeSE3PosePhotoDSO::computeError + linearizeOplus (dso_g2o_edge.cpp:395-500) for one edge
inline TrackEdgeVal trackEdge(const float* Xref, float meas_f, const float* dI, const orc_g2o_track_eval_t& ev) {
  TrackEdgeVal r;
  std::memset(&r, 0, sizeof(r));
  const double X[3] = {Xref[0], Xref[1], Xref[2]};
  double Xc[3];
  for (int i = 0; i < 3; i++) Xc[i] = ((ev.R[i * 3 + 0] * X[0] + ev.R[i * 3 + 1] * X[1]) + ev.R[i * 3 + 2] * X[2]) + ev.t[i];
  const double fx = ev.fx, fy = ev.fy, cx = ev.cx, cy = ev.cy;   // KG[level] is a Matrix3f (globalCalib.h:40)
  const double uvx = fx * (Xc[0] / Xc[2]) + cx;                   // dso_util.hpp:10-22
  const double uvy = fy * (Xc[1] / Xc[2]) + cy;
  if (CheckBoundary(uvx, uvy, ev.w, ev.h)) return r;              // error 0, Jacobians zero (:403-406, :431-435)
  r.inside = true;
  float hit[3];
  interp33(dI, (float)uvx, (float)uvy, ev.w, hit);
  const double meas = meas_f;
  if (std::isfinite(hit[0])) r.e = (double)hit[0] - ((double)ev.ab[0] * meas + (double)ev.ab[1]);   // :416-424 (else: left untouched -> 0 here)
  const double x = Xc[0], y = Xc[1], invz = 1.0 / Xc[2];
  const double u = x * invz, v = y * invz;
  const double dx = hit[1] * fx, dy = hit[2] * fy;
  r.J[0] = invz * dx;
  r.J[1] = invz * dy;
  r.J[2] = -invz * (u * dx + v * dy);
  r.J[3] = -(u * v * dx + (1 + v * v) * dy);
  r.J[4] = u * v * dy + (1 + u * u) * dx;
  r.J[5] = u * dy - v * dx;
  r.J[6] = (double)ev.ab[0] * (ev.b0 - meas);
  r.J[7] = -1;
  return r;
}
}  // namespace

// CoarseTracker::calcRes, fork-live body (CoarseTracker.cpp:600-792): culls with the pose it is CALLED with (float RKi, t),
// creates one edge per surviving point, evaluates it at the vertices' estimates and drops it when the signed error exceeds
// 10*cutoffTH.  Returns numTermsInE; rs = Vec6 of :783-789 (E is never accumulated in the live body -> 0).
extern "C" int orc_g2o_track_add_edges(int nl, const float* lpc_u, const float* lpc_v, const float* lpc_idepth, const float* lpc_color,
                                       const float* dINewl, const orc_g2o_track_eval_t* evp, double* rs, uint8_t* edge_mask, float* Xref_out) {
  const orc_g2o_track_eval_t& ev = *evp;
  const int lvl = ev.lvl, wl = ev.w, hl = ev.h;
  const float fxl = ev.fx, fyl = ev.fy, cxl = ev.cx, cyl = ev.cy;
  const float *RKi = ev.RKi, *Ki = ev.Ki, *t = ev.t_cull;
  float sumSquaredShiftT = 0, sumSquaredShiftRT = 0, sumSquaredShiftNum = 0;
  int numTermsInE = 0, numSaturated = 0;
  for (int i = 0; i < nl; i++) {
    edge_mask[i] = 0;
    Xref_out[i * 3] = Xref_out[i * 3 + 1] = Xref_out[i * 3 + 2] = 0;
    const float id = lpc_idepth[i], x = lpc_u[i], y = lpc_v[i];
    float pt[3], kp[3];
    for (int r = 0; r < 3; r++) {
      kp[r] = (Ki[r * 3 + 0] * x + Ki[r * 3 + 1] * y) + Ki[r * 3 + 2] * 1.0f;
      pt[r] = ((RKi[r * 3 + 0] * x + RKi[r * 3 + 1] * y) + RKi[r * 3 + 2] * 1.0f) + t[r] * id;
    }
    const float u = pt[0] / pt[2], v = pt[1] / pt[2];
    const float Ku = fxl * u + cxl, Kv = fyl * v + cyl;
    const float new_idepth = id / pt[2];
    if (lvl == 0 && i % 32 == 0) {  // :662-693
      float ptT[3], ptT2[3], pt3[3];
      for (int r = 0; r < 3; r++) {
        const float rp = (RKi[r * 3 + 0] * x + RKi[r * 3 + 1] * y) + RKi[r * 3 + 2] * 1.0f;
        ptT[r] = kp[r] + t[r] * id;
        ptT2[r] = kp[r] - t[r] * id;
        pt3[r] = rp - t[r] * id;
      }
      const float KuT = fxl * (ptT[0] / ptT[2]) + cxl, KvT = fyl * (ptT[1] / ptT[2]) + cyl;
      const float KuT2 = fxl * (ptT2[0] / ptT2[2]) + cxl, KvT2 = fyl * (ptT2[1] / ptT2[2]) + cyl;
      const float Ku3 = fxl * (pt3[0] / pt3[2]) + cxl, Kv3 = fyl * (pt3[1] / pt3[2]) + cyl;
      sumSquaredShiftT += (KuT - x) * (KuT - x) + (KvT - y) * (KvT - y);
      sumSquaredShiftT += (KuT2 - x) * (KuT2 - x) + (KvT2 - y) * (KvT2 - y);
      sumSquaredShiftRT += (Ku - x) * (Ku - x) + (Kv - y) * (Kv - y);
      sumSquaredShiftRT += (Ku3 - x) * (Ku3 - x) + (Kv3 - y) * (Kv3 - y);
      sumSquaredShiftNum += 2;
    }
    if (!(Ku > 2 && Kv > 2 && Ku < wl - 3 && Kv < hl - 3 && new_idepth > 0)) continue;   // :696
    float Xref[3];
    for (int r = 0; r < 3; r++) Xref[r] = kp[r] / id;                                     // :707  Ki[lvl] * Vec3f(x,y,1) / id
    const TrackEdgeVal ev1 = trackEdge(Xref, lpc_color[i], dINewl, ev);                   // edge->computeError() :721
    if (ev1.e > ev.cutoffTH * 10) { numSaturated++; continue; }                           // :724-727
    edge_mask[i] = 1;
    for (int r = 0; r < 3; r++) Xref_out[i * 3 + r] = Xref[r];
    numTermsInE++;
  }
  rs[0] = 0;
  rs[1] = numTermsInE;
  rs[2] = sumSquaredShiftT / (sumSquaredShiftNum + 0.1);
  rs[3] = 0;
  rs[4] = sumSquaredShiftRT / (sumSquaredShiftNum + 0.1);
  rs[5] = numSaturated / (float)numTermsInE;
  return numTermsInE;
}

// computeActiveErrors + linearizeOplus + constructQuadraticForm over the edges of one level.  H 8x8 row-major ordered
// [pose 6 | photometric 2] (vertex ids 0, 1: CoarseTracker.cpp:876-885), b, chi2 = {sum e^2, sum rho0 = activeRobustChi2}.
// err[n], J[n*8] optional, indexed like the pc arrays (zero where there is no edge).
extern "C" int orc_g2o_track_linearize(int nl, const uint8_t* edge_mask, const float* Xref, const float* lpc_color, const float* dINewl,
                                       const orc_g2o_track_eval_t* evp, double* H, double* b, double* chi2, double* err, double* J) {
  const orc_g2o_track_eval_t& ev = *evp;
  for (int i = 0; i < 64; i++) H[i] = 0;
  for (int i = 0; i < 8; i++) b[i] = 0;
  chi2[0] = chi2[1] = 0;
  int ne = 0;
  for (int i = 0; i < nl; i++) {
    if (err) err[i] = 0;
    if (J) for (int k = 0; k < 8; k++) J[i * 8 + k] = 0;
    if (!edge_mask[i]) continue;
    ne++;
    const TrackEdgeVal v = trackEdge(Xref + i * 3, lpc_color[i], dINewl, ev);
    if (err) err[i] = v.e;
    if (J) for (int k = 0; k < 8; k++) J[i * 8 + k] = v.J[k];
    double rho[2];
    huber(v.e * v.e, ev.huberTH, rho);
    chi2[0] += v.e * v.e;
    chi2[1] += rho[0];
    for (int r = 0; r < 8; r++) {
      b[r] -= rho[1] * v.J[r] * v.e;
      for (int c = 0; c < 8; c++) H[r * 8 + c] += v.J[r] * rho[1] * v.J[c];
    }
  }
  return ne;
}

// CoarseTracker::trackNewestCoarse, fork-live (CoarseTracker.cpp:827-1069): per level calcRes (always with the INITIAL pose,
// :890 — refToNew_current is never updated), initializeOptimization(lvl), optimize(2) with g2o's Levenberg-Marquardt.
extern "C" int orc_g2o_track_newest_coarse(const int* pc_n, const float* const* pc_u, const float* const* pc_v, const float* const* pc_idepth,
                                           const float* const* pc_color, const float* const* dIp, const orc_track_params_t* prm,
                                           orc_se3_t* lastToNew, orc_aff_t* aff_g2l, orc_track_result_t* out) {
  for (int i = 0; i < 5; i++) { out->lastResiduals[i] = NAN; out->iterations[i] = 0; }
  for (int i = 0; i < 3; i++) out->lastFlowIndicators[i] = 1000;
  out->good = 0; out->evaluations = 0; out->point_evals = 0;
  SE3 pose; std::memcpy(pose.R, lastToNew->R, sizeof(pose.R)); std::memcpy(pose.t, lastToNew->t, sizeof(pose.t));
  double aff[2] = {aff_g2l->a, aff_g2l->b};
  const orc_se3_t refToNew_current = *lastToNew;
  const int maxIterations[5] = {2, 2, 2, 2, 2};   // :861
  size_t total_edges = 0;

  auto make_ev = [&](int lvl, const SE3& T, const double* a1b1) {
    orc_g2o_track_eval_t ev;
    orc_track_eval_t base;
    orc_aff_t dummy = {a1b1[0], a1b1[1]};
    orc_track_make_eval(prm, lvl, &refToNew_current, &dummy, 1.0f, &base);
    ev.lvl = lvl; ev.w = base.w; ev.h = base.h; ev.fx = base.fx; ev.fy = base.fy; ev.cx = base.cx; ev.cy = base.cy;
    std::memcpy(ev.Ki, base.Ki, sizeof(ev.Ki)); std::memcpy(ev.RKi, base.RKi, sizeof(ev.RKi)); std::memcpy(ev.t_cull, base.t, sizeof(ev.t_cull));
    std::memcpy(ev.R, T.R, sizeof(ev.R)); std::memcpy(ev.t, T.t, sizeof(ev.t));
    ev.ab[0] = base.affLL[0]; ev.ab[1] = base.affLL[1];
    ev.b0 = prm->ref_aff_g2l.b;
    ev.cutoffTH = base.cutoffTH; ev.huberTH = base.huberTH;
    return ev;
  };

  for (int lvl = prm->coarsestLvl; lvl >= 0; lvl--) {
    const int n = pc_n[lvl];
    std::vector<uint8_t> mask(n > 0 ? n : 1);
    std::vector<float> Xref(3 * (n > 0 ? n : 1));
    double resOld[6];
    orc_g2o_track_eval_t ev = make_ev(lvl, pose, aff);
    const int ne = orc_g2o_track_add_edges(n, pc_u[lvl], pc_v[lvl], pc_idepth[lvl], pc_color[lvl], dIp[lvl], &ev, resOld, mask.data(), Xref.data());
    total_edges += ne;
    out->evaluations++; out->point_evals += n;

    double H[64], b[8], chi[2];
    auto linearize = [&](const SE3& T, const double* a1b1) {
      orc_g2o_track_eval_t e2 = make_ev(lvl, T, a1b1);
      orc_g2o_track_linearize(n, mask.data(), Xref.data(), pc_color[lvl], dIp[lvl], &e2, H, b, chi, nullptr, nullptr);
      out->evaluations++; out->point_evals += n;
    };

    // g2o SparseOptimizer::optimize(maxIterations[lvl]) with OptimizationAlgorithmLevenberg (userLambdaInit 0.01, :839-840)
    double lambda = 0, ni = 2, lastChi = 0;
    bool stop = false, ok = true;
    for (int it = 0; it < maxIterations[lvl] && !stop && ok; it++) {
      linearize(pose, aff);                       // computeActiveErrors + buildSystem
      double currentChi = chi[1];
      if (it == 0) { lambda = 0.01; ni = 2; }
      double rho = 0;
      int qmax = 0;
      double Hs[64], bs[8];
      std::memcpy(Hs, H, sizeof(H)); std::memcpy(bs, b, sizeof(b));
      do {
        MatX A(8, 8); VecX rhs(8), x(8);
        for (int r = 0; r < 8; r++) { rhs[r] = bs[r]; for (int c = 0; c < 8; c++) A(r, c) = Hs[r * 8 + c]; A(r, r) += lambda; }
        const bool ok2 = ldlt_solve(A, rhs, x);
        SE3 trial = pose; double affT[2] = {aff[0], aff[1]};
        if (ok2) {
          double xi[6]; for (int k = 0; k < 6; k++) xi[k] = x[k];
          trial = se3_mul(se3_exp(xi), pose);     // VertexSE3PoseDSO::oplusImpl (dso_g2o_vertex.cpp:15-18)
          affT[0] += x[6]; affT[1] += x[7];       // VertexPhotometricDSO::oplusImpl (:30-40)
        }
        linearize(trial, affT);                   // computeActiveErrors of the trial (H, b of the trial are not used)
        double tempChi = ok2 ? chi[1] : 1.7976931348623157e308;
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
      if (qmax == 10 || rho == 0 || !std::isfinite(lambda)) ok = false;   // SolverResult::Terminate ends optimize()
      // SparseOptimizerTerminateAction (gain threshold 1e-3, :846-848), run as post-iteration action
      linearize(pose, aff);
      if (it == 0) lastChi = chi[1];
      else {
        const double gain = (lastChi - chi[1]) / chi[1];
        lastChi = chi[1];
        if (gain >= 0 && gain < 1e-3) stop = true;
      }
    }
    if (maxIterations[lvl] == 0 || n == 0) linearize(pose, aff);
    out->lastResiduals[lvl] = sqrtf((float)chi[1] / total_edges);          // :1029  edges().size() counts every level so far
    for (int k = 0; k < 3; k++) out->lastFlowIndicators[k] = resOld[2 + k];
    if (out->lastResiduals[lvl] > 1.5 * prm->minResForAbort[lvl]) return 0;   // returns before the outputs are written (:1032-1033)
  }
  std::memcpy(lastToNew->R, pose.R, sizeof(pose.R)); std::memcpy(lastToNew->t, pose.t, sizeof(pose.t));
  aff_g2l->a = aff[0]; aff_g2l->b = aff[1];
  if ((prm->affineOptModeA != 0 && (fabsf((float)aff_g2l->a) > 1.2)) || (prm->affineOptModeB != 0 && (fabsf((float)aff_g2l->b) > 200))) return 0;
  double relAff[2];
  fromToVecExposure(prm->ref_exposure, prm->new_exposure, prm->ref_aff_g2l.a, prm->ref_aff_g2l.b, aff_g2l->a, aff_g2l->b, relAff);
  if ((prm->affineOptModeA == 0 && (fabsf(logf((float)relAff[0])) > 1.5)) || (prm->affineOptModeB == 0 && (fabsf((float)relAff[1]) > 200))) return 0;
  if (prm->affineOptModeA < 0) aff_g2l->a = 0;
  if (prm->affineOptModeB < 0) aff_g2l->b = 0;
  out->good = 1;
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// EdgeLBASE3PosePhotoIdepthCamDSO (dso_g2o_edge.cpp:5-282).  The edge object keeps state between calls: its error vector,
// the Jacobians of the last COMPLETED linearizeOplus (the locals are copied into _jacobianOplus only at the end, :278-281, so
// an early return leaves the previous ones in place), its g2o level (setLevel(1) is never undone), and through r_ the
// residual's state / energies, through the idepth vertex CenterProjectedTo.
namespace {
struct LbaEdge {
  double e[8];
  double J[104];          // rows [xi 6 | photometric 2 | idepth 1 | camera 4]
  uint8_t state, level;
  float energy[2];        // state_NewEnergy, state_NewEnergyWithOutlier
  float cpt[3];
  float ih;               // r_->point->idepth_hessian as last written by this edge
  void init() {
    std::memset(this, 0, sizeof(*this));
    cpt[0] = 2; cpt[1] = 2; cpt[2] = 0;       // VertexInverseDepthDSO(), dso_g2o_vertex.cpp:44-46
  }
};
struct LbaPair { const float* R; const float* t; const float* ab; };

// computeError :5-133
void lbaComputeError(const orc_g2o_lba_t* L, int ri, const LbaPair& P, const double* cam, double idepth, LbaEdge& S) {
  const double fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3];
  const int h = L->host[ri], tg = L->target[ri];
  const float *R = P.R, *t = P.t, *ab = P.ab;
  const float* dIl = L->dI[tg];
  const float pu = L->u[ri], pv = L->v[ri];
  const float* color = L->color + (size_t)ri * 8;
  const float* weights = L->weights + (size_t)ri * 8;
  const float eTH = std::max<float>(L->frameEnergyTH[h], L->frameEnergyTH[tg]);
  float energyLeft = 0, wJI2_sum = 0;
  for (int idx = 0; idx < patternNum; idx++) {
    const double u_host = pu + patternP[idx][0];
    const double v_host = pv + patternP[idx][1];
    const float Klip[3] = {(float)((u_host - cx) / fx), (float)((v_host - cy) / fy), 1};
    float ptp[3];
    for (int r = 0; r < 3; r++) ptp[r] = ((R[r * 3 + 0] * Klip[0] + R[r * 3 + 1] * Klip[1]) + R[r * 3 + 2] * Klip[2]) + t[r] * (float)idepth;
    const double drescale = 1.0f / ptp[2];
    if (drescale <= 0) { S.state = 1; for (int k = 0; k < 8; k++) S.e[k] = 0; return; }
    const double new_idepth = idepth * drescale;
    const double _u = ptp[0] * drescale, _v = ptp[1] * drescale;
    const double _Ku = _u * fx + cx, _Kv = _v * fy + cy;
    if (CheckBoundary(_Ku, _Kv, L->w - 3, L->h - 3)) { S.state = 1; for (int k = 0; k < 8; k++) S.e[k] = 0; S.level = 1; return; }
    else if (u_host == pu && v_host == pv) { S.cpt[0] = (float)_Ku; S.cpt[1] = (float)_Kv; S.cpt[2] = (float)new_idepth; }
    float hit[3];
    interp33(dIl, (float)_Ku, (float)_Kv, L->w, hit);
    if (!std::isfinite(hit[0])) { S.state = 1; S.e[idx] = 0; continue; }
    S.e[idx] = hit[0] - (ab[0] * color[idx] + ab[1]);
    float w = sqrtf(setting_outlierTHSumComponent / (setting_outlierTHSumComponent + (hit[1] * hit[1] + hit[2] * hit[2])));
    w = 0.5f * (w + weights[idx]);
    const float hw = fabsf((float)S.e[idx]) < setting_huberTH ? 1 : setting_huberTH / fabsf((float)S.e[idx]);
    energyLeft += w * w * hw * S.e[idx] * S.e[idx] * (2 - hw);
    wJI2_sum += hw * hw * (hit[1] * hit[1] + hit[2] * hit[2]);
  }
  S.energy[1] = energyLeft;
  if (energyLeft > eTH || wJI2_sum < 2) { energyLeft = eTH; S.state = 2; } else S.state = 0;
  S.energy[0] = energyLeft;
}

// linearizeOplus :135-282
void lbaLinearize(const orc_g2o_lba_t* L, int ri, const LbaPair& P, const double* cam, double idepth, double b0, LbaEdge& S) {
  if (S.level == 1 || S.state == 1) return;
  const double fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3];
  const int tg = L->target[ri];
  const float *R = P.R, *t = P.t, *ab = P.ab;
  const float* dIl = L->dI[tg];
  const float pu = L->u[ri], pv = L->v[ri];
  const float* color = L->color + (size_t)ri * 8;
  float H_idepth_idepth = 0;
  double Jtmp[104];
  for (int k = 0; k < 104; k++) Jtmp[k] = 0;
  for (int idx = 0; idx < patternNum; idx++) {
    const double u_host = pu + patternP[idx][0];
    const double v_host = pv + patternP[idx][1];
    const float Klip[3] = {(float)((u_host - cx) / fx), (float)((v_host - cy) / fy), 1};
    float ptp[3];
    for (int r = 0; r < 3; r++) ptp[r] = ((R[r * 3 + 0] * Klip[0] + R[r * 3 + 1] * Klip[1]) + R[r * 3 + 2] * Klip[2]) + t[r] * (float)idepth;
    const double drescale = 1.0f / ptp[2];
    if (drescale <= 0) { S.state = 1; return; }
    const double new_idepth = idepth * drescale;
    const double _u = ptp[0] * drescale, _v = ptp[1] * drescale;
    const double _Ku = _u * fx + cx, _Kv = _v * fy + cy;
    if (CheckBoundary(_Ku, _Kv, L->w - 3, L->h - 3)) { S.state = 1; return; }
    float hit[3];
    interp33(dIl, (float)_Ku, (float)_Kv, L->w, hit);
    if (!std::isfinite(hit[0])) { S.state = 1; return; }
    const double fxi = 1 / fx, fyi = 1 / fy;
    const double p0 = hit[1], p1 = hit[2];
    double C[2][4];
    C[0][2] = drescale * (R[6] * _u - R[0]);
    C[0][3] = fx * fyi * drescale * (R[7] * _u - R[1]);
    C[0][0] = Klip[0] * C[0][2];
    C[0][1] = Klip[1] * C[0][3];
    C[1][2] = fy * fxi * drescale * (R[6] * _v - R[3]);
    C[1][3] = drescale * (R[7] * _v - R[4]);
    C[1][0] = Klip[0] * C[1][2];
    C[1][1] = Klip[1] * C[1][3];
    double* row = Jtmp + idx * 13;
    for (int c = 0; c < 4; c++) row[9 + c] = p0 * C[0][c] + p1 * C[1][c];
    const double dx = hit[1] * fx, dy = hit[2] * fy;
    row[0] = new_idepth * dx;
    row[1] = new_idepth * dy;
    row[2] = -new_idepth * (_u * dx + _v * dy);
    row[3] = -(_u * _v * dx + (1 + _v * _v) * dy);
    row[4] = _u * _v * dy + (1 + _u * _u) * dx;
    row[5] = _u * dy - _v * dx;
    row[6] = ab[0] * (b0 - color[idx]);
    row[7] = -1;
    row[8] = dx * drescale * (t[0] - t[2] * _u) + dy * drescale * (t[1] - t[2] * _v);
    H_idepth_idepth += row[8] * row[8];
  }
  if (H_idepth_idepth < 1e-10) H_idepth_idepth = 1e-10;
  S.ih = H_idepth_idepth;
  for (int k = 0; k < 104; k++) S.J[k] = Jtmp[k];
}
}  // namespace

// One computeError + linearizeOplus of fresh edges (what the graph-building loop and the first buildSystem do).
//   error[nr*8], J[nr*8*13], state (0 IN, 1 OOB, 2 OUTLIER — ResState, Residuals.h:49), energy[nr*2], centerProjectedTo[nr*3],
//   idepth_hessian[nr], edge_level[nr].
extern "C" int orc_g2o_lba_eval(const orc_g2o_lba_t* L, double* error, double* J, uint8_t* state, float* energy, float* centerProjectedTo,
                                float* idepth_hessian, uint8_t* edge_level) {
  const int nf = L->nf;
  for (int ri = 0; ri < L->nr; ri++) {
    const int h = L->host[ri], tg = L->target[ri];
    const LbaPair P = {L->pair_R + (size_t)(h * nf + tg) * 9, L->pair_t + (size_t)(h * nf + tg) * 3, L->pair_ab + (size_t)(h * nf + tg) * 2};
    LbaEdge S;
    S.init();
    lbaComputeError(L, ri, P, L->cam, L->idepth[ri], S);
    lbaLinearize(L, ri, P, L->cam, L->idepth[ri], L->host_b0[h], S);
    for (int k = 0; k < 8; k++) error[(size_t)ri * 8 + k] = S.e[k];
    for (int k = 0; k < 104; k++) J[(size_t)ri * 104 + k] = S.J[k];
    state[ri] = S.state; edge_level[ri] = S.level;
    energy[ri * 2] = S.energy[0]; energy[ri * 2 + 1] = S.energy[1];
    for (int k = 0; k < 3; k++) centerProjectedTo[ri * 3 + k] = S.cpt[k];
    idepth_hessian[ri] = S.ih;
  }
  return 0;
}
