// sdso_shim.h — header-only C++ host layer that keeps the reference's call surface
// (CoarseTracker / EnergyFunctional / ImmaturePoint) on top of the C-ABI of
// libsdso_hip.so.  It is written against the reference's member NAMES through templates, so the same
// header compiles
//   * inside the reference tree against its real types (Eigen/Sophus: SE3 = Sophus::SE3d, AffLight,
//     FrameHessian, CalibHessian, EnergyFunctional, EFFrame/EFPoint/EFResidual, ImmaturePoint), and
//   * in this repository against the small stand-ins of host/test_shim.cpp (Eigen is not available here).
// Nothing here computes on the CPU what the library computes on the GPU; the shim only marshals.
//
// Reference signatures mirrored (paths under /root/reference/src):
//   bool CoarseTracker::trackNewestCoarse(FrameHessian*, SE3&, AffLight&, int coarsestLvl, Vec5 minResForAbort, ...)   FullSystem/CoarseTracker.h:62-66
//   void CoarseTracker::makeK(CalibHessian*)                                                                        FullSystem/CoarseTracker.h:76
//   Vec6 CoarseTracker::calcRes(int lvl, SE3 refToNew, AffLight aff_g2l, float cutoffTH) + calcGSSSE(lvl,H,b,...)    FullSystem/CoarseTracker.cpp:600, :537
//   void EnergyFunctional::solveSystemF(int iteration, double lambda, CalibHessian*)                                 OptimizationBackend/EnergyFunctional.h:74
//   void EnergyFunctional::marginalizePointsF()                                                                      OptimizationBackend/EnergyFunctional.h:69
//   float FullSystem::optimize(int mnumOptIts)                                                                       FullSystem/FullSystemOptimize.cpp:871
//   ImmaturePointStatus ImmaturePoint::traceStereo(FrameHessian* frame, Mat33f K, bool mode_right)                   FullSystem/ImmaturePoint.h:89
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>
#include <utility>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/sdso_abi.h"

namespace sdso_shim {

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

// One GPU + stream.  The reference serialises tracking under trackMutex and mapping under mapMutex;
// use one Device per such domain (or one shared Device and the same mutexes).
class Device {
 public:
  explicit Device(int ordinal = 0) {
    if (sdso_ctx_create(ordinal, &ctx_) != SDSO_OK) throw Error("sdso_ctx_create failed: no usable HIP device (there is no CPU fallback)");
  }
  ~Device() { sdso_ctx_destroy(ctx_); }
  Device(const Device&) = delete;
  Device& operator=(const Device&) = delete;
  sdso_ctx* ctx() const { return ctx_; }
  void check(int rc, const char* what) const {
    if (rc != SDSO_OK) throw Error(std::string(what) + ": " + sdso_last_error(ctx_));
  }
  // FrameHessian::dIp mirrors: immutable after makeImages, so upload once per frame (key = shell->id / frameID)
  template <class FrameHessianT>
  void uploadFrame(int slot, const FrameHessianT* fh, int levels, const int* w, const int* h) {
    std::vector<const float*> p(levels);
    for (int l = 0; l < levels; l++) p[l] = reinterpret_cast<const float*>(fh->dIp[l]);  // Eigen::Vector3f is 3 packed floats
    check(sdso_upload_pyramid(ctx_, slot, levels, w, h, p.data()), "sdso_upload_pyramid");
  }
  void releaseFrame(int slot) { sdso_release_pyramid(ctx_, slot); }
  // traceStereo's sub-pixel refinement: false = DSO-native GN (ImmaturePoint.cpp:707-769), true = the fork's g2o GN on
  // EdgeTracePointUVDSO (ImmaturePoint.cpp:309-412)
  // Multi-GPU: one process per GPU; rank 0 makes the id (static uniqueId()), the caller ships the 128 bytes to the other ranks,
  // then every rank calls commInit (collective).  RCCL is resolved at run time; a missing librccl is an error, not a fallback.
  static void uniqueId(unsigned char id[128]) {
    if (sdso_comm_unique_id(id) != SDSO_OK) throw Error("sdso_comm_unique_id failed (librccl.so.1 not loadable?)");
  }
  void commInit(int nranks, int rank, const unsigned char id[128]) { check(sdso_comm_init(ctx_, nranks, rank, id), "sdso_comm_init"); }
  void setForkLiveTraceRefinement(bool on) { check(sdso_trace_set_gn_mode(ctx_, on ? 1 : 0), "sdso_trace_set_gn_mode"); }

 private:
  sdso_ctx* ctx_ = nullptr;
};

// ---- SE3 / AffLight marshalling (Sophus::SE3d API: rotationMatrix()(i,j), translation()[i], ctor(R, t))
template <class SE3T>
inline sdso_se3_t toAbi(const SE3T& T) {
  sdso_se3_t o;
  const auto R = T.rotationMatrix();
  const auto t = T.translation();
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) o.R[i * 3 + j] = R(i, j); o.t[i] = t[i]; }
  return o;
}
template <class SE3T, class Mat33T, class Vec3T>
inline SE3T fromAbi(const sdso_se3_t& a) {
  Mat33T R;
  Vec3T t;
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R(i, j) = a.R[i * 3 + j]; t[i] = a.t[i]; }
  return SE3T(R, t);
}

// =================================================================================== CoarseTracker
// Drop-in for the tracking half of dso::CoarseTracker.  Template parameters are the reference's own
// types; Mat33/Vec3 are the Eigen double types used to rebuild an SE3.
template <class SE3T, class AffLightT, class Mat33T, class Vec3T>
class CoarseTracker {
 public:
  CoarseTracker(Device& dev, int ref_slot) : dev_(dev), ref_slot_(ref_slot) {
    std::memset(&prm_, 0, sizeof(prm_));
    prm_.coarseCutoffTH = 20.f; prm_.huberTH = 9.f;                 // settings.cpp:102, :95
    const int its[5] = {10, 20, 50, 50, 50};                         // CoarseTracker.cpp:861 (DSO-native)
    for (int i = 0; i < 5; i++) { prm_.maxIterations[i] = its[i]; lastResiduals[i] = NAN; }
    prm_.affineOptModeA = 1e12; prm_.affineOptModeB = 1e8;            // settings.cpp:90-91
    for (int i = 0; i < 3; i++) lastFlowIndicators[i] = 1000;
  }
  // makeK(CalibHessian*): per-level intrinsics exactly as CoarseTracker.cpp:108-136
  template <class CalibHessianT>
  void makeK(CalibHessianT* HCalib, int pyrLevelsUsed, int w0, int h0) {
    prm_.levels = pyrLevelsUsed;
    prm_.w[0] = w0; prm_.h[0] = h0;
    prm_.fx[0] = HCalib->fxl(); prm_.fy[0] = HCalib->fyl(); prm_.cx[0] = HCalib->cxl(); prm_.cy[0] = HCalib->cyl();
    for (int l = 1; l < pyrLevelsUsed; ++l) {
      prm_.w[l] = w0 >> l; prm_.h[l] = h0 >> l;
      prm_.fx[l] = prm_.fx[l - 1] * 0.5; prm_.fy[l] = prm_.fy[l - 1] * 0.5;
      prm_.cx[l] = (prm_.cx[0] + 0.5) / ((int)1 << l) - 0.5;
      prm_.cy[l] = (prm_.cy[0] + 0.5) / ((int)1 << l) - 0.5;
    }
  }
  // The template point cloud the reference builds in makeCoarseDepthL0 (CoarseTracker.cpp:507-532):
  // pc_u/pc_v/pc_idepth/pc_color[lvl], pc_n[lvl].  Called once per new reference keyframe.
  void setCoarseTrackingRef(int lvl, int pc_n, const float* pc_u, const float* pc_v, const float* pc_idepth, const float* pc_color,
                            float lastRef_ab_exposure, const AffLightT& lastRef_aff_g2l_, int refFrameID_) {
    dev_.check(sdso_track_set_ref(dev_.ctx(), ref_slot_, lvl, pc_n, pc_u, pc_v, pc_idepth, pc_color), "sdso_track_set_ref");
    prm_.ref_exposure = lastRef_ab_exposure;
    prm_.ref_aff_g2l.a = lastRef_aff_g2l_.a; prm_.ref_aff_g2l.b = lastRef_aff_g2l_.b;
    refFrameID = refFrameID_;
    firstCoarseRMSE = -1;
  }
  // makeCoarseDepthL0 (CoarseTracker.cpp:275-534) from STEP1's per-point results: integer pixel (u,v) on lastRef, the
  // (stereo-refined) new_idepth and weight = sqrtf(1e-3 / (HdiF + 1e-12)).  Splat, pyramid, dilation, normalisation and
  // the raster-order compaction run on the device; pc_n[lvl] comes back for the reference's bookkeeping.
  void makeCoarseDepthL0(int lastRef_slot, int n, const int* u, const int* v, const float* new_idepth, const float* weight, int* pc_n,
                         float lastRef_ab_exposure, const AffLightT& lastRef_aff_g2l_, int refFrameID_) {
    dev_.check(sdso_track_make_ref(dev_.ctx(), ref_slot_, lastRef_slot, n, u, v, new_idepth, weight, pc_n), "sdso_track_make_ref");
    prm_.ref_exposure = lastRef_ab_exposure;
    prm_.ref_aff_g2l.a = lastRef_aff_g2l_.a; prm_.ref_aff_g2l.b = lastRef_aff_g2l_.b;
    refFrameID = refFrameID_;
    firstCoarseRMSE = -1;
  }
  // bool trackNewestCoarse(FrameHessian* newFrameHessian, SE3& lastToNew_out, AffLight& aff_g2l_out, int coarsestLvl, Vec5 minResForAbort)
  template <class Vec5T>
  bool trackNewestCoarse(int newFrame_slot, float newFrame_ab_exposure, SE3T& lastToNew_out, AffLightT& aff_g2l_out, int coarsestLvl,
                         const Vec5T& minResForAbort) {
    prm_.new_exposure = newFrame_ab_exposure;
    prm_.coarsestLvl = coarsestLvl;
    for (int i = 0; i < 5; i++) prm_.minResForAbort[i] = minResForAbort[i];
    sdso_se3_t T = toAbi(lastToNew_out);
    sdso_aff_t aff{aff_g2l_out.a, aff_g2l_out.b};
    sdso_track_result_t out;
    if (forkLive) dev_.check(sdso_g2o_track_newest_coarse(dev_.ctx(), ref_slot_, newFrame_slot, &prm_, &T, &aff, &out), "sdso_g2o_track_newest_coarse");
    else dev_.check(sdso_track_newest_coarse(dev_.ctx(), ref_slot_, newFrame_slot, &prm_, &T, &aff, &out), "sdso_track_newest_coarse");
    for (int i = 0; i < 5; i++) lastResiduals[i] = out.lastResiduals[i];
    for (int i = 0; i < 3; i++) lastFlowIndicators[i] = out.lastFlowIndicators[i];
    lastToNew_out = fromAbi<SE3T, Mat33T, Vec3T>(T);
    aff_g2l_out.a = aff.a; aff_g2l_out.b = aff.b;
    lastStats = out;
    return out.good != 0;
  }
  // calcRes + calcGSSSE for one (level, pose): what the LM loop evaluates; returns calcRes' Vec6
  void calcResAndGS(int lvl, int newFrame_slot, float newFrame_ab_exposure, const SE3T& refToNew, const AffLightT& aff_g2l, float levelCutoffRepeat,
                    double H_out[64], double b_out[8], double res6[6], int* buf_warped_n = nullptr) {
    prm_.new_exposure = newFrame_ab_exposure;
    sdso_track_eval_t ev;
    const sdso_se3_t T = toAbi(refToNew);
    const sdso_aff_t a{aff_g2l.a, aff_g2l.b};
    sdso_track_make_eval(&prm_, lvl, &T, &a, levelCutoffRepeat, &ev);
    dev_.check(sdso_track_calc_res_gs(dev_.ctx(), ref_slot_, newFrame_slot, &ev, H_out, b_out, res6, buf_warped_n, nullptr), "sdso_track_calc_res_gs");
  }

  // false: DSO-native LM (the body kept as comments at CoarseTracker.cpp:908-1024).  true: the fork's live body — one
  // EdgeSE3PosePhotoDSO per point and g2o's Levenberg-Marquardt, 2 iterations per level (CoarseTracker.cpp:834-1047).
  bool forkLive = false;
  // public outputs of the reference (CoarseTracker.h:98-113)
  double lastResiduals[5];
  double lastFlowIndicators[3];
  double firstCoarseRMSE = -1;
  int refFrameID = -1;
  sdso_track_result_t lastStats{};
  sdso_track_params_t& params() { return prm_; }

 private:
  Device& dev_;
  int ref_slot_;
  sdso_track_params_t prm_;
};

// =================================================================================== EnergyFunctional
// Flattens the reference's pointer graph (EnergyFunctional::frames -> EFFrame::points -> EFPoint::residualsAll,
// i.e. makeIDX order, EnergyFunctional.cpp:998-1018) into sdso_ba_window_t and drives the device window.
// slot_of(FrameHessian*) returns the pyramid slot the frame was uploaded to.
template <class EnergyFunctionalT, class CalibHessianT>
class WindowedBA {
 public:
  WindowedBA(Device& dev, int win_id) : dev_(dev), win_(win_id) {}

  template <class SlotOf>
  void upload(EnergyFunctionalT* ef, CalibHessianT* HCalib, int w, int h, int solverMode, double affineOptModeA, double affineOptModeB,
              bool forceAcceptStep, SlotOf slot_of) {
    const int nf = (int)ef->frames.size();
    evalPT_.assign(nf * 12, 0); state_.assign(nf * 10, 0); state_zero_.assign(nf * 10, 0);
    exposure_.assign(nf, 1.f); energyTH_.assign(nf, 0.f); frameID_.assign(nf, 0); slots_.assign(nf, 0);
    for (int f = 0; f < nf; f++) {
      auto* fh = ef->frames[f]->data;
      const sdso_se3_t T = toAbi(fh->get_worldToCam_evalPT());
      std::memcpy(&evalPT_[f * 12], T.R, 72); std::memcpy(&evalPT_[f * 12 + 9], T.t, 24);
      for (int i = 0; i < 10; i++) { state_[f * 10 + i] = fh->get_state()[i]; state_zero_[f * 10 + i] = fh->get_state_zero()[i]; }
      exposure_[f] = fh->ab_exposure; energyTH_[f] = fh->frameEnergyTH; frameID_[f] = fh->frameID; slots_[f] = slot_of(fh);
    }
    u_.clear(); v_.clear(); idepth_.clear(); idepth_zero_.clear(); color_.clear(); weights_.clear(); host_.clear(); prior_.clear();
    res_point_.clear(); res_target_.clear(); res_state_.clear(); points_.clear(); residuals_.clear();
    for (int f = 0; f < nf; f++)
      for (auto* p : ef->frames[f]->points) {
        auto* ph = p->data;
        const int pi = (int)u_.size();
        u_.push_back(ph->u); v_.push_back(ph->v); idepth_.push_back(ph->idepth); idepth_zero_.push_back(ph->idepth_zero);
        for (int k = 0; k < 8; k++) { color_.push_back(ph->color[k]); weights_.push_back(ph->weights[k]); }
        host_.push_back(f); prior_.push_back(ph->hasDepthPrior ? 1 : 0);
        points_.push_back(p);
        for (auto* r : p->residualsAll) {
          res_point_.push_back(pi); res_target_.push_back(r->target->idx); res_state_.push_back((uint8_t)r->data->state_state);
          residuals_.push_back(r);
        }
      }
    sdso_ba_window_t W;
    std::memset(&W, 0, sizeof(W));
    W.nf = nf; W.np = (int)u_.size(); W.nr = (int)res_point_.size(); W.w = w; W.h = h;
    for (int i = 0; i < 4; i++) { W.calib_value_scaled[i] = HCalib->value_scaled[i]; W.calib_value_zero[i] = HCalib->value_zero[i]; }
    W.evalPT = evalPT_.data(); W.state = state_.data(); W.state_zero = state_zero_.data();
    W.ab_exposure = exposure_.data(); W.frameEnergyTH = energyTH_.data(); W.frameID = frameID_.data(); W.frame_slot = slots_.data();
    W.u = u_.data(); W.v = v_.data(); W.idepth = idepth_.data(); W.idepth_zero = idepth_zero_.data();
    W.color = color_.data(); W.weights = weights_.data(); W.host = host_.data(); W.hasDepthPrior = prior_.data();
    W.res_point = res_point_.data(); W.res_target = res_target_.data(); W.res_state = res_state_.data();
    const int n = 8 * nf + 4;
    HM_.assign((size_t)n * n, 0); bM_.assign(n, 0);
    for (int i = 0; i < n; i++) { bM_[i] = ef->bM[i]; for (int j = 0; j < n; j++) HM_[(size_t)i * n + j] = ef->HM(i, j); }
    W.HM = HM_.data(); W.bM = bM_.data();
    W.solverMode = solverMode; W.affineOptModeA = affineOptModeA; W.affineOptModeB = affineOptModeB; W.forceAcceptStep = forceAcceptStep ? 1 : 0;
    dev_.check(sdso_ba_upload_window(dev_.ctx(), win_, &W), "sdso_ba_upload_window");
    nf_ = nf;
  }

  // Vec3 FullSystem::linearizeAll(false): returns lastEnergyP
  double linearizeAll() { double e = 0; dev_.check(sdso_ba_linearize(dev_.ctx(), win_, &e), "sdso_ba_linearize"); return e; }
  // applyRes_Reductor(true, ...)
  void applyRes() { dev_.check(sdso_ba_apply_res(dev_.ctx(), win_), "sdso_ba_apply_res"); }
  // EnergyFunctional::solveSystemF(iteration, lambda, HCalib): fills lastX; frame / calib / point steps are fetched below
  // Multi-GPU: when this process holds only a contiguous range of allPoints, sum the packed accumulators over the
  // ranks before the stitch — what stitchDoubleMT does with the per-thread copies (AccumulatedTopHessian.cpp:299-308), across GPUs.
  // Device::commInit must have been called (sdso_comm_unique_id / sdso_comm_init); a no-op without a communicator.
  void allreduce() {
    int nranks = 0;
    dev_.check(sdso_comm_info(dev_.ctx(), &nranks, nullptr), "sdso_comm_info");
    if (nranks > 1) dev_.check(sdso_ba_allreduce_window(dev_.ctx(), win_), "sdso_ba_allreduce_window");
  }
  void solveSystemF(int iteration, double lambda, std::vector<double>& lastX, std::vector<double>& frame_step, double calib_step[4]) {
    const int n = 8 * nf_ + 4;
    lastX.assign(n, 0); frame_step.assign(nf_ * 8, 0);
    dev_.check(sdso_ba_accumulate(dev_.ctx(), win_), "sdso_ba_accumulate");
    allreduce();
    dev_.check(sdso_ba_solve(dev_.ctx(), win_, iteration, lambda, lastX.data(), nullptr, nullptr, frame_step.data(), calib_step), "sdso_ba_solve");
  }
  // float FullSystem::optimize(int mnumOptIts): writes states / idepths / residual states back into the reference objects
  template <class ApplyFrame, class ApplyPoint, class ApplyResidual>
  float optimize(int mnumOptIts, ApplyFrame apply_frame, ApplyPoint apply_point, ApplyResidual apply_residual) {
    std::vector<double> st(nf_ * 10);
    std::vector<float> idp(points_.size());
    std::vector<uint8_t> rs(residuals_.size());
    sdso_ba_opt_result_t out;
    dev_.check(sdso_ba_optimize(dev_.ctx(), win_, mnumOptIts, st.data(), idp.data(), rs.data(), &out), "sdso_ba_optimize");
    for (int f = 0; f < nf_; f++) apply_frame(f, &st[f * 10]);
    for (size_t p = 0; p < points_.size(); p++) apply_point(points_[p], idp[p]);
    for (size_t r = 0; r < residuals_.size(); r++) apply_residual(residuals_[r], rs[r]);
    lastResult = out;
    return (float)out.rmse;
  }
  // EnergyFunctional::marginalizePointsF(): points with stateFlag == PS_MARGINALIZE; updates ef->HM / ef->bM
  template <class IsMarg>
  void marginalizePointsF(EnergyFunctionalT* ef, IsMarg is_marg) {
    std::vector<uint8_t> flag(points_.size());
    for (size_t p = 0; p < points_.size(); p++) flag[p] = is_marg(points_[p]) ? 1 : 0;
    const int n = 8 * nf_ + 4;
    dev_.check(sdso_ba_marginalize_points(dev_.ctx(), win_, flag.data(), HM_.data(), bM_.data()), "sdso_ba_marginalize_points");
    for (int i = 0; i < n; i++) { ef->bM[i] = bM_[i]; for (int j = 0; j < n; j++) ef->HM(i, j) = HM_[(size_t)i * n + j]; }
  }
  sdso_ba_opt_result_t lastResult{};

 private:
  Device& dev_;
  int win_, nf_ = 0;
  std::vector<double> evalPT_, state_, state_zero_, HM_, bM_;
  std::vector<float> exposure_, energyTH_, u_, v_, idepth_, idepth_zero_, color_, weights_;
  std::vector<int> frameID_, slots_, host_, res_point_, res_target_;
  std::vector<uint8_t> prior_, res_state_;
  std::vector<std::decay_t<decltype(std::declval<EnergyFunctionalT&>().frames[0]->points[0])>> points_;          // EFPoint*
  std::vector<std::decay_t<decltype(std::declval<EnergyFunctionalT&>().frames[0]->points[0]->residualsAll[0])>> residuals_;  // EFResidual*
};

// =================================================================================== ImmaturePoint
// ImmaturePointStatus ImmaturePoint::traceStereo(FrameHessian* frame, Mat33f K, bool mode_right) for a whole
// vector of points (the callers loop over all immature points: FullSystem.cpp:581-613, :667-725).
template <class ImmaturePointT, class Mat33fT>
inline void traceStereoAll(Device& dev, std::vector<ImmaturePointT*>& pts, int frame_slot, const Mat33fT& K, float baseline, bool mode_right,
                           std::vector<uint8_t>& status_out) {
  const int n = (int)pts.size();
  std::vector<float> us(n), vs(n), imin(n), imins(n), imaxs(n), ids(n), col(n * 8), wgt(n * 8), gH(n * 4), eth(n), q(n), uv(n * 2), itv(n);
  std::vector<uint8_t> lts(n);
  for (int i = 0; i < n; i++) {
    const ImmaturePointT* p = pts[i];
    us[i] = p->u_stereo; vs[i] = p->v_stereo; imin[i] = p->idepth_min; imins[i] = p->idepth_min_stereo; imaxs[i] = p->idepth_max_stereo;
    ids[i] = p->idepth_stereo; eth[i] = p->energyTH; q[i] = p->quality; lts[i] = (uint8_t)p->lastTraceStatus;
    uv[2 * i] = p->lastTraceUV[0]; uv[2 * i + 1] = p->lastTraceUV[1]; itv[i] = p->lastTracePixelInterval;
    for (int k = 0; k < 8; k++) { col[i * 8 + k] = p->color[k]; wgt[i * 8 + k] = p->weights[k]; }
    gH[i * 4 + 0] = p->gradH(0, 0); gH[i * 4 + 1] = p->gradH(0, 1); gH[i * 4 + 2] = p->gradH(1, 0); gH[i * 4 + 3] = p->gradH(1, 1);
  }
  sdso_trace_points_t P{n, us.data(), vs.data(), imin.data(), imins.data(), imaxs.data(), ids.data(), col.data(), wgt.data(), gH.data(), eth.data(),
                        q.data(), lts.data(), uv.data(), itv.data()};
  const float K4[4] = {K(0, 0), K(1, 1), K(0, 2), K(1, 2)};
  status_out.assign(n, 0);
  dev.check(sdso_trace_stereo_batch(dev.ctx(), frame_slot, K4, baseline, mode_right ? 1 : 0, &P, status_out.data()), "sdso_trace_stereo_batch");
  for (int i = 0; i < n; i++) {
    ImmaturePointT* p = pts[i];
    p->idepth_min_stereo = imins[i]; p->idepth_max_stereo = imaxs[i]; p->idepth_stereo = ids[i]; p->quality = q[i];
    p->lastTraceStatus = static_cast<decltype(p->lastTraceStatus)>(lts[i]);
    p->lastTraceUV[0] = uv[2 * i]; p->lastTraceUV[1] = uv[2 * i + 1]; p->lastTracePixelInterval = itv[i];
  }
}


// =================================================================================== per-keyframe steps
// PixelSelector (src/FullSystem/PixelSelector2.h): int makeMaps(const FrameHessian* fh, float* map_out, float density,
// int recursionsLeft = 1, bool plot = false, float thFactor = 1); currentPotential is the public member the callers reset.
class PixelSelector {
 public:
  explicit PixelSelector(Device& dev) : dev_(dev) {}
  int currentPotential = 3;
  int makeMaps(int frame_slot, float* map_out, float density, int recursionsLeft = 1, bool /*plot*/ = false, float thFactor = 1) {
    int n = 0;
    dev_.check(sdso_pixel_select(dev_.ctx(), frame_slot, density, recursionsLeft, thFactor, &currentPotential, map_out, &n), "sdso_pixel_select");
    return n;
  }

 private:
  Device& dev_;
};

// ImmaturePoint::traceOn for all immature points of all host keyframes in the newest frame (FullSystem::traceNewCoarseKey / NonKey,
// FullSystem.cpp:632-790).  geom[h] = {KRKi, Kt, aff} of host h as computed at :654-665; host_of[i] selects it.
template <class ImmaturePointT>
inline void traceOnAll(Device& dev, std::vector<ImmaturePointT*>& pts, const std::vector<int>& host_of, const std::vector<sdso_trace_geom_t>& geom,
                       int frame_slot, std::vector<uint8_t>& status_out) {
  const int n = (int)pts.size();
  std::vector<float> us(n), vs(n), imin(n), imax(n), col(n * 8), wgt(n * 8), gH(n * 4), eth(n), q(n), uv(n * 2), itv(n);
  std::vector<uint8_t> lts(n);
  for (int i = 0; i < n; i++) {
    const ImmaturePointT* p = pts[i];
    us[i] = p->u; vs[i] = p->v; imin[i] = p->idepth_min; imax[i] = p->idepth_max; eth[i] = p->energyTH; q[i] = p->quality;
    lts[i] = (uint8_t)p->lastTraceStatus; uv[2 * i] = p->lastTraceUV[0]; uv[2 * i + 1] = p->lastTraceUV[1]; itv[i] = p->lastTracePixelInterval;
    for (int k = 0; k < 8; k++) { col[i * 8 + k] = p->color[k]; wgt[i * 8 + k] = p->weights[k]; }
    gH[i * 4 + 0] = p->gradH(0, 0); gH[i * 4 + 1] = p->gradH(0, 1); gH[i * 4 + 2] = p->gradH(1, 0); gH[i * 4 + 3] = p->gradH(1, 1);
  }
  sdso_trace_points_t P{n, us.data(), vs.data(), nullptr, imin.data(), imax.data(), nullptr, col.data(), wgt.data(), gH.data(), eth.data(),
                        q.data(), lts.data(), uv.data(), itv.data()};
  status_out.assign(n, 0);
  dev.check(sdso_trace_on_batch(dev.ctx(), frame_slot, (int)geom.size(), geom.data(), host_of.data(), &P, status_out.data()), "sdso_trace_on_batch");
  for (int i = 0; i < n; i++) {
    ImmaturePointT* p = pts[i];
    p->idepth_min = imin[i]; p->idepth_max = imax[i]; p->quality = q[i];
    p->lastTraceStatus = static_cast<decltype(p->lastTraceStatus)>(lts[i]);
    p->lastTraceUV[0] = uv[2 * i]; p->lastTraceUV[1] = uv[2 * i + 1]; p->lastTracePixelInterval = itv[i];
  }
}

// EnergyFunctional::marginalizeFrame's algebra (EnergyFunctional.cpp:554-660) on plain row-major arrays
inline void marginalizeFrame(int nFrames, int idx, const double* prior8, const double* delta_prior8, std::vector<double>& HM, std::vector<double>& bM) {
  const int m = 8 * (nFrames - 1) + 4;
  std::vector<double> Ho((size_t)m * m), bo(m);
  if (sdso_ba_marginalize_frame(nFrames, idx, prior8, delta_prior8, HM.data(), bM.data(), Ho.data(), bo.data()) != SDSO_OK)
    throw Error("sdso_ba_marginalize_frame: bad arguments");
  HM.swap(Ho); bM.swap(bo);
}

}  // namespace sdso_shim
