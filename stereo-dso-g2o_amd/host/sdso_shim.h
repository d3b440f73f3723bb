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
//   void CoarseTracker::setCoarseTrackingRef(std::vector<FrameHessian*>, FrameHessian* fh_right, CalibHessian)        FullSystem/CoarseTracker.h:71-72
//   void CoarseTracker::setCTRefForFirstFrame(std::vector<FrameHessian*>)                                           FullSystem/CoarseTracker.cpp:794-805
//   double PointFrameResidual::linearize(CalibHessian*) / void applyRes(bool)                                        FullSystem/Residuals.h:103, :113
//   AccumulatedTopHessianSSE::{setZero, addPoint<mode>, addPointsInternal<mode>, stitchDouble, stitchDoubleMT}        OptimizationBackend/AccumulatedTopHessian.h:66-97, :162-169
//   AccumulatedSCHessianSSE::{setZero, addPoint, addPointsInternal, stitchDouble, stitchDoubleMT}                     OptimizationBackend/AccumulatedSCHessian.h:66-96, :155-160
//   EnergyFunctional::{calcLEnergyF_MT, calcMEnergyF, setDeltaF, setAdjointsF}                                       OptimizationBackend/EnergyFunctional.h:75-86
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>
#include <utility>
#include <functional>
#include <stdexcept>
#include <string>
#include <unordered_map>
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
  // void setCoarseTrackingRef(std::vector<FrameHessian*> frameHessians, FrameHessian* fh_right, CalibHessian Hcalib) — CoarseTracker.h:71-72,
  // CoarseTracker.cpp:807-826 with makeCoarseDepthL0 (:275-534) entirely on the device.  STEP1 (:288-356): every point whose residual into
  // the newest keyframe is IN is re-observed by static stereo — ImmaturePoint at the rounded centerProjectedTo on fh_target, traceStereo
  // into fh_right with the interval [0.1, 1.9] * centerProjectedTo[2], and where that is IPS_GOOD back again from lastTraceUV —
  // (sdso_stereo_match_batch: ONE launch chain for all points), the accept rule of :329-341 picks idepth_stereo or centerProjectedTo[2],
  // weight = sqrtf(1e-3 / (HdiF + 1e-12)); STEP2-5 = sdso_track_make_ref.  `slot_of` maps a FrameHessian to the pyramid slot it was
  // uploaded to; `baseline` is the global of util/settings.h that traceStereo reads (ImmaturePoint.cpp:101).
  std::function<int(const void*)> slot_of;
  float baseline = 0.f;
  int pc_n[SDSO_PYR_LEVELS] = {0, 0, 0, 0, 0, 0};
  template <class FrameHessianT, class CalibHessianT>
  void setCoarseTrackingRef(std::vector<FrameHessianT*> frameHessians, FrameHessianT* fh_right, CalibHessianT Hcalib) {
    if (frameHessians.empty() || !slot_of) throw Error("setCoarseTrackingRef: no frames / slot_of not set");
    FrameHessianT* lastRef = frameHessians.back();           // (= fh_target of makeCoarseDepthL0)
    std::vector<float> fu, fv, imin, imax, cpt2, wgt;
    for (FrameHessianT* fh : frameHessians)
      for (auto* ph : fh->pointHessians) {
        if (ph->lastResiduals[0].first != 0 && (int)ph->lastResiduals[0].second == 0 /* ResState::IN */) {
          auto* r = ph->lastResiduals[0].first;
          const int u = r->centerProjectedTo[0] + 0.5f;      // :303-304
          const int v = r->centerProjectedTo[1] + 0.5f;
          fu.push_back((float)u); fv.push_back((float)v);
          imin.push_back(r->centerProjectedTo[2] * 0.1f);     // :311-312
          imax.push_back(r->centerProjectedTo[2] * 1.9f);
          cpt2.push_back(r->centerProjectedTo[2]);
          wgt.push_back(sqrtf(1e-3 / (ph->efPoint->HdiF + 1e-12)));   // :350
        }
      }
    const int n = (int)fu.size();
    std::vector<uint8_t> sf(n), sb(n);
    std::vector<float> ids(n), buv((size_t)n * 2);
    if (n) {
      sdso_stereo_match_t m;
      std::memset(&m, 0, sizeof(m));
      m.n = n; m.u = fu.data(); m.v = fv.data();
      m.idepth_min_stereo = imin.data(); m.idepth_max_stereo = imax.data();
      m.back_idepth_min_stereo = imin.data(); m.back_idepth_max_stereo = imax.data();   // :323-324
      m.status_fwd = sf.data(); m.status_back = sb.data(); m.idepth_stereo = ids.data(); m.back_uv = buv.data();
      const float K4[4] = {Hcalib.fxl(), Hcalib.fyl(), Hcalib.cxl(), Hcalib.cyl()};
      dev_.check(sdso_stereo_match_batch(dev_.ctx(), slot_of(lastRef), slot_of(fh_right), K4, baseline, 1, &m), "sdso_stereo_match_batch");
    }
    std::vector<int> iu(n), iv(n);
    std::vector<float> nid(n);
    for (int i = 0; i < n; i++) {
      iu[i] = (int)fu[i]; iv[i] = (int)fv[i];
      float new_idepth = cpt2[i];
      if (sf[i] == 0 /* IPS_GOOD */) {
        const float depth = 1.0f / ids[i];
        const float u_delta = std::fabs(fu[i] - buv[(size_t)i * 2]);
        if (u_delta < 1 && depth > 0 && depth < 50) new_idepth = ids[i];            // :329-332
      }
      nid[i] = new_idepth;
    }
    installRef_(lastRef, n, iu.data(), iv.data(), nid.data(), wgt.data());
  }
  // void setCTRefForFirstFrame(std::vector<FrameHessian*> frameHessians) — CoarseTracker.cpp:794-805 with makeCoarseDepthForFirstFrame
  // (:138-271): the first keyframe's own points at int(u + 0.5f), their idepth, no stereo refinement; STEP2-5 are makeCoarseDepthL0's.
  template <class FrameHessianT>
  void setCTRefForFirstFrame(std::vector<FrameHessianT*> frameHessians) {
    if (frameHessians.empty() || !slot_of) throw Error("setCTRefForFirstFrame: no frames / slot_of not set");
    FrameHessianT* lastRef = frameHessians.back();
    std::vector<int> iu, iv;
    std::vector<float> nid, wgt;
    for (auto* ph : lastRef->pointHessians) {
      const int u = ph->u + 0.5f;                            // :144-145
      const int v = ph->v + 0.5f;
      iu.push_back(u); iv.push_back(v); nid.push_back(ph->idepth);
      wgt.push_back(sqrtf(1e-3 / (ph->efPoint->HdiF + 1e-12)));
    }
    installRef_(lastRef, (int)iu.size(), iu.data(), iv.data(), nid.data(), wgt.data());
  }
  // bool trackNewestCoarse(FrameHessian* newFrameHessian, SE3& lastToNew_out, AffLight& aff_g2l_out, int coarsestLvl, Vec5 minResForAbort,
  //                        IOWrap::Output3DWrapper* wrap = 0) — CoarseTracker.h:62-66 verbatim (the debug wrapper is not used)
  template <class FrameHessianT, class Vec5T>
  bool trackNewestCoarse(FrameHessianT* newFrameHessian, SE3T& lastToNew_out, AffLightT& aff_g2l_out, int coarsestLvl, Vec5T minResForAbort,
                         void* /*wrap*/ = nullptr) {
    if (!slot_of) throw Error("trackNewestCoarse: slot_of not set");
    return trackNewestCoarse(slot_of(newFrameHessian), newFrameHessian->ab_exposure, lastToNew_out, aff_g2l_out, coarsestLvl, minResForAbort);
  }
  // bool trackNewestCoarse(...) with the frame given by its pyramid slot
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
  AffLightT lastRef_aff_g2l{};

 private:
  template <class FrameHessianT>
  void installRef_(FrameHessianT* lastRef, int n, const int* u, const int* v, const float* new_idepth, const float* weight) {
    dev_.check(sdso_track_make_ref(dev_.ctx(), ref_slot_, slot_of(lastRef), n, u, v, new_idepth, weight, pc_n), "sdso_track_make_ref");
    refFrameID = lastRef->shell->id;                         // :821-825
    lastRef_aff_g2l = lastRef->aff_g2l();
    prm_.ref_exposure = lastRef->ab_exposure;
    prm_.ref_aff_g2l.a = lastRef_aff_g2l.a; prm_.ref_aff_g2l.b = lastRef_aff_g2l.b;
    firstCoarseRMSE = -1;
  }
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
    maxRelBaseline_.clear(); numGood_.clear(); isNew_.clear();
    for (int f = 0; f < nf; f++)
      for (auto* p : ef->frames[f]->points) {
        auto* ph = p->data;
        const int pi = (int)u_.size();
        u_.push_back(ph->u); v_.push_back(ph->v); idepth_.push_back(ph->idepth); idepth_zero_.push_back(ph->idepth_zero);
        for (int k = 0; k < 8; k++) { color_.push_back(ph->color[k]); weights_.push_back(ph->weights[k]); }
        host_.push_back(f); prior_.push_back(ph->hasDepthPrior ? 1 : 0);
        maxRelBaseline_.push_back(ph->maxRelBaseline); numGood_.push_back(ph->numGoodResiduals);
        points_.push_back(p);
        for (auto* r : p->residualsAll) {                  // residualsAll order: the order of the reference's per-point float sums
          res_point_.push_back(pi); res_target_.push_back(r->target->idx); res_state_.push_back((uint8_t)r->data->state_state);
          isNew_.push_back(r->data->isNew ? 1 : 0);
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
    W.maxRelBaseline = maxRelBaseline_.data(); W.numGoodResiduals = numGood_.data(); W.res_isNew = isNew_.data();
    const int n = 8 * nf + 4;
    HM_.assign((size_t)n * n, 0); bM_.assign(n, 0);
    for (int i = 0; i < n; i++) { bM_[i] = ef->bM[i]; for (int j = 0; j < n; j++) HM_[(size_t)i * n + j] = ef->HM(i, j); }
    W.HM = HM_.data(); W.bM = bM_.data();
    W.solverMode = solverMode; W.affineOptModeA = affineOptModeA; W.affineOptModeB = affineOptModeB; W.forceAcceptStep = forceAcceptStep ? 1 : 0;
    dev_.check(sdso_ba_upload_window(dev_.ctx(), win_, &W), "sdso_ba_upload_window");
    nf_ = nf; resInM_seen_ = 0;
    ef_ = ef;
    res_index_.clear(); point_index_.clear();
    for (size_t i = 0; i < residuals_.size(); i++) res_index_[residuals_[i]->data] = (int)i;
    for (size_t i = 0; i < points_.size(); i++) point_index_[points_[i]] = (int)i;
    lin_valid_ = app_valid_ = acc_valid_ = marg_valid_ = false;
  }

  // Vec3 FullSystem::linearizeAll(false): returns lastEnergyP
  double linearizeAll() {
    double e = 0;
    dev_.check(sdso_ba_linearize(dev_.ctx(), win_, &e), "sdso_ba_linearize");
    lin_valid_ = app_valid_ = acc_valid_ = marg_valid_ = false;
    return e;
  }
  // applyRes_Reductor(true, ...)
  void applyRes() { dev_.check(sdso_ba_apply_res(dev_.ctx(), win_), "sdso_ba_apply_res"); app_valid_ = acc_valid_ = marg_valid_ = false; }

  // double PointFrameResidual::linearize(CalibHessian* HCalib) (Residuals.h:103, Residuals.cpp:83-336) and void applyRes(bool copyJacobians)
  // (Residuals.h:113, Residuals.cpp:367-385) PER OBJECT, the way linearizeAll_Reductor / applyRes_Reductor call them
  // (FullSystemOptimize.cpp:52-96): the device linearises / applies the whole window at the first call after a change and every call
  // hands its own residual's results out — state_NewEnergy, state_NewEnergyWithOutlier, state_NewState; then state_state, state_energy,
  // EFResidual::isActiveAndIsGoodNEW.  A linearised residual (EFResidual::isLinearized) is not touched — the reference's loops run over
  // activeResiduals, which holds none (FullSystemOptimize.cpp:880-889) —: both members return before any write.
  template <class PointFrameResidualT>
  double linearize(PointFrameResidualT* r, CalibHessianT* /*HCalib*/) {
    if (r->efResidual->isLinearized) return 0.0;
    if (!lin_valid_) {
      const int nr = (int)residuals_.size();
      linearizeAll();
      l_state_.assign(nr, 0); l_energy_.assign(nr, 0.f); l_energyWO_.assign(nr, 0.f);
      dev_.check(sdso_ba_get_linearization(dev_.ctx(), win_, nullptr, l_state_.data(), l_energy_.data(), l_energyWO_.data(), nullptr, nullptr), "sdso_ba_get_linearization");
      lin_valid_ = true;
    }
    const int i = index_of_(r);
    using ResStateT = std::decay_t<decltype(r->state_NewState)>;
    r->state_NewState = static_cast<ResStateT>(l_state_[i]);
    r->state_NewEnergyWithOutlier = l_energyWO_[i];
    // (an OOB residual returns its old state_energy without touching state_NewEnergy: Residuals.cpp:88-91, :226)
    if (l_state_[i] == 1) return r->state_energy;
    r->state_NewEnergy = l_energy_[i];
    return r->state_NewEnergy;
  }
  template <class PointFrameResidualT>
  void applyRes(PointFrameResidualT* r, bool copyJacobians) {
    if (r->efResidual->isLinearized) return;
    if (!copyJacobians) {   // Residuals.cpp:382-384 alone: setState(state_NewState); state_energy = state_NewEnergy — no OOB test, isActiveAndIsGoodNEW untouched
      r->state_state = r->state_NewState;
      r->state_energy = r->state_NewEnergy;
      return;
    }
    if (!app_valid_) {
      const int nr = (int)residuals_.size();
      applyRes();
      a_state_.assign(nr, 0); a_act_.assign(nr, 0);
      dev_.check(sdso_ba_get_residual_state(dev_.ctx(), win_, a_state_.data(), a_act_.data(), nullptr), "sdso_ba_get_residual_state");
      app_valid_ = true;
    }
    const int i = index_of_(r);
    using ResStateT = std::decay_t<decltype(r->state_state)>;
    if ((int)r->state_state == 1) return;                     // `if(state_state == ResState::OOB) return;` — can never go back from OOB
    r->state_state = static_cast<ResStateT>(a_state_[i]);
    r->state_energy = r->state_NewEnergy;
    r->efResidual->isActiveAndIsGoodNEW = a_act_[i] != 0;
  }
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
    ensureAccumulated();
    dev_.check(sdso_ba_solve(dev_.ctx(), win_, iteration, lambda, lastX.data(), nullptr, nullptr, frame_step.data(), calib_step), "sdso_ba_solve");
  }
  // void EnergyFunctional::solveSystemF(int iteration, double lambda, CalibHessian* HCalib) — EnergyFunctional.h:74, EnergyFunctional.cpp:838-995
  // with resubstituteF_MT (:272-341) — verbatim: everything the reference's function leaves in its objects is written into them:
  //   ef->lastX, lastHS, lastbS (:909-910, :992); ef->resInA / resInL (:219, :241); HCalib->step = -x.head<CPARS>() (:279);
  //   every EFFrame::data->step.head<8>() = -x.segment<8>(CPARS + 8 idx), tail<2>() = 0 (:283-286);
  //   every PointHessian::step (:336-338), EFPoint::HdiF / bdSumF (AccumulatedSCHessian.cpp:58-65)
  void solveSystemF(int iteration, double lambda, CalibHessianT* HCalib) {
    const int n = 8 * nf_ + 4, np = (int)points_.size();
    std::vector<double> x(n), HS((size_t)n * n), bS(n), fstep(nf_ * 8);
    double cstep[4];
    ensureAccumulated();
    dev_.check(sdso_ba_solve(dev_.ctx(), win_, iteration, lambda, x.data(), HS.data(), bS.data(), fstep.data(), cstep), "sdso_ba_solve");
    ef_->lastX.resize(n); ef_->lastbS.resize(n); ef_->lastHS.resize(n, n);
    for (int i = 0; i < n; i++) { ef_->lastX[i] = x[i]; ef_->lastbS[i] = bS[i]; for (int j = 0; j < n; j++) ef_->lastHS(i, j) = HS[(size_t)i * n + j]; }
    for (int i = 0; i < 4; i++) HCalib->step[i] = cstep[i];
    for (int f = 0; f < nf_; f++) {
      auto* fh = ef_->frames[f]->data;
      for (int i = 0; i < 8; i++) fh->step[i] = fstep[f * 8 + i];
      fh->step[8] = 0; fh->step[9] = 0;
    }
    std::vector<float> pstep(np), hdi(np), bds(np);
    if (np) {
      dev_.check(sdso_ba_get_point_steps(dev_.ctx(), win_, pstep.data()), "sdso_ba_get_point_steps");
      dev_.check(sdso_ba_get_point_terms(dev_.ctx(), win_, hdi.data(), bds.data(), nullptr, nullptr, nullptr), "sdso_ba_get_point_terms");
    }
    for (int p = 0; p < np; p++) { points_[p]->data->step = pstep[p]; points_[p]->HdiF = hdi[p]; points_[p]->bdSumF = bds[p]; }
    int ra = 0, rl = 0;
    dev_.check(sdso_ba_get_counts(dev_.ctx(), win_, &ra, &rl, nullptr), "sdso_ba_get_counts");
    ef_->resInA = ra; ef_->resInL = rl;
  }
  // double EnergyFunctional::calcLEnergyF_MT() / double calcMEnergyF() — EnergyFunctional.h:82-83, EnergyFunctional.cpp:420-442, :344-351
  double calcLEnergyF_MT() { double el = 0; dev_.check(sdso_ba_calc_energies(dev_.ctx(), win_, &el, nullptr), "sdso_ba_calc_energies"); return el; }
  double calcMEnergyF() { double em = 0; dev_.check(sdso_ba_calc_energies(dev_.ctx(), win_, nullptr, &em), "sdso_ba_calc_energies"); return em; }
  // void EnergyFunctional::setAdjointsF(CalibHessian* Hcalib) — EnergyFunctional.h:86, EnergyFunctional.cpp:41-119: the window's adjoints
  // (computed at upload / whenever the device loop moved the states) into ef->adHost / adTarget / adHostF / adTargetF, [h + t * nFrames]
  void setAdjointsF(CalibHessianT* /*Hcalib*/) {
    const int nf = nf_;
    std::vector<double> aH((size_t)nf * nf * 64), aT((size_t)nf * nf * 64);
    dev_.check(sdso_ba_get_tables(dev_.ctx(), win_, nullptr, aH.data(), aT.data(), nullptr), "sdso_ba_get_tables");
    using M88 = std::remove_pointer_t<decltype(ef_->adHost)>;
    using M88f = std::remove_pointer_t<decltype(ef_->adHostF)>;
    if (ef_->adHost != 0) delete[] ef_->adHost;
    if (ef_->adTarget != 0) delete[] ef_->adTarget;
    if (ef_->adHostF != 0) delete[] ef_->adHostF;
    if (ef_->adTargetF != 0) delete[] ef_->adTargetF;
    ef_->adHost = new M88[nf * nf]; ef_->adTarget = new M88[nf * nf]; ef_->adHostF = new M88f[nf * nf]; ef_->adTargetF = new M88f[nf * nf];
    for (int k = 0; k < nf * nf; k++)
      for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) {
          ef_->adHost[k](i, j) = aH[(size_t)k * 64 + i * 8 + j]; ef_->adTarget[k](i, j) = aT[(size_t)k * 64 + i * 8 + j];
          ef_->adHostF[k](i, j) = (float)aH[(size_t)k * 64 + i * 8 + j]; ef_->adTargetF[k](i, j) = (float)aT[(size_t)k * 64 + i * 8 + j];
        }
  }
  // void EnergyFunctional::setDeltaF(CalibHessian* HCalib) — EnergyFunctional.h:75, EnergyFunctional.cpp:173-207: adHTdeltaF[h + t * nFrames],
  // cDeltaF, EFFrame::delta / delta_prior, EFPoint::deltaF
  void setDeltaF(CalibHessianT* /*HCalib*/) {
    const int nf = nf_, np = (int)points_.size();
    std::vector<float> adhtd((size_t)nf * nf * 8), pd(np);
    std::vector<double> fd(nf * 8), fdp(nf * 8);
    float cd[4];
    dev_.check(sdso_ba_get_tables(dev_.ctx(), win_, nullptr, nullptr, nullptr, adhtd.data()), "sdso_ba_get_tables");
    dev_.check(sdso_ba_get_deltas(dev_.ctx(), win_, cd, fd.data(), fdp.data(), pd.data()), "sdso_ba_get_deltas");
    using M18f = std::remove_pointer_t<decltype(ef_->adHTdeltaF)>;
    if (ef_->adHTdeltaF != 0) delete[] ef_->adHTdeltaF;
    ef_->adHTdeltaF = new M18f[nf * nf];
    for (int k = 0; k < nf * nf; k++) for (int j = 0; j < 8; j++) ef_->adHTdeltaF[k](0, j) = adhtd[(size_t)k * 8 + j];
    for (int i = 0; i < 4; i++) ef_->cDeltaF[i] = cd[i];
    for (int f = 0; f < nf; f++) for (int i = 0; i < 8; i++) { ef_->frames[f]->delta[i] = fd[f * 8 + i]; ef_->frames[f]->delta_prior[i] = fdp[f * 8 + i]; }
    for (int p = 0; p < np; p++) points_[p]->deltaF = pd[p];
  }
  // the device pass behind accumulateAF_MT + accumulateLF_MT + accumulateSCF_MT: runs once per linearised / applied state
  void ensureAccumulated() {
    if (acc_valid_) return;
    dev_.check(sdso_ba_accumulate(dev_.ctx(), win_), "sdso_ba_accumulate");
    allreduce();
    acc_valid_ = true; marg_valid_ = false;
  }
  // marginalizePointsF's device pass for the points the accumulator façades collected (addPoint<2>): see AccumulatedTopHessianSSE below
  // (type-erased form for the accumulator façade, which holds the points as `const void*`)
  void ensureMarginalizedErased(const std::vector<const void*>& flagged) {
    if (marg_valid_) return;
    std::vector<uint8_t> flag(points_.size(), 0);
    for (const void* p : flagged) flag[point_index_.at(p)] = 1;
    const int before = countOf(2);
    dev_.check(sdso_ba_marginalize_points(dev_.ctx(), win_, flag.data(), HM_.data(), bM_.data()), "sdso_ba_marginalize_points");
    last_marg_res_ = countOf(2) - before;
    marg_valid_ = true; acc_valid_ = lin_valid_ = app_valid_ = false;
  }
  void requireMarginalized() const { if (!marg_valid_) throw Error("AccumulatedSCHessianSSE: the marginalisation pass has not run (stitch the top accumulator first, EnergyFunctional.cpp:707-708)"); }
  int lastMargResiduals() const { return last_marg_res_; }
  const std::vector<double>& HM() const { return HM_; }
  const std::vector<double>& bM() const { return bM_; }
  template <class MatXX, class VecX> void stitchedSystem(int which, MatXX& H, VecX& b) { stitched_(which, H, b); }
  int countOf(int which) {                                    // 0 resInA, 1 resInL, 2 residuals marginalised through this window so far
    int c[3] = {0, 0, 0};
    dev_.check(sdso_ba_get_counts(dev_.ctx(), win_, &c[0], &c[1], &c[2]), "sdso_ba_get_counts");
    return c[which];
  }
  int nFrames() const { return nf_; }
  // void EnergyFunctional::accumulateAF_MT(MatXX& H, VecX& b, bool MT) / accumulateLF_MT / accumulateSCF_MT (EnergyFunctional.cpp:212-269,
  // EnergyFunctional.h:103-105): what the reference's callers get back is the STITCHED system of AccumulatedTopHessianSSE::stitchDoubleMT
  // (mode 0 without priors, mode 1 with priors) / AccumulatedSCHessianSSE::stitchDoubleMT.  The device accumulates all three in one pass
  // (sdso_ba_accumulate) and its solver never materialises them; these members run the stitch kernels on demand.  MatXX / VecX: anything
  // with resize(rows[, cols]) and operator()(i[, j]) — Eigen's, or the stand-ins of host/test_shim.cpp.  `MT` is ignored.
  void accumulateAll() { acc_valid_ = false; ensureAccumulated(); }
  template <class MatXX, class VecX> void accumulateAF_MT(MatXX& H, VecX& b, bool /*MT*/) { stitched_(0, H, b); }
  template <class MatXX, class VecX> void accumulateLF_MT(MatXX& H, VecX& b, bool /*MT*/) { stitched_(1, H, b); }
  template <class MatXX, class VecX> void accumulateSCF_MT(MatXX& H, VecX& b, bool /*MT*/) { stitched_(2, H, b); }
  // float FullSystem::optimize(int mnumOptIts) — FullSystemOptimize.cpp:871-1041 from `activeResiduals.clear()` (:880) through the closing
  // `linearizeAll(true)` (:1008) — on the uploaded window, with EVERYTHING that function leaves behind written back into the reference's
  // own objects (sdso_ba_get_post_state):
  //   CalibHessian      setValue (value, value_scaled, ...), step                                   :218-222 via doStepFromBackup
  //   FrameHessian      setState (state, state_scaled, PRE_worldToCam, PRE_camToWorld), step; the newest frame's setEvalPT (:997-1003);
  //                     its frameEnergyTH (setNewFrameEnergyTH of the closing linearizeAll)
  //   PointHessian      setIdepth + setIdepthZero (:268-272), step, idepth_hessian, maxRelBaseline, numGoodResiduals (:64-77,
  //                     AccumulatedSCHessian.cpp:44-58); EFPoint::HdiF, bdSumF
  //   PointFrameResidual state_state / state_NewState, state_energy, centerProjectedTo, projectedTo; EFResidual::isActiveAndIsGoodNEW
  //   PointHessian::lastResiduals[k].second (:165-172); for every residual on toRemove: lastResiduals[k].first = 0,
  //                     ef->dropResidual(r->efResidual), deleteOut(ph->residuals, k) (:176-195) — in activeResiduals order
  //   EnergyFunctional  lastX, lastHS, lastbS, resInA, resInL
  // Left to the caller exactly as in the reference: `ef->setAdjointsF(&Hcalib); setPrecalcValues();` (:1005-1007, host tables other host
  // code reads), the isLost test, statistics_lastFineTrackRMSE and the shell poses (:1010-1037).  Not written back: the
  // RawResidualJacobian records (EFResidual::J) — only the accumulators this library replaces read them.
  // Returns sqrtf(lastEnergy[0] / (patternNum * ef->resInA)) like the reference (:1039); lastResult carries lastEnergy[0].
  float optimize(int mnumOptIts, EnergyFunctionalT* ef, CalibHessianT* HCalib) {
    const int nf = nf_, np = (int)points_.size(), nr = (int)residuals_.size(), n = 8 * nf + 4;
    sdso_ba_opt_result_t out;
    lin_valid_ = app_valid_ = acc_valid_ = marg_valid_ = false;
    if (nf < 2) { lastResult = sdso_ba_opt_result_t{0, 0, 0, 0}; lastRemoved = 0; return 0.f; }   // `if(frameHessians.size() < 2) return 0;` (:873-874): nothing is touched
    dev_.check(sdso_ba_optimize(dev_.ctx(), win_, mnumOptIts, nullptr, nullptr, nullptr, &out), "sdso_ba_optimize");
    std::vector<float> idp(np), pstep(np), hdi(np), bds(np), idh(np), mrb(np), energy(nr), cpt((size_t)nr * 3), prj((size_t)nr * 16), eth(nf);
    std::vector<int> ngood(np);
    std::vector<uint8_t> rs(nr), act(nr), rem(nr);
    std::vector<double> st(nf * 10), stz(nf * 10), ev(nf * 12), fstep(nf * 10), lx(n), lhs((size_t)n * n), lbs(n);
    sdso_ba_post_state_t P;
    std::memset(&P, 0, sizeof(P));
    P.idepth = idp.data(); P.step = pstep.data(); P.HdiF = hdi.data(); P.bdSumF = bds.data(); P.idepth_hessian = idh.data();
    P.maxRelBaseline = mrb.data(); P.numGoodResiduals = ngood.data();
    P.state_state = rs.data(); P.isActiveAndIsGoodNEW = act.data(); P.state_energy = energy.data(); P.centerProjectedTo = cpt.data();
    P.projectedTo = prj.data(); P.toRemove = rem.data();
    P.state = st.data(); P.state_zero = stz.data(); P.evalPT = ev.data(); P.frame_step = fstep.data(); P.frameEnergyTH = eth.data();
    P.lastX = lx.data(); P.lastHS = lhs.data(); P.lastbS = lbs.data();
    dev_.check(sdso_ba_get_post_state(dev_.ctx(), win_, &P), "sdso_ba_get_post_state");
    // ---- calibration and frames
    {
      auto v = HCalib->value_zero;                         // a VecC to fill
      for (int i = 0; i < 4; i++) v[i] = P.calib_value[i];
      HCalib->setValue(v);
      for (int i = 0; i < 4; i++) HCalib->step[i] = P.calib_step[i];
    }
    for (int f = 0; f < nf; f++) {
      auto* fh = ef->frames[f]->data;
      auto s = fh->get_state();                            // a Vec10 to fill
      for (int i = 0; i < 10; i++) s[i] = st[f * 10 + i];
      if (f == nf - 1) {
        // setEvalPT(PRE_worldToCam at the loop's final state, {0,..,0, a, b, 0, 0}) (:997-1003)
        sdso_se3_t T;
        std::memcpy(T.R, &ev[f * 12], 72); std::memcpy(T.t, &ev[f * 12 + 9], 24);
        fh->setEvalPT(like(fh->get_worldToCam_evalPT(), T), s);
      } else fh->setState(s);
      for (int i = 0; i < 10; i++) fh->step[i] = fstep[f * 10 + i];
      fh->frameEnergyTH = eth[f];
    }
    // ---- points
    for (int p = 0; p < np; p++) {
      auto* efp = points_[p];
      auto* ph = efp->data;
      ph->setIdepth(idp[p]); ph->setIdepthZero(idp[p]);
      ph->step = pstep[p];
      ph->idepth_hessian = idh[p]; ph->maxRelBaseline = mrb[p]; ph->numGoodResiduals = ngood[p];
      efp->HdiF = hdi[p]; efp->bdSumF = bds[p];
    }
    // ---- residuals
    for (int i = 0; i < nr; i++) {
      auto* r = residuals_[i];
      auto* pfr = r->data;
      if (r->isLinearized) continue;                       // not in activeResiduals (:880-889): optimize() never touches it
      using ResStateT = std::decay_t<decltype(pfr->state_state)>;
      pfr->state_state = static_cast<ResStateT>(rs[i]);
      pfr->state_NewState = static_cast<ResStateT>(rs[i]);
      pfr->state_energy = energy[i]; pfr->state_NewEnergy = energy[i];
      r->isActiveAndIsGoodNEW = act[i] != 0;
      if (act[i]) {
        for (int k = 0; k < 3; k++) pfr->centerProjectedTo[k] = cpt[(size_t)i * 3 + k];
        for (int k = 0; k < 8; k++) { pfr->projectedTo[k][0] = prj[(size_t)i * 16 + 2 * k]; pfr->projectedTo[k][1] = prj[(size_t)i * 16 + 2 * k + 1]; }
      }
      auto* ph = pfr->point;                               // :165-172
      if (ph->lastResiduals[0].first == pfr) ph->lastResiduals[0].second = pfr->state_state;
      else if (ph->lastResiduals[1].first == pfr) ph->lastResiduals[1].second = pfr->state_state;
    }
    int nResRemoved = 0;
    for (int i = 0; i < nr; i++) {                         // :176-195
      if (!rem[i] || residuals_[i]->isLinearized) continue;
      auto* pfr = residuals_[i]->data;
      auto* ph = pfr->point;
      if (ph->lastResiduals[0].first == pfr) ph->lastResiduals[0].first = 0;
      else if (ph->lastResiduals[1].first == pfr) ph->lastResiduals[1].first = 0;
      for (unsigned int k = 0; k < ph->residuals.size(); k++)
        if (ph->residuals[k] == pfr) {
          ef->dropResidual(pfr->efResidual);
          delete ph->residuals[k];                         // deleteOut<PointFrameResidual>(ph->residuals, k), FullSystem.h:62-71
          ph->residuals[k] = ph->residuals.back();
          ph->residuals.pop_back();
          nResRemoved++;
          break;
        }
    }
    residuals_.clear();                                    // (dropResidual deleted some of them: re-upload before the next call)
    lastRemoved = nResRemoved;
    // ---- EnergyFunctional
    ef->lastX.resize(n); ef->lastbS.resize(n); ef->lastHS.resize(n, n);
    for (int i = 0; i < n; i++) { ef->lastX[i] = lx[i]; ef->lastbS[i] = lbs[i]; for (int j = 0; j < n; j++) ef->lastHS(i, j) = lhs[(size_t)i * n + j]; }
    ef->resInA = P.resInA; ef->resInL = P.resInL;
    lastResult = out;
    return (float)out.rmse;
  }
  int lastRemoved = 0;
  // EnergyFunctional::marginalizePointsF() (EnergyFunctional.cpp:663-736): points with stateFlag == PS_MARGINALIZE; updates ef->HM / ef->bM /
  // ef->resInM.  The re-linearisation + fixLinearizationF that FullSystem::flagPointsForRemoval runs on those points beforehand
  // (FullSystem.cpp:1012-1021) is part of the call.  Call upload() first: optimize() dropped residuals and the reference drops points in between
  // (removeOutliers, flagPointsForRemoval + dropPointsF), so the window of the optimize call no longer matches the EnergyFunctional.
  // The caller goes on with the reference's removePoint loop (:730-735).
  template <class IsMarg>
  void marginalizePointsF(EnergyFunctionalT* ef, IsMarg is_marg) {
    std::vector<uint8_t> flag(points_.size());
    for (size_t p = 0; p < points_.size(); p++) flag[p] = is_marg(points_[p]) ? 1 : 0;
    const int n = 8 * nf_ + 4;
    dev_.check(sdso_ba_marginalize_points(dev_.ctx(), win_, flag.data(), HM_.data(), bM_.data()), "sdso_ba_marginalize_points");
    marg_valid_ = true; acc_valid_ = lin_valid_ = app_valid_ = false;
    for (int i = 0; i < n; i++) { ef->bM[i] = bM_[i]; for (int j = 0; j < n; j++) ef->HM(i, j) = HM_[(size_t)i * n + j]; }
    int resInM = 0;                                        // resInM += accSSE_top_A->nres[0] (EnergyFunctional.cpp:704)
    dev_.check(sdso_ba_get_counts(dev_.ctx(), win_, nullptr, nullptr, &resInM), "sdso_ba_get_counts");
    ef->resInM += resInM - resInM_seen_; resInM_seen_ = resInM;
  }
  sdso_ba_opt_result_t lastResult{};

 private:
  template <class MatXX, class VecX> void stitched_(int which, MatXX& H, VecX& b) {
    const int n = 8 * nf_ + 4;
    std::vector<double> Hs((size_t)n * n), bs(n);
    double* Hp[3] = {nullptr, nullptr, nullptr};
    double* bp[3] = {nullptr, nullptr, nullptr};
    Hp[which] = Hs.data(); bp[which] = bs.data();
    dev_.check(sdso_ba_get_stitched(dev_.ctx(), win_, Hp[0], bp[0], Hp[1], bp[1], Hp[2], bp[2]), "sdso_ba_get_stitched");
    H.resize(n, n); b.resize(n);
    for (int i = 0; i < n; i++) { b(i) = bs[i]; for (int j = 0; j < n; j++) H(i, j) = Hs[(size_t)i * n + j]; }
  }
  template <class PointFrameResidualT> int index_of_(PointFrameResidualT* r) const {
    auto it = res_index_.find(r);
    if (it == res_index_.end()) throw Error("residual is not part of the uploaded window");
    return it->second;
  }
  Device& dev_;
  int win_, nf_ = 0, resInM_seen_ = 0;
  EnergyFunctionalT* ef_ = nullptr;
  int last_marg_res_ = 0;
  bool lin_valid_ = false, app_valid_ = false, acc_valid_ = false, marg_valid_ = false;
  std::unordered_map<const void*, int> res_index_, point_index_;
  std::vector<uint8_t> l_state_, a_state_, a_act_;
  std::vector<float> l_energy_, l_energyWO_;
  std::vector<double> evalPT_, state_, state_zero_, HM_, bM_;
  std::vector<float> exposure_, energyTH_, u_, v_, idepth_, idepth_zero_, color_, weights_;
  std::vector<float> maxRelBaseline_;
  std::vector<int> frameID_, slots_, host_, res_point_, res_target_, numGood_;
  std::vector<uint8_t> prior_, res_state_, isNew_;
  // an SE3 of the reference's type from R, t (Sophus::SE3d(Matrix3d, Vector3d); the prototype only lends its types)
  template <class SE3T>
  static SE3T like(const SE3T& proto, const sdso_se3_t& a) {
    std::decay_t<decltype(proto.rotationMatrix())> R;
    std::decay_t<decltype(proto.translation())> t;
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R(i, j) = a.R[i * 3 + j]; t[i] = a.t[i]; }
    return SE3T(R, t);
  }
  std::vector<std::decay_t<decltype(std::declval<EnergyFunctionalT&>().frames[0]->points[0])>> points_;          // EFPoint*
  std::vector<std::decay_t<decltype(std::declval<EnergyFunctionalT&>().frames[0]->points[0]->residualsAll[0])>> residuals_;  // EFResidual*
};

// =================================================================================== accumulators
// AccumulatedTopHessianSSE / AccumulatedSCHessianSSE with the reference's member signatures (AccumulatedTopHessian.h:66-97, :162-169;
// AccumulatedSCHessian.h:66-96, :155-160) over the device window.  The device forms all three accumulations of a linearised state in one
// pass (k_ba_lin_fused / k_ba_accum_top + k_ba_sc_host), so setZero / addPoint<mode> / addPointsInternal<mode> are BOOKKEEPING — which mode
// the caller is accumulating, and for mode 2 (marginalizePointsF, EnergyFunctional.cpp:663-736) which points it passes — and the stitch
// members return the stitched systems of that pass (sdso_ba_get_stitched):
//   top, mode 0 : stitchDouble[MT](H, b, EF, usePrior = false, ..)   accumulateAF_MT  (EnergyFunctional.cpp:212-232)
//   top, mode 1 : stitchDouble[MT](H, b, EF, usePrior = true, ..)    accumulateLF_MT  (:236-254)
//   top, mode 2 : stitchDouble(M, Mb, EF, false, false)               marginalizePointsF (:707): runs sdso_ba_marginalize_points for the collected points
//   bottom      : stitchDouble[MT](H, b, EF, ..)                      accumulateSCF_MT (:256-269) / marginalizePointsF (:708)
// `tid` / `min` / `max` / `stats` / the IndexThreadReduce pointer are accepted and ignored (a GPU backend is free to ignore tid).
template <class BA>
class AccumulatedTopHessianSSE {
 public:
  explicit AccumulatedTopHessianSSE(BA& ba) : ba_(ba) { for (int i = 0; i < 6; i++) { nframes[i] = 0; nres[i] = 0; } }
  int nframes[6];   // NUM_THREADS (util/NumType.h:38)
  int nres[6];
  template <class StatsT = void>
  void setZero(int nFrames, int /*min*/ = 0, int /*max*/ = 1, StatsT* /*stats*/ = 0, int tid = 0) {
    nframes[tid] = nFrames; nres[tid] = 0; mode_ = -1; flagged_.clear();
  }
  template <int mode, class EFPointT, class EFT>
  void addPoint(EFPointT* p, EFT const* /*ef*/, int /*tid*/ = 0) {
    static_assert(mode >= 0 && mode <= 2, "addPoint<mode>: 0 active, 1 linearized, 2 marginalize");
    if (mode_ != -1 && mode_ != mode) throw Error("AccumulatedTopHessianSSE: one mode per setZero");
    mode_ = mode;
    if (mode == 2) flagged_.push_back(p);
  }
  template <int mode, class EFPointT, class EFT, class StatsT = void>
  void addPointsInternal(std::vector<EFPointT*>* points, EFT const* ef, int min = 0, int max = 1, StatsT* /*stats*/ = 0, int tid = 0) {
    for (int i = min; i < max; i++) addPoint<mode>((*points)[i], ef, tid);
  }
  template <class MatXX, class VecX, class EFT>
  void stitchDouble(MatXX& H, VecX& b, EFT const* /*EF*/, bool usePrior, bool /*useDelta*/, int /*tid*/ = 0) { stitch_(H, b, usePrior); }
  template <class RedT, class MatXX, class VecX, class EFT>
  void stitchDoubleMT(RedT* /*red*/, MatXX& H, VecX& b, EFT const* /*EF*/, bool usePrior, bool /*MT*/) { stitch_(H, b, usePrior); }
  const std::vector<const void*>& flagged() const { return flagged_; }

 private:
  template <class MatXX, class VecX> void stitch_(MatXX& H, VecX& b, bool usePrior) {
    if (mode_ == 2) {
      if (usePrior) throw Error("AccumulatedTopHessianSSE: the marginalisation stitch is called without priors (EnergyFunctional.cpp:707)");
      ba_.ensureMarginalizedErased(flagged_);
      ba_.stitchedSystem(0, H, b);
      nres[0] = ba_.lastMargResiduals();
      return;
    }
    // mode 0 is stitched without the priors, mode 1 with them — the only combinations the reference forms (EnergyFunctional.cpp:218, :240)
    const int which = mode_ == 1 ? 1 : 0;
    if (usePrior != (which == 1)) throw Error("AccumulatedTopHessianSSE: accumulateAF_MT stitches without priors, accumulateLF_MT with them");
    ba_.ensureAccumulated();
    ba_.stitchedSystem(which, H, b);
    nres[0] = ba_.countOf(which);
  }
  BA& ba_;
  int mode_ = -1;
  std::vector<const void*> flagged_;
};
template <class BA>
class AccumulatedSCHessianSSE {
 public:
  explicit AccumulatedSCHessianSSE(BA& ba) : ba_(ba) { for (int i = 0; i < 6; i++) nframes[i] = 0; }
  int nframes[6];
  template <class StatsT = void>
  void setZero(int n, int /*min*/ = 0, int /*max*/ = 1, StatsT* /*stats*/ = 0, int tid = 0) { nframes[tid] = n; marg_ = false; }
  template <class EFPointT>
  void addPoint(EFPointT* /*p*/, bool shiftPriorToZero, int /*tid*/ = 0) { if (!shiftPriorToZero) marg_ = true; }   // (false only in marginalizePointsF, :700)
  template <class EFPointT, class StatsT = void>
  void addPointsInternal(std::vector<EFPointT*>* points, bool shiftPriorToZero, int min = 0, int max = 1, StatsT* /*stats*/ = 0, int tid = 0) {
    for (int i = min; i < max; i++) addPoint((*points)[i], shiftPriorToZero, tid);
  }
  template <class MatXX, class VecX, class EFT>
  void stitchDouble(MatXX& H, VecX& b, EFT const* /*EF*/, int /*tid*/ = 0) { stitch_(H, b); }
  template <class RedT, class MatXX, class VecX, class EFT>
  void stitchDoubleMT(RedT* /*red*/, MatXX& H, VecX& b, EFT const* /*EF*/, bool /*MT*/) { stitch_(H, b); }

 private:
  template <class MatXX, class VecX> void stitch_(MatXX& H, VecX& b) {
    if (marg_) ba_.requireMarginalized();     // the top accumulator's stitch (called first, :707-708) ran the device pass
    else ba_.ensureAccumulated();
    ba_.stitchedSystem(2, H, b);
  }
  BA& ba_;
  bool marg_ = false;
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

// The member itself, ONE point (ImmaturePoint.h:89): `ImmaturePointStatus ImmaturePoint::traceStereo(FrameHessian* frame, Mat33f K,
// bool mode_right)` becomes `return sdso_shim::traceStereo(dev, this, slot_of(frame), K, baseline, mode_right);` — same state changes on
// the point, same return value.  A launch per point: the callers' loops (FullSystem.cpp:581-613, :667-725) want traceStereoAll.
template <class ImmaturePointT, class Mat33fT>
inline auto traceStereo(Device& dev, ImmaturePointT* p, int frame_slot, const Mat33fT& K, float baseline, bool mode_right) -> decltype(p->lastTraceStatus) {
  std::vector<ImmaturePointT*> one{p};
  std::vector<uint8_t> st;
  traceStereoAll(dev, one, frame_slot, K, baseline, mode_right, st);
  return static_cast<decltype(p->lastTraceStatus)>(st[0]);
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

// The member itself, ONE point (ImmaturePoint.h:90): `ImmaturePointStatus ImmaturePoint::traceOn(FrameHessian* frame, Mat33f
// hostToFrame_KRKi, Vec3f hostToFrame_Kt, Vec2f hostToFrame_affine, CalibHessian* HCalib, bool debugPrint)` becomes
// `return sdso_shim::traceOn(dev, this, slot_of(frame), KRKi, Kt, aff);` (HCalib is not read by the DSO-native body; debugPrint prints).
template <class ImmaturePointT, class Mat33fT, class Vec3fT, class Vec2fT>
inline auto traceOn(Device& dev, ImmaturePointT* p, int frame_slot, const Mat33fT& hostToFrame_KRKi, const Vec3fT& hostToFrame_Kt,
                    const Vec2fT& hostToFrame_affine) -> decltype(p->lastTraceStatus) {
  sdso_trace_geom_t g;
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) g.KRKi[r * 3 + c] = hostToFrame_KRKi(r, c);
  for (int r = 0; r < 3; r++) g.Kt[r] = hostToFrame_Kt[r];
  g.aff[0] = hostToFrame_affine[0]; g.aff[1] = hostToFrame_affine[1];
  std::vector<ImmaturePointT*> one{p};
  std::vector<uint8_t> st;
  traceOnAll(dev, one, std::vector<int>{0}, std::vector<sdso_trace_geom_t>{g}, frame_slot, st);
  return static_cast<decltype(p->lastTraceStatus)>(st[0]);
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
