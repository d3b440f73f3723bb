/*
 * sdso_abi.h — C-ABI of libsdso_hip.so, the MI355X (gfx950) implementation of Stereo-DSO's
 * photometric direct-alignment hot path.
 *
 * The reference (gyubeomim/stereo-dso-g2o) has no FFI layer: the path sits behind plain C++
 * member functions called by FullSystem.  Each entry point below names the reference member
 * function (file:line under /root/reference) whose work it performs; the header-only C++ shims
 * in stereo-dso-g2o_amd/host/ keep the reference's class/method names on top of these calls
 * (see INTEGRATION.md).
 *
 * Conventions
 *   - every function returns 0 (SDSO_OK) or a negative SDSO_ERR_* code; nothing throws;
 *     sdso_last_error() returns a static/ctx-owned message for the last failure on that ctx.
 *   - all pointers are caller-owned HOST buffers unless the name ends in _dev (device pointer).
 *   - matrices are row-major unless stated; SE3 = 3x3 rotation (row-major) + translation.
 *   - one ctx = one GPU + one HIP stream; calls on one ctx must come from one thread at a time
 *     (the reference serialises the same objects under trackMutex / mapMutex,
 *     src/FullSystem/FullSystem.cpp:1063, :1345).
 *   - there is NO CPU fallback: without a usable HIP device sdso_ctx_create fails with
 *     SDSO_ERR_NODEV and every other call fails with SDSO_ERR_STATE.
 */
#ifndef SDSO_ABI_H
#define SDSO_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDSO_OK 0
#define SDSO_ERR_ARG (-1)
#define SDSO_ERR_HIP (-2)
#define SDSO_ERR_NODEV (-3)
#define SDSO_ERR_STATE (-4)

#define SDSO_PYR_LEVELS 6      /* src/util/settings.h:46  PYR_LEVELS */
#define SDSO_PATTERN 8         /* src/util/settings.h:177 patternNum */
#define SDSO_MAX_RES 8         /* src/util/NumType.h:37  MAX_RES_PER_POINT */
#define SDSO_CPARS 4           /* src/util/NumType.h:47  CPARS */
#define SDSO_J_FLOATS 74       /* RawResidualJacobian payload, src/OptimizationBackend/RawResidualJacobian.h:32-65 */
#define SDSO_TOP_FLOATS 91     /* AccumulatorApprox packed sums 55+30+6, MatrixAccumulators.h:595-614 */

typedef struct sdso_ctx sdso_ctx;

/* ------------------------------------------------------------------ context */
int sdso_ctx_create(int device_ordinal, sdso_ctx** out);
void sdso_ctx_destroy(sdso_ctx* ctx);
const char* sdso_last_error(const sdso_ctx* ctx);
/* the HIP stream (hipStream_t) all kernels of this ctx are launched on */
void* sdso_ctx_stream(sdso_ctx* ctx);
int sdso_ctx_sync(sdso_ctx* ctx);
/* Split the device's CUs between two streams of the ctx (hipExtStreamCreateWithCUMask): aux_cus of them for the Schur accumulation and the
 * fused tail kernel of a batch's GN iteration (sdso_ba_batch_schur / sdso_ba_batch_solve_step), the rest for everything else — so that the
 * tail of one ctx's batch owns its CUs while another ctx's linearisation streams on the others.  stride 0: the lowest CU indices; stride s:
 * an equal share out of every s indices.  aux_cus 0 removes the partition.  Streams are re-created: call before sdso_ctx_stream is used.
 * No counterpart in the reference (a schedule); results are bit-identical with and without it. */
int sdso_ctx_partition_cus(sdso_ctx* ctx, int aux_cus, int stride);

/* Optional kernel timing with HIP events recorded on the ctx stream around launches.  on = 1: the dominant kernel of each workload
 * ("k_track_eval", "k_track_lm", "k_ba_lin_fused", "k_ba_linearize", "k_trace_stereo", ...); on = 2: also the secondary ones
 * ("k_ba_sc", "k_ba_tail", "k_ba_accum_top") — every bracket costs two event records on the stream, which a timed loop should only
 * pay for the kernel it reports; on = 0: off.
 * sdso_prof_read synchronises the stream and returns the accumulated milliseconds / launch count. */
int sdso_prof_enable(sdso_ctx* ctx, int on);
int sdso_prof_reset(sdso_ctx* ctx);
int sdso_prof_read(sdso_ctx* ctx, const char* kernel, double* total_ms, long* launches);

/* Self-test of the device's Lie-group arithmetic (the SE3::exp / product / inverse that the resident tracker LM and GN loops run in
 * kernels): for n tangents xi (6n; translation first, rotation last, se3.hpp:395-397) T_exp[i] = exp(xi_i), T_inv[i] = exp(xi_i)^-1,
 * T_mul_next[i] = exp(xi_i) * exp(xi_(i+1 mod n)), each as R (9, row-major) then t (3).  tests/test_oracle_se3.py puts the Sophus
 * fixtures of thirdparty/Sophus/sophus/test_se3.cpp:40-82 through it. */
int sdso_selftest_se3(sdso_ctx* ctx, int n, const double* xi, double* T_exp, double* T_inv, double* T_mul_next);

/* ------------------------------------------------------------------ pyramids
 * FrameHessian::dIp[lvl] (src/FullSystem/HessianBlocks.h:107-108): Vector3f {I, dx, dy} per pixel.
 * Images are immutable after FrameHessian::makeImages (HessianBlocks.cpp:141-203), so they are
 * mirrored once into device memory and addressed by an integer slot. */
int sdso_pyramid_levels(int w, int h); /* src/util/globalCalib.cpp:52-58 */
int sdso_upload_pyramid(sdso_ctx* ctx, int frame_slot, int levels, const int* w, const int* h,
                        const float* const* dIp /* [levels] AoS float3, w*h*3 floats each */);
/* FrameHessian::makeImages on the device from the level-0 irradiance image (w*h floats). */
int sdso_make_pyramid(sdso_ctx* ctx, int frame_slot, int w, int h, const float* color);
/* Photometric calibration for the pixel selection (setting_gammaWeightsPixelSelect == 1): absSquaredGrad of every pyramid built or
 * uploaded after the call is multiplied by CalibHessian::getBGradOnly(I)^2 (HessianBlocks.cpp:194-198, HessianBlocks.h:356-362).
 * B = CalibHessian::B (256 floats); NULL = identity response (weight exactly 1, the default).
 * sdso_gamma_from_binv = FullSystem::setGammaFunction (FullSystem.cpp:210-234): B from the inverse response table. */
int sdso_set_gamma(sdso_ctx* ctx, const float* B /* 256 or NULL */);
int sdso_gamma_from_binv(const float* BInv /* 256 */, float* B /* 256 */);
/* copy level `lvl` back as AoS float3 (tests) */
int sdso_download_pyramid_level(sdso_ctx* ctx, int frame_slot, int lvl, float* dI_out);
/* FrameHessian::absSquaredGrad[lvl] (HessianBlocks.h:109) as the device holds it: w_l*h_l floats */
int sdso_download_abs_grad(sdso_ctx* ctx, int frame_slot, int lvl, float* out);
int sdso_release_pyramid(sdso_ctx* ctx, int frame_slot);

/* ------------------------------------------------------------------ coarse tracker
 * CoarseTracker::calcRes + calcGSSSE (src/FullSystem/CoarseTracker.cpp:600-792, :537-596),
 * DSO-native arithmetic (the residual/Huber/buffer code kept as comments at :699-775). */
typedef struct {
  int lvl;
  int w, h;              /* w[lvl], h[lvl]                       CoarseTracker.cpp:609-610 */
  float fx, fy, cx, cy;  /* fx[lvl]..cy[lvl]                     CoarseTracker.cpp:612-615 */
  float Ki[9];           /* Ki[lvl] = K[lvl].inverse()           CoarseTracker.cpp:130     */
  float RKi[9];          /* R.cast<float>() * Ki[lvl]            CoarseTracker.cpp:617     */
  float t[3];            /* refToNew.translation().cast<float>() CoarseTracker.cpp:618     */
  float affLL[2];        /* AffLight::fromToVecExposure(...).cast<float>()  :621           */
  float ref_b0;          /* lastRef_aff_g2l.b                    CoarseTracker.cpp:542     */
  float cutoffTH;        /* setting_coarseCutoffTH*levelCutoffRepeat        :894           */
  float huberTH;         /* setting_huberTH                      settings.cpp:95           */
} sdso_track_eval_t;

/* pc_u/pc_v/pc_idepth/pc_color[lvl], pc_n[lvl] (CoarseTracker.h:132-140) */
int sdso_track_set_ref(sdso_ctx* ctx, int ref_slot, int lvl, int n, const float* pc_u,
                       const float* pc_v, const float* pc_idepth, const float* pc_color);
int sdso_track_release_ref(sdso_ctx* ctx, int ref_slot);

/* CoarseTracker::makeCoarseDepthL0 (CoarseTracker.cpp:352-534) on the device: STEP1 splat of n weighted inverse depths at
 * integer pixels (u,v) of the newest keyframe `frame_slot` (= lastRef), STEP2 2x2 pyramid sums, STEP3/4 dilation, STEP5
 * normalisation + compaction in raster order, for every level of that pyramid.  The result is installed as tracking
 * reference `ref_slot` exactly as sdso_track_set_ref would; pc_n_out[levels] receives pc_n[lvl].
 * (new_idepth / weight are what STEP1 computes per active point: the stereo-refined idepth — sdso_stereo_match_batch —
 * and sqrtf(1e-3 / (HdiF + 1e-12)), :350.) */
int sdso_track_make_ref(sdso_ctx* ctx, int ref_slot, int frame_slot, int n, const int* u, const int* v,
                        const float* new_idepth, const float* weight, int* pc_n_out);
/* read a template level back (tests); n_out = pc_n[lvl]; arrays may be NULL */
int sdso_track_get_ref(sdso_ctx* ctx, int ref_slot, int lvl, int* n_out, float* pc_u, float* pc_v,
                       float* pc_idepth, float* pc_color);

/* Host-only helper (no GPU work): fill the evaluation parameters for (level, pose, affine) the way
 * CoarseTracker::makeK (:108-136) and calcRes (:617-621) derive them; cutoffTH =
 * prm->coarseCutoffTH * levelCutoffRepeat. Declared after sdso_track_params_t below. */

/* One fused evaluation = calcRes followed by calcGSSSE on the buffers it filled.
 *   H[64], b[8]  : H_out / b_out after the scaling at CoarseTracker.cpp:581-595 (double)
 *   res[6]       : Vec6 (double) returned by calcRes (:783-789)
 *   n_warped     : buf_warped_n INCLUDING the zero padding to a multiple of 4 (:763-775)
 *   inlier_mask  : optional, n bytes; 1 where the point was appended to buf_warped_*          */
int sdso_track_calc_res_gs(sdso_ctx* ctx, int ref_slot, int frame_slot,
                           const sdso_track_eval_t* ev, double* H, double* b, double* res,
                           int* n_warped, uint8_t* inlier_mask);

/* The same for `nprob` independent (reference, frame, pose) problems in ONE launch. Outputs are
 * arrays of nprob entries (H: nprob*64, b: nprob*8, res: nprob*6, n_warped: nprob). */
int sdso_track_calc_res_gs_batch(sdso_ctx* ctx, int nprob, const int* ref_slots,
                                 const int* frame_slots, const sdso_track_eval_t* evs, double* H,
                                 double* b, double* res, int* n_warped);
/* Asynchronous form used by the benchmark: enqueue only (no host copy, no sync); results stay in
 * the ctx's device result buffer and are fetched with sdso_track_batch_fetch. */
int sdso_track_batch_prepare(sdso_ctx* ctx, int nprob, const int* ref_slots,
                             const int* frame_slots, const sdso_track_eval_t* evs);
int sdso_track_batch_enqueue(sdso_ctx* ctx);
int sdso_track_batch_fetch(sdso_ctx* ctx, double* H, double* b, double* res, int* n_warped);

typedef struct { double R[9]; double t[3]; } sdso_se3_t;
typedef struct { double a, b; } sdso_aff_t;

typedef struct {
  int levels;                               /* pyrLevelsUsed */
  int w[SDSO_PYR_LEVELS], h[SDSO_PYR_LEVELS];
  float fx[SDSO_PYR_LEVELS], fy[SDSO_PYR_LEVELS], cx[SDSO_PYR_LEVELS], cy[SDSO_PYR_LEVELS]; /* makeK :108-136 */
  float ref_exposure, new_exposure;         /* lastRef->ab_exposure, newFrame->ab_exposure */
  sdso_aff_t ref_aff_g2l;                   /* lastRef_aff_g2l */
  int coarsestLvl;
  double minResForAbort[5];
  float coarseCutoffTH;                     /* setting_coarseCutoffTH = 20 */
  float huberTH;                            /* setting_huberTH = 9 */
  int maxIterations[5];                     /* DSO-native {10,20,50,50,50} (CoarseTracker.cpp:861) */
  double affineOptModeA, affineOptModeB;    /* setting_affineOptModeA/B */
} sdso_track_params_t;

typedef struct {
  int good;                    /* the bool trackNewestCoarse returns */
  double lastResiduals[5];     /* NaN where the level was not reached */
  double lastFlowIndicators[3];
  int iterations[5];           /* LM iterations actually run per level */
  int evaluations;             /* total calcRes evaluations */
  long long point_evals;       /* sum of pc_n[lvl] over all evaluations */
} sdso_track_result_t;

void sdso_track_make_eval(const sdso_track_params_t* prm, int lvl, const sdso_se3_t* refToNew,
                          const sdso_aff_t* aff_g2l, float levelCutoffRepeat, sdso_track_eval_t* ev);

/* CoarseTracker::trackNewestCoarse (CoarseTracker.cpp:827-1069), DSO-native LM (the loop kept as
 * comments at :908-1024).  lastToNew / aff_g2l are in/out exactly like the reference's references. */
int sdso_track_newest_coarse(sdso_ctx* ctx, int ref_slot, int frame_slot,
                             const sdso_track_params_t* prm, sdso_se3_t* lastToNew,
                             sdso_aff_t* aff_g2l, sdso_track_result_t* out);
/* The same for `nhyp` independent (reference, frame, initial pose) hypotheses in lock-step: every round evaluates the pending
 * calcRes+calcGSSSE of all unfinished hypotheses in one launch (FullSystem::trackNewCoarse tries up to 53 initial motions one
 * after the other, FullSystem.cpp:305-441).  Every hypothesis follows exactly the evaluation sequence of the single call.
 * prms / lastToNew / aff_g2l / outs are arrays of nhyp entries (in/out like the single call). */
int sdso_track_newest_coarse_batch(sdso_ctx* ctx, int nhyp, const int* ref_slots, const int* frame_slots,
                                   const sdso_track_params_t* prms, sdso_se3_t* lastToNew, sdso_aff_t* aff_g2l,
                                   sdso_track_result_t* outs);

/* ------------------------------------------------------------------ windowed bundle adjustment
 * One "window" mirrors an EnergyFunctional (src/OptimizationBackend/EnergyFunctional.h:49-150) with
 * its frames / points / residuals flattened in EnergyFunctional::makeIDX order (:998-1018):
 * points in allPoints order, residuals grouped by point in residualsAll order. */
typedef struct {
  int nf, np, nr;
  int w, h;                        /* wG[0], hG[0] */
  double calib_value_scaled[4];    /* CalibHessian::value_scaled  fx fy cx cy   HessianBlocks.h:276-349 */
  double calib_value_zero[4];      /* CalibHessian::value_zero (unscaled)                              */
  /* frames */
  const double* evalPT;            /* nf*12  worldToCam_evalPT: R (9) then t (3)      HessianBlocks.h:132 */
  const double* state;             /* nf*10  FrameHessian::state                      HessianBlocks.h:137 */
  const double* state_zero;        /* nf*10  FrameHessian::state_zero                                     */
  const float* ab_exposure;        /* nf */
  const float* frameEnergyTH;      /* nf */
  const int* frameID;              /* nf     frameID==0 carries the pose prior        HessianBlocks.h:239-265 */
  const int* frame_slot;           /* nf     device pyramid slot of each keyframe (product) */
  const float* const* dI;          /* nf     host level-0 images (oracle only; product ignores) */
  /* points */
  const float* u;                  /* np */
  const float* v;                  /* np */
  const float* idepth;             /* np     PointHessian::idepth (== idepth_scaled, SCALE_IDEPTH=1) */
  const float* idepth_zero;        /* np */
  const float* color;              /* np*8 */
  const float* weights;            /* np*8 */
  const int* host;                 /* np     host frame index (EFFrame::idx) */
  const uint8_t* hasDepthPrior;    /* np */
  /* residuals */
  const int* res_point;            /* nr     non-decreasing */
  const int* res_target;           /* nr */
  const uint8_t* res_state;        /* nr     ResState {IN=0,OOB=1,OUTLIER=2}          Residuals.h:49 */
  /* marginalisation prior and nullspaces */
  const double* HM;                /* (8nf+4)^2 */
  const double* bM;                /* 8nf+4 */
  /* settings */
  int solverMode;                  /* setting_solverMode (settings.h:32-43): every bit solveSystemF reads is honoured (SVD, SVD_CUT7,
                                      ORTHOGONALIZE_SYSTEM, REMOVE_POSEPRIOR, USE_GN, FIX_LAMBDA, ORTHOGONALIZE_X[_LATER]); SVD and
                                      ORTHOGONALIZE_SYSTEM windows are solved per window (not in sdso_ba_batch_*) */
  double affineOptModeA, affineOptModeB;
  int forceAcceptStep;             /* setting_forceAceptStep */
  /* bookkeeping FullSystem::linearizeAll(true) updates at the end of optimize (FullSystemOptimize.cpp:64-77); each may be NULL */
  const float* maxRelBaseline;     /* np     PointHessian::maxRelBaseline   (NULL: 0)                              */
  const int* numGoodResiduals;     /* np     PointHessian::numGoodResiduals (NULL: 0)                              */
  const uint8_t* res_isNew;        /* nr     PointFrameResidual::isNew      (NULL: 1, what Residuals.cpp:79 sets)  */
} sdso_ba_window_t;

int sdso_ba_upload_window(sdso_ctx* ctx, int win, const sdso_ba_window_t* W);
int sdso_ba_release_window(sdso_ctx* ctx, int win);

/* FullSystem::linearizeAll(false) over every residual (FullSystemOptimize.cpp:142-203) =
 * PointFrameResidual::linearize (Residuals.cpp:83-336).  Returns the summed energy (stats[0]). */
int sdso_ba_linearize(sdso_ctx* ctx, int win, double* energy);
/* fetch what the latest linearisation produced (any pointer may be NULL).  After the FUSED linearise + applyRes + accumulate kernel
 * (sdso_ba_batch_accumulate, the resident loop) the records were written straight into EFResidual::J's slot — the swap of
 * takeDataF (EnergyFunctionalStructs.cpp:39) is skipped because the swapped-out copy is dead in the accepted-step flow — and are
 * returned from there, for applied (IN) and not applied (OUTLIER) residuals alike.
 *   J[nr*74] in RawResidualJacobian field order: resF8 Jpdxi0(6) Jpdxi1(6) Jpdc0(4) Jpdc1(4) Jpdd(2)
 *            JIdx0(8) JIdx1(8) JabF0(8) JabF1(8) JIdx2(4) JabJIdx(4) Jab2(4)
 *   newState[nr], newEnergy[nr], newEnergyWithOutlier[nr], projectedTo[nr*16], centerProjectedTo[nr*3] */
int sdso_ba_get_linearization(sdso_ctx* ctx, int win, float* J, uint8_t* newState, float* newEnergy,
                              float* newEnergyWithOutlier, float* projectedTo,
                              float* centerProjectedTo);
/* PointFrameResidual::applyRes(true) for every residual (Residuals.cpp:367-385) incl.
 * EFResidual::takeDataF (EnergyFunctionalStructs.cpp:37-51). */
int sdso_ba_apply_res(sdso_ctx* ctx, int win);
int sdso_ba_get_residual_state(sdso_ctx* ctx, int win, uint8_t* state, uint8_t* isActive,
                               float* JpJdF /* nr*8 */);
/* EFResidual::J (the record takeDataF swapped in, EnergyFunctionalStructs.cpp:39) of every residual, J[nr*74] in the field order of
 * sdso_ba_get_linearization; rows of residuals that are not active are unspecified (the reference never reads them either:
 * AccumulatedTopHessian.cpp:49 / EnergyFunctional.cpp:675 skip !isActive()). */
int sdso_ba_get_ef_jacobians(sdso_ctx* ctx, int win, float* J);

/* accumulateAF_MT + accumulateLF_MT + accumulateSCF_MT up to (not including) the stitch
 * (EnergyFunctional.cpp:212-269; AccumulatedTopHessian.cpp:36-198; AccumulatedSCHessian.cpp:34-103). */
int sdso_ba_accumulate(sdso_ctx* ctx, int win);
/* packed accumulator block of this window, as all-reduced across ranks:
 *   [ topA nf*nf*91 | topL nf*nf*91 | accD nf^3*64 | accE nf*nf*32 | accEB nf*nf*8 | accHcc 16 | accbc 4 | nresA nresL ]
 * sdso_ba_accum_floats gives its length; _dev returns the device pointer (float*) for RCCL. */
int sdso_ba_accum_floats(int nf);
int sdso_ba_accum_dev(sdso_ctx* ctx, int win, void** dev_ptr);
int sdso_ba_get_accumulators(sdso_ctx* ctx, int win, float* packed);
int sdso_ba_set_accumulators(sdso_ctx* ctx, int win, const float* packed); /* after a host-side reduction */
int sdso_ba_get_point_terms(sdso_ctx* ctx, int win, float* HdiF, float* bdSumF, float* Hdd_accAF,
                            float* bd_accAF, float* Hcd_accAF /* np*4 */);

/* stitch (AccumulatedTopHessian.cpp:265-337, AccumulatedSCHessian.cpp:106-195) + the rest of
 * EnergyFunctional::solveSystemF (EnergyFunctional.cpp:838-995) + resubstituteF_MT (:272-341).
 *   x[8nf+4]     : lastX
 *   H,b          : lastHS, lastbS (may be NULL)
 *   frame_step   : nf*8  (FrameHessian::step head<8>), calib_step: 4 */
int sdso_ba_solve(sdso_ctx* ctx, int win, int iteration, double lambda, double* x, double* HS,
                  double* bS, double* frame_step, double* calib_step);
/* EnergyFunctional::resubstituteF_MT (EnergyFunctional.cpp:272-341) alone, for a caller-supplied x (8nf+4): frame_step (nf*8) and
 * calib_step (4) = -x, the points' steps (resubstituteFPt, :305-341) on the device — sdso_ba_get_point_steps reads them.  Needs the
 * per-point terms of a preceding sdso_ba_accumulate. */
int sdso_ba_resubstitute(sdso_ctx* ctx, int win, const double* x, double* frame_step, double* calib_step);
int sdso_ba_get_point_steps(sdso_ctx* ctx, int win, float* step /* np */);

/* FullSystem::optimize, DSO-native GN loop (FullSystemOptimize.cpp:871-1041, the un-compiled #else):
 * returns final frame states, point idepths and residual states. */
typedef struct {
  int iterations;
  double lastEnergy;      /* lastEnergy[0] */
  double rmse;            /* sqrt(lastEnergy[0]/(patternNum*resInA)) */
  int resInA;
} sdso_ba_opt_result_t;
int sdso_ba_optimize(sdso_ctx* ctx, int win, int mnumOptIts, double* state_out /* nf*10 */,
                     float* idepth_out /* np */, uint8_t* res_state_out /* nr */,
                     sdso_ba_opt_result_t* out);

/* Everything FullSystem::optimize leaves behind in the reference's objects and that its callers read afterwards — the state after
 * the closing `linearizeAll(true)` (FullSystemOptimize.cpp:997-1041 with :52-87 and :142-203) and after the last solveSystemF's
 * AccumulatedSCHessianSSE::addPoint (AccumulatedSCHessian.cpp:34-60).  Valid after sdso_ba_optimize / sdso_ba_batch_optimize[_end]
 * of the window (SDSO_ERR_STATE before).  Every pointer may be NULL; arrays are caller-owned, in the window's point / residual order.
 * Consumers in the reference: CoarseTracker::makeCoarseDepthL0 reads lastResiduals[0].second (= state_state of that residual),
 * centerProjectedTo and efPoint->HdiF (CoarseTracker.cpp:295-350); FullSystem::flagPointsForRemoval reads residuals.size(),
 * isInlierNew() (numGoodResiduals), isOOB() (lastResiduals[].second, maxRelBaseline) and idepth_hessian (FullSystem.cpp:997-1040). */
typedef struct {
  /* per point [np] */
  float* idepth;                 /* PointHessian::idepth (== idepth_zero: doStepFromBackup sets both, :268-272)                       */
  float* step;                   /* PointHessian::step of the last solve                                                              */
  float* HdiF;                   /* EFPoint::HdiF          (AccumulatedSCHessian.cpp:58; 0 when the point had no active residual :44) */
  float* bdSumF;                 /* EFPoint::bdSumF        (:61-65)                                                                   */
  float* idepth_hessian;         /* PointHessian::idepth_hessian (:56; 0 at :46)                                                      */
  float* maxRelBaseline;         /* PointHessian::maxRelBaseline: reset at AccumulatedSCHessian.cpp:47, raised at FullSystemOptimize.cpp:68-74 */
  int* numGoodResiduals;         /* PointHessian::numGoodResiduals after :76                                                          */
  /* per residual [nr] */
  uint8_t* state_state;          /* PointFrameResidual::state_state after applyRes(true); also lastResiduals[k].second (:165-172)     */
  uint8_t* isActiveAndIsGoodNEW; /* EFResidual::isActiveAndIsGoodNEW (Residuals.cpp:367-385)                                          */
  float* state_energy;           /* PointFrameResidual::state_energy                                                                  */
  float* centerProjectedTo;      /* nr*3, meaningful where isActiveAndIsGoodNEW (the final linearisation reached :130-131); else 0    */
  float* projectedTo;            /* nr*16, same rule                                                                                  */
  uint8_t* toRemove;             /* 1: linearizeAll(true) put the residual on toRemove (:80-84) — the caller clears lastResiduals[].first,
                                    calls ef->dropResidual(r->efResidual) and deleteOut(ph->residuals, k) in this order (:176-195)   */
  /* frames */
  double* state;                 /* nf*10  FrameHessian::get_state()                                                                  */
  double* state_zero;            /* nf*10  (the newest frame's changes: setEvalPT at :1000-1003)                                      */
  double* evalPT;                /* nf*12  worldToCam_evalPT: R (9, row-major), t (3); the newest frame's is its PRE_worldToCam        */
  double* PRE_worldToCam;        /* nf*12  (shell->camToWorld = PRE_camToWorld at :1033-1037 is its inverse)                          */
  double* frame_step;            /* nf*10  FrameHessian::step of the last solve                                                       */
  float* frameEnergyTH;          /* nf     (the newest frame's: setNewFrameEnergyTH of the closing linearizeAll)                      */
  /* calibration */
  double calib_value[4];         /* CalibHessian::value (unscaled)                                                                    */
  double calib_value_scaled[4];
  double calib_step[4];
  /* EnergyFunctional members of the last solveSystemF */
  double* lastX;                 /* 8nf+4 */
  double* lastHS;                /* (8nf+4)^2; lastHS / lastbS (log-only in the reference, FullSystem.cpp:1691-1764) are kept by
                                    sdso_ba_optimize; inside a batch only after sdso_ba_batch_keep_system(ctx, 1): SDSO_ERR_STATE otherwise */
  double* lastbS;                /* 8nf+4 */
  int resInA, resInL, resInM;    /* nres of the last accumulateAF / LF; residuals marginalised through this window so far             */
  int n_toRemove;
  sdso_ba_opt_result_t result;   /* what the optimize call returned for this window                                                   */
} sdso_ba_post_state_t;
int sdso_ba_get_post_state(sdso_ctx* ctx, int win, sdso_ba_post_state_t* out);
/* EnergyFunctional::resInA / resInL of the latest accumulate (EnergyFunctional.cpp:219, :241) and resInM: the residuals marginalised
 * through this window by sdso_ba_marginalize_points so far (:704).  Any pointer may be NULL. */
int sdso_ba_get_counts(sdso_ctx* ctx, int win, int* resInA, int* resInL, int* resInM);

/* EnergyFunctional::marginalizePointsF (EnergyFunctional.cpp:663-736) for the points flagged in
 * marg_flag[np] (after flagPointsForRemoval's linearize + fixLinearizationF, FullSystem.cpp:1012-1021):
 * returns the updated HM / bM. */
int sdso_ba_marginalize_points(sdso_ctx* ctx, int win, const uint8_t* marg_flag, double* HM_out,
                               double* bM_out);

/* EnergyFunctional::marginalizeFrame (EnergyFunctional.cpp:554-660): remove keyframe `idx` (EFFrame::idx) from the
 * marginalisation prior.  prior8 / delta_prior8 are EFFrame::prior / delta_prior.  HM_in (8nf+4)^2, bM_in 8nf+4 ->
 * HM_out (8(nf-1)+4)^2, bM_out.  Host statement of the algebra on caller-owned arrays (no ctx); the device path is
 * sdso_ba_marginalize_frame_dev below. */
int sdso_ba_marginalize_frame(int nf, int idx, const double* prior8, const double* delta_prior8,
                              const double* HM_in, const double* bM_in, double* HM_out, double* bM_out);

/* EnergyFunctional::calcLEnergyF_MT (EnergyFunctional.cpp:420-442; EnergyFunctional.h:82) and calcMEnergyF (:344-351; .h:83) at the
 * window's current state: the values themselves (FullSystem::calcLEnergy / calcMEnergy return 0 under setting_forceAceptStep before
 * they ever call these, FullSystemOptimize.cpp:374-376, :1056).  Either pointer may be NULL. */
int sdso_ba_calc_energies(sdso_ctx* ctx, int win, double* EL, double* EM);
/* What EnergyFunctional::setDeltaF (EnergyFunctional.cpp:173-207; .h:75) leaves behind at the window's current state: cDeltaF (4),
 * EFFrame::delta and delta_prior (nf*8 each), EFPoint::deltaF (np); adHTdeltaF comes with sdso_ba_get_tables.  Any pointer may be NULL. */
int sdso_ba_get_deltas(sdso_ctx* ctx, int win, float* cDeltaF, double* frame_delta, double* frame_delta_prior, float* point_deltaF);

/* The same on the DEVICE-resident prior of an uploaded window: HM / bM as sdso_ba_marginalize_points left them on the device go through
 * the marginalisation of keyframe `idx` in one kernel (its EFFrame::prior / delta_prior are the window's own) and stay there; HM_out
 * (8(nf-1)+4)^2 / bM_out are optional host copies.  sdso_ba_adopt_prior(win, from_win) hands that prior to the next window — device to
 * device, the rows / columns of frames beyond it zero, as EnergyFunctional::insertFrame resizes HM / bM (EnergyFunctional.cpp:468-476);
 * upload `win` with HM = bM = NULL first.  The prior then never leaves the device between two keyframes.
 * Several frames leaving at one keyframe (FullSystem.cpp:1470-1476 marginalises every flagged frame in a loop): a call that follows another
 * one with no sdso_ba_marginalize_points in between continues from that result, `idx` then counting the frames the prior still covers
 * (EFFrame::idx after the earlier frame was erased); outputs are (8k+4)^2 / 8k+4 for the k frames left.  sdso_ba_adopt_prior fails (-1)
 * unless the adopting window's leading frames are exactly those k frames, in order (by frameID), and it was uploaded with a zero prior. */
int sdso_ba_marginalize_frame_dev(sdso_ctx* ctx, int win, int idx, double* HM_out, double* bM_out);
int sdso_ba_adopt_prior(sdso_ctx* ctx, int win, int from_win);

/* keep projectedTo / centerProjectedTo of PointFrameResidual (Residuals.h:96-99) for
 * sdso_ba_get_linearization; off by default (76 B of extra stores per residual). */
int sdso_ba_keep_projections(sdso_ctx* ctx, int win, int on);

/* Batches: any number of uploaded windows (same nf) advance through one GN iteration with ONE
 * launch per phase.  The packed accumulators of the batch are contiguous so that one RCCL
 * all-reduce over xGMI covers every window when the points of each window are sharded across
 * ranks:
 *   sdso_ba_batch_accumulate : linearizeAll + applyRes + accumulateAF/LF/SCF      (enqueue only)
 *   [all-reduce(sum) of sdso_ba_batch_accum_dev across ranks]
 *   sdso_ba_batch_solve      : stitch + solveSystemF + resubstituteF              (enqueue only)
 *   sdso_ba_batch_get_x      : lastX of every window (synchronises)
 * Windows in the SOLVER_SVD / SOLVER_ORTHOGONALIZE_SYSTEM modes (EnergyFunctional.cpp:876-900, 924-965) run on the device like the others
 * (one workgroup per window: projected system, parallel-order Jacobi eigen-decomposition or the pivoted LDL^T; enqueue only, resident
 * loop included); the members of a batch share one solver branch. */
int sdso_ba_batch_create(sdso_ctx* ctx, int nwin, const int* wins);
int sdso_ba_batch_accumulate(sdso_ctx* ctx);
/* = sdso_ba_batch_linearize (linearizeAll + applyRes + accumulateAF) followed by sdso_ba_batch_schur (accumulateLF/SCF + folds),
 * available separately for callers that interleave several batches on several contexts / streams */
int sdso_ba_batch_linearize(sdso_ctx* ctx);
int sdso_ba_batch_schur(sdso_ctx* ctx);
/* 1 (default): every linearization writes the RawResidualJacobian records to HBM like
 * PointFrameResidual::J; 0: they stay in registers of the fused linearize+accumulate kernel. */
int sdso_ba_batch_set_materialize(sdso_ctx* ctx, int materialize);
/* 1: every solve of the batch's resident loop also writes EnergyFunctional::lastHS / lastbS (EnergyFunctional.cpp:909-910; log-only in
 * the reference) so that sdso_ba_get_post_state can return them; 0 (default): they are not materialised (37 KB per window and iteration). */
int sdso_ba_batch_keep_system(sdso_ctx* ctx, int on);
/* Shape of the exchange of a SHARDED batch inside the resident loop (points of every window cut over the ranks of the ctx's
 * communicator; the accumulators are sums over points like the reference's per-thread copies, AccumulatedTopHessian.cpp:299-308):
 *   0 (default)  sdso_ba_allreduce = one all-reduce of the packed block; every rank stitches and solves every window;
 *   1            sdso_ba_allreduce = one reduce-scatter by window (rank r receives the sums of windows [r*nwin/N, (r+1)*nwin/N)), the
 *                solve of a window runs on that rank only, and x / xAd / nres of every window go round by one all-gather
 *                (136 + 8 nf^2 + 2 floats per window) inside sdso_ba_batch_solve_step: half the xGMI bytes per link.
 * Mode 1 applies when nwin divides by the ranks, the loop is the accepted-step flow and lastHS / lastbS are not kept; otherwise the call
 * falls back to the all-reduce.  The results are the same bits either way (the sums are the same numbers). */
int sdso_ba_batch_exchange_mode(sdso_ctx* ctx, int mode);
/* lambda is subject to the windows' solverMode exactly as in solveSystemF (SOLVER_USE_GN -> 0, SOLVER_FIX_LAMBDA -> 1e-5,
 * EnergyFunctional.cpp:840-846); the members of a batch must share one solverMode. */
int sdso_ba_batch_solve(sdso_ctx* ctx, double lambda, int orthogonalize_x);
int sdso_ba_batch_accum_dev(sdso_ctx* ctx, void** dev_ptr, long* nfloats);
int sdso_ba_batch_get_x(sdso_ctx* ctx, double* x /* nwin*(8nf+4) */);
/* EnergyFunctional::accumulateAF_MT / accumulateLF_MT / accumulateSCF_MT (EnergyFunctional.cpp:212-269) as the reference's callers see
 * them: the STITCHED systems (AccumulatedTopHessianSSE::stitchDoubleMT without / with priors, AccumulatedTopHessian.h:95-148;
 * AccumulatedSCHessianSSE::stitchDoubleMT, AccumulatedSCHessian.h:96-135) of the accumulators sdso_ba_accumulate left: (8nf+4)^2
 * row-major + (8nf+4) doubles each, any pointer may be NULL.  solveSystemF adds them up (:856-868); the fused kernels never
 * materialise them, this call runs the stitch kernels on demand (synchronises).  Right after sdso_ba_marginalize_points the same call
 * returns what marginalizePointsF stitches (EnergyFunctional.cpp:707-717): HA / bA = M, Mb of accSSE_top_A->stitchDouble(.., false, false)
 * over addPoint<2> of the flagged points, Hsc / bsc = Msc, Mbsc. */
int sdso_ba_get_stitched(sdso_ctx* ctx, int win, double* HA, double* bA, double* HL, double* bL, double* Hsc, double* bsc);

/* FullSystem::optimize (FullSystemOptimize.cpp:871-1041, DSO-native loop) for EVERY window of the batch, device-resident: the host
 * logic between the kernel phases — backupState / doStepFromBackup (:207-351), FrameHessian::setState, setPrecalcValues
 * (HessianBlocks.cpp:206-242), setDeltaF (EnergyFunctional.cpp:173-207), setNewFrameEnergyTH (:98-139), the break test — runs in one
 * workgroup per window (k_ba_opt_step), so a whole loop is enqueued without a host round trip.  The accepted-step flow
 * (setting_forceAceptStep = true, the reference's default, settings.cpp:53) runs through the fused linearise+accumulate kernel; with
 * forceAcceptStep = 0 the energy gate of :961-990 (calcLEnergy, calcMEnergy, accept or loadSateBackup + re-linearisation, lambda * 0.25 /
 * * 100) is taken on the device too (k_ba_opt_gate; single rank, sdso_ba_batch_optimize only).  The members of a batch share
 * forceAcceptStep.
 *   sdso_ba_batch_optimize       : the whole loop with the reference's lambda (1e-1 * 0.25^it, subject to solverMode) and
 *                                  orthogonalisation schedule; out[nwin].  With a communicator on ctx (points sharded over ranks) every
 *                                  iteration all-reduces the accumulators and all-gathers the newest-frame energies / break-test sums, so
 *                                  every rank takes the decisions of the unsharded window.
 *   ..._begin / sdso_ba_batch_step / ..._end : the same in pieces, for callers that drive the iterations themselves:
 *       begin ; per iteration { sdso_ba_batch_accumulate ; [sdso_ba_allreduce] ; sdso_ba_batch_solve ; sdso_ba_batch_step } ; end
 *     stop_on_convergence = 0 keeps every window iterating (benchmarks).
 *   sdso_ba_batch_solve_step     : sdso_ba_batch_solve + sdso_ba_batch_step as ONE enqueue — EnergyFunctional::solveSystemF (:838-995),
 *                                  resubstituteF (:272-341), doStepFromBackup, setPrecalcValues, setDeltaF, setNewFrameEnergyTH and the
 *                                  break test in one launch of the fused tail kernel (k_ba_tail: a persistent workgroup per window;
 *                                  the stitched system never leaves the CU).  lambda / orthogonalize_x as in sdso_ba_batch_solve.
 *   sdso_ba_get_state            : states / idepths / residual states of one window as they stand (synchronises). */
int sdso_ba_batch_optimize(sdso_ctx* ctx, int mnumOptIts, sdso_ba_opt_result_t* out /* nwin */);
int sdso_ba_batch_optimize_begin(sdso_ctx* ctx, int stop_on_convergence);
int sdso_ba_batch_step(sdso_ctx* ctx);
int sdso_ba_batch_solve_step(sdso_ctx* ctx, double lambda, int orthogonalize_x);
int sdso_ba_batch_optimize_end(sdso_ctx* ctx, sdso_ba_opt_result_t* out /* nwin, may be NULL */);
int sdso_ba_get_state(sdso_ctx* ctx, int win, double* state_out /* nf*10 */, float* idepth_out /* np */, uint8_t* res_state_out /* nr */);

/* ------------------------------------------------------------------ multi-GPU exchange
 * The points of every window are sharded over the ranks (one process per GPU, contiguous allPoints ranges, every rank holds all
 * pyramids and frame states); the packed accumulators are plain sums over points — the reference adds the per-thread copies the same
 * way before it stitches (AccumulatedTopHessian.cpp:299-308, AccumulatedSCHessian.cpp:136-185) — so ONE all-reduce(sum, float32) per
 * Gauss-Newton iteration over RCCL / xGMI makes every rank stitch and solve the same system:
 *     sdso_ba_batch_accumulate -> sdso_ba_allreduce -> sdso_ba_batch_solve           (all three only enqueue on the ctx stream)
 * RCCL is loaded (dlopen) by the first of these calls; single-GPU users never need it.
 *   sdso_comm_unique_id : rank 0 creates the 128-byte ncclUniqueId; the caller ships it to the other ranks (MPI, sockets, a file ...)
 *   sdso_comm_init      : collective over the nranks processes; binds the communicator to ctx (its device, its stream)
 *   sdso_comm_attach    : a second ctx of the same process / device uses the communicator of `owner`
 *   sdso_ba_allreduce   : in-place sum of the batch's contiguous block (sdso_ba_batch_accum_dev) over the ranks
 *   sdso_ba_allreduce_window : the same for one window's block (sdso_ba_accum_dev), between sdso_ba_accumulate and sdso_ba_solve
 * A single-iteration exchange sums everything solveSystemF needs.  The energy threshold of the newest frame (setNewFrameEnergyTH,
 * FullSystemOptimize.cpp:98-139: a 70 % quantile over its residuals) is NOT additive: a multi-iteration loop over sharded points has to
 * gather those energies as well — sdso_ba_batch_optimize / sdso_ba_batch_step do (one all-gather per iteration); sdso_ba_optimize is a
 * single-rank call. */
int sdso_comm_unique_id(void* id128);
int sdso_comm_init(sdso_ctx* ctx, int nranks, int rank, const void* id128);
int sdso_comm_attach(sdso_ctx* ctx, sdso_ctx* owner);
/* The same communicator over a transport of the caller's (MPI, sockets, a test harness) instead of RCCL: the library stages each
 * collective through host memory and calls allreduce (in-place sum of n floats) / allgather (n floats per rank, rank-major into recv);
 * both return 0 on success.  Slow path by construction (it synchronises the ctx stream around every collective). */
typedef int (*sdso_host_allreduce_fn)(void* user, float* buf, size_t n);
typedef int (*sdso_host_allgather_fn)(void* user, const float* send, float* recv, size_t n);
int sdso_comm_init_host(sdso_ctx* ctx, int nranks, int rank, sdso_host_allreduce_fn allreduce, sdso_host_allgather_fn allgather, void* user);
int sdso_comm_info(sdso_ctx* ctx, int* nranks, int* rank);   /* nranks = 0 without a communicator */
int sdso_comm_destroy(sdso_ctx* ctx);
int sdso_ba_allreduce(sdso_ctx* ctx);
int sdso_ba_allreduce_window(sdso_ctx* ctx, int win);

/* host tables derived from the frame states (tests): precalc nf*nf*27 floats per (host*nf+target)
 * {PRE_KRKiTll 9, PRE_KtTll 3, PRE_RTll_0 9, PRE_tTll_0 3, PRE_aff_mode 2, PRE_b0_mode 1}
 * (FrameFramePrecalc::set, HessianBlocks.cpp:206-242); adHost/adTarget nf*nf*64 doubles and
 * adHTdeltaF nf*nf*8 floats indexed h+t*nf (EnergyFunctional.cpp:41-119, :173-207). */
int sdso_ba_get_tables(sdso_ctx* ctx, int win, float* precalc, double* adHost, double* adTarget,
                       float* adHTdeltaF);

/* ------------------------------------------------------------------ static stereo
 * ImmaturePoint::ImmaturePoint (src/FullSystem/ImmaturePoint.cpp:33-62) for n pixels of a frame:
 * color[n*8], weights[n*8], gradH[n*4], energyTH[n] (NaN marks a rejected point). */
int sdso_immature_init_batch(sdso_ctx* ctx, int frame_slot, int n, const float* u, const float* v,
                             float* color, float* weights, float* gradH, float* energyTH);

/* in/out point state of ImmaturePoint::traceStereo (ImmaturePoint.h:60-102), SoA */
typedef struct {
  int n;
  float* u_stereo; float* v_stereo;
  float* idepth_min; /* the (unused by stereo) temporal idepth_min tested at ImmaturePoint.cpp:195 */
  float* idepth_min_stereo; float* idepth_max_stereo; float* idepth_stereo;
  float* color;      /* n*8 */
  float* weights;    /* n*8 */
  float* gradH;      /* n*4 */
  float* energyTH;
  float* quality;
  uint8_t* lastTraceStatus;     /* ImmaturePointStatus, in/out */
  float* lastTraceUV;           /* n*2 */
  float* lastTracePixelInterval;
} sdso_trace_points_t;

/* ImmaturePoint::traceStereo (ImmaturePoint.cpp:94-451) for every point; the sub-pixel GN is the
 * DSO-native one (twin at :707-769).  K = {fx,fy,cx,cy}. status[n] = returned ImmaturePointStatus. */
int sdso_trace_stereo_batch(sdso_ctx* ctx, int frame_slot, const float K[4], float baseline,
                            int mode_right, sdso_trace_points_t* pts, uint8_t* status);
/* The same in three steps, for callers that keep the point state resident in HBM:
 * prepare uploads it once, enqueue launches the trace (no copy, no sync; may be repeated),
 * fetch copies the in/out fields and the returned statuses back. */
int sdso_trace_stereo_prepare(sdso_ctx* ctx, int frame_slot, const float K[4], float baseline,
                              int mode_right, const sdso_trace_points_t* pts);
int sdso_trace_stereo_enqueue(sdso_ctx* ctx);
int sdso_trace_stereo_fetch(sdso_ctx* ctx, sdso_trace_points_t* pts, uint8_t* status);

/* PixelSelector::makeMaps (src/FullSystem/PixelSelector2.cpp:193-300, with makeHists :84-189 and select :330-540): candidate
 * pixels of the keyframe in `frame_slot` (needs pyramid levels 0..2; absSquaredGrad = dx*dx + dy*dy, identity response).
 *   potential : PixelSelector::currentPotential, in/out (3 after construction)
 *   map_out   : w*h floats, 0 / 1 / 2 / 4 = not selected / selected at level 0 / 1 / 2   (may be NULL)
 *   num_out   : the value makeMaps returns (points left after the random thinning)
 * The random pattern is glibc's rand() & 0xFF after srand(3141592) (:43-44), reproduced inside the library. */
int sdso_pixel_select(sdso_ctx* ctx, int frame_slot, float density, int recursionsLeft, float thFactor,
                      int* potential, float* map_out, int* num_out);
int sdso_pixel_selector_pattern(int n, unsigned char* out); /* host only: first n pattern bytes */

/* ImmaturePoint::traceOn (ImmaturePoint.cpp:459-828): the temporal epipolar search of every immature point of every host
 * keyframe in the newest frame (FullSystem::traceNewCoarseKey / NonKey, FullSystem.cpp:632-790).  geom[g] is the
 * hostToFrame geometry of host g (:654-665): KRKi = K R K^-1, Kt = K t, aff = AffLight::fromToVecExposure(...) as floats;
 * point_geom[i] selects it.  pts uses the traceStereo layout with u_stereo/v_stereo = u/v and idepth_min_stereo /
 * idepth_max_stereo = idepth_min / idepth_max (in/out); idepth_min and idepth_stereo are not used. */
typedef struct { float KRKi[9]; float Kt[3]; float aff[2]; } sdso_trace_geom_t;
int sdso_trace_on_batch(sdso_ctx* ctx, int frame_slot, int ngeom, const sdso_trace_geom_t* geom, const int* point_geom,
                        sdso_trace_points_t* pts, uint8_t* status);

/* FullSystem::optimizeImmaturePoint (FullSystemOptPoint.cpp:52-238) with ImmaturePoint::linearizeResidual
 * (ImmaturePoint.cpp:886-985), the DSO-native idepth-only Gauss-Newton, for n candidate points at once
 * (FullSystem::activatePointsMT, FullSystem.cpp:796-958).  pair_* are the FrameFramePrecalc members the reference reads
 * (host->targetPrecalc[target->idx]: PRE_RTll, PRE_tTll, PRE_aff_mode), indexed host*nf+target.
 *   status[n]      : 1 activated, 0 not well-constrained (the reference returns 0), -1 outlier (returns (PointHessian*)-1)
 *   idepth_out[n]  : currentIdepth at exit
 *   res_state[n*nf]: final ResState per target frame (255 for the host itself / not evaluated) */
typedef struct {
  int nf, w, h, n, minObs;
  float K[4];                              /* fxl fyl cxl cyl */
  const float* pair_R; const float* pair_t; const float* pair_aff;
  const int* frame_slot;                   /* nf pyramid slots */
  const float* const* dI;                  /* unused by the library (layout shared with the test oracle) */
  const int* host; const float* u; const float* v; const float* idepth_min; const float* idepth_max;
  const float* color; const float* weights; const float* energyTH;
} sdso_activate_t;
int sdso_activate_points_batch(sdso_ctx* ctx, const sdso_activate_t* A, int8_t* status, float* idepth_out, uint8_t* res_state);

/* Left-right-left matching as every caller of traceStereo performs it (FullSystem::stereoMatch FullSystem.cpp:581-613,
 * traceNewCoarseNonKey :667-725, CoarseTracker::makeCoarseDepthL0 CoarseTracker.cpp:295-347):
 *   forward: ImmaturePoint(u, v, frame A) traced into frame B   (interval idepth_min/max_stereo, NULL = fresh 0 / NaN)
 *   back   : where the forward trace is IPS_GOOD, ImmaturePoint(lastTraceUV, frame B) traced into frame A
 * The accept rule differs per caller (|u - back_uv[0]| < 1 and a depth bound), so it is left to the caller.
 * status_back is 255 where the back trace did not run.  Any output pointer may be NULL. */
typedef struct {
  int n;
  const float* u; const float* v;
  const float* idepth_min_stereo; const float* idepth_max_stereo;
  const float* back_idepth_min_stereo; const float* back_idepth_max_stereo;
  uint8_t* status_fwd; uint8_t* status_back;
  float* idepth_stereo; float* idepth_min_out; float* idepth_max_out;   /* of the forward trace */
  float* fwd_uv;   /* n*2 lastTraceUV of the forward trace */
  float* back_uv;  /* n*2 lastTraceUV of the back trace */
} sdso_stereo_match_t;
int sdso_stereo_match_batch(sdso_ctx* ctx, int slot_a, int slot_b, const float K[4], float baseline,
                            int mode_right_first, sdso_stereo_match_t* m);

/* ------------------------------------------------------------------ the fork's live g2o factors
 * gyubeomim/stereo-dso-g2o routes tracking, window optimisation and the sub-pixel trace refinement through g2o edges
 * (src/FullSystem/dso_g2o_edge.cpp, dso_g2o_vertex.cpp).  The edges' own arithmetic is specified by the reference tree and
 * is reproduced here exactly; g2o itself (robust kernel, quadratic form, Levenberg-Marquardt / Gauss-Newton control) is
 * neither vendored nor version-pinned (CMakeLists.txt:47-58) and is restated from its published algorithm — "parity
 * unpinned" for everything g2o decides.  `SE3 * Vec3` is evaluated as R*X + t with R = rotationMatrix(). */

/* S3: sub-pixel refinement used by every later traceStereo launch of this ctx (sdso_trace_stereo_*, sdso_stereo_match_batch):
 *   0  DSO-native GN, ImmaturePoint.cpp:707-769 (default)
 *   1  fork-live: VertexUVDSO + 8 EdgeTracePointUVDSO (dso_g2o_edge.cpp:571-619, dso_g2o_vertex.cpp:65-88), Huber(9),
 *      one g2o Gauss-Newton iteration per pass, 3 passes (ImmaturePoint.cpp:309-412) */
int sdso_trace_set_gn_mode(sdso_ctx* ctx, int mode);

/* T5: EdgeSE3PosePhotoDSO (dso_g2o_edge.cpp:395-500) at one evaluation point. */
typedef struct {
  int lvl, w, h;
  float fx, fy, cx, cy;      /* KG[level] (Matrix3f, globalCalib.cpp:90-107) = CoarseTracker fx[lvl]..cy[lvl] */
  float Ki[9];               /* Ki[lvl]                                              CoarseTracker.cpp:130  */
  float RKi[9], t_cull[3];   /* calcRes :617-618 from the pose calcRes is CALLED with (the fork always passes the
                                initial lastToNew_out, :890) — used for the border cull and the flow indicators only */
  double R[9], t[3];         /* VertexSE3PoseDSO::estimate(): rotationMatrix(), translation() */
  float ab[2];               /* AffLight::fromToVecExposure(ref_exposure, new_exposure, a0b0, VertexPhotometricDSO::estimate()).cast<float>() */
  double b0;                 /* a0b0_.b = lastRef_aff_g2l.b */
  float cutoffTH, huberTH;   /* setting_coarseCutoffTH (edges with error > 10*cutoffTH are dropped, :724), setting_huberTH */
} sdso_g2o_track_eval_t;
/* CoarseTracker::calcRes, fork-live body (CoarseTracker.cpp:600-792) for level ev->lvl of reference `ref_slot` against frame
 * `frame_slot`: culls the pc points, creates the edges, evaluates them at the vertices' estimates and keeps those that are
 * not saturated.  The edge set {mask, Xref, measurement} stays on the device, keyed by (ref_slot, level).
 *   res6     : the Vec6 calcRes returns (:783-789; E is never accumulated by the live body -> 0)
 *   n_edges  : numTermsInE;  edge_mask[pc_n], Xref[3*pc_n] optional copies (tests) */
int sdso_g2o_track_add_edges(sdso_ctx* ctx, int ref_slot, int frame_slot, const sdso_g2o_track_eval_t* ev, double* res6,
                             int* n_edges, uint8_t* edge_mask, float* Xref);
/* computeActiveErrors + linearizeOplus + g2o's robustified quadratic form over that edge set at the vertex estimates in ev:
 *   H[64] row-major over [pose 6 | photometric 2], b[8], chi2 = {sum e^2, sum rho0 (activeRobustChi2)};
 *   err[pc_n], J[8*pc_n] optional per-edge values (0 where the point carries no edge). */
int sdso_g2o_track_linearize(sdso_ctx* ctx, int ref_slot, int frame_slot, const sdso_g2o_track_eval_t* ev, double* H, double* b,
                             double* chi2, double* err, double* J);
/* CoarseTracker::trackNewestCoarse as the fork runs it (CoarseTracker.cpp:827-1069): per level add_edges, then g2o's
 * Levenberg-Marquardt (lambda0 0.01, 2 iterations, gain threshold 1e-3) on the 8-parameter system.  Same in/out convention
 * as sdso_track_newest_coarse; prm->maxIterations is ignored (the fork hard-codes {2,2,2,2,2}, :861). */
int sdso_g2o_track_newest_coarse(sdso_ctx* ctx, int ref_slot, int frame_slot, const sdso_track_params_t* prm,
                                 sdso_se3_t* lastToNew, sdso_aff_t* aff_g2l, sdso_track_result_t* out);

/* B13: EdgeLBASE3PosePhotoIdepthCamDSO (dso_g2o_edge.cpp:5-282) for the nr active residuals of a window, laid out as
 * FullSystem::optimize builds the graph (FullSystemOptimize.cpp:455-542): one pose + one photometric vertex per HOST frame,
 * one idepth vertex per RESIDUAL, one camera vertex; the target pose is a constant (PRE_worldToCam). */
typedef struct {
  int nf, nr, w, h;
  const int* frame_slot;         /* nf level-0 images */
  const float* const* dI;        /* unused by the library (kept so that the layout equals the test oracle's) */
  const float* pair_R;           /* [host*nf+target] 9: (Ttw * Twh).rotationMatrix().cast<float>()   :27-30 */
  const float* pair_t;           /* [host*nf+target] 3: translation */
  const float* pair_ab;          /* [host*nf+target] 2: fromToVecExposure(host, target, a0b0 = host vertex, a1b1 = target aff).cast<float>() :85-91 */
  const double* host_b0;         /* nf: b0_ (SetB, FullSystemOptimize.cpp:527-528) */
  const float* frameEnergyTH;    /* nf */
  double cam[4];                 /* VertexCamDSO::estimate(): fx fy cx cy */
  const int* host; const int* target;        /* nr */
  const float* u; const float* v;            /* nr: r->point->u, v */
  const double* idepth;          /* nr: VertexInverseDepthDSO::estimate() */
  const float* color; const float* weights;  /* nr*8: r->point->color (= measurement), weights */
} sdso_g2o_lba_t;
/* computeError + linearizeOplus for every residual:
 *   error[nr*8] (double), J[nr*8*13] rows = [xi 6 | photometric 2 | idepth 1 | camera 4] (zero where linearizeOplus returns early),
 *   state[nr] (0 IN, 1 OOB, 2 OUTLIER), energy[nr*2] = {state_NewEnergy, state_NewEnergyWithOutlier},
 *   centerProjectedTo[nr*3] ((2,2,0) if the centre pixel was not reached), idepth_hessian[nr], edge_level[nr] (1 = setLevel(1), :60-65) */
int sdso_g2o_lba_eval(sdso_ctx* ctx, const sdso_g2o_lba_t* L, double* error, double* J, uint8_t* state, float* energy,
                      float* centerProjectedTo, float* idepth_hessian, uint8_t* edge_level);

#ifdef __cplusplus
}
#endif
#endif /* SDSO_ABI_H */
