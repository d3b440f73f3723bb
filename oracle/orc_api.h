/* ORACLE — TEST INFRASTRUCTURE ONLY.
 * C API of liboracle.so: a dependency-free CPU restatement of the reference's DSO-native
 * arithmetic for the hot path.  It is the parity checker for libsdso_hip.so and
 * the "port" CPU baseline of bench.py; the product never links, loads or calls it.
 *
 * PARITY PINNING: the reference holds no golden vectors for this path and cannot be built here
 * (Eigen/Boost/g2o/OpenCV absent — SURVEY.md §8c).  The only reference test data, the Sophus SE3
 * group elements/tangents (thirdparty/Sophus/sophus/test_se3.cpp:40-82), pin se3 exp/log/Adj
 * (tests/test_oracle_se3.py).  Everything else is "parity unpinned": checked by finite-difference
 * Jacobians, double-precision re-evaluation and invariants only.
 *
 * The struct layouts mirror include/sdso_abi.h one-to-one so that the tests drive both libraries
 * with the same ctypes structures; the definitions are repeated here on purpose (the oracle does
 * not include product headers).
 */
#ifndef ORC_API_H
#define ORC_API_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int lvl;
  int w, h;
  float fx, fy, cx, cy;
  float Ki[9];
  float RKi[9];
  float t[3];
  float affLL[2];
  float ref_b0;
  float cutoffTH;
  float huberTH;
} orc_track_eval_t;

typedef struct { double R[9]; double t[3]; } orc_se3_t;
typedef struct { double a, b; } orc_aff_t;

typedef struct {
  int levels;
  int w[6], h[6];
  float fx[6], fy[6], cx[6], cy[6];
  float ref_exposure, new_exposure;
  orc_aff_t ref_aff_g2l;
  int coarsestLvl;
  double minResForAbort[5];
  float coarseCutoffTH;
  float huberTH;
  int maxIterations[5];
  double affineOptModeA, affineOptModeB;
} orc_track_params_t;

typedef struct {
  int good;
  double lastResiduals[5];
  double lastFlowIndicators[3];
  int iterations[5];
  int evaluations;
  long long point_evals;
} orc_track_result_t;

typedef struct {
  int nf, np, nr;
  int w, h;
  double calib_value_scaled[4];
  double calib_value_zero[4];
  const double* evalPT;
  const double* state;
  const double* state_zero;
  const float* ab_exposure;
  const float* frameEnergyTH;
  const int* frameID;
  const int* frame_slot;
  const float* const* dI;
  const float* u;
  const float* v;
  const float* idepth;
  const float* idepth_zero;
  const float* color;
  const float* weights;
  const int* host;
  const uint8_t* hasDepthPrior;
  const int* res_point;
  const int* res_target;
  const uint8_t* res_state;
  const double* HM;
  const double* bM;
  int solverMode;
  double affineOptModeA, affineOptModeB;
  int forceAcceptStep;
  const float* maxRelBaseline;     /* np, NULL: 0   PointHessian::maxRelBaseline   */
  const int* numGoodResiduals;     /* np, NULL: 0   PointHessian::numGoodResiduals */
  const uint8_t* res_isNew;        /* nr, NULL: 1   PointFrameResidual::isNew      */
} orc_ba_window_t;

typedef struct {
  int iterations;
  double lastEnergy;
  double rmse;
  int resInA;
} orc_ba_opt_result_t;

/* what FullSystem::optimize leaves behind (mirror of sdso_ba_post_state_t, include/sdso_abi.h) */
typedef struct {
  float* idepth; float* step; float* HdiF; float* bdSumF; float* idepth_hessian; float* maxRelBaseline; int* numGoodResiduals;
  uint8_t* state_state; uint8_t* isActiveAndIsGoodNEW; float* state_energy; float* centerProjectedTo; float* projectedTo; uint8_t* toRemove;
  double* state; double* state_zero; double* evalPT; double* PRE_worldToCam; double* frame_step; float* frameEnergyTH;
  double calib_value[4]; double calib_value_scaled[4]; double calib_step[4];
  double* lastX; double* lastHS; double* lastbS;
  int resInA, resInL, resInM;
  int n_toRemove;
  orc_ba_opt_result_t result;
} orc_ba_post_state_t;

typedef struct {
  int n;
  float* u_stereo; float* v_stereo;
  float* idepth_min;
  float* idepth_min_stereo; float* idepth_max_stereo; float* idepth_stereo;
  float* color;
  float* weights;
  float* gradH;
  float* energyTH;
  float* quality;
  uint8_t* lastTraceStatus;
  float* lastTraceUV;
  float* lastTracePixelInterval;
} orc_trace_points_t;

/* ---- math (Sophus restatement) */
void orc_se3_exp(const double xi[6], orc_se3_t* T);
void orc_se3_log(const orc_se3_t* T, double xi[6]);
void orc_se3_adj(const orc_se3_t* T, double A[36]);
void orc_se3_mul(const orc_se3_t* A, const orc_se3_t* B, orc_se3_t* C);
void orc_se3_inv(const orc_se3_t* A, orc_se3_t* C);
void orc_mat3f_inv(const float m[9], float inv[9]);
int orc_ldlt_solve(int n, const double* A, const double* rhs, double* x);

/* ---- pyramid: globalCalib.cpp:52-58 and HessianBlocks.cpp:141-203 */
int orc_pyramid_levels(int w, int h);
void orc_make_images(const float* color, int w, int h, int levels, float* const* dIp_out);

/* ---- tracker */
int orc_track_calc_res_gs(int n, const float* pc_u, const float* pc_v, const float* pc_idepth,
                          const float* pc_color, const float* dI, const orc_track_eval_t* ev,
                          double* H, double* b, double* res, int* n_warped, uint8_t* inlier_mask,
                          float* buf_warped /* optional 8*n_cap SoA: idepth,u,v,dx,dy,residual,weight,refColor */,
                          int n_cap);
/* host-side construction of one evaluation's parameters (CoarseTracker.cpp:108-136, :617-621) */
void orc_track_make_eval(const orc_track_params_t* prm, int lvl, const orc_se3_t* refToNew,
                         const orc_aff_t* aff_g2l, float levelCutoffRepeat, orc_track_eval_t* ev);
int orc_track_newest_coarse(const int* pc_n, const float* const* pc_u, const float* const* pc_v,
                            const float* const* pc_idepth, const float* const* pc_color,
                            const float* const* dIp, const orc_track_params_t* prm,
                            orc_se3_t* lastToNew, orc_aff_t* aff_g2l, orc_track_result_t* out);

/* CoarseTracker::makeCoarseDepthL0 from STEP1's splat on (CoarseTracker.cpp:352-534): n weighted inverse depths at integer pixels of
 * lastRef -> pc_n[lvl], pc_u / pc_v / pc_idepth / pc_color[lvl] (caller-allocated, w[lvl]*h[lvl] entries per level) */
int orc_make_coarse_depth(int levels, const int* w, const int* h, const float* const* dIp, int n, const int* u, const int* v,
                          const float* new_idepth, const float* weight, int* pc_n, float* const* pc_u, float* const* pc_v,
                          float* const* pc_idepth, float* const* pc_color);

/* test instrumentation: smallest relative margin of the LM loop's accept / stop / cut-off-repeat decisions per level of the last
 * orc_track_newest_coarse call (1e300 where a level took no decision) */
void orc_track_last_margins(double* out5);

/* ---- windowed BA (handle based; the handle deep-copies the window description) */
typedef struct orc_ba orc_ba;
orc_ba* orc_ba_create(const orc_ba_window_t* W);
void orc_ba_destroy(orc_ba* h);
/* CPU baseline only: run linearizeAll / applyRes / accumulate A,L,SC / stitch / resubstitute on n persistent workers the way the
 * reference's IndexThreadReduce does (src/util/IndexThreadReduce.h:34-196; NUM_THREADS = 6 there, util/NumType.h:38): index chunks
 * from a shared counter (50 points per chunk for the accumulators, EnergyFunctional.cpp:216-217), one private accumulator copy per
 * worker, copies summed in double in the stitch (AccumulatedTopHessian.cpp:299-308).  n <= 1: single thread (the parity path). */
int orc_ba_set_threads(orc_ba* h, int n);   /* returns the number of workers that were pinned to a core of their own */
int orc_ba_linearize(orc_ba* h, double* energy);
int orc_ba_get_linearization(orc_ba* h, float* J, uint8_t* newState, float* newEnergy,
                             float* newEnergyWithOutlier, float* projectedTo,
                             float* centerProjectedTo);
int orc_ba_apply_res(orc_ba* h);
int orc_ba_get_ef_jacobians(orc_ba* h, float* J /* nr*74: EFResidual::J */);
int orc_ba_get_residual_state(orc_ba* h, uint8_t* state, uint8_t* isActive, float* JpJdF);
int orc_ba_accumulate(orc_ba* h);
int orc_ba_accum_floats(int nf);
int orc_ba_get_accumulators(orc_ba* h, float* packed);
/* TRUTH MODE (not in the reference): orc_set_acc64(1) makes every later stitch read the BA accumulators' double shadow sums
 * (double products of the same float Jacobian entries, double accumulation) instead of the float 3-tier sums; the float path
 * itself is unchanged.  orc_ba_get_accumulators_f64 returns the packed block in double (shadow sums when the mode is on). */
void orc_set_acc64(int on);
int orc_ba_get_accumulators_f64(orc_ba* h, double* packed);
int orc_ba_get_point_terms(orc_ba* h, float* HdiF, float* bdSumF, float* Hdd_accAF, float* bd_accAF,
                           float* Hcd_accAF);
int orc_ba_solve(orc_ba* h, int iteration, double lambda, double* x, double* HS, double* bS,
                 double* frame_step, double* calib_step);
/* accumulate{AF,LF,SCF}_MT (EnergyFunctional.cpp:212-269): the stitched top-A (no priors), top-L (with priors) and Schur systems of the
 * accumulators as they stand after orc_ba_accumulate; n x n row-major + n each, any pointer may be NULL */
int orc_ba_get_stitched(orc_ba* h, double* HA, double* bA, double* HL, double* bL, double* Hsc, double* bsc);
int orc_ba_get_point_steps(orc_ba* h, float* step);
int orc_ba_optimize(orc_ba* h, int mnumOptIts, double* state_out, float* idepth_out,
                    uint8_t* res_state_out, orc_ba_opt_result_t* out);
/* the post-state of the last orc_ba_optimize: FullSystemOptimize.cpp:52-87, :142-203, :997-1041, AccumulatedSCHessian.cpp:34-60 */
int orc_ba_get_post_state(orc_ba* h, orc_ba_post_state_t* out);
/* lastX of every GN iteration of the latest orc_ba_optimize (iteration-major, 8nf+4 doubles each); returns the number of iterations */
int orc_ba_get_x_trace(orc_ba* h, double* x, int cap_iterations);
/* the loop's stepsize of every iteration (SOLVER_STEPMOMENTUM, FullSystemOptimize.cpp:936-948); returns the number of iterations */
int orc_ba_get_step_trace(orc_ba* h, float* stepsize, int cap_iterations);
/* calcLEnergyF_MT / calcMEnergyF (EnergyFunctional.cpp:420-442, :344-351) and what setDeltaF leaves behind (:173-207) */
int orc_ba_calc_energies(orc_ba* h, double* EL, double* EM);
int orc_ba_get_deltas(orc_ba* h, float* cDeltaF, double* frame_delta, double* frame_delta_prior, float* point_deltaF);
int orc_ba_marginalize_points(orc_ba* h, const uint8_t* marg_flag, double* HM_out, double* bM_out);
/* host tables the product also derives (for table-level parity): precalc nf*nf*27 floats
 * {KRKi9,Kt3,R0 9,t0 3,aff2,b0 1}, adHost/adTarget nf*nf*64 doubles, adHTdeltaF nf*nf*8 floats */
int orc_ba_get_tables(orc_ba* h, float* precalc, double* adHost, double* adTarget, float* adHTdeltaF);

/* ---- static stereo */
int orc_immature_init_batch(const float* dI, int w, int h, int n, const float* u, const float* v,
                            float* color, float* weights, float* gradH, float* energyTH);
/* PixelSelector::makeMaps (PixelSelector2.cpp:193-300) with makeHists / select; dIp = levels 0..2 (AoS float3) */
int orc_pixel_select(const float* const* dIp, int w, int h, float density, int recursionsLeft, float thFactor, int* potential,
                     float* map_out);
void orc_selector_random_pattern(int n, unsigned char* out);
/* CalibHessian::B (256 floats) for the gamma-weighted absSquaredGrad of orc_pixel_select (HessianBlocks.cpp:194-198); NULL = identity */
void orc_set_gamma(const float* B);
void orc_gamma_from_binv(const float* BInv, float* B);

/* EnergyFunctional::marginalizeFrame (EnergyFunctional.cpp:554-660) */
int orc_marginalize_frame(int nf, int idx, const double* prior8, const double* delta_prior8, const double* HM_in, const double* bM_in,
                          double* HM_out, double* bM_out);

/* FullSystem::optimizeImmaturePoint (FullSystemOptPoint.cpp:52-238), DSO-native; same layout as sdso_activate_t + host images */
typedef struct {
  int nf, w, h, n, minObs;
  float K[4];
  const float* pair_R; const float* pair_t; const float* pair_aff;   /* [host*nf+target] PRE_RTll 9, PRE_tTll 3, PRE_aff_mode 2 */
  const int* frame_slot;            /* product only */
  const float* const* dI;           /* oracle only: nf level-0 images, AoS float3 */
  const int* host; const float* u; const float* v; const float* idepth_min; const float* idepth_max;
  const float* color; const float* weights; const float* energyTH;
} orc_activate_t;
int orc_activate_points(const orc_activate_t* A, int8_t* status, float* idepth_out, uint8_t* res_state);

/* hostToFrame geometry of traceOn (FullSystem.cpp:654-665, :760-766) */
typedef struct { float KRKi[9]; float Kt[3]; float aff[2]; } orc_trace_geom_t;
/* ImmaturePoint::traceOn (ImmaturePoint.cpp:459-828); pts->u_stereo/v_stereo = u/v, idepth_min_stereo/idepth_max_stereo = idepth_min/idepth_max */
int orc_trace_on_batch(const float* dI, int w, int h, int ngeom, const orc_trace_geom_t* geom, const int* point_geom,
                       orc_trace_points_t* pts, uint8_t* status);
int orc_trace_stereo_batch(const float* dI, int w, int h, const float K[4], float baseline,
                           int mode_right, orc_trace_points_t* pts, uint8_t* status);

int orc_trace_stereo_batch_gn(const float* dI, int w, int h, const float K[4], float baseline,
                              int mode_right, orc_trace_points_t* pts, uint8_t* status, int gn_mode);

/* ---- the fork's live g2o factors (orc_g2o.cpp) */
typedef struct {
  int lvl, w, h;
  float fx, fy, cx, cy;      /* KG[level] (float, globalCalib.cpp:90-107) */
  float Ki[9];               /* Ki[lvl] */
  float RKi[9], t_cull[3];   /* calcRes :617-618, from the pose calcRes is called with */
  double R[9], t[3];         /* VertexSE3PoseDSO::estimate() */
  float ab[2];               /* fromToVecExposure(ref, new, a0b0, VertexPhotometricDSO::estimate()).cast<float>() */
  double b0;                 /* a0b0_.b = lastRef_aff_g2l.b */
  float cutoffTH, huberTH;
} orc_g2o_track_eval_t;
int orc_g2o_track_add_edges(int n, const float* pc_u, const float* pc_v, const float* pc_idepth, const float* pc_color,
                            const float* dI, const orc_g2o_track_eval_t* ev, double* res6, uint8_t* edge_mask, float* Xref);
int orc_g2o_track_linearize(int n, const uint8_t* edge_mask, const float* Xref, const float* pc_color, const float* dI,
                            const orc_g2o_track_eval_t* ev, double* H, double* b, double* chi2, double* err, double* J);
int orc_g2o_track_newest_coarse(const int* pc_n, const float* const* pc_u, const float* const* pc_v,
                                const float* const* pc_idepth, const float* const* pc_color,
                                const float* const* dIp, const orc_track_params_t* prm,
                                orc_se3_t* lastToNew, orc_aff_t* aff_g2l, orc_track_result_t* out);
typedef struct {
  int nf, nr, w, h;
  const int* frame_slot;         /* product only */
  const float* const* dI;        /* oracle only: nf level-0 images, AoS float3 */
  const float* pair_R;           /* [host*nf+target] (Ttw * Twh).rotationMatrix().cast<float>() */
  const float* pair_t;           /* [host*nf+target] translation */
  const float* pair_ab;          /* [host*nf+target] fromToVecExposure(host, target, a0b0, a1b1).cast<float>() */
  const double* host_b0;         /* nf: b0_ (EdgeLBA...::SetB, FullSystemOptimize.cpp:527-528) */
  const float* frameEnergyTH;    /* nf */
  double cam[4];                 /* VertexCamDSO::estimate() fx fy cx cy */
  const int* host; const int* target;
  const float* u; const float* v;
  const double* idepth;          /* VertexInverseDepthDSO::estimate(), one vertex per residual */
  const float* color; const float* weights;
} orc_g2o_lba_t;
int orc_g2o_lba_eval(const orc_g2o_lba_t* L, double* error, double* J, uint8_t* state, float* energy,
                     float* centerProjectedTo, float* idepth_hessian, uint8_t* edge_level);


#ifdef __cplusplus
}
#endif
#endif
