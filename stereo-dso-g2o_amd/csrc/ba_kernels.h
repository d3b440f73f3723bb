// Device side of the windowed BA (gfx950).  One BaDev describes one EnergyFunctional window that
// is resident in HBM; kernels take an array of BaDev and use blockIdx.y as the window index so
// that any number of independent windows run in ONE launch.
//
// Residual order on the device is (host,target)-pair sorted (htIDX = host + target*nf, stable in
// the reference's residualsAll order): every workgroup of the linearize / accumulate kernels
// sees one pair -> wave-uniform precalc tables, one target image, and a plain block reduction of
// the 91 AccumulatorApprox sums (no atomics).  Points keep the reference's allPoints order and
// reach their residuals through 64-byte per-residual records.
//
// Per-residual / per-point float arithmetic follows the reference's operation order (compiled
// with -ffp-contract=off), so J, energies, states, HdiF, steps are bit-identical to the CPU path;
// only cross-residual sums differ in order.
#pragma once
#include "sdso_internal.h"

namespace sdso {

// setting_solverMode bits (src/util/settings.h:32-43)
constexpr int SOLVER_SVD_CUT7 = 16;
constexpr int SOLVER_SVD = 1, SOLVER_ORTHOGONALIZE_SYSTEM = 2, SOLVER_ORTHOGONALIZE_POINTMARG = 4, SOLVER_ORTHOGONALIZE_FULL = 8,
              SOLVER_REMOVE_POSEPRIOR = 32, SOLVER_USE_GN = 64, SOLVER_FIX_LAMBDA = 128, SOLVER_ORTHOGONALIZE_X = 256,
              SOLVER_MOMENTUM = 512, SOLVER_STEPMOMENTUM = 1024, SOLVER_ORTHOGONALIZE_X_LATER = 2048;

constexpr int BA_BLOCK = 256;
constexpr int BA_CHUNK = 256;      // residuals per accumulate workgroup
constexpr int BA_SC_PTS = 32;      // points per SC wave item (<= 64)
constexpr int J_RESF = 0, J_XI0 = 8, J_XI1 = 14, J_C0 = 20, J_C1 = 24, J_DD = 28, J_IDX0 = 30, J_IDX1 = 38, J_AB0 = 46, J_AB1 = 54,
              J_IDX2 = 62, J_ABIDX = 66, J_AB2 = 70;
// p_out record (16 floats per point)
// [0..7] is what a plain Gauss-Newton iteration writes (one 32-byte piece per point), [8..13] the sums of the linearised / marginalised
// residuals (zero unless such residuals exist), [14..15] the back-substitution's own
constexpr int PO_HCD_A = 0, PO_HDD_A = 4, PO_BD_A = 5, PO_HDI = 6, PO_BDSUM = 7, PO_HCD_L = 8, PO_HDD_L = 12, PO_BD_L = 13, PO_STEP = 14, PO_BACKUP = 15;
// r_rec record (16 floats per residual): JpJdF[8], bd, Hdd, Hcd[4], flags (bit 0 active, bit 1 linearized), -
constexpr int RR_BD = 8, RR_HDD = 9, RR_HCD = 10, RR_FLAGS = 14;

// Device-resident state of FullSystem::optimize's Gauss-Newton loop for one window (ba_opt.hip): what the reference keeps in
// FrameHessian (state / state_backup / state_zero / worldToCam_evalPT / PRE_worldToCam, HessianBlocks.h:120-190) and
// CalibHessian (value / value_backup / value_zero, :276-349), plus the loop's own scalars.
struct BaOptDev {
  double state[8][10], state_backup[8][10], state_zero[8][10];
  double evalPT[8][12];          // worldToCam_evalPT: R (9, row-major) then t (3)
  double calib_value[4], calib_backup[4], calib_zero[4];
  float ab_exposure[8];
  double lastEnergy;             // energy of the latest linearizeAll
  float frameTH_new;             // frameEnergyTH of the newest frame after the latest setNewFrameEnergyTH
  int iterations;                // GN iterations run
  int phase;                     // 0 running; 1 the break test fired: the linearisation at the final state is still to be consumed; 2 finished;
                                 // 3 (energy-gated flow) an accepted step fired the break test: finished once its applyRes has run
  int resInA;                    // nres[0] of the latest accumulate
  int newest_first;              // first (pair-sorted) residual whose target is the newest frame; they run to nr
  // ---- energy-gated flow (setting_forceAceptStep = false, FullSystemOptimize.cpp:969-990)
  double lastEnergyL, lastEnergyM;   // calcLEnergy / calcMEnergy of the accepted state
  double lambda;                     // the loop's lambda: *0.25 on an accepted step, *100 on a rejected one
  int gate;                          // decision of the current iteration (k_ba_opt_gate): 1 accepted, 2 rejected; read by the conditional kernels
  int canbreak;                      // doStepFromBackup's return value of the current iteration
  // what loadSateBackup + setPrecalcValues restore on a rejected step
  float bk_precalc[64 * 27], bk_adHTdelta[64 * 8], bk_cdelta[4], bk_calib[6];
  double bk_prior[8 * 16 + 4 + 8 * 8 + 4];
  // ---- SOLVER_STEPMOMENTUM / SOLVER_MOMENTUM (k_ba_opt_momentum, ba_opt.hip)
  float stepsize;                    // the loop's stepsize (FullSystemOptimize.cpp:928, :936-948); 1 unless SOLVER_STEPMOMENTUM moves it
  double previousX[72];              // ef->lastX of the previous iteration (:929, :934), NaN before the first
  double x_backup[72];               // -step_backup of the frames and the calibration: the previous iteration's x, zeros in the first
                                     // (backupState(iteration != 0), :311-345)
};

struct BaDev {
  int nf, np, nr, nrp, w, h, nchunks, nitems, n;
  int tiledT;               // > 0: t_img are 4x2-tiled level-0 images with tiledT tiles per row; 0: row-major
  float wM3, hM3, fxl, fyl, cxl, cyl, fxli, fyli;
  int affA_fixed, affB_fixed;
  int jfix;                 // 1 (default): the fused kernel refreshes EFResidual::J IN PLACE (J[jsel]) instead of writing the other buffer and
                            // swapping the roles like PointFrameResidual::applyRes (Residuals.cpp:367-385) — in the accepted-step flow the
                            // swapped-out copy is dead until the next linearize overwrites it, and alternating between two 4 MB record buffers
                            // per window costs the kernel 5-15 % (measured: every other step 255 vs 305 us, 235 us in place; profiles/).
                            // 0: swap (SDSO_BA_JSWAP=1)
  // points
  float4* p_geo;            // u, v, idepth, idepth_zero
  const float* p_color;     // np*8
  const float* p_weights;   // np*8
  const int* p_host;
  float* p_prior;
  float* p_delta;
  const int* p_rbeg;        // np + 1: first residual (window order) of every point; p_rbeg[np] = nr
  const int* p_rcnt;
  const int* p_rlist;       // sorted residual index of every (point, slot)   (host bookkeeping / debug)
  const unsigned* p_order;  // per point: nibble k = target frame of its k-th residual in EFPoint::residualsAll order, 0xF past the end.  The
                            // reference's per-point loops (AccumulatedTopHessian.cpp:50-172, AccumulatedSCHessian.cpp:36-101, EnergyFunctional.cpp:
                            // 324-338) run in THAT order, which is target order only until dropResidual swaps the last entry into a freed slot
                            // (EnergyFunctional.cpp:524-533); the dense [target] records are visited through this word so that the float sums
                            // Hdd / bd / Hcd and the back-substitution come out bit-identical on such windows too
  float4* p_track;          // per point: x PointHessian::maxRelBaseline, y numGoodResiduals (int bits) — FullSystemOptimize.cpp:64-77 —,
                            // z idepth_hessian, w the active residuals of the latest AccumulatedSCHessianSSE::addPoint (bit t: target t; int bits)
  float* p_out;             // np*16
  float* p_stepbk;          // np: PointHessian::step_backup (SOLVER_MOMENTUM, FullSystemOptimize.cpp:311-345): the step the latest back-substitution
                            // replaced — written by k_ba_resub itself, zeros in a loop's first iteration
  // residuals (pair-sorted)
  const int* r_point;
  const int* r_orig;        // original (window-order) index of the pair-sorted residual
  const uint8_t* r_host;
  const uint8_t* r_target;
  uint8_t* r_state; uint8_t* r_newState; uint8_t* r_lin; uint8_t* r_act; uint8_t* r_jsel;
  const uint8_t* r_isnew;   // PointFrameResidual::isNew
  float* r_energy; float* r_newEnergy; float* r_newEnergyWO;
  float* J[2];              // 19 float4 groups x nrp each (layout: ba_kernels.hip); EFResidual::J = J[jsel], PointFrameResidual::J = J[1-jsel]
  float* r_toZero;          // 8 x nrp SoA
  // Per-residual records of the Schur part, in the WINDOW's residual order (grouped by point, EFPoint::residualsAll order inside a point:
  // record o belongs to point res_point[o]; point p owns records [p_rbeg[p], p_rbeg[p + 1]), nibble k of p_order[p] names the target of its
  // k-th record) — no slot for a target the point does not observe.  Zeros where the residual is not active (the sums add / multiply
  // records without looking at flags).  One 64-byte piece per residual: the linearisation's scattered store stays ONE half line (two
  // 32-byte pieces in two arrays cost k_ba_lin_fused 6 %: measured, profiles/README.md round 5).
  float* r_rec;             // nr x 16: EFResidual::JpJdF (EnergyFunctionalStructs.cpp:37-51), then RR_*: the residual's terms of Hdd / bd / Hcd
                            // (AccumulatedTopHessian.cpp:160-172) and its flags
  float* r_cj;              // nr x 8: the JpJdF halves alone, written by k_ba_sc_host while it has the records in LDS (its launches are latency-
                            // bound, the coalesced stores ride along) so that the back-substitution streams 32 instead of 64 bytes per residual
  float* r_proj;            // nr x 19 (projectedTo 16, centerProjectedTo 3)
  // tables
  const float* t_precalc;   // [host*nf+target][27]
  const float* t_adHTdelta; // [h+t*nf][8]
  const float* t_cdelta;    // 4
  float* t_frameTH;         // nf
  const float4* const* t_img;  // nf level-0 images
  const double* t_adHost; const double* t_adTarget;  // [h+t*nf][64]
  const float* t_xAd;       // [h*nf+t][8]  (resubstitute)
  const double* t_prior;    // nf*8 prior, nf*8 delta_prior, 4 cPrior, then delta (4+8nf)
  const double* t_HM; const double* t_bM; const double* t_P;   // marginalisation prior, nullspace projector
  // work lists
  const int4* chunks;       // {pair, start, count, 0}
  const int* pair_chunk_beg;  // nf*nf+1
  const int4* items;        // {host, pbeg, pend, 0}
  const int* host_item_beg; // nf+1
  int host_pt_beg[9];       // points of host h: [host_pt_beg[h], host_pt_beg[h + 1]) (points are in allPoints order: grouped by host)
  double* top_part;         // nchunks x 92 (91 sums + count), f64: rounded to float once, when a pair's chunks are folded
  float* sc_part;           // nf x 20: Hcc (16) and bc (4) of every host's Schur workgroup; the fold adds the hosts
  double* e_part;           // energy partials of linearize (per workgroup)
  float* accum;             // packed accumulators (see sdso_ba_accum_floats)
  double* sol;              // Htop_A n*n | btop_A n | Htop_L n*n | btop_L n | Hsc n*n | bsc n | x n | HS n*n | bS n
  BaOptDev* opt;            // resident optimizer state (ba_opt.hip)
  int solver_mode;          // setting_solverMode of the window (k_ba_solve_alt: which branch of solveSystemF)
  int have_first_frame;     // a frame with frameID == 0 is in the window (EnergyFunctional.cpp:881-884: then HT / bT are not projected)
  int finished;             // resident GN loop (k_ba_opt_step): 1 the break test fired (only the linearisation at the final state is
                            // still wanted), 2 the loop has ended; cleared by k_ba_opt_release
};
// the remaining iterations of a batch skip a window whose resident GN loop has ended
__device__ __forceinline__ bool ba_finished(const BaDev& B) { return B.finished >= 1; }       // accumulate / solve / step kernels
__device__ __forceinline__ bool ba_finished_lin(const BaDev& B) { return B.finished >= 2; }   // the linearisation runs once more
// conditional kernels of the energy-gated loop: cond = 0 always, 1 only when the step was accepted, 2 only when it was rejected
__device__ __forceinline__ bool ba_gate_skip(const BaDev& B, int cond) { return cond != 0 && (ba_finished_lin(B) || B.opt->gate != cond); }

// Float offset of group g (0..18, one float4) of residual i in a RawResidualJacobian buffer of S = nrp residual slots.
//   blocked (default): the 19 groups of 64 consecutive residuals form ONE contiguous 19 KiB block — a wave of the fused kernel streams
//                      its records into one stretch of memory instead of into 19 streams 4 S floats apart;
//   -DSDSO_J_SOA     : group-major over the whole window (rounds 1-2), kept for A/B.
__host__ __device__ inline size_t j_off(int S, int i, int g) {
#ifdef SDSO_J_SOA
  return (size_t)g * 4 * S + 4 * (size_t)i;
#else
  (void)S;
  return ((size_t)(i >> 6) * 19 + g) * 256 + (size_t)(i & 63) * 4;
#endif
}

__host__ __device__ inline size_t acc_off_topA(int nf) { return 0; }
__host__ __device__ inline size_t acc_off_topL(int nf) { return (size_t)nf * nf * 91; }
__host__ __device__ inline size_t acc_off_D(int nf) { return (size_t)nf * nf * 91 * 2; }
__host__ __device__ inline size_t acc_off_E(int nf) { return acc_off_D(nf) + (size_t)nf * nf * nf * 64; }
__host__ __device__ inline size_t acc_off_EB(int nf) { return acc_off_E(nf) + (size_t)nf * nf * 32; }
__host__ __device__ inline size_t acc_off_Hcc(int nf) { return acc_off_EB(nf) + (size_t)nf * nf * 8; }
__host__ __device__ inline size_t acc_off_bc(int nf) { return acc_off_Hcc(nf) + 16; }
__host__ __device__ inline size_t acc_off_nres(int nf) { return acc_off_bc(nf) + 4; }
__host__ __device__ inline size_t acc_floats(int nf) { return acc_off_nres(nf) + 2; }

}  // namespace sdso
