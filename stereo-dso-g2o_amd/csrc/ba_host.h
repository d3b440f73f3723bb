// Host mirror of one EnergyFunctional window (frames + calibration) and the small double-precision
// tables the reference derives on the CPU between kernel phases:
//   FrameHessian::setState / setStateZero / setEvalPT / getPrior   src/FullSystem/HessianBlocks.{h,cpp}
//   FrameFramePrecalc::set                                          HessianBlocks.cpp:206-242
//   EnergyFunctional::setAdjointsF / setDeltaF                      EnergyFunctional.cpp:41-119, :173-207
//   FullSystem::getNullspaces + EnergyFunctional::orthogonalize     FullSystemOptimize.cpp:1087-1147, EnergyFunctional.cpp:775-835
//   CalibHessian::setValue / setValueScaled                         HessianBlocks.h:318-349
#pragma once
#include "host_math.h"
#include "sdso_internal.h"

namespace sdso {

constexpr float kInitialRotPrior = 1e11f, kInitialTransPrior = 1e10f, kInitialAffBPrior = 1e14f, kInitialAffAPrior = 1e14f;
constexpr float kInitialCalibHessian = 5e9f;
constexpr double kSolverModeDelta = 0.00001;
// (the SOLVER_* bits of setting_solverMode: ba_kernels.h — the kernels read some of them too)

struct HostCalib {
  double value_zero[4], value_scaled[4], value[4], step[4] = {0, 0, 0, 0}, value_backup[4], value_minus_value_zero[4];
  float value_scaledf[4], value_scaledi[4];
  void finish() {
    for (int i = 0; i < 4; i++) value_scaledf[i] = (float)value_scaled[i];
    value_scaledi[0] = 1.0f / value_scaledf[0];
    value_scaledi[1] = 1.0f / value_scaledf[1];
    value_scaledi[2] = -value_scaledf[2] / value_scaledf[0];
    value_scaledi[3] = -value_scaledf[3] / value_scaledf[1];
    for (int i = 0; i < 4; i++) value_minus_value_zero[i] = value[i] - value_zero[i];
  }
  void setValue(const double* v) {
    for (int i = 0; i < 4; i++) value[i] = v[i];
    value_scaled[0] = SCALE_F * value[0]; value_scaled[1] = SCALE_F * value[1];
    value_scaled[2] = SCALE_C * value[2]; value_scaled[3] = SCALE_C * value[3];
    finish();
  }
  void setValueScaled(const double* vs) {
    for (int i = 0; i < 4; i++) value_scaled[i] = vs[i];
    value[0] = (1.0f / SCALE_F) * value_scaled[0]; value[1] = (1.0f / SCALE_F) * value_scaled[1];
    value[2] = (1.0f / SCALE_C) * value_scaled[2]; value[3] = (1.0f / SCALE_C) * value_scaled[3];
    finish();
  }
};

struct HostFrame {
  Se3 evalPT, PRE_worldToCam, PRE_camToWorld;
  double state_zero[10], state_scaled[10], state[10], step[10], state_backup[10], step_backup[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  float ab_exposure = 1, frameEnergyTH = 512;
  int frameID = 0, frame_slot = -1;
  double ns_pose[6][6], ns_scale[6];
  double prior[8], delta_prior[8], delta[8];
  double aff_a() const { return state_scaled[6]; }
  double aff_b() const { return state_scaled[7]; }
  double aff0_a() const { return state_zero[6] * SCALE_A; }
  double aff0_b() const { return state_zero[7] * SCALE_B; }
  void setState(const double* s) {
    for (int i = 0; i < 10; i++) state[i] = s[i];
    for (int i = 0; i < 3; i++) state_scaled[i] = SCALE_XI_TRANS * state[i];
    for (int i = 3; i < 6; i++) state_scaled[i] = SCALE_XI_ROT * state[i];
    state_scaled[6] = SCALE_A * state[6]; state_scaled[7] = SCALE_B * state[7];
    state_scaled[8] = SCALE_A * state[8]; state_scaled[9] = SCALE_B * state[9];
    PRE_worldToCam = expSe3(state_scaled) * evalPT;
    PRE_camToWorld = inverse(PRE_worldToCam);
  }
  void setStateZero(const double* sz) {
    for (int i = 0; i < 10; i++) state_zero[i] = sz[i];
    const Se3 Ti = inverse(evalPT);
    for (int i = 0; i < 6; i++) {
      double e[6] = {0, 0, 0, 0, 0, 0};
      e[i] = 1e-3;
      const Se3 P = (evalPT * expSe3(e)) * Ti;
      e[i] = -1e-3;
      const Se3 M = (evalPT * expSe3(e)) * Ti;
      double lp[6], lm[6];
      logSe3(P, lp); logSe3(M, lm);
      for (int r = 0; r < 6; r++) ns_pose[r][i] = (lp[r] - lm[r]) / (2e-3);
    }
    Se3 P = evalPT, M = evalPT;
    for (int i = 0; i < 3; i++) { P.t[i] *= 1.00001; M.t[i] /= 1.00001; }
    P = P * Ti; M = M * Ti;
    double lp[6], lm[6];
    logSe3(P, lp); logSe3(M, lm);
    for (int r = 0; r < 6; r++) ns_scale[r] = (lp[r] - lm[r]) / (2e-3);
  }
  void setEvalPT(const Se3& T, const double* s) { evalPT = T; setState(s); setStateZero(s); }
  void fillPrior(double optA, double optB, int solverMode) {
    double p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (frameID == 0) {
      for (int i = 0; i < 3; i++) p[i] = kInitialTransPrior;
      for (int i = 3; i < 6; i++) p[i] = kInitialRotPrior;
      if (solverMode & SOLVER_REMOVE_POSEPRIOR) for (int i = 0; i < 6; i++) p[i] = 0;
      p[6] = kInitialAffAPrior; p[7] = kInitialAffBPrior;
    } else {
      p[6] = optA < 0 ? (double)kInitialAffAPrior : optA;
      p[7] = optB < 0 ? (double)kInitialAffBPrior : optB;
    }
    for (int i = 0; i < 8; i++) prior[i] = p[i];
  }
};

// everything that changes when a frame state / calibration changes
struct HostTables {
  std::vector<float> precalc;     // [host*nf+target][27]
  std::vector<double> adHost, adTarget;   // [h+t*nf][64]
  std::vector<float> adHostF, adTargetF;
  std::vector<float> adHTdeltaF;  // [h+t*nf][8]
  float cDeltaF[4];
  double cPrior[4];
};

inline void buildPrecalc(const HostCalib& C, const std::vector<HostFrame>& F, HostTables& T) {
  const int nf = (int)F.size();
  T.precalc.assign((size_t)nf * nf * 27, 0.f);
  const float K[9] = {C.value_scaledf[0], 0, C.value_scaledf[2], 0, C.value_scaledf[1], C.value_scaledf[3], 0, 0, 1};
  float Ki[9];
  inv3f(K, Ki);
  for (int h = 0; h < nf; h++)
    for (int t = 0; t < nf; t++) {
      float* o = &T.precalc[(size_t)(h * nf + t) * 27];
      const Se3 l0 = F[t].evalPT * inverse(F[h].evalPT);
      const Se3 l = F[t].PRE_worldToCam * F[h].PRE_camToWorld;
      float R[9], tt[3], KR[9];
      for (int i = 0; i < 9; i++) { R[i] = (float)l.R[i]; o[12 + i] = (float)l0.R[i]; }
      for (int i = 0; i < 3; i++) { tt[i] = (float)l.t[i]; o[21 + i] = (float)l0.t[i]; }
      mul3f(K, R, KR);
      mul3f(KR, Ki, o);          // PRE_KRKiTll
      mulv3f(K, tt, o + 9);      // PRE_KtTll
      double a2[2];
      affFromTo(F[h].ab_exposure, F[t].ab_exposure, F[h].aff_a(), F[h].aff_b(), F[t].aff_a(), F[t].aff_b(), a2);
      o[24] = (float)a2[0]; o[25] = (float)a2[1];
      o[26] = (float)F[h].aff0_b();
    }
}

inline void buildAdjoints(const std::vector<HostFrame>& F, HostTables& T) {
  const int nf = (int)F.size();
  T.adHost.assign((size_t)nf * nf * 64, 0.0);
  T.adTarget.assign((size_t)nf * nf * 64, 0.0);
  for (int h = 0; h < nf; h++)
    for (int t = 0; t < nf; t++) {
      const Se3 h2t = F[t].evalPT * inverse(F[h].evalPT);
      double Ad[36];
      adjoint(h2t, Ad);
      double AH[64] = {0}, AT[64] = {0};
      for (int i = 0; i < 8; i++) AH[i * 8 + i] = AT[i * 8 + i] = 1;
      for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) AH[i * 8 + j] = -Ad[j * 6 + i];
      double a2[2];
      affFromTo(F[h].ab_exposure, F[t].ab_exposure, F[h].aff0_a(), F[h].aff0_b(), F[t].aff0_a(), F[t].aff0_b(), a2);
      const float a0 = (float)a2[0];
      AT[6 * 8 + 6] = -a0; AT[7 * 8 + 7] = -1;
      AH[6 * 8 + 6] = a0; AH[7 * 8 + 7] = a0;
      for (int j = 0; j < 8; j++) {
        for (int i = 0; i < 3; i++) { AH[i * 8 + j] *= SCALE_XI_TRANS; AT[i * 8 + j] *= SCALE_XI_TRANS; }
        for (int i = 3; i < 6; i++) { AH[i * 8 + j] *= SCALE_XI_ROT; AT[i * 8 + j] *= SCALE_XI_ROT; }
        AH[6 * 8 + j] *= SCALE_A; AT[6 * 8 + j] *= SCALE_A;
        AH[7 * 8 + j] *= SCALE_B; AT[7 * 8 + j] *= SCALE_B;
      }
      std::memcpy(&T.adHost[(size_t)(h + t * nf) * 64], AH, sizeof(AH));
      std::memcpy(&T.adTarget[(size_t)(h + t * nf) * 64], AT, sizeof(AT));
    }
  T.adHostF.resize(T.adHost.size()); T.adTargetF.resize(T.adTarget.size());
  for (size_t i = 0; i < T.adHost.size(); i++) { T.adHostF[i] = (float)T.adHost[i]; T.adTargetF[i] = (float)T.adTarget[i]; }
  for (int i = 0; i < 4; i++) T.cPrior[i] = kInitialCalibHessian;
}

inline void buildDelta(const HostCalib& C, std::vector<HostFrame>& F, HostTables& T) {
  const int nf = (int)F.size();
  T.adHTdeltaF.assign((size_t)nf * nf * 8, 0.f);
  for (int h = 0; h < nf; h++)
    for (int t = 0; t < nf; t++) {
      const int idx = h + t * nf;
      float dh[8], dt[8];
      for (int i = 0; i < 8; i++) { dh[i] = (float)(F[h].state[i] - F[h].state_zero[i]); dt[i] = (float)(F[t].state[i] - F[t].state_zero[i]); }
      for (int j = 0; j < 8; j++) {
        float sh = 0, st = 0;
        for (int i = 0; i < 8; i++) { sh += dh[i] * T.adHostF[(size_t)idx * 64 + i * 8 + j]; st += dt[i] * T.adTargetF[(size_t)idx * 64 + i * 8 + j]; }
        T.adHTdeltaF[(size_t)idx * 8 + j] = sh + st;
      }
    }
  for (int i = 0; i < 4; i++) T.cDeltaF[i] = (float)C.value_minus_value_zero[i];
  for (HostFrame& f : F)
    for (int i = 0; i < 8; i++) { f.delta[i] = f.state[i] - f.state_zero[i]; f.delta_prior[i] = f.state[i]; }
}

// N*pinv(N) over the 7 gauge directions (6 pose + scale), EnergyFunctional.cpp:775-835
inline Dense buildNullspaceProjector(const std::vector<HostFrame>& F) {
  const int nf = (int)F.size(), dim = 4 + 8 * nf, m = 7;
  std::vector<double> N((size_t)dim * m, 0.0);
  for (int c = 0; c < 7; c++) {
    for (int f = 0; f < nf; f++)
      for (int r = 0; r < 6; r++) {
        double v = c < 6 ? F[f].ns_pose[r][c] : F[f].ns_scale[r];
        v *= r < 3 ? (1.0f / SCALE_XI_TRANS) : (1.0f / SCALE_XI_ROT);
        N[(size_t)(4 + f * 8 + r) * m + c] = v;
      }
    double nrm = 0;
    for (int k = 0; k < dim; k++) nrm += N[(size_t)k * m + c] * N[(size_t)k * m + c];
    nrm = std::sqrt(nrm);
    for (int k = 0; k < dim; k++) N[(size_t)k * m + c] /= nrm;
  }
  return spanProjector(N, dim, m, kSolverModeDelta);
}

}  // namespace sdso
