// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_api.h).
// Constants and samplers shared by the oracle translation units.
//   src/FullSystem/HessianBlocks.h:54-61   SCALE_*
//   src/util/settings.cpp:29-251           setting_* values, patternP = staticPattern[8] (:216)
//   src/util/globalFuncs.h:73-86, :122-135, :160-184   bilinear samplers
//   src/util/NumType.h:159-170             AffLight::fromToVecExposure
#pragma once
#include <cmath>

namespace orc {

#define SCALE_IDEPTH 1.0f
#define SCALE_XI_ROT 1.0f
#define SCALE_XI_TRANS 0.5f
#define SCALE_F 50.0f
#define SCALE_C 50.0f
#define SCALE_W 1.0f
#define SCALE_A 10.0f
#define SCALE_B 1000.0f
#define SCALE_XI_ROT_INVERSE (1.0f / SCALE_XI_ROT)
#define SCALE_XI_TRANS_INVERSE (1.0f / SCALE_XI_TRANS)
#define SCALE_F_INVERSE (1.0f / SCALE_F)
#define SCALE_C_INVERSE (1.0f / SCALE_C)
#define SCALE_A_INVERSE (1.0f / SCALE_A)
#define SCALE_B_INVERSE (1.0f / SCALE_B)

static const int patternNum = 8;
static const int patternP[8][2] = {{0, -2}, {-1, -1}, {1, -1}, {-2, 0}, {0, 0}, {2, 0}, {-1, 1}, {0, 2}};

static const float setting_huberTH = 9;
static const float setting_outlierTH = 12 * 12;
static const float setting_outlierTHSumComponent = 50 * 50;
static const float setting_overallEnergyTHWeight = 1;
static const float setting_idepthFixPrior = 50 * 50;
static const float setting_idepthFixPriorMargFac = 600 * 600;
static const float setting_initialRotPrior = 1e11;
static const float setting_initialTransPrior = 1e10;
static const float setting_initialAffBPrior = 1e14;
static const float setting_initialAffAPrior = 1e14;
static const float setting_initialCalibHessian = 5e9;
static const double setting_solverModeDelta = 0.00001;
static const float setting_margWeightFac = 0.5 * 0.5;
static const float setting_frameEnergyTHConstWeight = 0.5;
static const float setting_frameEnergyTHN = 0.7f;
static const float setting_frameEnergyTHFacMedian = 1.5;
static const float setting_thOptIterations = 1.2;
static const int setting_minOptIterations = 1;
// trace (settings.cpp:111-120)
static const float setting_maxPixSearch = 0.027;
static const float setting_trace_stepsize = 1.0;
static const int setting_trace_GNIterations = 3;
static const float setting_minIdepthH_act = 100;        // settings.cpp:56
static const int setting_GNItsOnPointActivation = 3;    // settings.cpp:114
static const float setting_trace_GNThreshold = 0.1;
static const float setting_trace_extraSlackOnTH = 1.2;
static const float setting_trace_slackInterval = 1.5;
static const float setting_trace_minImprovementFactor = 2;
static const int setting_minTraceTestRadius = 2;

#define SOLVER_SVD 1
#define SOLVER_ORTHOGONALIZE_SYSTEM 2
#define SOLVER_ORTHOGONALIZE_POINTMARG 4
#define SOLVER_ORTHOGONALIZE_FULL 8
#define SOLVER_SVD_CUT7 16
#define SOLVER_REMOVE_POSEPRIOR 32
#define SOLVER_USE_GN 64
#define SOLVER_FIX_LAMBDA 128
#define SOLVER_ORTHOGONALIZE_X 256
#define SOLVER_MOMENTUM 512
#define SOLVER_STEPMOMENTUM 1024
#define SOLVER_ORTHOGONALIZE_X_LATER 2048

// globalFuncs.h:73-86 — AoS {I,dx,dy}
inline void interp33(const float* mat, float x, float y, int width, float* out) {
  int ix = (int)x;
  int iy = (int)y;
  float dx = x - ix;
  float dy = y - iy;
  float dxdy = dx * dy;
  const float* bp = mat + 3 * (ix + iy * width);
  float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
  for (int c = 0; c < 3; c++)
    out[c] = w11 * bp[3 * (1 + width) + c] + w01 * bp[3 * width + c] + w10 * bp[3 + c] + w00 * bp[c];
}
// globalFuncs.h:122-135
inline float interp31(const float* mat, float x, float y, int width) {
  int ix = (int)x;
  int iy = (int)y;
  float dx = x - ix;
  float dy = y - iy;
  float dxdy = dx * dy;
  const float* bp = mat + 3 * (ix + iy * width);
  return dxdy * bp[3 * (1 + width)] + (dy - dxdy) * bp[3 * width] + (dx - dxdy) * bp[3] +
         (1 - dx - dy + dxdy) * bp[0];
}
// globalFuncs.h:160-184 (the x==-1||y==-1 early-out returns an uninitialised vector in the
// reference; callers never pass -1 on this path)
inline void interp33BiLin(const float* mat, float x, float y, int width, float* out) {
  int ix = (int)x;
  int iy = (int)y;
  const float* bp = mat + 3 * (ix + iy * width);
  float tl = bp[0];
  float tr = bp[3];
  float bl = bp[3 * width];
  float br = bp[3 * (width + 1)];
  float dx = x - ix;
  float dy = y - iy;
  float topInt = dx * tr + (1 - dx) * tl;
  float botInt = dx * br + (1 - dx) * bl;
  float leftInt = dy * bl + (1 - dy) * tl;
  float rightInt = dy * br + (1 - dy) * tr;
  out[0] = dx * rightInt + (1 - dx) * leftInt;
  out[1] = rightInt - leftInt;
  out[2] = botInt - topInt;
}

// NumType.h:159-170
inline void fromToVecExposure(float exposureF, float exposureT, double g2F_a, double g2F_b,
                              double g2T_a, double g2T_b, double* out) {
  if (exposureF == 0 || exposureT == 0) exposureT = exposureF = 1;
  double a = std::exp(g2T_a - g2F_a) * exposureT / exposureF;
  double b = g2T_b - a * g2F_b;
  out[0] = a;
  out[1] = b;
}

}  // namespace orc
