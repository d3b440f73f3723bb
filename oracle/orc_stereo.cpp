// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_api.h).
// Static stereo + pyramid construction + math wrappers.  Follows (paths under /root/reference):
//   src/FullSystem/ImmaturePoint.cpp:33-62    ImmaturePoint ctor (patch colours, weights, gradH)
//   src/FullSystem/ImmaturePoint.cpp:94-451   traceStereo; sub-pixel GN = DSO-native twin at :707-769
//   src/FullSystem/HessianBlocks.cpp:141-203  FrameHessian::makeImages
//   src/util/globalCalib.cpp:52-58            pyramid level rule
#include "orc_api.h"
#include "orc_math.h"
#include "orc_common.h"
#include <cmath>
#include <utility>

using namespace orc;

enum { IPS_GOOD = 0, IPS_OOB, IPS_OUTLIER, IPS_SKIPPED, IPS_BADCONDITION, IPS_UNINITIALIZED };

extern "C" int orc_pyramid_levels(int w, int h) {
  int wlvl = w, hlvl = h, levels = 1;
  while (wlvl % 2 == 0 && hlvl % 2 == 0 && wlvl * hlvl > 5000 && levels < 6) { wlvl /= 2; hlvl /= 2; levels++; }
  return levels;
}

extern "C" void orc_make_images(const float* color, int w, int h, int levels, float* const* dIp) {
  float* dI = dIp[0];
  for (int i = 0; i < w * h; i++) { dI[3 * i] = color[i]; dI[3 * i + 1] = 0; dI[3 * i + 2] = 0; }
  for (int lvl = 0; lvl < levels; lvl++) {
    int wl = w >> lvl, hl = h >> lvl;
    float* dI_l = dIp[lvl];
    if (lvl > 0) {
      int wlm1 = w >> (lvl - 1);
      const float* dI_lm = dIp[lvl - 1];
      for (int y = 0; y < hl; y++)
        for (int x = 0; x < wl; x++) {
          dI_l[3 * (x + y * wl)] = 0.25f * (dI_lm[3 * (2 * x + 2 * y * wlm1)] + dI_lm[3 * (2 * x + 1 + 2 * y * wlm1)] +
                                            dI_lm[3 * (2 * x + 2 * y * wlm1 + wlm1)] + dI_lm[3 * (2 * x + 1 + 2 * y * wlm1 + wlm1)]);
          dI_l[3 * (x + y * wl) + 1] = 0;
          dI_l[3 * (x + y * wl) + 2] = 0;
        }
    }
    for (int idx = wl; idx < wl * (hl - 1); idx++) {
      float dx = 0.5f * (dI_l[3 * (idx + 1)] - dI_l[3 * (idx - 1)]);
      float dy = 0.5f * (dI_l[3 * (idx + wl)] - dI_l[3 * (idx - wl)]);
      if (!std::isfinite(dx)) dx = 0;
      if (!std::isfinite(dy)) dy = 0;
      dI_l[3 * idx + 1] = dx;
      dI_l[3 * idx + 2] = dy;
    }
  }
}

// ImmaturePoint.cpp:33-62
extern "C" int orc_immature_init_batch(const float* dI, int w, int h, int n, const float* u, const float* v, float* color,
                                       float* weights, float* gradH, float* energyTH) {
  (void)h;
  for (int p = 0; p < n; p++) {
    float gH[4] = {0, 0, 0, 0};
    bool bad = false;
    for (int idx = 0; idx < patternNum; idx++) {
      int dx = patternP[idx][0], dy = patternP[idx][1];
      float ptc[3];
      interp33BiLin(dI, u[p] + dx, v[p] + dy, w, ptc);
      color[p * 8 + idx] = ptc[0];
      if (!std::isfinite(ptc[0])) { bad = true; break; }
      gH[0] += ptc[1] * ptc[1]; gH[1] += ptc[1] * ptc[2]; gH[2] += ptc[2] * ptc[1]; gH[3] += ptc[2] * ptc[2];
      weights[p * 8 + idx] = sqrtf(setting_outlierTHSumComponent / (setting_outlierTHSumComponent + (ptc[1] * ptc[1] + ptc[2] * ptc[2])));
    }
    for (int k = 0; k < 4; k++) gradH[p * 4 + k] = gH[k];
    if (bad) { energyTH[p] = NAN; continue; }
    float e = patternNum * setting_outlierTH;
    e *= setting_overallEnergyTHWeight * setting_overallEnergyTHWeight;
    energyTH[p] = e;
  }
  return 0;
}

namespace {
// One point of ImmaturePoint::traceStereo.  Returns the new lastTraceStatus.
int traceStereoOne(const float* dI, int wG0, int hG0, const float* K4, float baseline, bool mode_right,
                   orc_trace_points_t* P, int i, int gn_mode = 0) {
  float& u_stereo = P->u_stereo[i]; float& v_stereo = P->v_stereo[i];
  float& idepth_min_stereo = P->idepth_min_stereo[i]; float& idepth_max_stereo = P->idepth_max_stereo[i];
  const float* color = P->color + i * 8; const float* weights = P->weights + i * 8; const float* gradH = P->gradH + i * 4;
  float* lastTraceUV = P->lastTraceUV + i * 2;
  float& lastTracePixelInterval = P->lastTracePixelInterval[i];
  uint8_t& lastTraceStatus = P->lastTraceStatus[i];
  const float idepth_min = P->idepth_min[i];
  const float energyTH = P->energyTH[i];
  float& quality = P->quality[i];

  // KRKi = I ; K = [fx 0 cx; 0 fy cy; 0 0 1]
  float bl[3] = {mode_right ? -baseline : baseline, 0, 0};
  float Kt[3];
  Kt[0] = (K4[0] * bl[0] + 0.0f * bl[1]) + K4[2] * bl[2];
  Kt[1] = (0.0f * bl[0] + K4[1] * bl[1]) + K4[3] * bl[2];
  Kt[2] = (0.0f * bl[0] + 0.0f * bl[1]) + 1.0f * bl[2];
  const float aff0 = 1, aff1 = 0;
  float bf = -K4[0] * bl[0];
  float pr[3];
  pr[0] = (1.0f * u_stereo + 0.0f * v_stereo) + 0.0f * 1.0f;
  pr[1] = (0.0f * u_stereo + 1.0f * v_stereo) + 0.0f * 1.0f;
  pr[2] = (0.0f * u_stereo + 0.0f * v_stereo) + 1.0f * 1.0f;
  float ptpMin[3];
  for (int k = 0; k < 3; k++) ptpMin[k] = pr[k] + Kt[k] * idepth_min_stereo;
  float uMin = ptpMin[0] / ptpMin[2];
  float vMin = ptpMin[1] / ptpMin[2];
  auto oob = [&]() { lastTraceUV[0] = -1; lastTraceUV[1] = -1; lastTracePixelInterval = 0; return (int)(lastTraceStatus = IPS_OOB); };
  if (!(uMin > 4 && vMin > 4 && uMin < wG0 - 5 && vMin < hG0 - 5)) return oob();

  float dist, uMax, vMax, ptpMax[3];
  float maxPixSearch = (wG0 + hG0) * setting_maxPixSearch;
  if (std::isfinite(idepth_max_stereo)) {
    for (int k = 0; k < 3; k++) ptpMax[k] = pr[k] + Kt[k] * idepth_max_stereo;
    uMax = ptpMax[0] / ptpMax[2];
    vMax = ptpMax[1] / ptpMax[2];
    if (!(uMax > 4 && vMax > 4 && uMax < wG0 - 5 && vMax < hG0 - 5)) return oob();
    dist = (uMin - uMax) * (uMin - uMax) + (vMin - vMax) * (vMin - vMax);
    dist = sqrtf(dist);
    if (dist < setting_trace_slackInterval) return lastTraceStatus = IPS_SKIPPED;
  } else {
    dist = maxPixSearch;
    for (int k = 0; k < 3; k++) ptpMax[k] = pr[k] + Kt[k] * 0.01f;
    uMax = ptpMax[0] / ptpMax[2];
    vMax = ptpMax[1] / ptpMax[2];
    float dx = uMax - uMin;
    float dy = vMax - vMin;
    float d = 1.0f / sqrtf(dx * dx + dy * dy);
    uMax = uMin + dist * dx * d;
    vMax = vMin + dist * dy * d;
    if (!(uMax > 4 && vMax > 4 && uMax < wG0 - 5 && vMax < hG0 - 5)) return oob();
  }
  if (!(idepth_min < 0 || (ptpMin[2] > 0.75 && ptpMin[2] < 1.5))) return oob();

  float dx = setting_trace_stepsize * (uMax - uMin);
  float dy = setting_trace_stepsize * (vMax - vMin);
  // Vec2f(dx,dy)^T * gradH * Vec2f(dx,dy): (row * matrix) then * vector
  float a = (dx * gradH[0] + dy * gradH[2]) * dx + (dx * gradH[1] + dy * gradH[3]) * dy;
  float b = (dy * gradH[0] + (-dx) * gradH[2]) * dy + (dy * gradH[1] + (-dx) * gradH[3]) * (-dx);
  float errorInPixel = 0.2f + 0.2f * (a + b) / a;
  if (errorInPixel * setting_trace_minImprovementFactor > dist && std::isfinite(idepth_max_stereo))
    return lastTraceStatus = IPS_BADCONDITION;
  if (errorInPixel > 10) errorInPixel = 10;

  dx /= dist;
  dy /= dist;
  if (dist > maxPixSearch) {
    uMax = uMin + maxPixSearch * dx;
    vMax = vMin + maxPixSearch * dy;
    dist = maxPixSearch;
  }
  int numSteps = 1.9999f + dist / setting_trace_stepsize;
  float randShift = uMin * 1000 - floorf(uMin * 1000);
  float ptx = uMin - randShift * dx;
  float pty = vMin - randShift * dy;
  float rotatetPattern[8][2];
  for (int idx = 0; idx < patternNum; idx++) {  // Rplane = identity 2x2
    rotatetPattern[idx][0] = 1.0f * patternP[idx][0] + 0.0f * patternP[idx][1];
    rotatetPattern[idx][1] = 0.0f * patternP[idx][0] + 1.0f * patternP[idx][1];
  }
  if (!std::isfinite(dx) || !std::isfinite(dy)) return oob();

  float errors[100];
  float bestU = 0, bestV = 0, bestEnergy = 1e10;
  int bestIdx = -1;
  if (numSteps >= 100) numSteps = 99;
  for (int s = 0; s < numSteps; s++) {
    float energy = 0;
    for (int idx = 0; idx < patternNum; idx++) {
      float hitColor = interp31(dI, (float)(ptx + rotatetPattern[idx][0]), (float)(pty + rotatetPattern[idx][1]), wG0);
      if (!std::isfinite(hitColor)) { energy += 1e5; continue; }
      float residual = hitColor - (float)(aff0 * color[idx] + aff1);
      float hw = fabs(residual) < setting_huberTH ? 1 : setting_huberTH / fabs(residual);
      energy += hw * residual * residual * (2 - hw);
    }
    errors[s] = energy;
    if (energy < bestEnergy) { bestU = ptx; bestV = pty; bestEnergy = energy; bestIdx = s; }
    ptx += dx;
    pty += dy;
  }
  float secondBest = 1e10;
  for (int s = 0; s < numSteps; s++)
    if ((s < bestIdx - setting_minTraceTestRadius || s > bestIdx + setting_minTraceTestRadius) && errors[s] < secondBest) secondBest = errors[s];
  float newQuality = secondBest / bestEnergy;
  if (newQuality < quality || numSteps > 10) quality = newQuality;

  if (gn_mode == 1) {
    // fork-live GN (ImmaturePoint.cpp:309-412): a VertexUVDSO at (bestU, bestV), per pass 8 NEW EdgeTracePointUVDSO
    // (dso_g2o_edge.cpp:571-619) with Huber(9) and one g2o Gauss-Newton iteration over all edges added so far.  The
    // duplicates of earlier passes scale H and b alike, so the step is b/H of the 8 current edges (restated g2o: unpinned).
    double U = bestU, V = bestV;
    const double ddx = dx, ddy = dy;
    if (setting_trace_GNIterations > 0) bestEnergy = 1e5;
    for (int it = 0; it < setting_trace_GNIterations; it++) {
      float energy = 0;
      double Hs = 0, bs = 0;
      for (int idx = 0; idx < patternNum; idx++) {
        double e = 0, J = 0;
        if (!((U - 2) < 0 || (U + 3) > (wG0 - 3) || (V - 2) < 0 || (V + 3) > (hG0 - 3))) {   // util::CheckBoundary(bestU, bestV, wG[0]-3, hG[0]-3)
          float hit[3];
          interp33(dI, (float)(U + rotatetPattern[idx][0]), (float)(V + rotatetPattern[idx][1]), wG0, hit);
          if (std::isfinite(hit[0])) {
            e = hit[0] - (aff0 * (double)color[idx] + aff1);
            J = ddx * hit[1] + ddy * hit[2];
          }
        }
        const float residual = e;
        const float hw = fabs(residual) < setting_huberTH ? 1 : setting_huberTH / fabs(residual);
        energy += weights[idx] * weights[idx] * hw * residual * residual * (2 - hw);
        const double e2 = e * e, dsqr = (double)setting_huberTH * setting_huberTH;
        const double rho1 = e2 <= dsqr ? 1. : setting_huberTH / std::sqrt(e2);
        bs -= rho1 * J * e;
        Hs += J * rho1 * J;
      }
      if (Hs != 0) {                                       // LDLT of a zero 1x1 system fails -> no update
        double update = bs / Hs;                           // VertexUVDSO::oplusImpl (dso_g2o_vertex.cpp:73-88)
        if (update < -0.5) update = -0.5;
        else if (update > 0.5) update = 0.5;
        else if (!std::isfinite(update)) update = 0;
        U += update * ddx;
        V += update * ddy;
      }
      if (!(energy > bestEnergy)) bestEnergy = energy;     // :381-407 (no step back in the live code)
    }
    bestU = U;
    bestV = V;
  } else {
  // DSO-native GN (ImmaturePoint.cpp:707-769)
  float uBak = bestU, vBak = bestV, gnstepsize = 1, stepBack = 0;
  if (setting_trace_GNIterations > 0) bestEnergy = 1e5;
  for (int it = 0; it < setting_trace_GNIterations; it++) {
    float H = 1, bb = 0, energy = 0;
    for (int idx = 0; idx < patternNum; idx++) {
      float hitColor[3];
      interp33(dI, (float)(bestU + rotatetPattern[idx][0]), (float)(bestV + rotatetPattern[idx][1]), wG0, hitColor);
      if (!std::isfinite((float)hitColor[0])) { energy += 1e5; continue; }
      float residual = hitColor[0] - (aff0 * color[idx] + aff1);
      float dResdDist = dx * hitColor[1] + dy * hitColor[2];
      float hw = fabs(residual) < setting_huberTH ? 1 : setting_huberTH / fabs(residual);
      H += hw * dResdDist * dResdDist;
      bb += hw * residual * dResdDist;
      energy += weights[idx] * weights[idx] * hw * residual * residual * (2 - hw);
    }
    if (energy > bestEnergy) {
      stepBack *= 0.5;
      bestU = uBak + stepBack * dx;
      bestV = vBak + stepBack * dy;
    } else {
      float step = -gnstepsize * bb / H;
      if (step < -0.5) step = -0.5;
      else if (step > 0.5) step = 0.5;
      if (!std::isfinite(step)) step = 0;
      uBak = bestU;
      vBak = bestV;
      stepBack = step;
      bestU += step * dx;
      bestV += step * dy;
      bestEnergy = energy;
    }
    if (fabsf(stepBack) < setting_trace_GNThreshold) break;
  }
  }

  if (!(bestEnergy < energyTH * setting_trace_extraSlackOnTH)) {
    lastTracePixelInterval = 0;
    lastTraceUV[0] = -1; lastTraceUV[1] = -1;
    if (lastTraceStatus == IPS_OUTLIER) return lastTraceStatus = IPS_OOB;
    else return lastTraceStatus = IPS_OUTLIER;
  }
  if (dx * dx > dy * dy) {
    idepth_min_stereo = (pr[2] * (bestU - errorInPixel * dx) - pr[0]) / (Kt[0] - Kt[2] * (bestU - errorInPixel * dx));
    idepth_max_stereo = (pr[2] * (bestU + errorInPixel * dx) - pr[0]) / (Kt[0] - Kt[2] * (bestU + errorInPixel * dx));
  } else {
    idepth_min_stereo = (pr[2] * (bestV - errorInPixel * dy) - pr[1]) / (Kt[1] - Kt[2] * (bestV - errorInPixel * dy));
    idepth_max_stereo = (pr[2] * (bestV + errorInPixel * dy) - pr[1]) / (Kt[1] - Kt[2] * (bestV + errorInPixel * dy));
  }
  if (idepth_min_stereo > idepth_max_stereo) std::swap(idepth_min_stereo, idepth_max_stereo);
  if (!std::isfinite(idepth_min_stereo) || !std::isfinite(idepth_max_stereo) || (idepth_max_stereo < 0)) {
    lastTracePixelInterval = 0;
    lastTraceUV[0] = -1; lastTraceUV[1] = -1;
    return lastTraceStatus = IPS_OUTLIER;
  }
  lastTracePixelInterval = 2 * errorInPixel;
  lastTraceUV[0] = bestU; lastTraceUV[1] = bestV;
  P->idepth_stereo[i] = (u_stereo - bestU) / bf;
  return lastTraceStatus = IPS_GOOD;
}
// ImmaturePoint::traceOn (src/FullSystem/ImmaturePoint.cpp:459-828), DSO-native sub-pixel GN (:707-769).
// P->u_stereo / v_stereo carry u / v, P->idepth_min_stereo / idepth_max_stereo carry idepth_min / idepth_max (in/out).
int traceOnOne(const float* dI, int wG0, int hG0, const orc_trace_geom_t* G, orc_trace_points_t* P, int i) {
  float& u_stereo = P->u_stereo[i]; float& v_stereo = P->v_stereo[i];
  float& idepth_min_stereo = P->idepth_min_stereo[i]; float& idepth_max_stereo = P->idepth_max_stereo[i];
  const float* color = P->color + i * 8; const float* weights = P->weights + i * 8; const float* gradH = P->gradH + i * 4;
  float* lastTraceUV = P->lastTraceUV + i * 2;
  float& lastTracePixelInterval = P->lastTracePixelInterval[i];
  uint8_t& lastTraceStatus = P->lastTraceStatus[i];
  const float energyTH = P->energyTH[i];
  float& quality = P->quality[i];
  if (lastTraceStatus == IPS_OOB) return lastTraceStatus;                               // :466-468
  const float idepth_min = idepth_min_stereo;
  const float* KRKi = G->KRKi; const float* Kt = G->Kt;
  const float aff0 = G->aff[0], aff1 = G->aff[1];
  float pr[3];
  for (int r = 0; r < 3; r++) pr[r] = (KRKi[r * 3 + 0] * u_stereo + KRKi[r * 3 + 1] * v_stereo) + KRKi[r * 3 + 2] * 1.0f;
  float ptpMin[3];
  for (int k = 0; k < 3; k++) ptpMin[k] = pr[k] + Kt[k] * idepth_min_stereo;
  float uMin = ptpMin[0] / ptpMin[2];
  float vMin = ptpMin[1] / ptpMin[2];
  auto oob = [&]() { lastTraceUV[0] = -1; lastTraceUV[1] = -1; lastTracePixelInterval = 0; return (int)(lastTraceStatus = IPS_OOB); };
  if (!(uMin > 4 && vMin > 4 && uMin < wG0 - 5 && vMin < hG0 - 5)) return oob();

  float dist, uMax, vMax, ptpMax[3];
  float maxPixSearch = (wG0 + hG0) * setting_maxPixSearch;
  if (std::isfinite(idepth_max_stereo)) {
    for (int k = 0; k < 3; k++) ptpMax[k] = pr[k] + Kt[k] * idepth_max_stereo;
    uMax = ptpMax[0] / ptpMax[2];
    vMax = ptpMax[1] / ptpMax[2];
    if (!(uMax > 4 && vMax > 4 && uMax < wG0 - 5 && vMax < hG0 - 5)) return oob();
    dist = (uMin - uMax) * (uMin - uMax) + (vMin - vMax) * (vMin - vMax);
    dist = sqrtf(dist);
    if (dist < setting_trace_slackInterval) {                                            // :525-531
      lastTraceUV[0] = (uMax + uMin) * 0.5f; lastTraceUV[1] = (vMax + vMin) * 0.5f;
      lastTracePixelInterval = dist;
      return lastTraceStatus = IPS_SKIPPED;
    }
  } else {
    dist = maxPixSearch;
    for (int k = 0; k < 3; k++) ptpMax[k] = pr[k] + Kt[k] * 0.01f;
    uMax = ptpMax[0] / ptpMax[2];
    vMax = ptpMax[1] / ptpMax[2];
    float dx = uMax - uMin;
    float dy = vMax - vMin;
    float d = 1.0f / sqrtf(dx * dx + dy * dy);
    uMax = uMin + dist * dx * d;
    vMax = vMin + dist * dy * d;
    if (!(uMax > 4 && vMax > 4 && uMax < wG0 - 5 && vMax < hG0 - 5)) return oob();
  }
  if (!(idepth_min < 0 || (ptpMin[2] > 0.75 && ptpMin[2] < 1.5))) return oob();

  float dx = setting_trace_stepsize * (uMax - uMin);
  float dy = setting_trace_stepsize * (vMax - vMin);
  // Vec2f(dx,dy)^T * gradH * Vec2f(dx,dy): (row * matrix) then * vector
  float a = (dx * gradH[0] + dy * gradH[2]) * dx + (dx * gradH[1] + dy * gradH[3]) * dy;
  float b = (dy * gradH[0] + (-dx) * gradH[2]) * dy + (dy * gradH[1] + (-dx) * gradH[3]) * (-dx);
  float errorInPixel = 0.2f + 0.2f * (a + b) / a;
  if (errorInPixel * setting_trace_minImprovementFactor > dist && std::isfinite(idepth_max_stereo)) {   // :596-603
    lastTraceUV[0] = (uMax + uMin) * 0.5f; lastTraceUV[1] = (vMax + vMin) * 0.5f;
    lastTracePixelInterval = dist;
    return lastTraceStatus = IPS_BADCONDITION;
  }
  if (errorInPixel > 10) errorInPixel = 10;

  dx /= dist;
  dy /= dist;
  if (dist > maxPixSearch) {
    uMax = uMin + maxPixSearch * dx;
    vMax = vMin + maxPixSearch * dy;
    dist = maxPixSearch;
  }
  int numSteps = 1.9999f + dist / setting_trace_stepsize;
  float randShift = uMin * 1000 - floorf(uMin * 1000);
  float ptx = uMin - randShift * dx;
  float pty = vMin - randShift * dy;
  float rotatetPattern[8][2];
  for (int idx = 0; idx < patternNum; idx++) {  // Rplane = KRKi.topLeftCorner<2,2>()  (:628, :636-637)
    rotatetPattern[idx][0] = KRKi[0] * patternP[idx][0] + KRKi[1] * patternP[idx][1];
    rotatetPattern[idx][1] = KRKi[3] * patternP[idx][0] + KRKi[4] * patternP[idx][1];
  }
  if (!std::isfinite(dx) || !std::isfinite(dy)) return oob();

  float errors[100];
  float bestU = 0, bestV = 0, bestEnergy = 1e10;
  int bestIdx = -1;
  if (numSteps >= 100) numSteps = 99;
  for (int s = 0; s < numSteps; s++) {
    float energy = 0;
    for (int idx = 0; idx < patternNum; idx++) {
      float hitColor = interp31(dI, (float)(ptx + rotatetPattern[idx][0]), (float)(pty + rotatetPattern[idx][1]), wG0);
      if (!std::isfinite(hitColor)) { energy += 1e5; continue; }
      float residual = hitColor - (float)(aff0 * color[idx] + aff1);
      float hw = fabs(residual) < setting_huberTH ? 1 : setting_huberTH / fabs(residual);
      energy += hw * residual * residual * (2 - hw);
    }
    errors[s] = energy;
    if (energy < bestEnergy) { bestU = ptx; bestV = pty; bestEnergy = energy; bestIdx = s; }
    ptx += dx;
    pty += dy;
  }
  float secondBest = 1e10;
  for (int s = 0; s < numSteps; s++)
    if ((s < bestIdx - setting_minTraceTestRadius || s > bestIdx + setting_minTraceTestRadius) && errors[s] < secondBest) secondBest = errors[s];
  float newQuality = secondBest / bestEnergy;
  if (newQuality < quality || numSteps > 10) quality = newQuality;

  // DSO-native GN (ImmaturePoint.cpp:707-769)
  float uBak = bestU, vBak = bestV, gnstepsize = 1, stepBack = 0;
  if (setting_trace_GNIterations > 0) bestEnergy = 1e5;
  for (int it = 0; it < setting_trace_GNIterations; it++) {
    float H = 1, bb = 0, energy = 0;
    for (int idx = 0; idx < patternNum; idx++) {
      float hitColor[3];
      interp33(dI, (float)(bestU + rotatetPattern[idx][0]), (float)(bestV + rotatetPattern[idx][1]), wG0, hitColor);
      if (!std::isfinite((float)hitColor[0])) { energy += 1e5; continue; }
      float residual = hitColor[0] - (aff0 * color[idx] + aff1);
      float dResdDist = dx * hitColor[1] + dy * hitColor[2];
      float hw = fabs(residual) < setting_huberTH ? 1 : setting_huberTH / fabs(residual);
      H += hw * dResdDist * dResdDist;
      bb += hw * residual * dResdDist;
      energy += weights[idx] * weights[idx] * hw * residual * residual * (2 - hw);
    }
    if (energy > bestEnergy) {
      stepBack *= 0.5;
      bestU = uBak + stepBack * dx;
      bestV = vBak + stepBack * dy;
    } else {
      float step = -gnstepsize * bb / H;
      if (step < -0.5) step = -0.5;
      else if (step > 0.5) step = 0.5;
      if (!std::isfinite(step)) step = 0;
      uBak = bestU;
      vBak = bestV;
      stepBack = step;
      bestU += step * dx;
      bestV += step * dy;
      bestEnergy = energy;
    }
    if (fabsf(stepBack) < setting_trace_GNThreshold) break;
  }

  if (!(bestEnergy < energyTH * setting_trace_extraSlackOnTH)) {
    lastTracePixelInterval = 0;
    lastTraceUV[0] = -1; lastTraceUV[1] = -1;
    if (lastTraceStatus == IPS_OUTLIER) return lastTraceStatus = IPS_OOB;
    else return lastTraceStatus = IPS_OUTLIER;
  }
  if (dx * dx > dy * dy) {
    idepth_min_stereo = (pr[2] * (bestU - errorInPixel * dx) - pr[0]) / (Kt[0] - Kt[2] * (bestU - errorInPixel * dx));
    idepth_max_stereo = (pr[2] * (bestU + errorInPixel * dx) - pr[0]) / (Kt[0] - Kt[2] * (bestU + errorInPixel * dx));
  } else {
    idepth_min_stereo = (pr[2] * (bestV - errorInPixel * dy) - pr[1]) / (Kt[1] - Kt[2] * (bestV - errorInPixel * dy));
    idepth_max_stereo = (pr[2] * (bestV + errorInPixel * dy) - pr[1]) / (Kt[1] - Kt[2] * (bestV + errorInPixel * dy));
  }
  if (idepth_min_stereo > idepth_max_stereo) std::swap(idepth_min_stereo, idepth_max_stereo);
  if (!std::isfinite(idepth_min_stereo) || !std::isfinite(idepth_max_stereo) || (idepth_max_stereo < 0)) {
    lastTracePixelInterval = 0;
    lastTraceUV[0] = -1; lastTraceUV[1] = -1;
    return lastTraceStatus = IPS_OUTLIER;
  }
  lastTracePixelInterval = 2 * errorInPixel;
  lastTraceUV[0] = bestU; lastTraceUV[1] = bestV;
  return lastTraceStatus = IPS_GOOD;
}
}  // namespace

extern "C" int orc_trace_on_batch(const float* dI, int w, int h, int ngeom, const orc_trace_geom_t* geom, const int* point_geom,
                                  orc_trace_points_t* pts, uint8_t* status) {
  for (int i = 0; i < pts->n; i++) {
    if (point_geom[i] < 0 || point_geom[i] >= ngeom) return -1;
    int s = traceOnOne(dI, w, h, geom + point_geom[i], pts, i);
    if (status) status[i] = (uint8_t)s;
  }
  return 0;
}

extern "C" int orc_trace_stereo_batch(const float* dI, int w, int h, const float K[4], float baseline, int mode_right,
                                      orc_trace_points_t* pts, uint8_t* status) {
  for (int i = 0; i < pts->n; i++) {
    int s = traceStereoOne(dI, w, h, K, baseline, mode_right != 0, pts, i, 0);
    if (status) status[i] = (uint8_t)s;
  }
  return 0;
}


// the same with the fork-live sub-pixel refinement (gn_mode 1: g2o Gauss-Newton on EdgeTracePointUVDSO, ImmaturePoint.cpp:309-412)
extern "C" int orc_trace_stereo_batch_gn(const float* dI, int w, int h, const float K[4], float baseline, int mode_right,
                                         orc_trace_points_t* pts, uint8_t* status, int gn_mode) {
  for (int i = 0; i < pts->n; i++) {
    int s = traceStereoOne(dI, w, h, K, baseline, mode_right != 0, pts, i, gn_mode);
    if (status) status[i] = (uint8_t)s;
  }
  return 0;
}

// ------------------------------------------------------------------ point activation
// FullSystem::optimizeImmaturePoint (src/FullSystem/FullSystemOptPoint.cpp:52-238) with ImmaturePoint::linearizeResidual
// (src/FullSystem/ImmaturePoint.cpp:886-985): the DSO-native idepth-only Gauss-Newton that the fork keeps as comments
// (FullSystemOptPoint.cpp:104-110, :124-169; ImmaturePoint.cpp:940-968), including the two "not well-constrained" exits.
namespace {
struct ActFrameGeom { const float* R; const float* t; const float* aff; };
// returns energy; accumulates Hdd / bd exactly like the reference (contributions of the pixels before an OOB pixel stay in)
static double linearizeResidualOne(const orc_activate_t* A, int p, int tgt, float outlierTHSlack, uint8_t state_state, float state_energy,
                                   uint8_t* newState, float* newEnergy, float& Hdd, float& bd, float idepth) {
  if (state_state == 1) { *newState = 1; return state_energy; }                       // :893-895
  const int host = A->host[p], nf = A->nf;
  const float* R = A->pair_R + (size_t)(host * nf + tgt) * 9;
  const float* t = A->pair_t + (size_t)(host * nf + tgt) * 3;
  const float* affLL = A->pair_aff + (size_t)(host * nf + tgt) * 2;
  const float* dIl = A->dI[tgt];
  const float fxl = A->K[0], fyl = A->K[1], cxl = A->K[2], cyl = A->K[3];
  const float fxli = 1.0f / fxl, fyli = 1.0f / fyl;
  const float wM3G = A->w - 3, hM3G = A->h - 3;
  const float* color = A->color + (size_t)p * 8; const float* weights = A->weights + (size_t)p * 8;
  const float energyTH = A->energyTH[p];
  float energyLeft = 0;
  for (int idx = 0; idx < patternNum; idx++) {
    int dx = patternP[idx][0], dy = patternP[idx][1];
    // projectPoint (ResidualProjections.h:64-96)
    float KliP[3] = {(A->u[p] + dx - cxl) * fxli, (A->v[p] + dy - cyl) * fyli, 1};
    float ptp[3];
    for (int r = 0; r < 3; r++) ptp[r] = ((R[r * 3 + 0] * KliP[0] + R[r * 3 + 1] * KliP[1]) + R[r * 3 + 2] * KliP[2]) + t[r] * idepth;
    float drescale = 1.0f / ptp[2];
    if (!(drescale > 0)) { *newState = 1; return state_energy; }
    float u = ptp[0] * drescale, v = ptp[1] * drescale;
    float Ku = u * fxl + cxl, Kv = v * fyl + cyl;
    if (!(Ku > 1.1f && Kv > 1.1f && Ku < wM3G && Kv < hM3G)) { *newState = 1; return state_energy; }
    float hitColor[3];
    interp33(dIl, Ku, Kv, A->w, hitColor);
    if (!std::isfinite((float)hitColor[0])) { *newState = 1; return state_energy; }
    float residual = hitColor[0] - (affLL[0] * color[idx] + affLL[1]);
    float hw = fabsf(residual) < setting_huberTH ? 1 : setting_huberTH / fabsf(residual);
    energyLeft += weights[idx] * weights[idx] * hw * residual * residual * (2 - hw);
    float dxInterp = hitColor[1] * fxl, dyInterp = hitColor[2] * fyl;
    float d_idepth = (dxInterp * drescale * (t[0] - t[2] * u) + dyInterp * drescale * (t[1] - t[2] * v)) * SCALE_IDEPTH;   // derive_idepth
    hw *= weights[idx] * weights[idx];
    Hdd += (hw * d_idepth) * d_idepth;
    bd += (hw * residual) * d_idepth;
  }
  if (energyLeft > energyTH * outlierTHSlack) { energyLeft = energyTH * outlierTHSlack; *newState = 2; }
  else *newState = 0;
  *newEnergy = energyLeft;
  return energyLeft;
}
}  // namespace

extern "C" int orc_activate_points(const orc_activate_t* A, int8_t* status, float* idepth_out, uint8_t* res_state) {
  const int nf = A->nf;
  for (int p = 0; p < A->n; p++) {
    uint8_t st[8], nst[8]; float en[8], nen[8]; int tg[8];
    int nres = 0;
    for (int f = 0; f < nf; f++) if (f != A->host[p]) { st[nres] = 0; nst[nres] = 2; en[nres] = 0; nen[nres] = 0; tg[nres] = f; nres++; }
    for (int f = 0; f < nf; f++) res_state[(size_t)p * nf + f] = 255;
    float lastEnergy = 0, lastHdd = 0, lastbd = 0;
    float currentIdepth = (A->idepth_max[p] + A->idepth_min[p]) * 0.5f;
    for (int i = 0; i < nres; i++) {
      lastEnergy += linearizeResidualOne(A, p, tg[i], 1000, st[i], en[i], &nst[i], &nen[i], lastHdd, lastbd, currentIdepth);
      st[i] = nst[i]; en[i] = nen[i];
    }
    idepth_out[p] = currentIdepth;
    if (!std::isfinite(lastEnergy) || lastHdd < setting_minIdepthH_act) { status[p] = 0; continue; }       // :104-110
    float lambda = 0.1;
    bool skip = false;
    for (int iteration = 0; iteration < setting_GNItsOnPointActivation; iteration++) {
      float H = lastHdd;
      H *= 1 + lambda;
      float step = (1.0 / H) * lastbd;
      float newIdepth = currentIdepth - step;
      float newHdd = 0, newbd = 0, newEnergy = 0;
      for (int i = 0; i < nres; i++) newEnergy += linearizeResidualOne(A, p, tg[i], 1, st[i], en[i], &nst[i], &nen[i], newHdd, newbd, newIdepth);
      if (!std::isfinite(lastEnergy) || newHdd < setting_minIdepthH_act) { skip = true; break; }            // :134-141
      if (newEnergy < lastEnergy) {
        currentIdepth = newIdepth; lastHdd = newHdd; lastbd = newbd; lastEnergy = newEnergy;
        for (int i = 0; i < nres; i++) { st[i] = nst[i]; en[i] = nen[i]; }
        lambda *= 0.5;
      } else lambda *= 5;
      if (fabsf(step) < 0.0001 * currentIdepth) break;
    }
    idepth_out[p] = currentIdepth;
    if (skip) { status[p] = 0; continue; }
    if (!std::isfinite(currentIdepth)) { status[p] = -1; continue; }
    int numGoodRes = 0;
    for (int i = 0; i < nres; i++) if (st[i] == 0) numGoodRes++;
    for (int i = 0; i < nres; i++) res_state[(size_t)p * nf + tg[i]] = st[i];
    if (numGoodRes < A->minObs) { status[p] = -1; continue; }
    if (!std::isfinite(A->energyTH[p])) { status[p] = -1; continue; }                                       // :199-202
    status[p] = 1;
  }
  return 0;
}

// ---- math wrappers
static SE3 toSE3(const orc_se3_t* T) { SE3 S; std::memcpy(S.R, T->R, 72); std::memcpy(S.t, T->t, 24); return S; }
static void fromSE3(const SE3& S, orc_se3_t* T) { std::memcpy(T->R, S.R, 72); std::memcpy(T->t, S.t, 24); }
extern "C" void orc_se3_exp(const double xi[6], orc_se3_t* T) { fromSE3(se3_exp(xi), T); }
extern "C" void orc_se3_log(const orc_se3_t* T, double xi[6]) { se3_log(toSE3(T), xi); }
extern "C" void orc_se3_adj(const orc_se3_t* T, double A[36]) { se3_adj(toSE3(T), A); }
extern "C" void orc_se3_mul(const orc_se3_t* A, const orc_se3_t* B, orc_se3_t* C) { fromSE3(se3_mul(toSE3(A), toSE3(B)), C); }
extern "C" void orc_se3_inv(const orc_se3_t* A, orc_se3_t* C) { fromSE3(se3_inv(toSE3(A)), C); }
extern "C" void orc_mat3f_inv(const float m[9], float inv[9]) { mat3_inv<float>(m, inv); }
extern "C" int orc_ldlt_solve(int n, const double* A, const double* rhs, double* x) {
  MatX M(n, n);
  for (int i = 0; i < n * n; i++) M.d[i] = A[i];
  VecX r(rhs, rhs + n), xs;
  bool ok = ldlt_solve(M, r, xs);
  for (int i = 0; i < n; i++) x[i] = xs[i];
  return ok ? 0 : 1;
}
