// Static stereo on gfx950: ImmaturePoint construction and ImmaturePoint::traceStereo for a batch
// of points in one launch.
//
// Reference (paths under /root/reference):
//   src/FullSystem/ImmaturePoint.cpp:33-62    ImmaturePoint::ImmaturePoint (colour patch, weights, gradH)
//   src/FullSystem/ImmaturePoint.cpp:94-451   ImmaturePoint::traceStereo; the sub-pixel Gauss-Newton is
//                                             the DSO-native one (twin in traceOn, :707-769)
//
// Kernel design: one 64-lane wave per point.  The search geometry (a few dozen scalar flops) is
// evaluated redundantly by every lane; the discrete epipolar search puts ONE search step on each
// lane (numSteps <= 99 -> at most two passes), so the 8-pixel x 4-tap gathers of neighbouring
// steps hit the same image rows and are served from L1/L2; best / second-best are wave
// reductions with the reference's first-minimum tie rule; the GN refinement evaluates the 8
// pattern pixels on lanes 0..7 and sums them in pattern order.  Per-point arithmetic keeps the
// reference's operation order (no FP contraction) => statuses, bestIdx and all outputs are
// bit-identical to the CPU path.
#include "sdso_internal.h"
#include <cmath>

using namespace sdso;

namespace sdso {

__constant__ int c_pat[8][2] = {{0, -2}, {-1, -1}, {1, -1}, {-2, 0}, {0, 0}, {2, 0}, {-1, 1}, {0, 2}};
constexpr float kMaxPixSearch = 0.027f, kTraceStepsize = 1.0f, kTraceGNThreshold = 0.1f, kTraceExtraSlack = 1.2f,
                kTraceSlackInterval = 1.5f, kTraceMinImprovement = 2.f, kOutlierTH = 144.f;
constexpr int kTraceGNIterations = 3, kMinTraceTestRadius = 2;
enum { IPS_GOOD = 0, IPS_OOB, IPS_OUTLIER, IPS_SKIPPED, IPS_BADCONDITION, IPS_UNINITIALIZED };

struct TraceDev {
  int n, w, h, mode_right;
  float fx, fy, cx, cy, baseline;
  const float4* img;
  const float* plane;   // level-0 intensities only (discrete search)
  float *u_stereo, *v_stereo, *idepth_min, *idepth_min_stereo, *idepth_max_stereo, *idepth_stereo;
  const float *color, *weights, *gradH, *energyTH;
  float* quality; uint8_t* lastTraceStatus; float* lastTraceUV; float* lastTracePixelInterval;
  uint8_t* status;
  const uint8_t* skip;   // optional: 1 = leave the point alone (status 255)
};

// getInterpolatedElement33BiLin (src/util/globalFuncs.h:160-184)
__device__ __forceinline__ float3 interp33BiLin(const float4* __restrict__ img, float x, float y, int width) {
  const int ix = (int)x, iy = (int)y;
  const float4* bp = img + ix + iy * width;
  const float tl = bp[0].x, tr = bp[1].x, bl = bp[width].x, br = bp[width + 1].x;
  const float dx = x - ix, dy = y - iy;
  const float topInt = dx * tr + (1 - dx) * tl;
  const float botInt = dx * br + (1 - dx) * bl;
  const float leftInt = dy * bl + (1 - dy) * tl;
  const float rightInt = dy * br + (1 - dy) * tr;
  return make_float3(dx * rightInt + (1 - dx) * leftInt, rightInt - leftInt, botInt - topInt);
}

}  // namespace sdso

__global__ __launch_bounds__(256) void k_immature_init(const float4* __restrict__ img, int w, int n, const float* __restrict__ u,
                                                       const float* __restrict__ v, float* __restrict__ color, float* __restrict__ weights,
                                                       float* __restrict__ gradH, float* __restrict__ energyTH) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  float g0 = 0, g1 = 0, g2 = 0, g3 = 0;
  bool bad = false;
  for (int idx = 0; idx < 8; idx++) {
    const float3 ptc = interp33BiLin(img, u[p] + c_pat[idx][0], v[p] + c_pat[idx][1], w);
    color[p * 8 + idx] = ptc.x;
    if (!isfinite(ptc.x)) { bad = true; break; }
    g0 += ptc.y * ptc.y; g1 += ptc.y * ptc.z; g2 += ptc.z * ptc.y; g3 += ptc.z * ptc.z;
    weights[p * 8 + idx] = sqrtf(kOutlierTHSumComponent / (kOutlierTHSumComponent + (ptc.y * ptc.y + ptc.z * ptc.z)));
  }
  gradH[p * 4 + 0] = g0; gradH[p * 4 + 1] = g1; gradH[p * 4 + 2] = g2; gradH[p * 4 + 3] = g3;
  float e = 8 * kOutlierTH;
  e *= 1.0f * 1.0f;
  energyTH[p] = bad ? NAN : e;
}

// ImmaturePoint::traceStereo (ImmaturePoint.cpp:94-451), block organisation: a 256-thread workgroup owns PTS (16) points.
//   phase 1  thread t < 64 : the search geometry of point t, one LANE per point (the reference's scalar code; every early exit of
//                            ImmaturePoint.cpp:118-238 is a per-lane exit) -> numSteps, start, direction in LDS
//   phase 2  wave w        : the discrete searches of its points, one after the other, lanes = steps
//   phase 3  thread t < 64 : sub-pixel refinement of point t with the 8 pattern pixels in a serial loop (the reference's order by
//                            construction), interval update, outputs
// (Rounds 1-3 ran one wave per point — every instruction of the geometry at 1 / 64 and of the refinement at 8 / 64 lanes, VALU-issue bound
// at 1 350 VALU instructions per point — and kept that kernel, an LDS-band variant of it and an 8-waves-per-SIMD build for A/B until
// round 5: 52-57 us against 33 us per 20 000 points, profiles/README.md.)  Same expressions, same operation order as the reference:
// bit-identical outputs.
// value of lane 8 (lane / 8) + K of a group of eight lanes (ds_swizzle_b32, bit mode: and 0x18, or K, inside each half of the wave)
template <int K>
__device__ __forceinline__ float tr_bcast8(float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x18 | (K << 5))); }
template <int K>
__device__ __forceinline__ int tr_bcast8(int v) { return __builtin_amdgcn_ds_swizzle(v, 0x18 | (K << 5)); }
template <int K>
__device__ __forceinline__ double tr_bcast8(double v) {
  const unsigned long long u = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_ds_swizzle((int)(unsigned)u, 0x18 | (K << 5)), hi = (unsigned)__builtin_amdgcn_ds_swizzle((int)(unsigned)(u >> 32), 0x18 | (K << 5));
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
template <int GN_MODE, int PTS>   // PTS points per 256-thread workgroup (PTS / 4 searches per wave)
__global__ __launch_bounds__(256) void k_trace_stereo_blk(TraceDev T) {
  __shared__ int s_steps[PTS];                                   // 0: the point does not reach the search
  __shared__ float s_bE[PTS], s_bX[PTS], s_bY[PTS], s_second[PTS];
  __shared__ int s_bI[PTS];
  __shared__ float s_dx[PTS], s_dy[PTS], s_rU[PTS], s_rV[PTS], s_rE[PTS];   // search direction; the refined position and energy (phase 3a -> 3b)
  // the sample positions of every step of every point: ptx is the reference's running sum (ptx += dx, one rounding per step), so step s needs
  // s dependent additions — run by the point's OWN lane in phase 1 (64 points per instruction) instead of by every step lane of phase 2
  // (where the 45-trip loop was 40 % of a search's instructions).  Row stride 101: the 16..64 point lanes write distinct banks.
  constexpr int kStepCap = 100, kStepLd = kStepCap + 1;
  __shared__ float s_px[PTS * kStepLd], s_py[PTS * kStepLd];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int i = blockIdx.x * PTS + t;
  const float4* __restrict__ dI = T.img;
  const int wG0 = T.w, hG0 = T.h;
  bool live = false;                                            // the point goes on to the search and the refinement
  // per-lane state that survives the barriers (thread t of phase 1 is thread t of phase 3)
  float u_stereo = 0, v_stereo = 0, idepth_min_stereo = 0, idepth_max_stereo = 0, energyTH = 0, quality = 0, errorInPixel = 0, dx = 0, dy = 0, bf = 0;
  float Kt[3] = {0, 0, 0}, pr[3] = {0, 0, 0};
  int numSteps = 0;
  uint8_t prevStatus = 0;
  auto finish = [&](int st, float uvx, float uvy, float interval, bool writeUV) {
    T.lastTraceStatus[i] = (uint8_t)st;
    if (T.status) T.status[i] = (uint8_t)st;
    if (writeUV) { T.lastTraceUV[i * 2] = uvx; T.lastTraceUV[i * 2 + 1] = uvy; T.lastTracePixelInterval[i] = interval; }
    T.quality[i] = quality;
  };
  if (t < PTS) {
    s_steps[t] = 0;
    if (i < T.n) {
      if (T.skip && T.skip[i]) { if (T.status) T.status[i] = 255; }
      else do {
        u_stereo = T.u_stereo[i]; v_stereo = T.v_stereo[i];
        idepth_min_stereo = T.idepth_min_stereo[i]; idepth_max_stereo = T.idepth_max_stereo[i];
        const float* gradH = T.gradH + (size_t)i * 4;
        const float idepth_min = T.idepth_min[i];
        energyTH = T.energyTH[i];
        quality = T.quality[i];
        prevStatus = T.lastTraceStatus[i];
        const float bl0 = T.mode_right ? -T.baseline : T.baseline;
        Kt[0] = (T.fx * bl0 + 0.0f * 0.0f) + T.cx * 0.0f;
        Kt[1] = (0.0f * bl0 + T.fy * 0.0f) + T.cy * 0.0f;
        Kt[2] = (0.0f * bl0 + 0.0f * 0.0f) + 1.0f * 0.0f;
        bf = -T.fx * bl0;
        pr[0] = (1.0f * u_stereo + 0.0f * v_stereo) + 0.0f * 1.0f;
        pr[1] = (0.0f * u_stereo + 1.0f * v_stereo) + 0.0f * 1.0f;
        pr[2] = (0.0f * u_stereo + 0.0f * v_stereo) + 1.0f * 1.0f;
        float ptpMin[3];
#pragma unroll
        for (int k = 0; k < 3; k++) ptpMin[k] = pr[k] + Kt[k] * idepth_min_stereo;
        const float uMin = ptpMin[0] / ptpMin[2];
        const float vMin = ptpMin[1] / ptpMin[2];
        if (!(uMin > 4 && vMin > 4 && uMin < wG0 - 5 && vMin < hG0 - 5)) { finish(IPS_OOB, -1, -1, 0, true); break; }
        float dist, uMax, vMax, ptpMax[3];
        const float maxPixSearch = (wG0 + hG0) * kMaxPixSearch;
        const bool finiteMax = isfinite(idepth_max_stereo);
        if (finiteMax) {
#pragma unroll
          for (int k = 0; k < 3; k++) ptpMax[k] = pr[k] + Kt[k] * idepth_max_stereo;
          uMax = ptpMax[0] / ptpMax[2];
          vMax = ptpMax[1] / ptpMax[2];
          if (!(uMax > 4 && vMax > 4 && uMax < wG0 - 5 && vMax < hG0 - 5)) { finish(IPS_OOB, -1, -1, 0, true); break; }
          dist = (uMin - uMax) * (uMin - uMax) + (vMin - vMax) * (vMin - vMax);
          dist = sqrtf(dist);
          if (dist < kTraceSlackInterval) { finish(IPS_SKIPPED, 0, 0, 0, false); break; }
        } else {
          dist = maxPixSearch;
#pragma unroll
          for (int k = 0; k < 3; k++) ptpMax[k] = pr[k] + Kt[k] * 0.01f;
          uMax = ptpMax[0] / ptpMax[2];
          vMax = ptpMax[1] / ptpMax[2];
          const float ddx = uMax - uMin;
          const float ddy = vMax - vMin;
          const float d = 1.0f / sqrtf(ddx * ddx + ddy * ddy);
          uMax = uMin + dist * ddx * d;
          vMax = vMin + dist * ddy * d;
          if (!(uMax > 4 && vMax > 4 && uMax < wG0 - 5 && vMax < hG0 - 5)) { finish(IPS_OOB, -1, -1, 0, true); break; }
        }
        if (!(idepth_min < 0 || (ptpMin[2] > 0.75 && ptpMin[2] < 1.5))) { finish(IPS_OOB, -1, -1, 0, true); break; }
        dx = kTraceStepsize * (uMax - uMin);
        dy = kTraceStepsize * (vMax - vMin);
        const float a = (dx * gradH[0] + dy * gradH[2]) * dx + (dx * gradH[1] + dy * gradH[3]) * dy;
        const float b = (dy * gradH[0] + (-dx) * gradH[2]) * dy + (dy * gradH[1] + (-dx) * gradH[3]) * (-dx);
        errorInPixel = 0.2f + 0.2f * (a + b) / a;
        if (errorInPixel * kTraceMinImprovement > dist && finiteMax) { finish(IPS_BADCONDITION, 0, 0, 0, false); break; }
        if (errorInPixel > 10) errorInPixel = 10;
        dx /= dist;
        dy /= dist;
        if (dist > maxPixSearch) {
          uMax = uMin + maxPixSearch * dx;
          vMax = vMin + maxPixSearch * dy;
          dist = maxPixSearch;
        }
        numSteps = 1.9999f + dist / kTraceStepsize;
        const float randShift = uMin * 1000 - floorf(uMin * 1000);
        const float ptx0 = uMin - randShift * dx;
        const float pty0 = vMin - randShift * dy;
        if (!isfinite(dx) || !isfinite(dy)) { finish(IPS_OOB, -1, -1, 0, true); break; }
        if (numSteps >= 100) numSteps = 99;
        s_steps[t] = numSteps > 0 ? numSteps : 0;
        s_dx[t] = dx; s_dy[t] = dy;
        {
          float px = ptx0, py = pty0;
          for (int k = 0; k < numSteps; k++) { s_px[t * kStepLd + k] = px; s_py[t * kStepLd + k] = py; px += dx; py += dy; }
        }
        live = true;
      } while (false);
    }
  }
  __syncthreads();

  // ---- phase 2: discrete search, one wave per point, lane = step (ptx is the reference's running sum ptx += dx)
  // (unrolled over the wave's points, no branch around a point that does not search — its lanes are simply inactive — so that the
  // taps of the next point are in flight while the minimum of the current one is reduced)
#pragma unroll
  for (int pp = 0; pp < PTS / 4; pp++) {
    const int pt = wv * (PTS / 4) + pp;
    const int nsteps = s_steps[pt];
    const float* color = T.color + (size_t)(blockIdx.x * PTS + pt) * 8;
    float myE[2] = {1e30f, 1e30f}, myX[2] = {0, 0}, myY[2] = {0, 0};
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      const int s = pass * 64 + lane;
      if (s < nsteps) {
        const float ptx = s_px[pt * kStepLd + s], pty = s_py[pt * kStepLd + s];
        float energy = 0;
#pragma unroll
        for (int idx = 0; idx < 8; idx++) {
          const float hitColor = interp31_plane(T.plane, (float)(ptx + (float)c_pat[idx][0]), (float)(pty + (float)c_pat[idx][1]), wG0);
          if (!isfinite(hitColor)) { energy += 1e5; continue; }
          const float residual = hitColor - (float)(1.0f * color[idx] + 0.0f);
          const float hw = fabsf(residual) < kHuberTH ? 1 : kHuberTH / fabsf(residual);
          energy += hw * residual * residual * (2 - hw);
        }
        myE[pass] = energy; myX[pass] = ptx; myY[pass] = pty;
      }
    }
    // first minimum (the reference takes strictly smaller energies only, in step order)
    float bE = 1e10f; int bI = -1; float bX = 0, bY = 0;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      const int s = pass * 64 + lane;
      if (s < nsteps && myE[pass] < bE) { bE = myE[pass]; bI = s; bX = myX[pass]; bY = myY[pass]; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float oE = __shfl_xor(bE, o, 64); const int oI = __shfl_xor(bI, o, 64);
      const float oX = __shfl_xor(bX, o, 64), oY = __shfl_xor(bY, o, 64);
      const bool take = (oI >= 0) && (bI < 0 || oE < bE || (oE == bE && oI < bI));
      if (take) { bE = oE; bI = oI; bX = oX; bY = oY; }
    }
    const int bestIdx = bI;
    float secondBest = 1e10f;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      const int s = pass * 64 + lane;
      if (s < nsteps && (s < bestIdx - kMinTraceTestRadius || s > bestIdx + kMinTraceTestRadius) && myE[pass] < secondBest) secondBest = myE[pass];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) secondBest = fminf(secondBest, __shfl_xor(secondBest, o, 64));
    if (lane == 0 && nsteps > 0) { s_bE[pt] = bE; s_bI[pt] = bI; s_bX[pt] = bX; s_bY[pt] = bY; s_second[pt] = secondBest; }
  }
  __syncthreads();

  // ---- phase 3a: sub-pixel refinement, lane = (point, pattern pixel): the 8 samples of a pass are taken by 8 lanes at once and added in
  // pattern order on every lane of the group (the reference's sums, term by term); one lane per point walked them one after the other
  // (840 instructions and 6 dependent round trips per 16 points, with the other three waves of the workgroup idle)
#define TR_ALL8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
  for (int q = t; q < PTS * 8; q += 256) {
    const int pl = q >> 3, idx = q & 7;
    if (s_steps[pl] <= 0) continue;                             // (all eight lanes of a point alike)
    const size_t gi = (size_t)blockIdx.x * PTS + pl;
    const float col = T.color[gi * 8 + idx], wgt = T.weights[gi * 8 + idx];
    const float dx = s_dx[pl], dy = s_dy[pl];
    float bestU = s_bX[pl], bestV = s_bY[pl], bestEnergy = s_bE[pl];
    if (s_bI[pl] < 0) { bestU = 0; bestV = 0; bestEnergy = 1e10f; }
    const float patx = (float)c_pat[idx][0], paty = (float)c_pat[idx][1];
    if constexpr (GN_MODE == 1) {
      // fork-live refinement (ImmaturePoint.cpp:309-412): VertexUVDSO in double, 8 EdgeTracePointUVDSO (dso_g2o_edge.cpp:571-619) with
      // Huber(9), one undamped g2o Gauss-Newton step per pass, update clamped by VertexUVDSO::oplusImpl (dso_g2o_vertex.cpp:73-88)
      double U = bestU, V = bestV;
      const double ddx = dx, ddy = dy;
      if (kTraceGNIterations > 0) bestEnergy = 1e5;
      for (int it = 0; it < kTraceGNIterations; it++) {
        double e = 0, J = 0;
        const bool inside = !((U - 2) < 0 || (U + 3) > (wG0 - 3) || (V - 2) < 0 || (V + 3) > (hG0 - 3));
        const float3 hit = interp33(dI, inside ? (float)(U + patx) : 2.5f, inside ? (float)(V + paty) : 2.5f, wG0);   // (outside: pixel (2,2), ignored)
        if (inside && isfinite(hit.x)) {
          e = hit.x - (1.0f * (double)col + 0.0f);
          J = ddx * hit.y + ddy * hit.z;
        }
        const float residual = e;
        const float hw = fabsf(residual) < kHuberTH ? 1 : kHuberTH / fabsf(residual);
        const float te = wgt * wgt * hw * residual * residual * (2 - hw);
        const double e2 = e * e;
        const double rho1 = e2 <= (double)kHuberTH * kHuberTH ? 1. : kHuberTH / sqrt(e2);
        const double tb = rho1 * J * e, tH = J * rho1 * J;
        float energy = 0;
        double Hs = 0, bs = 0;
#define TR_ADD(K) energy += tr_bcast8<K>(te); bs -= tr_bcast8<K>(tb); Hs += tr_bcast8<K>(tH);
        TR_ALL8(TR_ADD)
#undef TR_ADD
        if (Hs != 0) {
          double update = bs / Hs;
          if (update < -0.5) update = -0.5;
          else if (update > 0.5) update = 0.5;
          else if (!isfinite(update)) update = 0;
          U += update * ddx;
          V += update * ddy;
        }
        if (!(energy > bestEnergy)) bestEnergy = energy;
      }
      bestU = U;
      bestV = V;
    } else {
      // DSO-native GN (ImmaturePoint.cpp:707-769)
      float uBak = bestU, vBak = bestV, stepBack = 0;
      const float gnstepsize = 1;
      if (kTraceGNIterations > 0) bestEnergy = 1e5;
      for (int it = 0; it < kTraceGNIterations; it++) {
        float tH = 0, tb = 0, te = 0;
        int nan = 0;
        const float3 hit = interp33(dI, (float)(bestU + patx), (float)(bestV + paty), wG0);
        if (!isfinite(hit.x)) nan = 1;
        else {
          const float residual = hit.x - (1.0f * col + 0.0f);
          const float dResdDist = dx * hit.y + dy * hit.z;
          const float hw = fabsf(residual) < kHuberTH ? 1 : kHuberTH / fabsf(residual);
          tH = hw * dResdDist * dResdDist;
          tb = hw * residual * dResdDist;
          te = wgt * wgt * hw * residual * residual * (2 - hw);
        }
        float H = 1, bb = 0, energy = 0;
#define TR_ADD(K) { const float h_ = tr_bcast8<K>(tH), b_ = tr_bcast8<K>(tb), e_ = tr_bcast8<K>(te); const int nn = tr_bcast8<K>(nan); \
                    if (nn) energy += 1e5; else { H += h_; bb += b_; energy += e_; } }
        TR_ALL8(TR_ADD)
#undef TR_ADD
        if (energy > bestEnergy) {
          stepBack *= 0.5;
          bestU = uBak + stepBack * dx;
          bestV = vBak + stepBack * dy;
        } else {
          float step = -gnstepsize * bb / H;
          if (step < -0.5) step = -0.5;
          else if (step > 0.5) step = 0.5;
          if (!isfinite(step)) step = 0;
          uBak = bestU;
          vBak = bestV;
          stepBack = step;
          bestU += step * dx;
          bestV += step * dy;
          bestEnergy = energy;
        }
        if (fabsf(stepBack) < kTraceGNThreshold) break;
      }
    }
    if (idx == 0) { s_rU[pl] = bestU; s_rV[pl] = bestV; s_rE[pl] = bestEnergy; }
  }
#undef TR_ALL8
  __syncthreads();

  // ---- phase 3b: outputs, one lane per point
  if (!live) return;
  {
    float bestEnergy0 = s_bE[t];
    if (s_bI[t] < 0) bestEnergy0 = 1e10f;
    const float newQuality = s_second[t] / bestEnergy0;
    if (newQuality < quality || numSteps > 10) quality = newQuality;
  }
  const float bestU = s_rU[t], bestV = s_rV[t], bestEnergy = s_rE[t];

  if (!(bestEnergy < energyTH * kTraceExtraSlack)) {
    finish(prevStatus == IPS_OUTLIER ? IPS_OOB : IPS_OUTLIER, -1, -1, 0, true);
    return;
  }
  if (dx * dx > dy * dy) {
    idepth_min_stereo = (pr[2] * (bestU - errorInPixel * dx) - pr[0]) / (Kt[0] - Kt[2] * (bestU - errorInPixel * dx));
    idepth_max_stereo = (pr[2] * (bestU + errorInPixel * dx) - pr[0]) / (Kt[0] - Kt[2] * (bestU + errorInPixel * dx));
  } else {
    idepth_min_stereo = (pr[2] * (bestV - errorInPixel * dy) - pr[1]) / (Kt[1] - Kt[2] * (bestV - errorInPixel * dy));
    idepth_max_stereo = (pr[2] * (bestV + errorInPixel * dy) - pr[1]) / (Kt[1] - Kt[2] * (bestV + errorInPixel * dy));
  }
  if (idepth_min_stereo > idepth_max_stereo) { const float tmp = idepth_min_stereo; idepth_min_stereo = idepth_max_stereo; idepth_max_stereo = tmp; }
  T.idepth_min_stereo[i] = idepth_min_stereo; T.idepth_max_stereo[i] = idepth_max_stereo;
  if (!isfinite(idepth_min_stereo) || !isfinite(idepth_max_stereo) || (idepth_max_stereo < 0)) { finish(IPS_OUTLIER, -1, -1, 0, true); return; }
  T.idepth_stereo[i] = (u_stereo - bestU) / bf;
  finish(IPS_GOOD, bestU, bestV, 2 * errorInPixel, true);
}

// ImmaturePoint::traceOn (ImmaturePoint.cpp:459-828): the same search along a general epipolar line.  geom[pgeom[i]] is the
// hostToFrame geometry of the point's host; T.idepth_min_stereo / idepth_max_stereo hold idepth_min / idepth_max.
__global__ __launch_bounds__(256) void k_trace_on(TraceDev T, const sdso_trace_geom_t* __restrict__ geom, const int* __restrict__ pgeom) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wv;
  if (i >= T.n) return;
  if (T.skip && T.skip[i]) { if (lane == 0 && T.status) T.status[i] = 255; return; }
  __shared__ float s_err[4][128];
  volatile float* errors = s_err[wv];
  const float4* __restrict__ dI = T.img;
  const int wG0 = T.w, hG0 = T.h;
  const float u_stereo = T.u_stereo[i], v_stereo = T.v_stereo[i];
  float idepth_min_stereo = T.idepth_min_stereo[i], idepth_max_stereo = T.idepth_max_stereo[i];
  const float* color = T.color + (size_t)i * 8;
  const float* weights = T.weights + (size_t)i * 8;
  const float* gradH = T.gradH + (size_t)i * 4;
  const float idepth_min = idepth_min_stereo, energyTH = T.energyTH[i];
  float quality = T.quality[i];
  const uint8_t prevStatus = T.lastTraceStatus[i];
  if (prevStatus == IPS_OOB) { if (lane == 0 && T.status) T.status[i] = IPS_OOB; return; }   // :466-468
  const sdso_trace_geom_t G = geom[pgeom[i]];

  auto finish = [&](int st, float uvx, float uvy, float interval, bool writeUV) {
    if (lane == 0) {
      T.lastTraceStatus[i] = (uint8_t)st;
      if (T.status) T.status[i] = (uint8_t)st;
      if (writeUV) { T.lastTraceUV[i * 2] = uvx; T.lastTraceUV[i * 2 + 1] = uvy; T.lastTracePixelInterval[i] = interval; }
      T.quality[i] = quality;
    }
  };

  float Kt[3] = {G.Kt[0], G.Kt[1], G.Kt[2]};
  float pr[3];
#pragma unroll
  for (int r = 0; r < 3; r++) pr[r] = (G.KRKi[r * 3 + 0] * u_stereo + G.KRKi[r * 3 + 1] * v_stereo) + G.KRKi[r * 3 + 2] * 1.0f;
  const float aff0 = G.aff[0], aff1 = G.aff[1];
  float rot[8][2];
#pragma unroll
  for (int idx = 0; idx < 8; idx++) {   // Rplane * patternP  (:628, :636-637)
    rot[idx][0] = G.KRKi[0] * (float)c_pat[idx][0] + G.KRKi[1] * (float)c_pat[idx][1];
    rot[idx][1] = G.KRKi[3] * (float)c_pat[idx][0] + G.KRKi[4] * (float)c_pat[idx][1];
  }
  float ptpMin[3];
#pragma unroll
  for (int k = 0; k < 3; k++) ptpMin[k] = pr[k] + Kt[k] * idepth_min_stereo;
  const float uMin = ptpMin[0] / ptpMin[2];
  const float vMin = ptpMin[1] / ptpMin[2];
  if (!(uMin > 4 && vMin > 4 && uMin < wG0 - 5 && vMin < hG0 - 5)) { finish(IPS_OOB, -1, -1, 0, true); return; }

  float dist, uMax, vMax, ptpMax[3];
  const float maxPixSearch = (wG0 + hG0) * kMaxPixSearch;
  const bool finiteMax = isfinite(idepth_max_stereo);
  if (finiteMax) {
#pragma unroll
    for (int k = 0; k < 3; k++) ptpMax[k] = pr[k] + Kt[k] * idepth_max_stereo;
    uMax = ptpMax[0] / ptpMax[2];
    vMax = ptpMax[1] / ptpMax[2];
    if (!(uMax > 4 && vMax > 4 && uMax < wG0 - 5 && vMax < hG0 - 5)) { finish(IPS_OOB, -1, -1, 0, true); return; }
    dist = (uMin - uMax) * (uMin - uMax) + (vMin - vMax) * (vMin - vMax);
    dist = sqrtf(dist);
    if (dist < kTraceSlackInterval) { finish(IPS_SKIPPED, (uMax + uMin) * 0.5f, (vMax + vMin) * 0.5f, dist, true); return; }   // :525-531
  } else {
    dist = maxPixSearch;
#pragma unroll
    for (int k = 0; k < 3; k++) ptpMax[k] = pr[k] + Kt[k] * 0.01f;
    uMax = ptpMax[0] / ptpMax[2];
    vMax = ptpMax[1] / ptpMax[2];
    const float ddx = uMax - uMin;
    const float ddy = vMax - vMin;
    const float d = 1.0f / sqrtf(ddx * ddx + ddy * ddy);
    uMax = uMin + dist * ddx * d;
    vMax = vMin + dist * ddy * d;
    if (!(uMax > 4 && vMax > 4 && uMax < wG0 - 5 && vMax < hG0 - 5)) { finish(IPS_OOB, -1, -1, 0, true); return; }
  }
  if (!(idepth_min < 0 || (ptpMin[2] > 0.75 && ptpMin[2] < 1.5))) { finish(IPS_OOB, -1, -1, 0, true); return; }

  float dx = kTraceStepsize * (uMax - uMin);
  float dy = kTraceStepsize * (vMax - vMin);
  const float a = (dx * gradH[0] + dy * gradH[2]) * dx + (dx * gradH[1] + dy * gradH[3]) * dy;
  const float b = (dy * gradH[0] + (-dx) * gradH[2]) * dy + (dy * gradH[1] + (-dx) * gradH[3]) * (-dx);
  float errorInPixel = 0.2f + 0.2f * (a + b) / a;
  if (errorInPixel * kTraceMinImprovement > dist && finiteMax) { finish(IPS_BADCONDITION, (uMax + uMin) * 0.5f, (vMax + vMin) * 0.5f, dist, true); return; }   // :596-603
  if (errorInPixel > 10) errorInPixel = 10;
  dx /= dist;
  dy /= dist;
  if (dist > maxPixSearch) {
    uMax = uMin + maxPixSearch * dx;
    vMax = vMin + maxPixSearch * dy;
    dist = maxPixSearch;
  }
  int numSteps = 1.9999f + dist / kTraceStepsize;
  const float randShift = uMin * 1000 - floorf(uMin * 1000);
  const float ptx0 = uMin - randShift * dx;
  const float pty0 = vMin - randShift * dy;
  if (!isfinite(dx) || !isfinite(dy)) { finish(IPS_OOB, -1, -1, 0, true); return; }
  if (numSteps >= 100) numSteps = 99;

  // ---- discrete search: lane = step (ptx is the reference's running sum ptx += dx)
  float myE[2] = {1e30f, 1e30f}, myX[2] = {0, 0}, myY[2] = {0, 0};
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    const int s = pass * 64 + lane;
    if (s < numSteps) {
      float ptx = ptx0, pty = pty0;
      for (int k = 0; k < s; k++) { ptx += dx; pty += dy; }
      float energy = 0;
#pragma unroll
      for (int idx = 0; idx < 8; idx++) {
        const float hitColor = interp31_plane(T.plane, (float)(ptx + rot[idx][0]), (float)(pty + rot[idx][1]), wG0);
        if (!isfinite(hitColor)) { energy += 1e5; continue; }
        const float residual = hitColor - (float)(aff0 * color[idx] + aff1);
        const float hw = fabsf(residual) < kHuberTH ? 1 : kHuberTH / fabsf(residual);
        energy += hw * residual * residual * (2 - hw);
      }
      errors[s] = energy;
      myE[pass] = energy; myX[pass] = ptx; myY[pass] = pty;
    }
  }
  // first minimum (the reference takes strictly smaller energies only, in step order)
  float bE = 1e10f; int bI = -1; float bX = 0, bY = 0;
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    const int s = pass * 64 + lane;
    if (s < numSteps && myE[pass] < bE) { bE = myE[pass]; bI = s; bX = myX[pass]; bY = myY[pass]; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float oE = __shfl_xor(bE, o, 64); const int oI = __shfl_xor(bI, o, 64);
    const float oX = __shfl_xor(bX, o, 64), oY = __shfl_xor(bY, o, 64);
    const bool take = (oI >= 0) && (bI < 0 || oE < bE || (oE == bE && oI < bI));
    if (take) { bE = oE; bI = oI; bX = oX; bY = oY; }
  }
  float bestU = bX, bestV = bY, bestEnergy = bE;
  const int bestIdx = bI;
  if (bestIdx < 0) { bestU = 0; bestV = 0; bestEnergy = 1e10f; }
  float secondBest = 1e10f;
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    const int s = pass * 64 + lane;
    if (s < numSteps && (s < bestIdx - kMinTraceTestRadius || s > bestIdx + kMinTraceTestRadius) && myE[pass] < secondBest) secondBest = myE[pass];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) secondBest = fminf(secondBest, __shfl_xor(secondBest, o, 64));
  const float newQuality = secondBest / bestEnergy;
  if (newQuality < quality || numSteps > 10) quality = newQuality;

  // ---- DSO-native GN (ImmaturePoint.cpp:707-769): pattern pixel idx on lane idx, summed in order
  float uBak = bestU, vBak = bestV, stepBack = 0;
  const float gnstepsize = 1;
  if (kTraceGNIterations > 0) bestEnergy = 1e5;
  for (int it = 0; it < kTraceGNIterations; it++) {
    float tH = 0, tb = 0, te = 0;
    bool nan = false;
    if (lane < 8) {
      const float3 hit = interp33(dI, (float)(bestU + rot[lane][0]), (float)(bestV + rot[lane][1]), wG0);
      if (!isfinite(hit.x)) nan = true;
      else {
        const float residual = hit.x - (aff0 * color[lane] + aff1);
        const float dResdDist = dx * hit.y + dy * hit.z;
        const float hw = fabsf(residual) < kHuberTH ? 1 : kHuberTH / fabsf(residual);
        tH = hw * dResdDist * dResdDist;
        tb = hw * residual * dResdDist;
        te = weights[lane] * weights[lane] * hw * residual * residual * (2 - hw);
      }
    }
    float H = 1, bb = 0, energy = 0;
#pragma unroll
    for (int idx = 0; idx < 8; idx++) {
      const float h_ = lane_bcast(tH, idx), b_ = lane_bcast(tb, idx), e_ = lane_bcast(te, idx);
      const int nn = lane_bcast((int)nan, idx);
      if (nn) { energy += 1e5; continue; }
      H += h_; bb += b_; energy += e_;
    }
    if (energy > bestEnergy) {
      stepBack *= 0.5;
      bestU = uBak + stepBack * dx;
      bestV = vBak + stepBack * dy;
    } else {
      float step = -gnstepsize * bb / H;
      if (step < -0.5) step = -0.5;
      else if (step > 0.5) step = 0.5;
      if (!isfinite(step)) step = 0;
      uBak = bestU;
      vBak = bestV;
      stepBack = step;
      bestU += step * dx;
      bestV += step * dy;
      bestEnergy = energy;
    }
    if (fabsf(stepBack) < kTraceGNThreshold) break;
  }

  if (!(bestEnergy < energyTH * kTraceExtraSlack)) {
    finish(prevStatus == IPS_OUTLIER ? IPS_OOB : IPS_OUTLIER, -1, -1, 0, true);
    return;
  }
  if (dx * dx > dy * dy) {
    idepth_min_stereo = (pr[2] * (bestU - errorInPixel * dx) - pr[0]) / (Kt[0] - Kt[2] * (bestU - errorInPixel * dx));
    idepth_max_stereo = (pr[2] * (bestU + errorInPixel * dx) - pr[0]) / (Kt[0] - Kt[2] * (bestU + errorInPixel * dx));
  } else {
    idepth_min_stereo = (pr[2] * (bestV - errorInPixel * dy) - pr[1]) / (Kt[1] - Kt[2] * (bestV - errorInPixel * dy));
    idepth_max_stereo = (pr[2] * (bestV + errorInPixel * dy) - pr[1]) / (Kt[1] - Kt[2] * (bestV + errorInPixel * dy));
  }
  if (idepth_min_stereo > idepth_max_stereo) { const float t = idepth_min_stereo; idepth_min_stereo = idepth_max_stereo; idepth_max_stereo = t; }
  if (lane == 0) { T.idepth_min_stereo[i] = idepth_min_stereo; T.idepth_max_stereo[i] = idepth_max_stereo; }
  if (!isfinite(idepth_min_stereo) || !isfinite(idepth_max_stereo) || (idepth_max_stereo < 0)) { finish(IPS_OUTLIER, -1, -1, 0, true); return; }
  finish(IPS_GOOD, bestU, bestV, 2 * errorInPixel, true);
}

// ------------------------------------------------------------------ API
extern "C" int sdso_immature_init_batch(sdso_ctx* ctx, int frame_slot, int n, const float* u, const float* v, float* color, float* weights,
                                        float* gradH, float* energyTH) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  auto ip = ctx->pyr.find(frame_slot);
  SDSO_REQUIRE(ctx, ip != ctx->pyr.end(), "unknown frame slot");
  SDSO_REQUIRE(ctx, n >= 0 && (n == 0 || (u && v && color && weights && gradH && energyTH)), "null argument");
  if (n == 0) return SDSO_OK;
  const int w = ip->second.w[0], h = ip->second.h[0];
  for (int i = 0; i < n; i++)  // the reference dereferences the 3x3 neighbourhood unchecked; refuse instead of faulting
    SDSO_REQUIRE(ctx, u[i] >= 2 && v[i] >= 2 && u[i] < w - 3 && v[i] < h - 3, "immature point too close to the image border");
  int rc = ensure_scratch(ctx, sizeof(float) * (size_t)n * 23);
  if (rc) return rc;
  float* d = (float*)ctx->scratch;
  float *du = d, *dv = d + n, *dc = d + 2 * (size_t)n, *dw = d + 10 * (size_t)n, *dg = d + 18 * (size_t)n, *de = d + 22 * (size_t)n;
  SDSO_HIP(ctx, hipMemcpyAsync(du, u, sizeof(float) * n, hipMemcpyHostToDevice, ctx->stream));
  SDSO_HIP(ctx, hipMemcpyAsync(dv, v, sizeof(float) * n, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_immature_init, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ip->second.d[0], w, n, du, dv, dc, dw, dg, de);
  SDSO_HIP(ctx, hipGetLastError());
  SDSO_HIP(ctx, hipMemcpyAsync(color, dc, sizeof(float) * 8 * n, hipMemcpyDeviceToHost, ctx->stream));
  SDSO_HIP(ctx, hipMemcpyAsync(weights, dw, sizeof(float) * 8 * n, hipMemcpyDeviceToHost, ctx->stream));
  SDSO_HIP(ctx, hipMemcpyAsync(gradH, dg, sizeof(float) * 4 * n, hipMemcpyDeviceToHost, ctx->stream));
  SDSO_HIP(ctx, hipMemcpyAsync(energyTH, de, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SDSO_OK;
}

namespace sdso {
// device-resident batch used by both the host-buffer entry point and the benchmark
struct TraceBatch {
  TraceDev T;
  float* blob = nullptr;   // all float arrays
  uint8_t* bytes = nullptr;
  int n = 0;
};
static std::map<sdso_ctx*, TraceBatch> g_trace;

static int trace_reserve(sdso_ctx* ctx, TraceBatch& B, int n) {
  if (B.n >= n) return SDSO_OK;
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (B.blob) hipFree(B.blob);
  if (B.bytes) hipFree(B.bytes);
  B.n = n + n / 4 + 64;
  SDSO_HIP(ctx, hipMalloc(&B.blob, sizeof(float) * (size_t)B.n * 36));
  SDSO_HIP(ctx, hipMalloc(&B.bytes, (size_t)B.n * 3));
  return SDSO_OK;
}
static void trace_bind(TraceBatch& B, int n) {
  float* f = B.blob;
  const size_t N = B.n;
  TraceDev& T = B.T;
  T.n = n;
  // in/out fields {idepth_min_stereo, idepth_max_stereo, quality} are contiguous (3N floats at 3N) with a
  // pristine copy at 32N, so that enqueue can be repeated on identical input (benchmark)
  T.u_stereo = f; T.v_stereo = f + N; T.idepth_min = f + 2 * N; T.idepth_min_stereo = f + 3 * N; T.idepth_max_stereo = f + 4 * N; T.quality = f + 5 * N;
  T.idepth_stereo = f + 6 * N; T.color = f + 7 * N; T.weights = f + 15 * N; T.gradH = f + 23 * N; T.energyTH = f + 27 * N;
  T.lastTraceUV = f + 28 * N; T.lastTracePixelInterval = f + 30 * N;
  T.lastTraceStatus = B.bytes; T.status = B.bytes + N; T.skip = nullptr;
}
void release_trace(sdso_ctx* ctx) {
  TraceBatch tb;
  if (!reg_take(g_trace, ctx, tb)) return;
  if (tb.blob) hipFree(tb.blob);
  if (tb.bytes) hipFree(tb.bytes);
}
}  // namespace sdso

// upload the point state once (benchmark: state stays in HBM, enqueue-only launches follow)
extern "C" int sdso_trace_stereo_prepare(sdso_ctx* ctx, int frame_slot, const float K[4], float baseline, int mode_right, const sdso_trace_points_t* P) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  SDSO_REQUIRE(ctx, K && P && P->n >= 0, "null argument");
  auto ip = ctx->pyr.find(frame_slot);
  SDSO_REQUIRE(ctx, ip != ctx->pyr.end(), "unknown frame slot");
  const int n = P->n;
  TraceBatch& B = reg_get(g_trace, ctx);
  int rc = trace_reserve(ctx, B, std::max(n, 1));
  if (rc) return rc;
  trace_bind(B, n);
  TraceDev& T = B.T;
  rc = ensure_plane0(ctx, ip->second);
  if (rc) return rc;
  T.w = ip->second.w[0]; T.h = ip->second.h[0]; T.mode_right = mode_right; T.img = ip->second.d[0]; T.plane = ip->second.plane0;
  T.fx = K[0]; T.fy = K[1]; T.cx = K[2]; T.cy = K[3]; T.baseline = baseline;
#define UP(dst, src, cnt) if (n) SDSO_HIP(ctx, hipMemcpyAsync((void*)(dst), (src), sizeof(float) * (size_t)(cnt), hipMemcpyHostToDevice, ctx->stream))
  UP(T.u_stereo, P->u_stereo, n); UP(T.v_stereo, P->v_stereo, n); UP(T.idepth_min, P->idepth_min, n);
  UP(T.idepth_min_stereo, P->idepth_min_stereo, n); UP(T.idepth_max_stereo, P->idepth_max_stereo, n); UP(T.idepth_stereo, P->idepth_stereo, n);
  UP(T.color, P->color, 8 * n); UP(T.weights, P->weights, 8 * n); UP(T.gradH, P->gradH, 4 * n); UP(T.energyTH, P->energyTH, n); UP(T.quality, P->quality, n);
  UP(T.lastTraceUV, P->lastTraceUV, 2 * n); UP(T.lastTracePixelInterval, P->lastTracePixelInterval, n);
#undef UP
  if (n) SDSO_HIP(ctx, hipMemcpyAsync(T.lastTraceStatus, P->lastTraceStatus, n, hipMemcpyHostToDevice, ctx->stream));
  // pristine copies of the in/out fields
  SDSO_HIP(ctx, hipMemcpyAsync(B.blob + 32 * (size_t)B.n, B.blob + 3 * (size_t)B.n, sizeof(float) * 3 * (size_t)B.n, hipMemcpyDeviceToDevice, ctx->stream));
  SDSO_HIP(ctx, hipMemcpyAsync(B.bytes + 2 * (size_t)B.n, B.bytes, (size_t)B.n, hipMemcpyDeviceToDevice, ctx->stream));
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SDSO_OK;
}
// the traceStereo kernel in the ctx's refinement mode (0 DSO-native GN, 1 the fork's g2o passes)
static void launch_trace_stereo(sdso_ctx* ctx, const TraceDev& T, bool timed = false /* the benchmarked enqueue: the kernel's own duration under the name k_trace_stereo */) {
  constexpr int P = 16;
  const dim3 g((T.n + P - 1) / P), b(256);
  if (ctx->gn_mode == 1) { if (timed) launch_timed(ctx, "k_trace_stereo", 1, (k_trace_stereo_blk<1, P>), g, b, T); else hipLaunchKernelGGL((k_trace_stereo_blk<1, P>), g, b, 0, ctx->stream, T); }
  else { if (timed) launch_timed(ctx, "k_trace_stereo", 1, (k_trace_stereo_blk<0, P>), g, b, T); else hipLaunchKernelGGL((k_trace_stereo_blk<0, P>), g, b, 0, ctx->stream, T); }
}
extern "C" int sdso_trace_stereo_enqueue(sdso_ctx* ctx) {
  if (!ctx || !reg_has(g_trace, ctx)) return sdso::fail(ctx, SDSO_ERR_STATE, "no prepared trace batch");
  TraceBatch& B = reg_get(g_trace, ctx);
  if (B.T.n == 0) return SDSO_OK;
  SDSO_HIP(ctx, hipMemcpyAsync(B.blob + 3 * (size_t)B.n, B.blob + 32 * (size_t)B.n, sizeof(float) * 3 * (size_t)B.n, hipMemcpyDeviceToDevice, ctx->stream));
  SDSO_HIP(ctx, hipMemcpyAsync(B.bytes, B.bytes + 2 * (size_t)B.n, (size_t)B.n, hipMemcpyDeviceToDevice, ctx->stream));
  launch_trace_stereo(ctx, B.T, true);
  SDSO_HIP(ctx, hipGetLastError());
  return SDSO_OK;
}
extern "C" int sdso_trace_stereo_fetch(sdso_ctx* ctx, sdso_trace_points_t* P, uint8_t* status) {
  if (!ctx || !reg_has(g_trace, ctx)) return sdso::fail(ctx, SDSO_ERR_STATE, "no prepared trace batch");
  TraceBatch& B = reg_get(g_trace, ctx);
  const int n = B.T.n;
  TraceDev& T = B.T;
#define DN(dst, src, cnt) if (n && dst) SDSO_HIP(ctx, hipMemcpyAsync((dst), (src), sizeof(float) * (size_t)(cnt), hipMemcpyDeviceToHost, ctx->stream))
  if (P) {
    DN(P->idepth_min_stereo, T.idepth_min_stereo, n); DN(P->idepth_max_stereo, T.idepth_max_stereo, n); DN(P->idepth_stereo, T.idepth_stereo, n);
    DN(P->quality, T.quality, n); DN(P->lastTraceUV, T.lastTraceUV, 2 * n); DN(P->lastTracePixelInterval, T.lastTracePixelInterval, n);
    if (n && P->lastTraceStatus) SDSO_HIP(ctx, hipMemcpyAsync(P->lastTraceStatus, T.lastTraceStatus, n, hipMemcpyDeviceToHost, ctx->stream));
  }
#undef DN
  if (n && status) SDSO_HIP(ctx, hipMemcpyAsync(status, T.status, n, hipMemcpyDeviceToHost, ctx->stream));
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SDSO_OK;
}

extern "C" int sdso_trace_on_batch(sdso_ctx* ctx, int frame_slot, int ngeom, const sdso_trace_geom_t* geom, const int* point_geom,
                                   sdso_trace_points_t* pts, uint8_t* status) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  SDSO_REQUIRE(ctx, pts && pts->n >= 0 && ngeom >= 0, "null argument");
  const int n = pts->n;
  if (n == 0) return SDSO_OK;
  SDSO_REQUIRE(ctx, geom && point_geom && ngeom > 0, "null geometry");
  for (int i = 0; i < n; i++) SDSO_REQUIRE(ctx, point_geom[i] >= 0 && point_geom[i] < ngeom, "point_geom out of range");
  const float K0[4] = {1, 1, 0, 0};
  if (!pts->idepth_min) pts->idepth_min = pts->idepth_min_stereo;       // not used by traceOn; the upload needs a valid pointer
  if (!pts->idepth_stereo) pts->idepth_stereo = pts->idepth_min_stereo;
  int rc = sdso_trace_stereo_prepare(ctx, frame_slot, K0, 0.f, 1, pts);   // uploads the point state into the ctx's trace batch
  if (rc) return rc;
  TraceBatch& B = reg_get(g_trace, ctx);
  rc = ensure_scratch(ctx, sizeof(sdso_trace_geom_t) * (size_t)ngeom + sizeof(int) * (size_t)n);
  if (rc) return rc;
  sdso_trace_geom_t* d_geom = (sdso_trace_geom_t*)ctx->scratch;
  int* d_pg = (int*)(d_geom + ngeom);
  SDSO_HIP(ctx, hipMemcpyAsync(d_geom, geom, sizeof(sdso_trace_geom_t) * ngeom, hipMemcpyHostToDevice, ctx->stream));
  SDSO_HIP(ctx, hipMemcpyAsync(d_pg, point_geom, sizeof(int) * n, hipMemcpyHostToDevice, ctx->stream));
  {
    ProfScope ps(ctx, "k_trace_on");
    hipLaunchKernelGGL(k_trace_on, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, B.T, (const sdso_trace_geom_t*)d_geom, (const int*)d_pg);
  }
  SDSO_HIP(ctx, hipGetLastError());
  sdso_trace_points_t out = *pts;
  out.idepth_stereo = nullptr;
  return sdso_trace_stereo_fetch(ctx, &out, status);
}

// 0: DSO-native sub-pixel refinement (ImmaturePoint.cpp:707-769); 1: the fork's live g2o Gauss-Newton on
// EdgeTracePointUVDSO (ImmaturePoint.cpp:309-412).  Applies to every later traceStereo launch of this ctx.
extern "C" int sdso_trace_set_gn_mode(sdso_ctx* ctx, int mode) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_REQUIRE(ctx, mode == 0 || mode == 1, "gn mode must be 0 (DSO-native) or 1 (g2o fork)");
  ctx->gn_mode = mode;
  return SDSO_OK;
}

extern "C" int sdso_trace_stereo_batch(sdso_ctx* ctx, int frame_slot, const float K[4], float baseline, int mode_right, sdso_trace_points_t* pts, uint8_t* status) {
  int rc = sdso_trace_stereo_prepare(ctx, frame_slot, K, baseline, mode_right, pts);
  if (rc) return rc;
  rc = sdso_trace_stereo_enqueue(ctx);
  if (rc) return rc;
  return sdso_trace_stereo_fetch(ctx, pts, status);
}


// ------------------------------------------------------------------ left-right-left matching (S4)
// The callers of traceStereo all run the same three steps per point (FullSystem::stereoMatch FullSystem.cpp:581-613,
// traceNewCoarseNonKey :667-725, CoarseTracker::makeCoarseDepthL0 CoarseTracker.cpp:295-347):
//   ImmaturePoint(u, v, frameA) -> traceStereo(frameB)                       [forward]
//   if GOOD: ImmaturePoint(lastTraceUV, frameB) -> traceStereo(frameA)        [back]
// and then compare u with the back trace's lastTraceUV(0).  Everything stays on the device between the steps.
__global__ __launch_bounds__(256) void k_match_prepare(int n, const float* __restrict__ u, const float* __restrict__ v, const float* __restrict__ imin,
                                                       const float* __restrict__ imax, TraceDev T) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  T.u_stereo[p] = u[p]; T.v_stereo[p] = v[p];
  T.idepth_min[p] = 0.f;
  T.idepth_min_stereo[p] = imin ? imin[p] : 0.f;
  T.idepth_max_stereo[p] = imax ? imax[p] : NAN;
  T.idepth_stereo[p] = 0.f; T.quality[p] = 10000.f; T.lastTraceStatus[p] = IPS_UNINITIALIZED;   // ImmaturePoint.cpp:34-38
  T.lastTraceUV[2 * p] = 0.f; T.lastTraceUV[2 * p + 1] = 0.f; T.lastTracePixelInterval[p] = 0.f;
}
// back points at the forward trace's lastTraceUV; points whose forward trace was not GOOD are skipped (and parked on a
// harmless pixel so that the constructor kernel reads valid memory)
__global__ __launch_bounds__(256) void k_match_back_points(int n, TraceDev F, TraceDev Bk, uint8_t* __restrict__ skip, const float* __restrict__ bmin,
                                                           const float* __restrict__ bmax) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const bool good = F.status[p] == IPS_GOOD;
  skip[p] = good ? 0 : 1;
  Bk.u_stereo[p] = good ? F.lastTraceUV[2 * p] : 8.f;
  Bk.v_stereo[p] = good ? F.lastTraceUV[2 * p + 1] : 8.f;
  Bk.idepth_min[p] = 0.f;
  Bk.idepth_min_stereo[p] = bmin ? bmin[p] : 0.f;
  Bk.idepth_max_stereo[p] = bmax ? bmax[p] : NAN;
  Bk.idepth_stereo[p] = 0.f; Bk.quality[p] = 10000.f; Bk.lastTraceStatus[p] = IPS_UNINITIALIZED;
  Bk.lastTraceUV[2 * p] = 0.f; Bk.lastTraceUV[2 * p + 1] = 0.f; Bk.lastTracePixelInterval[p] = 0.f;
}

namespace sdso {
static std::map<sdso_ctx*, TraceBatch> g_match[2];
void release_match(sdso_ctx* ctx) {
  for (int k = 0; k < 2; k++) {
    TraceBatch tb;
    if (!reg_take(g_match[k], ctx, tb)) continue;
    if (tb.blob) hipFree(tb.blob);
    if (tb.bytes) hipFree(tb.bytes);
  }
}
}  // namespace sdso

extern "C" int sdso_stereo_match_batch(sdso_ctx* ctx, int slot_a, int slot_b, const float K[4], float baseline, int mode_right_first,
                                       sdso_stereo_match_t* M) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  SDSO_REQUIRE(ctx, K && M && M->n >= 0, "null argument");
  auto ia = ctx->pyr.find(slot_a), ib = ctx->pyr.find(slot_b);
  SDSO_REQUIRE(ctx, ia != ctx->pyr.end() && ib != ctx->pyr.end(), "unknown frame slot");
  SDSO_REQUIRE(ctx, ia->second.w[0] == ib->second.w[0] && ia->second.h[0] == ib->second.h[0], "the two frames differ in size");
  const int n = M->n, w = ia->second.w[0], h = ia->second.h[0];
  if (n == 0) return SDSO_OK;
  SDSO_REQUIRE(ctx, M->u && M->v, "null point arrays");
  for (int i = 0; i < n; i++)
    SDSO_REQUIRE(ctx, M->u[i] >= 2 && M->v[i] >= 2 && M->u[i] < w - 3 && M->v[i] < h - 3, "immature point too close to the image border");
  TraceBatch& A = reg_get(g_match[0], ctx);
  TraceBatch& Bk = reg_get(g_match[1], ctx);
  int rc = trace_reserve(ctx, A, n);
  if (rc) return rc;
  rc = trace_reserve(ctx, Bk, n);
  if (rc) return rc;
  trace_bind(A, n); trace_bind(Bk, n);
  rc = ensure_plane0(ctx, ia->second);
  if (rc) return rc;
  rc = ensure_plane0(ctx, ib->second);
  if (rc) return rc;
  auto geom = [&](TraceDev& T, const float4* img, const float* plane, int mode_right) {
    T.w = w; T.h = h; T.mode_right = mode_right; T.img = img; T.plane = plane;
    T.fx = K[0]; T.fy = K[1]; T.cx = K[2]; T.cy = K[3]; T.baseline = baseline;
  };
  geom(A.T, ib->second.d[0], ib->second.plane0, mode_right_first ? 1 : 0);       // forward: points of frame A searched in frame B
  geom(Bk.T, ia->second.d[0], ia->second.plane0, mode_right_first ? 0 : 1);      // back: points of frame B searched in frame A
  // host inputs -> device (the unused tail of the back batch's float blob is the staging area: 32N..36N)
  float* stage = Bk.blob + 32 * (size_t)Bk.n;
  const float* d_in[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  const float* h_in[6] = {M->u, M->v, M->idepth_min_stereo, M->idepth_max_stereo, M->back_idepth_min_stereo, M->back_idepth_max_stereo};
  rc = ensure_scratch(ctx, sizeof(float) * 6 * (size_t)n);
  if (rc) return rc;
  (void)stage;
  for (int k = 0; k < 6; k++)
    if (h_in[k]) {
      float* d = (float*)ctx->scratch + (size_t)k * n;
      SDSO_HIP(ctx, hipMemcpyAsync(d, h_in[k], sizeof(float) * n, hipMemcpyHostToDevice, ctx->stream));
      d_in[k] = d;
    }
  const dim3 g1((n + 255) / 256), b1(256), gw((n + 3) / 4);
  uint8_t* skip = Bk.bytes + 2 * (size_t)Bk.n;
  hipLaunchKernelGGL(k_match_prepare, g1, b1, 0, ctx->stream, n, d_in[0], d_in[1], d_in[2], d_in[3], A.T);
  hipLaunchKernelGGL(k_immature_init, g1, b1, 0, ctx->stream, ia->second.d[0], w, n, (const float*)A.T.u_stereo, (const float*)A.T.v_stereo,
                     (float*)A.T.color, (float*)A.T.weights, (float*)A.T.gradH, (float*)A.T.energyTH);
  launch_trace_stereo(ctx, A.T, true);      // (timed under k_trace_stereo when profiling is on: the match chain's two traces)
  hipLaunchKernelGGL(k_match_back_points, g1, b1, 0, ctx->stream, n, A.T, Bk.T, skip, d_in[4], d_in[5]);
  hipLaunchKernelGGL(k_immature_init, g1, b1, 0, ctx->stream, ib->second.d[0], w, n, (const float*)Bk.T.u_stereo, (const float*)Bk.T.v_stereo,
                     (float*)Bk.T.color, (float*)Bk.T.weights, (float*)Bk.T.gradH, (float*)Bk.T.energyTH);
  TraceDev Tb = Bk.T;
  Tb.skip = skip;
  launch_trace_stereo(ctx, Tb, true);
  SDSO_HIP(ctx, hipGetLastError());
#define DN(dst, src, cnt) if (dst) SDSO_HIP(ctx, hipMemcpyAsync((dst), (src), sizeof(float) * (size_t)(cnt), hipMemcpyDeviceToHost, ctx->stream))
  DN(M->idepth_stereo, A.T.idepth_stereo, n); DN(M->idepth_min_out, A.T.idepth_min_stereo, n); DN(M->idepth_max_out, A.T.idepth_max_stereo, n);
  DN(M->fwd_uv, A.T.lastTraceUV, 2 * n); DN(M->back_uv, Bk.T.lastTraceUV, 2 * n);
#undef DN
  if (M->status_fwd) SDSO_HIP(ctx, hipMemcpyAsync(M->status_fwd, A.T.status, n, hipMemcpyDeviceToHost, ctx->stream));
  if (M->status_back) SDSO_HIP(ctx, hipMemcpyAsync(M->status_back, Bk.T.status, n, hipMemcpyDeviceToHost, ctx->stream));
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SDSO_OK;
}

// ------------------------------------------------------------------ point activation (N3)
// FullSystem::optimizeImmaturePoint (src/FullSystem/FullSystemOptPoint.cpp:52-238) + ImmaturePoint::linearizeResidual
// (src/FullSystem/ImmaturePoint.cpp:886-985), the DSO-native idepth-only Gauss-Newton (kept as comments in the fork).
// One wave per immature point: lane (r, idx) = (lane>>3, lane&7) evaluates pattern pixel idx of the residual to the r-th
// other keyframe (<= 7 residuals x 8 pixels = 56 lanes) — projection, bilinear sample, Huber weight, d(res)/d(idepth) —
// and the reference's running sums (energy per residual; Hdd and bd across ALL residuals and pixels, including the pixels
// that precede an out-of-bounds pixel of the same residual) are rebuilt in its exact order from the 56 lane values.
namespace sdso {
struct ActDev {
  int nf, w, h, n, minObs;
  float fx, fy, cx, cy;
  const float* pair_R; const float* pair_t; const float* pair_aff;
  const float4* const* img;
  const int* host; const float* u; const float* v; const float* idepth_min; const float* idepth_max;
  const float* color; const float* weights; const float* energyTH;
  int8_t* status; float* idepth_out; uint8_t* res_state;
};
constexpr float kMinIdepthH_act = 100.f;
constexpr int kGNItsOnPointActivation = 3;
}  // namespace sdso

__global__ __launch_bounds__(256) void k_activate_points(ActDev A) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int p = blockIdx.x * 4 + wv;
  if (p >= A.n) return;
  const int nf = A.nf, host = A.host[p];
  const int r = lane >> 3, idx = lane & 7;
  const int nres = nf - 1;
  const int tgt = r < host ? r : r + 1;                  // r-th frame != host, in frame order
  const bool live = r < nres;
  const int pairi = host * nf + (live ? tgt : (host == 0 ? 1 % nf : 0));
  float R[9], t[3];
#pragma unroll
  for (int k = 0; k < 9; k++) R[k] = A.pair_R[(size_t)pairi * 9 + k];
#pragma unroll
  for (int k = 0; k < 3; k++) t[k] = A.pair_t[(size_t)pairi * 3 + k];
  const float aff0 = A.pair_aff[(size_t)pairi * 2], aff1 = A.pair_aff[(size_t)pairi * 2 + 1];
  const float4* __restrict__ dIl = A.img[live ? tgt : host];
  const float fxl = A.fx, fyl = A.fy, cxl = A.cx, cyl = A.cy;
  const float fxli = 1.0f / fxl, fyli = 1.0f / fyl;
  const float wM3G = A.w - 3, hM3G = A.h - 3;
  const float pu = A.u[p], pv = A.v[p];
  const float color = A.color[(size_t)p * 8 + idx], wgt = A.weights[(size_t)p * 8 + idx];
  const float energyTH = A.energyTH[p];
  const int pdx = c_pat[idx][0], pdy = c_pat[idx][1];

  // per-lane terms of one linearizeResidual pass at `idepth`
  auto eval = [&](float idepth, bool& fail, float& e, float& hh, float& bb) {
    fail = false; e = 0; hh = 0; bb = 0;
    const float KliP0 = (pu + pdx - cxl) * fxli, KliP1 = (pv + pdy - cyl) * fyli, KliP2 = 1;
    float ptp[3];
#pragma unroll
    for (int k = 0; k < 3; k++) ptp[k] = ((R[k * 3 + 0] * KliP0 + R[k * 3 + 1] * KliP1) + R[k * 3 + 2] * KliP2) + t[k] * idepth;
    const float drescale = 1.0f / ptp[2];
    if (!(drescale > 0)) { fail = true; return; }
    const float u = ptp[0] * drescale, v = ptp[1] * drescale;
    const float Ku = u * fxl + cxl, Kv = v * fyl + cyl;
    if (!(Ku > 1.1f && Kv > 1.1f && Ku < wM3G && Kv < hM3G)) { fail = true; return; }
    const float3 hit = interp33(dIl, Ku, Kv, A.w);
    if (!isfinite(hit.x)) { fail = true; return; }
    const float residual = hit.x - (aff0 * color + aff1);
    float hw = fabsf(residual) < kHuberTH ? 1 : kHuberTH / fabsf(residual);
    e = wgt * wgt * hw * residual * residual * (2 - hw);
    const float dxInterp = hit.y * fxl, dyInterp = hit.z * fyl;
    const float d_idepth = (dxInterp * drescale * (t[0] - t[2] * u) + dyInterp * drescale * (t[1] - t[2] * v)) * SCALE_IDEPTH;
    hw *= wgt * wgt;
    hh = (hw * d_idepth) * d_idepth;
    bb = (hw * residual) * d_idepth;
  };

  int st[7], nst[7];
  float en[7], nen[7];
#pragma unroll
  for (int i = 0; i < 7; i++) { st[i] = 0; nst[i] = 2; en[i] = 0; nen[i] = 0; }

  // one pass over all residuals in the reference's order; every lane ends with the same (uniform) results
  auto pass = [&](float idepth, float slack, float& Hdd, float& bd) -> float {
    bool f; float e, hh, bb;
    if (live) eval(idepth, f, e, hh, bb); else { f = false; e = 0; hh = 0; bb = 0; }
    float total = 0;
#pragma unroll
    for (int i = 0; i < 7; i++) {
      if (i >= nres) continue;
      if (st[i] == 1) { nst[i] = 1; total += en[i]; continue; }          // ImmaturePoint.cpp:893-895
      float energyLeft = 0;
      bool alive = true;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int src = i * 8 + k;
        const bool fk = lane_bcast((int)f, src) != 0;
        const float ek = lane_bcast(e, src), hk = lane_bcast(hh, src), bk = lane_bcast(bb, src);
        if (alive && fk) alive = false;
        if (alive) { energyLeft += ek; Hdd += hk; bd += bk; }
      }
      if (!alive) { nst[i] = 1; total += en[i]; continue; }               // OOB: returns the old state_energy
      if (energyLeft > energyTH * slack) { energyLeft = energyTH * slack; nst[i] = 2; }
      else nst[i] = 0;
      nen[i] = energyLeft;
      total += energyLeft;
    }
    return total;
  };

  float lastHdd = 0, lastbd = 0;
  float currentIdepth = (A.idepth_max[p] + A.idepth_min[p]) * 0.5f;
  float lastEnergy = pass(currentIdepth, 1000.f, lastHdd, lastbd);
#pragma unroll
  for (int i = 0; i < 7; i++) { st[i] = nst[i]; en[i] = nen[i]; }
  int8_t status = 1;
  bool done = false;
  if (!isfinite(lastEnergy) || lastHdd < kMinIdepthH_act) { status = 0; done = true; }     // FullSystemOptPoint.cpp:104-110
  float lambda = 0.1f;
  if (!done) {
    for (int iteration = 0; iteration < kGNItsOnPointActivation; iteration++) {
      float H = lastHdd;
      H *= 1 + lambda;
      const float step = (1.0 / H) * lastbd;
      const float newIdepth = currentIdepth - step;
      float newHdd = 0, newbd = 0;
      const float newEnergy = pass(newIdepth, 1.f, newHdd, newbd);
      if (!isfinite(lastEnergy) || newHdd < kMinIdepthH_act) { status = 0; done = true; break; }   // :134-141
      if (newEnergy < lastEnergy) {
        currentIdepth = newIdepth; lastHdd = newHdd; lastbd = newbd; lastEnergy = newEnergy;
#pragma unroll
        for (int i = 0; i < 7; i++) { st[i] = nst[i]; en[i] = nen[i]; }
        lambda *= 0.5;
      } else lambda *= 5;
      if (fabsf(step) < 0.0001 * currentIdepth) break;
    }
  }
  if (lane == 0) {
    for (int f = 0; f < nf; f++) A.res_state[(size_t)p * nf + f] = 255;
    A.idepth_out[p] = currentIdepth;
    if (!done) {
      if (!isfinite(currentIdepth)) status = -1;
      else {
        int good = 0;
#pragma unroll
        for (int i = 0; i < 7; i++) if (i < nres) { good += st[i] == 0; A.res_state[(size_t)p * nf + (i < host ? i : i + 1)] = (uint8_t)st[i]; }
        if (good < A.minObs || !isfinite(energyTH)) status = -1;
      }
    }
    A.status[p] = status;
  }
}

extern "C" int sdso_activate_points_batch(sdso_ctx* ctx, const sdso_activate_t* A, int8_t* status, float* idepth_out, uint8_t* res_state) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  SDSO_REQUIRE(ctx, A && status && idepth_out && res_state, "null argument");
  const int nf = A->nf, n = A->n;
  SDSO_REQUIRE(ctx, nf >= 2 && nf <= 8 && n >= 0, "2 <= nf <= 8 keyframes");
  if (n == 0) return SDSO_OK;
  SDSO_REQUIRE(ctx, A->pair_R && A->pair_t && A->pair_aff && A->frame_slot && A->host && A->u && A->v && A->idepth_min && A->idepth_max && A->color &&
                        A->weights && A->energyTH, "null array");
  std::vector<const float4*> imgs(nf);
  for (int f = 0; f < nf; f++) {
    auto ip = ctx->pyr.find(A->frame_slot[f]);
    SDSO_REQUIRE(ctx, ip != ctx->pyr.end(), "unknown frame slot");
    SDSO_REQUIRE(ctx, ip->second.w[0] == A->w && ip->second.h[0] == A->h, "pyramid size differs from w/h");
    imgs[f] = ip->second.d[0];
  }
  for (int i = 0; i < n; i++) SDSO_REQUIRE(ctx, A->host[i] >= 0 && A->host[i] < nf, "host out of range");
  // one scratch blob: tables, images, point arrays, outputs
  const size_t fl = (size_t)nf * nf * 14 + (size_t)n * 21;
  const size_t bytes = sizeof(float) * fl + sizeof(void*) * nf + sizeof(int) * n + sizeof(float) * n + (size_t)n * (1 + nf) + 64;
  int rc = ensure_scratch(ctx, bytes);
  if (rc) return rc;
  float* f0 = (float*)ctx->scratch;
  ActDev D;
  D.nf = nf; D.w = A->w; D.h = A->h; D.n = n; D.minObs = A->minObs;
  D.fx = A->K[0]; D.fy = A->K[1]; D.cx = A->K[2]; D.cy = A->K[3];
  float* q = f0;
  auto upf = [&](const float* src, size_t cnt) { float* d = q; q += cnt; hipMemcpyAsync(d, src, sizeof(float) * cnt, hipMemcpyHostToDevice, ctx->stream); return (const float*)d; };
  D.pair_R = upf(A->pair_R, (size_t)nf * nf * 9); D.pair_t = upf(A->pair_t, (size_t)nf * nf * 3); D.pair_aff = upf(A->pair_aff, (size_t)nf * nf * 2);
  D.u = upf(A->u, n); D.v = upf(A->v, n); D.idepth_min = upf(A->idepth_min, n); D.idepth_max = upf(A->idepth_max, n);
  D.color = upf(A->color, (size_t)n * 8); D.weights = upf(A->weights, (size_t)n * 8); D.energyTH = upf(A->energyTH, n);
  D.idepth_out = q; q += n;
  const float4** d_img = (const float4**)(((uintptr_t)q + 15) & ~(uintptr_t)15);
  SDSO_HIP(ctx, hipMemcpyAsync(d_img, imgs.data(), sizeof(void*) * nf, hipMemcpyHostToDevice, ctx->stream));
  D.img = d_img;
  int* d_host = (int*)(d_img + nf);
  SDSO_HIP(ctx, hipMemcpyAsync(d_host, A->host, sizeof(int) * n, hipMemcpyHostToDevice, ctx->stream));
  D.host = d_host;
  D.status = (int8_t*)(d_host + n);
  D.res_state = (uint8_t*)(D.status + n);
  hipLaunchKernelGGL(k_activate_points, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, D);
  SDSO_HIP(ctx, hipGetLastError());
  SDSO_HIP(ctx, hipMemcpyAsync(status, D.status, n, hipMemcpyDeviceToHost, ctx->stream));
  SDSO_HIP(ctx, hipMemcpyAsync(res_state, D.res_state, (size_t)n * nf, hipMemcpyDeviceToHost, ctx->stream));
  SDSO_HIP(ctx, hipMemcpyAsync(idepth_out, D.idepth_out, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SDSO_OK;
}
