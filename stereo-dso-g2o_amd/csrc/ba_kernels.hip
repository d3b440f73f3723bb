// Windowed-BA kernels for gfx950.  Reference arithmetic (paths under /root/reference):
//   k_ba_linearize      PointFrameResidual::linearize              src/FullSystem/Residuals.cpp:83-336
//   k_ba_apply          PointFrameResidual::applyRes + takeDataF   Residuals.cpp:367-385, EnergyFunctionalStructs.cpp:37-51
//   k_ba_fixlin         EFResidual::fixLinearizationF              EnergyFunctionalStructs.cpp:96-123
//   k_ba_accum_top      AccumulatedTopHessianSSE::addPoint<mode>   src/OptimizationBackend/AccumulatedTopHessian.cpp:36-198
//   k_ba_sc             AccumulatedSCHessianSSE::addPoint          AccumulatedSCHessian.cpp:34-103
//   k_ba_stitch         stitchDoubleInternal (top and SC)          AccumulatedTopHessian.cpp:265-337, AccumulatedSCHessian.cpp:106-195
//   k_ba_solve          EnergyFunctional::solveSystemF             EnergyFunctional.cpp:838-995
//   k_ba_resub          EnergyFunctional::resubstituteFPt          EnergyFunctional.cpp:305-341
//   k_ba_step_points    FullSystem::doStepFromBackup (points)      src/FullSystem/FullSystemOptimize.cpp:260-276
#include "ba_kernels.h"

namespace sdso {

__constant__ int c_pattern[8][2] = {{0, -2}, {-1, -1}, {1, -1}, {-2, 0}, {0, 0}, {2, 0}, {-1, 1}, {0, 2}};

__device__ __forceinline__ double block_sum_d(double v, double* lds) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) lds[wv] = v;
  __syncthreads();
  double s = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); w++) s += lds[w];
  __syncthreads();
  return s;
}

// ------------------------------------------------------------------ linearize
__device__ double linearize_one(const BaDev& B, int i) {
  B.r_newEnergyWO[i] = -1.f;
  const uint8_t st = B.r_state[i];
  if (st == 1) { B.r_newState[i] = 1; return (double)B.r_energy[i]; }
  const int pt = B.r_point[i], h = B.r_host[i], t = B.r_target[i];
  const float* __restrict__ pre = B.t_precalc + (size_t)(h * B.nf + t) * 27;
  const float* KRKi = pre; const float* Kt = pre + 9; const float* R0 = pre + 12; const float* t0 = pre + 21;
  const float affLL0 = pre[24], affLL1 = pre[25], b0 = pre[26];
  const float4 g = B.p_geo[pt];
  const float pu = g.x, pv = g.y, idepth_scaled = g.z, idepth_zero_scaled = g.w;
  const float4* __restrict__ dIl = B.t_img[t];
  float* __restrict__ J = B.J[1 - B.r_jsel[i]];
  const int S = B.nrp;
  const float fxl = B.fxl, fyl = B.fyl, cxl = B.cxl, cyl = B.cyl, fxli = B.fxli, fyli = B.fyli;

  // projectPoint (ResidualProjections.h:64-96) at the FEJ point
  float KliP[3];
  KliP[0] = (pu + 0 - cxl) * fxli;
  KliP[1] = (pv + 0 - cyl) * fyli;
  KliP[2] = 1;
  float ptp[3];
#pragma unroll
  for (int r = 0; r < 3; r++) ptp[r] = ((R0[r * 3 + 0] * KliP[0] + R0[r * 3 + 1] * KliP[1]) + R0[r * 3 + 2] * KliP[2]) + t0[r] * idepth_zero_scaled;
  const float drescale = 1.0f / ptp[2];
  const float new_idepth = idepth_zero_scaled * drescale;
  if (!(drescale > 0)) { B.r_newState[i] = 1; return (double)B.r_energy[i]; }
  const float u = ptp[0] * drescale;
  const float v = ptp[1] * drescale;
  const float Ku0 = u * fxl + cxl;
  const float Kv0 = v * fyl + cyl;
  if (!(Ku0 > 1.1f && Kv0 > 1.1f && Ku0 < B.wM3 && Kv0 < B.hM3)) { B.r_newState[i] = 1; return (double)B.r_energy[i]; }
  if (B.r_proj) { float* pj = B.r_proj + (size_t)i * 19; pj[16] = Ku0; pj[17] = Kv0; pj[18] = new_idepth; }

  {  // Residuals.cpp:135-185
    float d_C_x[4], d_C_y[4];
    const float d_d_x = drescale * (t0[0] - t0[2] * u) * SCALE_IDEPTH * fxl;
    const float d_d_y = drescale * (t0[1] - t0[2] * v) * SCALE_IDEPTH * fyl;
    d_C_x[2] = drescale * (R0[6] * u - R0[0]);
    d_C_x[3] = fxl * drescale * (R0[7] * u - R0[1]) * fyli;
    d_C_x[0] = KliP[0] * d_C_x[2];
    d_C_x[1] = KliP[1] * d_C_x[3];
    d_C_y[2] = fyl * drescale * (R0[6] * v - R0[3]) * fxli;
    d_C_y[3] = drescale * (R0[7] * v - R0[4]);
    d_C_y[0] = KliP[0] * d_C_y[2];
    d_C_y[1] = KliP[1] * d_C_y[3];
    d_C_x[0] = (d_C_x[0] + u) * SCALE_F;
    d_C_x[1] *= SCALE_F;
    d_C_x[2] = (d_C_x[2] + 1) * SCALE_C;
    d_C_x[3] *= SCALE_C;
    d_C_y[0] *= SCALE_F;
    d_C_y[1] = (d_C_y[1] + v) * SCALE_F;
    d_C_y[2] *= SCALE_C;
    d_C_y[3] = (d_C_y[3] + 1) * SCALE_C;
    J[(J_XI0 + 0) * S + i] = new_idepth * fxl;
    J[(J_XI0 + 1) * S + i] = 0;
    J[(J_XI0 + 2) * S + i] = -new_idepth * u * fxl;
    J[(J_XI0 + 3) * S + i] = -u * v * fxl;
    J[(J_XI0 + 4) * S + i] = (1 + u * u) * fxl;
    J[(J_XI0 + 5) * S + i] = -v * fxl;
    J[(J_XI1 + 0) * S + i] = 0;
    J[(J_XI1 + 1) * S + i] = new_idepth * fyl;
    J[(J_XI1 + 2) * S + i] = -new_idepth * v * fyl;
    J[(J_XI1 + 3) * S + i] = -(1 + v * v) * fyl;
    J[(J_XI1 + 4) * S + i] = u * v * fyl;
    J[(J_XI1 + 5) * S + i] = u * fyl;
#pragma unroll
    for (int k = 0; k < 4; k++) { J[(J_C0 + k) * S + i] = d_C_x[k]; J[(J_C1 + k) * S + i] = d_C_y[k]; }
    J[(J_DD + 0) * S + i] = d_d_x;
    J[(J_DD + 1) * S + i] = d_d_y;
  }

  float JIdxJIdx_00 = 0, JIdxJIdx_11 = 0, JIdxJIdx_10 = 0;
  float JabJIdx_00 = 0, JabJIdx_01 = 0, JabJIdx_10 = 0, JabJIdx_11 = 0;
  float JabJab_00 = 0, JabJab_01 = 0, JabJab_11 = 0;
  float wJI2_sum = 0, energyLeft = 0;
  const float4 c0 = *(const float4*)(B.p_color + (size_t)pt * 8), c1 = *(const float4*)(B.p_color + (size_t)pt * 8 + 4);
  const float4 w0 = *(const float4*)(B.p_weights + (size_t)pt * 8), w1 = *(const float4*)(B.p_weights + (size_t)pt * 8 + 4);
  const float color[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
  const float weights[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
  bool oob = false;
#pragma unroll
  for (int idx = 0; idx < 8; idx++) {
    if (oob) break;
    const float up = pu + c_pattern[idx][0], vp = pv + c_pattern[idx][1];
    float q[3];
#pragma unroll
    for (int r = 0; r < 3; r++) q[r] = ((KRKi[r * 3 + 0] * up + KRKi[r * 3 + 1] * vp) + KRKi[r * 3 + 2]) + Kt[r] * idepth_scaled;
    const float Ku = q[0] / q[2];
    const float Kv = q[1] / q[2];
    if (!(Ku > 1.1f && Kv > 1.1f && Ku < B.wM3 && Kv < B.hM3)) { oob = true; break; }
    if (B.r_proj) { B.r_proj[(size_t)i * 19 + idx * 2] = Ku; B.r_proj[(size_t)i * 19 + idx * 2 + 1] = Kv; }
    float3 hit = interp33(dIl, Ku, Kv, B.w);
    const float residual = hit.x - (affLL0 * color[idx] + affLL1);
    const float drdA = (color[idx] - b0);
    if (!isfinite(hit.x)) { oob = true; break; }
    float wgt = sqrtf(kOutlierTHSumComponent / (kOutlierTHSumComponent + (hit.y * hit.y + hit.z * hit.z)));
    wgt = 0.5f * (wgt + weights[idx]);
    float hw = fabsf(residual) < kHuberTH ? 1 : kHuberTH / fabsf(residual);
    energyLeft += wgt * wgt * hw * residual * residual * (2 - hw);
    if (hw < 1) hw = sqrtf(hw);
    hw = hw * wgt;
    hit.y *= hw;
    hit.z *= hw;
    J[(J_RESF + idx) * S + i] = residual * hw;
    J[(J_IDX0 + idx) * S + i] = hit.y;
    J[(J_IDX1 + idx) * S + i] = hit.z;
    J[(J_AB0 + idx) * S + i] = B.affA_fixed ? 0.f : drdA * hw;
    J[(J_AB1 + idx) * S + i] = B.affB_fixed ? 0.f : hw;
    JIdxJIdx_00 += hit.y * hit.y;
    JIdxJIdx_11 += hit.z * hit.z;
    JIdxJIdx_10 += hit.y * hit.z;
    JabJIdx_00 += drdA * hw * hit.y;
    JabJIdx_01 += drdA * hw * hit.z;
    JabJIdx_10 += hw * hit.y;
    JabJIdx_11 += hw * hit.z;
    JabJab_00 += drdA * drdA * hw * hw;
    JabJab_01 += drdA * hw * hw;
    JabJab_11 += hw * hw;
    wJI2_sum += hw * hw * (hit.y * hit.y + hit.z * hit.z);
  }
  if (oob) { B.r_newState[i] = 1; return (double)B.r_energy[i]; }
  J[(J_IDX2 + 0) * S + i] = JIdxJIdx_00; J[(J_IDX2 + 1) * S + i] = JIdxJIdx_10;
  J[(J_IDX2 + 2) * S + i] = JIdxJIdx_10; J[(J_IDX2 + 3) * S + i] = JIdxJIdx_11;
  J[(J_ABIDX + 0) * S + i] = JabJIdx_00; J[(J_ABIDX + 1) * S + i] = JabJIdx_01;
  J[(J_ABIDX + 2) * S + i] = JabJIdx_10; J[(J_ABIDX + 3) * S + i] = JabJIdx_11;
  J[(J_AB2 + 0) * S + i] = JabJab_00; J[(J_AB2 + 1) * S + i] = JabJab_01;
  J[(J_AB2 + 2) * S + i] = JabJab_01; J[(J_AB2 + 3) * S + i] = JabJab_11;

  B.r_newEnergyWO[i] = energyLeft;
  const float th = fmaxf(B.t_frameTH[h], B.t_frameTH[t]);
  if (energyLeft > th || wJI2_sum < 2) { energyLeft = th; B.r_newState[i] = 2; }
  else B.r_newState[i] = 0;
  B.r_newEnergy[i] = energyLeft;
  return (double)energyLeft;
}

__global__ __launch_bounds__(BA_BLOCK) void k_ba_linearize(const BaDev* __restrict__ wins) {
  const BaDev& B = wins[blockIdx.y];
  if ((int)(blockIdx.x * BA_BLOCK) >= B.nr) return;
  __shared__ double lds[BA_BLOCK / 64];
  const int i = blockIdx.x * BA_BLOCK + threadIdx.x;
  double e = 0;
  if (i < B.nr && !B.r_lin[i]) e = linearize_one(B, i);
  e = block_sum_d(e, lds);
  if (threadIdx.x == 0) B.e_part[blockIdx.x] = e;
}

// ------------------------------------------------------------------ applyRes(true) + takeDataF
// only_points: when non-null, restrict to residuals of flagged points (flagPointsForRemoval path)
__global__ __launch_bounds__(BA_BLOCK) void k_ba_apply(const BaDev* __restrict__ wins) {
  const BaDev& B = wins[blockIdx.y];
  const int i = blockIdx.x * BA_BLOCK + threadIdx.x;
  if (i >= B.nr) return;
  float* rec = B.r_rec + (size_t)i * 16;
  if (B.r_lin[i]) return;
  const uint8_t st = B.r_state[i];
  if (st == 1) return;  // can never go back from OOB
  const uint8_t ns = B.r_newState[i];
  uint8_t act = 0;
  if (ns == 0) {
    act = 1;
    const uint8_t sel = B.r_jsel[i] ^ 1;
    B.r_jsel[i] = sel;
    const float* __restrict__ J = B.J[sel];
    const int S = B.nrp;
    const float jdd0 = J[(J_DD + 0) * S + i], jdd1 = J[(J_DD + 1) * S + i];
    const float a00 = J[(J_IDX2 + 0) * S + i], a01 = J[(J_IDX2 + 1) * S + i], a10 = J[(J_IDX2 + 2) * S + i], a11 = J[(J_IDX2 + 3) * S + i];
    const float v0 = a00 * jdd0 + a01 * jdd1;
    const float v1 = a10 * jdd0 + a11 * jdd1;
    float out[8];
#pragma unroll
    for (int k = 0; k < 6; k++) out[k] = J[(J_XI0 + k) * S + i] * v0 + J[(J_XI1 + k) * S + i] * v1;
    out[6] = J[(J_ABIDX + 0) * S + i] * jdd0 + J[(J_ABIDX + 1) * S + i] * jdd1;
    out[7] = J[(J_ABIDX + 2) * S + i] * jdd0 + J[(J_ABIDX + 3) * S + i] * jdd1;
    *(float4*)(rec) = make_float4(out[0], out[1], out[2], out[3]);
    *(float4*)(rec + 4) = make_float4(out[4], out[5], out[6], out[7]);
  }
  B.r_act[i] = act;
  rec[RR_FLAGS] = (float)act;
  B.r_state[i] = ns;
  B.r_energy[i] = B.r_newEnergy[i];
}

// fixLinearizationF for the active residuals of flagged points; sets isLinearized
__global__ __launch_bounds__(BA_BLOCK) void k_ba_fixlin(const BaDev* __restrict__ wins, const uint8_t* __restrict__ pflag) {
  const BaDev& B = wins[blockIdx.y];
  const int i = blockIdx.x * BA_BLOCK + threadIdx.x;
  if (i >= B.nr) return;
  const int pt = B.r_point[i];
  if (!pflag[pt] || !B.r_act[i]) return;
  const float* __restrict__ J = B.J[B.r_jsel[i]];
  const int S = B.nrp;
  const float* dp = B.t_adHTdelta + (size_t)(B.r_host[i] + B.nf * B.r_target[i]) * 8;
  const float* dc = B.t_cdelta;
  float sx = 0, sy = 0, cx = 0, cy = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) { sx += J[(J_XI0 + k) * S + i] * dp[k]; sy += J[(J_XI1 + k) * S + i] * dp[k]; }
#pragma unroll
  for (int k = 0; k < 4; k++) { cx += J[(J_C0 + k) * S + i] * dc[k]; cy += J[(J_C1 + k) * S + i] * dc[k]; }
  const float dd = B.p_delta[pt];
  const float dx = sx + cx + J[(J_DD + 0) * S + i] * dd;
  const float dy = sy + cy + J[(J_DD + 1) * S + i] * dd;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    float rtz = J[(J_RESF + k) * S + i];
    rtz = rtz - J[(J_IDX0 + k) * S + i] * dx;
    rtz = rtz - J[(J_IDX1 + k) * S + i] * dy;
    rtz = rtz - J[(J_AB0 + k) * S + i] * dp[6];
    rtz = rtz - J[(J_AB1 + k) * S + i] * dp[7];
    B.r_toZero[k * S + i] = rtz;
  }
  B.r_lin[i] = 1;
  B.r_rec[(size_t)i * 16 + RR_FLAGS] = 3.f;  // active | linearized
}

// resetOOB + isLinearized=false for the residuals of flagged points (FullSystem.cpp:1012-1016)
__global__ __launch_bounds__(BA_BLOCK) void k_ba_reset_flagged(const BaDev* __restrict__ wins, const uint8_t* __restrict__ pflag) {
  const BaDev& B = wins[blockIdx.y];
  const int i = blockIdx.x * BA_BLOCK + threadIdx.x;
  if (i >= B.nr) return;
  if (!pflag[B.r_point[i]]) { B.r_lin[i] |= 2; return; }  // bit1: temporarily excluded from linearize/apply
  B.r_energy[i] = 0; B.r_newEnergy[i] = 0; B.r_newState[i] = 2; B.r_state[i] = 0; B.r_lin[i] = 0;
}
__global__ __launch_bounds__(BA_BLOCK) void k_ba_unmask(const BaDev* __restrict__ wins) {
  const BaDev& B = wins[blockIdx.y];
  const int i = blockIdx.x * BA_BLOCK + threadIdx.x;
  if (i < B.nr) B.r_lin[i] &= 1;
}
// resetOOB for every non-linearized residual (FullSystemOptimize.cpp:886-892)
__global__ __launch_bounds__(BA_BLOCK) void k_ba_reset_all(const BaDev* __restrict__ wins) {
  const BaDev& B = wins[blockIdx.y];
  const int i = blockIdx.x * BA_BLOCK + threadIdx.x;
  if (i >= B.nr || B.r_lin[i]) return;
  B.r_energy[i] = 0; B.r_newEnergy[i] = 0; B.r_newState[i] = 2; B.r_state[i] = 0;
}

// ------------------------------------------------------------------ top accumulation
// One workgroup per chunk of <=256 residuals of ONE (host,target) pair.  mode: 0 active, 1 linearized, 2 marginalise.
__global__ __launch_bounds__(BA_BLOCK) void k_ba_accum_top(const BaDev* __restrict__ wins, int mode, const uint8_t* __restrict__ pflag) {
  const BaDev& B = wins[blockIdx.y];
  if ((int)blockIdx.x >= B.nchunks) return;
  const int4 ch = B.chunks[blockIdx.x];
  const int i = ch.y + threadIdx.x;
  const int S = B.nrp;
  bool on = (int)threadIdx.x < ch.z;
  if (on) {
    const uint8_t lin = B.r_lin[i] & 1, act = B.r_act[i];
    if (mode == 0) on = !lin && act;
    else if (mode == 1) on = lin && act;
    else on = act && pflag[B.r_point[i]];
  }
  __shared__ float red[BA_BLOCK / 64][92];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float x[10], y[10], a = 0, b = 0, c = 0;
  float TR00 = 0, TR10 = 0, TR01 = 0, TR11 = 0, TR02 = 0, TR12 = 0;
  float br[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 10; k++) { x[k] = 0; y[k] = 0; }
  if (on) {
    const float* __restrict__ J = B.J[B.r_jsel[i]];
    const int pt = B.r_point[i];
    float resApprox[8];
    if (mode == 0) {
#pragma unroll
      for (int k = 0; k < 8; k++) resApprox[k] = J[(J_RESF + k) * S + i];
    } else if (mode == 2) {
#pragma unroll
      for (int k = 0; k < 8; k++) resApprox[k] = B.r_toZero[k * S + i];
    } else {
      const float* dp = B.t_adHTdelta + (size_t)ch.x * 8;
      const float* dc = B.t_cdelta;
      float sx = 0, sy = 0, cx = 0, cy = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) { sx += J[(J_XI0 + k) * S + i] * dp[k]; sy += J[(J_XI1 + k) * S + i] * dp[k]; }
#pragma unroll
      for (int k = 0; k < 4; k++) { cx += J[(J_C0 + k) * S + i] * dc[k]; cy += J[(J_C1 + k) * S + i] * dc[k]; }
      const float dd = B.p_delta[pt];
      const float dx = sx + cx + J[(J_DD + 0) * S + i] * dd;
      const float dy = sy + cy + J[(J_DD + 1) * S + i] * dd;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        float rtz = B.r_toZero[k * S + i];
        rtz = rtz + J[(J_IDX0 + k) * S + i] * dx;
        rtz = rtz + J[(J_IDX1 + k) * S + i] * dy;
        rtz = rtz + J[(J_AB0 + k) * S + i] * dp[6];
        rtz = rtz + J[(J_AB1 + k) * S + i] * dp[7];
        resApprox[k] = rtz;
      }
    }
    float JI_r0 = 0, JI_r1 = 0, Jab_r0 = 0, Jab_r1 = 0, rr = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      JI_r0 += resApprox[k] * J[(J_IDX0 + k) * S + i];
      JI_r1 += resApprox[k] * J[(J_IDX1 + k) * S + i];
      Jab_r0 += resApprox[k] * J[(J_AB0 + k) * S + i];
      Jab_r1 += resApprox[k] * J[(J_AB1 + k) * S + i];
      rr += resApprox[k] * resApprox[k];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) { x[k] = J[(J_C0 + k) * S + i]; y[k] = J[(J_C1 + k) * S + i]; }
#pragma unroll
    for (int k = 0; k < 6; k++) { x[4 + k] = J[(J_XI0 + k) * S + i]; y[4 + k] = J[(J_XI1 + k) * S + i]; }
    a = J[(J_IDX2 + 0) * S + i]; b = J[(J_IDX2 + 1) * S + i]; c = J[(J_IDX2 + 3) * S + i];
    TR00 = J[(J_ABIDX + 0) * S + i]; TR10 = J[(J_ABIDX + 1) * S + i]; TR01 = J[(J_ABIDX + 2) * S + i]; TR11 = J[(J_ABIDX + 3) * S + i];
    TR02 = JI_r0; TR12 = JI_r1;
    br[0] = J[(J_AB2 + 0) * S + i]; br[1] = J[(J_AB2 + 1) * S + i]; br[2] = Jab_r0; br[3] = J[(J_AB2 + 3) * S + i]; br[4] = Jab_r1; br[5] = rr;
    // per-residual idepth terms (AccumulatedTopHessian.cpp:160-172) -> record
    const float jdd0 = J[(J_DD + 0) * S + i], jdd1 = J[(J_DD + 1) * S + i];
    const float a10 = J[(J_IDX2 + 2) * S + i];
    const float q0 = a * jdd0 + b * jdd1;
    const float q1 = a10 * jdd0 + c * jdd1;
    float* rec = B.r_rec + (size_t)i * 16;
    rec[RR_BD] = JI_r0 * jdd0 + JI_r1 * jdd1;
    rec[RR_HDD] = q0 * jdd0 + q1 * jdd1;
#pragma unroll
    for (int k = 0; k < 4; k++) rec[RR_HCD + k] = x[k] * q0 + y[k] * q1;
  }
  // 55 + 30 + 6 sums (AccumulatorApprox::update / updateTopRight / updateBotRight) + residual count
  int idx = 0;
#pragma unroll
  for (int r = 0; r < 10; r++) {
#pragma unroll
    for (int cc = r; cc < 10; cc++) {
      const float val = a * x[cc] * x[r] + c * y[cc] * y[r] + b * (x[cc] * y[r] + y[cc] * x[r]);
      const float s = wave_sum(val);
      if (lane == 0) red[wv][idx] = s;
      idx++;
    }
  }
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const float s0 = wave_sum(x[r] * TR00 + y[r] * TR10);
    const float s1 = wave_sum(x[r] * TR01 + y[r] * TR11);
    const float s2 = wave_sum(x[r] * TR02 + y[r] * TR12);
    if (lane == 0) { red[wv][55 + 3 * r] = s0; red[wv][55 + 3 * r + 1] = s1; red[wv][55 + 3 * r + 2] = s2; }
  }
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const float s = wave_sum(br[k]);
    if (lane == 0) red[wv][85 + k] = s;
  }
  {
    const float s = wave_sum(on ? 1.f : 0.f);
    if (lane == 0) red[wv][91] = s;
  }
  __syncthreads();
  if (threadIdx.x < 92) {
    float s = red[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < BA_BLOCK / 64; w++) s += red[w][threadIdx.x];
    B.top_part[(size_t)blockIdx.x * 92 + threadIdx.x] = s;
  }
}

// fold chunk partials per pair (fixed order) into the packed accumulator; grid.x = nf*nf
__global__ __launch_bounds__(128) void k_ba_fold_top(const BaDev* __restrict__ wins, int which /*0 = A, 1 = L*/) {
  const BaDev& B = wins[blockIdx.y];
  const int pair = blockIdx.x;
  if (pair >= B.nf * B.nf) return;
  const int cb = B.pair_chunk_beg[pair], ce = B.pair_chunk_beg[pair + 1];
  float* out = B.accum + (which ? acc_off_topL(B.nf) : acc_off_topA(B.nf)) + (size_t)pair * 91;
  if (threadIdx.x < 91) {
    float s = 0;
    for (int ck = cb; ck < ce; ck++) s += B.top_part[(size_t)ck * 92 + threadIdx.x];
    out[threadIdx.x] = s;
  }
  if (pair == 0 && threadIdx.x == 127) {
    float s = 0;
    for (int ck = 0; ck < B.nchunks; ck++) s += B.top_part[(size_t)ck * 92 + 91];
    B.accum[acc_off_nres(B.nf) + which] = s;
  }
}

__global__ __launch_bounds__(128) void k_ba_zero_topL(const BaDev* __restrict__ wins) {
  const BaDev& B = wins[blockIdx.y];
  const int pair = blockIdx.x;
  if (pair >= B.nf * B.nf) return;
  if (threadIdx.x < 91) B.accum[acc_off_topL(B.nf) + (size_t)pair * 91 + threadIdx.x] = 0.f;
  if (pair == 0 && threadIdx.x == 127) B.accum[acc_off_nres(B.nf) + 1] = 0.f;
}

// ------------------------------------------------------------------ per-point Schur accumulation
// One wave per item (= up to BA_SC_PTS consecutive points of ONE host).  Lane (a,c)=(lane>>3,lane&7)
// owns element (a,c) of every 8x8 D block; the item's nf x nf D tiles live in LDS.
// pflag: when non-null only flagged points are processed (marginalizePointsF) and shiftPriorToZero=false.
__global__ __launch_bounds__(BA_BLOCK) void k_ba_sc(const BaDev* __restrict__ wins, const uint8_t* __restrict__ pflag, int shiftPriorToZero, int margMode) {
  const BaDev& B = wins[blockIdx.y];
  extern __shared__ float lds_all[];
  const int nf = B.nf;
  const int per_wave = nf * nf * 64 + nf * 32 + nf * 8;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int item = blockIdx.x * (BA_BLOCK / 64) + wv;
  if (item >= B.nitems) return;
  volatile float* D = lds_all + (size_t)wv * per_wave;
  volatile float* E = D + nf * nf * 64;
  volatile float* EB = E + nf * 32;
  for (int k = lane; k < per_wave; k += 64) D[k] = 0.f;
  const int4 it = B.items[item];
  const int la = lane >> 3, lc = lane & 7;
  float hcc = 0.f, bcv = 0.f;  // lanes 0..15: Hcc[a][c] (a=lane>>2,c=lane&3); lanes 0..3: bc
  for (int p = it.y; p < it.z; p++) {
    const int cnt = B.p_rcnt[p], beg = B.p_rbeg[p];
    float* po = B.p_out + (size_t)p * 16;
    if (pflag && !pflag[p]) continue;
    // gather the point's residual records (uniform addresses -> broadcast loads)
    float Ja[8], Jc[8], Je[8];
    int tg[8];
    bool act[8];
    float Hdd_A = 0, bd_A = 0, Hdd_L = 0, bd_L = 0, HcdA[4] = {0, 0, 0, 0}, HcdL[4] = {0, 0, 0, 0};
    int ngood = 0;
#pragma unroll
    for (int s = 0; s < 8; s++) {
      act[s] = false; Ja[s] = 0; Jc[s] = 0; Je[s] = 0; tg[s] = 0;
      if (s < cnt) {
        const int ri = __builtin_amdgcn_readfirstlane(B.p_rlist[beg + s]);
        const float* rec = B.r_rec + (size_t)ri * 16;
        const int fl = (int)rec[RR_FLAGS];
        if (fl & 1) {
          act[s] = true;
          ngood++;
          tg[s] = B.r_target[ri];
          Ja[s] = rec[la];
          Jc[s] = rec[lc];
          Je[s] = rec[(lane >> 2) & 7];
          const bool lin = (fl & 2) != 0;
          // mode 0 sums non-linearized, mode 1/2 linearized residuals (AccumulatedTopHessian.cpp:54-71, 177-192)
          if (!lin && !margMode) {
            bd_A += rec[RR_BD]; Hdd_A += rec[RR_HDD];
#pragma unroll
            for (int k = 0; k < 4; k++) HcdA[k] += rec[RR_HCD + k];
          } else {
            bd_L += rec[RR_BD]; Hdd_L += rec[RR_HDD];
#pragma unroll
            for (int k = 0; k < 4; k++) HcdL[k] += rec[RR_HCD + k];
          }
        }
      }
    }
    float HdiF = 0, bdSumF = 0;
    float Hcd[4] = {0, 0, 0, 0};
    if (ngood > 0) {
      float H = Hdd_A + Hdd_L + B.p_prior[p];
      if (H < 1e-10) H = 1e-10;
      HdiF = 1.0 / H;
      bdSumF = bd_A + bd_L;
      if (shiftPriorToZero) bdSumF += B.p_prior[p] * B.p_delta[p];
#pragma unroll
      for (int k = 0; k < 4; k++) Hcd[k] = HcdA[k] + HcdL[k];
    }
    if (lane == 0) {
      po[PO_HDD_A] = Hdd_A; po[PO_BD_A] = bd_A; po[PO_HDD_L] = Hdd_L; po[PO_BD_L] = bd_L;
#pragma unroll
      for (int k = 0; k < 4; k++) { po[PO_HCD_A + k] = HcdA[k]; po[PO_HCD_L + k] = HcdL[k]; }
      po[PO_HDI] = HdiF; po[PO_BDSUM] = bdSumF;
    }
    if (ngood == 0) continue;
    if (lane < 16) hcc += (HdiF * Hcd[lane >> 2]) * Hcd[lane & 3];
    if (lane < 4) bcv += (bdSumF * HdiF) * Hcd[lane];
#pragma unroll
    for (int s1 = 0; s1 < 8; s1++) {
      if (!act[s1]) continue;
      const float wl = HdiF * Ja[s1];
      const int t1 = tg[s1];
#pragma unroll
      for (int s2 = 0; s2 < 8; s2++) {
        if (!act[s2]) continue;
        const int bin = t1 * nf + tg[s2];
        D[bin * 64 + lane] = D[bin * 64 + lane] + wl * Jc[s2];
      }
    }
    // accE: (HdiF*JpJdF[a]) * Hcd[c], a = lane>>2 (0..7), c = lane&3 ; accEB: (HdiF*bdSumF) * JpJdF[a], a = lane (0..7)
#pragma unroll
    for (int s1 = 0; s1 < 8; s1++) {
      if (!act[s1]) continue;
      const int t1 = tg[s1];
      if (lane < 32) E[t1 * 32 + lane] = E[t1 * 32 + lane] + (HdiF * Je[s1]) * Hcd[lane & 3];
      if (lane < 8) EB[t1 * 8 + lane] = EB[t1 * 8 + lane] + (HdiF * bdSumF) * Jc[s1];
    }
  }
  // flush this item's partial tiles
  float* out = B.sc_part + (size_t)item * sc_part_floats(nf);
  for (int k = lane; k < per_wave; k += 64) out[k] = D[k];
  if (lane < 16) out[per_wave + lane] = hcc;
  if (lane < 4) out[per_wave + 16 + lane] = bcv;
}

// Register-tile variant for nf known at compile time (nf <= 8): lane (a,c) keeps its element of all
// NF x NF D tiles in VGPRs; a point's residual vectors are first scattered to per-TARGET slots (a point
// has at most one residual per target), so the tile index is a compile-time constant and targets
// without a residual add an exact +0.  Same accumulation order per tile as the LDS variant.
template <int NF>
__global__ __launch_bounds__(BA_BLOCK) void k_ba_sc_reg(const BaDev* __restrict__ wins, const uint8_t* __restrict__ pflag, int shiftPriorToZero, int margMode) {
  const BaDev& B = wins[blockIdx.y];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int item = blockIdx.x * (BA_BLOCK / 64) + wv;
  if (item >= B.nitems) return;
  const int4 it = B.items[item];
  const int la = lane >> 3, lc = lane & 7, le = (lane >> 2) & 7;
  float D[NF][NF], E[NF], EB[NF];
#pragma unroll
  for (int i = 0; i < NF; i++) {
    E[i] = 0.f; EB[i] = 0.f;
#pragma unroll
    for (int j = 0; j < NF; j++) D[i][j] = 0.f;
  }
  float hcc = 0.f, bcv = 0.f;
  for (int p = it.y; p < it.z; p++) {
    if (pflag && !pflag[p]) continue;
    const int cnt = B.p_rcnt[p], beg = B.p_rbeg[p];
    float* po = B.p_out + (size_t)p * 16;
    float JaT[NF], JcT[NF], JeT[NF];
    bool has[NF];
#pragma unroll
    for (int t = 0; t < NF; t++) { JaT[t] = 0.f; JcT[t] = 0.f; JeT[t] = 0.f; has[t] = false; }
    float Hdd_A = 0, bd_A = 0, Hdd_L = 0, bd_L = 0, HcdA[4] = {0, 0, 0, 0}, HcdL[4] = {0, 0, 0, 0};
    int ngood = 0;
#pragma unroll
    for (int s = 0; s < 8; s++) {
      if (s < cnt) {
        const int ri = __builtin_amdgcn_readfirstlane(B.p_rlist[beg + s]);
        const float* rec = B.r_rec + (size_t)ri * 16;
        const int fl = (int)rec[RR_FLAGS];
        if (fl & 1) {
          ngood++;
          const int tg = __builtin_amdgcn_readfirstlane((int)B.r_target[ri]);
          const float ja = rec[la], jc = rec[lc], je = rec[le];
#pragma unroll
          for (int t = 0; t < NF; t++)
            if (t == tg) { JaT[t] = ja; JcT[t] = jc; JeT[t] = je; has[t] = true; }
          if (!(fl & 2) && !margMode) {
            bd_A += rec[RR_BD]; Hdd_A += rec[RR_HDD];
#pragma unroll
            for (int k = 0; k < 4; k++) HcdA[k] += rec[RR_HCD + k];
          } else {
            bd_L += rec[RR_BD]; Hdd_L += rec[RR_HDD];
#pragma unroll
            for (int k = 0; k < 4; k++) HcdL[k] += rec[RR_HCD + k];
          }
        }
      }
    }
    float HdiF = 0, bdSumF = 0;
    float Hcd[4] = {0, 0, 0, 0};
    if (ngood > 0) {
      float H = Hdd_A + Hdd_L + B.p_prior[p];
      if (H < 1e-10) H = 1e-10;
      HdiF = 1.0 / H;
      bdSumF = bd_A + bd_L;
      if (shiftPriorToZero) bdSumF += B.p_prior[p] * B.p_delta[p];
#pragma unroll
      for (int k = 0; k < 4; k++) Hcd[k] = HcdA[k] + HcdL[k];
    }
    if (lane == 0) {
      *(float4*)(po + 0) = make_float4(Hdd_A, bd_A, HcdA[0], HcdA[1]);
      *(float4*)(po + 4) = make_float4(HcdA[2], HcdA[3], Hdd_L, bd_L);
      *(float4*)(po + 8) = make_float4(HcdL[0], HcdL[1], HcdL[2], HcdL[3]);
      po[PO_HDI] = HdiF; po[PO_BDSUM] = bdSumF;
    }
    if (ngood == 0) continue;
    if (lane < 16) hcc += (HdiF * Hcd[lane >> 2]) * Hcd[lane & 3];
    if (lane < 4) bcv += (bdSumF * HdiF) * Hcd[lane];
    const float hb = HdiF * bdSumF;
    const float hc = Hcd[lane & 3];
#pragma unroll
    for (int t1 = 0; t1 < NF; t1++) {
      if (!has[t1]) continue;   // wave-uniform: targets without a residual contribute exact zeros
      const float wl = HdiF * JaT[t1];
#pragma unroll
      for (int t2 = 0; t2 < NF; t2++) D[t1][t2] = D[t1][t2] + wl * JcT[t2];
      E[t1] = E[t1] + (HdiF * JeT[t1]) * hc;
      EB[t1] = EB[t1] + hb * JcT[t1];
    }
  }
  float* out = B.sc_part + (size_t)item * sc_part_floats(NF);
#pragma unroll
  for (int t1 = 0; t1 < NF; t1++) {
#pragma unroll
    for (int t2 = 0; t2 < NF; t2++) out[(t1 * NF + t2) * 64 + lane] = D[t1][t2];
    if (lane < 32) out[NF * NF * 64 + t1 * 32 + lane] = E[t1];
    if (lane < 8) out[NF * NF * 64 + NF * 32 + t1 * 8 + lane] = EB[t1];
  }
  const int per_wave = NF * NF * 64 + NF * 32 + NF * 8;
  if (lane < 16) out[per_wave + lane] = hcc;
  if (lane < 4) out[per_wave + 16 + lane] = bcv;
}

}  // namespace sdso
