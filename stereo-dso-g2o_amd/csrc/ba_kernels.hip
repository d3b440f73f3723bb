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

// Workgroup barrier for exchanges that go through LDS only: waits for this wave's LDS traffic, not for its outstanding global
// stores (`__syncthreads()` drains vmcnt too — microseconds per barrier while a wave still has Jacobian records in flight).
template <bool LDS_ONLY>
__device__ __forceinline__ void wg_barrier() {
  if (LDS_ONLY) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else __syncthreads();
}
template <bool LDS_ONLY = false>
__device__ __forceinline__ double block_sum_d(double v, double* lds) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) lds[wv] = v;
  wg_barrier<LDS_ONLY>();
  double s = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); w++) s += lds[w];
  wg_barrier<LDS_ONLY>();
  return s;
}

// Device layout of one RawResidualJacobian: 19 float4 groups (76 floats; group g of residual i at J + j_off(S, i, g), ba_kernels.h:
// blocks of 64 residuals x 19 groups), ordered so that linearize can store every group as soon as it is known:
//   g0..g2 Jpdxi[0][0..5],Jpdxi[1][0..5] | g3 Jpdc[0] | g4 Jpdc[1] | g5 {Jpdd[0],Jpdd[1],-,-}
//   g6+k {resF[k], JIdx[0][k], JIdx[1][k], JabF[0][k]} k=0..7 | g14,g15 JabF[1][0..7] | g16 JIdx2 | g17 JabJIdx | g18 Jab2
__device__ __forceinline__ float4& JQ(float* J, int S, int i, int g) { return *(float4*)(J + j_off(S, i, g)); }
__device__ __forceinline__ float4 JQ(const float* J, int S, int i, int g) { return *(const float4*)(J + j_off(S, i, g)); }
__device__ __forceinline__ void load_J(const float* J, int S, int i, float* jl) {
#pragma unroll
  for (int g = 0; g < 19; g++) {
    const float4 q = *(const float4*)(J + j_off(S, i, g));
    jl[4 * g] = q.x; jl[4 * g + 1] = q.y; jl[4 * g + 2] = q.z; jl[4 * g + 3] = q.w;
  }
}
// ABI field index (include/sdso_abi.h order) -> index in the device layout
__host__ __device__ constexpr int jdev(int f) {
  return f < 8 ? 24 + 4 * f : f < 14 ? (f - 8) : f < 20 ? 6 + (f - 14) : f < 24 ? 12 + (f - 20) : f < 28 ? 16 + (f - 24) : f < 30 ? 20 + (f - 28)
       : f < 38 ? 24 + 4 * (f - 30) + 1 : f < 46 ? 24 + 4 * (f - 38) + 2 : f < 54 ? 24 + 4 * (f - 46) + 3 : f < 62 ? 56 + (f - 54)
       : f < 66 ? 64 + (f - 62) : f < 70 ? 68 + (f - 66) : 72 + (f - 70);
}
#define JV(f) jl[jdev(f)]
typedef float te_f4 __attribute__((ext_vector_type(4)));

// getInterpolatedElement33 on the 4x2-tiled level-0 image: the four taps of a sample and the samples of one 8-pixel pattern
// fall into fewer 128-B lines than in the row-major image (6.4 instead of 8.3 per residual on average), the arithmetic is
// the same expression as interp33
__device__ __forceinline__ float3 interp33_tiled(const float4* __restrict__ img, float x, float y, int T) {
  const int ix = (int)x;
  const int iy = (int)y;
  const float dx = x - ix;
  const float dy = y - iy;
  const float dxdy = dx * dy;
  const float4 p00 = img[tiled_index(ix, iy, T)], p10 = img[tiled_index(ix + 1, iy, T)], p01 = img[tiled_index(ix, iy + 1, T)],
               p11 = img[tiled_index(ix + 1, iy + 1, T)];
  const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
  float3 r;
  r.x = w11 * p11.x + w01 * p01.x + w10 * p10.x + w00 * p00.x;
  r.y = w11 * p11.y + w01 * p01.y + w10 * p10.y + w00 * p00.y;
  r.z = w11 * p11.z + w01 * p01.z + w10 * p10.z + w00 * p00.z;
  return r;
}

// ------------------------------------------------------------------ linearize
// STORE: write the RawResidualJacobian groups to HBM (PointFrameResidual::J);  KEEP 1: leave them in jl[76]
// (device layout) for the caller;  KEEP 2: leave only the geometric groups and the 2x2 blocks in jl and return the
// five per-pixel sums of AccumulatedTopHessianSSE::addPoint<0> (resApprox = resF: JI_r[2], Jab_r[2], rr; same
// k = 0..7 order as AccumulatedTopHessian.cpp:119-128) in rs[5] — 35 fewer live registers.  ns_out = state_NewState.
#define SETQ(g, a, b, c, d)                                                              \
  do {                                                                                   \
    const float4 _q = make_float4(a, b, c, d);                                           \
    if (STORE) __builtin_nontemporal_store((te_f4){_q.x, _q.y, _q.z, _q.w}, (te_f4*)&JQ(J, S, i, g)); /* streamed, not re-read this iteration */ \
    if (KEEP == 1 || (KEEP == 2 && ((g) < 6 || (g) > 15))) { jl[4 * (g)] = _q.x; jl[4 * (g) + 1] = _q.y; jl[4 * (g) + 2] = _q.z; jl[4 * (g) + 3] = _q.w; } \
  } while (0)
template <bool STORE, int KEEP, bool TILED>
__device__ __forceinline__ double linearize_one(const BaDev& B, int i, int h, int t, float* jl, int& ns_out, float* rs = nullptr, bool inplace = false /* fused kernel with BaDev::jfix */) {
  B.r_newEnergyWO[i] = -1.f;
  ns_out = 1;
  const uint8_t st = B.r_state[i];
  if (st == 1) { B.r_newState[i] = 1; return (double)B.r_energy[i]; }
  const int pt = B.r_point[i];
  const float* __restrict__ pre = B.t_precalc + (size_t)(h * B.nf + t) * 27;
  const float* KRKi = pre; const float* Kt = pre + 9; const float* R0 = pre + 12; const float* t0 = pre + 21;
  const float affLL0 = pre[24], affLL1 = pre[25], b0 = pre[26];
  const float4 g = B.p_geo[pt];
  const float pu = g.x, pv = g.y, idepth_scaled = g.z, idepth_zero_scaled = g.w;
  const float4* __restrict__ dIl = B.t_img[t];
  float* __restrict__ J = STORE ? ((B.r_jsel[i] != 0) != inplace ? B.J[0] : B.J[1]) : nullptr;
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
    SETQ(0, new_idepth * fxl, 0, -new_idepth * u * fxl, -u * v * fxl);                       // Jpdxi[0][0..3]
    SETQ(1, (1 + u * u) * fxl, -v * fxl, 0, new_idepth * fyl);                                // Jpdxi[0][4..5], Jpdxi[1][0..1]
    SETQ(2, -new_idepth * v * fyl, -(1 + v * v) * fyl, u * v * fyl, u * fyl);                 // Jpdxi[1][2..5]
    SETQ(3, d_C_x[0], d_C_x[1], d_C_x[2], d_C_x[3]);
    SETQ(4, d_C_y[0], d_C_y[1], d_C_y[2], d_C_y[3]);
    SETQ(5, d_d_x, d_d_y, 0.f, 0.f);
  }

  float JIdxJIdx_00 = 0, JIdxJIdx_11 = 0, JIdxJIdx_10 = 0;
  float JabJIdx_00 = 0, JabJIdx_01 = 0, JabJIdx_10 = 0, JabJIdx_11 = 0;
  float JabJab_00 = 0, JabJab_01 = 0, JabJab_11 = 0;
  float wJI2_sum = 0, energyLeft = 0;
  const float4 c0 = *(const float4*)(B.p_color + (size_t)pt * 8), c1 = *(const float4*)(B.p_color + (size_t)pt * 8 + 4);
  const float4 w0 = *(const float4*)(B.p_weights + (size_t)pt * 8), w1 = *(const float4*)(B.p_weights + (size_t)pt * 8 + 4);
  const float color[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
  const float weights[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
  // Pass 1 (no memory traffic): project the 8 pattern pixels; the residual is OOB as soon as one leaves
  // the image (Residuals.cpp:215-225).  Doing this first removes the early exit from the sampling loop, so
  // the 32 bilinear taps below are independent loads the hardware can keep in flight together.
  float Kus[8], Kvs[8];
  bool oob = false;
#pragma unroll
  for (int idx = 0; idx < 8; idx++) {
    const float up = pu + c_pattern[idx][0], vp = pv + c_pattern[idx][1];
    float q[3];
#pragma unroll
    for (int r = 0; r < 3; r++) q[r] = ((KRKi[r * 3 + 0] * up + KRKi[r * 3 + 1] * vp) + KRKi[r * 3 + 2]) + Kt[r] * idepth_scaled;
    Kus[idx] = q[0] / q[2];
    Kvs[idx] = q[1] / q[2];
    if (!(Kus[idx] > 1.1f && Kvs[idx] > 1.1f && Kus[idx] < B.wM3 && Kvs[idx] < B.hM3)) oob = true;
  }
  if (oob) { B.r_newState[i] = 1; return (double)B.r_energy[i]; }
  float jab1[8];
  // the taps are fetched in two groups of 4 pattern pixels (16 gathers in flight per lane): half the registers
  // of one group of 32
#pragma unroll
  for (int hb = 0; hb < 8; hb += 4) {
  float3 hits[4];
#pragma unroll
  for (int k = 0; k < 4; k++) hits[k] = TILED ? interp33_tiled(dIl, Kus[hb + k], Kvs[hb + k], B.tiledT) : interp33(dIl, Kus[hb + k], Kvs[hb + k], B.w);
#pragma unroll
  for (int idx = hb; idx < hb + 4; idx++) {
    const float Ku = Kus[idx], Kv = Kvs[idx];
    if (B.r_proj) { B.r_proj[(size_t)i * 19 + idx * 2] = Ku; B.r_proj[(size_t)i * 19 + idx * 2 + 1] = Kv; }
    float3 hit = hits[idx - hb];
    if (!isfinite(hit.x)) oob = true;
    const float residual = hit.x - (affLL0 * color[idx] + affLL1);
    const float drdA = (color[idx] - b0);
    float wgt = sqrtf(kOutlierTHSumComponent / (kOutlierTHSumComponent + (hit.y * hit.y + hit.z * hit.z)));
    wgt = 0.5f * (wgt + weights[idx]);
    float hw = fabsf(residual) < kHuberTH ? 1 : kHuberTH / fabsf(residual);
    energyLeft += wgt * wgt * hw * residual * residual * (2 - hw);
    if (hw < 1) hw = sqrtf(hw);
    hw = hw * wgt;
    hit.y *= hw;
    hit.z *= hw;
    SETQ(6 + idx, residual * hw, hit.y, hit.z, B.affA_fixed ? 0.f : drdA * hw);   // resF, JIdx[0], JIdx[1], JabF[0]
    jab1[idx] = B.affB_fixed ? 0.f : hw;
    if (KEEP == 2) {
      const float ra = residual * hw;
      rs[0] += ra * hit.y; rs[1] += ra * hit.z;
      rs[2] += ra * (B.affA_fixed ? 0.f : drdA * hw); rs[3] += ra * jab1[idx];
      rs[4] += ra * ra;
    }
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
  __builtin_amdgcn_sched_barrier(0);   // keep the second group's gathers behind the first group's arithmetic
  }
  if (oob) { B.r_newState[i] = 1; return (double)B.r_energy[i]; }
  SETQ(14, jab1[0], jab1[1], jab1[2], jab1[3]);
  SETQ(15, jab1[4], jab1[5], jab1[6], jab1[7]);
  SETQ(16, JIdxJIdx_00, JIdxJIdx_10, JIdxJIdx_10, JIdxJIdx_11);
  SETQ(17, JabJIdx_00, JabJIdx_01, JabJIdx_10, JabJIdx_11);
  SETQ(18, JabJab_00, JabJab_01, JabJab_01, JabJab_11);

  B.r_newEnergyWO[i] = energyLeft;
  const float th = fmaxf(B.t_frameTH[h], B.t_frameTH[t]);
  if (energyLeft > th || wJI2_sum < 2) { energyLeft = th; ns_out = 2; }
  else ns_out = 0;
  B.r_newState[i] = (uint8_t)ns_out;
  B.r_newEnergy[i] = energyLeft;
  return (double)energyLeft;
}
// The same function for the fused kernel, with the 32 bilinear taps of every residual fetched COOPERATIVELY: the taps of one
// residual sit on 32 neighbouring lanes of ONE load instruction (pixel-major, the four corners of a pixel adjacent), so an
// instruction touches ~16 distinct 128-B lines instead of 64 and no line is asked for twice.  On gfx950 a divergent 16-byte load costs per distinct line, not
// per byte: with one residual per lane the 32 taps run at 105 G lane-taps/s whatever HBM could deliver (tools/mix_bw.hip);
// spread over lanes the same taps run at the HBM rate.  The interpolated samples travel through a wave-private LDS stage back
// to the residual's own lane, which then does exactly the arithmetic of linearize_one — every per-residual result stays
// bit-identical.  Must be called by ALL lanes of the wave
// (`live` = this lane has a residual to linearise); early exits of linearize_one become the `dead` flag.
// Stage of one wave: the 8 pattern-pixel coordinates of its 64 residuals (float2 each), and the interpolated {I, dx, dy}
// of every pattern pixel in rows of 65 floats ([pixel*3 + channel][residual]).
constexpr int CG_COORD_FLOATS = 64 * 8 * 2, CG_ROW = 64, CG_HIT_FLOATS = 8 * 3 * CG_ROW, CG_WAVE_FLOATS = CG_COORD_FLOATS + CG_HIT_FLOATS;
constexpr int CG_BATCH = 8;                                   // loads in flight per lane
__device__ __forceinline__ void cg_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// value of lane (quad base + k) of the caller's quad (v_mov_b32 with DPP quad_perm broadcast)
template <int K>
__device__ __forceinline__ float quad_bcast(float v) {
  return __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), K * 0x55, 0xf, 0xf, true));
}
// All 32 bilinear taps of a residual in ONE load instruction: round r serves residuals 2r and 2r+1 of the wave, lane = (residual,
// pattern pixel, corner).  The four corner lanes of a pixel exchange their samples inside the quad and evaluate
// getInterpolatedElement33 (same expression, same order: bit-identical to interp33); the corner-0 lane parks the result for the
// residual's own lane.  A residual's image lines are touched by exactly one instruction, so they are fetched once.
template <bool TILED>
__device__ __forceinline__ void coop_gather_hits(const float4* __restrict__ img, int Tw, bool dead, const float* Ku, const float* Kv, float* wstage) {
  const int lane = threadIdx.x & 63;
  float2* coords = (float2*)wstage;
  float* hits = wstage + CG_COORD_FLOATS;
#pragma unroll
  for (int k = 0; k < 8; k++) coords[lane * 8 + k] = dead ? make_float2(-1.f, -1.f) : make_float2(Ku[k], Kv[k]);
  cg_wave_sync();
  struct __attribute__((packed, aligned(4))) px3 { float x, y, z; };
  const int sub = lane >> 5, px = (lane >> 2) & 7, c = lane & 3;
#pragma unroll
  for (int r0 = 0; r0 < 32; r0 += CG_BATCH) {
    px3 q[CG_BATCH];
    float2 cx[CG_BATCH];
#pragma unroll
    for (int r = 0; r < CG_BATCH; r++) {
      const int unit = (r0 + r) * 2 + sub;
      cx[r] = coords[unit * 8 + px];
      q[r].x = 0; q[r].y = 0; q[r].z = 0;
      if (cx[r].x >= 0) {
        const int x = (int)cx[r].x + (c & 1), y = (int)cx[r].y + (c >> 1);
        q[r] = *(const px3*)(img + (TILED ? tiled_index(x, y, Tw) : x + y * Tw));
      }
    }
#pragma unroll
    for (int r = 0; r < CG_BATCH; r++) {
      const int unit = (r0 + r) * 2 + sub;
      const float x = cx[r].x, y = cx[r].y;
      const int ix = (int)x;
      const int iy = (int)y;
      const float dx = x - ix;
      const float dy = y - iy;
      const float dxdy = dx * dy;
      // corners: lane c = 0 -> p00 (x, y), 1 -> p10 (x+1, y), 2 -> p01 (x, y+1), 3 -> p11.  Every corner lane scales its own
      // sample by its own weight; lane 0 adds the four products in the order of interp33: ((w11 p11 + w01 p01) + w10 p10) + w00 p00
      const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
      const float wa = (c & 1) ? w11 : w01, wb = (c & 1) ? w10 : w00;
      const float wc = (c & 2) ? wa : wb;
      const float tx = wc * q[r].x, ty = wc * q[r].y, tz = wc * q[r].z;
      const float hx = ((quad_bcast<3>(tx) + quad_bcast<2>(tx)) + quad_bcast<1>(tx)) + tx;
      const float hy = ((quad_bcast<3>(ty) + quad_bcast<2>(ty)) + quad_bcast<1>(ty)) + ty;
      const float hz = ((quad_bcast<3>(tz) + quad_bcast<2>(tz)) + quad_bcast<1>(tz)) + tz;
      if (c == 0) {
        hits[(px * 3 + 0) * CG_ROW + unit] = hx;
        hits[(px * 3 + 1) * CG_ROW + unit] = hy;
        hits[(px * 3 + 2) * CG_ROW + unit] = hz;
      }
    }
  }
  cg_wave_sync();
}
// what a persistent workgroup requested for this chunk while it worked on the previous one (SDSO_LIN_PERSIST): r_state, r_point, p_geo
struct LinPre { int st, pt; float4 geo; };
template <bool STORE, int KEEP, bool TILED>
__device__ __forceinline__ double linearize_coop(const BaDev& B, int i, bool live, int h, int t, float* jl, int& ns_out, float* rs, float* wstage, const LinPre* have = nullptr) {
  ns_out = 1;
  bool dead = !live;
  double ret = 0;
  float Kus[8], Kvs[8], color[8], weights[8], jab1[8];
  bool oob = false;
  float g_u = 0, g_v = 0, g_dr = 0, g_nid = 0, g_k0 = 0, g_k1 = 0;
  float JIdxJIdx_00 = 0, JIdxJIdx_11 = 0, JIdxJIdx_10 = 0;
  float JabJIdx_00 = 0, JabJIdx_01 = 0, JabJIdx_10 = 0, JabJIdx_11 = 0;
  float JabJab_00 = 0, JabJab_01 = 0, JabJab_11 = 0;
  float wJI2_sum = 0, energyLeft = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) { Kus[k] = 0; Kvs[k] = 0; color[k] = 0; weights[k] = 0; jab1[k] = 0; }
  const float* __restrict__ pre = B.t_precalc + (size_t)(h * B.nf + t) * 27;
  const float affLL0 = pre[24], affLL1 = pre[25], b0 = pre[26];
  const float4* __restrict__ dIl = B.t_img[t];
  const int S = B.nrp;
  float* __restrict__ J = nullptr;
  if (!dead) do {
  B.r_newEnergyWO[i] = -1.f;
  const uint8_t st = have ? (uint8_t)have->st : B.r_state[i];
  if (st == 1) { B.r_newState[i] = 1; ret = (double)B.r_energy[i]; dead = true; break; }
  const int pt = have ? have->pt : B.r_point[i];
  const float* KRKi = pre; const float* Kt = pre + 9; const float* R0 = pre + 12; const float* t0 = pre + 21;
#if defined(SDSO_LIN_PERSIST) && SDSO_LIN_PERSIST < 2
  const float4 g = B.p_geo[pt];                  // (light form: only the chunk descriptor, r_state and r_point are requested ahead)
#else
  const float4 g = have ? have->geo : B.p_geo[pt];
#endif
  const float pu = g.x, pv = g.y, idepth_scaled = g.z, idepth_zero_scaled = g.w;
  J = STORE ? ((B.r_jsel[i] != 0) != (B.jfix != 0) ? B.J[0] : B.J[1]) : nullptr;   // jfix: EFResidual::J refreshed in place (ba_kernels.h)
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
  if (!(drescale > 0)) { B.r_newState[i] = 1; ret = (double)B.r_energy[i]; dead = true; break; }
  const float u = ptp[0] * drescale;
  const float v = ptp[1] * drescale;
  const float Ku0 = u * fxl + cxl;
  const float Kv0 = v * fyl + cyl;
  if (!(Ku0 > 1.1f && Kv0 > 1.1f && Ku0 < B.wM3 && Kv0 < B.hM3)) { B.r_newState[i] = 1; ret = (double)B.r_energy[i]; dead = true; break; }
  g_u = u; g_v = v; g_dr = drescale; g_nid = new_idepth; g_k0 = KliP[0]; g_k1 = KliP[1];   // the geometric Jacobians are built after the taps (fewer live registers during the gathers)

  const float4 c0 = *(const float4*)(B.p_color + (size_t)pt * 8), c1 = *(const float4*)(B.p_color + (size_t)pt * 8 + 4);
  const float4 w0 = *(const float4*)(B.p_weights + (size_t)pt * 8), w1 = *(const float4*)(B.p_weights + (size_t)pt * 8 + 4);
  color[0] = c0.x; color[1] = c0.y; color[2] = c0.z; color[3] = c0.w; color[4] = c1.x; color[5] = c1.y; color[6] = c1.z; color[7] = c1.w;
  weights[0] = w0.x; weights[1] = w0.y; weights[2] = w0.z; weights[3] = w0.w; weights[4] = w1.x; weights[5] = w1.y; weights[6] = w1.z; weights[7] = w1.w;
  // Pass 1 (no memory traffic): project the 8 pattern pixels; the residual is OOB as soon as one leaves
  // the image (Residuals.cpp:215-225).  Doing this first removes the early exit from the sampling loop, so
  // the 32 bilinear taps below are independent loads the hardware can keep in flight together.
#pragma unroll
  for (int idx = 0; idx < 8; idx++) {
    const float up = pu + c_pattern[idx][0], vp = pv + c_pattern[idx][1];
    float q[3];
#pragma unroll
    for (int r = 0; r < 3; r++) q[r] = ((KRKi[r * 3 + 0] * up + KRKi[r * 3 + 1] * vp) + KRKi[r * 3 + 2]) + Kt[r] * idepth_scaled;
    Kus[idx] = q[0] / q[2];
    Kvs[idx] = q[1] / q[2];
    if (!(Kus[idx] > 1.1f && Kvs[idx] > 1.1f && Kus[idx] < B.wM3 && Kvs[idx] < B.hM3)) oob = true;
  }
  if (oob) { B.r_newState[i] = 1; ret = (double)B.r_energy[i]; dead = true; break; }
  } while (0);
  coop_gather_hits<TILED>(dIl, TILED ? B.tiledT : B.w, dead, Kus, Kvs, wstage);
  {
  if (!dead) {
  const float* hstage = wstage + CG_COORD_FLOATS + (threadIdx.x & 63);
#pragma unroll
  for (int idx = 0; idx < 8; idx++) {
    float3 hit = make_float3(hstage[(idx * 3 + 0) * CG_ROW], hstage[(idx * 3 + 1) * CG_ROW], hstage[(idx * 3 + 2) * CG_ROW]);
    if (!isfinite(hit.x)) oob = true;
    const float residual = hit.x - (affLL0 * color[idx] + affLL1);
    const float drdA = (color[idx] - b0);
    float wgt = sqrtf(kOutlierTHSumComponent / (kOutlierTHSumComponent + (hit.y * hit.y + hit.z * hit.z)));
    wgt = 0.5f * (wgt + weights[idx]);
    float hw = fabsf(residual) < kHuberTH ? 1 : kHuberTH / fabsf(residual);
    energyLeft += wgt * wgt * hw * residual * residual * (2 - hw);
    if (hw < 1) hw = sqrtf(hw);
    hw = hw * wgt;
#ifdef SDSO_LIN_PK
    // A/B (round 6, verdict item 5b): the (dx, dy) pairs of the per-pixel sums as two-float vectors -> v_pk_mul_f32 / v_pk_add_f32.  Every product
    // and every sum is the scalar statement's, in its order (contraction is off: no packed fma): bit-identical.  profiles/r06_lin_pk_ab.txt
    typedef float pk2 __attribute__((ext_vector_type(2)));
    pk2 hyz = {hit.y, hit.z};
    hyz = hyz * hw;
    hit.y = hyz.x; hit.z = hyz.y;
    SETQ(6 + idx, residual * hw, hit.y, hit.z, B.affA_fixed ? 0.f : drdA * hw);   // resF, JIdx[0], JIdx[1], JabF[0]
    jab1[idx] = B.affB_fixed ? 0.f : hw;
    if (KEEP == 2) {
      const float ra = residual * hw;
      const pk2 r01 = (pk2){rs[0], rs[1]} + ra * hyz;
      rs[0] = r01.x; rs[1] = r01.y;
      rs[2] += ra * (B.affA_fixed ? 0.f : drdA * hw); rs[3] += ra * jab1[idx];
      rs[4] += ra * ra;
    }
    const pk2 sq = hyz * hyz;
    const pk2 jj = (pk2){JIdxJIdx_00, JIdxJIdx_11} + sq;
    JIdxJIdx_00 = jj.x; JIdxJIdx_11 = jj.y;
    JIdxJIdx_10 += hit.y * hit.z;
    const pk2 a0 = (pk2){JabJIdx_00, JabJIdx_01} + (drdA * hw) * hyz;
    JabJIdx_00 = a0.x; JabJIdx_01 = a0.y;
    const pk2 a1 = (pk2){JabJIdx_10, JabJIdx_11} + hw * hyz;
    JabJIdx_10 = a1.x; JabJIdx_11 = a1.y;
    JabJab_00 += drdA * drdA * hw * hw;
    JabJab_01 += drdA * hw * hw;
    JabJab_11 += hw * hw;
    wJI2_sum += hw * hw * (sq.x + sq.y);
#else
    hit.y *= hw;
    hit.z *= hw;
    SETQ(6 + idx, residual * hw, hit.y, hit.z, B.affA_fixed ? 0.f : drdA * hw);   // resF, JIdx[0], JIdx[1], JabF[0]
    jab1[idx] = B.affB_fixed ? 0.f : hw;
    if (KEEP == 2) {
      const float ra = residual * hw;
      rs[0] += ra * hit.y; rs[1] += ra * hit.z;
      rs[2] += ra * (B.affA_fixed ? 0.f : drdA * hw); rs[3] += ra * jab1[idx];
      rs[4] += ra * ra;
    }
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
#endif
  }
  }
  }
  if (dead) return ret;
  if (oob) { B.r_newState[i] = 1; return (double)B.r_energy[i]; }
  {  // Residuals.cpp:135-185
    const float* R0 = pre + 12; const float* t0 = pre + 21;
    const float u = g_u, v = g_v, drescale = g_dr, new_idepth = g_nid;
    const float KliP[2] = {g_k0, g_k1};
    const float fxl = B.fxl, fyl = B.fyl, fxli = B.fxli, fyli = B.fyli;
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
    SETQ(0, new_idepth * fxl, 0, -new_idepth * u * fxl, -u * v * fxl);                       // Jpdxi[0][0..3]
    SETQ(1, (1 + u * u) * fxl, -v * fxl, 0, new_idepth * fyl);                                // Jpdxi[0][4..5], Jpdxi[1][0..1]
    SETQ(2, -new_idepth * v * fyl, -(1 + v * v) * fyl, u * v * fyl, u * fyl);                 // Jpdxi[1][2..5]
    SETQ(3, d_C_x[0], d_C_x[1], d_C_x[2], d_C_x[3]);
    SETQ(4, d_C_y[0], d_C_y[1], d_C_y[2], d_C_y[3]);
    SETQ(5, d_d_x, d_d_y, 0.f, 0.f);
  }
  SETQ(14, jab1[0], jab1[1], jab1[2], jab1[3]);
  SETQ(15, jab1[4], jab1[5], jab1[6], jab1[7]);
  SETQ(16, JIdxJIdx_00, JIdxJIdx_10, JIdxJIdx_10, JIdxJIdx_11);
  SETQ(17, JabJIdx_00, JabJIdx_01, JabJIdx_10, JabJIdx_11);
  SETQ(18, JabJab_00, JabJab_01, JabJab_01, JabJab_11);

  B.r_newEnergyWO[i] = energyLeft;
  const float th = fmaxf(B.t_frameTH[h], B.t_frameTH[t]);
  if (energyLeft > th || wJI2_sum < 2) { energyLeft = th; ns_out = 2; }
  else ns_out = 0;
  B.r_newState[i] = (uint8_t)ns_out;
  B.r_newEnergy[i] = energyLeft;
  return (double)energyLeft;
}
#undef SETQ

template <bool TILED>
__global__ __launch_bounds__(BA_BLOCK) void k_ba_linearize(const BaDev* __restrict__ wins, int cond = 0) {
  const BaDev& B = wins[blockIdx.y];
  if (ba_finished_lin(B) || ba_gate_skip(B, cond)) return;   // (a window whose resident loop has ended is left alone)
  if ((int)(blockIdx.x * BA_BLOCK) >= B.nr) return;
  __shared__ double lds[BA_BLOCK / 64];
  const int i = blockIdx.x * BA_BLOCK + threadIdx.x;
  double e = 0;
  int ns;
  if (i < B.nr && !B.r_lin[i]) e = linearize_one<true, 0, TILED>(B, i, B.r_host[i], B.r_target[i], nullptr, ns);
  e = block_sum_d(e, lds);
  if (threadIdx.x == 0) B.e_part[blockIdx.x] = e;
}

// ------------------------------------------------------------------ applyRes(true) + takeDataF
// only_points: when non-null, restrict to residuals of flagged points (flagPointsForRemoval path)
__global__ __launch_bounds__(BA_BLOCK) void k_ba_apply(const BaDev* __restrict__ wins, int cond = 0) {
  const BaDev& B = wins[blockIdx.y];
  if (ba_gate_skip(B, cond)) return;
  const int i = blockIdx.x * BA_BLOCK + threadIdx.x;
  if (i >= B.nr) return;
  float* jp = B.r_rec + (size_t)B.r_orig[i] * 16;
  float* tm = jp + 8;
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
    const float4 g0 = JQ(J, S, i, 0), g1 = JQ(J, S, i, 1), g2 = JQ(J, S, i, 2), gd = JQ(J, S, i, 5), i2 = JQ(J, S, i, 16), ai = JQ(J, S, i, 17);
    const float jdd0 = gd.x, jdd1 = gd.y;
    const float v0 = i2.x * jdd0 + i2.y * jdd1;
    const float v1 = i2.z * jdd0 + i2.w * jdd1;
    const float xi0[6] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y}, xi1[6] = {g1.z, g1.w, g2.x, g2.y, g2.z, g2.w};
    float out[8];
#pragma unroll
    for (int k = 0; k < 6; k++) out[k] = xi0[k] * v0 + xi1[k] * v1;
    out[6] = ai.x * jdd0 + ai.y * jdd1;
    out[7] = ai.z * jdd0 + ai.w * jdd1;
    *(float4*)(jp) = make_float4(out[0], out[1], out[2], out[3]);
    *(float4*)(jp + 4) = make_float4(out[4], out[5], out[6], out[7]);
  }
  else {
    // the records of a residual that is not active hold zeros (like the fused kernel's): the Schur kernel adds / multiplies records without
    // looking at their flags
    *(float4*)(jp) = make_float4(0.f, 0.f, 0.f, 0.f);
    *(float4*)(jp + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    *(float4*)(tm) = make_float4(0.f, 0.f, 0.f, 0.f);
    tm[4] = 0.f; tm[5] = 0.f;
  }
  B.r_act[i] = act;
  jp[RR_FLAGS] = (float)act;
  B.r_state[i] = ns;
  B.r_energy[i] = B.r_newEnergy[i];
}

// What FullSystem::linearizeAll_Reductor(fixLinearization = true) does per residual after applyRes (FullSystemOptimize.cpp:62-78) at the end
// of FullSystem::optimize, and the projections that closing linearisation left in the PointFrameResidual (centerProjectedTo, projectedTo:
// Residuals.cpp:130-131, :215-225 — the expressions of linearize_one, re-evaluated here so that the hot kernels need not store them).
// One lane per residual; PointHessian::maxRelBaseline / numGoodResiduals through integer atomics (order-independent: non-negative floats
// compare like their bit patterns).  proj: nr x 19 (projectedTo 16, centerProjectedTo 3), zeros for residuals that are not active.
__global__ __launch_bounds__(BA_BLOCK) void k_ba_post_state(const BaDev* __restrict__ wins, float* __restrict__ proj, int counters /* 0: the projections only */) {
  const BaDev& B = wins[blockIdx.y];
  const int i = blockIdx.x * BA_BLOCK + threadIdx.x;
  if (i >= B.nr) return;
  if (!proj && !counters) return;
  float scratch[19];
  float* pj = proj ? proj + (size_t)i * 19 : scratch;   // proj == nullptr: the counters only (the end of an optimize call)
#pragma unroll
  for (int k = 0; k < 19; k++) pj[k] = 0.f;
  if ((B.r_lin[i] & 1) || !B.r_act[i]) return;       // not in activeResiduals (:880-889) / on toRemove (:80-84)
  const int h = B.r_host[i], t = B.r_target[i], pt = B.r_point[i];
  const float* __restrict__ pre = B.t_precalc + (size_t)(h * B.nf + t) * 27;
  const float* KRKi = pre; const float* Kt = pre + 9; const float* R0 = pre + 12; const float* t0 = pre + 21;
  const float4 g = B.p_geo[pt];
  const float pu = g.x, pv = g.y, idepth_scaled = g.z, idepth_zero_scaled = g.w;
  {  // projectPoint at the FEJ point (ResidualProjections.h:64-96)
    float KliP[3];
    KliP[0] = (pu + 0 - B.cxl) * B.fxli;
    KliP[1] = (pv + 0 - B.cyl) * B.fyli;
    KliP[2] = 1;
    float ptp[3];
#pragma unroll
    for (int r = 0; r < 3; r++) ptp[r] = ((R0[r * 3 + 0] * KliP[0] + R0[r * 3 + 1] * KliP[1]) + R0[r * 3 + 2] * KliP[2]) + t0[r] * idepth_zero_scaled;
    const float drescale = 1.0f / ptp[2];
    const float u = ptp[0] * drescale, v = ptp[1] * drescale;
    pj[16] = u * B.fxl + B.cxl; pj[17] = v * B.fyl + B.cyl; pj[18] = idepth_zero_scaled * drescale;
  }
#pragma unroll
  for (int idx = 0; idx < 8; idx++) {
    const float up = pu + c_pattern[idx][0], vp = pv + c_pattern[idx][1];
    float q[3];
#pragma unroll
    for (int r = 0; r < 3; r++) q[r] = ((KRKi[r * 3 + 0] * up + KRKi[r * 3 + 1] * vp) + KRKi[r * 3 + 2]) + Kt[r] * idepth_scaled;
    pj[idx * 2] = q[0] / q[2]; pj[idx * 2 + 1] = q[1] / q[2];
  }
  if (!counters || !B.r_isnew[i]) return;
  float pinf[3], pp[3];                               // FullSystemOptimize.cpp:64-76
#pragma unroll
  for (int r = 0; r < 3; r++) { pinf[r] = (KRKi[r * 3 + 0] * pu + KRKi[r * 3 + 1] * pv) + KRKi[r * 3 + 2] * 1.0f; pp[r] = pinf[r] + Kt[r] * idepth_scaled; }
  const float dx = pinf[0] / pinf[2] - pp[0] / pp[2], dy = pinf[1] / pinf[2] - pp[1] / pp[2];
  const float relBS = (float)(0.01 * (double)sqrtf(dx * dx + dy * dy));
  int* tr = (int*)(B.p_track + pt);
  if (relBS == relBS) atomicMax(tr, __float_as_int(relBS));   // if (relBS > p->maxRelBaseline) p->maxRelBaseline = relBS
  atomicAdd(tr + 1, 1);                                       // p->numGoodResiduals++
}

// fixLinearizationF for the active residuals of flagged points; sets isLinearized
__global__ __launch_bounds__(BA_BLOCK) void k_ba_fixlin(const BaDev* __restrict__ wins, const uint8_t* __restrict__ pflag) {
  const BaDev& B = wins[blockIdx.y];
  const int i = blockIdx.x * BA_BLOCK + threadIdx.x;
  if (i >= B.nr) return;
  const int pt = B.r_point[i];
  if (!pflag[pt] || !B.r_act[i]) return;
  float jl[76];
  load_J(B.J[B.r_jsel[i]], B.nrp, i, jl);
  const int S = B.nrp;
  const float* dp = B.t_adHTdelta + (size_t)(B.r_host[i] + B.nf * B.r_target[i]) * 8;
  const float* dc = B.t_cdelta;
  float sx = 0, sy = 0, cx = 0, cy = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) { sx += JV(J_XI0 + k) * dp[k]; sy += JV(J_XI1 + k) * dp[k]; }
#pragma unroll
  for (int k = 0; k < 4; k++) { cx += JV(J_C0 + k) * dc[k]; cy += JV(J_C1 + k) * dc[k]; }
  const float dd = B.p_delta[pt];
  const float dx = sx + cx + JV(J_DD + 0) * dd;
  const float dy = sy + cy + JV(J_DD + 1) * dd;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    float rtz = JV(J_RESF + k);
    rtz = rtz - JV(J_IDX0 + k) * dx;
    rtz = rtz - JV(J_IDX1 + k) * dy;
    rtz = rtz - JV(J_AB0 + k) * dp[6];
    rtz = rtz - JV(J_AB1 + k) * dp[7];
    B.r_toZero[k * S + i] = rtz;
  }
  B.r_lin[i] = 1;
  B.r_rec[(size_t)B.r_orig[i] * 16 + RR_FLAGS] = 3.f;  // active | linearized
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
// upload: state_NewState = OUTLIER, state_NewEnergyWithOutlier = -1 (Residuals.cpp:40-52)
__global__ __launch_bounds__(BA_BLOCK) void k_ba_init_res(const BaDev* __restrict__ wins) {
  const BaDev& B = wins[blockIdx.y];
  const int i = blockIdx.x * BA_BLOCK + threadIdx.x;
  if (i >= B.nr) return;
  B.r_newState[i] = 2;
  B.r_newEnergyWO[i] = -1.f;
}
// resetOOB for every non-linearized residual (FullSystemOptimize.cpp:886-892)
__global__ __launch_bounds__(BA_BLOCK) void k_ba_reset_all(const BaDev* __restrict__ wins) {
  const BaDev& B = wins[blockIdx.y];
  const int i = blockIdx.x * BA_BLOCK + threadIdx.x;
  if (i >= B.nr || B.r_lin[i]) return;
  B.r_energy[i] = 0; B.r_newEnergy[i] = 0; B.r_newState[i] = 2; B.r_state[i] = 0;
}

// ------------------------------------------------------------------ top accumulation
// Block reduction of the 55 + 30 + 6 AccumulatorApprox sums (+ residual count) of one chunk -> top_part[chunk].
//
// The sums over the residuals of a chunk are  D = sum_i  u0_i (x) v0_i + u1_i (x) v1_i  with
//   u0 = [Jpdc0|Jpdxi0] (10), u1 = [Jpdc1|Jpdxi1] (10),
//   v0 = [a*u0 + b*u1 (10) | JabJIdx col 0 (2), JI_r0 | Jab2_00, Jab2_01, Jab_r0],
//   v1 = [b*u0 + c*u1 (10) | JabJIdx col 1 (2), JI_r1 | Jab2_11, Jab_r1, rr],
// plus a constant-1 row per half that turns the last three columns into plain sums: a 12 x 16 x (2*64) product
// per wave.  Summing 92 values over 64 lanes with shuffles costs ~1100 issue slots per residual; instead every
// lane parks its 26 values per half in a wave-private LDS panel (row stride 68 floats: conflict-free for the
// lane-contiguous writes and for the (row = lane%16, k = lane/16) reads) and the matrix cores do the reduction.
//
// The CROSS-RESIDUAL sums run in f64 (round 6): 2 x 16 v_mfma_f64_16x16x4_f64 per wave on the float operands converted exactly, two
// interleaved accumulators, the waves' tiles added in f64, top_part in f64, ONE rounding to float when the pair's chunks have been
// folded into the packed block.  The reference limits the growth of these float sums with its three-tier carry (MatrixAccumulators.h:
// 872-903) and its thread partials (AccumulatedTopHessian.cpp:299-308); a 128-term fmaf chain per wave (rounds 1-5) sat as far from an
// order-independent sum as the CPU path does but not on the same side, and north_star's 1e-5 on the pose update is at that distance
// (profiles/r05_truth_updates.txt).  In f64 the device's sums ARE the order-independent value up to one float rounding; what is left
// between device and CPU is the CPU float path's own distance from it (profiles/r06_truth_updates.txt).  The per-residual operands
// (v0 / v1 included) stay the float expressions they were: their roundings do not accumulate.
constexpr int TE_STRIDE = 68;
constexpr int TE_ROWS = 26;
constexpr int TE_WAVE_FLOATS = TE_ROWS * TE_STRIDE;
constexpr int TE_LDS_FLOATS = (BA_BLOCK / 64) * TE_WAVE_FLOATS;
typedef double te_d4 __attribute__((ext_vector_type(4)));
// How the cross-residual / cross-point sums are carried (compile-time, A/B: tools/mk_variant.sh <name> -DSDSO_ACC_MODE=k; measured on the
// 24 windows of tests/diag/truth_spread.py and the 256-window step, profiles/r06_acc_modes.txt):
//   1 (default)  v_mfma_f64_16x16x4_f64 throughout: first pose update 1.4e-6 (median) from the f64-accumulator truth, the CPU float path
//                1.3e-5; +8 us on k_ba_lin_fused, +10 us on k_ba_sc_host per 256-window step (the f64 form runs at half the fp32 rate);
//   2            fp32 MFMA over 16-term chains from a zero accumulator, the chains added in f64 on the VALU: free in time, but a chain's
//                error grows with its partial sums, not with its length alone — 6.4e-6, half way;
//   0            one fp32 chain per wave (rounds 1-5): 1.1e-5.
#ifndef SDSO_ACC_MODE
#define SDSO_ACC_MODE 1
#endif
constexpr int ACC_MODE = SDSO_ACC_MODE;
#ifndef SDSO_ACC_MODE_SC
#define SDSO_ACC_MODE_SC SDSO_ACC_MODE
#endif
constexpr int ACC_MODE_SC = SDSO_ACC_MODE_SC;      // the Schur kernel's own choice (A/B: top sums in f64 MFMA, Schur sums in short fp32 chains)
static_assert(TE_LDS_FLOATS * 4 >= ((BA_BLOCK / 64) * 256 + BA_BLOCK / 64) * 8, "the waves' f64 tiles lie over the panels");

template <bool LDS_ONLY = false>
__device__ __forceinline__ void top_emit(const BaDev& B, const float* x, const float* y, float a, float b, float c, float TR00, float TR10, float TR01,
                                         float TR11, float TR02, float TR12, const float* br, bool on, float* stage, int chunk = -1 /* default: blockIdx.x */) {
  if (chunk < 0) chunk = (int)blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* S = stage + wv * TE_WAVE_FLOATS;
  const int m = lane & 15, kq = lane >> 4;
  const int mu = m < 10 ? m : 9;
  te_d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};   // ACC_MODE 1: D[i][j], i = lane/16 + 4 v; otherwise i = 4 (lane/16) + v
  te_f4 accf = {0.f, 0.f, 0.f, 0.f};
  const unsigned long long onmask = __ballot(on);
#pragma unroll
  for (int ph = 0; ph < 2; ph++) {
    const float* u = ph ? y : x;
#pragma unroll
    for (int r = 0; r < 10; r++) S[r * TE_STRIDE + lane] = u[r];
#pragma unroll
    for (int cc = 0; cc < 10; cc++) S[(10 + cc) * TE_STRIDE + lane] = ph ? (b * x[cc] + c * y[cc]) : (a * x[cc] + b * y[cc]);
    S[20 * TE_STRIDE + lane] = ph ? TR10 : TR00;
    S[21 * TE_STRIDE + lane] = ph ? TR11 : TR01;
    S[22 * TE_STRIDE + lane] = ph ? TR12 : TR02;
#pragma unroll
    for (int q = 0; q < 3; q++) S[(23 + q) * TE_STRIDE + lane] = br[3 * ph + q];
    wg_barrier<LDS_ONLY>();
    const float one = (m == 10 + ph) ? 1.f : 0.f;
    if (ACC_MODE == 1) {
#pragma unroll
      for (int s4 = 0; s4 < 16; s4 += 2) {
        float av0 = S[mu * TE_STRIDE + 4 * s4 + kq], av1 = S[mu * TE_STRIDE + 4 * s4 + 4 + kq];
        av0 = m < 10 ? av0 : one; av1 = m < 10 ? av1 : one;
        const float bv0 = S[(10 + m) * TE_STRIDE + 4 * s4 + kq], bv1 = S[(10 + m) * TE_STRIDE + 4 * s4 + 4 + kq];
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av0, (double)bv0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av1, (double)bv1, acc1, 0, 0, 0);
      }
    } else {
      // four chains of four steps (16 residuals each), two in flight; a chain's result joins the f64 sum while the next ones run
      te_f4 ch[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        ch[q] = (te_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 4 * q; s4 < 4 * q + 4; s4++) {
          float av = S[mu * TE_STRIDE + 4 * s4 + kq];
          av = m < 10 ? av : one;
          const float bv = S[(10 + m) * TE_STRIDE + 4 * s4 + kq];
          if (ACC_MODE == 0) accf = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, accf, 0, 0, 0);
          else ch[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, ch[q], 0, 0, 0);
        }
      }
      if (ACC_MODE == 2) {
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
          for (int v = 0; v < 4; v++) acc0[v] += (double)ch[q][v];
      }
    }
    wg_barrier<LDS_ONLY>();
  }
  double* R = (double*)stage;
#pragma unroll
  for (int v = 0; v < 4; v++) {
    if (ACC_MODE == 1) R[wv * 256 + (kq + 4 * v) * 16 + m] = acc0[v] + acc1[v];
    else R[wv * 256 + (4 * kq + v) * 16 + m] = ACC_MODE == 0 ? (double)accf[v] : acc0[v];
  }
  if (lane == 0) R[(BA_BLOCK / 64) * 256 + wv] = (double)__popcll(onmask);
  wg_barrier<LDS_ONLY>();
  if (threadIdx.x < 92) {
    const int t = threadIdx.x;
    int off;
    if (t < 55) { int r = 0, rem = t; while (rem >= 10 - r) { rem -= 10 - r; r++; } off = r * 16 + r + rem; }
    else if (t < 85) off = ((t - 55) / 3) * 16 + 10 + (t - 55) % 3;
    else if (t < 88) off = 10 * 16 + 13 + (t - 85);
    else if (t < 91) off = 11 * 16 + 13 + (t - 88);
    else off = (BA_BLOCK / 64) * 256;   // count slots follow the tiles
    double s = R[off];
#pragma unroll
    for (int w = 1; w < BA_BLOCK / 64; w++) s += R[off + (t < 91 ? w * 256 : w)];
    B.top_part[(size_t)chunk * 92 + t] = s;
  }
}

// One workgroup per chunk of <=256 residuals of ONE (host,target) pair.  mode: 0 active, 1 linearized, 2 marginalise.
__global__ __launch_bounds__(BA_BLOCK) void k_ba_accum_top(const BaDev* __restrict__ wins, int mode, const uint8_t* __restrict__ pflag) {
  const BaDev& B = wins[blockIdx.y];
  if (ba_finished(B)) return;
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
  __shared__ float red[TE_LDS_FLOATS];
  float x[10], y[10], a = 0, b = 0, c = 0;
  float TR00 = 0, TR10 = 0, TR01 = 0, TR11 = 0, TR02 = 0, TR12 = 0;
  float br[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 10; k++) { x[k] = 0; y[k] = 0; }
  if (on) {
    float jl[76];
    load_J(B.J[B.r_jsel[i]], S, i, jl);
    const int pt = B.r_point[i];
    (void)pt;
    float resApprox[8];
    if (mode == 0) {
#pragma unroll
      for (int k = 0; k < 8; k++) resApprox[k] = JV(J_RESF + k);
    } else if (mode == 2) {
#pragma unroll
      for (int k = 0; k < 8; k++) resApprox[k] = B.r_toZero[k * S + i];
    } else {
      const float* dp = B.t_adHTdelta + (size_t)ch.x * 8;
      const float* dc = B.t_cdelta;
      float sx = 0, sy = 0, cx = 0, cy = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) { sx += JV(J_XI0 + k) * dp[k]; sy += JV(J_XI1 + k) * dp[k]; }
#pragma unroll
      for (int k = 0; k < 4; k++) { cx += JV(J_C0 + k) * dc[k]; cy += JV(J_C1 + k) * dc[k]; }
      const float dd = B.p_delta[pt];
      const float dx = sx + cx + JV(J_DD + 0) * dd;
      const float dy = sy + cy + JV(J_DD + 1) * dd;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        float rtz = B.r_toZero[k * S + i];
        rtz = rtz + JV(J_IDX0 + k) * dx;
        rtz = rtz + JV(J_IDX1 + k) * dy;
        rtz = rtz + JV(J_AB0 + k) * dp[6];
        rtz = rtz + JV(J_AB1 + k) * dp[7];
        resApprox[k] = rtz;
      }
    }
    float JI_r0 = 0, JI_r1 = 0, Jab_r0 = 0, Jab_r1 = 0, rr = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      JI_r0 += resApprox[k] * JV(J_IDX0 + k);
      JI_r1 += resApprox[k] * JV(J_IDX1 + k);
      Jab_r0 += resApprox[k] * JV(J_AB0 + k);
      Jab_r1 += resApprox[k] * JV(J_AB1 + k);
      rr += resApprox[k] * resApprox[k];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) { x[k] = JV(J_C0 + k); y[k] = JV(J_C1 + k); }
#pragma unroll
    for (int k = 0; k < 6; k++) { x[4 + k] = JV(J_XI0 + k); y[4 + k] = JV(J_XI1 + k); }
    a = JV(J_IDX2 + 0); b = JV(J_IDX2 + 1); c = JV(J_IDX2 + 3);
    TR00 = JV(J_ABIDX + 0); TR10 = JV(J_ABIDX + 1); TR01 = JV(J_ABIDX + 2); TR11 = JV(J_ABIDX + 3);
    TR02 = JI_r0; TR12 = JI_r1;
    br[0] = JV(J_AB2 + 0); br[1] = JV(J_AB2 + 1); br[2] = Jab_r0; br[3] = JV(J_AB2 + 3); br[4] = Jab_r1; br[5] = rr;
    // per-residual idepth terms (AccumulatedTopHessian.cpp:160-172) -> record
    const float jdd0 = JV(J_DD + 0), jdd1 = JV(J_DD + 1);
    const float a10 = JV(J_IDX2 + 2);
    const float q0 = a * jdd0 + b * jdd1;
    const float q1 = a10 * jdd0 + c * jdd1;
    float* rec = B.r_rec + (size_t)B.r_orig[i] * 16;
    rec[RR_BD] = JI_r0 * jdd0 + JI_r1 * jdd1;
    rec[RR_HDD] = q0 * jdd0 + q1 * jdd1;
#pragma unroll
    for (int k = 0; k < 4; k++) rec[RR_HCD + k] = x[k] * q0 + y[k] * q1;
  }
  top_emit(B, x, y, a, b, c, TR00, TR10, TR01, TR11, TR02, TR12, br, on, red);
}

// ------------------------------------------------------------------ fused linearize + applyRes + accumulate (mode 0)
// FullSystem::optimize with setting_forceAceptStep (the reference default, settings.cpp:53) always applies
// the fresh linearization, so linearizeAll -> applyRes_Reductor(true) -> accumulateAF of the next
// solveSystemF can run back to back on the same residual while its Jacobian is still in registers:
// one workgroup per chunk of one (host,target) pair, J is written to HBM only when MATERIALIZE
// (the reference API keeps RawResidualJacobian; the solver itself never reads it again).
// Linearized residuals are untouched (their accumulation is the separate mode-1 pass).
// Workgroups per CU: four without the Jacobian stores (124 VGPRs, 40 KB of LDS each); with them the kernel needs 152 VGPRs
// (capping it at 128 spills 22-38 of them and loses more than the fourth wave per SIMD gains), so three.
// The 32 taps of a residual come in through the cooperative quad gather (linearize_coop).  (Rounds 1-4 kept two more gathers for A/B — one
// residual's taps on one lane, and LDS-DMA rounds; both lost or tied, profiles/README.md.)
// The workgroup barriers behind the linearisation exchange data through LDS only (the gather stage handed over to the reduction's panels, the
// energy sum, the waves' tiles).  -DSDSO_LIN_LDS_BARRIERS=1 makes them wait for the wave's LDS traffic only, not for its outstanding
// Jacobian-record stores (`__syncthreads()` is s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier) — measured: no difference (0.772-0.777 against
// 0.770-0.774 ms, profiles/r06_lin_barriers_ab.txt: two other workgroups per CU run under a wave's wait), so the plain barriers stay.
#ifndef SDSO_LIN_LDS_BARRIERS
#define SDSO_LIN_LDS_BARRIERS 0
#endif
constexpr bool LIN_LDS_BARRIERS = SDSO_LIN_LDS_BARRIERS != 0;
template <bool MATERIALIZE, bool TILED>
__global__ __launch_bounds__(BA_BLOCK, (MATERIALIZE ? 3 : 4)) void k_ba_lin_fused(const BaDev* __restrict__ wins) {
  constexpr bool COOP = true;
  // by-value copy first: every pointer of the descriptor is read before the kernel's first store, so the
  // compiler can prove them global (global_load / s_load instead of flat_load) and keep them in SGPRs
  const BaDev B = wins[blockIdx.y];
  if (ba_finished_lin(B)) return;
  if ((int)blockIdx.x >= B.nchunks) return;
  constexpr int STAGE_FLOATS = (BA_BLOCK / 64) * (COOP ? CG_WAVE_FLOATS : 0);
  constexpr int RED_FLOATS = TE_LDS_FLOATS > STAGE_FLOATS ? TE_LDS_FLOATS : STAGE_FLOATS;
  __shared__ float red[RED_FLOATS];    // the gather stage of the linearisation, then the MFMA panels of the reduction
  double* const lds = (double*)red;    // (the energy reduction runs between the two uses; 40 KB in all = four workgroups per CU)
#ifdef SDSO_LIN_PERSIST
  // A/B (round 6, verdict item 5a): a persistent workgroup walks the chunks blockIdx.x, blockIdx.x + gridDim.x, ... of its window (launched with
  // half the chunks as grid: two chunks per workgroup) and requests the NEXT chunk's descriptor, r_state / r_lin / r_point at the top of the current
  // one and its p_geo once the current chunk's taps are through — three dependent round trips of the next chunk under the current one's work.
  int4 chN = B.chunks[blockIdx.x];
  bool liveN = false; LinPre preN{1, 0, make_float4(0.f, 0.f, 0.f, 0.f)};
  {
    const int i0 = chN.y + threadIdx.x;
    liveN = (int)threadIdx.x < chN.z && !B.r_lin[i0];
    if (liveN) { preN.st = B.r_state[i0]; preN.pt = B.r_point[i0]; }
#if SDSO_LIN_PERSIST >= 2
    if (liveN) preN.geo = B.p_geo[preN.pt];
#endif
  }
  for (int chunk = blockIdx.x; chunk < B.nchunks; chunk += gridDim.x) {
  const int4 ch = chN;
  const bool live = liveN;
  const LinPre preC = preN;
  const int nxt = chunk + (int)gridDim.x;
  if (nxt < B.nchunks) {
    chN = B.chunks[nxt];
    const int in = chN.y + threadIdx.x;
    liveN = (int)threadIdx.x < chN.z && !B.r_lin[in];
    if (liveN) { preN.st = B.r_state[in]; preN.pt = B.r_point[in]; }
  }
#else
  const int chunk = blockIdx.x;
  const int4 ch = B.chunks[blockIdx.x];
#endif
  const int pair = __builtin_amdgcn_readfirstlane(ch.x);   // one (host,target) per workgroup: precalc, image, thresholds are wave-uniform
  const int i = ch.y + threadIdx.x;
  float x[10], y[10], a = 0, b = 0, c = 0;
  float TR00 = 0, TR10 = 0, TR01 = 0, TR11 = 0, TR02 = 0, TR12 = 0;
  float br[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 10; k++) { x[k] = 0; y[k] = 0; }
  bool on = false;
  double e = 0;
#ifndef SDSO_LIN_PERSIST
  const bool live = (int)threadIdx.x < ch.z && !B.r_lin[i];
#endif
  float jl[76];
  float rs5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  int ns = 1;
  uint8_t st = 1;
  if (COOP) {
#ifdef SDSO_LIN_PERSIST
    if (live) st = (uint8_t)preC.st;
    e = linearize_coop<MATERIALIZE, 2, TILED>(B, i, live, pair % B.nf, pair / B.nf, jl, ns, rs5, red + (threadIdx.x >> 6) * CG_WAVE_FLOATS, &preC);
#if SDSO_LIN_PERSIST >= 2
    if (nxt < B.nchunks && liveN) preN.geo = B.p_geo[preN.pt];      // (the next chunk's point index arrived long ago)
#endif
#else
    if (live) st = B.r_state[i];
    e = linearize_coop<MATERIALIZE, 2, TILED>(B, i, live, pair % B.nf, pair / B.nf, jl, ns, rs5, red + (threadIdx.x >> 6) * CG_WAVE_FLOATS);
#endif
    wg_barrier<LIN_LDS_BARRIERS>();   // the reduction below reuses the stage of all waves
  }
  if (live) {
    float* rec = B.r_rec + (size_t)B.r_orig[i] * 16;
    // the 64-byte record of this residual (BaDev::r_rec): written once, whole, at the end (four 16-byte stores of one half line)
    float o8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, rbd = 0.f, rhdd = 0.f, rhcd[4] = {0.f, 0.f, 0.f, 0.f};
    uint8_t act = 0;
    if (st != 1) {  // applyRes(true): OOB is sticky
      if (ns == 0) {
        act = 1;
        if (MATERIALIZE && !B.jfix) B.r_jsel[i] ^= 1;
        const float jdd0 = JV(J_DD + 0), jdd1 = JV(J_DD + 1);
        const float v0 = JV(J_IDX2 + 0) * jdd0 + JV(J_IDX2 + 1) * jdd1;
        const float v1 = JV(J_IDX2 + 2) * jdd0 + JV(J_IDX2 + 3) * jdd1;
#pragma unroll
        for (int k = 0; k < 6; k++) o8[k] = JV(J_XI0 + k) * v0 + JV(J_XI1 + k) * v1;
        o8[6] = JV(J_ABIDX + 0) * jdd0 + JV(J_ABIDX + 1) * jdd1;
        o8[7] = JV(J_ABIDX + 2) * jdd0 + JV(J_ABIDX + 3) * jdd1;
      }
      B.r_act[i] = act;
      B.r_state[i] = (uint8_t)ns;
      B.r_energy[i] = B.r_newEnergy[i];
      on = act != 0;
    } else {
      on = B.r_act[i] != 0;   // (an OOB residual is never active)
    }
    if (on) {   // AccumulatedTopHessianSSE::addPoint<0>, resApprox = resF
      const float JI_r0 = rs5[0], JI_r1 = rs5[1], Jab_r0 = rs5[2], Jab_r1 = rs5[3], rr = rs5[4];
#pragma unroll
      for (int k = 0; k < 4; k++) { x[k] = JV(J_C0 + k); y[k] = JV(J_C1 + k); }
#pragma unroll
      for (int k = 0; k < 6; k++) { x[4 + k] = JV(J_XI0 + k); y[4 + k] = JV(J_XI1 + k); }
      a = JV(J_IDX2 + 0); b = JV(J_IDX2 + 1); c = JV(J_IDX2 + 3);
      TR00 = JV(J_ABIDX + 0); TR10 = JV(J_ABIDX + 1); TR01 = JV(J_ABIDX + 2); TR11 = JV(J_ABIDX + 3);
      TR02 = JI_r0; TR12 = JI_r1;
      br[0] = JV(J_AB2 + 0); br[1] = JV(J_AB2 + 1); br[2] = Jab_r0; br[3] = JV(J_AB2 + 3); br[4] = Jab_r1; br[5] = rr;
      const float jdd0 = JV(J_DD + 0), jdd1 = JV(J_DD + 1);
      const float q0 = a * jdd0 + b * jdd1;
      const float q1 = JV(J_IDX2 + 2) * jdd0 + c * jdd1;
      rbd = JI_r0 * jdd0 + JI_r1 * jdd1;
      rhdd = q0 * jdd0 + q1 * jdd1;
#pragma unroll
      for (int k = 0; k < 4; k++) rhcd[k] = x[k] * q0 + y[k] * q1;
    }
    if (st != 1) {   // (a sticky-OOB residual keeps the records its last applyRes wrote: flags 0)
      *(float4*)(rec) = make_float4(o8[0], o8[1], o8[2], o8[3]);
      *(float4*)(rec + 4) = make_float4(o8[4], o8[5], o8[6], o8[7]);
      *(float4*)(rec + 8) = make_float4(rbd, rhdd, rhcd[0], rhcd[1]);
      *(float4*)(rec + 12) = make_float4(rhcd[2], rhcd[3], (float)act, 0.f);
    }
  }
  e = block_sum_d<LIN_LDS_BARRIERS>(e, lds);
  if (threadIdx.x == 0) B.e_part[chunk] = e;
  wg_barrier<LIN_LDS_BARRIERS>();
  top_emit<LIN_LDS_BARRIERS>(B, x, y, a, b, c, TR00, TR10, TR01, TR11, TR02, TR12, br, on, red, chunk);
#ifdef SDSO_LIN_PERSIST
  __syncthreads();   // the next chunk's gather stage lies over the panels
  }
#endif
}

// ------------------------------------------------------------------ linearised energy (EnergyFunctional::calcLEnergyPt, EnergyFunctional.cpp:354-417)
// sum over linearized & active residuals of (2*res_toZeroF + J*delta) * J*delta, plus deltaF^2 * priorF per point.
// grid.x = nchunks (one (host,target) pair each: adHTdeltaF is uniform) + point blocks; one float partial per workgroup.
__global__ __launch_bounds__(BA_BLOCK) void k_ba_lenergy(const BaDev* __restrict__ wins, float* __restrict__ out, int out_stride = 0 /* floats between the windows' partials */,
                                                         int cond = 0) {
  const BaDev& B = wins[blockIdx.y];
  if (ba_finished_lin(B) || ba_gate_skip(B, cond)) return;
  if ((int)blockIdx.x >= B.nchunks + (B.np + BA_BLOCK - 1) / BA_BLOCK) return;
  out += (size_t)blockIdx.y * out_stride;
  float e = 0.f;
  if ((int)blockIdx.x < B.nchunks) {
    const int4 ch = B.chunks[blockIdx.x];
    const int i = ch.y + threadIdx.x;
    if ((int)threadIdx.x < ch.z && (B.r_lin[i] & 1) && B.r_act[i]) {
      float jl[76];
      load_J(B.r_jsel[i] ? B.J[1] : B.J[0], B.nrp, i, jl);
      const float* dp = B.t_adHTdelta + (size_t)ch.x * 8;
      const float* dc = B.t_cdelta;
      const float dd = B.p_delta[B.r_point[i]];
      float sx = 0, sy = 0, cx = 0, cy = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) { sx += JV(J_XI0 + k) * dp[k]; sy += JV(J_XI1 + k) * dp[k]; }
#pragma unroll
      for (int k = 0; k < 4; k++) { cx += JV(J_C0 + k) * dc[k]; cy += JV(J_C1 + k) * dc[k]; }
      const float dx = sx + cx + JV(J_DD + 0) * dd;
      const float dy = sy + cy + JV(J_DD + 1) * dd;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        float Jdelta = JV(J_IDX0 + k) * dx;
        Jdelta = Jdelta + JV(J_IDX1 + k) * dy;
        Jdelta = Jdelta + JV(J_AB0 + k) * dp[6];
        Jdelta = Jdelta + JV(J_AB1 + k) * dp[7];
        float r0 = B.r_toZero[k * B.nrp + i];
        r0 = r0 + r0;
        r0 = r0 + Jdelta;
        e += Jdelta * r0;
      }
    }
  } else {
    const int p = ((int)blockIdx.x - B.nchunks) * BA_BLOCK + threadIdx.x;
    if (p < B.np) { const float d = B.p_delta[p]; e = d * d * B.p_prior[p]; }
  }
  __shared__ float red[BA_BLOCK / 64];
  e = wave_sum(e);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = e;
  __syncthreads();
  if (threadIdx.x == 0) { float s = 0; for (int w = 0; w < BA_BLOCK / 64; w++) s += red[w]; out[blockIdx.x] = s; }
}

// fold chunk partials per pair (fixed order) into the packed accumulator; one 128-thread workgroup per pair
__device__ __forceinline__ void fold_top_body(const BaDev& B, int pair, int which /*0 = A, 1 = L*/, int tid) {
  if (pair >= B.nf * B.nf) return;
  const int cb = B.pair_chunk_beg[pair], ce = B.pair_chunk_beg[pair + 1];
  float* out = B.accum + (which ? acc_off_topL(B.nf) : acc_off_topA(B.nf)) + (size_t)pair * 91;
  if (tid < 91) {
    double s = 0;
    for (int ck = cb; ck < ce; ck++) s += B.top_part[(size_t)ck * 92 + tid];
    out[tid] = (float)s;
  }
  if (pair == 0 && tid == 127) {
    double s = 0;
    for (int ck = 0; ck < B.nchunks; ck++) s += B.top_part[(size_t)ck * 92 + 91];
    B.accum[acc_off_nres(B.nf) + which] = (float)s;
  }
}
__device__ __forceinline__ void zero_topL_body(const BaDev& B, int pair, int tid) {
  if (pair >= B.nf * B.nf) return;
  if (tid < 91) B.accum[acc_off_topL(B.nf) + (size_t)pair * 91 + tid] = 0.f;
  if (pair == 0 && tid == 127) B.accum[acc_off_nres(B.nf) + 1] = 0.f;
}
__global__ __launch_bounds__(128) void k_ba_fold_top(const BaDev* __restrict__ wins, int which) { if (ba_finished(wins[blockIdx.y])) return; fold_top_body(wins[blockIdx.y], blockIdx.x, which, threadIdx.x); }
__global__ __launch_bounds__(128) void k_ba_zero_topL(const BaDev* __restrict__ wins) { if (ba_finished(wins[blockIdx.y])) return; zero_topL_body(wins[blockIdx.y], blockIdx.x, threadIdx.x); }

// ------------------------------------------------------------------ Schur accumulation, one workgroup per host frame
// AccumulatedSCHessianSSE::addPoint (AccumulatedSCHessian.cpp:34-103) for all points of ONE host frame of ONE window, together with the
// per-point sums addPoint<mode> of the top accumulator leaves in the EFPoint (AccumulatedTopHessian.cpp:160-192).
//
// The four waves of the workgroup share the host's points (64-point slices dealt round-robin, 16-point groups inside a slice).  A group's
// records — the contiguous stretch [p_rbeg[p0], p_rbeg[p0 + 16]) of BaDev::r_rec, 64 bytes per residual, no padding for targets a point does
// not observe — travel straight into LDS by `global_load_lds_dwordx4` (no destination registers, nothing to park), one group ahead of the
// arithmetic, and are split on the way: the DMA takes a global address per lane and writes lane-contiguous LDS, so the JpJdF halves land in
// one buffer (double-buffered: phase 2 of group g reads it while group g + 1 arrives) and the term halves in another (single: phase 1 is
// done with it before the next DMA is issued).  The JpJdF halves also go back out, compact, for the back-substitution (BaDev::r_cj).
//   phase 1  per-point terms on ALL 64 lanes: lane = (point, j), j = 0 {bd, Hdd}, 1 {Hcd0, Hcd1}, 2 {Hcd2, Hcd3}, 3 {flags}.  A lane walks
//            its float pair of the point's (<= 8) records in EFPoint::residualsAll order — which IS the order the records lie in — so Hdd / bd /
//            Hcd, HdiF and bdSumF are the reference's additions one after the other, bit for bit, whatever dropResidual did to that order
//            (EnergyFunctional.cpp:524-533); the eight LDS reads of a lane are independent and issued together.  Lane 3 also builds the
//            point's target -> record map (a word of nibbles) for phase 2.  The quad exchanges H / `any` by DPP and stores the point's 32 bytes
//            of BaDev::p_out ([8..13], the sums of linearised / marginalised residuals, only when such residuals exist or their stale values
//            have to be cleared) and idepth_hessian / the active-record mask / `maxRelBaseline = 0` (:44-48) in BaDev::p_track.
//   phase 2  Z^T diag(HdiF) Z on the matrix cores (v_mfma_f64_16x16x4_f64 since round 6 — ACC_MODE_SC —, K = 4 points), Z = [JpJdF of the 7 targets a
//            point can observe (never its host) | Hcd | bdSumF] (61 columns in a 4 x 4 grid of 16-column tiles): only the 10 tiles on and above the
//            diagonal are accumulated (80 accumulator registers in f64); the bins below the
//            diagonal are written as mirror images, which makes accD(i,j,k) == accD(i,k,j)^T hold EXACTLY as it does in the reference
//            (AccumulatorXX::update multiplies a_i * b_j * w: the same product for both, MatrixAccumulators.h:31-80) — the stitch relies on it
//            (ba_tail.hip: S2 = S1^T).  Absent targets, residuals that are not active and points without an active residual are exact zeros.
// The waves' tiles are added through LDS in a fixed order (3+2 -> 1+0 -> 0), laid out the way the bins lie in memory and written by all
// four waves as whole 256-byte blocks straight into the packed accumulator block: no per-item partials in HBM, no fold pass.  Only Hcc / bc
// (sums over ALL hosts) leave a 20-float partial per host.
// PLAIN: no marginalisation pass, no point filter, no linearized residual in the launch (every GN iteration of the reference's live flow):
// all active residuals go to the A sums, and the records of a residual that is not active hold zeros — x + 0 is exact — so no flag is read
// on the value lanes.  clearL: also store zeros into the L sums of p_out (a launch with linearised residuals left values there).
// Two workgroups per CU: 217 VGPRs (the f64 tiles), 51 KB of LDS.
constexpr int SCH_REC = 1024;                         // floats of one record buffer: 16 points x 8 records x 8 floats
constexpr int SCH_WAVE = 3 * SCH_REC + 16 * 8;        // two JpJdF buffers (double-buffered), one term buffer, the points' phase-2 operands
// WPH ("wave per host", the form of a large batch): every WAVE takes a whole host frame — all its 16-point groups in a row — and a workgroup
// four hosts of one window.  The host's sums never leave the wave's accumulators: no tree over the waves, no barrier; the bins go from the
// registers straight to memory (every (t1, t2) block of 256 bytes is covered by the four stores of one tile; the blocks below the diagonal
// as float4s of the mirrored tile; the five tiles ON the diagonal are mirrored through 1 KB of the wave's stage first).  A batch of 256
// windows is 2 048 waves on 3 072 slots — one round — where 2 048 workgroups on 768 slots need three: 116 -> see DESIGN.md §4.  A single
// window keeps the workgroup per host (a quarter of the latency).
template <bool PLAIN, bool WPH>
__global__ __launch_bounds__(BA_BLOCK, 2) void k_ba_sc_host(const BaDev* __restrict__ wins, const uint8_t* __restrict__ pflag, int shiftPriorToZero, int margMode, int clearL) {
  constexpr int NW = BA_BLOCK / 64;
  const BaDev& B = wins[blockIdx.y];
  if (ba_finished(B)) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nf = B.nf, h = WPH ? (int)blockIdx.x * NW + wv : (int)blockIdx.x;
  if (h >= nf) return;                       // (WPH: a wave of its own — the form has no workgroup barrier)
#ifdef SDSO_SC_STAMPS   // diagnostic build (tools/mk_variant.sh scst -DSDSO_SC_STAMPS): shader-clock ticks of wave 0 of two workgroups per phase, printed
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int stn = 0;
#define SCS() do { if (stn < 8) st[stn++] = __builtin_amdgcn_s_memtime(); } while (0)
  unsigned long long sub[6] = {0, 0, 0, 0, 0, 0}, sub_t = 0;     // inside the group loop: wait for the records | r_cj | phase 1 | DMA issue + inputs | phase 2
#define SCSUB(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); sub[i] += tn_ - sub_t; sub_t = tn_; } while (0)
#define SCSUB0() do { sub_t = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SCS() do { } while (0)
#define SCSUB(i) do { } while (0)
#define SCSUB0() do { } while (0)
#endif
  SCS();
  constexpr int SC_NT = 10;                                  // tile t of pair (a <= b) of the 4 x 4 tile grid: sc_ut(a, b)
  constexpr int SC_NEED = 2 * 2 * SC_NT * 256;                // the tree's two tile buffers (f64)
  constexpr int SC_LDS = NW * SCH_WAVE > SC_NEED ? NW * SCH_WAVE : SC_NEED;
  __shared__ __align__(16) float stage_all[SC_LDS];
  float* const bufA = stage_all + wv * SCH_WAVE;             // [2][SCH_REC]
  float* const bufT = bufA + 2 * SCH_REC;
  float (*pt)[8] = reinterpret_cast<float (*)[8]>(bufT + SCH_REC);
  const int pb = B.host_pt_beg[h], pe = B.host_pt_beg[h + 1];     // (in the descriptor itself: no dependent round trip before the first DMA)
  // the sums over the points are carried in f64 like the top sums (top_emit, ACC_MODE_SC): a group's 16 points are one short fp32 MFMA chain
  // per tile, the groups' tiles are added in f64 (mode 1: f64 MFMAs on the float operands converted exactly, (w z_a) z_b with the
  // product w z_a exact); ONE rounding to float when the bins are written
  te_d4 acc[SC_NT];
  te_f4 grp[SC_NT];                              // mode 2: the tiles of one group; mode 0: the running float sums
#pragma unroll
  for (int t = 0; t < SC_NT; t++) { acc[t] = (te_d4){0.0, 0.0, 0.0, 0.0}; grp[t] = (te_f4){0.f, 0.f, 0.f, 0.f}; }
  auto sc_ut = [](int a, int b) { return a * 4 - (a * (a - 1)) / 2 + (b - a); };   // index of the upper tile (a <= b): 0..9
  const int ci = lane & 15, kq = lane >> 4;
  const int tsub = ci >> 3, asub = ci & 7;
  const int pl = lane >> 2, jq = lane & 3;                   // phase 1: point of the group, float pair of the term record
  // a wave's 16-point groups: 64-point slices dealt round-robin over the waves, four groups per slice
  auto group_p0 = [&](int gidx) { return WPH ? pb + 16 * gidx : pb + 64 * (wv + NW * (gidx >> 2)) + 16 * (gidx & 3); };
  // what a group needs beside its records, fetched one group ahead: the first record of its points (lane L: point min(L, npts), so lane 16
  // and everything past the group's end holds the END of its stretch) and, in the phase-1 layout, prior / delta / order / filter of the point
  struct GroupIn { int rb; float prior, delta; unsigned order; int on; };
  auto fetch_in = [&](int p0, GroupIn& g) {
    const int npts = min(16, pe - p0);
    g.rb = 0; g.prior = 0.f; g.delta = 0.f; g.order = 0xffffffffu; g.on = 0;
    if (npts <= 0) return;
    g.rb = B.p_rbeg[p0 + min(lane, npts)];
    if (pl < npts) {
      g.prior = B.p_prior[p0 + pl]; g.delta = B.p_delta[p0 + pl]; g.order = B.p_order[p0 + pl];
      g.on = pflag ? (int)pflag[p0 + pl] : 1;
    }
  };
  // records [r0, r0 + nrec) -> LDS: 16-byte piece c (record c / 2, half c % 2) of the JpJdF halves lands at float 4 c of dstA, of the term
  // halves at float 4 c of dstT
  auto dma_records = [&](int r0, int nrec, float* dstA, float* dstT) {
    const int npieces = nrec * 2;
    const float* g = B.r_rec + (size_t)r0 * 16;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int c = lane + 64 * k;
      if (64 * k < npieces) {      // (wave-uniform)
        if (c < npieces) {
          const float* src = g + (size_t)(c >> 1) * 16 + 4 * (c & 1);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(dstA + 256 * k), 16, 0, 0);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 8), (__attribute__((address_space(3))) void*)(dstT + 256 * k), 16, 0, 0);
        }
      }
    }
  };
  GroupIn gcur, gnext;
  fetch_in(group_p0(0), gcur);
  {
    const int r0 = __builtin_amdgcn_readlane(gcur.rb, 0), r1 = __builtin_amdgcn_readlane(gcur.rb, 16);
    if (group_p0(0) < pe) dma_records(r0, r1 - r0, bufA, bufT);
  }
  fetch_in(group_p0(1), gnext);
  SCS();
  for (int gidx = 0;; gidx++) {
    const int p0 = group_p0(gidx);
    if (p0 >= pe) break;
    const int npts = min(16, pe - p0);
    float* const curA = bufA + (gidx & 1) * SCH_REC;
    const int r0 = __builtin_amdgcn_readlane(gcur.rb, 0);
    SCSUB0();
    // this group's records (and the inputs of the next one) have landed; the LDS-DMA is invisible to the compiler's own scoreboard
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    SCSUB(0);
    {  // the JpJdF halves, compact, for the back-substitution: coalesced 1-KB stores straight from the stage
      const int npieces = (__builtin_amdgcn_readlane(gcur.rb, 16) - r0) * 2;
      float* dst = B.r_cj + (size_t)r0 * 8;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int c = lane + 64 * k;
        if (c < npieces) *(float4*)(dst + (size_t)c * 4) = *(const float4*)(curA + c * 4);
      }
    }
    SCSUB(1);
    {  // ---- phase 1
      const int base = __shfl(gcur.rb, pl, 64) - r0;           // first record of the lane's point inside the stage
      const int cnt = 8 - (__clz((int)~gcur.order) >> 2);           // its records: the nibbles of `order` that are not 0xF (targets are < 8)
      const bool onp = gcur.on != 0;
      float ax = 0.f, ay = 0.f, lx = 0.f, ly = 0.f;            // A and L sums of the lane's float pair
      int ngood = 0, mbits = 0;
      unsigned inv = 0xffffffffu;                              // nibble t: the record of the point's residual to target t
      float2 v[8];
      float fl[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const float* rec = bufT + (base + (k < cnt ? k : 0)) * 8;
        v[k] = *(const float2*)(rec + 2 * jq);
        fl[k] = PLAIN ? 0.f : rec[RR_FLAGS - 8];
      }
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const bool in = k < cnt;
        const unsigned t = (gcur.order >> (4 * k)) & 15u;
        if (PLAIN) {
          // (jq == 3: v.x is the flags word; the value lanes add records as they are — zeros where the residual is not active)
          const int act = in ? ((int)v[k].x & 1) : 0;
          ngood += act; mbits |= act << k;
          ax += in ? v[k].x : 0.f; ay += in ? v[k].y : 0.f;
        } else {
          const int f = in ? (int)fl[k] : 0;
          const bool m = (f & 1) != 0 && onp;                  // residual active (and the point taking part)
          const bool mA = m && !(f & 2) && !margMode, mL = m && !mA;   // mode 0 vs mode 1 / 2 sums (AccumulatedTopHessian.cpp:54-71)
          if (m) { ngood++; mbits |= 1 << k; }
          // x + 0 is exact: masked-off residuals leave the sums untouched
          ax += mA ? v[k].x : 0.f; ay += mA ? v[k].y : 0.f;
          lx += mL ? v[k].x : 0.f; ly += mL ? v[k].y : 0.f;
        }
        if (in) inv = (inv & ~(15u << (4 * t))) | ((unsigned)k << (4 * t));
      }
      // lane 0 of the quad: bd (x), Hdd (y)
      float H = ay + ly + gcur.prior;
      if (H < 1e-10) H = 1e-10;
      const float hdi = 1.0 / H;
      float bds = ax + lx;
      if (shiftPriorToZero) bds += gcur.prior * gcur.delta;
      const bool any = __builtin_amdgcn_mov_dpp(ngood, 0xff, 0xf, 0xf, true) > 0;     // quad lane 3 counted the active residuals
      const float H0 = quad_bcast<0>(H);
      if (pl < npts && onp) {
        float* po = B.p_out + (size_t)(p0 + pl) * 16;
        if (jq == 0) *(float4*)(po + PO_HDD_A) = make_float4(ay, ax, any ? hdi : 0.f, any ? bds : 0.f);
        else if (jq < 3) *(float2*)(po + PO_HCD_A + 2 * (jq - 1)) = make_float2(ax, ay);
        else {
          float* tr = (float*)(B.p_track + p0 + pl);
          *(float2*)(tr + 2) = make_float2(any ? H0 : 0.f, __int_as_float(mbits));   // p->data->idepth_hessian (:46, :56); the active records for k_ba_resub*
          if (!any) tr[0] = 0.f;                                                       // p->data->maxRelBaseline = 0 (:47)
        }
        if (!PLAIN || clearL) {
          if (jq == 0) *(float2*)(po + PO_HDD_L) = make_float2(ly, lx);
          else if (jq < 3) *(float2*)(po + PO_HCD_L + 2 * (jq - 1)) = make_float2(lx, ly);
        }
      }
      // the point's operands of phase 2: HdiF, bdSumF, Hcd[4] (zeros without an active residual), the target -> record map, its first record
      float2 o2;
      if (jq == 0) o2 = make_float2(any ? hdi : 0.f, any ? bds : 0.f);
      else if (jq < 3) o2 = make_float2(any ? ax + lx : 0.f, any ? ay + ly : 0.f);
      else o2 = make_float2(__int_as_float((int)inv), __int_as_float(base));
      *(float2*)(&pt[pl][2 * jq]) = o2;
    }
    // the term buffer and (since the last iteration's phase 2) the other JpJdF buffer are free: the next group's records may come
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    SCSUB(2);
    {
      const int pn = group_p0(gidx + 1);
      if (pn < pe) {
        const int n0 = __builtin_amdgcn_readlane(gnext.rb, 0), n1 = __builtin_amdgcn_readlane(gnext.rb, 16);
        dma_records(n0, n1 - n0, bufA + ((gidx + 1) & 1) * SCH_REC, bufT);
      }
    }
    GroupIn gnn;
    fetch_in(group_p0(gidx + 2), gnn);
    SCSUB(3);
    {  // ---- phase 2: the group's four MFMA operand sets (points 4 u + kq).  Column 16 tt + ci of Z: compact target 2 tt + tsub (the
       // targets without the host itself, which no residual of these points observes), element asub; the last eight columns: Hcd, bdSumF
      float zz[4][4], hx[4];
      // three straight stages — the points' words, the sixteen record words, the products — with the scheduler kept from sinking a read
      // next to its use (left alone it waits for sixteen LDS round trips one after the other instead of two).  Every lane reads ONE word
      // per tile column, unconditionally (a load under a condition becomes a branch with its own wait): the record of its target — any
      // record when the point does not observe it, masked afterwards — or, on the special lanes of the last tile column, the point's
      // Hcd / bdSumF word in pt[]
      unsigned invs[4];
      int bases[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int q = 4 * u + kq;
        hx[u] = pt[q][0];
        const float2 ib = *(const float2*)(&pt[q][6]);
        invs[u] = (unsigned)__float_as_int(ib.x); bases[u] = __float_as_int(ib.y);
      }
      __builtin_amdgcn_sched_barrier(0);
      bool oks[4][4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int q = 4 * u + kq;
        const float* rb = curA + bases[u] * 8 + asub;
#pragma unroll
        for (int tt = 0; tt < 4; tt++) {
          const int tp = 2 * tt + tsub, t = tp + (tp >= h ? 1 : 0);       // (tp = 7 is the special block: its lanes take the other address)
          const unsigned k = (invs[u] >> (4 * (t & 7))) & 15u;
          const float* ad = rb + (k & 7u) * 8;
          bool ok = k != 15u;
          if (tt == 3) { ad = tsub ? &pt[q][asub < 4 ? 2 + asub : 1] : ad; ok = tsub ? asub <= 4 : ok; }
          zz[u][tt] = *ad;
          oks[u][tt] = ok;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int tt = 0; tt < 4; tt++) zz[u][tt] = oks[u][tt] ? zz[u][tt] : 0.f;
      if (ACC_MODE_SC == 1) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
          double za[4], zd[4];
#pragma unroll
          for (int tt = 0; tt < 4; tt++) { zd[tt] = (double)zz[u][tt]; za[tt] = (double)hx[u] * zd[tt]; }
#pragma unroll
          for (int a = 0; a < 4; a++) {
#pragma unroll
            for (int b = a; b < 4; b++) acc[sc_ut(a, b)] = __builtin_amdgcn_mfma_f64_16x16x4f64(za[a], zd[b], acc[sc_ut(a, b)], 0, 0, 0);
          }
        }
      } else {
        if (ACC_MODE_SC == 2) {
#pragma unroll
          for (int t = 0; t < SC_NT; t++) grp[t] = (te_f4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          float za[4];
#pragma unroll
          for (int tt = 0; tt < 4; tt++) za[tt] = hx[u] * zz[u][tt];
#pragma unroll
          for (int a = 0; a < 4; a++) {
#pragma unroll
            for (int b = a; b < 4; b++) grp[sc_ut(a, b)] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[a], zz[u][b], grp[sc_ut(a, b)], 0, 0, 0);
          }
        }
        if (ACC_MODE_SC == 2) {
#pragma unroll
          for (int t = 0; t < SC_NT; t++)
#pragma unroll
            for (int v = 0; v < 4; v++) acc[t][v] += (double)grp[t][v];
        }
      }
    }
    gcur = gnext; gnext = gnn;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();          // the next group overwrites pt[]
    SCSUB(4);
  }
  SCS();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (ACC_MODE_SC == 0) {
#pragma unroll
    for (int t = 0; t < SC_NT; t++)
#pragma unroll
      for (int v = 0; v < 4; v++) acc[t][v] = (double)grp[t][v];
  }
  // ---- the host's bins from ONE wave's accumulators (WPH: this wave's; otherwise wave 0's after the tree), rounded to float: d[v] of lane
  // (kq, ci) = D[16 a + 4 kq + v][16 b + ci] of tile (a, b) — four consecutive rows per lane (ACC_MODE_SC 1: the f64 MFMA leaves rows
  // kq + 4 v on a lane; its tiles are turned through 1 KB of LDS into that layout first).  Every (t1, t2) block of 256 bytes is covered by
  // the four stores of its tile; the blocks below the diagonal are float4s of the mirrored tile; a tile ON the diagonal takes its lower
  // triangle from the mirror image too: D(r, c) and D(c, r) differ in the last bit ((w z_r) z_c against (w z_c) z_r), and the stitch relies
  // on accD(h, i, j) == accD(h, j, i)^T exactly.  The blocks of the host's own (absent) target are zeros.
  auto store_bins = [&](float* T /* 16 x 17 floats of LDS of this wave */) {
    const int nf2 = nf * nf;
    float* accD = B.accum + acc_off_D(nf);
    float* accE = B.accum + acc_off_E(nf);
    float* accEB = B.accum + acc_off_EB(nf);
    const int ro = 4 * (kq & 1), cc = ci & 7;                 // row (+ v) and column inside an 8 x 8 block
#pragma unroll
    for (int a = 0; a < 4; a++) {
      const int t1p = 2 * a + (kq >> 1), t1 = t1p + (t1p >= h ? 1 : 0);
#pragma unroll
      for (int b = a; b < 4; b++) {
        const int t2p = 2 * b + (ci >> 3), t2 = t2p + (t2p >= h ? 1 : 0);
        te_f4 d;
        {
          const te_d4 dd = acc[sc_ut(a, b)];
#pragma unroll
          for (int v = 0; v < 4; v++) d[v] = (float)dd[v];
          if (ACC_MODE_SC == 1 || a == b) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int v = 0; v < 4; v++) T[(ACC_MODE_SC == 1 ? kq + 4 * v : 4 * kq + v) * 17 + ci] = d[v];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (ACC_MODE_SC == 1) {
#pragma unroll
              for (int v = 0; v < 4; v++) d[v] = T[(4 * kq + v) * 17 + ci];
            }
            if (a == b) {
#pragma unroll
              for (int v = 0; v < 4; v++) { const float m = T[ci * 17 + 4 * kq + v]; d[v] = (4 * kq + v > ci) ? m : d[v]; }
            }
          }
        }
        if (t1p < 7 && t1 < nf) {
          if (t2p < 7) {
            if (t2 < nf) {
              float* blk = accD + (size_t)(h + t1 * nf + t2 * nf2) * 64 + cc;
#pragma unroll
              for (int v = 0; v < 4; v++) blk[(ro + v) * 8] = d[v];
              if (a != b) *(float4*)(accD + (size_t)(h + t2 * nf + t1 * nf2) * 64 + cc * 8 + ro) = make_float4(d[0], d[1], d[2], d[3]);   // the mirror image
            }
          } else if (cc < 4) {                                 // the special block's columns: Hcd ...
#pragma unroll
            for (int v = 0; v < 4; v++) accE[(size_t)(h + t1 * nf) * 32 + (ro + v) * 4 + cc] = d[v];
          } else if (cc == 4) {                                // ... and bdSumF
#pragma unroll
            for (int v = 0; v < 4; v++) accEB[(size_t)(h + t1 * nf) * 8 + ro + v] = d[v];
          }
        } else if (b == 3 && t1p == 7 && t2p == 7 && ro == 0) {   // special x special: Hcc (16) and bc (4) of this host; the fold adds the hosts
          float* hp = B.sc_part + (size_t)h * 20;
          if (cc < 4) {
#pragma unroll
            for (int v = 0; v < 4; v++) hp[v * 4 + cc] = d[v];
          } else if (cc == 4) {
#pragma unroll
            for (int v = 0; v < 4; v++) hp[16 + v] = d[v];
          }
        }
      }
    }
    // the host's own target: 2 nf - 1 blocks of accD, one of accE and accEB
    for (int k = 0; k < 2 * nf; k++) {
      const int t = k >> 1;
      if ((k & 1) && t == h) continue;
      const int t1 = (k & 1) ? t : h, t2 = (k & 1) ? h : t;
      accD[(size_t)(h + t1 * nf + t2 * nf2) * 64 + lane] = 0.f;
    }
    if (lane < 32) accE[(size_t)(h + h * nf) * 32 + lane] = 0.f;
    else if (lane < 40) accEB[(size_t)(h + h * nf) * 8 + lane - 32] = 0.f;
  };
  if (WPH) {
    store_bins(bufA);
#ifdef SDSO_SC_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SCS();
    if ((h == 0 || h == 5) && (blockIdx.y == 0 || blockIdx.y == 100) && lane == 0)
      printf("sc_host wave-per-host (%d,%d) ticks: prologue %llu  group loop %llu  bins %llu | total %llu  (points %d)  loop: wait %llu  r_cj %llu  phase 1 %llu  dma + inputs %llu  phase 2 %llu\n",
             h, blockIdx.y, st[1] - st[0], st[2] - st[1], st[3] - st[2], st[3] - st[0], pe - pb, sub[0], sub[1], sub[2], sub[3], sub[4]);
#endif
    return;
  }
  __syncthreads();                            // the tiles lie over the waves' stages
  // ---- fixed-order tree over the four waves: (3 -> 1, 2 -> 0), then (1 -> 0); wave 0 writes the bins
  double (*tiles)[SC_NT * 256] = reinterpret_cast<double (*)[SC_NT * 256]>(stage_all);
  auto put = [&](double* dst) {
#pragma unroll
    for (int t = 0; t < SC_NT; t++)
#pragma unroll
      for (int v = 0; v < 4; v++) dst[(t * 4 + v) * 64 + lane] = acc[t][v];
  };
  auto add = [&](const double* src) {
#pragma unroll
    for (int t = 0; t < SC_NT; t++)
#pragma unroll
      for (int v = 0; v < 4; v++) acc[t][v] += src[(t * 4 + v) * 64 + lane];
  };
  if (wv >= 2) put(tiles[wv - 2]);
  __syncthreads();
  if (wv < 2) add(tiles[wv]);
  __syncthreads();
  if (wv == 1) put(tiles[1]);                 // (1 + 3)
  __syncthreads();
  SCS();
  if (wv == 0) {
    // (0 + 2) + (1 + 3): the order the tree always had
#pragma unroll
    for (int t = 0; t < SC_NT; t++)
#pragma unroll
      for (int v = 0; v < 4; v++) acc[t][v] += tiles[1][(t * 4 + v) * 64 + lane];
    store_bins((float*)tiles[0]);
  }
#ifdef SDSO_SC_STAMPS
  if (wv == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SCS();
    if ((blockIdx.x == 0 || blockIdx.x == 5) && (blockIdx.y == 0 || blockIdx.y == 100) && lane == 0)
      printf("sc_host (%d,%d) ticks: prologue %llu  group loop %llu  wave tree %llu  bins %llu | total %llu  (points %d)  loop: wait %llu  r_cj %llu  phase 1 %llu  dma + inputs %llu  phase 2 %llu\n",
             blockIdx.x, blockIdx.y, st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[4] - st[0], pe - pb, sub[0], sub[1], sub[2], sub[3], sub[4]);
  }
#endif
}

}  // namespace sdso
