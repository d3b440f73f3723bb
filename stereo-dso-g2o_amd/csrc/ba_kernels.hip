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
template <bool STORE, int KEEP, bool TILED>
__device__ __forceinline__ double linearize_coop(const BaDev& B, int i, bool live, int h, int t, float* jl, int& ns_out, float* rs, float* wstage) {
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
  const uint8_t st = B.r_state[i];
  if (st == 1) { B.r_newState[i] = 1; ret = (double)B.r_energy[i]; dead = true; break; }
  const int pt = B.r_point[i];
  const float* KRKi = pre; const float* Kt = pre + 9; const float* R0 = pre + 12; const float* t0 = pre + 21;
  const float4 g = B.p_geo[pt];
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
// ------------------------------------------------------------------ linearize with the taps brought in by LDS-DMA
// Third gather scheme (GATHER = 2, 4x2-tiled images): the 32 taps of a residual are fetched by `global_load_lds_dwordx4`
// (one 16-byte pixel per lane, written by the memory pipeline straight into LDS — no destination VGPRs, nothing to wait for at
// issue).  A ROUND is one pattern pixel of all 64 residuals of the wave: 4 instructions, lane = (residual % 16, corner) so that the
// four corners of a bilinear sample sit on neighbouring lanes (one or two 128-byte lines per quad, as in the cooperative gather),
// 4 KiB of LDS.  DM_DEPTH rounds are in flight per wave; while they travel the wave interpolates and does the per-pixel arithmetic
// of the round that has landed ON THE RESIDUAL'S OWN LANE (four ds_read_b128 of its 64 contiguous bytes, then exactly
// interp33's expression) and stores that pixel's Jacobian group: no coordinate / sample stage, no quad shuffles, and a third
// fewer vector instructions than the cooperative scheme (which spends 32 x ~40 instructions per wave on redundant weights).
// The compiler cannot track the LDS dependency of the DMA finer than `s_waitcnt vmcnt(0)`, so the ring is read with inline
// assembly behind a counted wait: vector-memory operations complete in issue order, hence `vmcnt(4 x rounds issued later)` is a
// sufficient wait for round p whatever stores were issued in between (they only make the wait earlier than necessary).
constexpr int DM_DEPTH = 3;
constexpr int DM_ROUND_FLOATS = 4 * 64 * 4;
constexpr int DM_WAVE_FLOATS = DM_DEPTH * DM_ROUND_FLOATS;     // 12 KiB per wave
template <int N>
__device__ __forceinline__ void dm_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// one round: xy = (int)Ku | (int)Kv << 16 of THIS lane's residual for the round's pattern pixel (0 for a dead lane: pixel (0,0))
__device__ __forceinline__ void dm_issue_round(const float4* __restrict__ img, int Tw, int xy, float* slot) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int v = __shfl(xy, 16 * q + (lane >> 2), 64);
    const int x = (v & 0xffff) + (lane & 1), y = (v >> 16) + ((lane >> 1) & 1);
    const float4* a = img + tiled_index(x, y, Tw);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a, (__attribute__((address_space(3))) void*)(slot + q * 256), 16, 0, 0);
  }
}
// the four corner pixels of this lane's sample: 64 contiguous bytes at slot + 16 * lane floats (instruction lane/16, its lanes 4*(lane%16)..+3)
__device__ __forceinline__ void dm_read_taps(const float* slot_lane, te_f4& p00, te_f4& p10, te_f4& p01, te_f4& p11) {
  const unsigned addr = (unsigned)(unsigned long)(const __attribute__((address_space(3))) float*)slot_lane;
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(p00), "=&v"(p10), "=&v"(p01), "=&v"(p11)
               : "v"(addr)
               : "memory");
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
  float* rec = B.r_rec + ((size_t)B.r_point[i] * B.nf + B.r_target[i]) * 16;
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
    *(float4*)(rec) = make_float4(out[0], out[1], out[2], out[3]);
    *(float4*)(rec + 4) = make_float4(out[4], out[5], out[6], out[7]);
  }
  else {
    // the record of a residual that is not active holds zeros (like the fused kernel's): the Schur kernel adds / multiplies records without
    // looking at their flags
    *(float4*)(rec) = make_float4(0.f, 0.f, 0.f, 0.f);
    *(float4*)(rec + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    *(float4*)(rec + 8) = make_float4(0.f, 0.f, 0.f, 0.f);
    rec[12] = 0.f; rec[13] = 0.f;
  }
  B.r_act[i] = act;
  rec[RR_FLAGS] = (float)act;
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
  B.r_rec[((size_t)pt * B.nf + B.r_target[i]) * 16 + RR_FLAGS] = 3.f;  // active | linearized
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
// upload: state_NewState = OUTLIER, state_NewEnergyWithOutlier = -1 (Residuals.cpp:40-52), and the target of every existing
// (point, target) slot of the dense per-point records
__global__ __launch_bounds__(BA_BLOCK) void k_ba_init_res(const BaDev* __restrict__ wins) {
  const BaDev& B = wins[blockIdx.y];
  const int i = blockIdx.x * BA_BLOCK + threadIdx.x;
  if (i >= B.nr) return;
  B.r_newState[i] = 2;
  B.r_newEnergyWO[i] = -1.f;
  const int t = B.r_target[i];
  B.r_rec[((size_t)B.r_point[i] * B.nf + t) * 16 + RR_TARGET] = (float)t;
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
// lane-contiguous writes and for the (row = lane%16, k = lane/16) reads) and 2 x 16 v_mfma_f32_16x16x4_f32
// (full fp32) do the reduction.  Only the ORDER of the cross-residual float sums differs from the CPU path.
constexpr int TE_STRIDE = 68;
constexpr int TE_ROWS = 26;
constexpr int TE_WAVE_FLOATS = TE_ROWS * TE_STRIDE;
constexpr int TE_LDS_FLOATS = (BA_BLOCK / 64) * TE_WAVE_FLOATS;

template <bool LDS_ONLY = false>
__device__ __forceinline__ void top_emit(const BaDev& B, const float* x, const float* y, float a, float b, float c, float TR00, float TR10, float TR01,
                                         float TR11, float TR02, float TR12, const float* br, bool on, float* stage) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* S = stage + wv * TE_WAVE_FLOATS;
  const int m = lane & 15, kq = lane >> 4;
  const int mu = m < 10 ? m : 9;
  te_f4 acc = {0.f, 0.f, 0.f, 0.f};
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
#pragma unroll
    for (int s4 = 0; s4 < 16; s4++) {
      float av = S[mu * TE_STRIDE + 4 * s4 + kq];
      av = m < 10 ? av : one;
      const float bv = S[(10 + m) * TE_STRIDE + 4 * s4 + kq];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    }
    wg_barrier<LDS_ONLY>();
  }
  // D[i][j]: i = 4*(lane/16) + v, j = lane%16
  float* R = stage;
#pragma unroll
  for (int v = 0; v < 4; v++) R[wv * 256 + (4 * kq + v) * 16 + m] = acc[v];
  if (lane == 0) R[(BA_BLOCK / 64) * 256 + wv] = (float)__popcll(onmask);
  wg_barrier<LDS_ONLY>();
  if (threadIdx.x < 92) {
    const int t = threadIdx.x;
    int off;
    if (t < 55) { int r = 0, rem = t; while (rem >= 10 - r) { rem -= 10 - r; r++; } off = r * 16 + r + rem; }
    else if (t < 85) off = ((t - 55) / 3) * 16 + 10 + (t - 55) % 3;
    else if (t < 88) off = 10 * 16 + 13 + (t - 85);
    else if (t < 91) off = 11 * 16 + 13 + (t - 88);
    else off = (BA_BLOCK / 64) * 256;   // count slots follow the tiles
    float s = R[off];
#pragma unroll
    for (int w = 1; w < BA_BLOCK / 64; w++) s += R[off + (t < 91 ? w * 256 : w)];
    B.top_part[(size_t)blockIdx.x * 92 + t] = s;
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
    float* rec = B.r_rec + ((size_t)pt * B.nf + ch.x / B.nf) * 16;   // pair = host + target*nf
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
// GATHER: 0 = every lane fetches the 32 taps of its own residual (A/B: SDSO_BA_DIRECT_TAPS); 1 = cooperative quad gather
// (linearize_coop), the library's default.  k_ba_lin_dma below (LDS-DMA rounds, tiled images) is a third, equally fast variant.
template <bool MATERIALIZE, bool TILED, int GATHER = 1>
__global__ __launch_bounds__(BA_BLOCK, (MATERIALIZE ? 3 : 4)) void k_ba_lin_fused(const BaDev* __restrict__ wins) {
  constexpr bool COOP = GATHER == 1;
  // by-value copy first: every pointer of the descriptor is read before the kernel's first store, so the
  // compiler can prove them global (global_load / s_load instead of flat_load) and keep them in SGPRs
  const BaDev B = wins[blockIdx.y];
  if (ba_finished_lin(B)) return;
  if ((int)blockIdx.x >= B.nchunks) return;
  const int4 ch = B.chunks[blockIdx.x];
  const int pair = __builtin_amdgcn_readfirstlane(ch.x);   // one (host,target) per workgroup: precalc, image, thresholds are wave-uniform
  const int i = ch.y + threadIdx.x;
  constexpr int STAGE_FLOATS = (BA_BLOCK / 64) * (COOP ? CG_WAVE_FLOATS : 0);
  constexpr int RED_FLOATS = TE_LDS_FLOATS > STAGE_FLOATS ? TE_LDS_FLOATS : STAGE_FLOATS;
  __shared__ float red[RED_FLOATS];    // the gather stage of the linearisation, then the MFMA panels of the reduction
  double* const lds = (double*)red;    // (the energy reduction runs between the two uses; 40 KB in all = four workgroups per CU)
  float x[10], y[10], a = 0, b = 0, c = 0;
  float TR00 = 0, TR10 = 0, TR01 = 0, TR11 = 0, TR02 = 0, TR12 = 0;
  float br[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 10; k++) { x[k] = 0; y[k] = 0; }
  bool on = false;
  double e = 0;
  const bool live = (int)threadIdx.x < ch.z && !B.r_lin[i];
  float jl[76];
  float rs5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  int ns = 1;
  uint8_t st = 1;
  if (COOP) {
    if (live) st = B.r_state[i];
    e = linearize_coop<MATERIALIZE, 2, TILED>(B, i, live, pair % B.nf, pair / B.nf, jl, ns, rs5, red + (threadIdx.x >> 6) * CG_WAVE_FLOATS);
    __syncthreads();   // the reduction below reuses the stage of all waves
  }
  if (live) {
    if (GATHER == 0) {
      st = B.r_state[i];
      e = linearize_one<MATERIALIZE, 2, TILED>(B, i, pair % B.nf, pair / B.nf, jl, ns, rs5, B.jfix == 1);
    }
    const int pt = B.r_point[i];
    float* rec = B.r_rec + ((size_t)pt * B.nf + pair / B.nf) * 16;
    // the 64-byte record of this (point, target): written once, whole, at the end (four 16-byte stores of one line)
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
    if (st != 1) {   // (a sticky-OOB residual keeps the record its last applyRes wrote: flags 0)
      *(float4*)(rec) = make_float4(o8[0], o8[1], o8[2], o8[3]);
      *(float4*)(rec + 4) = make_float4(o8[4], o8[5], o8[6], o8[7]);
      *(float4*)(rec + 8) = make_float4(rbd, rhdd, rhcd[0], rhcd[1]);
      *(float4*)(rec + 12) = make_float4(rhcd[2], rhcd[3], (float)act, (float)(pair / B.nf));
    }
  }
  e = block_sum_d(e, lds);
  if (threadIdx.x == 0) B.e_part[blockIdx.x] = e;
  __syncthreads();
  top_emit(B, x, y, a, b, c, TR00, TR10, TR01, TR11, TR02, TR12, br, on, red);
}

// ------------------------------------------------------------------ fused linearize + applyRes + accumulateAF, taps by LDS-DMA
// A/B variant of the fused kernel (SDSO_BA_GATHER=2, 4x2-tiled level-0 images; measured equal to the cooperative gather of
// k_ba_lin_fused<.,.,1>, which stays the default).  Same work per workgroup as k_ba_lin_fused — one chunk of <= 256
// residuals of ONE (host,target) pair — organised around what the profile of its predecessor showed: a workgroup lived ~60 us for
// ~3.5 k instructions per wave, i.e. it sat in a chain of a dozen DEPENDENT memory round trips at three waves per SIMD.
//   * prologue: every per-residual input is fetched by two batches of independent, unconditional loads (indices / states, then the
//     point), lanes past the end of the chunk read the chunk's first residual; the OOB / sticky-state decisions become flags;
//   * taps: LDS-DMA rounds (see dm_issue_round above), DM_DEPTH pixel rounds in flight per wave, interpolation and the per-pixel
//     arithmetic on the residual's own lane while the next rounds travel; `s_waitcnt vmcnt(N)` with N = exactly the younger
//     vector-memory operations (later rounds + the record stores issued since), given through the builtin so that the compiler's
//     own scoreboard knows when the last DMA has landed and does not drain the record stores before the reduction;
//   * the record stores are `global_store ... nt` straight from the registers of the pixel just finished (one live quad);
//   * workgroup barriers wait for LDS traffic only (wg_barrier<true>): the Jacobian records keep streaming out underneath the
//     energy sum and the MFMA reduction of the 91 top sums.
// Per-residual arithmetic is expression for expression that of linearize_one: every output stays bit-identical.
template <int VM>
__device__ __forceinline__ void dm_wait_builtin() {
  static_assert(VM >= 0 && VM < 64, "vmcnt is a 6-bit field");
  __builtin_amdgcn_s_waitcnt(0x0F70 | (VM & 15) | ((VM >> 4) << 14));   // vmcnt(VM), expcnt / lgkmcnt untouched
}
template <bool MATERIALIZE>
__global__ __launch_bounds__(BA_BLOCK, 3) void k_ba_lin_dma(const BaDev* __restrict__ wins) {
  // explicit address spaces: global_* instead of flat_* (flat operations complete out of order, so every use of a flat load
  // would force `s_waitcnt vmcnt(0)` and drain the record stores), s_load for the wave-uniform tables
#define GP(T, p) ((__attribute__((address_space(1))) T*)(p))
#define CP(T, p) ((const __attribute__((address_space(4))) T*)(p))
  const BaDev B = wins[blockIdx.y];
  if (ba_finished_lin(B)) return;
  if ((int)blockIdx.x >= B.nchunks) return;
  typedef int te_i4 __attribute__((ext_vector_type(4)));
  const te_i4 ch = CP(te_i4, B.chunks)[blockIdx.x];          // {pair, start, count, -}
  const int pair = __builtin_amdgcn_readfirstlane(ch.x);
  const int nf = B.nf, h = pair % nf, t = pair / nf;
  constexpr int STAGE_FLOATS = (BA_BLOCK / 64) * DM_WAVE_FLOATS;
  constexpr int RED_FLOATS = TE_LDS_FLOATS > STAGE_FLOATS ? TE_LDS_FLOATS : STAGE_FLOATS;
  __shared__ float red[RED_FLOATS];     // the DMA rings of the four waves, then the MFMA panels of the reduction
  double* const lds = (double*)red;
  float* const ring = red + (threadIdx.x >> 6) * DM_WAVE_FLOATS;
  const bool inb = (int)threadIdx.x < ch.z;
  const int i = ch.y + (inb ? (int)threadIdx.x : 0);
  // ---- batch 1: indices and states (independent loads)
  const uint8_t lin = GP(const uint8_t, B.r_lin)[i], st = GP(const uint8_t, B.r_state)[i], jsel = GP(const uint8_t, B.r_jsel)[i];
  const int pt = GP(const int, B.r_point)[i];
  const float e_old = GP(const float, B.r_energy)[i], ne_old = GP(const float, B.r_newEnergy)[i];
  const __attribute__((address_space(4))) float* pre = CP(float, B.t_precalc) + (size_t)(h * nf + t) * 27;   // wave-uniform: SGPRs
  const float affLL0 = pre[24], affLL1 = pre[25], b0 = pre[26];
  const float4* __restrict__ dIl = (const float4*)CP(unsigned long long, B.t_img)[t];
  const float th = fmaxf(GP(const float, B.t_frameTH)[h], GP(const float, B.t_frameTH)[t]);
  // ---- batch 2: the point
  const te_f4 g = GP(const te_f4, B.p_geo)[pt];
  const te_f4 c0 = GP(const te_f4, B.p_color)[2 * (size_t)pt], c1 = GP(const te_f4, B.p_color)[2 * (size_t)pt + 1];
  const te_f4 w0 = GP(const te_f4, B.p_weights)[2 * (size_t)pt], w1 = GP(const te_f4, B.p_weights)[2 * (size_t)pt + 1];
  const bool live = inb && !lin;                     // linearizeAll skips linearized residuals (FullSystemOptimize.cpp:52-87)
  const float color[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
  const float weights[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
  const float pu = g.x, pv = g.y, idepth_scaled = g.z, idepth_zero_scaled = g.w;
  const float fxl = B.fxl, fyl = B.fyl, cxl = B.cxl, cyl = B.cyl, fxli = B.fxli, fyli = B.fyli;
  const auto KRKi = pre; const auto Kt = pre + 9; const auto R0 = pre + 12; const auto t0 = pre + 21;
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
  const float u = ptp[0] * drescale;
  const float v = ptp[1] * drescale;
  const float Ku0 = u * fxl + cxl;
  const float Kv0 = v * fyl + cyl;
  // a residual is sampled when it is live, not sticky-OOB (Residuals.cpp:88-91), its centre projects into the image (:96-100) and
  // so do all 8 pattern pixels (:215-225)
  bool comp = live && st != 1 && (drescale > 0) && (Ku0 > 1.1f && Kv0 > 1.1f && Ku0 < B.wM3 && Kv0 < B.hM3);
  float Kus[8], Kvs[8];
#pragma unroll
  for (int idx = 0; idx < 8; idx++) {
    const float up = pu + c_pattern[idx][0], vp = pv + c_pattern[idx][1];
    float q[3];
#pragma unroll
    for (int r = 0; r < 3; r++) q[r] = ((KRKi[r * 3 + 0] * up + KRKi[r * 3 + 1] * vp) + KRKi[r * 3 + 2]) + Kt[r] * idepth_scaled;
    Kus[idx] = q[0] / q[2];
    Kvs[idx] = q[1] / q[2];
    if (!(Kus[idx] > 1.1f && Kvs[idx] > 1.1f && Kus[idx] < B.wM3 && Kvs[idx] < B.hM3)) comp = false;
  }
  __attribute__((address_space(1))) float* const J = MATERIALIZE ? (__attribute__((address_space(1))) float*)((jsel != 0) != (B.jfix != 0) ? B.J[0] : B.J[1]) : nullptr;   // (jfix: in place, see ba_kernels.h)
  const int S = B.nrp;
#define JSTORE(grp, a, b, c, d)                                                                                                      \
  do {                                                                                                                               \
    if (MATERIALIZE) __builtin_nontemporal_store((te_f4){a, b, c, d}, (__attribute__((address_space(1))) te_f4*)(J + j_off(S, i, (grp)))); \
  } while (0)
  float JIdxJIdx_00 = 0, JIdxJIdx_11 = 0, JIdxJIdx_10 = 0;
  float JabJIdx_00 = 0, JabJIdx_01 = 0, JabJIdx_10 = 0, JabJIdx_11 = 0;
  float JabJab_00 = 0, JabJab_01 = 0, JabJab_11 = 0;
  float wJI2_sum = 0, energyLeft = 0;
  float JI_r0 = 0, JI_r1 = 0, Jab_r0 = 0, Jab_r1 = 0, rr = 0;      // addPoint<0>'s per-pixel sums with resApprox = resF (AccumulatedTopHessian.cpp:119-128)
  float jab1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  bool nonfinite = false;
  if (__ballot(comp) != 0ull) {      // (wave-uniform)
    int xy[8];
#pragma unroll
    for (int k = 0; k < 8; k++) xy[k] = comp ? ((int)Kus[k] | ((int)Kvs[k] << 16)) : 0;
    const int Tw = B.tiledT;
    const float* my = ring + 16 * (threadIdx.x & 63);
#pragma unroll
    for (int p = 0; p < DM_DEPTH; p++) dm_issue_round(dIl, Tw, xy[p], ring + p * DM_ROUND_FLOATS);
#pragma unroll
    for (int idx = 0; idx < 8; idx++) {
      // vector-memory operations younger than round idx: the rounds issued after it, and (when the records are materialised) the
      // record stores issued since it was issued — exec is non-zero here (some lane samples), so each of those was one instruction
      constexpr int D1 = DM_DEPTH - 1;
      const int later = (7 - idx < D1 ? 7 - idx : D1);
      const int stores = MATERIALIZE ? (idx < DM_DEPTH ? idx : DM_DEPTH) : 0;
      switch (4 * later + stores) {
#define DMW(N) case N: dm_wait_builtin<N>(); break;
        DMW(0) DMW(1) DMW(2) DMW(3) DMW(4) DMW(5) DMW(6) DMW(7) DMW(8) DMW(9) DMW(10) DMW(11) DMW(12) DMW(13) DMW(14) DMW(15)
#undef DMW
        default: dm_wait_builtin<0>(); break;
      }
      te_f4 p00, p10, p01, p11;
      dm_read_taps(my + (idx % DM_DEPTH) * DM_ROUND_FLOATS, p00, p10, p01, p11);
      if (idx + DM_DEPTH < 8) dm_issue_round(dIl, Tw, xy[idx + DM_DEPTH], ring + (idx % DM_DEPTH) * DM_ROUND_FLOATS);   // the slot has just been read
      // getInterpolatedElement33 (same expression and order as interp33)
      const float x = Kus[idx], y = Kvs[idx];
      const int ix = (int)x;
      const int iy = (int)y;
      const float dx = x - ix;
      const float dy = y - iy;
      const float dxdy = dx * dy;
      const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
      float3 hit;
      hit.x = w11 * p11.x + w01 * p01.x + w10 * p10.x + w00 * p00.x;
      hit.y = w11 * p11.y + w01 * p01.y + w10 * p10.y + w00 * p00.y;
      hit.z = w11 * p11.z + w01 * p01.z + w10 * p10.z + w00 * p00.z;
      if (!isfinite(hit.x)) nonfinite = true;
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
      const float ra = residual * hw, ja0 = B.affA_fixed ? 0.f : drdA * hw;
      jab1[idx] = B.affB_fixed ? 0.f : hw;
      if (comp) JSTORE(6 + idx, ra, hit.y, hit.z, ja0);        // resF, JIdx[0], JIdx[1], JabF[0]
      JI_r0 += ra * hit.y; JI_r1 += ra * hit.z;
      Jab_r0 += ra * ja0; Jab_r1 += ra * jab1[idx];
      rr += ra * ra;
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
  }
  const bool done = comp && !nonfinite;             // linearize ran to its end (Residuals.cpp:302-336)
  // ---- geometric Jacobians at the FEJ point (Residuals.cpp:135-185) and the rest of the record
  float xv[10], yv[10], jdd0 = 0.f, jdd1 = 0.f;
#pragma unroll
  for (int k = 0; k < 10; k++) { xv[k] = 0.f; yv[k] = 0.f; }
  int ns = 1;
  if (done) {
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
#pragma unroll
    for (int k = 0; k < 4; k++) { xv[k] = d_C_x[k]; yv[k] = d_C_y[k]; }
    xv[4] = new_idepth * fxl; xv[5] = 0; xv[6] = -new_idepth * u * fxl; xv[7] = -u * v * fxl; xv[8] = (1 + u * u) * fxl; xv[9] = -v * fxl;
    yv[4] = 0; yv[5] = new_idepth * fyl; yv[6] = -new_idepth * v * fyl; yv[7] = -(1 + v * v) * fyl; yv[8] = u * v * fyl; yv[9] = u * fyl;
    jdd0 = d_d_x; jdd1 = d_d_y;
    JSTORE(0, xv[4], xv[5], xv[6], xv[7]);                    // Jpdxi[0][0..3]
    JSTORE(1, xv[8], xv[9], yv[4], yv[5]);                    // Jpdxi[0][4..5], Jpdxi[1][0..1]
    JSTORE(2, yv[6], yv[7], yv[8], yv[9]);                    // Jpdxi[1][2..5]
    JSTORE(3, xv[0], xv[1], xv[2], xv[3]);
    JSTORE(4, yv[0], yv[1], yv[2], yv[3]);
    JSTORE(5, jdd0, jdd1, 0.f, 0.f);
    JSTORE(14, jab1[0], jab1[1], jab1[2], jab1[3]);
    JSTORE(15, jab1[4], jab1[5], jab1[6], jab1[7]);
    JSTORE(16, JIdxJIdx_00, JIdxJIdx_10, JIdxJIdx_10, JIdxJIdx_11);
    JSTORE(17, JabJIdx_00, JabJIdx_01, JabJIdx_10, JabJIdx_11);
    JSTORE(18, JabJab_00, JabJab_01, JabJab_01, JabJab_11);
    ns = (energyLeft > th || wJI2_sum < 2) ? 2 : 0;           // OUTLIER / IN (Residuals.cpp:325-332)
  }
#undef JSTORE
  const float e_new = ns == 2 ? th : energyLeft;              // state_NewEnergy when linearize ran through
  double e = 0;
  bool on = false;
  float a = 0, b = 0, c = 0, TR00 = 0, TR10 = 0, TR01 = 0, TR11 = 0, TR02 = 0, TR12 = 0;
  float br[6] = {0, 0, 0, 0, 0, 0};
  if (live) {
    GP(float, B.r_newEnergyWO)[i] = done ? energyLeft : -1.f;
    GP(uint8_t, B.r_newState)[i] = (uint8_t)ns;
    if (done) GP(float, B.r_newEnergy)[i] = e_new;
    e = done ? (double)e_new : (double)e_old;
    if (st != 1) {                                            // applyRes(true): OOB is sticky (Residuals.cpp:367-385)
      __attribute__((address_space(1))) te_f4* rec = GP(te_f4, B.r_rec) + ((size_t)pt * nf + t) * 4;
      float o8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, rbd = 0.f, rhdd = 0.f, rhcd[4] = {0.f, 0.f, 0.f, 0.f};
      const uint8_t act = ns == 0;
      if (act) {                                              // takeDataF (EnergyFunctionalStructs.cpp:37-51), then addPoint<0>
        if (MATERIALIZE && !B.jfix) GP(uint8_t, B.r_jsel)[i] = jsel ^ 1;
        const float v0 = JIdxJIdx_00 * jdd0 + JIdxJIdx_10 * jdd1;
        const float v1 = JIdxJIdx_10 * jdd0 + JIdxJIdx_11 * jdd1;
#pragma unroll
        for (int k = 0; k < 6; k++) o8[k] = xv[4 + k] * v0 + yv[4 + k] * v1;
        o8[6] = JabJIdx_00 * jdd0 + JabJIdx_01 * jdd1;
        o8[7] = JabJIdx_10 * jdd0 + JabJIdx_11 * jdd1;
        on = true;
        a = JIdxJIdx_00; b = JIdxJIdx_10; c = JIdxJIdx_11;
        TR00 = JabJIdx_00; TR10 = JabJIdx_01; TR01 = JabJIdx_10; TR11 = JabJIdx_11;
        TR02 = JI_r0; TR12 = JI_r1;
        br[0] = JabJab_00; br[1] = JabJab_01; br[2] = Jab_r0; br[3] = JabJab_11; br[4] = Jab_r1; br[5] = rr;
        const float q0 = a * jdd0 + b * jdd1;
        const float q1 = JIdxJIdx_10 * jdd0 + c * jdd1;
        rbd = JI_r0 * jdd0 + JI_r1 * jdd1;
        rhdd = q0 * jdd0 + q1 * jdd1;
#pragma unroll
        for (int k = 0; k < 4; k++) rhcd[k] = xv[k] * q0 + yv[k] * q1;
      }
      GP(uint8_t, B.r_act)[i] = act;
      GP(uint8_t, B.r_state)[i] = (uint8_t)ns;
      GP(float, B.r_energy)[i] = done ? e_new : ne_old;        // state_energy = state_NewEnergy
      rec[0] = (te_f4){o8[0], o8[1], o8[2], o8[3]};
      rec[1] = (te_f4){o8[4], o8[5], o8[6], o8[7]};
      rec[2] = (te_f4){rbd, rhdd, rhcd[0], rhcd[1]};
      rec[3] = (te_f4){rhcd[2], rhcd[3], (float)act, (float)t};
    }
  }
  if (!on) {
#pragma unroll
    for (int k = 0; k < 10; k++) { xv[k] = 0.f; yv[k] = 0.f; }
  }
  wg_barrier<true>();     // every wave is done with its ring: the reductions reuse the space
  e = block_sum_d<true>(e, lds);
  if (threadIdx.x == 0) GP(double, B.e_part)[blockIdx.x] = e;
  wg_barrier<true>();
#undef GP
#undef CP
  top_emit<true>(B, xv, yv, a, b, c, TR00, TR10, TR01, TR11, TR02, TR12, br, on, red);
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
    float s = 0;
    for (int ck = cb; ck < ce; ck++) s += B.top_part[(size_t)ck * 92 + tid];
    out[tid] = s;
  }
  if (pair == 0 && tid == 127) {
    float s = 0;
    for (int ck = 0; ck < B.nchunks; ck++) s += B.top_part[(size_t)ck * 92 + 91];
    B.accum[acc_off_nres(B.nf) + which] = s;
  }
}
__device__ __forceinline__ void zero_topL_body(const BaDev& B, int pair, int tid) {
  if (pair >= B.nf * B.nf) return;
  if (tid < 91) B.accum[acc_off_topL(B.nf) + (size_t)pair * 91 + tid] = 0.f;
  if (pair == 0 && tid == 127) B.accum[acc_off_nres(B.nf) + 1] = 0.f;
}
__global__ __launch_bounds__(128) void k_ba_fold_top(const BaDev* __restrict__ wins, int which) { if (ba_finished(wins[blockIdx.y])) return; fold_top_body(wins[blockIdx.y], blockIdx.x, which, threadIdx.x); }
__global__ __launch_bounds__(128) void k_ba_zero_topL(const BaDev* __restrict__ wins) { if (ba_finished(wins[blockIdx.y])) return; zero_topL_body(wins[blockIdx.y], blockIdx.x, threadIdx.x); }

// ------------------------------------------------------------------ per-point terms of the Schur accumulation
// The first half of AccumulatedSCHessianSSE::addPoint (AccumulatedSCHessian.cpp:34-71) for ONE point on ONE lane, together with the
// per-point sums addPoint<mode> of the top accumulator leaves in the EFPoint (AccumulatedTopHessian.cpp:160-192): the residuals are
// visited in EFPoint::residualsAll order (`ord`: BaDev::p_order), exactly the additions of the reference, so Hdd / bd / Hcd, HdiF and
// bdSumF are bit-identical to the CPU path whatever dropResidual did to that order.  recs: the point's dense [target] records (LDS or
// global).  Also keeps PointHessian::idepth_hessian and the `maxRelBaseline = 0` of :44-48 (p_track).
struct ScPointTerms {
  float Hdd_A, bd_A, Hdd_L, bd_L, HcdA[4], HcdL[4];   // p->{Hdd,bd,Hcd}_acc{A,L}F
  float HdiF, bdSumF, Hcd[4];                          // what the cross-point sums use (zeros when the point has no active residual)
  int mbits;                                           // bit t: the residual to target t is present and active
};
// PLAIN: no marginalisation pass, no point filter, no linearized residual in the launch (every GN iteration of the reference's live flow): all
// active residuals go to the A sums, and — the record of a residual that is not active holds zeros (k_ba_apply, the fused kernels) — the
// records are added as they are: x + 0 is exact, so the sums are the reference's to the bit without a select per term (the kernel is bound
// by instruction issue: this loop on 16 of 64 lanes was half of its vector instructions).
template <bool PLAIN = false>
__device__ __forceinline__ void sc_point_terms(const BaDev& B, int p, const float* recs, unsigned ord, float onf, float prior, float delta,
                                               int shiftPriorToZero, int margMode, ScPointTerms& o) {
  float Hdd_A = 0, bd_A = 0, Hdd_L = 0, bd_L = 0, HcdA[4] = {0, 0, 0, 0}, HcdL[4] = {0, 0, 0, 0};
  int ngood = 0, mbits = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const unsigned t = (ord >> (4 * k)) & 15u;
    if (t == 15u) break;
    const float4 q0 = *(const float4*)(recs + t * 16 + 8);    // bd, Hdd, Hcd0, Hcd1
    const float4 q1 = *(const float4*)(recs + t * 16 + 12);   // Hcd2, Hcd3, flags, target
    const int fl = (int)q1.z;
    if (PLAIN) {
      const int act = fl & 1;
      ngood += act; mbits |= act << t;
      bd_A += q0.x; Hdd_A += q0.y;
      HcdA[0] += q0.z; HcdA[1] += q0.w; HcdA[2] += q1.x; HcdA[3] += q1.y;
      continue;
    }
    const bool m = (fl & 1) != 0 && onf != 0.f;                // residual active (and the point taking part)
    const bool mA = m && !(fl & 2) && !margMode, mL = m && !mA;   // mode 0 vs mode 1 / 2 sums (AccumulatedTopHessian.cpp:54-71)
    const float rh[4] = {q0.z, q0.w, q1.x, q1.y};
    if (m) { ngood++; mbits |= 1 << t; }
    // x + 0 is exact: masked-off residuals leave the sums untouched
    bd_A += mA ? q0.x : 0.f; Hdd_A += mA ? q0.y : 0.f;
    bd_L += mL ? q0.x : 0.f; Hdd_L += mL ? q0.y : 0.f;
#pragma unroll
    for (int c = 0; c < 4; c++) { HcdA[c] += mA ? rh[c] : 0.f; HcdL[c] += mL ? rh[c] : 0.f; }
  }
  float H = Hdd_A + Hdd_L + prior;
  if (H < 1e-10) H = 1e-10;
  const float hdi = 1.0 / H;
  float bds = bd_A + bd_L;
  if (shiftPriorToZero) bds += prior * delta;
  const bool any = ngood > 0;
  o.Hdd_A = Hdd_A; o.bd_A = bd_A; o.Hdd_L = Hdd_L; o.bd_L = bd_L;
  o.HdiF = any ? hdi : 0.f; o.bdSumF = any ? bds : 0.f;
#pragma unroll
  for (int c = 0; c < 4; c++) { o.HcdA[c] = HcdA[c]; o.HcdL[c] = HcdL[c]; o.Hcd[c] = any ? HcdA[c] + HcdL[c] : 0.f; }
  o.mbits = mbits;
  if (onf != 0.f) {
    float* po = B.p_out + (size_t)p * 16;
    *(float4*)(po + 0) = make_float4(Hdd_A, bd_A, HcdA[0], HcdA[1]);
    *(float4*)(po + 4) = make_float4(HcdA[2], HcdA[3], Hdd_L, bd_L);
    *(float4*)(po + 8) = make_float4(HcdL[0], HcdL[1], HcdL[2], HcdL[3]);
    *(float2*)(po + PO_HDI) = make_float2(o.HdiF, o.bdSumF);
    float* tr = (float*)(B.p_track + p);
    *(float2*)(tr + 2) = make_float2(any ? H : 0.f, __int_as_float(mbits));   // p->data->idepth_hessian (:46, :56); the active targets for k_ba_resub*
    if (!any) tr[0] = 0.f;                                                     // p->data->maxRelBaseline = 0 (:47)
  }
}

// ------------------------------------------------------------------ per-point Schur accumulation
// AccumulatedSCHessianSSE::addPoint for nf <= 8 (template NF).  One wave per item (<= 64 consecutive
// points of ONE host); lane (a,c) = (lane>>3, lane&7) keeps element (a,c) of all NF x NF 8x8 D tiles of
// that host in VGPRs, lanes 0..31 / 0..7 the E / EB rows.
// The per-residual records of a point form a dense [target] table (r_rec[p][t], 64 B each, flags 0 when
// the point has no residual to t), so "slot == target": tile indices are compile-time constants, nothing
// is scattered, there is no branch in the loop, and one point costs two coalesced 256-B loads (issued one
// point ahead) + cross-lane moves.  Absent residuals contribute exact zeros.
// Per-point sums (Hdd/bd/Hcd) run in EFPoint::residualsAll order (BaDev::p_order).
template <int NF>
__global__ __launch_bounds__(BA_BLOCK) void k_ba_sc_reg(const BaDev* __restrict__ wins, const uint8_t* __restrict__ pflag, int shiftPriorToZero, int margMode) {
  const BaDev& B = wins[blockIdx.y];
  if (ba_finished(B)) return;
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
  const int npts = it.z - it.y;
  float my_prior = 0.f, my_delta = 0.f; int my_on = 0;
  unsigned my_ord = 0xffffffffu;
  if (lane < npts) {
    const int p = it.y + lane;
    my_prior = B.p_prior[p]; my_delta = B.p_delta[p]; my_ord = B.p_order[p];
    my_on = pflag ? (int)pflag[p] : 1;
  }
  auto fetch = [&](int q, float& a, float& b) {
    const float* base = B.r_rec + (size_t)(it.y + q) * NF * 16;
    a = lane < NF * 16 ? base[lane] : 0.f;
    b = lane + 64 < NF * 16 ? base[lane + 64] : 0.f;
  };
  float vA = 0.f, vB = 0.f, nA = 0.f, nB = 0.f;
  // per-point outputs are parked in the registers of lane q and stored once after the loop
  float o_hddA = 0, o_bdA = 0, o_hddL = 0, o_bdL = 0, o_hdi = 0, o_bds = 0;
  float o_hcA0 = 0, o_hcA1 = 0, o_hcA2 = 0, o_hcA3 = 0, o_hcL0 = 0, o_hcL1 = 0, o_hcL2 = 0, o_hcL3 = 0;
  float o_idh = 0; int o_ng = 0;
  if (npts > 0) fetch(0, vA, vB);
  for (int q = 0; q < npts; q++) {
    if (q + 1 < npts) fetch(q + 1, nA, nB);   // in flight while point q is processed
    const float prior = __shfl(my_prior, q, 64), delta = __shfl(my_delta, q, 64);
    const float onf = __shfl(my_on, q, 64) ? 1.f : 0.f;
    float ja[NF], jc[NF], je[NF];
    float Hdd_A = 0, bd_A = 0, Hdd_L = 0, bd_L = 0, HcdA[4] = {0, 0, 0, 0}, HcdL[4] = {0, 0, 0, 0};
    float ngood = 0;
    // the per-point sums in EFPoint::residualsAll order (BaDev::p_order; the order word of point q is wave-uniform)
    const unsigned ord = (unsigned)__builtin_amdgcn_readfirstlane(__shfl((int)my_ord, q, 64));
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int t = (int)((ord >> (4 * k)) & 15u);
      if (t == 15) break;
      const int isrc = __float_as_int(t < 4 ? vA : vB);
      const int o = (t & 3) * 16;
      const int fl = (int)__int_as_float(__builtin_amdgcn_readlane(isrc, o + RR_FLAGS));
      const float m = ((fl & 1) ? 1.f : 0.f) * onf;                          // residual present and active
      const float mA = (!(fl & 2) && !margMode) ? m : 0.f, mL = m - mA;      // mode 0 vs mode 1/2 sums (AccumulatedTopHessian.cpp:54-71)
      const float rbd = __int_as_float(__builtin_amdgcn_readlane(isrc, o + RR_BD));
      const float rhdd = __int_as_float(__builtin_amdgcn_readlane(isrc, o + RR_HDD));
      float rh[4];
#pragma unroll
      for (int c = 0; c < 4; c++) rh[c] = __int_as_float(__builtin_amdgcn_readlane(isrc, o + RR_HCD + c));
      ngood += m;
      // masked adds: x + 0 is exact, so inactive slots leave the sums untouched
      bd_A += mA != 0.f ? rbd : 0.f; Hdd_A += mA != 0.f ? rhdd : 0.f;
      bd_L += mL != 0.f ? rbd : 0.f; Hdd_L += mL != 0.f ? rhdd : 0.f;
#pragma unroll
      for (int c = 0; c < 4; c++) { HcdA[c] += mA != 0.f ? rh[c] : 0.f; HcdL[c] += mL != 0.f ? rh[c] : 0.f; }
    }
    // the JpJdF rows by target (tile indices are compile-time constants; absent / inactive residuals enter as zeros)
#pragma unroll
    for (int t = 0; t < NF; t++) {
      const float src = t < 4 ? vA : vB;
      const int o = (t & 3) * 16;
      const int fl = (int)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(src), o + RR_FLAGS));
      const float m = ((fl & 1) ? 1.f : 0.f) * onf;
      const float sa = __shfl(src, o + la, 64), sc = __shfl(src, o + lc, 64), se = __shfl(src, o + le, 64);
      ja[t] = m != 0.f ? sa : 0.f; jc[t] = m != 0.f ? sc : 0.f; je[t] = m != 0.f ? se : 0.f;
    }
    float HdiF = 0, bdSumF = 0;
    float Hcd[4] = {0, 0, 0, 0};
    float o_H = 0.f;
    {
      float H = Hdd_A + Hdd_L + prior;
      if (H < 1e-10) H = 1e-10;
      o_H = H;
      const float hdi = 1.0 / H;
      float bds = bd_A + bd_L;
      if (shiftPriorToZero) bds += prior * delta;
      const bool any = ngood > 0.f;
      HdiF = any ? hdi : 0.f; bdSumF = any ? bds : 0.f;
#pragma unroll
      for (int k = 0; k < 4; k++) Hcd[k] = any ? HcdA[k] + HcdL[k] : 0.f;
    }
    if (lane == q) {
      o_idh = ngood > 0.f ? o_H : 0.f; o_ng = (int)ngood;
      o_hddA = Hdd_A; o_bdA = bd_A; o_hddL = Hdd_L; o_bdL = bd_L; o_hdi = HdiF; o_bds = bdSumF;
      o_hcA0 = HcdA[0]; o_hcA1 = HcdA[1]; o_hcA2 = HcdA[2]; o_hcA3 = HcdA[3];
      o_hcL0 = HcdL[0]; o_hcL1 = HcdL[1]; o_hcL2 = HcdL[2]; o_hcL3 = HcdL[3];
    }
    // ngood == 0 => HdiF = 0 and every ja/jc/je is 0: all updates below add exact zeros
    hcc += (HdiF * Hcd[(lane >> 2) & 3]) * Hcd[lane & 3];
    bcv += (bdSumF * HdiF) * Hcd[lane & 3];
    const float hb = HdiF * bdSumF;
    const float hc = Hcd[lane & 3];
#pragma unroll
    for (int t1 = 0; t1 < NF; t1++) {
      const float wl = HdiF * ja[t1];
#pragma unroll
      for (int t2 = 0; t2 < NF; t2++) D[t1][t2] = D[t1][t2] + wl * jc[t2];
      E[t1] = E[t1] + (HdiF * je[t1]) * hc;
      EB[t1] = EB[t1] + hb * jc[t1];
    }
    vA = nA; vB = nB;
  }
  if (lane < npts && my_on) {
    float* po = B.p_out + (size_t)(it.y + lane) * 16;
    *(float4*)(po + 0) = make_float4(o_hddA, o_bdA, o_hcA0, o_hcA1);
    *(float4*)(po + 4) = make_float4(o_hcA2, o_hcA3, o_hddL, o_bdL);
    *(float4*)(po + 8) = make_float4(o_hcL0, o_hcL1, o_hcL2, o_hcL3);
    po[PO_HDI] = o_hdi; po[PO_BDSUM] = o_bds;
    float* tr = (float*)(B.p_track + it.y + lane);
    *(float2*)(tr + 2) = make_float2(o_idh, __int_as_float(o_ng));   // p->data->idepth_hessian (AccumulatedSCHessian.cpp:46, :56)  (w: the count — this variant's launches use the records' flags in k_ba_resub*)
    if (o_ng == 0) tr[0] = 0.f;                                       // p->data->maxRelBaseline = 0 (:47)
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


// ------------------------------------------------------------------ per-point Schur accumulation on the matrix cores
// All Schur sums of a host are ONE symmetric product over its points,
//     D' = Z^T diag(HdiF) Z,   Z[p] = [ JpJdF of target 0..7 (64) | Hcd (4) | bdSumF (1) ]  (69 columns, padded to 80):
//     accD[t1][t2] = D'[8t1.., 8t2..],  accE[t1] = D'[8t1.., 64..67],  accEB[t1] = D'[8t1.., 68],  accHcc = D'[64..67, 64..67],
//     accbc = D'[64..67, 68]                                                     (AccumulatedSCHessian.cpp:75-101).
// One wave per item (<= 64 consecutive points of one host).  Phase 1, lane = point: the per-point terms
// (Hdd/bd/Hcd sums in target-slot order, HdiF, bdSumF — unchanged arithmetic, bit-exact with the CPU path) go to
// p_out and to a small LDS table.  Phase 2, 4 points per step: lane (i, k) = (lane%16, lane/16) loads column
// 16*tt+i of point 4g+k for the 5 column tiles and 21 v_mfma_f32_16x16x4_f32 accumulate the 4x5 (+1) output tiles
// in 84 VGPRs.  Absent / inactive residuals enter as exact zeros.  Only the order of the cross-point float sums
// differs from the CPU path (and the product HdiF*J1*J2 is fused in the MFMA: one rounding less).
__global__ __launch_bounds__(BA_BLOCK) void k_ba_sc_mfma(const BaDev* __restrict__ wins, const uint8_t* __restrict__ pflag, int shiftPriorToZero, int margMode) {
  const BaDev& B = wins[blockIdx.y];
  if (ba_finished(B)) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int item = blockIdx.x * (BA_BLOCK / 64) + wv;
  __shared__ float pt_all[BA_BLOCK / 64][64][8];   // HdiF, bdSumF, Hcd[4], mask bits, -
  if (item >= B.nitems) return;                    // (no block-level barrier below: waves are independent)
  float (*pt)[8] = pt_all[wv];
  const int4 it = B.items[item];
  const int nf = B.nf, npts = it.z - it.y;
  // ---- phase 1: per-point terms
  {
    float HdiF = 0.f, bdSumF = 0.f, Hcd[4] = {0.f, 0.f, 0.f, 0.f};
    int mbits = 0;
    if (lane < npts) {
      const int p = it.y + lane;
      ScPointTerms T;
      sc_point_terms(B, p, B.r_rec + (size_t)p * nf * 16, B.p_order[p], (pflag ? (int)pflag[p] : 1) ? 1.f : 0.f, B.p_prior[p], B.p_delta[p], shiftPriorToZero, margMode, T);
      HdiF = T.HdiF; bdSumF = T.bdSumF; mbits = T.mbits;
#pragma unroll
      for (int k = 0; k < 4; k++) Hcd[k] = T.Hcd[k];
    }
    *(float4*)(&pt[lane][0]) = make_float4(HdiF, bdSumF, Hcd[0], Hcd[1]);
    *(float4*)(&pt[lane][4]) = make_float4(Hcd[2], Hcd[3], __int_as_float(mbits), 0.f);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // ---- phase 2: D' += Z^T diag(HdiF) Z, four points per step
  te_f4 acc[4][5], acc44 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 5; b++) acc[a][b] = (te_f4){0.f, 0.f, 0.f, 0.f};
  const int ci = lane & 15, kq = lane >> 4;
  const int tsub = ci >> 3, asub = ci & 7;
  for (int g = 0; g < npts; g += 4) {
    const int q = g + kq;                      // q < 64 always: rows of points >= npts hold zeros (HdiF = 0, mask = 0)
    const float4 h0 = *(const float4*)(&pt[q][0]);
    const float4 h1 = *(const float4*)(&pt[q][4]);
    const int mb = __float_as_int(h1.z);
    const float* base = B.r_rec + (size_t)(it.y + (q < npts ? q : 0)) * nf * 16 + asub;
    float z[5], za[5];
#pragma unroll
    for (int tt = 0; tt < 4; tt++) {
      const int t = 2 * tt + tsub;
      z[tt] = ((mb >> t) & 1) ? base[t * 16] : 0.f;
    }
    z[4] = ci == 0 ? h0.z : ci == 1 ? h0.w : ci == 2 ? h1.x : ci == 3 ? h1.y : ci == 4 ? h0.y : 0.f;
#pragma unroll
    for (int tt = 0; tt < 5; tt++) za[tt] = h0.x * z[tt];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 5; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[a], z[b], acc[a][b], 0, 0, 0);
    acc44 = __builtin_amdgcn_mfma_f32_16x16x4f32(za[4], z[4], acc44, 0, 0, 0);
  }
  // ---- phase 3: tiles -> the partial layout k_ba_fold_sc expects.  acc[a][b][v] = D'[16a + 4*kq + v][16b + ci]
  float* out = B.sc_part + (size_t)item * sc_part_floats(nf);
  const int nf2 = nf * nf;
#pragma unroll
  for (int a = 0; a < 4; a++) {
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const int R = 16 * a + 4 * kq + v, t1 = R >> 3, ra = R & 7;
      if (t1 < nf) {
#pragma unroll
        for (int b = 0; b < 4; b++) {
          const int C = 16 * b + ci, t2 = C >> 3, cc = C & 7;
          if (t2 < nf) out[(t1 * nf + t2) * 64 + ra * 8 + cc] = acc[a][b][v];
        }
        if (ci < 4) out[nf2 * 64 + t1 * 32 + ra * 4 + ci] = acc[a][4][v];
        if (ci == 4) out[nf2 * 64 + nf * 32 + t1 * 8 + ra] = acc[a][4][v];
      }
    }
  }
  const int per_wave = nf2 * 64 + nf * 32 + nf * 8;
  if (kq == 0) {
#pragma unroll
    for (int v = 0; v < 4; v++) {
      if (ci < 4) out[per_wave + v * 4 + ci] = acc44[v];
      if (ci == 4) out[per_wave + 16 + v] = acc44[v];
    }
  }
}


// ------------------------------------------------------------------ Schur accumulation, one workgroup per host frame
// Same arithmetic as k_ba_sc_mfma (phase 1 per-point terms, phase 2 Z^T diag(HdiF) Z on the matrix cores), but the four waves
// of a workgroup share ALL points of one host (64-point slices dealt round-robin), add their 21 accumulator tiles through LDS
// in a fixed order (3+2 -> 1+0 -> 0) and wave 0 writes the host's accD / accE / accEB bins straight into the packed accumulator
// block: no per-item partials in HBM and no fold pass.  Only Hcc / bc (sums over ALL hosts) leave a 20-float partial per host.
// (Round 3 tried the transposed split — wave a owns tile ROW a for all points of the host: 116 VGPRs and 8 KB of LDS, so all 1024
//  workgroups of a 128-window launch are resident at once instead of two rounds of 512 — and measured it SLOWER, 91 against 72 us: the
//  kernel is bound by the per-wave chain load -> MFMA over its point groups, which that split makes four times longer.)
// NW: waves per workgroup.  4 (default).  2 (SDSO_SC_WAVES=2, A/B): half as many waves share a host, so a wave sees twice the point groups
// (prologue, tree and bins amortise over eight instead of four) and four workgroups instead of two fit a CU.  Measured on MI355X, 256 windows:
// 142 against 134 us — the per-wave chain load -> terms -> MFMA over its groups is what bounds the kernel, and this doubles it.
template <bool PLAIN, int NW = 4>
__global__ __launch_bounds__(64 * NW, 2) void k_ba_sc_host(const BaDev* __restrict__ wins, const uint8_t* __restrict__ pflag, int shiftPriorToZero, int margMode, int signal = 0) {
  const BaDev& B = wins[blockIdx.y];       // (by value — all pointers in SGPRs, no scalar re-loads in the loop — measured equal: 131 vs 133 us)
  if (ba_finished(B)) return;
  const int nf = B.nf, h = blockIdx.x;
  if (h >= nf) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#ifdef SDSO_SC_STAMPS   // diagnostic build: shader-clock ticks of wave 0 of workgroup (0, 0) per phase, printed
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int stn = 0;
#define SCS() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (stn < 8) st[stn++] = __builtin_amdgcn_s_memtime(); } while (0)
  unsigned long long sub[6] = {0, 0, 0, 0, 0, 0}, sub_t = 0;     // inside the group loop: wait for the records + park | request + barrier | r_cj | phase 1 | phase 2 | closing barrier
#define SCSUB(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); sub[i] += tn_ - sub_t; sub_t = tn_; } while (0)
#define SCSUB0() do { sub_t = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SCS() do { } while (0)
#define SCSUB(i) do { } while (0)
#define SCSUB0() do { } while (0)
#endif
  SCS();
  __shared__ float pt_all[NW][16][8];
  // The records of 16 points (16 x nf x 16 floats, contiguous in r_rec) are brought in ONCE by coalesced 16-byte loads and parked in LDS;
  // both phases read them there.  (Before: phase 1 read floats 8..15 of every record with lane = point and phase 2 floats 0..7 with its
  // own lane layout, straight from memory — every 128-byte line crossed the memory system twice, and the launch, two rounds of 512
  // resident workgroups, ran at the rate those 2 x 131 MB stream: the stamps showed each phase at ~16 us per round.)
  // 160 floats per point: the four point groups of an MFMA operand read (kq) land 32 banks apart.  The accumulator tiles of the final
  // tree lie over the same memory.
  constexpr int SC_PSTRIDE = 160;
  constexpr int SC_STAGE = 16 * SC_PSTRIDE;                  // floats per wave
  // D' = Z^T diag(HdiF) Z is symmetric: only the tiles on and above the diagonal of its 4 x 4 grid over the JpJdF columns are accumulated
  // (10 instead of 16; + 4 tiles of the Hcd / bdSumF columns + 1 corner = 15 tiles, 60 accumulator registers instead of 84), and the bins
  // below the diagonal are written as their mirror images.  That also makes accD(i,j,k) == accD(i,k,j)^T hold EXACTLY, as it does in the
  // reference (AccumulatorXX::update multiplies a_i * b_j * w: the same product for both, MatrixAccumulators.h:31-80) — two MFMA tiles
  // (HdiF z_a) z_b and (HdiF z_b) z_a round differently, and the stitch relies on that symmetry (ba_tail.hip: S2 = S1^T).
  constexpr int SC_NT = 15;                                  // tile t of pair (a <= b): sc_ut(a, b); 10 + a: column tile 4 of row a; 14: the corner
  constexpr int BIN_E = 64 * 64, BIN_EB = BIN_E + 8 * 32;
  constexpr int SC_NEED = SC_NT * 256 + BIN_EB + 64;          // one tile buffer + the bins laid out behind it
  constexpr int SC_LDS0 = NW * SC_STAGE > (NW / 2) * SC_NT * 256 ? NW * SC_STAGE : (NW / 2) * SC_NT * 256;
  constexpr int SC_LDS = SC_LDS0 > SC_NEED ? SC_LDS0 : SC_NEED;
  __shared__ __align__(16) float stage_all[SC_LDS];
  float (*tiles)[SC_NT * 256] = reinterpret_cast<float (*)[SC_NT * 256]>(stage_all);   // two waves' worth of accumulator tiles
  float* stg = stage_all + wv * SC_STAGE;
  float (*pt)[8] = pt_all[wv];
  const int ib = B.host_item_beg[h], ie = B.host_item_beg[h + 1];
  const int pb = ib < ie ? B.items[ib].y : 0, pe = ib < ie ? B.items[ie - 1].z : 0;
  te_f4 acc[SC_NT];
#pragma unroll
  for (int t = 0; t < SC_NT; t++) acc[t] = (te_f4){0.f, 0.f, 0.f, 0.f};
  auto sc_ut = [](int a, int b) { return a * 4 - (a * (a - 1)) / 2 + (b - a); };   // index of the upper tile (a <= b): 0..9
  const int ci = lane & 15, kq = lane >> 4;
  const int tsub = ci >> 3, asub = ci & 7;
  // a wave's 16-point groups: 64-point slices dealt round-robin over the waves (as before), four groups per slice
  auto group_p0 = [&](int gidx) { return pb + 64 * (wv + NW * (gidx >> 2)) + 16 * (gidx & 3); };
  // the records of group g0: float4 chunk c = lane + 64 k of its (<= 16 nf 4) chunks
  float pr_next = 0.f, de_next = 0.f;        // prior, delta, residual order and the marginalisation flag of the lane's point of the NEXT group: they travel with its records
  int pf_next = 1;
  unsigned or_next = 0xffffffffu;
  auto request = [&](int p0, float4 (&v)[8]) {
    const int n16 = min(16, pe - p0);
    const int nchunks = n16 > 0 ? n16 * nf * 4 : 0;
    const float* rbase = B.r_rec + (size_t)(n16 > 0 ? p0 : pb) * nf * 16;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int c = lane + 64 * k;
      v[k] = c < nchunks ? *(const float4*)(rbase + (size_t)c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (lane < n16) { pr_next = B.p_prior[p0 + lane]; de_next = B.p_delta[p0 + lane]; or_next = B.p_order[p0 + lane]; pf_next = pflag ? (int)pflag[p0 + lane] : 1; }
  };
  float4 vnext[8];
  SCS();
  request(group_p0(0), vnext);
  SCS();
  for (int gidx = 0;; gidx++) {
    const int p0 = group_p0(gidx);
    if (p0 >= pe) break;
    const int npts = min(16, pe - p0);
    SCSUB0();
    {  // park the group's records: chunk c -> record c / 4 = point * nf + target, floats 4 (c & 3) ..
      const int nchunks = npts * nf * 4;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int c = lane + 64 * k;
        if (c < nchunks) {
          const int rec = c >> 2, pl = nf == 8 ? rec >> 3 : rec / nf, t = rec - pl * nf;
          *(float4*)(stg + pl * SC_PSTRIDE + t * 16 + 4 * (c & 3)) = vnext[k];
        }
      }
    }
    const float prior = pr_next, delta = de_next;
    const float onf = pf_next ? 1.f : 0.f;
    const unsigned order = or_next;
    SCSUB(0);
    request(group_p0(gidx + 1), vnext);          // the next group's records travel while this one is worked on.  (Requested behind this group's
                                                 // stores instead — so that the wait for them never waits for a younger store — measured equal.)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    SCSUB(1);
    {  // the JpJdF halves of the parked records -> the compact copy the back-substitution streams (BaDev::r_cj): four fully coalesced 1-KB
       // stores per group, issued BEHIND the prefetch loads (gfx9 counts stores in vmcnt: in front of them the wait for the next group's
       // records would also wait for these stores' acknowledgements)
      const int nhalf = npts * nf * 2;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int c2 = lane + 64 * k;
        if (c2 < nhalf) {
          const int rec = c2 >> 1, pl = nf == 8 ? rec >> 3 : rec / nf, t = rec - pl * nf;
          const float4 q = *(const float4*)(stg + pl * SC_PSTRIDE + t * 16 + 4 * (c2 & 1));
          *(float4*)(B.r_cj + ((size_t)p0 * nf + rec) * 8 + 4 * (c2 & 1)) = q;
        }
      }
    }
    SCSUB(2);
    {  // phase 1: per-point terms, lane = point (identical to k_ba_sc_mfma)
      float HdiF = 0.f, bdSumF = 0.f, Hcd[4] = {0.f, 0.f, 0.f, 0.f};
      int mbits = 0;
      if (lane < npts) {
        ScPointTerms T;
        sc_point_terms<PLAIN>(B, p0 + lane, stg + lane * SC_PSTRIDE, order, onf, prior, delta, shiftPriorToZero, margMode, T);
        HdiF = T.HdiF; bdSumF = T.bdSumF; mbits = T.mbits;
#pragma unroll
        for (int k = 0; k < 4; k++) Hcd[k] = T.Hcd[k];
      }
      if (lane < 16) {
        *(float4*)(&pt[lane][0]) = make_float4(HdiF, bdSumF, Hcd[0], Hcd[1]);
        *(float4*)(&pt[lane][4]) = make_float4(Hcd[2], Hcd[3], __int_as_float(mbits), 0.f);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    SCSUB(3);
    {  // phase 2: the group's four MFMA operand sets (points 4 u + kq), read from the parked records as they are: the record of a
       // residual that is not active (or of a target the point does not observe) holds zeros, and a point without an active residual,
       // outside the marginalisation filter or past npts has HdiF = 0 and zero terms in pt[] — exact no-ops, no select needed.
      float zz[4][5], hx[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int q = 4 * u + kq;
        const float4 h0 = *(const float4*)(&pt[q][0]);
        const float4 h1 = *(const float4*)(&pt[q][4]);
        const float* base = stg + q * SC_PSTRIDE + asub;
#pragma unroll
        for (int tt = 0; tt < 4; tt++) zz[u][tt] = q < npts ? base[(2 * tt + tsub) * 16] : 0.f;   // (rows past npts are stale LDS)
        zz[u][4] = ci == 0 ? h0.z : ci == 1 ? h0.w : ci == 2 ? h1.x : ci == 3 ? h1.y : ci == 4 ? h0.y : 0.f;
        hx[u] = h0.x;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        float za[5];
#pragma unroll
        for (int tt = 0; tt < 5; tt++) za[tt] = hx[u] * zz[u][tt];
#pragma unroll
        for (int a = 0; a < 4; a++) {
#pragma unroll
          for (int b = a; b < 4; b++) acc[sc_ut(a, b)] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[a], zz[u][b], acc[sc_ut(a, b)], 0, 0, 0);
          acc[10 + a] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[a], zz[u][4], acc[10 + a], 0, 0, 0);
        }
        acc[14] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[4], zz[u][4], acc[14], 0, 0, 0);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    SCSUB(4);
    __builtin_amdgcn_wave_barrier();          // the next group overwrites the stage and pt[]
    if (gidx == 0) SCS();
  }
  SCS();
  __syncthreads();                            // the tiles lie over the other waves' stages
  // ---- fixed-order tree over the four waves: (3 -> 1, 2 -> 0), then (1 -> 0)
  auto put = [&](float* dst) {
#pragma unroll
    for (int t = 0; t < SC_NT; t++)
#pragma unroll
      for (int v = 0; v < 4; v++) dst[(t * 4 + v) * 64 + lane] = acc[t][v];
  };
  auto add = [&](const float* src) {
#pragma unroll
    for (int t = 0; t < SC_NT; t++)
#pragma unroll
      for (int v = 0; v < 4; v++) acc[t][v] += src[(t * 4 + v) * 64 + lane];
  };
  if constexpr (NW == 4) {
    if (wv >= 2) put(tiles[wv - 2]);
    __syncthreads();
    if (wv < 2) add(tiles[wv]);
    __syncthreads();
  }
  if (wv == 1) put(tiles[0]);
  __syncthreads();
  // ---- the host's bins.  acc[a][b][v] = D'[16a + 4*kq + v][16b + ci]: wave 0 lays the finished tiles out in LDS the way the bins lie in
  // memory (64-float blocks per (t1, t2)), then all four waves write them out, one 256-byte block per store (from the accumulator
  // layout a store covered eight 32-byte pieces of four blocks, 84 such stores on one wave: 9 k of the kernel's 75 k cycles)
  const int nf2 = nf * nf;
  float* accD = B.accum + acc_off_D(nf);
  float* accE = B.accum + acc_off_E(nf);
  float* accEB = B.accum + acc_off_EB(nf);
  float* bins = stage_all + SC_NT * 256;       // the second tile buffer: free since the tree's second barrier
  static_assert(SC_NEED <= SC_LDS, "the bins fit behind the first tile buffer");
  if (wv == 0) {
    add(tiles[0]);
    SCS();
#pragma unroll
    for (int a = 0; a < 4; a++) {
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const int Rr = 16 * a + 4 * kq + v, t1 = Rr >> 3, ra = Rr & 7;
#pragma unroll
        for (int b = a; b < 4; b++) {              // element (Rr, Cc) of the upper triangle and its mirror image (Cc, Rr)
          const int Cc = 16 * b + ci, t2 = Cc >> 3, cc = Cc & 7;
          const float val = acc[sc_ut(a, b)][v];
          if (b > a || Rr <= Cc) {
            bins[(t1 * 8 + t2) * 64 + ra * 8 + cc] = val;
            if (Rr != Cc) bins[(t2 * 8 + t1) * 64 + cc * 8 + ra] = val;
          }
        }
        if (ci < 4) bins[BIN_E + t1 * 32 + ra * 4 + ci] = acc[10 + a][v];
        if (ci == 4) bins[BIN_EB + t1 * 8 + ra] = acc[10 + a][v];
      }
    }
    float* hp = B.sc_part + (size_t)h * 20;      // Hcc (16) and bc (4) of this host; k_ba_fold_all adds the hosts
    if (kq == 0) {
#pragma unroll
      for (int v = 0; v < 4; v++) {
        if (ci < 4) hp[v * 4 + ci] = acc[14][v];
        if (ci == 4) hp[16 + v] = acc[14][v];
      }
    }
  }
  __syncthreads();
#pragma unroll 4
  for (int blk = wv; blk < 64; blk += NW) {
    const int t1 = blk >> 3, t2 = blk & 7;
    if (t1 < nf && t2 < nf) accD[(size_t)(h + t1 * nf + t2 * nf2) * 64 + lane] = bins[blk * 64 + lane];
  }
#pragma unroll
  for (int idx = 64 * wv + lane; idx < 256; idx += 64 * NW) {
    const int t1 = idx >> 5;
    if (t1 < nf) accE[(size_t)(h + t1 * nf) * 32 + (idx & 31)] = bins[BIN_E + idx];
  }
  if (wv == 1 && (lane >> 3) < nf) accEB[(size_t)(h + (lane >> 3) * nf) * 8 + (lane & 7)] = bins[BIN_EB + lane];
  if (signal) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }   // (every wave's stores are out before wave 0 signals)
  if (wv != 0) return;
  SCS();
#ifdef SDSO_SC_STAMPS
  if ((blockIdx.x == 0 || blockIdx.x == 5) && (blockIdx.y == 0 || blockIdx.y == 100) && lane == 0)
    printf("sc_host (%d,%d) ticks: prologue %llu  first records %llu  group 0 %llu  other groups %llu  wave tree %llu  bins %llu | total %llu  (points %d)\n", blockIdx.x, blockIdx.y,
           st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6] - st[5], st[6] - st[0], pe - pb);
  if ((blockIdx.x == 0 || blockIdx.x == 5) && (blockIdx.y == 0 || blockIdx.y == 100) && lane == 0)
    printf("sc_host (%d,%d) inside the group loop, all groups of wave 0: wait + park %llu  request + barrier %llu  r_cj %llu  phase 1 %llu  phase 2 %llu\n", blockIdx.x, blockIdx.y, sub[0], sub[1], sub[2], sub[3], sub[4]);
#endif
  if (signal) {
    // side-stream launch: tell the tail kernel (already resident, polling) that this host's bins, Hcc partial and per-point terms are
    // in memory.  The other waves' p_out stores were drained before the workgroup barriers above (__syncthreads waits for vmcnt(0));
    // this wave drains its own, releases at agent scope (L2 write-back) and only then bumps the counter.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(&B.opt->sc_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}


}  // namespace sdso
