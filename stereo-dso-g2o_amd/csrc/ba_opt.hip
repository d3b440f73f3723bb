// The host logic of FullSystem::optimize's Gauss-Newton loop between the kernel phases, on the device — one 256-thread workgroup
// per window, so that a whole loop (any number of windows) is enqueued without a single host round trip.
// Reference (paths under /root/reference):
//   FullSystem::optimize loop body                                   src/FullSystem/FullSystemOptimize.cpp:927-991
//   FullSystem::backupState / doStepFromBackup (break test)          FullSystemOptimize.cpp:309-351, :207-305
//   FullSystem::setNewFrameEnergyTH (70 % quantile, newest frame)    FullSystemOptimize.cpp:98-139
//   FrameHessian::setState -> PRE_worldToCam                         src/FullSystem/HessianBlocks.h:197-214
//   FrameFramePrecalc::set (setPrecalcValues)                        src/FullSystem/HessianBlocks.cpp:206-242
//   EnergyFunctional::setDeltaF                                      src/OptimizationBackend/EnergyFunctional.cpp:173-207
//   CalibHessian::setValue                                           HessianBlocks.h:318-333
// This path is the accepted-step flow (setting_forceAceptStep = true, the reference's default, settings.cpp:53): every step is
// taken, so linearizeAll + applyRes + the next accumulate collapse into the fused kernel.  The energy-gated loop stays on the host
// (sdso_ba_optimize).  Sharded windows (one rank per GPU): the quantile and the break test need every rank's residuals / points, so
// each rank packs them (k_ba_opt_pack) and ONE all-gather per iteration feeds the same k_ba_opt_step everywhere.
#include "ba_kernels.h"
#include "host_math.h"

namespace sdso {

// floats of one window's packed record: [0,cap) newEnergyWithOutlier of the residuals that enter the quantile (-1 = none),
// then the energy of the linearisation (a double in two float slots), sum |idepth backup|, number of points, and (energy-gated flow)
// this rank's part of calcLEnergy's point / residual terms
constexpr int OPT_PACK_TAIL = 5;
__host__ __device__ inline int opt_pack_floats(int cap) { return cap + OPT_PACK_TAIL; }

// after the points' step: this rank's contribution to setNewFrameEnergyTH and to the break test
__global__ __launch_bounds__(256) void k_ba_opt_pack(const BaDev* __restrict__ wins, float* __restrict__ out, int cap, int nparts /* energy partials: 0 -> nchunks (fused), else ceil(nr/256) */,
                                                     const float* __restrict__ sums, int sums_stride, const float* __restrict__ lpart = nullptr /* k_ba_lenergy's partials (gated flow) */, int lstride = 0) {
  const BaDev& B = wins[blockIdx.y];
  if (ba_finished_lin(B)) return;
  float* o = out + (size_t)blockIdx.y * opt_pack_floats(cap);
  const int first = B.opt->newest_first, nf = B.nf;
  for (int k = threadIdx.x; k < cap; k += 256) {
    const int i = first + k;
    float e = -1.f;
    if (i < B.nr && B.r_target[i] == nf - 1 && !B.r_lin[i]) e = B.r_newEnergyWO[i];
    o[k] = e;
  }
  if (threadIdx.x == 0) {
    double s = 0;
    const int np_ = nparts ? (B.nr + BA_BLOCK - 1) / BA_BLOCK : B.nchunks;
    for (int b = 0; b < np_; b++) s += B.e_part[b];
    __builtin_memcpy(o + cap, &s, 8);
    float sumNID = 0;
    if (sums) {
      const float* sm = sums + (size_t)blockIdx.y * sums_stride;
      for (int b = 0; b < (B.np + BA_BLOCK - 1) / BA_BLOCK; b++) sumNID += sm[2 * b + 1];
    }
    o[cap + 2] = sumNID;
    o[cap + 3] = (float)B.np;
    float Ept = 0;
    if (lpart) {
      const float* lp = lpart + (size_t)blockIdx.y * lstride;
      const int nlp = B.nchunks + (B.np + BA_BLOCK - 1) / BA_BLOCK;
      for (int b = 0; b < nlp; b++) Ept += lp[b];
    }
    o[cap + 4] = Ept;
  }
}

// the energies that enter setNewFrameEnergyTH: the all-gathered records of every rank, or (single rank) the window's own residuals
struct OptEnergies {
  const float* g; size_t rstride; int nranks, cap;       // gathered records
  const BaDev* B; int first;                             // local residuals (g == nullptr)
  __device__ __forceinline__ int ranks() const { return g ? nranks : 1; }
  __device__ __forceinline__ float at(int r, int j) const {
    if (g) return g[r * rstride + j];
    const int i = first + j;
    return (i < B->nr && B->r_target[i] == B->nf - 1 && !B->r_lin[i]) ? B->r_newEnergyWO[i] : -1.f;
  }
};
// the k-th smallest (0-based) of the non-negative values: 4 radix passes over the bit patterns.  NT threads call it together
// (NT = 256 or 512); the 256 bins are owned by the first 256 threads.
template <int NT>
__device__ inline float opt_select(const OptEnergies& v, int k, unsigned* hist /* LDS 256 */, unsigned* sh /* LDS 6 */) {
  const int nranks = v.ranks(), cap = v.cap;
  const int tid = threadIdx.x;
  unsigned prefix = 0, mask = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int r = 0; r < nranks; r++)
      for (int j = tid; j < cap; j += NT) {
        const float e = v.at(r, j);
        if (!(e >= 0)) continue;
        const unsigned key = __float_as_uint(e);
        if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1u);
      }
    __syncthreads();
    // bin b with  cum(b) <= k < cum(b) + hist[b]  (cum = exclusive prefix sum): thread b owns bin b; wave-level scan + the wave totals
    unsigned hb = 0, inc = 0;
    if (tid < 256) {
      hb = hist[tid];
      inc = hb;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_up(inc, o, 64); if ((int)(tid & 63) >= o) inc += up; }
      if ((tid & 63) == 63) sh[2 + (tid >> 6)] = inc;
    }
    __syncthreads();
    if (tid < 256) {
      unsigned base = 0;
      for (int w = 0; w < (int)(tid >> 6); w++) base += sh[2 + w];
      const unsigned excl = base + inc - hb;
      if (hb > 0 && excl <= (unsigned)k && (unsigned)k < excl + hb) { sh[0] = tid; sh[1] = excl; }
    }
    __syncthreads();
    prefix |= sh[0] << shift;
    mask |= 255u << shift;
    k -= (int)sh[1];
    __syncthreads();
  }
  return __uint_as_float(prefix);
}

// LDS of opt_step_body (the caller provides it: a kernel of its own declares one, the fused tail kernel aliases it onto dead buffers)
struct OptStepSmem {
  static constexpr int kStage = 8192;                   // energies staged for the radix passes (32 KiB); larger sets re-read global
  float en[kStage];
  unsigned hist[256], sh[8];
  int cnt[8];
  double esum[8];
  float nid[8];
  double w2c[8][12], c2w[8][12], step[8][8], affd[8][2];
  float K[9], Ki[9], dlt[8][8], zb[8];
};

// ---- the part of opt_step_body that does not depend on the solver's x — the energies of the linearisation, their 70 % quantile, the
// energy sum — by ONE wave without any workgroup barrier, so that the fused tail kernel can run it on an idle wave beside the
// factorisation.  Same exact order statistic as opt_select (four radix passes over the bit patterns), the energy sum in another order.
struct OptPreSmem {
  static constexpr int kStage = 4096;                   // larger sets: the workgroup path of opt_step_body
  float en[kStage];
  unsigned hist[256];
};
struct OptPre { float valid, th; double esum; };        // valid != 0: th (the new frameEnergyTH of the newest frame) and esum are set
__device__ __forceinline__ void opt_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ inline float opt_wave_select(const float* v, int n, int k, unsigned* hist) {
  const int lane = threadIdx.x & 63;
  unsigned prefix = 0, mask = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
#pragma unroll
    for (int q = 0; q < 4; q++) hist[64 * q + lane] = 0;
    opt_wave_sync();
    for (int j = lane; j < n; j += 64) {
      const float e = v[j];
      if (!(e >= 0)) continue;
      const unsigned key = __float_as_uint(e);
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1u);
    }
    opt_wave_sync();
    // lane owns bins 4 lane .. 4 lane + 3; exclusive prefix over the lanes, then inside the lane
    unsigned h[4], tot = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) { h[q] = hist[4 * lane + q]; tot += h[q]; }
    unsigned inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_up(inc, o, 64); if (lane >= o) inc += up; }
    unsigned excl = inc - tot;
    int bin = -1; unsigned bex = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (h[q] > 0 && excl <= (unsigned)k && (unsigned)k < excl + h[q]) { bin = 4 * lane + q; bex = excl; }
      excl += h[q];
    }
    const unsigned long long who = __ballot(bin >= 0);
    const int src = who ? __ffsll((long long)who) - 1 : 0;
    const int b = __shfl(bin, src, 64);
    const unsigned be = __shfl(bex, src, 64);
    prefix |= (unsigned)(b < 0 ? 0 : b) << shift;
    mask |= 255u << shift;
    k -= (int)be;
    opt_wave_sync();
  }
  return __uint_as_float(prefix);
}
// The energies of the residuals into the newest frame (-1: not one of them), requested by ALL threads of the workgroup in one go — kU per
// thread in registers — and parked in P.en later (opt_pre_park): a single wave walking them was a chain of 25 dependent round trips,
// the longest thing beside the factorisation once a whole batch shares HBM.
template <int NT>
struct OptPreRegs { static constexpr int kU = OptPreSmem::kStage / NT; float e[kU]; };
template <int NT>
__device__ __forceinline__ void opt_pre_request(const BaDev& B, OptPreRegs<NT>& R) {
  const int first = B.opt->newest_first, cap = max(B.nr - first, 0), nf = B.nf;
#pragma unroll
  for (int u = 0; u < OptPreRegs<NT>::kU; u++) {
    const int j = (int)threadIdx.x + u * NT, i = first + j;
    R.e[u] = -1.f;
    if (cap <= OptPreSmem::kStage && j < cap) R.e[u] = (B.r_target[i] == nf - 1 && !B.r_lin[i]) ? B.r_newEnergyWO[i] : -1.f;
  }
}
template <int NT>
__device__ __forceinline__ void opt_pre_park(const BaDev& B, OptPreSmem& P, const OptPreRegs<NT>& R) {
  const int cap = max(B.nr - B.opt->newest_first, 0);
  if (cap > OptPreSmem::kStage) return;
#pragma unroll
  for (int u = 0; u < OptPreRegs<NT>::kU; u++) { const int j = (int)threadIdx.x + u * NT; if (j < cap) P.en[j] = R.e[u]; }
}
__device__ inline void opt_pre_wave(const BaDev& B, OptPreSmem& P, OptPre* out, bool parked = false) {
  const BaOptDev& O = *B.opt;
  const int lane = threadIdx.x & 63, nf = B.nf;
  const int first = O.newest_first, cap = max(B.nr - first, 0);
  if (cap > OptPreSmem::kStage) { if (lane == 0) out->valid = 0.f; return; }
  int cnt = 0;
  for (int j = lane; j < cap; j += 64) {
    float e;
    if (parked) e = P.en[j];
    else {
      const int i = first + j;
      e = (B.r_target[i] == nf - 1 && !B.r_lin[i]) ? B.r_newEnergyWO[i] : -1.f;
      P.en[j] = e;
    }
    cnt += e >= 0 ? 1 : 0;
  }
  double esum = 0;
  for (int b = lane; b < B.nchunks; b += 64) esum += B.e_part[b];     // (the fused kernel's partials: one per chunk)
  esum = wave_sum(esum);
  const int M = (int)wave_sum((float)cnt);
  opt_wave_sync();
  float th = 12 * 12 * 8;
  if (M > 0) {
    const int nth = (int)(0.7f * M);
    const float q = opt_wave_select(P.en, cap, nth, P.hist);
    const float nthElement = sqrtf(q);
    th = nthElement * 1.5f;
    th = 26.0f * 0.5f + th * (1 - 0.5f);
    th = th * th;
    th *= 1.0f * 1.0f;
  }
  if (lane == 0) { out->th = th; out->esum = esum; out->valid = 1.f; }
}

// One GN iteration's host part for every window.  gathered: [nranks][nwin][opt_pack_floats(cap)]; nullptr on a single rank: the
// energies, energy partials and point sums are read where the kernels left them (no pack launch).
//   last == 0:  consume the energies of the linearisation at the current state (lastEnergy, setNewFrameEnergyTH), take the step the
//               solver left in sol (backupState + doStepFromBackup for frames and calibration; the points were stepped by
//               k_ba_points_op), rebuild the tables (setPrecalcValues + setDeltaF), evaluate the break test.
//   last == 1 or the break test fired in the previous call: consume the energies only, then mark the window finished.
//   last == 2 (energy-gated flow): the step only — the energies of the trial linearisation are consumed by k_ba_opt_gate afterwards;
//               the tables and calibration scalars of the current state are kept for loadSateBackup, the break test is only recorded.
// xsol: the solver's x of this iteration (global B.sol segment, or the fused tail kernel's copy in LDS); nres_f: nres[0] of the
// accumulate (read from the packed block by the caller); wsums: this window's per-block point sums (sumID, sumNID pairs; nwsums pairs)
template <int NT>
__device__ __forceinline__ void opt_step_body(BaDev& Bw, const BaDev& B, OptStepSmem& S, const float* __restrict__ gathered, int nranks, int cap, int iteration, int last,
                                              int stop_on_convergence, float stepsize, int unfused_parts, const float* __restrict__ wsums, int nwsums,
                                              const double* __restrict__ x, float p_nres, int win, int nwin, const OptPre* pre = nullptr) {
  BaOptDev& O = *B.opt;
  const int tid = threadIdx.x, nf = B.nf;
  const int pf = opt_pack_floats(cap);
  const size_t rstride = (size_t)nwin * pf;
  const float* g = gathered ? gathered + (size_t)win * pf : nullptr;
  OptEnergies en{g, rstride, nranks, cap, &B, O.newest_first};
  if (!g) { nranks = 1; en.cap = cap = max(B.nr - O.newest_first, 0); }
  const int phase = O.phase;

  // ---- every global input of the step below is requested here, before the quantile (LDS work) — the kernel is a chain of memory round
  // trips otherwise, and each of them stretches several-fold while another batch's linearisation saturates HBM
  double p_x[8], p_st[10], p_zero[8], p_ev[12];
  constexpr int U = 512 / NT;                            // nf^2 * 8 <= 512 adjoint columns over NT threads
  float p_ah[U][8], p_at[U][8], p_exp_h = 1, p_exp_t = 1;
  if (tid < nf) {
#pragma unroll
    for (int i = 0; i < 8; i++) { p_x[i] = x[4 + 8 * tid + i]; p_zero[i] = O.state_zero[tid][i]; }
#pragma unroll
    for (int i = 0; i < 10; i++) p_st[i] = O.state[tid][i];
#pragma unroll
    for (int i = 0; i < 12; i++) p_ev[i] = O.evalPT[tid][i];
  } else if (tid == 64) {
#pragma unroll
    for (int i = 0; i < 4; i++) { p_x[i] = x[i]; p_st[i] = O.calib_value[i]; p_zero[i] = O.calib_zero[i]; }
  }
  if (tid < nf * nf) { p_exp_h = O.ab_exposure[tid / nf]; p_exp_t = O.ab_exposure[tid % nf]; }
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int e = tid + NT * u;
    if (e < nf * nf * 8) {
      const int idx = e >> 3, j = e & 7;
#pragma unroll
      for (int i = 0; i < 8; i++) { p_ah[u][i] = (float)B.t_adHost[(size_t)idx * 64 + i * 8 + j]; p_at[u][i] = (float)B.t_adTarget[(size_t)idx * 64 + i * 8 + j]; }
    }
  }
  const int p_its = O.iterations;
  // SOLVER_STEPMOMENTUM: the loop's own stepsize; SOLVER_MOMENTUM: the whole step plus half of the previous one on the poses, no step
  // factor (doStepFromBackup, FullSystemOptimize.cpp:225-236) — both left in BaOptDev by k_ba_opt_momentum, which ran after the solve
  const bool momentum = (B.solver_mode & SOLVER_MOMENTUM) != 0;
  if (B.solver_mode & (SOLVER_MOMENTUM | SOLVER_STEPMOMENTUM)) stepsize = momentum ? 1.0f : O.stepsize;
  double p_xb[6] = {0, 0, 0, 0, 0, 0};
  if (momentum && tid < nf) {
#pragma unroll
    for (int i = 0; i < 6; i++) p_xb[i] = O.x_backup[4 + 8 * tid + i];
  }

  // ---- setNewFrameEnergyTH over every rank's residuals into the newest frame
  // (the fused tail kernel has done this part on an idle wave beside the factorisation: pre)
  const bool have_pre = pre != nullptr && !g && pre->valid != 0.f;      // (uniform: the barriers below are skipped by every thread)
  float th = 12 * 12 * 8;
  float nidsum = 0;
  if (have_pre) {
    th = pre->th;
    if (wsums) for (int b = tid; b < nwsums; b += NT) nidsum += wsums[2 * b + 1];
    nidsum = wave_sum(nidsum);
    if ((tid & 63) == 0) { S.esum[tid >> 6] = tid == 0 ? pre->esum : 0.0; S.nid[tid >> 6] = nidsum; }
    __syncthreads();
  } else {
  // one pass over global memory: the energies go to LDS (all loads of a thread in flight together), the radix passes read LDS
  const bool staged = nranks * cap <= OptStepSmem::kStage;
  int cnt = 0;
  for (int r = 0; r < nranks; r++)
    for (int j = tid; j < cap; j += NT) {
      const float e = en.at(r, j);
      if (staged) S.en[r * cap + j] = e;
      cnt += e >= 0 ? 1 : 0;
    }
  // the energy partials and the points' |idepth| sums of this rank (single-rank path), summed by all threads
  double esum = 0;
  if (!g) {
    const int np_ = unfused_parts ? (B.nr + BA_BLOCK - 1) / BA_BLOCK : B.nchunks;
    for (int b = tid; b < np_; b += NT) esum += B.e_part[b];
    if (wsums) for (int b = tid; b < nwsums; b += NT) nidsum += wsums[2 * b + 1];
    esum = wave_sum(esum); nidsum = wave_sum(nidsum);
    if ((tid & 63) == 0) { S.esum[tid >> 6] = esum; S.nid[tid >> 6] = nidsum; }
  }
  if (staged) { en.g = S.en; en.rstride = cap; en.nranks = nranks; en.cap = cap; }
  cnt = (int)wave_sum((float)cnt);                       // <= nranks * cap < 2^24: exact in float
  if ((tid & 63) == 0) S.cnt[tid >> 6] = cnt;
  __syncthreads();
  int M = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; w++) M += S.cnt[w];
  if (M > 0) {
    const int nth = (int)(0.7f * M);
    const float q = opt_select<NT>(en, nth, S.hist, S.sh);
    const float nthElement = sqrtf(q);
    th = nthElement * 1.5f;
    th = 26.0f * 0.5f + th * (1 - 0.5f);
    th = th * th;
    th *= 1.0f * 1.0f;
  }
  }
  const bool gated = last == 2;
  if (tid == 0 && !gated) {
    B.t_frameTH[nf - 1] = th;
    O.frameTH_new = th;
    double e = 0;
    if (g) for (int r = 0; r < nranks; r++) { double er; __builtin_memcpy(&er, g + r * rstride + cap, 8); e += er; }
    else { e = (S.esum[0] + S.esum[1]) + (S.esum[2] + S.esum[3]); if (NT > 256) e += (S.esum[4] + S.esum[5]) + (S.esum[6] + S.esum[7]); }
    O.lastEnergy = e;
  }
  if (!gated && (phase == 1 || last)) {
    if (tid == 0) { O.phase = 2; Bw.finished = 2; }
    return;
  }
  if (gated) {   // what a rejected step puts back (loadSateBackup + setPrecalcValues, FullSystemOptimize.cpp:355-370)
    for (int e = tid; e < nf * nf * 27; e += NT) O.bk_precalc[e] = B.t_precalc[e];
    for (int e = tid; e < nf * nf * 8; e += NT) O.bk_adHTdelta[e] = B.t_adHTdelta[e];
    for (int e = tid; e < nf * 16 + 8 + nf * 8; e += NT) O.bk_prior[e] = B.t_prior[e];
    if (tid < 4) O.bk_cdelta[tid] = B.t_cdelta[tid];
    if (tid == 0) { O.bk_calib[0] = B.fxl; O.bk_calib[1] = B.fyl; O.bk_calib[2] = B.cxl; O.bk_calib[3] = B.cyl; O.bk_calib[4] = B.fxli; O.bk_calib[5] = B.fyli; }
    __syncthreads();
  }

  // ---- backupState + doStepFromBackup: frames on threads 0..nf-1, the calibration on thread 64
  double* tp = const_cast<double*>(B.t_prior);
  if (tid < nf) {
    const int f = tid;
    double ns[10], sc[10];
#pragma unroll
    for (int i = 0; i < 10; i++) {
      double st = i < 8 ? -p_x[i < 8 ? i : 0] : 0.0;           // step = -x (EnergyFunctional.cpp:978-985)
      if (momentum && i < 6) st += 0.5f * -p_xb[i];            // step.head<6>() += 0.5f * step_backup.head<6>()
      if (i < 8) S.step[f][i] = st;
      ns[i] = p_st[i] + (double)stepsize * st;
    }
#pragma unroll
    for (int i = 0; i < 10; i++) { O.state_backup[f][i] = p_st[i]; O.state[f][i] = ns[i]; }
    for (int i = 0; i < 3; i++) sc[i] = SCALE_XI_TRANS * ns[i];
    for (int i = 3; i < 6; i++) sc[i] = SCALE_XI_ROT * ns[i];
    sc[6] = SCALE_A * ns[6]; sc[7] = SCALE_B * ns[7]; sc[8] = SCALE_A * ns[8]; sc[9] = SCALE_B * ns[9];
    Se3 E;
    for (int i = 0; i < 9; i++) E.R[i] = p_ev[i];
    for (int i = 0; i < 3; i++) E.t[i] = p_ev[9 + i];
    const Se3 Wc = expSe3(sc) * E;                             // PRE_worldToCam = SE3::exp(w2c_leftEps()) * worldToCam_evalPT
    const Se3 Cw = inverse(Wc);
    for (int i = 0; i < 9; i++) { S.w2c[f][i] = Wc.R[i]; S.c2w[f][i] = Cw.R[i]; }
    for (int i = 0; i < 3; i++) { S.w2c[f][9 + i] = Wc.t[i]; S.c2w[f][9 + i] = Cw.t[i]; }
    // setDeltaF: delta_prior = state, delta = state - state_zero
#pragma unroll
    for (int i = 0; i < 8; i++) {
      tp[nf * 8 + f * 8 + i] = ns[i];
      const double dl = ns[i] - p_zero[i];
      tp[nf * 16 + 8 + f * 8 + i] = dl;
      S.dlt[f][i] = (float)dl;
    }
    S.zb[f] = (float)(p_zero[7] * SCALE_B);
  }
  if (tid < nf) { S.affd[tid][0] = SCALE_A * (p_st[6] + (double)stepsize * -p_x[6]); S.affd[tid][1] = SCALE_B * (p_st[7] + (double)stepsize * -p_x[7]); }
  if (tid == 64) {
    double v[4], vs[4];
    float vsf[4];
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = p_st[i] + stepsize * -p_x[i];
#pragma unroll
    for (int i = 0; i < 4; i++) { O.calib_backup[i] = p_st[i]; O.calib_value[i] = v[i]; }
    vs[0] = SCALE_F * v[0]; vs[1] = SCALE_F * v[1]; vs[2] = SCALE_C * v[2]; vs[3] = SCALE_C * v[3];
    for (int i = 0; i < 4; i++) vsf[i] = (float)vs[i];
    Bw.fxl = vsf[0]; Bw.fyl = vsf[1]; Bw.cxl = vsf[2]; Bw.cyl = vsf[3];
    Bw.fxli = 1.0f / vsf[0]; Bw.fyli = 1.0f / vsf[1];
    const float K[9] = {vsf[0], 0, vsf[2], 0, vsf[1], vsf[3], 0, 0, 1};
    float Ki[9];
    inv3f(K, Ki);
    for (int i = 0; i < 9; i++) { S.K[i] = K[i]; S.Ki[i] = Ki[i]; }
    float* cd = const_cast<float*>(B.t_cdelta);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float c = (float)(v[i] - p_zero[i]);
      cd[i] = c;
      tp[nf * 16 + 4 + i] = (double)c;
    }
  }
  __syncthreads();

  // ---- setPrecalcValues: one (host, target) pair per thread; the evalPT parts (PRE_RTll_0, PRE_tTll_0) do not change in the loop
  if (tid < nf * nf) {
    const int h = tid / nf, t = tid % nf;
    float* o = const_cast<float*>(B.t_precalc) + (size_t)(h * nf + t) * 27;
    Se3 Tw, Ch;
    for (int i = 0; i < 9; i++) { Tw.R[i] = S.w2c[t][i]; Ch.R[i] = S.c2w[h][i]; }
    for (int i = 0; i < 3; i++) { Tw.t[i] = S.w2c[t][9 + i]; Ch.t[i] = S.c2w[h][9 + i]; }
    const Se3 l = Tw * Ch;
    float R[9], tt[3], KR[9], K[9], Ki[9], o9[9], o3[3];
    for (int i = 0; i < 9; i++) { R[i] = (float)l.R[i]; K[i] = S.K[i]; Ki[i] = S.Ki[i]; }
    for (int i = 0; i < 3; i++) tt[i] = (float)l.t[i];
    mul3f(K, R, KR);
    mul3f(KR, Ki, o9);         // PRE_KRKiTll
    mulv3f(K, tt, o3);         // PRE_KtTll
    double a2[2];
    affFromTo(p_exp_h, p_exp_t, S.affd[h][0], S.affd[h][1], S.affd[t][0], S.affd[t][1], a2);
    for (int i = 0; i < 9; i++) o[i] = o9[i];
    for (int i = 0; i < 3; i++) o[9 + i] = o3[i];
    o[24] = (float)a2[0]; o[25] = (float)a2[1];
    o[26] = S.zb[h];
  }
  // adHTdeltaF[h + t*nf] = delta_h^T adHostF + delta_t^T adTargetF  (float arithmetic)
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int e = tid + NT * u;
    if (e < nf * nf * 8) {
      const int idx = e >> 3, h = idx % nf, t = idx / nf;
      float shh = 0, stt = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        shh += S.dlt[h][i] * p_ah[u][i];
        stt += S.dlt[t][i] * p_at[u][i];
      }
      const_cast<float*>(B.t_adHTdelta)[e] = shh + stt;
    }
  }

  // ---- the loop's break test (doStepFromBackup's return value) and bookkeeping
  if (tid == 0) {
    float sumA = 0, sumB = 0, sumT = 0, sumR = 0;
    for (int f = 0; f < nf; f++) {
      const double* st = S.step[f];
      sumA += st[6] * st[6];
      sumB += st[7] * st[7];
      sumT += st[0] * st[0] + st[1] * st[1] + st[2] * st[2];
      sumR += st[3] * st[3] + st[4] * st[4] + st[5] * st[5];
    }
    float sumNID = 0, numID = 0;
    if (g) for (int r = 0; r < nranks; r++) { sumNID += g[r * rstride + cap + 2]; numID += g[r * rstride + cap + 3]; }
    else {
      sumNID = (S.nid[0] + S.nid[1]) + (S.nid[2] + S.nid[3]);
      if (NT > 256) sumNID += (S.nid[4] + S.nid[5]) + (S.nid[6] + S.nid[7]);
      numID = (float)B.np;
    }
    sumA /= nf; sumB /= nf; sumR /= nf; sumT /= nf;
    sumNID /= numID;
    const bool canbreak = sqrtf(sumA) < 0.0005 * 1.2f && sqrtf(sumB) < 0.00005 * 1.2f && sqrtf(sumR) < 0.00005 * 1.2f && sqrtf(sumT) * sumNID < 0.00005 * 1.2f;
    O.iterations = p_its + 1;
    O.resInA = (int)p_nres;
    if (gated) O.canbreak = (stop_on_convergence && canbreak && iteration >= 1) ? 1 : 0;
    else if (stop_on_convergence && canbreak && iteration >= 1) { O.phase = 1; Bw.finished = 1; }
  }
}


// SOLVER_STEPMOMENTUM / SOLVER_MOMENTUM, between the solve and the step of one iteration (one workgroup per window, after x is in B.sol):
//   incDirChange = (1e-20 + previousX . lastX) / (1e-20 + |previousX| |lastX|), previousX = lastX        FullSystemOptimize.cpp:933-934
//   stepsize <- (exp(1.4 incDirChange) stepsize^3)^(1/4), reset to 1 on a direction change, kept in [0.25, 2]   :936-948 (STEPMOMENTUM only)
//   x_backup = the previous iteration's x (frames' / calibration's step_backup = -x_backup), zeros in the first  :311-345
__global__ __launch_bounds__(128) void k_ba_opt_momentum(const BaDev* __restrict__ wins) {
  const BaDev& B = wins[blockIdx.y];
  if (ba_finished(B)) return;
  BaOptDev& O = *B.opt;
  const int n = B.n, tid = threadIdx.x;
  const double* x = B.sol + 3 * ((size_t)n * n + n);
  __shared__ double xs[72], ps[72];
  if (tid < n) { xs[tid] = x[tid]; ps[tid] = O.previousX[tid]; }
  __syncthreads();
  if (tid == 0) {
    double dot = 0, n0 = 0, n1 = 0;
    for (int i = 0; i < n; i++) { dot += ps[i] * xs[i]; n0 += ps[i] * ps[i]; n1 += xs[i] * xs[i]; }
    const double incDirChange = (1e-20 + dot) / (1e-20 + sqrt(n0) * sqrt(n1));
    float stepsize = O.stepsize;
    if (isfinite(incDirChange) && (B.solver_mode & SOLVER_STEPMOMENTUM)) {
      const float newStepsize = (float)exp(incDirChange * 1.4);
      if (incDirChange < 0 && stepsize > 1) stepsize = 1;
      stepsize = sqrtf(sqrtf(newStepsize * stepsize * stepsize * stepsize));
      if (stepsize > 2) stepsize = 2;
      if (stepsize < 0.25f) stepsize = 0.25f;
    }
    O.stepsize = stepsize;
  }
  if (tid < n) {
    O.x_backup[tid] = O.iterations != 0 ? ps[tid] : 0.0;
    O.previousX[tid] = xs[tid];
  }
}

__global__ __launch_bounds__(256) void k_ba_opt_step(const BaDev* __restrict__ wins, const float* __restrict__ gathered, int nranks, int cap, int iteration, int last,
                                                     int stop_on_convergence, float stepsize, int unfused_parts, const float* __restrict__ sums, int sums_stride) {
  BaDev& Bw = const_cast<BaDev&>(wins[blockIdx.y]);   // written: calibration scalars, finished
  const BaDev B = Bw;                                   // read once: later loads need not be repeated after the stores below
  if (ba_finished_lin(B)) return;
  __shared__ OptStepSmem S;
  opt_step_body<256>(Bw, B, S, gathered, nranks, cap, iteration, last, stop_on_convergence, stepsize, unfused_parts,
                     sums ? sums + (size_t)blockIdx.y * sums_stride : nullptr, (B.np + BA_BLOCK - 1) / BA_BLOCK,
                     B.sol + 3 * ((size_t)B.n * B.n + B.n), B.accum[acc_off_nres(B.nf)], blockIdx.y, gridDim.y);
}

// The energy gate of one GN iteration (FullSystemOptimize.cpp:961-990) for every window, after the trial linearisation:
//   which 0: the loop's start — lastEnergy, setNewFrameEnergyTH, lastEnergyL, lastEnergyM of the uploaded state (:895-906)
//   which 1: newEnergy (+ threshold), newEnergyL, newEnergyM; accepted when their sum is below the last one: the new values become the
//            last ones, lambda *= 0.25, gate = 1 (applyRes follows); otherwise loadSateBackup — states, calibration and tables go back —,
//            lambda *= 1e2, gate = 2 (the points are restored and the state re-linearised by the conditional kernels that follow)
//   which 2: (rejected windows only) the energies of the re-linearisation at the restored state become the last ones (:983-985)
// After which 1 (accepted) / which 2 the recorded break test ends the window's loop.  lpart: k_ba_lenergy's per-workgroup partials.
// Sharded windows: gathered = every rank's k_ba_opt_pack record ([nranks][nwin][opt_pack_floats(pcap)]) — the newest frame's energies, the
// energy of the linearisation and the point part of calcLEnergy are then taken from there, rank by rank, so every rank takes the same decision.
__global__ __launch_bounds__(256) void k_ba_opt_gate(const BaDev* __restrict__ wins, const float* __restrict__ lpart, int lstride, int which, int stop_on_convergence,
                                                     const float* __restrict__ gathered = nullptr, int nranks = 1, int pcap = 0) {
  BaDev& Bw = const_cast<BaDev&>(wins[blockIdx.y]);
  const BaDev B = Bw;
  if (ba_finished_lin(B)) return;
  BaOptDev& O = *B.opt;
  if (which == 2 && O.phase == 3) {                          // an accepted step that also fired the break test: its applyRes (the conditional
    if (threadIdx.x == 0) { O.phase = 2; Bw.finished = 2; } //   kernel between which 1 and which 2) has run — only now is the window finished
    return;
  }
  if (which == 2 && O.gate != 2) return;
  const int tid = threadIdx.x, nf = B.nf, n = B.n;
  __shared__ unsigned hist[256], sh[6];
  __shared__ int s_cnt[4];
  __shared__ double s_esum[4], s_term[72], s_delta[72];
  __shared__ float s_lp[1024];
  constexpr int kStage = 8192;
  __shared__ float s_en[kStage];
  // ---- energy of the linearisation (k_ba_linearize: one partial per 256 residuals) and the newest frame's threshold
  const int pf = opt_pack_floats(pcap);
  const size_t grs = (size_t)gridDim.y * pf;
  const float* g = gathered ? gathered + (size_t)blockIdx.y * pf : nullptr;
  OptEnergies en{g, grs, g ? nranks : 1, g ? pcap : max(B.nr - O.newest_first, 0), &B, O.newest_first};
  const int cap = en.cap * en.ranks();                        // values that enter the quantile, rank after rank
  const bool staged = cap <= kStage;
  int cnt = 0;
  for (int j = tid; j < cap; j += 256) { const float e = en.at(j / en.cap, j % en.cap); if (staged) s_en[j] = e; cnt += e >= 0 ? 1 : 0; }
  double esum = 0;
  if (g) { if (tid < nranks) { double er; __builtin_memcpy(&er, g + tid * grs + pcap, 8); esum = er; } }
  else for (int b = tid; b < (B.nr + BA_BLOCK - 1) / BA_BLOCK; b += 256) esum += B.e_part[b];
  esum = wave_sum(esum);
  cnt = (int)wave_sum((float)cnt);
  if ((tid & 63) == 0) { s_esum[tid >> 6] = esum; s_cnt[tid >> 6] = cnt; }
  // calcLEnergy's partials and the operands of calcMEnergy (EnergyFunctional.cpp:344-442), fetched before the quantile
  const int nlp = B.nchunks + (B.np + BA_BLOCK - 1) / BA_BLOCK;
  const float* lp = lpart + (size_t)blockIdx.y * lstride;
  for (int b = tid; b < nlp && b < 1024; b += 256) s_lp[b] = lp[b];
  if (tid < n) s_delta[tid] = tid < 4 ? O.calib_value[tid] - O.calib_zero[tid] : B.t_prior[nf * 16 + 8 + (tid - 4)];   // getStitchedDeltaF (:1021-1032)
  __syncthreads();
  if (staged) { en.g = s_en; en.rstride = cap; en.nranks = 1; en.cap = cap; }
  else if (g) { /* read in place: opt_select walks the ranks */ }
  const int M = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
  float th = 12 * 12 * 8;
  if (M > 0) {
    const int nth = (int)(0.7f * M);
    const float q = opt_select<256>(en, nth, hist, sh);
    const float nthElement = sqrtf(q);
    th = nthElement * 1.5f;
    th = 26.0f * 0.5f + th * (1 - 0.5f);
    th = th * th;
    th *= 1.0f * 1.0f;
  }
  if (tid < n) {                                              // delta^T (2 bM + HM delta), row by row like the host loop
    double s2 = 0;
    for (int k = 0; k < n; k++) s2 += B.t_HM[(size_t)tid * n + k] * s_delta[k];
    s_term[tid] = s_delta[tid] * (2 * B.t_bM[tid] + s2);
  }
  __syncthreads();
  if (tid != 0) return;
  const double newE = (s_esum[0] + s_esum[1]) + (s_esum[2] + s_esum[3]);
  Bw.t_frameTH[nf - 1] = th;
  O.frameTH_new = th;
  double EL = 0;
  for (int f = 0; f < nf; f++) for (int i = 0; i < 8; i++) { const double dp = B.t_prior[nf * 8 + f * 8 + i]; EL += dp * B.t_prior[f * 8 + i] * dp; }
  { float s = 0; for (int i = 0; i < 4; i++) { const float cd = B.t_cdelta[i]; s += cd * (float)B.t_prior[nf * 16 + i] * cd; } EL += s; }
  { float Ept = 0; if (g) { for (int r = 0; r < nranks; r++) Ept += g[r * grs + pcap + 4]; } else for (int b = 0; b < nlp; b++) Ept += b < 1024 ? s_lp[b] : lp[b]; EL += Ept; }
  double EM = 0;
  for (int i = 0; i < n; i++) EM += s_term[i];
  if (which != 1) {                                           // the loop's start, or the state a rejected step went back to
    O.lastEnergy = newE; O.lastEnergyL = EL; O.lastEnergyM = EM;
    if (which == 0) { O.gate = 0; return; }
  } else if (newE + EL + EM < O.lastEnergy + O.lastEnergyL + O.lastEnergyM) {      // :969
    O.lastEnergy = newE; O.lastEnergyL = EL; O.lastEnergyM = EM;
    O.lambda *= 0.25;
    O.gate = 1;
  } else {
    O.lambda *= 1e2;
    O.gate = 2;
    // loadSateBackup: frames, calibration, and everything setPrecalcValues / setDeltaF derived from them
    for (int f = 0; f < nf; f++) for (int i = 0; i < 10; i++) O.state[f][i] = O.state_backup[f][i];
    for (int i = 0; i < 4; i++) O.calib_value[i] = O.calib_backup[i];
    float* pc = const_cast<float*>(B.t_precalc); float* ad = const_cast<float*>(B.t_adHTdelta); float* cd = const_cast<float*>(B.t_cdelta);
    double* tp = const_cast<double*>(B.t_prior);
    for (int e = 0; e < nf * nf * 27; e++) pc[e] = O.bk_precalc[e];
    for (int e = 0; e < nf * nf * 8; e++) ad[e] = O.bk_adHTdelta[e];
    for (int e = 0; e < nf * 16 + 8 + nf * 8; e++) tp[e] = O.bk_prior[e];
    for (int e = 0; e < 4; e++) cd[e] = O.bk_cdelta[e];
    Bw.fxl = O.bk_calib[0]; Bw.fyl = O.bk_calib[1]; Bw.cxl = O.bk_calib[2]; Bw.cyl = O.bk_calib[3]; Bw.fxli = O.bk_calib[4]; Bw.fyli = O.bk_calib[5];
    return;                                                   // the break test waits for the re-linearisation (which 2)
  }
  // :990 `if(canbreak && iteration >= setting_minOptIterations) break;` — after the accepted step's applyRes (:969-981 run before the break):
  // which 1 only records it (phase 3), the which-2 call that follows the conditional applyRes ends the loop; a rejected step's call (which 2) ends it directly
  if (O.canbreak) { if (which == 1) O.phase = 3; else { O.phase = 2; Bw.finished = 2; } }
}

// what the host needs back from a resident loop, one record per window (one D2H copy per batch)
struct BaOptOut {
  double state[8][10];
  double calib_value[4];
  double lastEnergy;
  float frameTH_new;
  int iterations, resInA, resInL;   // resInL: nres[0] of the loop's last accumulateLF (EnergyFunctional.cpp:241), read when the loop ends
};
__global__ __launch_bounds__(128) void k_ba_opt_release(const BaDev* __restrict__ wins, BaOptOut* __restrict__ out) {
  BaDev& B = const_cast<BaDev&>(wins[blockIdx.x]);
  BaOptDev& O = *B.opt;
  BaOptOut& R = out[blockIdx.x];
  const int tid = threadIdx.x;
  if (tid < 80) R.state[tid / 10][tid % 10] = O.state[tid / 10][tid % 10];
  if (tid >= 96 && tid < 100) R.calib_value[tid - 96] = O.calib_value[tid - 96];
  if (tid == 127) {
    R.lastEnergy = O.lastEnergy; R.frameTH_new = O.frameTH_new; R.iterations = O.iterations; R.resInA = O.resInA; R.resInL = (int)B.accum[acc_off_nres(B.nf) + 1];
    O.phase = 0;
    B.finished = 0;
  }
}

}  // namespace sdso
