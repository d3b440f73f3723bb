// The tail of one windowed-BA Gauss-Newton iteration as ONE kernel: a persistent 512-thread workgroup per window folds the
// accumulator partials, stitches the (8 nf + 4)^2 system straight into LDS, factorises it in registers, solves and runs the
// loop's host part — nothing between these phases leaves the CU.  It replaces the launches
//   k_ba_fold_all, k_ba_stitch_pre, k_ba_stitch, k_ba_solve, k_ba_opt_step       (k_ba_resub + k_ba_points_op become k_ba_resub_step)
// (seven dependent launches whose intermediate results — three stitched 68x68 blocks, the Schur pre-products, x — round-tripped
// through global memory; profiles/r02_*: 0.45 ms of a 0.83 ms step).  Reference (paths under /root/reference/src):
//   AccumulatedTopHessianSSE::stitchDoubleInternal / stitchDoubleMT   OptimizationBackend/AccumulatedTopHessian.cpp:265-337, .h:95-148
//   AccumulatedSCHessianSSE::stitchDoubleInternal / stitchDoubleMT    OptimizationBackend/AccumulatedSCHessian.cpp:106-195, .h:96-135
//   EnergyFunctional::solveSystemF (default LDLT branch)              OptimizationBackend/EnergyFunctional.cpp:838-995
//   EnergyFunctional::resubstituteF_MT / resubstituteFPt              OptimizationBackend/EnergyFunctional.cpp:272-341
//   FullSystem::backupState / doStepFromBackup (points)               FullSystem/FullSystemOptimize.cpp:207-351
//   the loop's host part: see ba_opt.hip (opt_step_body)
// Phases (B = workgroup barrier):
//   P1  all threads: fold the per-chunk top partials per (host,target) pair into LDS (fixed order), Hcc / bc over the hosts, the
//       diagonals of adTarget, bM + HM delta; then wave h: the Schur pre-products of host h, S1(h,y) = sum_j adHost(h,j) D(h,j,y)
//       (lane = column, eight running sums per lane, adHost rows through the scalar cache)                                   B
//   P2  one wave per output tile (8x8 frame-frame, frame-calibration, calibration): top-A + top-L (+ priors) + marginalisation
//       prior - Schur complement / (1 + lambda), written as ONE finished element into the LDS matrix M                       B
//   P3  SVecI, Eigen's pivot order from the scaled diagonal (ba_ldlt.h)                                                       B
//   P4  the scaled, permuted system As (mirrored from the lower triangle, like Eigen reads it)                               B
//   P5  wave 0: register-resident LDL^T + both substitutions (ba_ldlt.h)                                                      B
//   P6  x = SVecI * y back in original order, nullspace projection, x and the adjoint products xAd                           B
//   P7  opt_step_body: setNewFrameEnergyTH, frames / calibration step, SE3::exp, precalc + delta tables, break test
// The back-substitution and the points' step stay a kernel of their own over the points (k_ba_resub_step): they stream 0.6 KB per
// point (1.2 MB per window), which ONE CU pulls at ~50 GB/s — 24 us — while a grid over the points takes a few; and the break test
// does not wait for them: doStepFromBackup's sumNID is the sum of the BACKUP idepths, known before the step (summed in P1).
// S2(y,x) of the previous kernels is S1(y,x)^T here: D(i,j,k) = D(i,k,j)^T holds exactly in the reference (the same products in
// the same order) and up to float rounding in the MFMA accumulation, so the transposed pre-product stands in for the second one.
#include "ba_ldlt.h"

namespace sdso {

constexpr int TAIL_NT = 512;
constexpr int TAIL_FOLD = 1;        // fold the top partials / per-host Hcc here (else: the packed block holds folded, possibly all-reduced sums)
constexpr int TAIL_HS = 2;          // write lastHS / lastbS (sdso_ba_solve's HS, bS outputs)
constexpr int TAIL_STEP = 16;       // the loop's host part (opt_step_body), single rank
constexpr int TAIL_ORTH = 32;       // x -= P x
constexpr int TAIL_LAMBDA_DEV = 64; // lambda of the window's resident loop (energy-gated flow)
constexpr int TAIL_TOPL = 128;      // the linearised (L) top sums are not zero: read them from the packed block
constexpr int TAIL_RESUB = 256;     // with TAIL_STEP, last == 0: the points' back-substitution and step too (k_ba_resub_step's work, by this workgroup)

// LDS layout (bytes); the big regions are reused by phases that do not overlap
struct TailLds {
  static constexpr int kCol = 0;                                  // 64 d   column exchange of the factorisation (low address: immediate offsets)
  static constexpr int kVec = kCol + 64 * 8;                      // 8 vectors of 72 d: bF sv bp xp xv bMt dg tmp
  static constexpr int kPos = kVec + 8 * 72 * 8;                  // 72 i pos, 72 i perm
  static constexpr int kKeys = kPos + 2 * 72 * 4;                 // 72 u64 (P1: the nf^2 + 1 chunk offsets of the pairs)
  static constexpr int kAtd = kKeys + 72 * 8;                     // 64 pairs x 8 d: diagonals of adTarget
  static constexpr int kXad = kAtd + 64 * 8 * 8;                  // 512 f
  static constexpr int kMisc = kXad + 512 * 4;                    // 64 f: Hcc 16, bc 4, nres 2, sums 2, flags
  static constexpr int kM = kMisc + 64 * 4;                       // 68 x 70 d: the finished system (then L^T of the factorisation)
  static constexpr int kR1 = kM + LDLT_NMAX * LDLT_LD * 8;        // union: {accA 64 x 92 f | S1 64 x 64 d}, As 68 x 70 d, OptStepSmem
  static constexpr int kAccA = kR1;
  static constexpr int kS1 = kAccA + 64 * 92 * 4;
  static constexpr int kG = kS1 + 64 * 64 * 8;                    // 64 pairs x 64 d: adHost, staged once (lives until the end of P6)
  static constexpr int kE = kG + 64 * 64 * 8;                     // 64 pairs x (32 + 8) f: the Schur frame-calibration sums accE, accEB
  static constexpr int kEnd1 = kE + 64 * 40 * 4;
  static constexpr int kEnd2 = kR1 + LDLT_NMAX * LDLT_LD * 8;
  static constexpr int kEnd3 = kR1 + (int)sizeof(OptStepSmem) + 16;
  static constexpr int kPre = kEnd2;                              // OptPreSmem: behind As, on the dead tail of S1 (P5 only)
  static_assert(kPre + (int)sizeof(OptPreSmem) <= kG, "the pre-wave's stage must fit between As and the adHost stage");
  static_assert(kEnd2 <= kG && kEnd3 <= kG, "As / OptStepSmem must not reach the adHost stage");
  static constexpr int kBytes = kEnd1 > kEnd2 ? (kEnd1 > kEnd3 ? kEnd1 : kEnd3) : (kEnd2 > kEnd3 ? kEnd2 : kEnd3);
};

#ifdef SDSO_TAIL_STAMPS   // diagnostic build only (make EXTRA=-DSDSO_TAIL_STAMPS, tools/dbg_tail_stamps.py): x[0..] carry cycle counts of the phases
#define TSTAMP(i) do { if (threadIdx.x == 0) tstamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ double tail_dbg_stamps[1024][12];   // with the loop's host part (TAIL_STEP): phases incl. P7 per workgroup, read by sdso_dbg_tail_stamps
#else
#define TSTAMP(i) do { } while (0)
#endif
typedef const double __attribute__((address_space(4)))* tail_cdptr;   // uniform read-only data through the scalar cache

// element (r,c) of the 13x13 AccumulatorApprox matrix from its 91 packed sums, any address space
template <class P> __device__ __forceinline__ double tail_acc13(P p, int r, int c) { return (double)p[acc13_index(r, c)]; }

// NFT > 0 fixes the keyframe count at compile time (the full window of 8: every loop over hosts / targets unrolls, so the loads of all
// its trips are in flight together and independent dependency chains interleave); NFT = 0: runtime nf.
template <int NFT>
__global__ __launch_bounds__(TAIL_NT) void k_ba_tail(const BaDev* __restrict__ wins, double lambda, int flags, int iteration, int last,
                                                     int stop_on_convergence) {
  BaDev& Bw = const_cast<BaDev&>(wins[blockIdx.x]);
  const BaDev B = Bw;
  if (ba_finished_lin(B)) return;
  __shared__ __attribute__((aligned(16))) char tail_smem[TailLds::kBytes];
  double* col = (double*)(tail_smem + TailLds::kCol);
  double* vec = (double*)(tail_smem + TailLds::kVec);
  double *bF = vec, *sv = vec + 72, *bp = vec + 144, *xp = vec + 216, *xv = vec + 288, *bMt = vec + 360, *dg = vec + 432, *tmpv = vec + 504;
  int* pos = (int*)(tail_smem + TailLds::kPos);
  unsigned long long* keys = (unsigned long long*)(tail_smem + TailLds::kKeys);
  double* atd = (double*)(tail_smem + TailLds::kAtd);
  float* xAd_s = (float*)(tail_smem + TailLds::kXad);
  float* misc = (float*)(tail_smem + TailLds::kMisc);
  double* M = (double*)(tail_smem + TailLds::kM);
  float* accA = (float*)(tail_smem + TailLds::kAccA);
  double* S1 = (double*)(tail_smem + TailLds::kS1);
  double* G = (double*)(tail_smem + TailLds::kG);
  float* Es = (float*)(tail_smem + TailLds::kE);
  double* As = (double*)(tail_smem + TailLds::kR1);
  OptStepSmem& OS = *(OptStepSmem*)(tail_smem + ((TailLds::kR1 + 15) & ~15));
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nf = NFT ? NFT : B.nf, nf2 = nf * nf, n = NFT ? 8 * NFT + 4 : B.n;
  float nres_f = 0.f;
#ifdef SDSO_TAIL_STAMPS
  unsigned long long tstamps[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tjk[4] = {0, 0, 0, 0};        // wave 0's tile jobs by kind
#endif
  TSTAMP(0);

  if (!ba_finished(B)) {     // (a window whose break test fired only consumes the energies of its final linearisation: P8)
  if (flags & TAIL_LAMBDA_DEV) lambda = B.opt->lambda;
  const double f = (double)1.0f / (1 + lambda);
  const double* adH = B.t_adHost;
  const double* adT = B.t_adTarget;
  const float* accum = B.accum;
  // ------------------------------------------------------------------ P1
  {
    const bool fold = (flags & TAIL_FOLD) != 0;
    // everything that does not hang on the chunk ranges is requested FIRST and parked after the fold below ("request everything, then
    // park it": written as separate loops each of these was a round trip of its own — the LDS store of a loop waits for its load — five
    // in a row; requested behind the fold they were a third round trip after its two)
    constexpr int GT = 8, ET = 5;                              // 4096 adHost doubles / 2560 accE+EB floats over 512 threads
    double gv[GT]; float ev[ET], hcc = 0.f, hpart[8];
    double av = 0.0;
#pragma unroll
    for (int u = 0; u < GT; u++) { const int e = tid + u * TAIL_NT; gv[u] = e < nf2 * 64 ? adH[e] : 0.0; }
#pragma unroll
    for (int u = 0; u < ET; u++) { const int e = tid + u * TAIL_NT; ev[u] = e < nf2 * 40 ? accum[acc_off_E(nf) + e] : 0.f; }   // accE (nf2 x 32) and accEB (nf2 x 8) are contiguous
    if (tid < nf2 * 8) av = adT[(size_t)(tid >> 3) * 64 + (tid & 7) * 9];
    if (tid < 20) {
      if (fold) {
#pragma unroll
        for (int h = 0; h < 8; h++) hpart[h] = h < nf ? B.sc_part[(size_t)h * 20 + tid] : 0.f;
      } else hcc = accum[acc_off_Hcc(nf) + tid];
    }
    float nid = 0.f;                          // sum |idepth| of the points as they stand: the break test's sumNID (doStepFromBackup sums the
    if (flags & TAIL_STEP)                    //   BACKUP values, FullSystemOptimize.cpp:262 — known before the step is taken)
      for (int p = tid; p < B.np; p += TAIL_NT) nid += fabsf(B.p_geo[p].z);
    // the fold of the top partials: a thread's trips are independent, but each is two dependent global round trips (chunk range of the
    // pair, then the partial) — 12 trips in a row were the longest part of this phase.  The chunk ranges go to LDS first, then every
    // trip's first partial is requested before any is consumed (a pair has one chunk unless it holds more than 256 residuals).
    int* pcb = (int*)(tail_smem + TailLds::kKeys);            // nf2 + 1 chunk offsets (the keys of the pivot rank are not alive yet)
    if (fold) for (int e = tid; e <= nf2; e += TAIL_NT) pcb[e] = B.pair_chunk_beg[e];
    __syncthreads();
    constexpr int FT = 12;                                   // 64 pairs x 92 sums over 512 threads
    for (int e0 = tid; e0 < nf2 * 92; e0 += FT * TAIL_NT) {
      double first[FT];                                      // (top_part is f64: a pair's chunks are added in f64, rounded to float once)
      int cbv[FT], cev[FT];
#pragma unroll
      for (int u = 0; u < FT; u++) {
        const int e = e0 + u * TAIL_NT;
        first[u] = 0.0; cbv[u] = 0; cev[u] = 0;
        if (e < nf2 * 92) {
          const int pair = e / 92, k = e - pair * 92;
          if (fold) {
            cbv[u] = pcb[pair]; cev[u] = pcb[pair + 1];
            if (cbv[u] < cev[u]) first[u] = B.top_part[(size_t)cbv[u] * 92 + k];
          } else if (k < 91) first[u] = (double)accum[acc_off_topA(nf) + (size_t)pair * 91 + k];
        }
      }
#pragma unroll
      for (int u = 0; u < FT; u++) {
        const int e = e0 + u * TAIL_NT;
        if (e < nf2 * 92) {
          double s = first[u];
          const int k = e % 92;
          for (int ck = cbv[u] + 1; ck < cev[u]; ck++) s += B.top_part[(size_t)ck * 92 + k];
          accA[e] = (float)s;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < GT; u++) { const int e = tid + u * TAIL_NT; if (e < nf2 * 64) G[e] = gv[u]; }
#pragma unroll
    for (int u = 0; u < ET; u++) { const int e = tid + u * TAIL_NT; if (e < nf2 * 40) Es[e] = ev[u]; }
    if (tid < nf2 * 8) atd[tid] = av;
    if (tid < 20) {
      if (fold) { hcc = 0.f; for (int h = 0; h < nf; h++) hcc += hpart[h]; }     // (the float sum k_ba_fold_hcc leaves in the packed block: both paths give the same bits)
      misc[tid] = hcc;
    }
    if (tid >= 64 && tid < 64 + 4 * 72) {     // bM_top = bM + HM * delta   (EnergyFunctional.cpp:870): four threads per row, every load of a
      const int i = (tid - 64) >> 2, q = tid & 3;   //   thread in flight together; their partial sums are added in a fixed order
      const double* delta = B.t_prior + nf * 16 + 4;
      double part = 0;
      if (i < n) {
        const double* hm = B.t_HM + (size_t)i * n;
        double hv[17], dl[17];
#pragma unroll
        for (int u = 0; u < 17; u++) { const int k = q + 4 * u; hv[u] = k < n ? hm[k] : 0.0; dl[u] = k < n ? delta[k] : 0.0; }
#pragma unroll
        for (int u = 0; u < 17; u++) part += hv[u] * dl[u];
      }
      const double p1 = __shfl_xor(part, 1, 64);
      part += p1;
      const double p2 = __shfl_xor(part, 2, 64);
      part += p2;
      if (i < n && q == 0) bMt[i] = B.t_bM[i] + part;
    }
    nid = wave_sum(nid);
    if (lane == 0) misc[24 + wv] = nid;
    __syncthreads();
    TSTAMP(1);
    // S1(h,y)[a][c], h = this wave, lane = (y,c): eight sums per lane, j outer / m inner like k_ba_stitch_pre.  The rows of adHost(h,j)
    // are wave-uniform LDS reads (broadcast); the eight D values of a (j, lane) are requested before the FMAs of the previous j finish.
    for (int h = wv; h < nf; h += TAIL_NT / 64) {
      const int y = lane >> 3;
      double w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      const float* accD = accum + acc_off_D(nf);
      const float* Dbase = accD + (size_t)(h + nf2 * (y < nf ? y : 0)) * 64 + (lane & 7);
      float d[8], dn[8];
#pragma unroll
      for (int m = 0; m < 8; m++) d[m] = Dbase[m * 8];
      for (int j = 0; j < nf; j++) {
        if (j + 1 < nf) {
          const float* Dp = Dbase + (size_t)nf * (j + 1) * 64;
#pragma unroll
          for (int m = 0; m < 8; m++) dn[m] = Dp[m * 8];
        }
        const double* L = G + (size_t)(h + nf * j) * 64;
#pragma unroll
        for (int a = 0; a < 8; a++) {
#pragma unroll
          for (int m = 0; m < 8; m++) w[a] = __builtin_fma(L[a * 8 + m], (double)d[m], w[a]);
        }
#pragma unroll
        for (int m = 0; m < 8; m++) d[m] = dn[m];
      }
      if (y < nf) {
#pragma unroll
        for (int a = 0; a < 8; a++) S1[(size_t)(h + nf * y) * 64 + a * 8 + (lane & 7)] = w[a];
      }
    }
  }
  __syncthreads();
  TSTAMP(2);
  if (flags & TAIL_FOLD) {                  // nres[0] (the packed block's slot too: sdso_ba_optimize reports it), nres[1] = 0
    if (tid == 0) {
      float s = 0.f;
      for (int p = 0; p < nf2; p++) s += accA[p * 92 + 91];
      misc[20] = s; misc[21] = 0.f;
      B.accum[acc_off_nres(nf)] = s; B.accum[acc_off_nres(nf) + 1] = 0.f;
    }
  } else if (tid == 0) { misc[20] = accum[acc_off_nres(nf)]; misc[21] = accum[acc_off_nres(nf) + 1]; }

  // ------------------------------------------------------------------ P2: one wave per output tile
  {
    const int a = lane >> 3, c = lane & 7;
    const bool topL = (flags & TAIL_TOPL) != 0;
    const float* accL = accum + acc_off_topL(nf);
    const float* accD = accum + acc_off_D(nf);
    const float* accE = Es;
    const float* accEB = Es + nf2 * 32;
    const double* prior = B.t_prior;
    const size_t blk = (size_t)n * n + n;
    double* lastHS = B.sol + 3 * blk + n;
    double* lastbS = lastHS + (size_t)n * n;
    const bool wr_hs = (flags & TAIL_HS) != 0;
    auto at = [&](int p, int q, int d) { return atd[(p + nf * q) * 8 + d]; };
    const double* adHs = G;                             // adHost from its LDS stage
    int idxA[8];                                       // packed index of A8[a][nn], nn = 0..7 (row a of the frame block)
#pragma unroll
    for (int nn = 0; nn < 8; nn++) idxA[nn] = acc13_index(4 + a, 4 + nn);
    // element (i, j) of the finished system from its three parts
    auto finish = [&](int i, int j, double oA, double oL, double oS, double hm) {
      double v = oL + hm + oA;                              // HFinal_top = HL + HM + HA        (:906)
      if (wr_hs) lastHS[(size_t)i * n + j] = v - oS;       // :909
      if (i == j) v *= (1 + lambda);                        // :914-916
      v -= oS * f;                                          // :918
      M[i * LDLT_LD + j] = v;
    };
    // top part of a frame-frame tile from 91-sum blocks `acc` (LDS or global): the expressions of k_ba_stitch
    auto top_ff = [&](auto acc, int x, int y, bool isL) -> double {
      double out = 0;
      if (x == y) {
#pragma unroll
        for (int t = 0; t < nf; t++) {                      // H[h,h] += AH A AH^T over the targets of host x
          const int aidx = x + nf * t;
          const double* AH = adHs + (size_t)aidx * 64;
          auto ap = acc + (size_t)aidx * (isL ? 91 : 92);
          double ra[8], rc[8];
#pragma unroll
          for (int q = 0; q < 8; q++) { ra[q] = AH[a * 8 + q]; rc[q] = AH[c * 8 + q]; }
          double tv = 0;                                    // T[a][c] = sum_nn A8[a][nn] AH[c][nn]: this lane's element of A AH^T
#pragma unroll
          for (int nn = 0; nn < 8; nn++) tv += (double)ap[idxA[nn]] * rc[nn];
          double s = 0;                                     // (AH T)[a][c]: column c of T sits on the lanes (mm, c)
#pragma unroll
          for (int mm = 0; mm < 8; mm++) s += ra[mm] * __shfl(tv, mm * 8 + c, 64);
          out += s;
        }
#pragma unroll
        for (int h = 0; h < nf; h++)                        // H[t,t] += AT A AT^T over the hosts of target x
          out += at(h, x, a) * tail_acc13(acc + (size_t)(h + nf * x) * (isL ? 91 : 92), 4 + a, 4 + c) * at(h, x, c);
        {                                                   // H[h,t] += AH A AT^T of the pair (x,x)
          const double* AH = adHs + (size_t)(x + nf * x) * 64;
          auto ap = acc + (size_t)(x + nf * x) * (isL ? 91 : 92);
          double s = 0;
#pragma unroll
          for (int mm = 0; mm < 8; mm++) s += AH[a * 8 + mm] * tail_acc13(ap, 4 + mm, 4 + c);
          out += s * at(x, x, c);
        }
      } else {
        const double* AH1 = adHs + (size_t)(x + nf * y) * 64;
        auto ap1 = acc + (size_t)(x + nf * y) * (isL ? 91 : 92);
        const double* AH2 = adHs + (size_t)(y + nf * x) * 64;
        auto ap2 = acc + (size_t)(y + nf * x) * (isL ? 91 : 92);
        double s1 = 0, s2 = 0;
#pragma unroll
        for (int mm = 0; mm < 8; mm++) { s1 += AH1[a * 8 + mm] * tail_acc13(ap1, 4 + mm, 4 + c); s2 += AH2[c * 8 + mm] * tail_acc13(ap2, 4 + mm, 4 + a); }
        const double m_lo = x < y ? s1 * at(x, y, c) : s2 * at(y, x, a);      // M(lo,hi) first, then M(hi,lo)^T: the order the CPU adds them in
        const double m_hi = x < y ? s2 * at(y, x, a) : s1 * at(x, y, c);
        out = m_lo + m_hi;
      }
      return out;
    };
    auto top_fc = [&](auto acc, int x, bool isL) -> double {   // lanes (a, c < 4): H[frame x row a][calib c]; c == 4: b
      double hv = 0;
      if (c < 5) {
        const int colx = c < 4 ? c : 12;
#pragma unroll
        for (int k = 0; k < 2 * nf; k++) {
          const int h = k < nf ? x : k - nf, t = k < nf ? k : x;
          auto ap = acc + (size_t)(h + nf * t) * (isL ? 91 : 92);
          double s = 0;
          if (k < nf) {
            const double* Am = adHs + (size_t)(h + nf * t) * 64;
#pragma unroll
            for (int kk = 0; kk < 8; kk++) s += Am[a * 8 + kk] * tail_acc13(ap, 4 + kk, colx);
          } else s = at(h, t, a) * tail_acc13(ap, 4 + a, colx);      // adTarget is diagonal
          hv += s;
        }
      }
      return hv;
    };
    // ---- block row x of the system on wave x (round 5; rounds 3-4 ran one JOB per tile and wave, each rebuilding its operands through
    // run-time acc13 index arithmetic and cross-row shuffles: 64 k of the kernel's 156 k cycles).  Lane (a, c) owns element (a, c) of every
    // tile (x, y), y = 0 .. nf - 1, of the frame-calibration strip (c < 5) and keeps, for the whole row:
    //   W1 = (AH_p F_p)[a][c], p = (host x, target y): row a of adHost(p) times column c of the pair's frame block — the operand of
    //        H[x,y] += AH F AT^T (x != y), of the pair's share of H[x,x] += AH F AH^T (its row a sits in the lane's own group of
    //        eight: ds_swizzle broadcasts, no LDS memory round trip) and, with the calibration / residual columns, of H[x,c] += AH X, b[x] += AH r;
    //   W2 = (AH_q F_q)[c][a], q = (host y, target x): the transposed partner H[y,x]^T that the reference adds into the same block
    //        (AccumulatedTopHessian.h:134-147);
    //   the target-side terms AT F AT^T, AT X (adTarget is diagonal) of pair q, and the Schur terms of the same pairs.
    // Index tables into the 91 packed sums are per-lane constants (registers); the next tile's global operands (HM element, the eight accD
    // values of its AT D AT term) are requested before the current tile's arithmetic.
    // value of lane 8 (lane / 8) + K of the lane's group of eight (ds_swizzle_b32, bit mode: and 0x18, or K, inside each half of the wave;
    // the LDS crossbar without a memory access — gfx9 has no DPP8)
    auto bcast8 = [](double v, auto kk) {
      constexpr int pat = 0x18 | (decltype(kk)::value << 5);
      const unsigned long long u = __double_as_longlong(v);
      const unsigned lo = (unsigned)__builtin_amdgcn_ds_swizzle((int)(unsigned)u, pat), hi = (unsigned)__builtin_amdgcn_ds_swizzle((int)(unsigned)(u >> 32), pat);
      return __longlong_as_double(((unsigned long long)hi << 32) | lo);
    };
    if (wv < nf) {
      const int x = wv;
      const int colx = c < 4 ? c : 12;
      int iFc[8], iFa[8], iX[8];
#pragma unroll
      for (int nn = 0; nn < 8; nn++) { iFc[nn] = acc13_index(4 + nn, 4 + c); iFa[nn] = acc13_index(4 + nn, 4 + a); iX[nn] = acc13_index(4 + nn, colx); }
      const int iFac = acc13_index(4 + a, 4 + c), iXa = acc13_index(4 + a, colx);
      double atx_a[8];                                           // at(i, x, a): adTarget diagonal of pair (host i, target x), row a
#pragma unroll
      for (int i = 0; i < 8; i++) atx_a[i] = i < nf ? at(i, x, a) : 0.0;
      double dgA = 0, dgS = 0, hvA = 0, hvS = 0;                 // running sums of the diagonal tile (top, Schur) and of the strip (top, Schur)
      double hmDiag = 0;                                         // the diagonal tile's element of HM (the tile is finished after the row)
      double hmN = B.t_HM[(size_t)(4 + x * 8 + a) * n + (4 + 0 * 8 + c)];
      float dvN[8];
#pragma unroll
      for (int i = 0; i < 8; i++) dvN[i] = i < nf ? accD[(size_t)(i + nf * x + nf2 * 0) * 64 + lane] : 0.f;
#pragma unroll
      for (int y = 0; y < 8; y++) {
        if (y < nf) {
          const double hm = hmN;
          float dv[8];
#pragma unroll
          for (int i = 0; i < 8; i++) dv[i] = dvN[i];
          if (y + 1 < nf) {
            hmN = B.t_HM[(size_t)(4 + x * 8 + a) * n + (4 + (y + 1) * 8 + c)];
#pragma unroll
            for (int i = 0; i < 8; i++) if (i < nf) dvN[i] = accD[(size_t)(i + nf * x + nf2 * (y + 1)) * 64 + lane];
          }
          const int pp = x + nf * y, qq = y + nf * x;
          const double* AHp = adHs + (size_t)pp * 64;
          const float* ap = accA + (size_t)pp * 92;
          const float* aq = accA + (size_t)qq * 92;
          double AHa[8], AHc[8];
#pragma unroll
          for (int m = 0; m < 8; m++) { AHa[m] = AHp[a * 8 + m]; AHc[m] = AHp[c * 8 + m]; }
          double W1 = 0, V1 = 0, sd = 0;
#pragma unroll
          for (int m = 0; m < 8; m++) W1 += AHa[m] * (double)ap[iFc[m]];
          if (c < 5) {
#pragma unroll
            for (int m = 0; m < 8; m++) V1 += AHa[m] * (double)ap[iX[m]];
          }
          hvA += V1;
          // the pair's share of H[x,x] += AH F AH^T: sum_m W1[a][m] AH[c][m]
          dgA += bcast8(W1, std::integral_constant<int, 0>()) * AHc[0];
          dgA += bcast8(W1, std::integral_constant<int, 1>()) * AHc[1];
          dgA += bcast8(W1, std::integral_constant<int, 2>()) * AHc[2];
          dgA += bcast8(W1, std::integral_constant<int, 3>()) * AHc[3];
          dgA += bcast8(W1, std::integral_constant<int, 4>()) * AHc[4];
          dgA += bcast8(W1, std::integral_constant<int, 5>()) * AHc[5];
          dgA += bcast8(W1, std::integral_constant<int, 6>()) * AHc[6];
          dgA += bcast8(W1, std::integral_constant<int, 7>()) * AHc[7];
          // ... and of the Schur complement's H[x,x] += sum_k S1(x,k) AH(x,k)^T (AccumulatedSCHessian.cpp:151-171)
          {
            const double* s1 = S1 + (size_t)pp * 64 + a * 8;
#pragma unroll
            for (int m = 0; m < 8; m++) sd += s1[m] * AHc[m];
            dgS += sd;
          }
          // frame-calibration strip, Schur side: H[x,c] += AH E, b[x] += AH EB of pair p; AT E, AT EB of pair q
          if (c < 5) {
            double s = 0;
#pragma unroll
            for (int m = 0; m < 8; m++) s += AHa[m] * (double)(c < 4 ? accE[(size_t)pp * 32 + m * 4 + c] : accEB[(size_t)pp * 8 + m]);
            hvS += s;
          }
          const double at_pc = at(x, y, c);                     // adTarget diagonal of pair p at c
          const double at_qa = atx_a[y], at_qc = at(y, x, c);   // ... of pair q at a, c
          // target-side terms of pair q = (host y, target x) into the diagonal tile and the strip
          dgA += at_qa * (double)aq[iFac] * at_qc;
          if (c < 5) { hvA += at_qa * (double)aq[iXa]; hvS += at_qa * (double)(c < 4 ? accE[(size_t)qq * 32 + a * 4 + c] : accEB[(size_t)qq * 8 + a]); }
          // the tile's AT D AT term
          double dS = 0;
#pragma unroll
          for (int i = 0; i < 8; i++) if (i < nf) dS += atx_a[i] * (double)dv[i] * at(i, y, c);
          if (y != x) {
            const double* AHq = adHs + (size_t)qq * 64;
            double W2 = 0;
#pragma unroll
            for (int m = 0; m < 8; m++) W2 += AHq[c * 8 + m] * (double)aq[iFa[m]];
            const double z1 = W1 * at_pc, z2 = W2 * at_qa;
            const double oA = x < y ? z1 + z2 : z2 + z1;          // M(lo,hi) first, then M(hi,lo)^T: the order the CPU adds them in
            const double oL = topL ? top_ff(accL, x, y, true) : 0.0;
            double oS = S1[(size_t)pp * 64 + lane] * at_pc;
            oS += at_qa * S1[(size_t)qq * 64 + c * 8 + a];
            oS += dS;
            finish(4 + x * 8 + a, 4 + y * 8 + c, oA, oL, oS, hm);
          } else {
            // the diagonal tile is finished after the row: keep its pair term, Schur pair terms and the HM element
            dgA += W1 * at_pc;
            dgS += S1[(size_t)pp * 64 + lane] * at_pc + at_qa * S1[(size_t)qq * 64 + c * 8 + a] + dS;
            hmDiag = hm;
          }
        }
      }
      {
        double oL = topL ? top_ff(accL, x, x, true) : 0.0;
        if (a == c) oL += prior[x * 8 + a];
        finish(4 + x * 8 + a, 4 + x * 8 + c, dgA, oL, dgS, hmDiag);
      }
      {   // the strip: H[x, calibration] (c < 4) and b[x] (c == 4)
        double hL = topL ? top_fc(accL, x, true) : 0.0;
        const int i = 4 + x * 8 + a;
        if (c < 4) {
          const double hmic = B.t_HM[(size_t)i * n + c], hmci = B.t_HM[(size_t)c * n + i];
          double v = hL + hmic + hvA;
          if (wr_hs) { lastHS[(size_t)i * n + c] = v - hvS; lastHS[(size_t)c * n + i] = (hL + hmci + hvA) - hvS; }
          v -= hvS * f;
          M[i * LDLT_LD + c] = v;
          M[c * LDLT_LD + i] = (hL + hmci + hvA) - hvS * f;
        } else if (c == 4) {
          hL += prior[x * 8 + a] * prior[nf * 8 + x * 8 + a];
          const double v = hL + bMt[i] + hvA - hvS;            // bFinal = bL + bM_top + bA - b_sc      (:907)
          bF[i] = v;
          if (wr_hs) lastbS[i] = v;
        }
      }
    }
    if (wv == (nf < TAIL_NT / 64 ? nf : 0) && lane < 20) {       // calibration block and its b: a wave without a frame row, wave 0 for the full window
      const int r = lane < 16 ? lane >> 2 : lane - 16, colc = lane < 16 ? (lane & 3) : 12;
      const int ic = acc13_index(r, colc);
      double sA = 0, sL = 0;
#pragma unroll 16
      for (int p = 0; p < nf2; p++) sA += (double)accA[p * 92 + ic];
      if (topL) for (int p = 0; p < nf2; p++) sL += (double)accL[(size_t)p * 91 + ic];
      const double sS = (double)misc[lane];                  // Hcc (16) then bc (4)
      if (lane < 16) {
        if (r == colc) sL += prior[nf * 16 + r];
        finish(r, colc, sA, sL, sS, B.t_HM[(size_t)r * n + colc]);
      } else {
        sL += prior[nf * 16 + r] * (double)B.t_cdelta[r];
        const double v = sL + bMt[r] + sA - sS;
        bF[r] = v;
        if (wr_hs) lastbS[r] = v;
      }
    }
  }
  __syncthreads();
  TSTAMP(3);
  // ------------------------------------------------------------------ P3: SVecI (:967) and Eigen's pivot order of the scaled system
  OptPreRegs<TAIL_NT> pre_regs;                // (the energies the idle wave of P5 sorts: requested here, parked behind As at the end of P4)
  if (flags & TAIL_STEP) opt_pre_request<TAIL_NT>(B, pre_regs);
  if (tid < 72) {
    double s = 0, d = 0;
    if (tid < n) { const double mii = M[tid * LDLT_LD + tid]; s = 1.0 / sqrt(mii + 10); d = s * mii * s; }
    sv[tid] = s; dg[tid] = d; bp[tid] = 0.0;
  }
  for (int e = tid; e < LDLT_NMAX * LDLT_LD; e += TAIL_NT) As[e] = 0.0;     // (accA / S1 are dead: P2 is behind a barrier)
  __syncthreads();
  ldlt_pivot_rank(dg, n, pos, keys);
  TSTAMP(4);
  // ------------------------------------------------------------------ P4: As = SVecI H SVecI, permuted; the lower triangle mirrored
  for (int e = tid; e < n * n; e += TAIL_NT) {
    const int i = e / n, j = e - i * n;
    if (i >= j) {
      const double v = sv[i] * M[i * LDLT_LD + j] * sv[j];
      const int pi = pos[i], pj = pos[j];
      As[pi * LDLT_LD + pj] = v;
      As[pj * LDLT_LD + pi] = v;
    }
  }
  if (tid < n) bp[pos[tid]] = sv[tid] * bF[tid];
  if (flags & TAIL_STEP) opt_pre_park<TAIL_NT>(B, *(OptPreSmem*)(tail_smem + TailLds::kPre), pre_regs);
  __syncthreads();
  TSTAMP(5);
  // ------------------------------------------------------------------ P5: wave 0 factorises and solves; wave 1 meanwhile does the part of the
  // loop's host part that does not need x (energies of the linearisation, their 70 % quantile, the energy sum)
  OptPre* pre = (OptPre*)(misc + 44);
  if (wv == 0) ldlt_solve_regs(As, bp, M /* L^T */, col, xp, n);
  else if (wv == 1 && (flags & TAIL_STEP)) opt_pre_wave(B, *(OptPreSmem*)(tail_smem + TailLds::kPre), pre, true);
  __syncthreads();
  TSTAMP(6);
  // ------------------------------------------------------------------ P6: x = SVecI * solve(...) (:976), x -= P x (:980-984, :824-826)
  if (tid < n) xv[tid] = sv[tid] * xp[pos[tid]];
  __syncthreads();
  if (flags & TAIL_ORTH) {
    if (tid < n) {
      double s = 0;
      const double* pr = B.t_P + (size_t)tid * n;
      for (int k = 0; k < n; k++) s += pr[k] * xv[k];
      tmpv[tid] = xv[tid] - s;
    }
    __syncthreads();
    if (tid < n) xv[tid] = tmpv[tid];
    __syncthreads();
  }
  {
    double* xout = B.sol + 3 * ((size_t)n * n + n);
    if (tid < n) xout[tid] = xv[tid];
    // xAd[nf*h+t] = xF(h)^T adHostF[h+nf*t] + xF(t)^T adTargetF[h+nf*t]   (:289-291), float arithmetic
    float* xAd = const_cast<float*>(B.t_xAd);
    for (int e = tid; e < nf2 * 8; e += TAIL_NT) {
      const int j = e & 7, ht = e >> 3, h = ht / nf, t = ht % nf;
      const double* AH = G + (size_t)(h + nf * t) * 64;
      float sh = 0, st = 0;                  // (adTarget is diagonal: its column j has one entry; the other products are exact zeros)
      for (int i = 0; i < 8; i++) { sh += (float)xv[4 + 8 * h + i] * (float)AH[i * 8 + j]; st += (float)xv[4 + 8 * t + i] * (i == j ? (float)atd[(h + nf * t) * 8 + j] : 0.f); }
      const float v = sh + st;
      xAd[e] = v; xAd_s[e] = v;
    }
  }
  __syncthreads();
  TSTAMP(7);
  nres_f = misc[20];
  if (tid == 0) { float t = 0.f; for (int w = 0; w < TAIL_NT / 64; w++) t += misc[24 + w]; misc[22] = 0.f; misc[23] = t; }
  TSTAMP(8);
#ifdef SDSO_TAIL_STAMPS
  __syncthreads();
  if (threadIdx.x == 0 && !(flags & TAIL_STEP)) {
    double* xo = B.sol + 3 * ((size_t)n * n + n);
    for (int i = 0; i < 8; i++) xo[i] = (double)(tstamps[i + 1] - tstamps[i]);   // stage | S1 | tiles | SVecI+order | assemble | factor+solve | x, xAd | resub
    for (int i = 0; i < 4; i++) xo[8 + i] = (double)tjk[i];
  }
#endif
  }  // !ba_finished
  // ------------------------------------------------------------------ P6b: resubstituteF_MT + the points' backupState / doStepFromBackup for this window,
  // x and xAd straight from LDS — k_ba_resub_step's pass (one launch fewer per iteration; a workgroup starts on its points the moment its
  // own x exists).  Only when this call takes the step: the window's break test has not fired (phase 1 consumes energies only).
  if ((flags & TAIL_RESUB) && !ba_finished(B) && B.opt->phase != 1) {
    const double* xc = xv;
    float a_, b_;
    if (flags & TAIL_TOPL) { for (int p = tid; p < B.np; p += TAIL_NT) resub_step_point<true>(B, p, xc, xAd_s, a_, b_); }
    else { for (int p = tid; p < B.np; p += TAIL_NT) resub_step_point<false>(B, p, xc, xAd_s, a_, b_); }
  }
  // ------------------------------------------------------------------ P7
  if (flags & TAIL_STEP) {
    __syncthreads();
    opt_step_body<TAIL_NT>(Bw, B, OS, nullptr, 1, 0, iteration, last, stop_on_convergence, 1.0f, 0, misc + 22, 1, xv, nres_f, blockIdx.x, gridDim.x, ba_finished(B) ? nullptr : (const OptPre*)(misc + 44));
#ifdef SDSO_TAIL_STAMPS
    __syncthreads();
    TSTAMP(9);
    if (threadIdx.x == 0 && blockIdx.x < 1024) {
      for (int i = 0; i < 8; i++) tail_dbg_stamps[blockIdx.x][i] = (double)(tstamps[i + 1] - tstamps[i]);
      tail_dbg_stamps[blockIdx.x][8] = (double)(tstamps[9] - tstamps[8]);
    }
#endif
  }
}

}  // namespace sdso
#ifdef SDSO_TAIL_STAMPS
extern "C" int sdso_dbg_tail_stamps(double* out, int nwin) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sdso::tail_dbg_stamps), sizeof(double) * 12 * nwin, 0, hipMemcpyDeviceToHost);
}
#endif
