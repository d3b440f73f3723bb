// Host API of the windowed bundle adjustment (C-ABI entry points sdso_ba_*).
// The window mirrors an EnergyFunctional (src/OptimizationBackend/EnergyFunctional.h:49-150):
// frames / calibration live on the host in double (ba_host.h), points and residuals live in HBM
// (ba_kernels.h).  Call-surface mapping: see include/sdso_abi.h and INTEGRATION.md.
#include "ba_kernels.hip"   // single translation unit: kernels + host API
#include "ba_solve.hip"
#include "ba_opt.hip"
#include "ba_tail.hip"
#include "ba_host.h"
#include "ba_solve_alt.hip"   // (after ba_host.h: the SOLVER_* bits and setting_solverModeDelta)
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <limits>
#include <functional>
#include <chrono>
#include <atomic>
#include <thread>

namespace sdso {

struct BaBatch;

struct BaWindowDev {
  BaDev d;                 // host copy of the device descriptor
  BaDev* d_self = nullptr; // device copy (array of 1)
  std::vector<std::pair<void*, size_t>> allocs;
  // host mirror
  HostCalib calib;
  std::vector<HostFrame> frames;
  HostTables tab;
  Dense P;
  std::vector<double> HM, bM;    // host mirror of the marginalisation prior; the MASTER copy is the device's (dt_HM / dt_bM): see hm_host_valid
  bool hm_host_valid = true;     // false: a device kernel changed the prior since the mirror was filled (sync_prior_host brings it up to date)
  double* d_marg = nullptr;      // the prior after sdso_ba_marginalize_frame_dev: marg_dim^2 + marg_dim doubles, adopted by the next window
  double* d_marg2 = nullptr;     // (the other half of the ping-pong when several frames leave at one keyframe)
  int marg_dim = 0;              // its dimension (0: none)
  std::vector<int> marg_frames;  // the window's frames that prior still covers, in order (indices into `frames`)
  bool marg_chain = false;       // the next sdso_ba_marginalize_frame_dev continues from d_marg (no sdso_ba_marginalize_points since the last one)
  bool prior_pristine = false;   // uploaded with HM = bM = NULL and untouched since: what sdso_ba_adopt_prior requires of the adopting window
  int solverMode = 0, forceAccept = 1;
  double affA = 0, affB = 0;
  std::vector<int> perm, inv;     // sorted -> original, original -> sorted
  std::vector<uint8_t> h_target;  // sorted order
  std::vector<int> h_point;       // sorted order
  std::vector<uint8_t> h_lin;     // sorted order mirror of isLinearized
  std::vector<float> h_prior;
  int nblk_res = 0, nblk_pts = 0;
  // device table blocks that are re-uploaded when frame states change
  float* dt_precalc = nullptr; float* dt_adHTdelta = nullptr; float* dt_cdelta = nullptr; float* dt_frameTH = nullptr;
  double* dt_adHost = nullptr; double* dt_adTarget = nullptr; double* dt_prior = nullptr; double* dt_HM = nullptr; double* dt_bM = nullptr; double* dt_P = nullptr;
  float* dt_xAd = nullptr;
  uint8_t* d_pflag = nullptr;
  float* d_sums = nullptr;
  BaOptDev* d_opt = nullptr;    // resident GN loop state (ba_opt.hip)
  BaOptDev h_opt;               // staging of its upload
  std::vector<double> h_prstage;  // staging of the dt_prior upload (upload_tables)
  int newest_first = 0;         // first pair-sorted residual whose target is the newest frame
  char* tbl_first = nullptr;    // the tables upload_tables refreshes are one contiguous block of the window's slab:
  size_t tbl_bytes = 0;         //   [precalc | adHTdelta | cdelta | adHost | adTarget | P | prior | BaDev], 256-byte aligned each
  float* accum_own = nullptr;   // the window's own packed accumulator block (d.accum points into the batch block while batched)
  bool in_batch = false;
  bool accumulated = false;
  bool marg_accumulated = false;  // the packed block holds the sums of the latest sdso_ba_marginalize_points (addPoint<2> + the Schur addPoint of the flagged points)
  bool has_lin_cached = false;  // some residual of the window is linearized (updated wherever h_lin changes)
  bool l_dirty = false;         // p_out's L sums (linearised / marginalised residuals) may be non-zero: the next plain Schur launch clears them
  bool j_inplace_last = false;  // the latest linearisation was the fused kernel's, written IN PLACE into EFResidual::J's slot (BaDev::jfix):
                                // sdso_ba_get_linearization reads the records from there
  // post-state of FullSystem::optimize (sdso_ba_get_post_state)
  bool post_valid = false;      // an optimize call has ended on this window
  bool hs_valid = false;        // the last solveSystemF of that call wrote lastHS / lastbS
  sdso_ba_opt_result_t last_result{0, 0, 0, 0};
  float* d_post = nullptr;      // nr x 19: projectedTo, centerProjectedTo of the closing linearisation
  int resInL = 0, resInM = 0;
};

// zeroed device buffer for a window: reuse a pooled buffer of a released window when one of a similar size exists
// (hipMalloc / hipFree of ~40 buffers cost more than the whole upload otherwise)
static int dmalloc(sdso_ctx* ctx, BaWindowDev* W, void** p, size_t bytes, bool zero = true) {
  const size_t want = ((bytes ? bytes : 16) + 255) & ~(size_t)255;
  int best = -1;
  for (int i = 0; i < (int)ctx->ba_pool.size(); i++) {
    const size_t have = ctx->ba_pool[i].second;
    if (have >= want && have <= 2 * want + 4096 && (best < 0 || have < ctx->ba_pool[best].second)) best = i;
  }
  size_t got = want;
  if (best >= 0) { *p = ctx->ba_pool[best].first; got = ctx->ba_pool[best].second; ctx->ba_pool.erase(ctx->ba_pool.begin() + best); }
  else SDSO_HIP(ctx, hipMalloc(p, want));
  if (zero) SDSO_HIP(ctx, hipMemsetAsync(*p, 0, want, ctx->stream));
  W->allocs.emplace_back(*p, got);
  return SDSO_OK;
}
// pinned host staging of the ctx (window uploads, table refreshes): grown on demand, released with the ctx's windows.  The caller
// synchronises the stream before the next reservation is written.
// Two buffers taken in turn, each with an event that marks the last copy enqueued from it (stage_commit): a caller that commits need not
// synchronise the stream — the buffer is only waited for when its turn comes again, two reservations later.  A caller that does not commit
// synchronises the stream itself before the ctx reserves again (upload_tables, opt_finish).
struct StageBuf { char* p[2] = {nullptr, nullptr}; size_t cap[2] = {0, 0}; hipEvent_t ev[2] = {nullptr, nullptr}; bool busy[2] = {false, false}; int cur = 0; };
static std::map<sdso_ctx*, StageBuf> g_stage;
static int stage_reserve(sdso_ctx* ctx, size_t bytes, char** out) {
  StageBuf& b = reg_get(g_stage, ctx);
  b.cur ^= 1;
  const int k = b.cur;
  if (b.busy[k]) { SDSO_HIP(ctx, hipEventSynchronize(b.ev[k])); b.busy[k] = false; }
  if (bytes > b.cap[k]) {
    if (b.p[k]) { SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream)); hipHostFree(b.p[k]); }
    b.p[k] = nullptr; b.cap[k] = 0;
    const size_t want = (bytes * 3 / 2 + 4095) & ~(size_t)4095;
    SDSO_HIP(ctx, hipHostMalloc((void**)&b.p[k], want));
    b.cap[k] = want;
  }
  *out = b.p[k];
  return SDSO_OK;
}
static int stage_commit(sdso_ctx* ctx) {      // everything that reads the latest reservation has been enqueued on ctx->stream
  StageBuf& b = reg_get(g_stage, ctx);
  const int k = b.cur;
  if (!b.ev[k]) SDSO_HIP(ctx, hipEventCreateWithFlags(&b.ev[k], hipEventDisableTiming));
  SDSO_HIP(ctx, hipEventRecord(b.ev[k], ctx->stream));
  b.busy[k] = true;
  return SDSO_OK;
}
static void stage_free(sdso_ctx* ctx) {
  StageBuf b;
  if (!reg_take(g_stage, ctx, b)) return;
  for (int k = 0; k < 2; k++) { if (b.ev[k]) hipEventDestroy(b.ev[k]); if (b.p[k]) hipHostFree(b.p[k]); }
}
#define DM(ptr, T, count)                                                   \
  do {                                                                      \
    void* _p = nullptr;                                                     \
    int _rc = dmalloc(ctx, W, &_p, sizeof(T) * (size_t)(count));            \
    if (_rc) return _rc;                                                    \
    ptr = (T*)_p;                                                           \
  } while (0)
#define H2D(dst, src, bytes) SDSO_HIP(ctx, hipMemcpyAsync((void*)(dst), (src), (bytes), hipMemcpyHostToDevice, ctx->stream))

static void free_window(sdso_ctx* ctx, BaWindowDev* W) {
  for (auto& a : W->allocs) {
    if (ctx->ba_pool.size() < 4096) ctx->ba_pool.push_back(a); else hipFree(a.first);
  }
  delete W;
}
struct BaLaunch {
  const BaDev* d_arr; int nwin; int max_nblk_res, max_nblk_pts, max_chunks, max_items, nf, n;
  bool any_lin;   // some window holds linearized residuals -> the mode-1 accumulation has work to do
  bool tiled;     // the windows' t_img are 4x2-tiled level-0 images (same for every window of a launch)
  bool alt;       // the windows' solverMode takes solveSystemF's SVD / orthogonalised-system branches (ba_solve_alt.hip): never the fused tail kernel
  std::vector<BaWindowDev*> Ws;   // the windows behind d_arr (host bookkeeping of a launch: BaWindowDev::l_dirty)
};
struct BaBatch {
  std::vector<int> wins;
  std::vector<BaWindowDev*> W;   // valid while the batch lives: releasing / re-uploading a member frees the batch first
  BaDev* d_arr = nullptr;
  float* d_accum = nullptr;
  BaLaunch L;
  bool materialize = true;
  bool eager_fold = false;       // sdso_ba_batch_accum_dev handed the block's address out: never defer the folds
  bool folded = true;            // the packed accumulator block holds the folded sums of the latest accumulate (false: the top partials and the
                                 // per-host Hcc / bc are still unfolded — the fused tail kernel folds them itself; ensure_folded() for anyone else)
  int exchange_mode = 0;         // sdso_ba_batch_exchange_mode: 0 all-reduce + the solve on every rank, 1 reduce-scatter by window + all-gather of x
  bool scattered = false;        // the latest sdso_ba_allreduce was the reduce-scatter: only this rank's windows hold summed accumulators
  bool keep_system = false;      // sdso_ba_batch_keep_system: the resident loop's solves also write lastHS / lastbS (37 KB per window and iteration)
};
static std::map<sdso_ctx*, BaBatch*> g_batches;
void free_optrun(sdso_ctx* ctx);    // the resident GN loop's bookkeeping (end of this file)
void free_optbufs(sdso_ctx* ctx);
int optimize_resident_single(sdso_ctx* ctx, BaWindowDev* W, int mnumOptIts, sdso_ba_opt_result_t* res);
// Dissolve the ctx's batch: every member window gets its own accumulator block back (host descriptor and its device copy),
// so later per-window calls never touch the freed batch block.
static void free_batch(sdso_ctx* ctx) {
  BaBatch* taken = nullptr;
  free_optrun(ctx);   // a resident loop over the batch ends with it
  if (!reg_take(g_batches, ctx, taken) || !taken) return;
  hipStreamSynchronize(ctx->stream);
  for (BaWindowDev* W : taken->W) {
    W->d.accum = W->accum_own;
    W->in_batch = false;
    W->accumulated = false;
    hipMemcpyAsync(W->d_self, &W->d, sizeof(BaDev), hipMemcpyHostToDevice, ctx->stream);
  }
  hipStreamSynchronize(ctx->stream);
  hipFree(taken->d_arr); hipFree(taken->d_accum);
  delete taken;
}
void release_all_windows(sdso_ctx* ctx) {
  free_batch(ctx);
  free_optbufs(ctx);
  stage_free(ctx);
  for (auto& kv : ctx->wins) free_window(ctx, kv.second);
  ctx->wins.clear();
}

// the CPU half of upload_tables: everything derived from the frame states / calibration, into the window's own staging members
// (no HIP call: safe to run for several windows on several host threads)
static void build_tables(BaWindowDev* W, bool adjoints) {
  const int nf = W->d.nf, n = W->d.n;
  buildPrecalc(W->calib, W->frames, W->tab);
  if (adjoints) { buildAdjoints(W->frames, W->tab); W->P = buildNullspaceProjector(W->frames); }
  buildDelta(W->calib, W->frames, W->tab);
  std::vector<double>& pr = W->h_prstage;   // member: the copy may still be in flight when upload_tables returns (sync == false)
  pr.assign((size_t)nf * 16 + 4 + n, 0.0);
  for (int f = 0; f < nf; f++)
    for (int i = 0; i < 8; i++) { pr[f * 8 + i] = W->frames[f].prior[i]; pr[nf * 8 + f * 8 + i] = W->frames[f].delta_prior[i]; }
  for (int i = 0; i < 4; i++) pr[nf * 16 + i] = W->tab.cPrior[i];
  for (int i = 0; i < 4; i++) pr[nf * 16 + 4 + i] = (double)W->tab.cDeltaF[i];
  for (int f = 0; f < nf; f++) for (int i = 0; i < 8; i++) pr[nf * 16 + 4 + 4 + f * 8 + i] = W->frames[f].delta[i];
  // calibration scalars live in the descriptor
  W->d.fxl = W->calib.value_scaledf[0]; W->d.fyl = W->calib.value_scaledf[1];
  W->d.cxl = W->calib.value_scaledf[2]; W->d.cyl = W->calib.value_scaledf[3];
  W->d.fxli = W->calib.value_scaledi[0]; W->d.fyli = W->calib.value_scaledi[1];
}
// tables -> device.  The block is contiguous in the window's slab, so it travels as ONE copy from a pinned staging area: `stage`
// (tbl_bytes of the caller's reservation; it must stay untouched until the stream has passed the copy), or the ctx staging buffer,
// in which case the call synchronises.
static int upload_tables(sdso_ctx* ctx, BaWindowDev* W, bool adjoints, bool sync = true, bool built = false, char* stage = nullptr) {
  const int nf = W->d.nf, n = W->d.n;
  if (!built) build_tables(W, adjoints);
  if (!stage) {
    int rc = stage_reserve(ctx, W->tbl_bytes, &stage);
    if (rc) return rc;
    sync = true;
  }
  auto put = [&](const void* dst, const void* src, size_t bytes) { std::memcpy(stage + ((const char*)dst - W->tbl_first), src, bytes); };
  put(W->dt_precalc, W->tab.precalc.data(), sizeof(float) * nf * nf * 27);
  put(W->dt_adHTdelta, W->tab.adHTdeltaF.data(), sizeof(float) * nf * nf * 8);
  put(W->dt_cdelta, W->tab.cDeltaF, sizeof(float) * 4);
  put(W->dt_adHost, W->tab.adHost.data(), sizeof(double) * nf * nf * 64);       // unchanged unless `adjoints`: the host copies persist
  put(W->dt_adTarget, W->tab.adTarget.data(), sizeof(double) * nf * nf * 64);
  put(W->dt_P, W->P.a.data(), sizeof(double) * n * n);
  put(W->dt_prior, W->h_prstage.data(), sizeof(double) * W->h_prstage.size());
  put(W->d_self, &W->d, sizeof(BaDev));
  SDSO_HIP(ctx, hipMemcpyAsync(W->tbl_first, stage, W->tbl_bytes, hipMemcpyHostToDevice, ctx->stream));
  if (sync) SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SDSO_OK;
}

// host mirror of the marginalisation prior <- device (the kernels that change it leave the mirror stale)
static int sync_prior_host(sdso_ctx* ctx, BaWindowDev* W) {
  if (W->hm_host_valid) return SDSO_OK;
  const int n = W->d.n;
  SDSO_HIP(ctx, hipMemcpyAsync(W->HM.data(), W->dt_HM, sizeof(double) * n * n, hipMemcpyDeviceToHost, ctx->stream));
  SDSO_HIP(ctx, hipMemcpyAsync(W->bM.data(), W->dt_bM, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  W->hm_host_valid = true;
  return SDSO_OK;
}

static BaWindowDev* find_win(sdso_ctx* ctx, int win) {
  auto it = ctx->wins.find(win);
  return it == ctx->wins.end() ? nullptr : it->second;
}

}  // namespace sdso

using namespace sdso;

extern "C" int sdso_ba_accum_floats(int nf) { return (int)acc_floats(nf); }

extern "C" int sdso_ba_release_window(sdso_ctx* ctx, int win) {
  if (!ctx) return SDSO_ERR_STATE;
  auto it = ctx->wins.find(win);
  if (it == ctx->wins.end()) return SDSO_OK;
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (it->second->in_batch) free_batch(ctx);   // the batch holds a snapshot of this window's buffers
  free_window(ctx, it->second);
  ctx->wins.erase(it);
  return SDSO_OK;
}

static int upload_window_impl(sdso_ctx* ctx, int win, const sdso_ba_window_t* Win) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  SDSO_REQUIRE(ctx, Win, "null window");
  const int nf = Win->nf, np = Win->np, nr = Win->nr;
  SDSO_REQUIRE(ctx, nf >= 1 && nf <= 8 && np >= 0 && nr >= 0, "window sizes out of range (nf <= 8: setting_maxFrames is 7, settings.cpp:65)");
  SDSO_REQUIRE(ctx, Win->evalPT && Win->state && Win->state_zero && Win->ab_exposure && Win->frameEnergyTH && Win->frameID && Win->frame_slot, "null frame arrays");
  SDSO_REQUIRE(ctx, np == 0 || (Win->u && Win->v && Win->idepth && Win->idepth_zero && Win->color && Win->weights && Win->host && Win->hasDepthPrior), "null point arrays");
  SDSO_REQUIRE(ctx, nr == 0 || (Win->res_point && Win->res_target && Win->res_state), "null residual arrays");
  // (every bit of setting_solverMode has its branch: solveSystemF's in launch_solve, STEPMOMENTUM / MOMENTUM in the GN loops, ORTHOGONALIZE_POINTMARG /
  // _FULL in sdso_ba_marginalize_points)
  const bool timing = dbg_env("SDSO_BA_UPLOAD_TIMING") != nullptr;   // phase times of the upload on stderr (diagnostic)
  auto t_prev = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (!timing) return;
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[sdso_ba_upload_window] %-28s %7.1f us\n", what, std::chrono::duration<double, std::micro>(t - t_prev).count());
    t_prev = t;
  };
  int rc = sdso_ba_release_window(ctx, win);
  if (rc) return rc;
  mark("release of the old window");
  const bool use_tiled = dbg_env("SDSO_BA_ROWMAJOR") == nullptr;   // 4x2-tiled level-0 images for the linearisation (default)

  BaWindowDev* W = new BaWindowDev();
  ctx->wins[win] = W;
  BaDev& d = W->d;
  std::memset(&d, 0, sizeof(d));
  d.nf = nf; d.np = np; d.nr = nr; d.nrp = (nr + 63) & ~63; d.w = Win->w; d.h = Win->h; d.n = 8 * nf + 4;
  d.wM3 = (float)(Win->w - 3); d.hM3 = (float)(Win->h - 3);
  d.affA_fixed = Win->affineOptModeA < 0; d.affB_fixed = Win->affineOptModeB < 0;
  d.jfix = dbg_env("SDSO_BA_JSWAP") && atoi(dbg_env("SDSO_BA_JSWAP")) ? 0 : 1;
  W->solverMode = Win->solverMode; W->forceAccept = Win->forceAcceptStep; W->affA = Win->affineOptModeA; W->affB = Win->affineOptModeB;
  d.solver_mode = Win->solverMode;
  d.have_first_frame = 0;
  for (int f = 0; f < nf; f++) if (Win->frameID[f] == 0) d.have_first_frame = 1;
  const int n = d.n;

  // ---- host mirror
  for (int i = 0; i < 4; i++) W->calib.value_zero[i] = Win->calib_value_zero[i];
  W->calib.setValueScaled(Win->calib_value_scaled);
  W->frames.resize(nf);
  std::vector<const float4*> imgs(nf);
  for (int f = 0; f < nf; f++) {
    HostFrame& F = W->frames[f];
    std::memcpy(F.evalPT.R.data(), Win->evalPT + f * 12, 72);
    std::memcpy(F.evalPT.t.data(), Win->evalPT + f * 12 + 9, 24);
    F.ab_exposure = Win->ab_exposure[f]; F.frameEnergyTH = Win->frameEnergyTH[f]; F.frameID = Win->frameID[f]; F.frame_slot = Win->frame_slot[f];
    F.setState(Win->state + f * 10);
    F.setStateZero(Win->state_zero + f * 10);
    for (int i = 0; i < 10; i++) F.step[i] = 0;
    F.fillPrior(W->affA, W->affB, W->solverMode);
    auto ip = ctx->pyr.find(F.frame_slot);
    SDSO_REQUIRE(ctx, ip != ctx->pyr.end(), "window references a frame slot without an uploaded pyramid");
    SDSO_REQUIRE(ctx, ip->second.w[0] == Win->w && ip->second.h[0] == Win->h, "pyramid level-0 size differs from the window's w/h");
    if (use_tiled) {
      int rc = ensure_tiled0(ctx, ip->second);
      if (rc) return rc;
      imgs[f] = ip->second.tiled0;
    } else imgs[f] = ip->second.d[0];
  }
  W->HM.assign((size_t)n * n, 0.0); W->bM.assign(n, 0.0);
  if (Win->HM) std::memcpy(W->HM.data(), Win->HM, sizeof(double) * n * n);
  if (Win->bM) std::memcpy(W->bM.data(), Win->bM, sizeof(double) * n);
  W->prior_pristine = std::all_of(W->HM.begin(), W->HM.end(), [](double v) { return v == 0.0; }) && std::all_of(W->bM.begin(), W->bM.end(), [](double v) { return v == 0.0; });

  mark("host mirror of the frames");
  // ---- validate + sort residuals by (host,target) pair, stable.  Two passes over the residuals (validation + keys + counts, then the
  // placement with the sorted arrays written on the way) — a keyframe's upload is on the caller's critical path (round 5: ten passes and a
  // np x nf scratch array were 66 of its 160 us)
  for (int p = 0; p < np; p++) {
    SDSO_REQUIRE(ctx, Win->host[p] >= 0 && Win->host[p] < nf, "point host out of range");
    SDSO_REQUIRE(ctx, p == 0 || Win->host[p] >= Win->host[p - 1], "points must be in allPoints order (host index non-decreasing)");
  }
  std::vector<int> rbeg(np + 1, 0), rcnt(np, 0);
  std::vector<uint8_t> rkey(nr);
  int cnt[65] = {0};
  {  // every validation runs before the first H2D copy
    int cur = -1; unsigned seen = 0;       // the targets the current point's residuals have named so far
    for (int i = 0; i < nr; i++) {
      const int p = Win->res_point[i], t = Win->res_target[i];
      SDSO_REQUIRE(ctx, p >= 0 && p < np && p >= cur, "residuals must be grouped by point in point order");
      SDSO_REQUIRE(ctx, t >= 0 && t < nf, "residual target out of range");
      if (p != cur) { cur = p; seen = 0; }
      SDSO_REQUIRE(ctx, !((seen >> t) & 1u), "two residuals of one point observe the same target frame");
      seen |= 1u << t;
      const int h = Win->host[p];
      // (the reference never creates one: `if(fh != point->host)`, FullSystemOptPoint.cpp:74; the Schur kernel has no column for it)
      SDSO_REQUIRE(ctx, t != h, "a residual observes its own host frame");
      SDSO_REQUIRE(ctx, ++rcnt[p] <= SDSO_MAX_RES, "more than MAX_RES_PER_POINT residuals on a point");
      const int key = h + t * nf;           // htIDX (nf^2 <= 64 keys)
      rkey[i] = (uint8_t)key;
      cnt[key + 1]++;
    }
  }
  for (int p = 0; p < np; p++) rbeg[p + 1] = rbeg[p] + rcnt[p];     // (residuals are grouped by point: a point's first residual, nr behind the last)
  for (int k = 0; k < nf * nf; k++) cnt[k + 1] += cnt[k];
  int pair_first[65];
  for (int k = 0; k <= nf * nf; k++) pair_first[k] = cnt[k];          // first sorted residual of every pair (the chunk lists below)
  W->perm.resize(nr); W->inv.resize(nr);
  std::vector<int> s_point(nr);
  std::vector<uint8_t> s_host(nr), s_target(nr), s_state(nr);
  for (int i = 0; i < nr; i++) {            // stable counting sort: placement, the inverse and the sorted arrays in one pass
    const int key = rkey[i], j = cnt[key]++;
    W->perm[j] = i; W->inv[i] = j;
    s_point[j] = Win->res_point[i]; s_target[j] = (uint8_t)Win->res_target[i]; s_host[j] = (uint8_t)(key - s_target[j] * nf); s_state[j] = Win->res_state[i];
  }
  W->h_target = s_target;
  W->h_point = s_point;
  W->newest_first = nr > 0 && nf > 0 ? pair_first[(nf - 1) * nf] : nr;    // first pair-sorted residual whose target is the newest frame (keys host + target * nf)
  if (W->newest_first > nr) W->newest_first = nr;
  W->h_lin.assign(nr, 0);
  W->has_lin_cached = false;
  // chunks per pair
  std::vector<int4> chunks;
  std::vector<int> pair_beg(nf * nf + 1, 0);
  {
    chunks.reserve(nf * nf + nr / BA_CHUNK + 1);
    for (int pair = 0; pair < nf * nf; pair++) {
      pair_beg[pair] = (int)chunks.size();
      const int start = pair_first[pair], j = pair_first[pair + 1];
      for (int s = start; s < j; s += BA_CHUNK) chunks.push_back(make_int4(pair, s, std::min(BA_CHUNK, j - s), 0));
    }
    pair_beg[nf * nf] = (int)chunks.size();
  }
  // point ranges per host, in 64-point items (the Schur kernel deals 64-point slices to its waves)
  const int sc_pts = 64;
  std::vector<int4> items;
  std::vector<int> host_beg(nf + 1, 0);
  {
    int p = 0;
    for (int h = 0; h < nf; h++) {
      host_beg[h] = (int)items.size();
      int start = p;
      while (p < np && Win->host[p] == h) p++;
      for (int s = start; s < p; s += sc_pts) items.push_back(make_int4(h, s, std::min(s + sc_pts, p), 0));
    }
    host_beg[nf] = (int)items.size();
  }
  {
    int p = 0;
    for (int h = 0; h <= 8; h++) {
      d.host_pt_beg[h] = p;
      while (h < nf && p < np && Win->host[p] == h) p++;
    }
  }
  d.nchunks = (int)chunks.size();
  d.nitems = (int)items.size();
  W->nblk_res = (nr + BA_BLOCK - 1) / BA_BLOCK;
  W->nblk_pts = (np + BA_BLOCK - 1) / BA_BLOCK;

  mark("validation, sort, work lists");
  // ---- device memory: ONE slab per window.  Segments whose content comes from the host sit at its front and are filled by one staged
  // H2D copy (a pinned staging buffer of the ctx, same layout); everything behind them is cleared by one memset.  (~60 separate buffers
  // with a memset each and ~40 small pageable copies cost 0.7 of the 0.96 ms an upload took.)
  struct Seg { size_t bytes; bool init; std::function<void(char*)> set; size_t off; };
  std::vector<Seg> segs;
#define PL(ptr, T, count, init) segs.push_back(Seg{sizeof(T) * (size_t)(count), (init), [&](char* b) { ptr = (T*)b; }, 0})
  float4* p_geo; float *p_color, *p_weights, *p_prior, *p_delta, *p_out; int *p_host, *p_rbeg, *p_rcnt, *p_rlist;
  unsigned* p_order; float4* p_track; uint8_t* r_isnew;
  PL(p_geo, float4, np, true); PL(p_color, float, np * 8, true); PL(p_weights, float, np * 8, true); PL(p_host, int, np, true);
  PL(p_prior, float, np, true); PL(p_delta, float, np, true); PL(p_rbeg, int, np + 1, true); PL(p_rcnt, int, np, true); PL(p_rlist, int, nr, true);
  PL(p_order, unsigned, np, true); PL(p_track, float4, np, true); PL(r_isnew, uint8_t, nr, true);
  PL(p_out, float, (size_t)np * 16, false); PL(d.p_stepbk, float, np, false);
  int* r_point; int* r_orig; uint8_t *r_host, *r_target;
  PL(r_point, int, nr, true); PL(r_orig, int, nr, true); PL(r_host, uint8_t, nr, true); PL(r_target, uint8_t, nr, true);
  PL(d.r_state, uint8_t, nr, true); PL(d.r_newState, uint8_t, nr, false); PL(d.r_lin, uint8_t, nr, false); PL(d.r_act, uint8_t, nr, false); PL(d.r_jsel, uint8_t, nr, false);
  PL(d.r_energy, float, nr, false); PL(d.r_newEnergy, float, nr, false); PL(d.r_newEnergyWO, float, nr, false);
  PL(d.J[0], float, (size_t)76 * d.nrp, false); PL(d.J[1], float, (size_t)76 * d.nrp, false); PL(d.r_toZero, float, (size_t)8 * d.nrp, false);
  PL(d.r_rec, float, (size_t)(nr + 16) * 16, false);  // per-residual records of the Schur part, window order (ba_kernels.h)
  PL(d.r_cj, float, (size_t)(nr + 16) * 8, false);    // their JpJdF halves, compact (written by k_ba_sc_host)
  d.r_proj = nullptr;
  // the tables upload_tables refreshes: contiguous, in this order (one staged copy there too)
  PL(W->dt_precalc, float, nf * nf * 27, true); PL(W->dt_adHTdelta, float, nf * nf * 8, true); PL(W->dt_cdelta, float, 4, true);
  PL(W->dt_adHost, double, nf * nf * 64, true); PL(W->dt_adTarget, double, nf * nf * 64, true); PL(W->dt_P, double, (size_t)n * n, true);
  PL(W->dt_prior, double, nf * 16 + 4 + n, true); PL(W->d_self, BaDev, 1, true);
  PL(W->dt_frameTH, float, nf, true);
  PL(W->dt_HM, double, (size_t)n * n, true); PL(W->dt_bM, double, n, true); PL(W->dt_xAd, float, nf * nf * 8, false);
  const float4** d_img; PL(d_img, const float4*, nf, true);
  int4* d_chunks; int* d_pair_beg; int4* d_items; int* d_host_beg;
  PL(d_chunks, int4, chunks.size(), true); PL(d_pair_beg, int, nf * nf + 1, true); PL(d_items, int4, items.size(), true); PL(d_host_beg, int, nf + 1, true);
  PL(d.top_part, double, (size_t)d.nchunks * 92, false); PL(d.sc_part, float, (size_t)nf * 20, false);
  PL(d.e_part, double, std::max(W->nblk_res, d.nchunks) + 1, false);
  PL(d.accum, float, acc_floats(nf), false);
  PL(d.sol, double, sol_doubles(n, nf), false);
  PL(W->d_pflag, uint8_t, np, false); PL(W->d_sums, float, 2 * (W->nblk_pts + 1), false);
  PL(W->d_opt, BaOptDev, 1, false);
#undef PL
  size_t init_bytes = 0, total = 0;
  for (int pass = 0; pass < 2; pass++) {
    for (Seg& sg : segs)
      if (sg.init == (pass == 0)) { sg.off = total; total += ((sg.bytes ? sg.bytes : 16) + 255) & ~(size_t)255; }
    if (pass == 0) init_bytes = total;
  }
  char* slab = nullptr;
  {
    void* sp = nullptr;
    int rc2 = dmalloc(ctx, W, &sp, total, false);
    if (rc2) return rc2;
    slab = (char*)sp;
  }
  for (Seg& sg : segs) sg.set(slab + sg.off);
  W->accum_own = d.accum;
  d.opt = W->d_opt; d.finished = 0;
  W->tbl_first = (char*)W->dt_precalc; W->tbl_bytes = (size_t)((char*)W->d_self + sizeof(BaDev) - (char*)W->dt_precalc);
  char* stage = nullptr;
  {
    int rc2 = stage_reserve(ctx, init_bytes, &stage);
    if (rc2) return rc2;
    std::memset(stage, 0, init_bytes);
  }
#define STG(dst, src, bytes) std::memcpy(stage + ((const char*)(dst) - slab), (src), (bytes))

  d.p_geo = p_geo; d.p_color = p_color; d.p_weights = p_weights; d.p_host = p_host; d.p_prior = p_prior; d.p_delta = p_delta;
  d.p_rbeg = p_rbeg; d.p_rcnt = p_rcnt; d.p_rlist = p_rlist; d.p_out = p_out;
  d.p_order = p_order; d.p_track = p_track; d.r_isnew = r_isnew;
  d.r_point = r_point; d.r_orig = r_orig; d.r_host = r_host; d.r_target = r_target;
  d.tiledT = use_tiled ? (Win->w + 3) / 4 : 0;
  d.t_precalc = W->dt_precalc; d.t_adHTdelta = W->dt_adHTdelta; d.t_cdelta = W->dt_cdelta; d.t_frameTH = W->dt_frameTH; d.t_img = d_img;
  d.t_adHost = W->dt_adHost; d.t_adTarget = W->dt_adTarget; d.t_xAd = W->dt_xAd; d.t_prior = W->dt_prior; d.t_HM = W->dt_HM; d.t_bM = W->dt_bM; d.t_P = W->dt_P;
  d.chunks = d_chunks; d.pair_chunk_beg = d_pair_beg; d.items = d_items; d.host_item_beg = d_host_beg;

  mark("slab + staging reservation");
  // ---- uploads
  std::vector<float4> geo(np);
  W->h_prior.resize(np);
  std::vector<float> delta(np);
  for (int p = 0; p < np; p++) {
    geo[p] = make_float4(Win->u[p], Win->v[p], SCALE_IDEPTH * Win->idepth[p], SCALE_IDEPTH * Win->idepth_zero[p]);
    float pr = Win->hasDepthPrior[p] ? 50.f * 50.f * SCALE_IDEPTH * SCALE_IDEPTH : 0.f;  // EFPoint::takeData, setting_idepthFixPrior
    if (W->solverMode & SOLVER_REMOVE_POSEPRIOR) pr = 0;
    W->h_prior[p] = pr;
    delta[p] = Win->idepth[p] - Win->idepth_zero[p];
  }
  std::vector<int> rlist(nr);
  for (int i = 0; i < nr; i++) rlist[i] = W->inv[i];   // slot order == original order (grouped by point)
  // EFPoint::residualsAll order of every point as a word of target nibbles (BaDev::p_order); PointHessian::maxRelBaseline / numGoodResiduals
  std::vector<unsigned> order(np, 0xffffffffu);
  for (int p = 0; p < np; p++)
    for (int k = 0; k < rcnt[p]; k++) order[p] = (order[p] & ~(15u << (4 * k))) | ((unsigned)Win->res_target[rbeg[p] + k] << (4 * k));
  std::vector<float4> track(np);
  for (int p = 0; p < np; p++) {
    const int ng = Win->numGoodResiduals ? Win->numGoodResiduals[p] : 0;
    float ngf; std::memcpy(&ngf, &ng, 4);
    track[p] = make_float4(Win->maxRelBaseline ? Win->maxRelBaseline[p] : 0.f, ngf, 0.f, 0.f);
  }
  std::vector<uint8_t> isnew(nr, 1);
  if (Win->res_isNew) for (int j = 0; j < nr; j++) isnew[j] = Win->res_isNew[W->perm[j]] ? 1 : 0;
  std::vector<float> frameTH(nf);
  for (int f = 0; f < nf; f++) frameTH[f] = W->frames[f].frameEnergyTH;
  STG(p_geo, geo.data(), sizeof(float4) * np); STG(p_color, Win->color, sizeof(float) * np * 8); STG(p_weights, Win->weights, sizeof(float) * np * 8);
  STG(p_host, Win->host, sizeof(int) * np); STG(p_prior, W->h_prior.data(), sizeof(float) * np); STG(p_delta, delta.data(), sizeof(float) * np);
  STG(p_rbeg, rbeg.data(), sizeof(int) * (np + 1)); STG(p_rcnt, rcnt.data(), sizeof(int) * np); STG(p_rlist, rlist.data(), sizeof(int) * nr);
  STG(p_order, order.data(), sizeof(unsigned) * np); STG(p_track, track.data(), sizeof(float4) * np); STG(r_isnew, isnew.data(), nr);
  STG(r_point, s_point.data(), sizeof(int) * nr); STG(r_orig, W->perm.data(), sizeof(int) * nr); STG(r_host, s_host.data(), nr); STG(r_target, s_target.data(), nr); STG(d.r_state, s_state.data(), nr);
  STG(W->dt_frameTH, frameTH.data(), sizeof(float) * nf); STG(d_img, imgs.data(), sizeof(float4*) * nf);
  STG(d_chunks, chunks.data(), sizeof(int4) * chunks.size()); STG(d_pair_beg, pair_beg.data(), sizeof(int) * (nf * nf + 1));
  STG(d_items, items.data(), sizeof(int4) * items.size()); STG(d_host_beg, host_beg.data(), sizeof(int) * (nf + 1));
  STG(W->dt_HM, W->HM.data(), sizeof(double) * n * n); STG(W->dt_bM, W->bM.data(), sizeof(double) * n);
  mark("staging of points / residuals");
  // the tables at the uploaded state, staged with everything else
  build_tables(W, true);
  STG(W->dt_precalc, W->tab.precalc.data(), sizeof(float) * nf * nf * 27);
  STG(W->dt_adHTdelta, W->tab.adHTdeltaF.data(), sizeof(float) * nf * nf * 8);
  STG(W->dt_cdelta, W->tab.cDeltaF, sizeof(float) * 4);
  STG(W->dt_adHost, W->tab.adHost.data(), sizeof(double) * nf * nf * 64);
  STG(W->dt_adTarget, W->tab.adTarget.data(), sizeof(double) * nf * nf * 64);
  STG(W->dt_P, W->P.a.data(), sizeof(double) * n * n);
  STG(W->dt_prior, W->h_prstage.data(), sizeof(double) * W->h_prstage.size());
  STG(W->d_self, &W->d, sizeof(BaDev));
#undef STG
  mark("tables (adjoints, projector)");
  SDSO_HIP(ctx, hipMemcpyAsync(slab, stage, init_bytes, hipMemcpyHostToDevice, ctx->stream));
  if (total > init_bytes) SDSO_HIP(ctx, hipMemsetAsync(slab + init_bytes, 0, total - init_bytes, ctx->stream));
  // per-residual record: target in slot 15, newState OUTLIER, newEnergyWO -1
  if (nr) hipLaunchKernelGGL(k_ba_init_res, dim3(W->nblk_res), dim3(BA_BLOCK), 0, ctx->stream, W->d_self);
  SDSO_HIP(ctx, hipGetLastError());
  // no synchronisation: the upload is ENQUEUED (copy, clear, init kernel) and whatever the caller does next on this ctx queues behind it;
  // the staging buffer is marked in flight (round 5 waited here: 38 of the call's 160 us)
  { const int rcc = stage_commit(ctx); if (rcc) return rcc; }
  mark("copy + clear + init kernel (enqueue)");
  return SDSO_OK;
}
// a window that failed half-way through its upload must not stay registered (later calls would launch on null arrays)
extern "C" int sdso_ba_upload_window(sdso_ctx* ctx, int win, const sdso_ba_window_t* Win) {
  const int rc = upload_window_impl(ctx, win, Win);
  if (rc && ctx) {
    const std::string why = ctx->err;
    sdso_ba_release_window(ctx, win);
    ctx->err = why;
  }
  return rc;
}

// optional: keep projectedTo / centerProjectedTo (tests); costs 76 B of stores per residual
extern "C" int sdso_ba_keep_projections(sdso_ctx* ctx, int win, int on) {
  if (!ctx) return SDSO_ERR_STATE;
  BaWindowDev* W = find_win(ctx, win);
  SDSO_REQUIRE(ctx, W, "unknown window");
  if (on && !W->d.r_proj) { DM(W->d.r_proj, float, (size_t)W->d.nr * 19); }
  if (!on) W->d.r_proj = nullptr;
  H2D(W->d_self, &W->d, sizeof(BaDev));
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SDSO_OK;
}

// ------------------------------------------------------------------ launches on an array of windows
namespace sdso {
static void launch_linearize(sdso_ctx* ctx, const BaLaunch& L) {
  ProfScope ps(ctx, "k_ba_linearize");
  if (L.tiled) hipLaunchKernelGGL(k_ba_linearize<true>, dim3(L.max_nblk_res, L.nwin), dim3(BA_BLOCK), 0, ctx->stream, L.d_arr);
  else hipLaunchKernelGGL(k_ba_linearize<false>, dim3(L.max_nblk_res, L.nwin), dim3(BA_BLOCK), 0, ctx->stream, L.d_arr);
}
static void launch_apply(sdso_ctx* ctx, const BaLaunch& L) {
  hipLaunchKernelGGL(k_ba_apply, dim3(L.max_nblk_res, L.nwin), dim3(BA_BLOCK), 0, ctx->stream, L.d_arr);
}
static bool launch_sc_and_folds(sdso_ctx* ctx, const BaLaunch& L, const uint8_t* pflag, bool marg, bool fold_top_too = false, bool defer_fold = false);
// the back-substitution kernels read the points' L sums (p_out[8..13]) only when linearised residuals exist in the launch
#define LAUNCH_RESUB(L_, ...) do { if ((L_).any_lin) hipLaunchKernelGGL(k_ba_resub<true>, __VA_ARGS__); else hipLaunchKernelGGL(k_ba_resub<false>, __VA_ARGS__); } while (0)
#define LAUNCH_RESUB_STEP(L_, ...) do { if ((L_).any_lin) hipLaunchKernelGGL(k_ba_resub_step<true>, __VA_ARGS__); else hipLaunchKernelGGL(k_ba_resub_step<false>, __VA_ARGS__); } while (0)
static void launch_accumulate(sdso_ctx* ctx, const BaLaunch& L, const uint8_t* pflag, bool marg) {
  const int nf = L.nf;
  // the folds run even without a single chunk: they are what clears the top bins of the previous call
  if (!marg) {
    if (L.max_chunks > 0) { ProfScope ps(ctx, "k_ba_accum_top", 2); hipLaunchKernelGGL(k_ba_accum_top, dim3(L.max_chunks, L.nwin), dim3(BA_BLOCK), 0, ctx->stream, L.d_arr, 0, (const uint8_t*)nullptr); }
    hipLaunchKernelGGL(k_ba_fold_top, dim3(nf * nf, L.nwin), dim3(128), 0, ctx->stream, L.d_arr, 0);
    if (L.any_lin && L.max_chunks > 0) {
      hipLaunchKernelGGL(k_ba_accum_top, dim3(L.max_chunks, L.nwin), dim3(BA_BLOCK), 0, ctx->stream, L.d_arr, 1, (const uint8_t*)nullptr);
      hipLaunchKernelGGL(k_ba_fold_top, dim3(nf * nf, L.nwin), dim3(128), 0, ctx->stream, L.d_arr, 1);
    } else {
      // accumulateLF_MT over zero linearized residuals: only the priors survive (added in the stitch)
      hipLaunchKernelGGL(k_ba_zero_topL, dim3(nf * nf, L.nwin), dim3(128), 0, ctx->stream, L.d_arr);
    }
  } else {
    if (L.max_chunks > 0) hipLaunchKernelGGL(k_ba_accum_top, dim3(L.max_chunks, L.nwin), dim3(BA_BLOCK), 0, ctx->stream, L.d_arr, 2, pflag);
    hipLaunchKernelGGL(k_ba_fold_top, dim3(nf * nf, L.nwin), dim3(128), 0, ctx->stream, L.d_arr, 0);
  }
  launch_sc_and_folds(ctx, L, pflag, marg);
}
// The Schur kernel (one workgroup per host frame and window) and what is left to fold afterwards: Hcc / bc over the hosts, and — with
// fold_top_too — the top partials of the fused kernel.  Returns false when those folds were left to the fused tail kernel (defer_fold).
static bool launch_sc_and_folds(sdso_ctx* ctx, const BaLaunch& L, const uint8_t* pflag, bool marg, bool fold_top_too, bool defer_fold) {
  const int nf = L.nf;
  const int shift = marg ? 0 : 1, mm = marg ? 1 : 0;
  // the launch's common case — no marginalisation pass, no point filter, no linearized residual — takes the kernel's lean per-point loop
  const bool plain = !marg && !pflag && !L.any_lin;
  int clear_l = 0;
  for (BaWindowDev* W : L.Ws) {
    if (plain && W->l_dirty) clear_l = 1;
    W->l_dirty = !plain;          // (a plain launch with clear_l zeroes the L sums of every point it visits: all of them)
  }
  {
    ProfScope ps(ctx, "k_ba_sc", 2);
    // a wave per host (see the kernel) once the workgroups-per-host form would need more than three rounds of two workgroups per CU;
    // SDSO_BA_SC_WPH=0 / 1 forces one form (A/B)
    static const int wph_env = dbg_env("SDSO_BA_SC_WPH") ? atoi(dbg_env("SDSO_BA_SC_WPH")) : -1;
    const int cus = (ctx->aux && ctx->stream == ctx->aux) ? ctx->aux_cus : ctx->n_cu;     // the CUs this launch may use (CU-partitioned ctx: the aux share)
    const bool wph = wph_env >= 0 ? wph_env != 0 : 2 * nf * L.nwin > 13 * cus;   // (round 6, two workgroups per CU since the f64 accumulators — µs,
                                                                                   //  workgroup / wave form: 128 windows 70 / 76, 192: 101 / 108, 224: 122 / 111, 256: 136 / 120: profiles/r06_sc_batch_ab.txt)
    const dim3 g(wph ? (nf + BA_BLOCK / 64 - 1) / (BA_BLOCK / 64) : nf, L.nwin);
    if (plain) { if (wph) hipLaunchKernelGGL((k_ba_sc_host<true, true>), g, dim3(BA_BLOCK), 0, ctx->stream, L.d_arr, pflag, shift, mm, clear_l);
                 else hipLaunchKernelGGL((k_ba_sc_host<true, false>), g, dim3(BA_BLOCK), 0, ctx->stream, L.d_arr, pflag, shift, mm, clear_l); }
    else { if (wph) hipLaunchKernelGGL((k_ba_sc_host<false, true>), g, dim3(BA_BLOCK), 0, ctx->stream, L.d_arr, pflag, shift, mm, 0);
           else hipLaunchKernelGGL((k_ba_sc_host<false, false>), g, dim3(BA_BLOCK), 0, ctx->stream, L.d_arr, pflag, shift, mm, 0); }
  }
  if (fold_top_too && defer_fold) return false;
  if (fold_top_too) hipLaunchKernelGGL(k_ba_fold_all, dim3(1 + 2 * nf * nf, L.nwin), dim3(128), 0, ctx->stream, L.d_arr);
  else hipLaunchKernelGGL(k_ba_fold_hcc, dim3(1, L.nwin), dim3(64), 0, ctx->stream, L.d_arr);
  return true;
}
// linearizeAll + applyRes + accumulateAF in one kernel, then the (normally empty) linearized pass and the Schur part
// returns false when the folds were deferred to the tail kernel (defer_fold)
static bool launch_fused(sdso_ctx* ctx, const BaLaunch& L, bool materialize, int part = 3 /* bit 0: linearize+top, bit 1: Schur+folds */, bool defer_fold = false) {
  const int nf = L.nf;
  if ((part & 1) && L.max_chunks > 0) {
    {
#ifdef SDSO_LIN_PERSIST
      const dim3 g((L.max_chunks + 1) / 2, L.nwin), b(BA_BLOCK);      // A/B: two chunks per (persistent) workgroup
#else
      const dim3 g(L.max_chunks, L.nwin), b(BA_BLOCK);
#endif
#define LT(K) launch_timed(ctx, "k_ba_lin_fused", 1, K, g, b, (const BaDev*)L.d_arr)
      if (materialize) { if (L.tiled) LT((k_ba_lin_fused<true, true>)); else LT((k_ba_lin_fused<true, false>)); }
      else { if (L.tiled) LT((k_ba_lin_fused<false, true>)); else LT((k_ba_lin_fused<false, false>)); }
#undef LT
    }
    if (L.any_lin) {
      hipLaunchKernelGGL(k_ba_fold_top, dim3(nf * nf, L.nwin), dim3(128), 0, ctx->stream, L.d_arr, 0);
      hipLaunchKernelGGL(k_ba_accum_top, dim3(L.max_chunks, L.nwin), dim3(BA_BLOCK), 0, ctx->stream, L.d_arr, 1, (const uint8_t*)nullptr);
      hipLaunchKernelGGL(k_ba_fold_top, dim3(nf * nf, L.nwin), dim3(128), 0, ctx->stream, L.d_arr, 1);
    }
  }
  // without linearized residuals the top partials are folded together with the Schur partials, after the Schur kernel
  if (part & 2) return launch_sc_and_folds(ctx, L, nullptr, false, !L.any_lin, defer_fold);
  return true;
}
// stitchDouble of the three accumulator groups: the Schur pre-products, then one wave per output tile
static void launch_stitch(sdso_ctx* ctx, const BaLaunch& L) {
  const int nf = L.nf;
  hipLaunchKernelGGL(k_ba_stitch_pre, dim3((2 * nf * nf + 3) / 4, L.nwin), dim3(256), 0, ctx->stream, L.d_arr);
  const dim3 sg((3 * (nf * nf + nf + 1) + ST_WAVES - 1) / ST_WAVES, L.nwin), sb(64 * ST_WAVES);
  // NF = 0 (runtime nf): the fully unrolled NF = 8 instantiation was measured 2x slower (register pressure: 259 vs 127 us per 64 windows)
  hipLaunchKernelGGL(k_ba_stitch<0>, sg, sb, 0, ctx->stream, L.d_arr);
}
// the fused tail kernel (ba_tail.hip); SDSO_BA_TAIL=0 keeps the chain of separate kernels (A/B)
static bool tail_enabled() { static const bool on = !(dbg_env("SDSO_BA_TAIL") && atoi(dbg_env("SDSO_BA_TAIL")) == 0); return on; }
static void launch_tail(sdso_ctx* ctx, const BaLaunch& L, double lambda, int flags, int iteration = 0, int last = 0, int stop = 0) {
  ProfScope ps(ctx, "k_ba_tail", 2);
  if (L.nf == 8) hipLaunchKernelGGL(k_ba_tail<8>, dim3(L.nwin), dim3(TAIL_NT), 0, ctx->stream, L.d_arr, lambda, flags, iteration, last, stop);
  else hipLaunchKernelGGL(k_ba_tail<0>, dim3(L.nwin), dim3(TAIL_NT), 0, ctx->stream, L.d_arr, lambda, flags, iteration, last, stop);
}
static void launch_fold_deferred(sdso_ctx* ctx, const BaLaunch& L) {   // what launch_fused left out under defer_fold
  hipLaunchKernelGGL(k_ba_fold_all, dim3(1 + 2 * L.nf * L.nf, L.nwin), dim3(128), 0, ctx->stream, L.d_arr);
}
// stitch + solveSystemF (default branch) + resubstitute.  orth bit 0: x -= P x; bit 1: lambda of the window's resident loop.
// folded = false: the accumulate left the folds to the tail kernel (launch_fused with defer_fold)
static bool solve_on_host() { return dbg_env("SDSO_BA_SOLVE_HOST") != nullptr; }   // A/B: the SVD / orthogonalised-system branches through solve_system_host
static void launch_solve(sdso_ctx* ctx, const BaLaunch& L, double lambda, int orth, bool folded = true) {
  const int n = L.n;
  if (L.alt) {   // solveSystemF's SVD / orthogonalised-system branches: stitch, then one workgroup per window (ba_solve_alt.hip)
    if (!folded) launch_fold_deferred(ctx, L);
    launch_stitch(ctx, L);
    hipLaunchKernelGGL(k_ba_solve_alt, dim3(L.nwin), dim3(ALT_NT), 0, ctx->stream, L.d_arr, lambda, orth);
    if (L.max_nblk_pts > 0) LAUNCH_RESUB(L, dim3(L.max_nblk_pts, L.nwin), dim3(BA_BLOCK), 0, ctx->stream, L.d_arr);
    return;
  }
  if (tail_enabled()) {
    const int flags = TAIL_HS | ((orth & 1) ? TAIL_ORTH : 0) | ((orth & 2) ? TAIL_LAMBDA_DEV : 0) | (L.any_lin ? TAIL_TOPL : 0) | (folded ? 0 : TAIL_FOLD);
    launch_tail(ctx, L, lambda, flags);
    if (L.max_nblk_pts > 0) LAUNCH_RESUB(L, dim3(L.max_nblk_pts, L.nwin), dim3(BA_BLOCK), 0, ctx->stream, L.d_arr);
    return;
  }
  if (!folded) launch_fold_deferred(ctx, L);
  launch_stitch(ctx, L);
  const size_t lds = sizeof(double) * ((size_t)n * ((n + 2) & ~1) + 6 * n + 16) + sizeof(int) * n;   // matrix, six vectors (+16 pad), perm
  hipLaunchKernelGGL(k_ba_solve, dim3(1, L.nwin), dim3(BA_BLOCK), lds, ctx->stream, L.d_arr, lambda, orth);
  if (L.max_nblk_pts > 0) LAUNCH_RESUB(L, dim3(L.max_nblk_pts, L.nwin), dim3(BA_BLOCK), 0, ctx->stream, L.d_arr);
}
// the packed block of a batch whose latest accumulate deferred its folds: fold now (anyone but the tail kernel reads folded sums)
static void ensure_folded(sdso_ctx* ctx, BaBatch* Bt) {
  if (!Bt || Bt->folded) return;
  launch_fold_deferred(ctx, Bt->L);
  Bt->folded = true;
}
static void ensure_folded_win(sdso_ctx* ctx, BaWindowDev* W) { if (W->in_batch && reg_has(g_batches, ctx)) ensure_folded(ctx, reg_get(g_batches, ctx)); }
// bookkeeping for sdso_ba_get_linearization: where the latest linearisation's records are (fetch_jacobians)
static void mark_linearized(const std::vector<BaWindowDev*>& Ws, bool fused_materialized) {
  for (BaWindowDev* W : Ws) W->j_inplace_last = fused_materialized && W->d.jfix != 0;
}
static BaLaunch single(BaWindowDev* W) {
  BaLaunch L;
  L.d_arr = W->d_self; L.nwin = 1; L.max_nblk_res = std::max(W->nblk_res, 1); L.max_nblk_pts = W->nblk_pts;
  L.max_chunks = W->d.nchunks; L.max_items = W->d.nitems; L.nf = W->d.nf; L.n = W->d.n;
  L.any_lin = W->has_lin_cached;
  L.tiled = W->d.tiledT > 0;
  L.alt = (W->solverMode & (SOLVER_SVD | SOLVER_ORTHOGONALIZE_SYSTEM)) != 0;
  L.Ws = {W};
  return L;
}

// setNewFrameEnergyTH (FullSystemOptimize.cpp:98-139) from the energies the linearize kernel wrote
static int update_frame_energy_th(sdso_ctx* ctx, BaWindowDev* W) {
  const int nr = W->d.nr, nf = W->d.nf;
  std::vector<float> e(nr);
  if (nr) SDSO_HIP(ctx, hipMemcpyAsync(e.data(), W->d.r_newEnergyWO, sizeof(float) * nr, hipMemcpyDeviceToHost, ctx->stream));
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<float> all;
  all.reserve(nr);
  for (int j = 0; j < nr; j++)
    if (!W->h_lin[j] && e[j] >= 0 && W->h_target[j] == nf - 1) all.push_back(e[j]);
  float th;
  if (all.empty()) th = 12 * 12 * 8;
  else {
    const int nth = (int)(0.7f * all.size());
    std::nth_element(all.begin(), all.begin() + nth, all.end());
    const float nthElement = sqrtf(all[nth]);
    th = nthElement * 1.5f;
    th = 26.0f * 0.5f + th * (1 - 0.5f);
    th = th * th;
    th *= 1.0f * 1.0f;
  }
  W->frames[nf - 1].frameEnergyTH = th;
  SDSO_HIP(ctx, hipMemcpyAsync(W->dt_frameTH + (nf - 1), &W->frames[nf - 1].frameEnergyTH, sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SDSO_OK;
}

// FullSystem::linearizeAll(fixLinearization) (FullSystemOptimize.cpp:142-203)
static int linearize_all(sdso_ctx* ctx, BaWindowDev* W, bool fix, double* energy) {
  BaLaunch L = single(W);
  launch_linearize(ctx, L);
  W->j_inplace_last = false;
  if (fix) launch_apply(ctx, L);
  SDSO_HIP(ctx, hipGetLastError());
  std::vector<double> ep(W->nblk_res);
  if (W->nblk_res) SDSO_HIP(ctx, hipMemcpyAsync(ep.data(), W->d.e_part, sizeof(double) * W->nblk_res, hipMemcpyDeviceToHost, ctx->stream));
  int rc = update_frame_energy_th(ctx, W);  // synchronises
  if (rc) return rc;
  double s = 0;
  for (double v : ep) s += v;
  if (energy) *energy = s;
  W->accumulated = false;
  return SDSO_OK;
}
}  // namespace sdso

#define GET_WIN()                                   \
  if (!ctx) return SDSO_ERR_STATE;                  \
  SDSO_HIP(ctx, hipSetDevice(ctx->device));         \
  BaWindowDev* W = find_win(ctx, win);              \
  SDSO_REQUIRE(ctx, W, "unknown window")

extern "C" int sdso_ba_linearize(sdso_ctx* ctx, int win, double* energy) {
  GET_WIN();
  return linearize_all(ctx, W, false, energy);
}

namespace sdso {
// RawResidualJacobian records in the ABI's field order; ef = false: PointFrameResidual::J (= J[1 - jsel], what linearize wrote
// last — or J[jsel] when that was the fused kernel refreshing the record in place), ef = true: EFResidual::J (= J[jsel], what takeDataF swapped in)
static int fetch_jacobians(sdso_ctx* ctx, BaWindowDev* W, bool ef, float* J) {
  const int nr = W->d.nr, S = W->d.nrp;
  std::vector<float> j0((size_t)76 * S), j1((size_t)76 * S);
  std::vector<uint8_t> sel(nr);
  SDSO_HIP(ctx, hipMemcpy(j0.data(), W->d.J[0], sizeof(float) * j0.size(), hipMemcpyDeviceToHost));
  SDSO_HIP(ctx, hipMemcpy(j1.data(), W->d.J[1], sizeof(float) * j1.size(), hipMemcpyDeviceToHost));
  if (nr) SDSO_HIP(ctx, hipMemcpy(sel.data(), W->d.r_jsel, nr, hipMemcpyDeviceToHost));
  const bool both_ef = W->j_inplace_last;     // fused kernel, in place: "what linearize wrote last" sits in the EF slot too
  for (int j = 0; j < nr; j++) {
    const std::vector<float>& src = ((sel[j] != 0) != (ef || both_ef)) ? j0 : j1;
    float* o = J + (size_t)W->perm[j] * 74;
    for (int f = 0; f < 74; f++) { const int dv = jdev(f); o[f] = src[j_off(S, j, dv >> 2) + (dv & 3)]; }
  }
  return SDSO_OK;
}
}  // namespace sdso

extern "C" int sdso_ba_get_ef_jacobians(sdso_ctx* ctx, int win, float* J) {
  GET_WIN();
  SDSO_REQUIRE(ctx, J, "null buffer");
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return fetch_jacobians(ctx, W, true, J);
}

extern "C" int sdso_ba_get_linearization(sdso_ctx* ctx, int win, float* J, uint8_t* newState, float* newEnergy, float* newEnergyWithOutlier,
                                         float* projectedTo, float* centerProjectedTo) {
  GET_WIN();
  const int nr = W->d.nr;
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (J) { const int rcj = fetch_jacobians(ctx, W, false, J); if (rcj) return rcj; }
  auto fetch = [&](auto* dst, const auto* dsrc, int width) -> int {
    using T = std::remove_pointer_t<decltype(dst)>;
    std::vector<T> tmp((size_t)nr * width);
    if (nr) SDSO_HIP(ctx, hipMemcpy(tmp.data(), dsrc, sizeof(T) * tmp.size(), hipMemcpyDeviceToHost));
    for (int j = 0; j < nr; j++) std::memcpy(dst + (size_t)W->perm[j] * width, tmp.data() + (size_t)j * width, sizeof(T) * width);
    return SDSO_OK;
  };
  int rc = SDSO_OK;
  if (newState) rc |= fetch(newState, W->d.r_newState, 1);
  if (newEnergy) rc |= fetch(newEnergy, W->d.r_newEnergy, 1);
  if (newEnergyWithOutlier) rc |= fetch(newEnergyWithOutlier, W->d.r_newEnergyWO, 1);
  if (projectedTo || centerProjectedTo) {
    SDSO_REQUIRE(ctx, W->d.r_proj, "projections were not kept: call sdso_ba_keep_projections(ctx, win, 1) before linearize");
    std::vector<float> tmp((size_t)nr * 19);
    if (nr) SDSO_HIP(ctx, hipMemcpy(tmp.data(), W->d.r_proj, sizeof(float) * tmp.size(), hipMemcpyDeviceToHost));
    for (int j = 0; j < nr; j++) {
      if (projectedTo) std::memcpy(projectedTo + (size_t)W->perm[j] * 16, &tmp[(size_t)j * 19], 64);
      if (centerProjectedTo) std::memcpy(centerProjectedTo + (size_t)W->perm[j] * 3, &tmp[(size_t)j * 19 + 16], 12);
    }
  }
  return rc;
}

extern "C" int sdso_ba_apply_res(sdso_ctx* ctx, int win) {
  GET_WIN();
  launch_apply(ctx, single(W));
  SDSO_HIP(ctx, hipGetLastError());
  W->accumulated = false;
  return SDSO_OK;
}

extern "C" int sdso_ba_get_residual_state(sdso_ctx* ctx, int win, uint8_t* state, uint8_t* isActive, float* JpJdF) {
  GET_WIN();
  const int nr = W->d.nr;
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<uint8_t> t(nr);
  if (state && nr) { SDSO_HIP(ctx, hipMemcpy(t.data(), W->d.r_state, nr, hipMemcpyDeviceToHost)); for (int j = 0; j < nr; j++) state[W->perm[j]] = t[j]; }
  if (isActive && nr) { SDSO_HIP(ctx, hipMemcpy(t.data(), W->d.r_act, nr, hipMemcpyDeviceToHost)); for (int j = 0; j < nr; j++) isActive[W->perm[j]] = t[j]; }
  if (JpJdF && nr) {   // (the records lie in the window's order)
    std::vector<float> rec((size_t)nr * 16);
    SDSO_HIP(ctx, hipMemcpy(rec.data(), W->d.r_rec, sizeof(float) * rec.size(), hipMemcpyDeviceToHost));
    for (int o = 0; o < nr; o++) std::memcpy(JpJdF + (size_t)o * 8, &rec[(size_t)o * 16], 32);
  }
  return SDSO_OK;
}

extern "C" int sdso_ba_accumulate(sdso_ctx* ctx, int win) {
  GET_WIN();
  launch_accumulate(ctx, single(W), nullptr, false);
  SDSO_HIP(ctx, hipGetLastError());
  W->accumulated = true; W->marg_accumulated = false;
  return SDSO_OK;
}

extern "C" int sdso_ba_accum_dev(sdso_ctx* ctx, int win, void** dev_ptr) {
  GET_WIN();
  SDSO_REQUIRE(ctx, dev_ptr, "null out pointer");
  ensure_folded_win(ctx, W);
  *dev_ptr = W->d.accum;
  return SDSO_OK;
}

extern "C" int sdso_ba_get_accumulators(sdso_ctx* ctx, int win, float* packed) {
  GET_WIN();
  SDSO_REQUIRE(ctx, packed, "null buffer");
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ensure_folded_win(ctx, W);
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  SDSO_HIP(ctx, hipMemcpy(packed, W->d.accum, sizeof(float) * acc_floats(W->d.nf), hipMemcpyDeviceToHost));
  return SDSO_OK;
}

// overwrite the packed accumulators (after a host-side / non-RCCL reduction across ranks)
extern "C" int sdso_ba_set_accumulators(sdso_ctx* ctx, int win, const float* packed) {
  GET_WIN();
  SDSO_REQUIRE(ctx, packed, "null buffer");
  ensure_folded_win(ctx, W);
  SDSO_HIP(ctx, hipMemcpyAsync(W->d.accum, packed, sizeof(float) * acc_floats(W->d.nf), hipMemcpyHostToDevice, ctx->stream));
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  W->accumulated = true;
  return SDSO_OK;
}

extern "C" int sdso_ba_get_point_terms(sdso_ctx* ctx, int win, float* HdiF, float* bdSumF, float* Hdd_accAF, float* bd_accAF, float* Hcd_accAF) {
  GET_WIN();
  ensure_folded_win(ctx, W);      // (joins a Schur kernel that is still on the side stream)
  const int np = W->d.np;
  std::vector<float> po((size_t)np * 16);
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (np) SDSO_HIP(ctx, hipMemcpy(po.data(), W->d.p_out, sizeof(float) * po.size(), hipMemcpyDeviceToHost));
  for (int p = 0; p < np; p++) {
    const float* o = &po[(size_t)p * 16];
    if (HdiF) HdiF[p] = o[PO_HDI];
    if (bdSumF) bdSumF[p] = o[PO_BDSUM];
    if (Hdd_accAF) Hdd_accAF[p] = o[PO_HDD_A];
    if (bd_accAF) bd_accAF[p] = o[PO_BD_A];
    if (Hcd_accAF) for (int k = 0; k < 4; k++) Hcd_accAF[p * 4 + k] = o[PO_HCD_A + k];
  }
  return SDSO_OK;
}

namespace sdso {
// solveSystemF's non-default branches (EnergyFunctional.cpp:876-900 SOLVER_ORTHOGONALIZE_SYSTEM, :924-965 SOLVER_SVD [_CUT7]):
// the stitched 68x68 blocks come back from the device, the assembly and the solve run on the host in double (a Jacobi
// eigen-decomposition stands in for Eigen::JacobiSVD of the symmetric matrix), x / lastHS / lastbS go back for the
// back-substitution kernel.  Single-window path only; the batch entry points keep the default branch.
static int solve_system_host(sdso_ctx* ctx, BaWindowDev* W, int iteration, double lambda) {
  { const int rcs = sync_prior_host(ctx, W); if (rcs) return rcs; }
  const BaLaunch L = single(W);
  const int nf = L.nf, n = L.n;
  launch_stitch(ctx, L);
  SDSO_HIP(ctx, hipGetLastError());
  const size_t blk = (size_t)n * n + n;
  std::vector<double> st(3 * blk);
  SDSO_HIP(ctx, hipMemcpyAsync(st.data(), W->d.sol, sizeof(double) * st.size(), hipMemcpyDeviceToHost, ctx->stream));
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const double *HA = st.data(), *bA = HA + (size_t)n * n, *HL = st.data() + blk, *bL = HL + (size_t)n * n, *HS = st.data() + 2 * blk, *bS = HS + (size_t)n * n;
  std::vector<double> delta(n), bM_top(n);
  for (int i = 0; i < 4; i++) delta[i] = (double)W->tab.cDeltaF[i];
  for (int f = 0; f < nf; f++) for (int i = 0; i < 8; i++) delta[4 + 8 * f + i] = W->frames[f].delta[i];
  for (int i = 0; i < n; i++) { double s = 0; for (int k = 0; k < n; k++) s += W->HM[(size_t)i * n + k] * delta[k]; bM_top[i] = W->bM[i] + s; }
  Dense Hf(n);
  std::vector<double> bf(n), lastHS((size_t)n * n), lastbS(n);
  auto orthogonalize = [&](std::vector<double>* b, Dense* H) {   // EnergyFunctional.cpp:775-835 with the window's projector
    const Dense& P = W->P;
    if (b) { std::vector<double> Pb(n, 0.0); for (int i = 0; i < n; i++) { double s = 0; for (int k = 0; k < n; k++) s += P(i, k) * (*b)[k]; Pb[i] = s; } for (int i = 0; i < n; i++) (*b)[i] -= Pb[i]; }
    if (H) {
      Dense PH(n), PHP(n);
      for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { double s = 0; for (int k = 0; k < n; k++) s += P(i, k) * (*H)(k, j); PH(i, j) = s; }
      for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { double s = 0; for (int k = 0; k < n; k++) s += PH(i, k) * P(k, j); PHP(i, j) = s; }
      for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) (*H)(i, j) -= PHP(i, j);
    }
  };
  if (W->solverMode & SOLVER_ORTHOGONALIZE_SYSTEM) {
    bool haveFirstFrame = false;
    for (const HostFrame& f : W->frames) if (f.frameID == 0) haveFirstFrame = true;
    Dense HT(n);
    std::vector<double> bT(n);
    for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) HT(i, j) = HL[(size_t)i * n + j] + HA[(size_t)i * n + j] - HS[(size_t)i * n + j]; bT[i] = bL[i] + bA[i] - bS[i]; }
    if (!haveFirstFrame) orthogonalize(&bT, &HT);
    for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) Hf(i, j) = HT(i, j) + W->HM[(size_t)i * n + j]; bf[i] = bT[i] + bM_top[i]; }
    for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) lastHS[(size_t)i * n + j] = Hf(i, j); lastbS[i] = bf[i]; }
    for (int i = 0; i < n; i++) Hf(i, i) *= (1 + lambda);
  } else {
    for (int i = 0; i < n; i++) {
      for (int j = 0; j < n; j++) Hf(i, j) = HL[(size_t)i * n + j] + W->HM[(size_t)i * n + j] + HA[(size_t)i * n + j];
      bf[i] = bL[i] + bM_top[i] + bA[i] - bS[i];
    }
    for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) lastHS[(size_t)i * n + j] = Hf(i, j) - HS[(size_t)i * n + j]; lastbS[i] = bf[i]; }
    for (int i = 0; i < n; i++) Hf(i, i) *= (1 + lambda);
    const double f = (double)(1.0f / (1 + lambda));
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) Hf(i, j) -= HS[(size_t)i * n + j] * f;
  }
  std::vector<double> x(n, 0.0);
  if (W->solverMode & SOLVER_SVD) {
    std::vector<double> sv(n), bs(n), w;
    for (int i = 0; i < n; i++) sv[i] = 1.0 / std::sqrt(Hf(i, i));
    Dense Hs(n), V;
    for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) Hs(i, j) = sv[i] * Hf(i, j) * sv[j]; bs[i] = sv[i] * bf[i]; }
    symEigen(Hs, w, V);
    std::vector<int> ord(n);
    for (int i = 0; i < n; i++) ord[i] = i;
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return std::fabs(w[a]) > std::fabs(w[b]); });
    double maxSv = 0;
    for (int i = 0; i < n; i++) maxSv = std::max(maxSv, std::fabs(w[i]));
    for (int i = 0; i < n; i++) {
      const int c = ord[i];
      const double S = std::fabs(w[c]);
      double ub = 0;
      for (int k = 0; k < n; k++) ub += V(k, c) * bs[k];
      if (w[c] < 0) ub = -ub;
      if (S < kSolverModeDelta * maxSv) ub = 0;                            // setting_solverModeDelta, settings.cpp:52
      if ((W->solverMode & SOLVER_SVD_CUT7) && (i >= n - 7)) ub = 0;
      else ub /= S;
      for (int k = 0; k < n; k++) x[k] += V(k, c) * ub;
    }
    for (int k = 0; k < n; k++) x[k] *= sv[k];
  } else {
    std::vector<double> sv(n), bs(n), y;
    for (int i = 0; i < n; i++) sv[i] = 1.0 / std::sqrt(Hf(i, i) + 10);
    Dense Hs(n);
    for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) Hs(i, j) = sv[i] * Hf(i, j) * sv[j]; bs[i] = sv[i] * bf[i]; }
    solveLdlt(Hs, bs, y);
    for (int i = 0; i < n; i++) x[i] = sv[i] * y[i];
  }
  if ((W->solverMode & SOLVER_ORTHOGONALIZE_X) || (iteration >= 2 && (W->solverMode & SOLVER_ORTHOGONALIZE_X_LATER))) orthogonalize(&x, nullptr);
  double* xout = W->d.sol + 3 * blk;
  SDSO_HIP(ctx, hipMemcpyAsync(xout, x.data(), sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
  SDSO_HIP(ctx, hipMemcpyAsync(xout + n, lastHS.data(), sizeof(double) * n * n, hipMemcpyHostToDevice, ctx->stream));
  SDSO_HIP(ctx, hipMemcpyAsync(xout + n + (size_t)n * n, lastbS.data(), sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
  // xAd[nf*h+t] = xF(h)^T adHostF[h+nf*t] + xF(t)^T adTargetF[h+nf*t]  (EnergyFunctional.cpp:289-291), as k_ba_solve leaves it
  std::vector<float> xAd((size_t)nf * nf * 8);
  for (int h = 0; h < nf; h++)
    for (int t = 0; t < nf; t++)
      for (int j = 0; j < 8; j++) {
        float sh = 0, stt = 0;
        for (int i = 0; i < 8; i++) {
          sh += (float)x[4 + 8 * h + i] * (float)W->tab.adHost[(size_t)(h + nf * t) * 64 + i * 8 + j];
          stt += (float)x[4 + 8 * t + i] * (float)W->tab.adTarget[(size_t)(h + nf * t) * 64 + i * 8 + j];
        }
        xAd[(size_t)(nf * h + t) * 8 + j] = sh + stt;
      }
  SDSO_HIP(ctx, hipMemcpyAsync(W->dt_xAd, xAd.data(), sizeof(float) * xAd.size(), hipMemcpyHostToDevice, ctx->stream));
  if (L.max_nblk_pts > 0) LAUNCH_RESUB(L, dim3(L.max_nblk_pts, 1), dim3(BA_BLOCK), 0, ctx->stream, L.d_arr);
  SDSO_HIP(ctx, hipGetLastError());
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));   // x / lastHS / lastbS / xAd are stack-local
  return SDSO_OK;
}
static int solve_system(sdso_ctx* ctx, BaWindowDev* W, int iteration, double lambda) {
  if (W->solverMode & SOLVER_USE_GN) lambda = 0;
  if (W->solverMode & SOLVER_FIX_LAMBDA) lambda = 1e-5;
  if ((W->solverMode & (SOLVER_SVD | SOLVER_ORTHOGONALIZE_SYSTEM)) && solve_on_host()) return solve_system_host(ctx, W, iteration, lambda);
  const int orth = (W->solverMode & SOLVER_ORTHOGONALIZE_X) || (iteration >= 2 && (W->solverMode & SOLVER_ORTHOGONALIZE_X_LATER));
  launch_solve(ctx, single(W), lambda, orth ? 1 : 0);
  SDSO_HIP(ctx, hipGetLastError());
  return SDSO_OK;
}
static int fetch_x(sdso_ctx* ctx, BaWindowDev* W, std::vector<double>& x) {
  const int n = W->d.n;
  x.resize(n);
  SDSO_HIP(ctx, hipMemcpyAsync(x.data(), W->d.sol + 3 * ((size_t)n * n + n), sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 4; i++) W->calib.step[i] = -x[i];
  for (int f = 0; f < W->d.nf; f++) {
    for (int i = 0; i < 8; i++) W->frames[f].step[i] = -x[4 + 8 * f + i];
    W->frames[f].step[8] = W->frames[f].step[9] = 0;
  }
  return SDSO_OK;
}
}  // namespace sdso

extern "C" int sdso_ba_solve(sdso_ctx* ctx, int win, int iteration, double lambda, double* x, double* HS, double* bS, double* frame_step, double* calib_step) {
  GET_WIN();
  SDSO_REQUIRE(ctx, W->accumulated, "sdso_ba_solve needs sdso_ba_accumulate (and, across ranks, the all-reduce of the packed accumulators) first");
  int rc = solve_system(ctx, W, iteration, lambda);
  if (rc) return rc;
  std::vector<double> xs;
  rc = fetch_x(ctx, W, xs);
  if (rc) return rc;
  const int n = W->d.n;
  if (x) std::memcpy(x, xs.data(), sizeof(double) * n);
  const double* base = W->d.sol + 3 * ((size_t)n * n + n) + n;
  if (HS) SDSO_HIP(ctx, hipMemcpy(HS, base, sizeof(double) * n * n, hipMemcpyDeviceToHost));
  if (bS) SDSO_HIP(ctx, hipMemcpy(bS, base + (size_t)n * n, sizeof(double) * n, hipMemcpyDeviceToHost));
  if (frame_step) for (int f = 0; f < W->d.nf; f++) for (int i = 0; i < 8; i++) frame_step[f * 8 + i] = W->frames[f].step[i];
  if (calib_step) for (int i = 0; i < 4; i++) calib_step[i] = W->calib.step[i];
  return SDSO_OK;
}

extern "C" int sdso_ba_get_stitched(sdso_ctx* ctx, int win, double* HA, double* bA, double* HL, double* bL, double* Hsc, double* bsc) {
  GET_WIN();
  SDSO_REQUIRE(ctx, W->accumulated || W->marg_accumulated, "sdso_ba_get_stitched needs sdso_ba_accumulate (or sdso_ba_marginalize_points) first");
  ensure_folded_win(ctx, W);
  launch_stitch(ctx, single(W));
  SDSO_HIP(ctx, hipGetLastError());
  const int n = W->d.n;
  const size_t blk = (size_t)n * n + n;
  std::vector<double> st(3 * blk);
  SDSO_HIP(ctx, hipMemcpyAsync(st.data(), W->d.sol, sizeof(double) * st.size(), hipMemcpyDeviceToHost, ctx->stream));
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double* Hs[3] = {HA, HL, Hsc};
  double* bs[3] = {bA, bL, bsc};
  for (int k = 0; k < 3; k++) {
    if (Hs[k]) std::memcpy(Hs[k], st.data() + k * blk, sizeof(double) * n * n);
    if (bs[k]) std::memcpy(bs[k], st.data() + k * blk + (size_t)n * n, sizeof(double) * n);
  }
  return SDSO_OK;
}

// EnergyFunctional::resubstituteF_MT (EnergyFunctional.cpp:272-341) for a caller-supplied x: frame / calibration steps = -x, xAd from the
// float adjoints (:283-292), then resubstituteFPt for every point on the device
extern "C" int sdso_ba_resubstitute(sdso_ctx* ctx, int win, const double* x, double* frame_step, double* calib_step) {
  GET_WIN();
  SDSO_REQUIRE(ctx, x, "null x");
  SDSO_REQUIRE(ctx, W->accumulated, "sdso_ba_resubstitute needs the per-point terms of sdso_ba_accumulate");
  const int nf = W->d.nf, n = W->d.n;
  const size_t blk = (size_t)n * n + n;
  ensure_folded_win(ctx, W);
  SDSO_HIP(ctx, hipMemcpyAsync(W->d.sol + 3 * blk, x, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
  std::vector<float> xAd((size_t)nf * nf * 8);
  for (int h = 0; h < nf; h++)
    for (int t = 0; t < nf; t++)
      for (int j = 0; j < 8; j++) {
        float sh = 0, stt = 0;
        for (int i = 0; i < 8; i++) {
          sh += (float)x[4 + 8 * h + i] * (float)W->tab.adHost[(size_t)(h + nf * t) * 64 + i * 8 + j];
          stt += (float)x[4 + 8 * t + i] * (float)W->tab.adTarget[(size_t)(h + nf * t) * 64 + i * 8 + j];
        }
        xAd[(size_t)(nf * h + t) * 8 + j] = sh + stt;
      }
  SDSO_HIP(ctx, hipMemcpyAsync(W->dt_xAd, xAd.data(), sizeof(float) * xAd.size(), hipMemcpyHostToDevice, ctx->stream));
  const BaLaunch L = single(W);
  if (L.max_nblk_pts > 0) LAUNCH_RESUB(L, dim3(L.max_nblk_pts, 1), dim3(BA_BLOCK), 0, ctx->stream, L.d_arr);
  SDSO_HIP(ctx, hipGetLastError());
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 4; i++) W->calib.step[i] = -x[i];
  for (int f = 0; f < nf; f++) { for (int i = 0; i < 8; i++) W->frames[f].step[i] = -x[4 + 8 * f + i]; W->frames[f].step[8] = W->frames[f].step[9] = 0; }
  if (frame_step) for (int f = 0; f < nf; f++) for (int i = 0; i < 8; i++) frame_step[f * 8 + i] = W->frames[f].step[i];
  if (calib_step) for (int i = 0; i < 4; i++) calib_step[i] = W->calib.step[i];
  return SDSO_OK;
}

extern "C" int sdso_ba_get_point_steps(sdso_ctx* ctx, int win, float* step) {
  GET_WIN();
  const int np = W->d.np;
  std::vector<float> po((size_t)np * 16);
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (np) SDSO_HIP(ctx, hipMemcpy(po.data(), W->d.p_out, sizeof(float) * po.size(), hipMemcpyDeviceToHost));
  for (int p = 0; p < np; p++) step[p] = po[(size_t)p * 16 + PO_STEP];
  return SDSO_OK;
}

extern "C" int sdso_ba_get_tables(sdso_ctx* ctx, int win, float* precalc, double* adHost, double* adTarget, float* adHTdeltaF) {
  GET_WIN();
  const int nf = W->d.nf;
  if (precalc) std::memcpy(precalc, W->tab.precalc.data(), sizeof(float) * nf * nf * 27);
  if (adHost) std::memcpy(adHost, W->tab.adHost.data(), sizeof(double) * nf * nf * 64);
  if (adTarget) std::memcpy(adTarget, W->tab.adTarget.data(), sizeof(double) * nf * nf * 64);
  if (adHTdeltaF) std::memcpy(adHTdeltaF, W->tab.adHTdeltaF.data(), sizeof(float) * nf * nf * 8);
  return SDSO_OK;
}

// EnergyFunctional::calcLEnergyF_MT (EnergyFunctional.cpp:420-442) and calcMEnergyF (:344-351); both are 0 under
// setting_forceAceptStep (FullSystemOptimize.cpp:374-376, :1056)
static int calc_energies(sdso_ctx* ctx, BaWindowDev* W, double* EL, double* EM, bool always = false) {
  *EL = 0; *EM = 0;
  if (W->forceAccept && !always) return SDSO_OK;
  const int nf = W->d.nf, n = W->d.n;
  const int nblk = W->d.nchunks + W->nblk_pts;
  double E = 0;
  for (const HostFrame& f : W->frames) for (int i = 0; i < 8; i++) E += f.delta_prior[i] * f.prior[i] * f.delta_prior[i];
  { float s = 0; for (int i = 0; i < 4; i++) s += W->tab.cDeltaF[i] * (float)W->tab.cPrior[i] * W->tab.cDeltaF[i]; E += s; }
  if (nblk > 0) {
    int rc = ensure_scratch(ctx, sizeof(float) * nblk);
    if (rc) return rc;
    BaLaunch L = single(W);
    hipLaunchKernelGGL(k_ba_lenergy, dim3(nblk, 1), dim3(BA_BLOCK), 0, ctx->stream, L.d_arr, (float*)ctx->scratch);
    std::vector<float> part(nblk);
    SDSO_HIP(ctx, hipMemcpyAsync(part.data(), ctx->scratch, sizeof(float) * nblk, hipMemcpyDeviceToHost, ctx->stream));
    SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float Ept = 0;
    for (int b = 0; b < nblk; b++) Ept += part[b];
    E += Ept;
  }
  *EL = E;
  { const int rcs = sync_prior_host(ctx, W); if (rcs) return rcs; }
  std::vector<double> delta(n);                               // getStitchedDeltaF (:1021-1032)
  for (int i = 0; i < 4; i++) delta[i] = (double)W->tab.cDeltaF[i];       // d.head<CPARS>() = cDeltaF.cast<double>()
  for (int f = 0; f < nf; f++) for (int i = 0; i < 8; i++) delta[4 + 8 * f + i] = W->frames[f].delta[i];
  double em = 0;
  for (int i = 0; i < n; i++) { double s = 0; for (int k = 0; k < n; k++) s += W->HM[(size_t)i * n + k] * delta[k]; em += delta[i] * (2 * W->bM[i] + s); }
  *EM = em;
  return SDSO_OK;
}

// EnergyFunctional::calcLEnergyF_MT (EnergyFunctional.cpp:420-442) and calcMEnergyF (:344-351) as members a caller may invoke: the values
// themselves, whatever setting_forceAceptStep says (that test lives in FullSystem::calcLEnergy / calcMEnergy, FullSystemOptimize.cpp:374-376)
extern "C" int sdso_ba_calc_energies(sdso_ctx* ctx, int win, double* EL, double* EM) {
  GET_WIN();
  double el = 0, em = 0;
  const int rc = calc_energies(ctx, W, &el, &em, true);
  if (rc) return rc;
  if (EL) *EL = el;
  if (EM) *EM = em;
  return SDSO_OK;
}

// What EnergyFunctional::setDeltaF leaves in the reference's objects (EnergyFunctional.cpp:173-207) at the window's current state:
// cDeltaF (4 floats), EFFrame::delta / delta_prior (nf*8 doubles each), EFPoint::deltaF (np floats).  Any pointer may be NULL.
extern "C" int sdso_ba_get_deltas(sdso_ctx* ctx, int win, float* cDeltaF, double* frame_delta, double* frame_delta_prior, float* point_deltaF) {
  GET_WIN();
  const int nf = W->d.nf, np = W->d.np;
  if (cDeltaF) for (int i = 0; i < 4; i++) cDeltaF[i] = W->tab.cDeltaF[i];
  for (int f = 0; f < nf; f++)
    for (int i = 0; i < 8; i++) {
      if (frame_delta) frame_delta[f * 8 + i] = W->frames[f].delta[i];
      if (frame_delta_prior) frame_delta_prior[f * 8 + i] = W->frames[f].delta_prior[i];
    }
  if (point_deltaF && np) {
    SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SDSO_HIP(ctx, hipMemcpy(point_deltaF, W->d.p_delta, sizeof(float) * np, hipMemcpyDeviceToHost));
  }
  return SDSO_OK;
}

// FullSystem::optimize, DSO-native GN loop (FullSystemOptimize.cpp:871-1041)
extern "C" int sdso_ba_optimize(sdso_ctx* ctx, int win, int mnumOptIts, double* state_out, float* idepth_out, uint8_t* res_state_out, sdso_ba_opt_result_t* out) {
  GET_WIN();
  const int nf = W->d.nf, np = W->d.np, nr = W->d.nr;
  sdso_ba_opt_result_t res{0, 0, 0, 0};
  BaLaunch L = single(W);
  // The whole loop runs on the device (ba_opt.hip) without a host round trip: the accepted-step flow (setting_forceAceptStep, the
  // reference's default) through the fused kernel, the energy-gated flow through the un-fused ones with the decision taken by
  // k_ba_opt_gate.  The SVD / orthogonalised-system solver modes and SDSO_BA_HOST_LOOP=1 (A/B) take the host loop below.
  const bool host_loop = dbg_env("SDSO_BA_HOST_LOOP") != nullptr;   // read per call: tests flip it
  if (nf >= 2 && !host_loop && ((W->solverMode & (SOLVER_SVD | SOLVER_ORTHOGONALIZE_SYSTEM)) == 0 || !solve_on_host())) {
    int rc = optimize_resident_single(ctx, W, mnumOptIts, &res);
    if (rc) return rc;
  } else if (nf >= 2) {
    if (nf < 3) mnumOptIts = 20;
    if (nf < 4) mnumOptIts = 15;
    hipLaunchKernelGGL(k_ba_reset_all, dim3(L.max_nblk_res, 1), dim3(BA_BLOCK), 0, ctx->stream, L.d_arr);
    double lastEnergy = 0;
    int rc = linearize_all(ctx, W, false, &lastEnergy);
    if (rc) return rc;
    double lastEnergyL = 0, lastEnergyM = 0;
    rc = calc_energies(ctx, W, &lastEnergyL, &lastEnergyM);
    if (rc) return rc;
    launch_apply(ctx, L);
    double lambda = 1e-1;
    float stepsize = 1;
    const bool momentum = (W->solverMode & SOLVER_MOMENTUM) != 0;
    std::vector<double> x, previousX(W->d.n, std::numeric_limits<double>::quiet_NaN());
    std::vector<float> sums(2 * (W->nblk_pts + 1));
    for (int iteration = 0; iteration < mnumOptIts; iteration++) {
      res.iterations++;
      // backupState(iteration != 0) (:309-351); SOLVER_MOMENTUM also keeps the previous steps (the points': k_ba_resub, which looks at the
      // iteration count of the window's BaOptDev)
      for (int i = 0; i < 4; i++) W->calib.value_backup[i] = W->calib.value[i];
      for (HostFrame& f : W->frames) for (int i = 0; i < 10; i++) { f.step_backup[i] = (momentum && iteration != 0) ? f.step[i] : 0.0; f.state_backup[i] = f.state[i]; }
      W->h_opt.iterations = iteration;
      H2D(&W->d_opt->iterations, &W->h_opt.iterations, sizeof(int));
      if (L.max_nblk_pts) hipLaunchKernelGGL(k_ba_points_op, dim3(L.max_nblk_pts, 1), dim3(BA_BLOCK), 0, ctx->stream, L.d_arr, 0, 0.f, (float*)nullptr);
      // solveSystem
      launch_accumulate(ctx, L, nullptr, false);
      rc = solve_system(ctx, W, iteration, lambda);
      if (rc) return rc;
      rc = fetch_x(ctx, W, x);
      if (rc) return rc;
      {  // incDirChange and the step size (:933-948)
        double dot = 0, n0 = 0, n1 = 0;
        for (int i = 0; i < W->d.n; i++) { dot += previousX[i] * x[i]; n0 += previousX[i] * previousX[i]; n1 += x[i] * x[i]; }
        const double incDirChange = (1e-20 + dot) / (1e-20 + std::sqrt(n0) * std::sqrt(n1));
        previousX = x;
        if (std::isfinite(incDirChange) && (W->solverMode & SOLVER_STEPMOMENTUM)) {
          const float newStepsize = (float)std::exp(incDirChange * 1.4);
          if (incDirChange < 0 && stepsize > 1) stepsize = 1;
          stepsize = sqrtf(sqrtf(newStepsize * stepsize * stepsize * stepsize));
          if (stepsize > 2) stepsize = 2;
          if (stepsize < 0.25f) stepsize = 0.25f;
        }
      }
      // doStepFromBackup (:207-305)
      double nv[4];
      for (int i = 0; i < 4; i++) nv[i] = W->calib.value_backup[i] + (momentum ? 1.0f : stepsize) * W->calib.step[i];
      W->calib.setValue(nv);
      float sumA = 0, sumB = 0, sumT = 0, sumR = 0;
      for (HostFrame& fh : W->frames) {
        double ns[10], st[10];
        for (int i = 0; i < 10; i++) st[i] = fh.step[i];
        if (momentum) for (int i = 0; i < 6; i++) st[i] += 0.5f * fh.step_backup[i];     // :231
        for (int i = 0; i < 10; i++) ns[i] = fh.state_backup[i] + (momentum ? 1.0 : (double)stepsize) * st[i];
        fh.setState(ns);
        sumA += st[6] * st[6];
        sumB += st[7] * st[7];
        sumT += st[0] * st[0] + st[1] * st[1] + st[2] * st[2];
        sumR += st[3] * st[3] + st[4] * st[4] + st[5] * st[5];
      }
      float sumNID = 0, numID = (float)np;
      if (L.max_nblk_pts) {
        hipLaunchKernelGGL(k_ba_points_op, dim3(L.max_nblk_pts, 1), dim3(BA_BLOCK), 0, ctx->stream, L.d_arr, 1, stepsize, W->d_sums);
        SDSO_HIP(ctx, hipMemcpyAsync(sums.data(), W->d_sums, sizeof(float) * 2 * W->nblk_pts, hipMemcpyDeviceToHost, ctx->stream));
        SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int b = 0; b < W->nblk_pts; b++) sumNID += sums[2 * b + 1];
      }
      sumA /= nf; sumB /= nf; sumR /= nf; sumT /= nf;
      sumNID /= numID;
      rc = upload_tables(ctx, W, false);  // setPrecalcValues
      if (rc) return rc;
      const bool canbreak = sqrtf(sumA) < 0.0005 * 1.2f && sqrtf(sumB) < 0.00005 * 1.2f && sqrtf(sumR) < 0.00005 * 1.2f && sqrtf(sumT) * sumNID < 0.00005 * 1.2f;
      double newEnergy = 0;
      rc = linearize_all(ctx, W, false, &newEnergy);
      if (rc) return rc;
      double newEnergyL = 0, newEnergyM = 0;
      rc = calc_energies(ctx, W, &newEnergyL, &newEnergyM);
      if (rc) return rc;
      if (W->forceAccept || (newEnergy + newEnergyL + newEnergyM < lastEnergy + lastEnergyL + lastEnergyM)) {   // :978
        launch_apply(ctx, L);
        lastEnergy = newEnergy; lastEnergyL = newEnergyL; lastEnergyM = newEnergyM;
        lambda *= 0.25;
      } else {
        // loadSateBackup (:355-370), then re-linearize at the restored state
        W->calib.setValue(W->calib.value_backup);
        for (HostFrame& fh : W->frames) { double bs[10]; for (int i = 0; i < 10; i++) bs[i] = fh.state_backup[i]; fh.setState(bs); }
        if (L.max_nblk_pts) hipLaunchKernelGGL(k_ba_points_op, dim3(L.max_nblk_pts, 1), dim3(BA_BLOCK), 0, ctx->stream, L.d_arr, 2, 0.f, (float*)nullptr);
        rc = upload_tables(ctx, W, false);
        if (rc) return rc;
        rc = linearize_all(ctx, W, false, &lastEnergy);
        if (rc) return rc;
        rc = calc_energies(ctx, W, &lastEnergyL, &lastEnergyM);
        if (rc) return rc;
        lambda *= 1e2;
      }
      if (canbreak && iteration >= 1) break;
    }
    double nsz[10] = {0};
    nsz[6] = W->frames[nf - 1].state[6];
    nsz[7] = W->frames[nf - 1].state[7];
    W->frames[nf - 1].setEvalPT(W->frames[nf - 1].PRE_worldToCam, nsz);
    rc = upload_tables(ctx, W, true);
    if (rc) return rc;
    rc = linearize_all(ctx, W, true, &lastEnergy);
    if (rc) return rc;
    float nresA = 0;
    SDSO_HIP(ctx, hipMemcpy(&nresA, W->d.accum + acc_off_nres(nf), sizeof(float), hipMemcpyDeviceToHost));
    res.lastEnergy = lastEnergy;
    res.resInA = (int)nresA;
    res.rmse = sqrtf((float)(lastEnergy / (8 * res.resInA)));
    // linearizeAll_Reductor(true)'s per-residual bookkeeping (maxRelBaseline, numGoodResiduals; FullSystemOptimize.cpp:62-78): once per optimize, now
    if (nr) hipLaunchKernelGGL(k_ba_post_state, dim3(std::max(W->nblk_res, 1), 1), dim3(BA_BLOCK), 0, ctx->stream, (const BaDev*)W->d_self, (float*)nullptr, 1);
    SDSO_HIP(ctx, hipGetLastError());
    {
      float nres2[2] = {0, 0};
      SDSO_HIP(ctx, hipMemcpy(nres2, W->d.accum + acc_off_nres(nf), sizeof(nres2), hipMemcpyDeviceToHost));
      W->resInL = (int)nres2[1];
    }
    W->post_valid = true; W->hs_valid = true; W->last_result = res;
  } else {
    // fewer than two keyframes: the reference returns 0 before touching anything (FullSystemOptimize.cpp:873-874)
    W->post_valid = true; W->hs_valid = false; W->last_result = res; W->resInL = 0;
  }
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (state_out) for (int f = 0; f < nf; f++) for (int i = 0; i < 10; i++) state_out[f * 10 + i] = W->frames[f].state[i];
  if (idepth_out && np) {
    std::vector<float4> geo(np);
    SDSO_HIP(ctx, hipMemcpy(geo.data(), W->d.p_geo, sizeof(float4) * np, hipMemcpyDeviceToHost));
    for (int p = 0; p < np; p++) idepth_out[p] = geo[p].z;
  }
  if (res_state_out && nr) {
    std::vector<uint8_t> t(nr);
    SDSO_HIP(ctx, hipMemcpy(t.data(), W->d.r_state, nr, hipMemcpyDeviceToHost));
    for (int j = 0; j < nr; j++) res_state_out[W->perm[j]] = t[j];
  }
  if (out) *out = res;
  return SDSO_OK;
}

// flagPointsForRemoval core (FullSystem.cpp:1004-1021) + EnergyFunctional::marginalizePointsF (:663-736)
extern "C" int sdso_ba_marginalize_points(sdso_ctx* ctx, int win, const uint8_t* marg_flag, double* HM_out, double* bM_out) {
  GET_WIN();
  SDSO_REQUIRE(ctx, marg_flag, "null flags");
  const int np = W->d.np, nr = W->d.nr, n = W->d.n;
  BaLaunch L = single(W);
  H2D(W->d_pflag, marg_flag, np);
  hipLaunchKernelGGL(k_ba_reset_flagged, dim3(L.max_nblk_res, 1), dim3(BA_BLOCK), 0, ctx->stream, L.d_arr, W->d_pflag);
  if (L.tiled) hipLaunchKernelGGL(k_ba_linearize<true>, dim3(L.max_nblk_res, 1), dim3(BA_BLOCK), 0, ctx->stream, L.d_arr);
  else hipLaunchKernelGGL(k_ba_linearize<false>, dim3(L.max_nblk_res, 1), dim3(BA_BLOCK), 0, ctx->stream, L.d_arr);
  W->j_inplace_last = false;
  launch_apply(ctx, L);
  hipLaunchKernelGGL(k_ba_unmask, dim3(L.max_nblk_res, 1), dim3(BA_BLOCK), 0, ctx->stream, L.d_arr);
  hipLaunchKernelGGL(k_ba_fixlin, dim3(L.max_nblk_res, 1), dim3(BA_BLOCK), 0, ctx->stream, L.d_arr, W->d_pflag);
  for (int p = 0; p < np; p++) if (marg_flag[p]) W->h_prior[p] *= 600.f * 600.f;   // setting_idepthFixPriorMargFac (:674)
  H2D(W->d.p_prior, W->h_prior.data(), sizeof(float) * np);
  launch_accumulate(ctx, L, W->d_pflag, true);
  launch_stitch(ctx, L);
  SDSO_HIP(ctx, hipGetLastError());
  // HM += setting_margWeightFac * (M - Msc), bM likewise (:727-728): on the device copy, which is the master — the prior stays resident
  // from here through sdso_ba_marginalize_frame_dev into the next window (sdso_ba_adopt_prior); the host mirror follows on demand
  if (W->solverMode & (SOLVER_ORTHOGONALIZE_POINTMARG | SOLVER_ORTHOGONALIZE_FULL))     // (:707-731; POINTMARG only when frame 0 has left the window)
    hipLaunchKernelGGL(k_ba_prior_orth, dim3(1, 1), dim3(256), 0, ctx->stream, L.d_arr, (double)(0.5f * 0.5f),
                       ((W->solverMode & SOLVER_ORTHOGONALIZE_POINTMARG) && !W->d.have_first_frame) ? 1 : 0, (W->solverMode & SOLVER_ORTHOGONALIZE_FULL) ? 1 : 0);
  else
  hipLaunchKernelGGL(k_ba_prior_add, dim3(8, 1), dim3(256), 0, ctx->stream, L.d_arr, (double)(0.5f * 0.5f));   // setting_margWeightFac
  SDSO_HIP(ctx, hipGetLastError());
  W->hm_host_valid = false;
  W->prior_pristine = false;
  W->marg_chain = false;            // the resident prior changed: the next marginalizeFrame starts from it again
  std::vector<uint8_t> lin(nr);
  if (nr) SDSO_HIP(ctx, hipMemcpyAsync(lin.data(), W->d.r_lin, nr, hipMemcpyDeviceToHost, ctx->stream));
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  W->h_lin = lin;
  W->has_lin_cached = std::any_of(lin.begin(), lin.end(), [](uint8_t v) { return v != 0; });
  {  // resInM += accSSE_top_A->nres[0] (EnergyFunctional.cpp:704)
    float nresM = 0;
    SDSO_HIP(ctx, hipMemcpy(&nresM, W->d.accum + acc_off_nres(W->d.nf), sizeof(float), hipMemcpyDeviceToHost));
    W->resInM += (int)nresM;
  }
  if (HM_out || bM_out) {
    const int rcs = sync_prior_host(ctx, W);
    if (rcs) return rcs;
    if (HM_out) std::memcpy(HM_out, W->HM.data(), sizeof(double) * n * n);
    if (bM_out) std::memcpy(bM_out, W->bM.data(), sizeof(double) * n);
  }
  W->accumulated = false;
  W->marg_accumulated = true;
  return SDSO_OK;
}

// ------------------------------------------------------------------ batches of windows (one launch per phase)

extern "C" int sdso_ba_batch_create(sdso_ctx* ctx, int nwin, const int* wins) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  SDSO_REQUIRE(ctx, nwin > 0 && wins, "bad batch");
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  free_batch(ctx);
  // validate every member before anything is registered or rebound
  std::vector<BaWindowDev*> Ws(nwin);
  for (int i = 0; i < nwin; i++) {
    Ws[i] = find_win(ctx, wins[i]);
    SDSO_REQUIRE(ctx, Ws[i], "unknown window in batch");
    SDSO_REQUIRE(ctx, Ws[i]->d.nf == Ws[0]->d.nf, "batch windows must share nf");
    SDSO_REQUIRE(ctx, (Ws[i]->d.tiledT > 0) == (Ws[0]->d.tiledT > 0), "batch windows must share the image layout");
    SDSO_REQUIRE(ctx, Ws[i]->solverMode == Ws[0]->solverMode, "batch windows must share solverMode (one lambda per launch)");
    for (int k = 0; k < i; k++) SDSO_REQUIRE(ctx, Ws[k] != Ws[i], "a window may appear only once in a batch");
  }
  const int nf = Ws[0]->d.nf;
  const size_t af = acc_floats(nf);
  BaDev* d_arr = nullptr; float* d_accum = nullptr;
  SDSO_HIP(ctx, hipMalloc(&d_arr, sizeof(BaDev) * nwin));
  if (hipMalloc(&d_accum, sizeof(float) * af * nwin) != hipSuccess) { hipFree(d_arr); return sdso::fail(ctx, SDSO_ERR_HIP, "hipMalloc of the batch accumulator block failed"); }
  BaBatch* Bt = new BaBatch();
  Bt->d_arr = d_arr; Bt->d_accum = d_accum; Bt->W = Ws;
  Bt->wins.assign(wins, wins + nwin);
  reg_get(g_batches, ctx) = Bt;
  hipMemsetAsync(Bt->d_accum, 0, sizeof(float) * af * nwin, ctx->stream);
  std::vector<BaDev> h(nwin);
  BaLaunch L{};
  L.nwin = nwin; L.nf = nf; L.n = Ws[0]->d.n; L.tiled = Ws[0]->d.tiledT > 0;
  for (int i = 0; i < nwin; i++) {
    BaWindowDev* W = Ws[i];
    W->d.accum = Bt->d_accum + af * i;   // contiguous accumulators: ONE all-reduce covers the batch
    W->in_batch = true;
    h[i] = W->d;
    hipMemcpyAsync(W->d_self, &W->d, sizeof(BaDev), hipMemcpyHostToDevice, ctx->stream);
    L.max_nblk_res = std::max(L.max_nblk_res, std::max(W->nblk_res, 1)); L.max_nblk_pts = std::max(L.max_nblk_pts, W->nblk_pts);
    L.max_chunks = std::max(L.max_chunks, W->d.nchunks); L.max_items = std::max(L.max_items, W->d.nitems);
    W->accumulated = true;
  }
  hipMemcpyAsync(Bt->d_arr, h.data(), sizeof(BaDev) * nwin, hipMemcpyHostToDevice, ctx->stream);
  L.d_arr = Bt->d_arr;
  L.any_lin = false;   // recomputed at every launch (marginalisation may linearize residuals of a member later)
  L.alt = (Ws[0]->solverMode & (SOLVER_SVD | SOLVER_ORTHOGONALIZE_SYSTEM)) != 0;   // (the members of a batch share one solverMode)
  L.Ws = Ws;
  Bt->L = L;
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) { free_batch(ctx); return sdso::fail(ctx, SDSO_ERR_HIP, "batch descriptor upload failed"); }
  return SDSO_OK;
}
namespace sdso {
static BaBatch* get_batch(sdso_ctx* ctx) { return ctx && reg_has(g_batches, ctx) ? reg_get(g_batches, ctx) : nullptr; }
// launch descriptor of the batch with the state-dependent flags refreshed
static const BaLaunch& batch_launch(BaBatch* Bt) {
  Bt->L.any_lin = false;
  for (BaWindowDev* W : Bt->W) if (W->has_lin_cached) { Bt->L.any_lin = true; break; }
  return Bt->L;
}
}  // namespace sdso
namespace sdso {
// comm.hip: the blocks the RCCL all-reduce sums in place
void* ba_batch_accum_block(sdso_ctx* ctx, size_t* nfloats) {
  BaBatch* Bt = get_batch(ctx);
  if (!Bt) return nullptr;
  *nfloats = acc_floats(Bt->L.nf) * Bt->wins.size();
  ensure_folded(ctx, Bt);
  return Bt->d_accum;
}
void* ba_window_accum_block(sdso_ctx* ctx, int win, size_t* nfloats) {
  BaWindowDev* W = find_win(ctx, win);
  if (!W) return nullptr;
  *nfloats = acc_floats(W->d.nf);
  W->accumulated = true;
  ensure_folded_win(ctx, W);
  return W->d.accum;
}
}  // namespace sdso
namespace sdso {
struct OptRun;
static bool batch_defers_fold(sdso_ctx* ctx, BaBatch* Bt);   // below, next to the resident loop
}
// phase 1 of one GN iteration for every window of the batch: linearize + applyRes + accumulate A/L/SC (enqueue only).
// Inside a single-rank resident loop (sdso_ba_batch_optimize_begin) the folds of the partial sums are left to the fused tail kernel of
// sdso_ba_batch_solve / sdso_ba_batch_solve_step; whoever else looks at the packed block gets it folded first (ensure_folded).
extern "C" int sdso_ba_batch_accumulate(sdso_ctx* ctx) {
  BaBatch* Bt = get_batch(ctx);
  if (!Bt) return sdso::fail(ctx, SDSO_ERR_STATE, "no batch");
  Bt->scattered = false;
  Bt->folded = launch_fused(ctx, batch_launch(Bt), Bt->materialize, 3, batch_defers_fold(ctx, Bt));
  mark_linearized(Bt->W, Bt->materialize);
  SDSO_HIP(ctx, hipGetLastError());
  return SDSO_OK;
}
// the two halves of sdso_ba_batch_accumulate as separate enqueues, for callers that overlap batches on several streams: the
// bandwidth-bound linearisation of one batch is best followed immediately by the linearisation of the next one, with the Schur
// accumulation and the folds of the first running underneath it
// CU-partitioned ctx (sdso_ctx_partition_cus): the launches inside the scope go to the ctx's aux stream, ordered behind everything the main
// stream holds so far; at the end of the scope the main stream is ordered behind them again (its next consumer — the next linearisation of
// THIS batch — needs their results anyway; another ctx's linearisation, on its own stream with the same large CU mask, does not wait).
struct AuxScope {
  sdso_ctx* ctx; hipStream_t main = nullptr;
  explicit AuxScope(sdso_ctx* c) : ctx(c) {
    if (!ctx->aux) return;
    hipEventRecord(ctx->ev_main, ctx->stream);
    hipStreamWaitEvent(ctx->aux, ctx->ev_main, 0);
    main = ctx->stream; ctx->stream = ctx->aux;
  }
  ~AuxScope() {
    if (!main) return;
    hipEventRecord(ctx->ev_aux, ctx->aux);
    ctx->stream = main;
    hipStreamWaitEvent(ctx->stream, ctx->ev_aux, 0);
  }
};
extern "C" int sdso_ba_batch_linearize(sdso_ctx* ctx) {
  BaBatch* Bt = get_batch(ctx);
  if (!Bt) return sdso::fail(ctx, SDSO_ERR_STATE, "no batch");
  launch_fused(ctx, batch_launch(Bt), Bt->materialize, 1);
  mark_linearized(Bt->W, Bt->materialize);
  SDSO_HIP(ctx, hipGetLastError());
  return SDSO_OK;
}
extern "C" int sdso_ba_batch_schur(sdso_ctx* ctx) {
  BaBatch* Bt = get_batch(ctx);
  if (!Bt) return sdso::fail(ctx, SDSO_ERR_STATE, "no batch");
  Bt->scattered = false;
  AuxScope aux(ctx);
  Bt->folded = launch_fused(ctx, batch_launch(Bt), Bt->materialize, 2, batch_defers_fold(ctx, Bt));
  SDSO_HIP(ctx, hipGetLastError());
  return SDSO_OK;
}
// materialize = 1 (default): every linearization also writes the RawResidualJacobian records to HBM
// (what PointFrameResidual::J holds in the reference); 0: they stay in registers (the solver never
// re-reads them) — 296 B less store traffic per point-residual.
extern "C" int sdso_ba_batch_set_materialize(sdso_ctx* ctx, int materialize) {
  BaBatch* Bt = get_batch(ctx);
  if (!Bt) return sdso::fail(ctx, SDSO_ERR_STATE, "no batch");
  Bt->materialize = materialize != 0;
  return SDSO_OK;
}
// phase 2: stitch + solve + resubstitute (enqueue only). Between the phases the caller may all-reduce
// the packed accumulators (sdso_ba_batch_accum_dev) across ranks.
extern "C" int sdso_ba_batch_solve(sdso_ctx* ctx, double lambda, int orthogonalize_x) {
  BaBatch* Bt = get_batch(ctx);
  if (!Bt) return sdso::fail(ctx, SDSO_ERR_STATE, "no batch");
  SDSO_REQUIRE(ctx, !Bt->scattered, "the accumulators were reduce-scattered by window (exchange mode 1): sdso_ba_batch_solve_step consumes them");
  // solveSystem's overrides of lambda (EnergyFunctional.cpp:840-846), as in the single-window call
  if (Bt->W[0]->solverMode & SOLVER_USE_GN) lambda = 0;
  if (Bt->W[0]->solverMode & SOLVER_FIX_LAMBDA) lambda = 1e-5;
  if ((Bt->W[0]->solverMode & (SOLVER_SVD | SOLVER_ORTHOGONALIZE_SYSTEM)) && solve_on_host()) {
    // SDSO_BA_SOLVE_HOST=1 (A/B): solveSystemF's SVD / orthogonalised-system branches (EnergyFunctional.cpp:876-900, 924-965) with the
    // assembly and the eigen-decomposition on the host, window by window (solve_system_host), the back-substitution on the device.  The
    // host mirrors (deltas, projector) are those of the upload.  Default: k_ba_solve_alt for the whole batch (launch_solve).
    ensure_folded(ctx, Bt);
    for (BaWindowDev* W : Bt->W) {
      const int rc = solve_system_host(ctx, W, orthogonalize_x ? 2 : 0, lambda);   // (iteration >= 2 is how the single call spells ORTHOGONALIZE_X_LATER)
      if (rc) return rc;
    }
    return SDSO_OK;
  }
  const bool no_tail = !tail_enabled() || batch_launch(Bt).alt;
  if (batch_launch(Bt).alt) {   // as the single call spells it for these branches: the argument is "iteration >= 2", the mode decides (EnergyFunctional.cpp:980)
    const int sm = Bt->W[0]->solverMode;
    orthogonalize_x = ((sm & SOLVER_ORTHOGONALIZE_X) || (orthogonalize_x && (sm & SOLVER_ORTHOGONALIZE_X_LATER))) ? 1 : 0;
  }
  launch_solve(ctx, batch_launch(Bt), lambda, orthogonalize_x, Bt->folded);   // (the tail kernel folds for itself: the block stays as it is)
  if (no_tail) Bt->folded = true;
  SDSO_HIP(ctx, hipGetLastError());
  return SDSO_OK;
}
extern "C" int sdso_ba_batch_accum_dev(sdso_ctx* ctx, void** dev_ptr, long* nfloats) {
  BaBatch* Bt = get_batch(ctx);
  if (!Bt) return sdso::fail(ctx, SDSO_ERR_STATE, "no batch");
  ensure_folded(ctx, Bt);
  Bt->eager_fold = true;          // the caller holds the address: every later accumulate leaves folded sums there
  if (dev_ptr) *dev_ptr = Bt->d_accum;
  if (nfloats) *nfloats = (long)(acc_floats(Bt->L.nf) * Bt->wins.size());
  return SDSO_OK;
}
extern "C" int sdso_ba_batch_get_x(sdso_ctx* ctx, double* x /* nwin * (8nf+4) */) {
  BaBatch* Bt = get_batch(ctx);
  if (!Bt) return sdso::fail(ctx, SDSO_ERR_STATE, "no batch");
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int n = Bt->L.n;
  for (size_t i = 0; i < Bt->wins.size(); i++) {
    BaWindowDev* W = find_win(ctx, Bt->wins[i]);
    SDSO_HIP(ctx, hipMemcpy(x + i * n, W->d.sol + 3 * ((size_t)n * n + n), sizeof(double) * n, hipMemcpyDeviceToHost));
  }
  return SDSO_OK;
}

// EnergyFunctional::marginalizeFrame (EnergyFunctional.cpp:554-660): drop frame `idx` from the marginalisation prior
// HM / bM by a scaled Schur complement.  ~70x70 doubles once per keyframe: host algebra, no device work.
extern "C" int sdso_ba_marginalize_frame(int nf, int idx, const double* prior8, const double* delta_prior8, const double* HM_in,
                                         const double* bM_in, double* HM_out, double* bM_out) {
  if (nf < 1 || idx < 0 || idx >= nf || !prior8 || !delta_prior8 || !HM_in || !bM_in || !HM_out || !bM_out) return SDSO_ERR_ARG;
  const int odim = nf * 8 + 4, ndim = odim - 8;
  // step 1: move the frame's 8 rows / columns to the end (order of the others unchanged)
  std::vector<int> ord;
  for (int i = 0; i < odim; i++) if (i < idx * 8 + 4 || i >= idx * 8 + 12) ord.push_back(i);
  for (int i = 0; i < 8; i++) ord.push_back(idx * 8 + 4 + i);
  std::vector<double> H((size_t)odim * odim), b(odim);
  for (int i = 0; i < odim; i++) { b[i] = bM_in[ord[i]]; for (int j = 0; j < odim; j++) H[(size_t)i * odim + j] = HM_in[(size_t)ord[i] * odim + ord[j]]; }
  // step 2: the frame's prior
  for (int i = 0; i < 8; i++) { H[(size_t)(ndim + i) * odim + ndim + i] += prior8[i]; b[ndim + i] += prior8[i] * delta_prior8[i]; }
  // step 3: scale, invert the 8x8 corner, Schur complement, unscale
  std::vector<double> S(odim), Si(odim);
  for (int i = 0; i < odim; i++) { S[i] = std::sqrt(std::fabs(H[(size_t)i * odim + i]) + 10); Si[i] = 1.0 / S[i]; }
  for (int i = 0; i < odim; i++) { b[i] = Si[i] * b[i]; for (int j = 0; j < odim; j++) H[(size_t)i * odim + j] = Si[i] * H[(size_t)i * odim + j] * Si[j]; }
  double A[8][8], inv[8][8];
  for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) { const double v = H[(size_t)(ndim + i) * odim + ndim + j]; A[i][j] = 0.5f * (v + v); inv[i][j] = i == j; }
  for (int k = 0; k < 8; k++) {   // Gauss-Jordan, partial pivoting (Eigen's fixed-size inverse() is PartialPivLU)
    int pv = k;
    for (int i = k + 1; i < 8; i++) if (std::fabs(A[i][k]) > std::fabs(A[pv][k])) pv = i;
    if (pv != k) for (int j = 0; j < 8; j++) { std::swap(A[k][j], A[pv][j]); std::swap(inv[k][j], inv[pv][j]); }
    const double d = A[k][k];
    for (int j = 0; j < 8; j++) { A[k][j] /= d; inv[k][j] /= d; }
    for (int i = 0; i < 8; i++) {
      if (i == k) continue;
      const double f = A[i][k];
      if (f == 0) continue;
      for (int j = 0; j < 8; j++) { A[i][j] -= f * A[k][j]; inv[i][j] -= f * inv[k][j]; }
    }
  }
  for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) inv[i][j] = 0.5f * (inv[i][j] + inv[i][j]);
  std::vector<double> bli((size_t)ndim * 8);   // bottomLeft^T * hpi
  for (int r = 0; r < ndim; r++)
    for (int c = 0; c < 8; c++) { double s = 0; for (int k = 0; k < 8; k++) s += H[(size_t)(ndim + k) * odim + r] * inv[k][c]; bli[(size_t)r * 8 + c] = s; }
  for (int r = 0; r < ndim; r++) {
    for (int c = 0; c < ndim; c++) { double s = 0; for (int k = 0; k < 8; k++) s += bli[(size_t)r * 8 + k] * H[(size_t)(ndim + k) * odim + c]; H[(size_t)r * odim + c] -= s; }
    double s = 0;
    for (int k = 0; k < 8; k++) s += bli[(size_t)r * 8 + k] * b[ndim + k];
    b[r] -= s;
  }
  for (int i = 0; i < odim; i++) { b[i] = S[i] * b[i]; for (int j = 0; j < odim; j++) H[(size_t)i * odim + j] = S[i] * H[(size_t)i * odim + j] * S[j]; }
  for (int r = 0; r < ndim; r++) { bM_out[r] = b[r]; for (int c = 0; c < ndim; c++) HM_out[(size_t)r * ndim + c] = 0.5 * (H[(size_t)r * odim + c] + H[(size_t)c * odim + r]); }
  return SDSO_OK;
}

// EnergyFunctional::marginalizeFrame (EnergyFunctional.cpp:554-660) on the window's DEVICE-resident prior (k_ba_marg_frame): what
// sdso_ba_marginalize_points left in dt_HM / dt_bM goes through the frame's marginalisation without visiting the host; the result stays in
// the window (BaWindowDev::d_marg) until the next window adopts it (sdso_ba_adopt_prior).  prior / delta_prior are the frame's own
// (EFFrame::prior, delta_prior = the host mirror's, as uploaded / as the resident loop left them).  HM_out / bM_out: optional copies.
extern "C" int sdso_ba_marginalize_frame_dev(sdso_ctx* ctx, int win, int idx, double* HM_out, double* bM_out) {
  GET_WIN();
  const int nf = W->d.nf, n = W->d.n;
  // Several frames may leave at one keyframe (FullSystem.cpp:1470-1476 calls marginalizeFrame for every flagged frame, each on the prior
  // the previous one left): a call that follows another one — with no sdso_ba_marginalize_points in between — continues from that result,
  // and `idx` then counts the frames the prior still covers, as the reference's frames[] does after the earlier frame was erased.
  if (!W->marg_chain) { W->marg_frames.resize(nf); std::iota(W->marg_frames.begin(), W->marg_frames.end(), 0); }
  const int cur = (int)W->marg_frames.size(), odim = 8 * cur + 4, m = odim - 8;
  SDSO_REQUIRE(ctx, idx >= 0 && idx < cur, "frame index out of range (it counts the frames the prior still covers)");
  if (!W->d_marg) { DM(W->d_marg, double, (size_t)n * n + n); DM(W->d_marg2, double, (size_t)n * n + n); }
  const HostFrame& Fm = W->frames[W->marg_frames[idx]];
  double pr[16];
  for (int i = 0; i < 8; i++) { pr[i] = Fm.prior[i]; pr[8 + i] = Fm.delta_prior[i]; }
  int rc = ensure_scratch(ctx, sizeof(pr));
  if (rc) return rc;
  SDSO_HIP(ctx, hipMemcpyAsync(ctx->scratch, pr, sizeof(pr), hipMemcpyHostToDevice, ctx->stream));
  const double* srcH = W->marg_chain ? W->d_marg : W->dt_HM;
  const double* srcb = W->marg_chain ? W->d_marg + (size_t)odim * odim : W->dt_bM;
  hipLaunchKernelGGL(k_ba_marg_frame, dim3(1), dim3(256), 0, ctx->stream, srcH, srcb, odim, idx, (const double*)ctx->scratch, W->d_marg2);
  SDSO_HIP(ctx, hipGetLastError());
  std::swap(W->d_marg, W->d_marg2);
  W->marg_frames.erase(W->marg_frames.begin() + idx);
  W->marg_chain = true;
  W->marg_dim = m;
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));     // (pr is stack-local)
  if (HM_out) SDSO_HIP(ctx, hipMemcpy(HM_out, W->d_marg, sizeof(double) * m * m, hipMemcpyDeviceToHost));
  if (bM_out) SDSO_HIP(ctx, hipMemcpy(bM_out, W->d_marg + (size_t)m * m, sizeof(double) * m, hipMemcpyDeviceToHost));
  return SDSO_OK;
}

// The next window takes over the prior sdso_ba_marginalize_frame_dev left in `from_win`: device to device, the new keyframe's 8 rows /
// columns zero — what EnergyFunctional::insertFrame does to HM / bM (EnergyFunctional.cpp:468-476: conservativeResize + setZero of the new
// rows and columns).  `win` must have been uploaded with HM = bM = NULL (zeros) and its LEADING frames must be the frames the prior covers,
// in the same order (checked by frameID): a prior attached to other frames is an error, never a silent result.
__global__ __launch_bounds__(256) void k_ba_prior_adopt(double* __restrict__ HM, double* __restrict__ bM, int n, const double* __restrict__ src, int m) {
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n * n + n; e += gridDim.x * 256) {
    if (e < n * n) { const int i = e / n, j = e - i * n; HM[e] = (i < m && j < m) ? src[(size_t)i * m + j] : 0.0; }
    else { const int i = e - n * n; bM[i] = i < m ? src[(size_t)m * m + i] : 0.0; }
  }
}
extern "C" int sdso_ba_adopt_prior(sdso_ctx* ctx, int win, int from_win) {
  GET_WIN();
  BaWindowDev* F = find_win(ctx, from_win);
  SDSO_REQUIRE(ctx, F && F->d_marg && F->marg_dim > 0, "the source window holds no marginalised prior (sdso_ba_marginalize_frame_dev first)");
  SDSO_REQUIRE(ctx, F != W, "a window cannot adopt its own prior");
  const int n = W->d.n, m = F->marg_dim, k = (int)F->marg_frames.size();
  SDSO_REQUIRE(ctx, m == 8 * k + 4 && m <= n, "the prior covers more frames than the window holds");
  SDSO_REQUIRE(ctx, W->prior_pristine, "the adopting window must have been uploaded with HM = bM = NULL and not have changed its prior since");
  for (int i = 0; i < k; i++)
    SDSO_REQUIRE(ctx, W->frames[i].frameID == F->frames[F->marg_frames[i]].frameID, "the window's leading frames are not the frames the prior covers (frameID mismatch)");
  hipLaunchKernelGGL(k_ba_prior_adopt, dim3(8), dim3(256), 0, ctx->stream, W->dt_HM, W->dt_bM, n, (const double*)F->d_marg, m);
  SDSO_HIP(ctx, hipGetLastError());
  W->hm_host_valid = false;
  W->accumulated = false;
  W->prior_pristine = false;
  W->marg_chain = false;
  return SDSO_OK;
}

// ------------------------------------------------------------------ device-resident Gauss-Newton loop (ba_opt.hip)
namespace sdso {
int comm_nranks(sdso_ctx* ctx);                                                            // comm.hip
int comm_rank(sdso_ctx* ctx);                                                              // comm.hip
bool comm_present(sdso_ctx* ctx);                                                          // comm.hip
int comm_allgather_floats(sdso_ctx* ctx, const float* send, float* recv, size_t nfloats);  // comm.hip
int comm_max_int(sdso_ctx* ctx, int* value);                                               // comm.hip

// scratch of the resident loop, one set per ctx (grown on demand, freed with the ctx's windows)
struct OptBufs {
  float* d_sums = nullptr; size_t sums_cap = 0;
  float* d_pack = nullptr; size_t pack_cap = 0;
  float* d_gather = nullptr; size_t gather_cap = 0;
  BaOptOut* d_out = nullptr; BaOptOut* h_out = nullptr; size_t out_cap = 0;
  float* d_lpart = nullptr; size_t lpart_cap = 0;
  float* d_solrec = nullptr; size_t solrec_cap = 0;
};
static std::map<sdso_ctx*, OptBufs*> g_optbufs;
void free_optbufs(sdso_ctx* ctx) {
  OptBufs* b = nullptr;
  if (!reg_take(g_optbufs, ctx, b) || !b) return;
  hipFree(b->d_sums); hipFree(b->d_pack); hipFree(b->d_gather); hipFree(b->d_out); hipFree(b->d_lpart); hipFree(b->d_solrec);
  if (b->h_out) hipHostFree(b->h_out);
  delete b;
}
template <class T> static int grow(sdso_ctx* ctx, T*& p, size_t& cap, size_t want) {
  if (want <= cap) return SDSO_OK;
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  hipFree(p); p = nullptr; cap = 0;
  SDSO_HIP(ctx, hipMalloc(&p, sizeof(T) * want));
  cap = want;
  return SDSO_OK;
}

// one resident loop on a ctx: a batch (sdso_ba_batch_optimize*) or a single window (sdso_ba_optimize)
struct OptRun {
  BaLaunch L{};
  std::vector<BaWindowDev*> W;
  bool materialize = true;
  int cap = 0, nranks = 1, sums_stride = 0, iteration = 0, stop = 1;
  bool exchange = false;   // pack + all-gather between the ranks (always when nranks > 1)
  bool gated = false;      // energy-gated flow (setting_forceAceptStep = false): un-fused kernels + k_ba_opt_gate
  int lstride = 0;         // floats between the windows' calcLEnergy partials
  bool active = false;
  bool local_only = false; // single-window call: never a collective, whatever communicator the ctx carries
  bool failed = false;     // a collective of the gated flow failed (sdso_last_error says which)
  bool keep_hs = false;    // every solve also writes lastHS / lastbS (EnergyFunctional.cpp:909-910): sdso_ba_get_post_state hands them out
  bool scatter_local = false;  // this rank's view: the batch asks for the reduce-scatter exchange and its loop can take it
  bool scatter = false;    // ... and every rank agreed (opt_begin)
  int momentum = 0;        // SOLVER_STEPMOMENTUM / SOLVER_MOMENTUM bits of the windows: k_ba_opt_momentum between solve and step, never the fused step
  OptBufs* B = nullptr;
};

static int opt_begin(sdso_ctx* ctx, OptRun& R, int stop_on_convergence) {
  const int nwin = (int)R.W.size(), nf = R.L.nf;
  SDSO_REQUIRE(ctx, nf >= 2, "the Gauss-Newton loop needs at least two keyframes (FullSystemOptimize.cpp:873)");
  int cap = 1;
  for (BaWindowDev* W : R.W) {
    SDSO_REQUIRE(ctx, (W->forceAccept != 0) == (R.W[0]->forceAccept != 0), "the windows of a resident loop must share setting_forceAceptStep");
    SDSO_REQUIRE(ctx, (W->solverMode & (SOLVER_SVD | SOLVER_ORTHOGONALIZE_SYSTEM)) == (R.W[0]->solverMode & (SOLVER_SVD | SOLVER_ORTHOGONALIZE_SYSTEM)), "the windows of a resident loop must share the solver branch");
    SDSO_REQUIRE(ctx, (W->solverMode & (SOLVER_MOMENTUM | SOLVER_STEPMOMENTUM)) == (R.W[0]->solverMode & (SOLVER_MOMENTUM | SOLVER_STEPMOMENTUM)), "the windows of a resident loop must share SOLVER_MOMENTUM / SOLVER_STEPMOMENTUM");
    cap = std::max(cap, W->d.nr - W->newest_first);
  }
  R.momentum = R.W[0]->solverMode & (SOLVER_MOMENTUM | SOLVER_STEPMOMENTUM);
  R.nranks = comm_nranks(ctx);
  // SDSO_OPT_FORCE_EXCHANGE: take the pack / all-gather path on a 1-rank communicator too (tests: the collectives of a 1-GPU box)
  R.exchange = !R.local_only && (R.nranks > 1 || (comm_present(ctx) && dbg_env("SDSO_OPT_FORCE_EXCHANGE") != nullptr));
  R.gated = !R.W[0]->forceAccept;
  if (R.exchange) { int rc = comm_max_int(ctx, &cap); if (rc) return rc; }
  R.cap = cap;
  // shape of the accumulators' exchange: one decision for the whole loop, the same on every rank or an error (never mismatched collectives)
  R.scatter = false;
  if (R.exchange) {
    const int want = (R.scatter_local && !R.gated && !R.keep_hs && tail_enabled() && !R.L.alt && !R.momentum && nwin % R.nranks == 0) ? 1 : 0;
    int hi = want, lo = -want;
    int rc = comm_max_int(ctx, &hi); if (rc) return rc;
    rc = comm_max_int(ctx, &lo); if (rc) return rc;
    SDSO_REQUIRE(ctx, hi == -lo, "the ranks disagree on the shape of the accumulators' exchange (sdso_ba_batch_exchange_mode / sdso_ba_batch_keep_system / SDSO_BA_TAIL differ between ranks)");
    R.scatter = want != 0;
  }
  R.sums_stride = 2 * (R.L.max_nblk_pts + 1);
  if (!reg_has(g_optbufs, ctx)) reg_get(g_optbufs, ctx) = new OptBufs();
  OptBufs* B = reg_get(g_optbufs, ctx);
  R.B = B;
  const size_t pf = opt_pack_floats(cap);
  int rc;
  if ((rc = grow(ctx, B->d_sums, B->sums_cap, (size_t)nwin * R.sums_stride))) return rc;
  if ((rc = grow(ctx, B->d_pack, B->pack_cap, (size_t)nwin * pf))) return rc;
  if (R.exchange && (rc = grow(ctx, B->d_gather, B->gather_cap, (size_t)R.nranks * nwin * pf))) return rc;
  R.lstride = R.L.max_chunks + R.L.max_nblk_pts + 1;
  if (R.gated && (rc = grow(ctx, B->d_lpart, B->lpart_cap, (size_t)nwin * R.lstride))) return rc;
  if ((size_t)nwin > B->out_cap) {
    SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    hipFree(B->d_out); if (B->h_out) hipHostFree(B->h_out);
    B->d_out = nullptr; B->h_out = nullptr; B->out_cap = 0;
    SDSO_HIP(ctx, hipMalloc(&B->d_out, sizeof(BaOptOut) * nwin));
    SDSO_HIP(ctx, hipHostMalloc(&B->h_out, sizeof(BaOptOut) * nwin));
    B->out_cap = nwin;
  }
  for (BaWindowDev* W : R.W) {
    BaOptDev& O = W->h_opt;
    std::memset(&O, 0, sizeof(O));
    for (int f = 0; f < nf; f++) {
      const HostFrame& F = W->frames[f];
      for (int i = 0; i < 10; i++) { O.state[f][i] = F.state[i]; O.state_backup[f][i] = F.state[i]; O.state_zero[f][i] = F.state_zero[i]; }
      for (int i = 0; i < 9; i++) O.evalPT[f][i] = F.evalPT.R[i];
      for (int i = 0; i < 3; i++) O.evalPT[f][9 + i] = F.evalPT.t[i];
      O.ab_exposure[f] = F.ab_exposure;
    }
    for (int i = 0; i < 4; i++) { O.calib_value[i] = W->calib.value[i]; O.calib_backup[i] = W->calib.value[i]; O.calib_zero[i] = W->calib.value_zero[i]; }
    O.newest_first = W->newest_first;
    O.lambda = 1e-1;
    O.stepsize = 1;                                                               // FullSystemOptimize.cpp:928
    for (double& v : O.previousX) v = std::numeric_limits<double>::quiet_NaN();   // :929
    H2D(W->d_opt, &W->h_opt, sizeof(BaOptDev));
  }
  hipLaunchKernelGGL(k_ba_reset_all, dim3(R.L.max_nblk_res, nwin), dim3(BA_BLOCK), 0, ctx->stream, R.L.d_arr);
  SDSO_HIP(ctx, hipGetLastError());
  R.iteration = 0; R.stop = stop_on_convergence; R.active = true;
  return SDSO_OK;
}

// pack -> [all-gather] -> k_ba_opt_step.  unfused: the energies come from k_ba_linearize's workgroups, not from the fused kernel's chunks
static int opt_consume(sdso_ctx* ctx, OptRun& R, int last, bool unfused, bool with_sums) {
  const int nwin = (int)R.W.size();
  OptBufs* B = R.B;
  const float* gathered = nullptr;     // single rank: k_ba_opt_step reads the energies where the kernels left them
  const float* sums = with_sums ? B->d_sums : (const float*)nullptr;
  if (R.exchange) {
    hipLaunchKernelGGL(k_ba_opt_pack, dim3(1, nwin), dim3(256), 0, ctx->stream, R.L.d_arr, B->d_pack, R.cap, unfused ? 1 : 0, sums, R.sums_stride);
    int rc = comm_allgather_floats(ctx, B->d_pack, B->d_gather, (size_t)nwin * opt_pack_floats(R.cap));
    if (rc) return rc;
    gathered = B->d_gather;
  }
  hipLaunchKernelGGL(k_ba_opt_step, dim3(1, nwin), dim3(256), 0, ctx->stream, R.L.d_arr, gathered, R.nranks, R.cap, R.iteration, last, R.stop, 1.0f, unfused ? 1 : 0, sums, R.sums_stride);
  SDSO_HIP(ctx, hipGetLastError());
  return SDSO_OK;
}

// after the solve of one iteration: doStepFromBackup for points, frames and calibration, tables, break test
static int opt_step(sdso_ctx* ctx, OptRun& R) {
  const int nwin = (int)R.W.size();
  if (R.momentum) hipLaunchKernelGGL(k_ba_opt_momentum, dim3(1, nwin), dim3(128), 0, ctx->stream, R.L.d_arr);   // the stepsize / the kept previous step of this iteration
  if (R.L.max_nblk_pts) hipLaunchKernelGGL(k_ba_points_op, dim3(R.L.max_nblk_pts, nwin), dim3(BA_BLOCK), 0, ctx->stream, R.L.d_arr, 3, R.momentum ? -1.0f : 1.0f, R.B->d_sums, R.sums_stride);
  int rc = opt_consume(ctx, R, 0, false, R.L.max_nblk_pts > 0);
  R.iteration++;
  return rc;
}

// ---- energy-gated flow (setting_forceAceptStep = false): the un-fused kernels, with the decision taken by k_ba_opt_gate on the device and
// the kernels of the two branches (applyRes / loadSateBackup + re-linearisation) launched unconditionally, each looking at the decision
static void gated_linearize(sdso_ctx* ctx, OptRun& R, int cond, int which) {
  const int nwin = (int)R.W.size();
  const dim3 g(R.L.max_nblk_res, nwin), b(BA_BLOCK);
  if (R.L.tiled) hipLaunchKernelGGL(k_ba_linearize<true>, g, b, 0, ctx->stream, R.L.d_arr, cond);
  else hipLaunchKernelGGL(k_ba_linearize<false>, g, b, 0, ctx->stream, R.L.d_arr, cond);
  mark_linearized(R.W, false);
  const int nblk = R.L.max_chunks + R.L.max_nblk_pts;
  if (nblk > 0) hipLaunchKernelGGL(k_ba_lenergy, dim3(nblk, nwin), b, 0, ctx->stream, R.L.d_arr, R.B->d_lpart, R.lstride, cond);
  if (R.exchange) {
    // sharded windows: every rank hands over its newest-frame energies, the energy of its residuals and its part of calcLEnergy; the
    // gate reads the gathered records rank by rank, so all ranks accept / reject together.  (Unconditional on every rank: a collective.)
    hipLaunchKernelGGL(k_ba_opt_pack, dim3(1, nwin), dim3(256), 0, ctx->stream, R.L.d_arr, R.B->d_pack, R.cap, 1, (const float*)nullptr, 0,
                       nblk > 0 ? (const float*)R.B->d_lpart : (const float*)nullptr, R.lstride);
    if (comm_allgather_floats(ctx, R.B->d_pack, R.B->d_gather, (size_t)nwin * opt_pack_floats(R.cap))) { R.failed = true; return; }
    hipLaunchKernelGGL(k_ba_opt_gate, dim3(1, nwin), dim3(256), 0, ctx->stream, R.L.d_arr, (const float*)R.B->d_lpart, R.lstride, which, R.stop,
                       (const float*)R.B->d_gather, R.nranks, R.cap);
    return;
  }
  hipLaunchKernelGGL(k_ba_opt_gate, dim3(1, nwin), dim3(256), 0, ctx->stream, R.L.d_arr, (const float*)R.B->d_lpart, R.lstride, which, R.stop);
}
static int opt_gated_start(sdso_ctx* ctx, OptRun& R) {   // linearizeAll(false) + the energies of the uploaded state + applyRes (:894-908)
  const int nwin = (int)R.W.size();
  gated_linearize(ctx, R, 0, 0);
  if (R.failed) return SDSO_ERR_STATE;
  hipLaunchKernelGGL(k_ba_apply, dim3(R.L.max_nblk_res, nwin), dim3(BA_BLOCK), 0, ctx->stream, R.L.d_arr, 0);
  SDSO_HIP(ctx, hipGetLastError());
  return SDSO_OK;
}
static int opt_gated_iteration(sdso_ctx* ctx, OptRun& R, int it) {
  const int nwin = (int)R.W.size();
  const dim3 gp(std::max(R.L.max_nblk_pts, 1), nwin), b(BA_BLOCK);
  if (R.L.max_nblk_pts) hipLaunchKernelGGL(k_ba_points_op, gp, b, 0, ctx->stream, R.L.d_arr, 0, 0.f, (float*)nullptr, 0, 0);   // backupState
  launch_accumulate(ctx, R.L, nullptr, false);
  if (R.exchange) {                                            // sharded windows: the packed accumulators of every rank, summed
    const int rc = sdso_ba_allreduce(ctx);
    if (rc) return rc;
  }
  const int sm = R.W[0]->solverMode;
  double lambda = 0;
  int flags = ((sm & SOLVER_ORTHOGONALIZE_X) || (it >= 2 && (sm & SOLVER_ORTHOGONALIZE_X_LATER))) ? 1 : 0;
  if (sm & SOLVER_FIX_LAMBDA) lambda = 1e-5;
  else if (sm & SOLVER_USE_GN) lambda = 0;
  else flags |= 2;                                             // the loop's own lambda, kept on the device (it depends on the decisions)
  launch_solve(ctx, R.L, lambda, flags);
  if (R.momentum) hipLaunchKernelGGL(k_ba_opt_momentum, dim3(1, nwin), dim3(128), 0, ctx->stream, R.L.d_arr);
  if (R.L.max_nblk_pts) hipLaunchKernelGGL(k_ba_points_op, gp, b, 0, ctx->stream, R.L.d_arr, 1, R.momentum ? -1.0f : 1.0f, R.B->d_sums, R.sums_stride, 0);
  if (R.exchange) {                                            // the break-test sums of every rank's points: pack -> all-gather -> step
    const int rc = opt_consume(ctx, R, 2, true, R.L.max_nblk_pts > 0);
    if (rc) return rc;
  } else
    hipLaunchKernelGGL(k_ba_opt_step, dim3(1, nwin), dim3(256), 0, ctx->stream, R.L.d_arr, (const float*)nullptr, 1, R.cap, it, 2, R.stop, 1.0f, 1,
                       R.L.max_nblk_pts ? (const float*)R.B->d_sums : (const float*)nullptr, R.sums_stride);
  gated_linearize(ctx, R, 0, 1);                               // trial linearisation, energies, decision
  hipLaunchKernelGGL(k_ba_apply, dim3(R.L.max_nblk_res, nwin), b, 0, ctx->stream, R.L.d_arr, 1);                                  // accepted: applyRes
  if (R.L.max_nblk_pts) hipLaunchKernelGGL(k_ba_points_op, gp, b, 0, ctx->stream, R.L.d_arr, 2, 0.f, (float*)nullptr, 0, 2);   // rejected: the points go back,
  gated_linearize(ctx, R, 2, 2);                               //           the restored state is linearised again and its energies kept
  if (R.failed) return SDSO_ERR_STATE;
  SDSO_HIP(ctx, hipGetLastError());
  R.iteration++;
  return SDSO_OK;
}

static int opt_collect(sdso_ctx* ctx, OptRun& R) {
  const int nwin = (int)R.W.size();
  hipLaunchKernelGGL(k_ba_opt_release, dim3(nwin), dim3(128), 0, ctx->stream, R.L.d_arr, R.B->d_out);
  SDSO_HIP(ctx, hipMemcpyAsync(R.B->h_out, R.B->d_out, sizeof(BaOptOut) * nwin, hipMemcpyDeviceToHost, ctx->stream));
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SDSO_OK;
}

// the end of FullSystem::optimize (FullSystemOptimize.cpp:993-1041): consume the linearisation at the final state, bring the host
// mirrors up to date, newest frame's setEvalPT, linearizeAll(true)
static int opt_finish(sdso_ctx* ctx, OptRun& R, sdso_ba_opt_result_t* out) {
  const int nwin = (int)R.W.size(), nf = R.L.nf;
  int rc = SDSO_OK;
  if (!R.gated) {   // (the gated loop leaves every window linearised at its final state)
    launch_fused(ctx, R.L, R.materialize, 1);
    mark_linearized(R.W, R.materialize);
    if ((rc = opt_consume(ctx, R, 1, false, false))) return rc;
  }
  if ((rc = opt_collect(ctx, R))) return rc;
  std::vector<int> its(nwin), resInA(nwin);
  // host mirrors + the tables at the final state: CPU-only per window (numeric nullspaces, adjoints, the gauge projector), spread
  // over host threads for a batch; the H2D enqueues follow on this thread
  auto finalize = [&](int w) {
    BaWindowDev* W = R.W[w];
    const BaOptOut& o = R.B->h_out[w];
    its[w] = o.iterations; resInA[w] = o.resInA;
    W->calib.setValue(o.calib_value);
    for (int f = 0; f < nf; f++) W->frames[f].setState(o.state[f]);
    W->frames[nf - 1].frameEnergyTH = o.frameTH_new;
    double nsz[10] = {0};
    nsz[6] = W->frames[nf - 1].state[6];
    nsz[7] = W->frames[nf - 1].state[7];
    W->frames[nf - 1].setEvalPT(W->frames[nf - 1].PRE_worldToCam, nsz);
    build_tables(W, true);
  };
  const int nthreads = std::max(1, std::min({nwin / 4, 16, (int)std::thread::hardware_concurrency()}));
  if (nthreads <= 1) for (int w = 0; w < nwin; w++) finalize(w);
  else {
    std::atomic<int> next{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; t++) pool.emplace_back([&] { for (int w; (w = next.fetch_add(1)) < nwin;) finalize(w); });
    for (std::thread& t : pool) t.join();
  }
  size_t tb = 0;
  for (BaWindowDev* W : R.W) tb = std::max(tb, (W->tbl_bytes + 255) & ~(size_t)255);
  char* tstage = nullptr;
  if ((rc = stage_reserve(ctx, tb * nwin, &tstage))) return rc;      // released for reuse by the synchronisation of opt_collect below
  for (int w = 0; w < nwin; w++) {
    BaWindowDev* W = R.W[w];
    if ((rc = upload_tables(ctx, W, true, false, true, tstage + tb * w))) return rc;
    if (W->in_batch) H2D(const_cast<BaDev*>(R.L.d_arr) + w, &W->d, sizeof(BaDev));   // the batch's descriptor copy carries the calibration scalars too
    W->accumulated = false;
  }
  if (R.L.tiled) hipLaunchKernelGGL(k_ba_linearize<true>, dim3(R.L.max_nblk_res, nwin), dim3(BA_BLOCK), 0, ctx->stream, R.L.d_arr);
  else hipLaunchKernelGGL(k_ba_linearize<false>, dim3(R.L.max_nblk_res, nwin), dim3(BA_BLOCK), 0, ctx->stream, R.L.d_arr);
  mark_linearized(R.W, false);
  hipLaunchKernelGGL(k_ba_apply, dim3(R.L.max_nblk_res, nwin), dim3(BA_BLOCK), 0, ctx->stream, R.L.d_arr);
  if ((rc = opt_consume(ctx, R, 1, true, false))) return rc;
  // linearizeAll_Reductor(true)'s per-residual bookkeeping (maxRelBaseline, numGoodResiduals; FullSystemOptimize.cpp:62-78) belongs to THIS
  // optimize call: it runs now, once, for every window — not when (and if) somebody asks for the post-state
  hipLaunchKernelGGL(k_ba_post_state, dim3(R.L.max_nblk_res, nwin), dim3(BA_BLOCK), 0, ctx->stream, R.L.d_arr, (float*)nullptr, 1);
  SDSO_HIP(ctx, hipGetLastError());
  if ((rc = opt_collect(ctx, R))) return rc;
  for (int w = 0; w < nwin; w++) {
    BaWindowDev* W = R.W[w];
    const BaOptOut& o = R.B->h_out[w];
    W->frames[nf - 1].frameEnergyTH = o.frameTH_new;
    W->resInL = o.resInL;
    sdso_ba_opt_result_t r;
    r.iterations = its[w];
    r.lastEnergy = o.lastEnergy;
    r.resInA = resInA[w];
    r.rmse = sqrtf((float)(o.lastEnergy / (8 * resInA[w])));
    if (out) out[w] = r;
    W->post_valid = true; W->hs_valid = R.keep_hs || R.gated || !tail_enabled() || R.L.alt; W->last_result = r;
  }
  R.active = false;
  return SDSO_OK;
}

static double opt_lambda(int iteration) { double l = 1e-1; for (int i = 0; i < iteration; i++) l *= 0.25; return l; }
static int opt_iterations(int nf, int mnumOptIts) {
  if (nf < 3) mnumOptIts = 20;
  if (nf < 4) mnumOptIts = 15;
  return mnumOptIts;
}
static bool batch_defers_fold(sdso_ctx* ctx, BaBatch* Bt) { (void)ctx; return tail_enabled() && !Bt->eager_fold && !Bt->L.alt; }

// solveSystem + doStepFromBackup + the loop's host part of iteration R.iteration.  Single rank: ONE launch of the fused tail kernel.
// Sharded windows: tail kernel (stitch, solve, resubstitute, points' step) -> pack -> all-gather -> k_ba_opt_step, as before.
static std::map<sdso_ctx*, OptRun*> g_optruns;   // the batch loop in flight between sdso_ba_batch_optimize_begin and _end
static int opt_solve_step(sdso_ctx* ctx, OptRun& R, double lambda, int orth, bool folded) {
  if (!tail_enabled() || R.L.alt || R.momentum) {   // (momentum: the stepsize / the kept step come between the solve and the step — opt_step)
    launch_solve(ctx, R.L, lambda, orth, folded);
    SDSO_HIP(ctx, hipGetLastError());
    return opt_step(ctx, R);
  }
  const int flags = ((orth & 1) ? TAIL_ORTH : 0) | (R.L.any_lin ? TAIL_TOPL : 0) | (folded ? 0 : TAIL_FOLD) | (R.keep_hs ? TAIL_HS : 0);
  const int nwin = (int)R.W.size();
  const dim3 gp(std::max(R.L.max_nblk_pts, 1), nwin);
  if (!R.exchange) {
    // the points' back-substitution and step inside the tail kernel (TAIL_RESUB) once every CU has a tail workgroup anyway: 143 -> 132 us
    // for the two at 256 windows; below that the separate kernel spreads a window's points over idle CUs (one window: 0.64 against
    // 0.70 ms per optimize).  SDSO_BA_TAIL_RESUB=0 / 1 forces one form (A/B)
    static const int fuse_env = dbg_env("SDSO_BA_TAIL_RESUB") ? atoi(dbg_env("SDSO_BA_TAIL_RESUB")) : -1;
    const bool fuse_resub = fuse_env >= 0 ? fuse_env != 0 : nwin >= (ctx->aux ? ctx->aux_cus : ctx->n_cu);   // (the CUs this launch may use)
    launch_tail(ctx, R.L, lambda, flags | TAIL_STEP | (fuse_resub ? TAIL_RESUB : 0), R.iteration, 0, R.stop);
    if (R.L.max_nblk_pts && !fuse_resub) { ProfScope ps(ctx, "k_ba_resub", 2); LAUNCH_RESUB_STEP(R.L, gp, dim3(BA_BLOCK), 0, ctx->stream, R.L.d_arr, R.iteration + 1, (float*)nullptr, 0); }
    SDSO_HIP(ctx, hipGetLastError());
    R.iteration++;
    return SDSO_OK;
  }
  BaBatch* Bt = R.W[0]->in_batch ? get_batch(ctx) : nullptr;
  if (Bt && Bt->scattered) {
    // reduce-scatter exchange: this rank holds the summed accumulators of its own windows only — it solves those, and the solutions
    // (x, xAd, nres: one record per window) go round by all-gather; every rank then steps its own points of every window, as below
    Bt->scattered = false;
    const int per = nwin / R.nranks, first = comm_rank(ctx) * per;
    BaLaunch own = R.L;
    own.d_arr = R.L.d_arr + first; own.nwin = per;
    launch_tail(ctx, own, lambda, flags);
    const size_t rf = (size_t)sol_rec_floats(R.L.n, R.L.nf);
    int rc = grow(ctx, R.B->d_solrec, R.B->solrec_cap, rf * nwin);
    if (rc) return rc;
    hipLaunchKernelGGL(k_ba_sol_record, dim3(nwin), dim3(256), 0, ctx->stream, R.L.d_arr, first, per, R.B->d_solrec, 0);
    if ((rc = comm_allgather_floats(ctx, R.B->d_solrec + rf * first, R.B->d_solrec, rf * per))) return rc;
    hipLaunchKernelGGL(k_ba_sol_record, dim3(nwin), dim3(256), 0, ctx->stream, R.L.d_arr, first, per, R.B->d_solrec, 1);
  } else
    launch_tail(ctx, R.L, lambda, flags);
  if (R.L.max_nblk_pts) LAUNCH_RESUB_STEP(R.L, gp, dim3(BA_BLOCK), 0, ctx->stream, R.L.d_arr, -1, R.B->d_sums, R.sums_stride);
  const int rc = opt_consume(ctx, R, 0, false, R.L.max_nblk_pts > 0);
  R.iteration++;
  return rc;
}
// one whole GN iteration: accumulate (fused linearisation + Schur part) -> [all-reduce] -> solve + step
static int opt_iteration(sdso_ctx* ctx, OptRun& R, int it) {
  BaBatch* Bt = R.W[0]->in_batch ? get_batch(ctx) : nullptr;
  const bool defer = tail_enabled() && !R.L.alt && !R.exchange && !(Bt && Bt->eager_fold);
  bool folded = launch_fused(ctx, R.L, R.materialize, 3, defer);
  mark_linearized(R.W, R.materialize);
  if (Bt) Bt->folded = folded;
  if (R.exchange) {
    int rc = Bt ? sdso_ba_allreduce(ctx) : SDSO_ERR_STATE;
    if (rc) return rc;
    folded = true;
  }
  const int sm = R.W[0]->solverMode;
  double lambda = opt_lambda(it);
  if (sm & SOLVER_USE_GN) lambda = 0;
  if (sm & SOLVER_FIX_LAMBDA) lambda = 1e-5;
  const int orth = (sm & SOLVER_ORTHOGONALIZE_X) || (it >= 2 && (sm & SOLVER_ORTHOGONALIZE_X_LATER));
  return opt_solve_step(ctx, R, lambda, orth ? 1 : 0, folded);
}

int optimize_resident_single(sdso_ctx* ctx, BaWindowDev* W, int mnumOptIts, sdso_ba_opt_result_t* res) {
  OptRun R;
  R.L = single(W); R.W = {W};
  // RawResidualJacobian records on demand: nothing inside the loop reads the records of a residual that is being re-linearised (the
  // accumulators take them from registers; linearised residuals keep the records fixLinearizationF saw), and the closing
  // linearizeAll(true) — k_ba_linearize + k_ba_apply in opt_finish — writes the records of the final state, which is what
  // PointFrameResidual::J / EFResidual::J hold when FullSystem::optimize returns.  296 B less store traffic per residual and iteration.
  R.materialize = false; R.keep_hs = true;
  // refused before anything is touched: opt_begin would already issue a collective and reset the window's residuals
  if (comm_nranks(ctx) > 1) return sdso::fail(ctx, SDSO_ERR_STATE, "sdso_ba_optimize is a single-rank call; sharded windows use sdso_ba_batch_optimize");
  R.local_only = true;
  int rc = opt_begin(ctx, R, 1);
  if (rc) return rc;
  const int N = opt_iterations(W->d.nf, mnumOptIts);
  if (R.gated) {
    if ((rc = opt_gated_start(ctx, R))) return rc;
    for (int it = 0; it < N; it++) if ((rc = opt_gated_iteration(ctx, R, it))) return rc;
    return opt_finish(ctx, R, res);
  }
  for (int it = 0; it < N; it++) if ((rc = opt_iteration(ctx, R, it))) return rc;
  return opt_finish(ctx, R, res);
}
// sdso_ba_allreduce asks: is this exchange the reduce-scatter by window?  Only inside the accepted-step resident loop of a sharded batch
// whose windows divide over the ranks, with the fused tail kernel, and without lastHS / lastbS being kept (they exist on the solving
// rank only); anything else takes the all-reduce, whatever mode the batch carries.
// The decision itself is taken ONCE, in sdso_ba_batch_optimize_begin, from this rank's state AND agreed on by all ranks (opt_begin:
// ranks that disagree — another exchange mode, SDSO_BA_TAIL, keep_system — would issue ncclReduceScatter against ncclAllReduce and hang).
bool ba_batch_scatter_wanted(sdso_ctx* ctx) {
  BaBatch* Bt = get_batch(ctx);
  if (!Bt || !reg_has(g_optruns, ctx)) return false;
  OptRun* R = reg_get(g_optruns, ctx);
  return R && R->active && R->scatter && R->W == Bt->W;
}
void ba_batch_scatter_done(sdso_ctx* ctx) {
  if (BaBatch* Bt = get_batch(ctx)) Bt->scattered = true;
}
void free_optrun(sdso_ctx* ctx) {
  OptRun* r = nullptr;
  if (reg_take(g_optruns, ctx, r) && r) delete r;
}
}  // namespace sdso

// FullSystem::optimize for every window of the batch, device-resident (no host round trip inside the loop):
//   begin : backupState's initial copy of the states on the device, resetOOB of every residual
//   then per iteration  sdso_ba_batch_accumulate -> [sdso_ba_allreduce] -> sdso_ba_batch_solve -> sdso_ba_batch_step
//   end   : the linearisation at the final state, setEvalPT of the newest frame, linearizeAll(true); results per window
// sdso_ba_batch_optimize runs the whole sequence with the reference's lambda / orthogonalisation schedule.
extern "C" int sdso_ba_batch_optimize_begin(sdso_ctx* ctx, int stop_on_convergence) {
  BaBatch* Bt = get_batch(ctx);
  if (!Bt) return sdso::fail(ctx, SDSO_ERR_STATE, "no batch");
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  free_optrun(ctx);
  OptRun* R = new OptRun();
  R->L = batch_launch(Bt); R->W = Bt->W; R->materialize = Bt->materialize; R->keep_hs = Bt->keep_system;
  R->scatter_local = Bt->exchange_mode == 1;
  int rc = opt_begin(ctx, *R, stop_on_convergence);
  if (rc) { delete R; return rc; }
  reg_get(g_optruns, ctx) = R;
  return SDSO_OK;
}
extern "C" int sdso_ba_batch_step(sdso_ctx* ctx) {
  if (!ctx || !reg_has(g_optruns, ctx)) return sdso::fail(ctx, SDSO_ERR_STATE, "sdso_ba_batch_optimize_begin first");
  OptRun* R = reg_get(g_optruns, ctx);
  SDSO_REQUIRE(ctx, get_batch(ctx) && get_batch(ctx)->W == R->W, "the batch changed since sdso_ba_batch_optimize_begin");
  SDSO_REQUIRE(ctx, !R->gated, "sdso_ba_batch_step drives the accepted-step flow; energy-gated windows run through sdso_ba_batch_optimize");
  return opt_step(ctx, *R);
}
// sdso_ba_batch_solve + sdso_ba_batch_step as ONE enqueue: solveSystemF, resubstituteF, doStepFromBackup, setPrecalcValues / setDeltaF /
// setNewFrameEnergyTH and the break test of the batch's resident loop run in one launch of the fused tail kernel (ba_tail.hip)
extern "C" int sdso_ba_batch_solve_step(sdso_ctx* ctx, double lambda, int orthogonalize_x) {
  if (!ctx || !reg_has(g_optruns, ctx)) return sdso::fail(ctx, SDSO_ERR_STATE, "sdso_ba_batch_optimize_begin first");
  OptRun* R = reg_get(g_optruns, ctx);
  BaBatch* Bt = get_batch(ctx);
  SDSO_REQUIRE(ctx, Bt && Bt->W == R->W, "the batch changed since sdso_ba_batch_optimize_begin");
  SDSO_REQUIRE(ctx, !R->gated, "sdso_ba_batch_solve_step drives the accepted-step flow; energy-gated windows run through sdso_ba_batch_optimize");
  if (Bt->W[0]->solverMode & SOLVER_USE_GN) lambda = 0;
  if (Bt->W[0]->solverMode & SOLVER_FIX_LAMBDA) lambda = 1e-5;
  R->L = batch_launch(Bt);
  AuxScope aux(ctx);
  return opt_solve_step(ctx, *R, lambda, orthogonalize_x ? 1 : 0, Bt->folded);
}
extern "C" int sdso_ba_batch_optimize_end(sdso_ctx* ctx, sdso_ba_opt_result_t* out) {
  if (!ctx || !reg_has(g_optruns, ctx)) return sdso::fail(ctx, SDSO_ERR_STATE, "sdso_ba_batch_optimize_begin first");
  OptRun* R = reg_get(g_optruns, ctx);
  SDSO_REQUIRE(ctx, get_batch(ctx) && get_batch(ctx)->W == R->W, "the batch changed since sdso_ba_batch_optimize_begin");
  R->L = batch_launch(get_batch(ctx));
  const int rc = opt_finish(ctx, *R, out);
  free_optrun(ctx);
  return rc;
}
extern "C" int sdso_ba_batch_optimize(sdso_ctx* ctx, int mnumOptIts, sdso_ba_opt_result_t* out) {
  if (BaBatch* Bt = get_batch(ctx)) {
    if ((Bt->W[0]->solverMode & (SOLVER_SVD | SOLVER_ORTHOGONALIZE_SYSTEM)) && solve_on_host()) {
      // SDSO_BA_SOLVE_HOST=1 (A/B): the SVD / orthogonalised-system solver modes host-driven (one round trip per iteration,
      // solve_system_host) — the batch call runs the single-window host loop window by window.  Default: the resident loop below, with
      // k_ba_solve_alt in the place of the tail kernel's stitch and solve
      SDSO_HIP(ctx, hipSetDevice(ctx->device));
      free_optrun(ctx);
      for (size_t i = 0; i < Bt->wins.size(); i++) {
        const int rc = sdso_ba_optimize(ctx, Bt->wins[i], mnumOptIts, nullptr, nullptr, nullptr, out ? &out[i] : nullptr);
        if (rc) return rc;
      }
      Bt->folded = true;
      return SDSO_OK;
    }
  }
  int rc = sdso_ba_batch_optimize_begin(ctx, 1);
  if (rc) return rc;
  OptRun* R = reg_get(g_optruns, ctx);
  const int N = opt_iterations(R->L.nf, mnumOptIts);
  if (R->gated) {
    if ((rc = opt_gated_start(ctx, *R))) { free_optrun(ctx); return rc; }
    for (int it = 0; it < N; it++) if ((rc = opt_gated_iteration(ctx, *R, it))) { free_optrun(ctx); return rc; }
    return sdso_ba_batch_optimize_end(ctx, out);
  }
  for (int it = 0; it < N; it++) {
    if ((rc = opt_iteration(ctx, *R, it))) { free_optrun(ctx); return rc; }
  }
  return sdso_ba_batch_optimize_end(ctx, out);
}
// FrameHessian::state, PointHessian::idepth and the residual states of one window as they stand (after sdso_ba_optimize /
// sdso_ba_batch_optimize); synchronises
extern "C" int sdso_ba_get_state(sdso_ctx* ctx, int win, double* state_out /* nf*10 */, float* idepth_out /* np */, uint8_t* res_state_out /* nr */) {
  GET_WIN();
  const int nf = W->d.nf, np = W->d.np, nr = W->d.nr;
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (state_out) for (int f = 0; f < nf; f++) for (int i = 0; i < 10; i++) state_out[f * 10 + i] = W->frames[f].state[i];
  if (idepth_out && np) {
    std::vector<float4> geo(np);
    SDSO_HIP(ctx, hipMemcpy(geo.data(), W->d.p_geo, sizeof(float4) * np, hipMemcpyDeviceToHost));
    for (int p = 0; p < np; p++) idepth_out[p] = geo[p].z;
  }
  if (res_state_out && nr) {
    std::vector<uint8_t> t(nr);
    SDSO_HIP(ctx, hipMemcpy(t.data(), W->d.r_state, nr, hipMemcpyDeviceToHost));
    for (int j = 0; j < nr; j++) res_state_out[W->perm[j]] = t[j];
  }
  return SDSO_OK;
}

extern "C" int sdso_ba_batch_exchange_mode(sdso_ctx* ctx, int mode) {
  BaBatch* Bt = get_batch(ctx);
  if (!Bt) return sdso::fail(ctx, SDSO_ERR_STATE, "no batch");
  SDSO_REQUIRE(ctx, mode == 0 || mode == 1, "exchange mode: 0 all-reduce, 1 reduce-scatter by window");
  Bt->exchange_mode = mode;
  return SDSO_OK;
}

extern "C" int sdso_ba_batch_keep_system(sdso_ctx* ctx, int on) {
  BaBatch* Bt = get_batch(ctx);
  if (!Bt) return sdso::fail(ctx, SDSO_ERR_STATE, "no batch");
  Bt->keep_system = on != 0;
  return SDSO_OK;
}

// Everything FullSystem::optimize leaves behind for its callers (include/sdso_abi.h: sdso_ba_post_state_t).  The per-residual part of
// linearizeAll_Reductor(true) (maxRelBaseline, numGoodResiduals; FullSystemOptimize.cpp:62-78) runs here, once per optimize call.
extern "C" int sdso_ba_get_post_state(sdso_ctx* ctx, int win, sdso_ba_post_state_t* out) {
  GET_WIN();
  SDSO_REQUIRE(ctx, out, "null post-state");
  SDSO_REQUIRE(ctx, W->post_valid, "sdso_ba_get_post_state needs a finished sdso_ba_optimize / sdso_ba_batch_optimize on this window");
  SDSO_REQUIRE(ctx, (!out->lastHS && !out->lastbS) || W->hs_valid, "lastHS / lastbS were not kept: sdso_ba_batch_keep_system(ctx, 1) before the batch loop");
  const int nf = W->d.nf, np = W->d.np, nr = W->d.nr, n = W->d.n;
  if (nr && (out->centerProjectedTo || out->projectedTo)) {
    // (the projections are re-evaluated on every call that asks for them; the counters moved when the optimize call ended)
    if (!W->d_post) { DM(W->d_post, float, (size_t)std::max(nr, 1) * 19); }
    hipLaunchKernelGGL(k_ba_post_state, dim3(std::max(W->nblk_res, 1), 1), dim3(BA_BLOCK), 0, ctx->stream, (const BaDev*)W->d_self, W->d_post, 0);
    SDSO_HIP(ctx, hipGetLastError());
  }
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // ---- points
  if (np && (out->idepth || out->step || out->HdiF || out->bdSumF || out->idepth_hessian || out->maxRelBaseline || out->numGoodResiduals)) {
    std::vector<float4> geo(np), tr(np);
    std::vector<float> po((size_t)np * 16);
    SDSO_HIP(ctx, hipMemcpy(geo.data(), W->d.p_geo, sizeof(float4) * np, hipMemcpyDeviceToHost));
    SDSO_HIP(ctx, hipMemcpy(tr.data(), W->d.p_track, sizeof(float4) * np, hipMemcpyDeviceToHost));
    SDSO_HIP(ctx, hipMemcpy(po.data(), W->d.p_out, sizeof(float) * po.size(), hipMemcpyDeviceToHost));
    for (int p = 0; p < np; p++) {
      const float* o = &po[(size_t)p * 16];
      if (out->idepth) out->idepth[p] = geo[p].z;
      if (out->step) out->step[p] = o[PO_STEP];
      if (out->HdiF) out->HdiF[p] = o[PO_HDI];
      if (out->bdSumF) out->bdSumF[p] = o[PO_BDSUM];
      if (out->idepth_hessian) out->idepth_hessian[p] = tr[p].z;
      if (out->maxRelBaseline) out->maxRelBaseline[p] = tr[p].x;
      if (out->numGoodResiduals) std::memcpy(&out->numGoodResiduals[p], &tr[p].y, 4);
    }
  }
  // ---- residuals (pair-sorted on the device -> the window's order)
  out->n_toRemove = 0;
  if (nr) {
    std::vector<uint8_t> st(nr), act(nr), lin(nr);
    SDSO_HIP(ctx, hipMemcpy(st.data(), W->d.r_state, nr, hipMemcpyDeviceToHost));
    SDSO_HIP(ctx, hipMemcpy(act.data(), W->d.r_act, nr, hipMemcpyDeviceToHost));
    SDSO_HIP(ctx, hipMemcpy(lin.data(), W->d.r_lin, nr, hipMemcpyDeviceToHost));
    for (int j = 0; j < nr; j++) {
      const int o = W->perm[j];
      const bool rem = !(lin[j] & 1) && !act[j];      // in activeResiduals and not isActive(): toRemove (:80-84)
      if (out->state_state) out->state_state[o] = st[j];
      if (out->isActiveAndIsGoodNEW) out->isActiveAndIsGoodNEW[o] = act[j];
      if (out->toRemove) out->toRemove[o] = rem ? 1 : 0;
      out->n_toRemove += rem ? 1 : 0;
    }
    if (out->state_energy) {
      std::vector<float> e(nr);
      SDSO_HIP(ctx, hipMemcpy(e.data(), W->d.r_energy, sizeof(float) * nr, hipMemcpyDeviceToHost));
      for (int j = 0; j < nr; j++) out->state_energy[W->perm[j]] = e[j];
    }
    if (out->centerProjectedTo || out->projectedTo) {
      std::vector<float> pj((size_t)nr * 19);
      SDSO_HIP(ctx, hipMemcpy(pj.data(), W->d_post, sizeof(float) * pj.size(), hipMemcpyDeviceToHost));
      for (int j = 0; j < nr; j++) {
        if (out->projectedTo) std::memcpy(out->projectedTo + (size_t)W->perm[j] * 16, &pj[(size_t)j * 19], 64);
        if (out->centerProjectedTo) std::memcpy(out->centerProjectedTo + (size_t)W->perm[j] * 3, &pj[(size_t)j * 19 + 16], 12);
      }
    }
  }
  // ---- frames, calibration (host mirror: brought up to date when the loop ended)
  std::vector<double> x(n);
  SDSO_HIP(ctx, hipMemcpy(x.data(), W->d.sol + 3 * ((size_t)n * n + n), sizeof(double) * n, hipMemcpyDeviceToHost));
  for (int f = 0; f < nf; f++) {
    const HostFrame& F = W->frames[f];
    for (int i = 0; i < 10; i++) {
      if (out->state) out->state[f * 10 + i] = F.state[i];
      if (out->state_zero) out->state_zero[f * 10 + i] = F.state_zero[i];
      if (out->frame_step) out->frame_step[f * 10 + i] = i < 8 ? -x[4 + 8 * f + i] : 0.0;   // EnergyFunctional.cpp:283-286
    }
    if (out->evalPT) { std::memcpy(out->evalPT + f * 12, F.evalPT.R.data(), 72); std::memcpy(out->evalPT + f * 12 + 9, F.evalPT.t.data(), 24); }
    if (out->PRE_worldToCam) { std::memcpy(out->PRE_worldToCam + f * 12, F.PRE_worldToCam.R.data(), 72); std::memcpy(out->PRE_worldToCam + f * 12 + 9, F.PRE_worldToCam.t.data(), 24); }
    if (out->frameEnergyTH) out->frameEnergyTH[f] = F.frameEnergyTH;
  }
  for (int i = 0; i < 4; i++) { out->calib_value[i] = W->calib.value[i]; out->calib_value_scaled[i] = W->calib.value_scaled[i]; out->calib_step[i] = -x[i]; }
  if (out->lastX) std::memcpy(out->lastX, x.data(), sizeof(double) * n);
  const double* hsb = W->d.sol + 3 * ((size_t)n * n + n) + n;
  if (out->lastHS) SDSO_HIP(ctx, hipMemcpy(out->lastHS, hsb, sizeof(double) * n * n, hipMemcpyDeviceToHost));
  if (out->lastbS) SDSO_HIP(ctx, hipMemcpy(out->lastbS, hsb + (size_t)n * n, sizeof(double) * n, hipMemcpyDeviceToHost));
  // (resInL: nres[0] of the last accumulateLF, EnergyFunctional.cpp:241 — recorded when the optimize call ended, next to resInA)
  out->resInA = W->last_result.resInA; out->resInL = W->resInL; out->resInM = W->resInM;
  out->result = W->last_result;
  return SDSO_OK;
}

// EnergyFunctional::resInA / resInL (nres[0] of the latest accumulateAF / LF, EnergyFunctional.cpp:219, :241) and resInM (residuals
// marginalised through this window so far, :704).  Any pointer may be NULL.
extern "C" int sdso_ba_get_counts(sdso_ctx* ctx, int win, int* resInA, int* resInL, int* resInM) {
  GET_WIN();
  SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (resInA || resInL) {
    ensure_folded_win(ctx, W);
    SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float nres2[2] = {0, 0};
    SDSO_HIP(ctx, hipMemcpy(nres2, W->d.accum + acc_off_nres(W->d.nf), sizeof(nres2), hipMemcpyDeviceToHost));
    if (resInA) *resInA = (int)nres2[0];
    if (resInL) *resInL = (int)nres2[1];
  }
  if (resInM) *resInM = W->resInM;
  return SDSO_OK;
}
