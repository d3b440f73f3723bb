// Exercises host/sdso_shim.h the way the reference's classes would call it, with small stand-ins
// for the Eigen / Sophus / DSO types (none of which exist in this image).  tests/test_host_shim.py
// writes a problem as raw arrays into a directory, runs this program on the GPU box and compares
// what it prints with the same problem pushed through the C-ABI from Python.
//
//   test_shim <dir> tracker|tracker_g2o|tracker_ref|stereo|stereo_g2o|ba|ba_members|selector
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <fstream>
#include <memory>
#include "sdso_shim.h"

template <class T>
static std::vector<T> load(const std::string& dir, const char* name) {
  std::ifstream f(dir + "/" + name + ".bin", std::ios::binary | std::ios::ate);
  if (!f) { std::fprintf(stderr, "missing %s\n", name); std::exit(2); }
  const size_t bytes = (size_t)f.tellg();
  std::vector<T> v(bytes / sizeof(T));
  f.seekg(0);
  f.read(reinterpret_cast<char*>(v.data()), bytes);
  return v;
}

// ---- stand-ins with the reference's member names ------------------------------------------------
struct Mat33 { double m[9]; double& operator()(int i, int j) { return m[i * 3 + j]; } double operator()(int i, int j) const { return m[i * 3 + j]; } };
struct Vec3 { double v[3]; double& operator[](int i) { return v[i]; } double operator[](int i) const { return v[i]; } };
struct Mat33f { float m[9]; float operator()(int i, int j) const { return m[i * 3 + j]; } };
struct Mat22f { float m[4]; float operator()(int i, int j) const { return m[i * 2 + j]; } };
struct SE3 {
  Mat33 R; Vec3 t;
  SE3() { for (int i = 0; i < 9; i++) R.m[i] = (i % 4 == 0); t = {{0, 0, 0}}; }
  SE3(const Mat33& R_, const Vec3& t_) : R(R_), t(t_) {}
  const Mat33& rotationMatrix() const { return R; }
  const Vec3& translation() const { return t; }
};
struct AffLight { double a = 0, b = 0; };
struct Vec3f { float v[3]; };
struct Vec10 { double v[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; double& operator[](int i) { return v[i]; } double operator[](int i) const { return v[i]; } };
struct VecC { double v[4] = {0, 0, 0, 0}; double& operator[](int i) { return v[i]; } double operator[](int i) const { return v[i]; } };
struct Vec2f { float v[2] = {0, 0}; float& operator[](int i) { return v[i]; } };
struct Vec3fv { float v[3] = {0, 0, 0}; float& operator[](int i) { return v[i]; } };
struct CalibHessian {
  VecC value_scaled, value_zero, value, step;
  void setValue(const VecC& val) {                               // HessianBlocks.h:318-333 (SCALE_F = SCALE_C = 50)
    value = val;
    for (int i = 0; i < 4; i++) value_scaled[i] = 50.0 * val[i];
  }
  float fxl() const { return (float)value_scaled[0]; } float fyl() const { return (float)value_scaled[1]; }
  float cxl() const { return (float)value_scaled[2]; } float cyl() const { return (float)value_scaled[3]; }
};
struct PointHessian;
struct FrameShell { int id = 0; };
struct FrameHessian {
  Vec3f* dIp[SDSO_PYR_LEVELS];
  std::vector<std::vector<float>> store;
  SE3 worldToCam_evalPT; Vec10 state, state_zero, step;
  float ab_exposure = 1, frameEnergyTH = 0; int frameID = 0, idx = 0, slot = 0;
  std::vector<PointHessian*> pointHessians;                      // HessianBlocks.h:125
  FrameShell shellStore; FrameShell* shell = &shellStore;
  AffLight aff; AffLight aff_g2l() const { return aff; }         // HessianBlocks.h:185
  const SE3& get_worldToCam_evalPT() const { return worldToCam_evalPT; }
  const Vec10& get_state() const { return state; }
  const Vec10& get_state_zero() const { return state_zero; }
  void setState(const Vec10& s) { state = s; }                   // (the reference also refreshes state_scaled / PRE_worldToCam here: host math)
  void setEvalPT(const SE3& T, const Vec10& s) { worldToCam_evalPT = T; state = s; state_zero = s; }   // HessianBlocks.h:216-222
};
struct EFFrame; struct EFPoint; struct EFResidual; struct PointHessian;
struct PointFrameResidual {
  int state_state = 0, state_NewState = 0;
  double state_energy = 0, state_NewEnergy = 0, state_NewEnergyWithOutlier = 0;
  bool isNew = true;
  Vec2f projectedTo[SDSO_MAX_RES];
  Vec3fv centerProjectedTo;
  PointHessian* point = nullptr; EFResidual* efResidual = nullptr;
  int id = -1;                                                   // index in the uploaded window (the test's bookkeeping)
};
struct PointHessian {
  float u, v, idepth, idepth_zero, color[8], weights[8], step = 0, idepth_hessian = 0, maxRelBaseline = 0;
  int numGoodResiduals = 0;
  bool hasDepthPrior = false;
  std::vector<PointFrameResidual*> residuals;
  std::pair<PointFrameResidual*, int> lastResiduals[2] = {{nullptr, 2}, {nullptr, 2}};
  EFPoint* efPoint = nullptr;
  void setIdepth(float x) { idepth = x; }
  void setIdepthZero(float x) { idepth_zero = x; }
  bool isInlierNew() const { return (int)residuals.size() >= 3 && numGoodResiduals >= 4; }   // HessianBlocks.h:465-469; setting_minGoodActiveResForMarg = 3, setting_minGoodResForMarg = 4 (settings.cpp:82-83)
};
struct EFResidual { PointFrameResidual* data; EFFrame* target; bool isActiveAndIsGoodNEW = false; int idxInAll = 0; EFPoint* point = nullptr; bool isLinearized = false; };
struct EFPoint { PointHessian* data; std::vector<EFResidual*> residualsAll; int stateFlag = 0; float HdiF = 0, bdSumF = 0, deltaF = 0; };
struct Vec8 { double v[8] = {0, 0, 0, 0, 0, 0, 0, 0}; double& operator[](int i) { return v[i]; } };
struct EFFrame { FrameHessian* data; std::vector<EFPoint*> points; int idx; Vec8 delta, delta_prior; };
struct Mat88 { double m[64]; double& operator()(int i, int j) { return m[i * 8 + j]; } };
struct Mat88f { float m[64]; float& operator()(int i, int j) { return m[i * 8 + j]; } };
struct Mat18f { float m[8]; float& operator()(int, int j) { return m[j]; } };
struct DynMat {
  int n = 0; std::vector<double> d;
  void resize(int r, int c) { n = c; d.assign((size_t)r * c, 0.0); }
  double& operator()(int i, int j) { return d[(size_t)i * n + j]; }
};
struct DynVec {
  std::vector<double> d;
  void resize(int r) { d.assign((size_t)r, 0.0); }
  double& operator()(int i) { return d[(size_t)i]; }
};
struct EnergyFunctional {
  std::vector<EFFrame*> frames; DynMat HM, lastHS; std::vector<double> bM, lastbS, lastX;
  int resInA = 0, resInL = 0, resInM = 0, nResiduals = 0;
  Mat88 *adHost = 0, *adTarget = 0; Mat88f *adHostF = 0, *adTargetF = 0; Mat18f* adHTdeltaF = 0;   // EnergyFunctional.h:115-135
  float cDeltaF[4] = {0, 0, 0, 0};
  std::vector<EFPoint*> allPoints;
  ~EnergyFunctional() { delete[] adHost; delete[] adTarget; delete[] adHostF; delete[] adTargetF; delete[] adHTdeltaF; }
  void dropResidual(EFResidual* r) {                             // EnergyFunctional.cpp:519-548
    EFPoint* p = r->point;
    p->residualsAll[r->idxInAll] = p->residualsAll.back();
    p->residualsAll[r->idxInAll]->idxInAll = r->idxInAll;
    p->residualsAll.pop_back();
    nResiduals--;
    r->data->efResidual = nullptr;
    delete r;
  }
};
struct ImmaturePoint {
  float u, v, idepth_max;
  float u_stereo, v_stereo, idepth_min, idepth_min_stereo, idepth_max_stereo, idepth_stereo, energyTH, quality, color[8], weights[8];
  Mat22f gradH; int lastTraceStatus; float lastTraceUV[2]; float lastTracePixelInterval;
};
struct Vec5 { double v[5]; double operator[](int i) const { return v[i]; } };
// compile the temporal-trace wrapper against the stand-in as well (it is exercised through the C-ABI in tests/test_stereo.py)
template void sdso_shim::traceOnAll<ImmaturePoint>(sdso_shim::Device&, std::vector<ImmaturePoint*>&, const std::vector<int>&, const std::vector<sdso_trace_geom_t>&, int,
                                                   std::vector<uint8_t>&);

struct Vec3fc { float v[3]; float operator[](int i) const { return v[i]; } };
struct Vec2fc { float v[2]; float operator[](int i) const { return v[i]; } };
// ... and ImmaturePoint::traceOn's one-point form (ImmaturePoint.h:90)
template int sdso_shim::traceOn<ImmaturePoint, Mat33f, Vec3fc, Vec2fc>(sdso_shim::Device&, ImmaturePoint*, int, const Mat33f&, const Vec3fc&, const Vec2fc&);

static void load_pyramid(const std::string& dir, const char* prefix, FrameHessian& fh, int levels) {
  fh.store.resize(levels);
  for (int l = 0; l < levels; l++) {
    char nm[64]; std::snprintf(nm, sizeof nm, "%s_l%d", prefix, l);
    fh.store[l] = load<float>(dir, nm);
    fh.dIp[l] = reinterpret_cast<Vec3f*>(fh.store[l].data());
  }
}

static int run_tracker(const std::string& dir, bool fork_live) {
  auto meta = load<int>(dir, "meta");             // levels, w0, h0, coarsestLvl
  auto calib = load<double>(dir, "calib");        // fx fy cx cy
  auto misc = load<double>(dir, "misc");          // ref_exposure new_exposure ref_a ref_b, T0 (12), aff0 (2), minResForAbort (5)
  const int levels = meta[0], w0 = meta[1], h0 = meta[2];
  sdso_shim::Device dev(0);
  FrameHessian ref, cur;
  load_pyramid(dir, "ref", ref, levels); load_pyramid(dir, "new", cur, levels);
  int w[SDSO_PYR_LEVELS], h[SDSO_PYR_LEVELS];
  for (int l = 0; l < levels; l++) { w[l] = w0 >> l; h[l] = h0 >> l; }
  dev.uploadFrame(0, &ref, levels, w, h); dev.uploadFrame(1, &cur, levels, w, h);
  CalibHessian HC; for (int i = 0; i < 4; i++) HC.value_scaled[i] = HC.value_zero[i] = calib[i];
  sdso_shim::CoarseTracker<SE3, AffLight, Mat33, Vec3> tracker(dev, 0);
  tracker.makeK(&HC, levels, w0, h0);
  tracker.forkLive = fork_live;
  AffLight refAff; refAff.a = misc[2]; refAff.b = misc[3];
  for (int l = 0; l < levels; l++) {
    char nm[32];
    std::snprintf(nm, sizeof nm, "pc_u_l%d", l); auto pu = load<float>(dir, nm);
    std::snprintf(nm, sizeof nm, "pc_v_l%d", l); auto pv = load<float>(dir, nm);
    std::snprintf(nm, sizeof nm, "pc_idepth_l%d", l); auto pi = load<float>(dir, nm);
    std::snprintf(nm, sizeof nm, "pc_color_l%d", l); auto pc = load<float>(dir, nm);
    tracker.setCoarseTrackingRef(l, (int)pu.size(), pu.data(), pv.data(), pi.data(), pc.data(), (float)misc[0], refAff, 7);
  }
  Mat33 R; Vec3 t;
  for (int i = 0; i < 9; i++) R.m[i] = misc[4 + i];
  for (int i = 0; i < 3; i++) t.v[i] = misc[13 + i];
  SE3 lastToNew(R, t);
  AffLight aff; aff.a = misc[16]; aff.b = misc[17];
  Vec5 minRes; for (int i = 0; i < 5; i++) minRes.v[i] = misc[18 + i];
  const bool good = tracker.trackNewestCoarse(1, (float)misc[1], lastToNew, aff, meta[3], minRes);
  std::printf("good %d\n", good ? 1 : 0);
  std::printf("T"); for (int i = 0; i < 9; i++) std::printf(" %.17g", lastToNew.R.m[i]); for (int i = 0; i < 3; i++) std::printf(" %.17g", lastToNew.t.v[i]);
  std::printf("\naff %.17g %.17g\nres", aff.a, aff.b);
  for (int i = 0; i < 5; i++) std::printf(" %.17g", tracker.lastResiduals[i]);
  std::printf("\nflow %.17g %.17g %.17g\n", tracker.lastFlowIndicators[0], tracker.lastFlowIndicators[1], tracker.lastFlowIndicators[2]);
  return 0;
}

template <class T>
static void dump(const std::string& dir, const char* name, const std::vector<T>& v);

// CoarseTracker::setCoarseTrackingRef(frameHessians, fh_right, Hcalib) and setCTRefForFirstFrame(frameHessians) with the reference's
// signatures on a small pointer graph: frames with pointHessians whose lastResiduals[0] / centerProjectedTo / efPoint->HdiF are what
// FullSystem::optimize left.  The template levels come back through sdso_track_get_ref and are dumped for tests/test_host_shim.py, which
// rebuilds them from the oracle (ImmaturePoint ctor, traceStereo there and back, the accept rule, orc_make_coarse_depth).
static int run_tracker_ref(const std::string& dir) {
  auto meta = load<int>(dir, "meta");             // levels w0 h0 n nframes
  auto kf = load<float>(dir, "K");                // fx fy cx cy baseline
  const int levels = meta[0], w0 = meta[1], h0 = meta[2], n = meta[3], nfr = meta[4];
  auto cpt = load<float>(dir, "cpt"), hdi = load<float>(dir, "HdiF"), pu = load<float>(dir, "pu"), pv = load<float>(dir, "pv"), pid = load<float>(dir, "pidepth");
  auto rstate = load<int>(dir, "rstate"), frame_of = load<int>(dir, "frame_of");
  auto has_last = load<uint8_t>(dir, "has_last");
  sdso_shim::Device dev(0);
  std::vector<std::unique_ptr<FrameHessian>> fhs;
  FrameHessian right;
  int w[SDSO_PYR_LEVELS], h[SDSO_PYR_LEVELS];
  for (int l = 0; l < levels; l++) { w[l] = w0 >> l; h[l] = h0 >> l; }
  for (int f = 0; f < nfr; f++) { fhs.emplace_back(new FrameHessian); fhs.back()->slot = 20 + f; fhs.back()->shellStore.id = 100 + f; fhs.back()->ab_exposure = 1.f + 0.01f * f; }
  FrameHessian& target = *fhs.back();
  target.aff.a = 0.02; target.aff.b = 1.5;
  load_pyramid(dir, "left", target, levels);
  dev.uploadFrame(target.slot, &target, levels, w, h);
  load_pyramid(dir, "right", right, 1);
  right.slot = 40;
  dev.uploadFrame(right.slot, &right, 1, w, h);
  std::vector<std::unique_ptr<PointHessian>> phs;
  std::vector<std::unique_ptr<EFPoint>> efps;
  std::vector<std::unique_ptr<PointFrameResidual>> pfrs;
  for (int i = 0; i < n; i++) {
    phs.emplace_back(new PointHessian);
    PointHessian& ph = *phs.back();
    ph.u = pu[i]; ph.v = pv[i]; ph.idepth = pid[i];
    efps.emplace_back(new EFPoint{&ph, {}, 0});
    ph.efPoint = efps.back().get();
    ph.efPoint->HdiF = hdi[i];
    pfrs.emplace_back(new PointFrameResidual);
    for (int k = 0; k < 3; k++) pfrs.back()->centerProjectedTo[k] = cpt[(size_t)i * 3 + k];
    ph.lastResiduals[0] = {has_last[i] ? pfrs.back().get() : nullptr, rstate[i]};
    fhs[frame_of[i]]->pointHessians.push_back(&ph);
  }
  CalibHessian HC;
  for (int i = 0; i < 4; i++) HC.value_scaled[i] = HC.value_zero[i] = kf[i];
  std::vector<FrameHessian*> frameHessians;
  for (auto& f : fhs) frameHessians.push_back(f.get());
  std::vector<int> o_pcn;
  for (int variant = 0; variant < 2; variant++) {
    sdso_shim::CoarseTracker<SE3, AffLight, Mat33, Vec3> tracker(dev, 5 + variant);
    tracker.makeK(&HC, levels, w0, h0);
    tracker.slot_of = [](const void* fh) { return static_cast<const FrameHessian*>(fh)->slot; };
    tracker.baseline = kf[4];
    if (variant == 0) tracker.setCoarseTrackingRef(frameHessians, &right, HC);
    else tracker.setCTRefForFirstFrame(frameHessians);
    if (tracker.refFrameID != target.shell->id || tracker.lastRef_aff_g2l.a != target.aff.a || tracker.firstCoarseRMSE != -1) { std::fprintf(stderr, "bookkeeping of setCoarseTrackingRef\n"); return 1; }
    for (int l = 0; l < levels; l++) {
      const int pn = tracker.pc_n[l];
      o_pcn.push_back(pn);
      std::vector<float> pc((size_t)4 * pn);
      int got = 0;
      dev.check(sdso_track_get_ref(dev.ctx(), 5 + variant, l, &got, pc.data(), pc.data() + pn, pc.data() + 2 * (size_t)pn, pc.data() + 3 * (size_t)pn), "sdso_track_get_ref");
      if (got != pn) { std::fprintf(stderr, "pc_n mismatch\n"); return 1; }
      char nm[32]; std::snprintf(nm, sizeof nm, "pc_%d_l%d", variant, l);
      dump(dir, nm, pc);
    }
  }
  dump(dir, "pcn", o_pcn);
  std::printf("tracker_ref ok\n");
  return 0;
}

static int run_stereo(const std::string& dir, bool fork_live) {
  auto meta = load<int>(dir, "meta");             // w, h, n, mode_right
  auto kf = load<float>(dir, "K");                // fx fy cx cy baseline
  FrameHessian fr; load_pyramid(dir, "right", fr, 1);
  sdso_shim::Device dev(0);
  int w[1] = {meta[0]}, h[1] = {meta[1]};
  dev.uploadFrame(3, &fr, 1, w, h);
  const int n = meta[2];
  auto us = load<float>(dir, "u_stereo"), vs = load<float>(dir, "v_stereo"), imin = load<float>(dir, "idepth_min"),
       imins = load<float>(dir, "idepth_min_stereo"), imaxs = load<float>(dir, "idepth_max_stereo"), col = load<float>(dir, "color"),
       wg = load<float>(dir, "weights"), gH = load<float>(dir, "gradH"), eth = load<float>(dir, "energyTH");
  std::vector<ImmaturePoint> store(n);
  std::vector<ImmaturePoint*> pts(n);
  for (int i = 0; i < n; i++) {
    ImmaturePoint& p = store[i];
    p.u_stereo = us[i]; p.v_stereo = vs[i]; p.idepth_min = imin[i]; p.idepth_min_stereo = imins[i]; p.idepth_max_stereo = imaxs[i];
    p.idepth_stereo = 0; p.energyTH = eth[i]; p.quality = 10000; p.lastTraceStatus = 5; p.lastTraceUV[0] = p.lastTraceUV[1] = 0; p.lastTracePixelInterval = 0;
    for (int k = 0; k < 8; k++) { p.color[k] = col[i * 8 + k]; p.weights[k] = wg[i * 8 + k]; }
    for (int k = 0; k < 4; k++) p.gradH.m[k] = gH[i * 4 + k];
    pts[i] = &p;
  }
  Mat33f K{{kf[0], 0, kf[2], 0, kf[1], kf[3], 0, 0, 1}};
  std::vector<uint8_t> status;
  dev.setForkLiveTraceRefinement(fork_live);
  std::vector<ImmaturePoint> fresh(store.begin(), store.begin() + std::min(n, 24));     // (before the call: the member form below starts from the same state)
  sdso_shim::traceStereoAll(dev, pts, 3, K, kf[4], meta[3] != 0, status);
  // ImmaturePoint::traceStereo as the member it is — one point per call — must be the batch's result for that point, bit for bit
  for (size_t i = 0; i < fresh.size(); i++) {
    const int st = sdso_shim::traceStereo(dev, &fresh[i], 3, K, kf[4], meta[3] != 0);
    const ImmaturePoint &a = fresh[i], &b = store[i];
    if (st != (int)status[i] || a.lastTraceStatus != b.lastTraceStatus || std::memcmp(&a.idepth_min_stereo, &b.idepth_min_stereo, 4) || std::memcmp(&a.idepth_max_stereo, &b.idepth_max_stereo, 4) ||
        std::memcmp(&a.idepth_stereo, &b.idepth_stereo, 4) || std::memcmp(&a.quality, &b.quality, 4) || std::memcmp(a.lastTraceUV, b.lastTraceUV, 8) ||
        std::memcmp(&a.lastTracePixelInterval, &b.lastTracePixelInterval, 4)) {
      std::fprintf(stderr, "traceStereo (one point) differs from traceStereoAll at point %zu\n", i);
      return 3;
    }
  }
  for (int i = 0; i < n; i++)
    std::printf("%d %d %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", (int)status[i], store[i].lastTraceStatus, store[i].idepth_min_stereo, store[i].idepth_max_stereo,
                store[i].idepth_stereo, store[i].quality, store[i].lastTraceUV[0], store[i].lastTraceUV[1], store[i].lastTracePixelInterval);
  return 0;
}

template <class T>
static void dump(const std::string& dir, const char* name, const std::vector<T>& v) {
  std::ofstream f(dir + "/out_" + name + ".bin", std::ios::binary);
  f.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
}

// FullSystem::optimize through the shim on a pointer graph built like the reference's (FrameHessian / PointHessian / PointFrameResidual /
// EFFrame / EFPoint / EFResidual with residuals, residualsAll, lastResiduals), then everything the reference's callers read afterwards,
// dumped as raw arrays (out_*.bin) for tests/test_host_shim.py to compare with the ORACLE's post-state:
//   makeCoarseDepthL0 STEP1 (CoarseTracker.cpp:295-350): which points enter, their pixel and weight sqrtf(1e-3 / (HdiF + 1e-12))
//   flagPointsForRemoval (FullSystem.cpp:1004-1035): marginalise / drop / keep decision of a point whose host is being marginalised
// the pointer graph of one window, built like the reference's (FrameHessian / PointHessian / PointFrameResidual / EFFrame / EFPoint /
// EFResidual with residuals, residualsAll, lastResiduals) from the raw arrays tests/test_host_shim.py wrote
struct BaGraph {
  std::vector<int> meta;
  int nf = 0, np = 0, nr = 0, w = 0, h = 0;
  std::vector<std::unique_ptr<FrameHessian>> fhs;
  std::vector<std::unique_ptr<EFFrame>> effs;
  std::vector<std::unique_ptr<PointHessian>> phs;
  std::vector<std::unique_ptr<EFPoint>> efps;
  std::vector<PointFrameResidual*> pfrs;                        // window order
  std::vector<int> host;
  EnergyFunctional ef;
  CalibHessian HC;
  void build(const std::string& dir, sdso_shim::Device& dev) {
    meta = load<int>(dir, "meta");             // nf np nr w h its solverMode
    nf = meta[0]; np = meta[1]; nr = meta[2]; w = meta[3]; h = meta[4];
    auto calib = load<double>(dir, "calib");        // value_scaled(4) value_zero(4)
    auto evalPT = load<double>(dir, "evalPT"), state = load<double>(dir, "state"), state_zero = load<double>(dir, "state_zero"),
         HM = load<double>(dir, "HM"), bM = load<double>(dir, "bM");
    auto exposure = load<float>(dir, "ab_exposure"), eTH = load<float>(dir, "frameEnergyTH");
    auto frameID = load<int>(dir, "frameID"), res_point = load<int>(dir, "res_point"), res_target = load<int>(dir, "res_target");
    host = load<int>(dir, "host");
    auto u = load<float>(dir, "u"), v = load<float>(dir, "v"), idepth = load<float>(dir, "idepth"), idz = load<float>(dir, "idepth_zero"),
         color = load<float>(dir, "color"), weights = load<float>(dir, "weights"), mrb = load<float>(dir, "maxRelBaseline");
    auto prior = load<uint8_t>(dir, "hasDepthPrior"), res_state = load<uint8_t>(dir, "res_state"), isnew = load<uint8_t>(dir, "res_isNew");
    auto ngood = load<int>(dir, "numGoodResiduals");
    int wv[1] = {w}, hv[1] = {h};
    for (int f = 0; f < nf; f++) {
      fhs.emplace_back(new FrameHessian);
      FrameHessian& fh = *fhs.back();
      char nm[32]; std::snprintf(nm, sizeof nm, "img%d", f);
      load_pyramid(dir, nm, fh, 1);
      for (int i = 0; i < 9; i++) fh.worldToCam_evalPT.R.m[i] = evalPT[f * 12 + i];
      for (int i = 0; i < 3; i++) fh.worldToCam_evalPT.t.v[i] = evalPT[f * 12 + 9 + i];
      for (int i = 0; i < 10; i++) { fh.state[i] = state[f * 10 + i]; fh.state_zero[i] = state_zero[f * 10 + i]; }
      fh.ab_exposure = exposure[f]; fh.frameEnergyTH = eTH[f]; fh.frameID = frameID[f]; fh.idx = f; fh.slot = 10 + f;
      dev.uploadFrame(fh.slot, &fh, 1, wv, hv);
      effs.emplace_back(new EFFrame{&fh, {}, f});
      ef.frames.push_back(effs.back().get());
    }
    int r = 0;
    for (int p = 0; p < np; p++) {
      phs.emplace_back(new PointHessian);
      PointHessian& ph = *phs.back();
      ph.u = u[p]; ph.v = v[p]; ph.idepth = idepth[p]; ph.idepth_zero = idz[p]; ph.hasDepthPrior = prior[p] != 0;
      ph.maxRelBaseline = mrb[p]; ph.numGoodResiduals = ngood[p];
      for (int k = 0; k < 8; k++) { ph.color[k] = color[p * 8 + k]; ph.weights[k] = weights[p * 8 + k]; }
      efps.emplace_back(new EFPoint{&ph, {}, 0});
      ph.efPoint = efps.back().get();
      for (; r < nr && res_point[r] == p; r++) {
        // raw new: the shim deletes dropped residuals like the reference does (deleteOut / dropResidual)
        PointFrameResidual* pfr = new PointFrameResidual;
        pfr->state_state = (int)res_state[r]; pfr->isNew = isnew[r] != 0; pfr->point = &ph; pfr->id = r;
        EFResidual* efr = new EFResidual{pfr, ef.frames[res_target[r]]};
        efr->point = ph.efPoint; efr->idxInAll = (int)ph.efPoint->residualsAll.size();
        pfr->efResidual = efr;
        ph.efPoint->residualsAll.push_back(efr);
        ph.residuals.push_back(pfr);
        pfrs.push_back(pfr);
        ef.nResiduals++;
        // lastResiduals: [0] the residual into the newest frame, [1] into the one before (FullSystem.cpp:1400-1410)
        if (res_target[r] == nf - 1) ph.lastResiduals[0] = {pfr, 0};
        if (res_target[r] == nf - 2) ph.lastResiduals[1] = {pfr, 0};
      }
      ef.frames[host[p]]->points.push_back(ph.efPoint);   // points arrive grouped by host (makeIDX order)
      ef.allPoints.push_back(ph.efPoint);
    }
    const int n = 8 * nf + 4;
    ef.HM.n = n; ef.HM.d = HM; ef.bM = bM;
    for (int i = 0; i < 4; i++) { HC.value_scaled[i] = calib[i]; HC.value_zero[i] = calib[4 + i]; }
  }
  ~BaGraph() { for (auto& ph : phs) for (PointFrameResidual* rr : ph->residuals) { delete rr->efResidual; delete rr; } }
};

static int run_ba(const std::string& dir) {
  sdso_shim::Device dev(0);
  BaGraph G;
  G.build(dir, dev);
  const int nf = G.nf, np = G.np, nr = G.nr, w = G.w, h = G.h;
  auto& meta = G.meta; auto& fhs = G.fhs; auto& phs = G.phs; EnergyFunctional& ef = G.ef; CalibHessian& HC = G.HC;
  sdso_shim::WindowedBA<EnergyFunctional, CalibHessian> ba(dev, 0);
  ba.upload(&ef, &HC, w, h, /*solverMode=*/meta[6], 1e12, 1e8, true, [](FrameHessian* fh) { return fh->slot; });
  const float rmse = ba.optimize(meta[5], &ef, &HC);
  std::printf("rmse %.9g iterations %d resInA %d energy %.17g removed %d nResiduals %d\n", rmse, ba.lastResult.iterations, ef.resInA, ba.lastResult.lastEnergy,
              ba.lastRemoved, ef.nResiduals);
  // ---- dump what the reference's callers read
  std::vector<double> o_state(nf * 10), o_zero(nf * 10), o_eval(nf * 12), o_fstep(nf * 10), o_calib(12);
  std::vector<float> o_eth(nf);
  for (int f = 0; f < nf; f++) {
    for (int i = 0; i < 10; i++) { o_state[f * 10 + i] = fhs[f]->state[i]; o_zero[f * 10 + i] = fhs[f]->state_zero[i]; o_fstep[f * 10 + i] = fhs[f]->step[i]; }
    for (int i = 0; i < 9; i++) o_eval[f * 12 + i] = fhs[f]->worldToCam_evalPT.R.m[i];
    for (int i = 0; i < 3; i++) o_eval[f * 12 + 9 + i] = fhs[f]->worldToCam_evalPT.t.v[i];
    o_eth[f] = fhs[f]->frameEnergyTH;
  }
  for (int i = 0; i < 4; i++) { o_calib[i] = HC.value[i]; o_calib[4 + i] = HC.value_scaled[i]; o_calib[8 + i] = HC.step[i]; }
  std::vector<float> o_pt((size_t)np * 8);        // idepth idepth_zero step idepth_hessian maxRelBaseline HdiF bdSumF weight
  std::vector<int> o_pi((size_t)np * 8);          // numGood nResiduals last0_alive last0_state last1_alive last1_state coarse_uv(packed, -1 = not used) flag_decision
  std::vector<int> o_lists;                       // per point: residuals ids ..., -1, residualsAll ids ..., -2
  std::vector<int> o_alive(nr, 0), o_rstate(nr, -1), o_ract(nr, -1);
  std::vector<float> o_renergy(nr, 0.f), o_cpt((size_t)nr * 3, 0.f), o_prj((size_t)nr * 16, 0.f);
  for (int p = 0; p < np; p++) {
    PointHessian& ph = *phs[p];
    float* f = &o_pt[(size_t)p * 8]; int* q = &o_pi[(size_t)p * 8];
    f[0] = ph.idepth; f[1] = ph.idepth_zero; f[2] = ph.step; f[3] = ph.idepth_hessian; f[4] = ph.maxRelBaseline; f[5] = ph.efPoint->HdiF; f[6] = ph.efPoint->bdSumF;
    q[0] = ph.numGoodResiduals; q[1] = (int)ph.residuals.size();
    q[2] = ph.lastResiduals[0].first != nullptr; q[3] = ph.lastResiduals[0].second;
    q[4] = ph.lastResiduals[1].first != nullptr; q[5] = ph.lastResiduals[1].second;
    // makeCoarseDepthL0 STEP1 (CoarseTracker.cpp:295-350)
    q[6] = -1; f[7] = 0.f;
    if (ph.lastResiduals[0].first != nullptr && ph.lastResiduals[0].second == 0 /* ResState::IN */) {
      PointFrameResidual* rr = ph.lastResiduals[0].first;
      const int uu = (int)(rr->centerProjectedTo[0] + 0.5f), vv = (int)(rr->centerProjectedTo[1] + 0.5f);
      q[6] = vv * 65536 + uu;
      f[7] = sqrtf(1e-3 / (ph.efPoint->HdiF + 1e-12));
    }
    // flagPointsForRemoval for a point whose host is flagged (FullSystem.cpp:1004-1035): 0 drop (no residuals / idepth < 0), 1 marginalise, 2 drop (inlier, small Hessian), 3 drop (not an inlier)
    if (ph.idepth < 0 || ph.residuals.empty()) q[7] = 0;
    else if (ph.isInlierNew()) q[7] = ph.idepth_hessian > 50.f /* setting_minIdepthH_marg */ ? 1 : 2;
    else q[7] = 3;
    for (PointFrameResidual* rr : ph.residuals) {
      o_lists.push_back(rr->id);
      o_alive[rr->id] = 1; o_rstate[rr->id] = rr->state_state; o_ract[rr->id] = rr->efResidual->isActiveAndIsGoodNEW ? 1 : 0;
      o_renergy[rr->id] = (float)rr->state_energy;
      for (int k = 0; k < 3; k++) o_cpt[(size_t)rr->id * 3 + k] = rr->centerProjectedTo[k];
      for (int k = 0; k < 8; k++) { o_prj[(size_t)rr->id * 16 + 2 * k] = rr->projectedTo[k][0]; o_prj[(size_t)rr->id * 16 + 2 * k + 1] = rr->projectedTo[k][1]; }
    }
    o_lists.push_back(-1);
    for (EFResidual* er : ph.efPoint->residualsAll) o_lists.push_back(er->data->id);
    o_lists.push_back(-2);
  }
  std::vector<double> o_lastX = ef.lastX, o_lastbS = ef.lastbS, o_lastHS = ef.lastHS.d;
  std::vector<int> o_counts = {ef.resInA, ef.resInL, ef.resInM, ef.nResiduals, ba.lastRemoved, ba.lastResult.iterations};
  dump(dir, "state", o_state); dump(dir, "state_zero", o_zero); dump(dir, "evalPT", o_eval); dump(dir, "frame_step", o_fstep); dump(dir, "calib", o_calib);
  dump(dir, "frameEnergyTH", o_eth); dump(dir, "pt", o_pt); dump(dir, "pi", o_pi); dump(dir, "lists", o_lists); dump(dir, "alive", o_alive);
  dump(dir, "rstate", o_rstate); dump(dir, "ract", o_ract); dump(dir, "renergy", o_renergy); dump(dir, "cpt", o_cpt); dump(dir, "prj", o_prj);
  dump(dir, "lastX", o_lastX); dump(dir, "lastbS", o_lastbS); dump(dir, "lastHS", o_lastHS); dump(dir, "counts", o_counts);
  // EnergyFunctional::accumulate{AF,LF,SCF}_MT at the state optimize() left (EnergyFunctional.cpp:212-269): the three stitched systems
  ba.linearizeAll(); ba.applyRes(); ba.accumulateAll();
  DynMat H3[3]; DynVec b3[3];
  ba.accumulateAF_MT(H3[0], b3[0], false); ba.accumulateLF_MT(H3[1], b3[1], false); ba.accumulateSCF_MT(H3[2], b3[2], false);
  std::vector<double> o_st;
  for (int k = 0; k < 3; k++) { o_st.insert(o_st.end(), H3[k].d.begin(), H3[k].d.end()); o_st.insert(o_st.end(), b3[k].d.begin(), b3[k].d.end()); }
  dump(dir, "stitched", o_st);
  return 0;
}

// The members the round-4 verdict found missing from the shim, driven the way the reference's own bodies drive them:
//   linearizeAll_Reductor / applyRes_Reductor: PointFrameResidual::linearize(&HCalib) and applyRes(true) per object (FullSystemOptimize.cpp:52-96)
//   setAdjointsF / setDeltaF (EnergyFunctional.cpp:41-119, :173-207)
//   accumulateAF_MT / accumulateLF_MT / accumulateSCF_MT through AccumulatedTopHessianSSE / AccumulatedSCHessianSSE (:212-269)
//   solveSystemF(iteration, lambda, &HCalib) writing into the objects (:838-995, :272-341); calcLEnergyF_MT / calcMEnergyF (:344-442)
//   marginalizePointsF's accumulator calls (:663-736) for the points of the oldest keyframe
static int run_ba_members(const std::string& dir) {
  sdso_shim::Device dev(0);
  BaGraph G;
  G.build(dir, dev);
  const int nf = G.nf, np = G.np, nr = G.nr, n = 8 * nf + 4;
  EnergyFunctional& ef = G.ef; CalibHessian& HC = G.HC;
  using BA = sdso_shim::WindowedBA<EnergyFunctional, CalibHessian>;
  BA ba(dev, 0);
  ba.upload(&ef, &HC, G.w, G.h, /*solverMode=*/G.meta[6], 1e12, 1e8, true, [](FrameHessian* fh) { return fh->slot; });
  // ---- linearizeAll(false) + applyRes, object by object
  double E = 0;
  for (PointFrameResidual* r : G.pfrs) E += ba.linearize(r, &HC);
  for (PointFrameResidual* r : G.pfrs) ba.applyRes(r, true);
  std::vector<int> o_ns(nr), o_st(nr), o_act(nr);
  std::vector<double> o_ne(nr), o_nw(nr), o_en(nr);
  for (int i = 0; i < nr; i++) {
    PointFrameResidual* r = G.pfrs[i];
    o_ns[i] = r->state_NewState; o_st[i] = r->state_state; o_act[i] = r->efResidual->isActiveAndIsGoodNEW ? 1 : 0;
    o_ne[i] = r->state_NewEnergy; o_nw[i] = r->state_NewEnergyWithOutlier; o_en[i] = r->state_energy;
  }
  {  // the two object-level branches of Residuals.cpp:367-385 that do not go to the device: applyRes(false) moves state_state / state_energy
     // only (no OOB return, isActiveAndIsGoodNEW untouched); a linearised residual is left alone by linearize and applyRes alike
    PointFrameResidual probe = *G.pfrs[0];
    EFResidual ep = *G.pfrs[0]->efResidual;
    probe.efResidual = &ep;
    probe.state_state = 1 /* OOB */; probe.state_NewState = 2; probe.state_NewEnergy = 42.0; probe.state_energy = 1.0; ep.isActiveAndIsGoodNEW = true;
    ba.applyRes(&probe, false);
    if (probe.state_state != 2 || probe.state_energy != 42.0 || !ep.isActiveAndIsGoodNEW) { std::fprintf(stderr, "applyRes(r, false) semantics\n"); return 1; }
    ep.isLinearized = true;
    probe.state_state = 0; probe.state_NewState = 2; probe.state_energy = 3.0; probe.state_NewEnergy = 5.0; probe.state_NewEnergyWithOutlier = 7.0;
    const double e0 = ba.linearize(&probe, &HC);
    ba.applyRes(&probe, true);
    if (e0 != 0.0 || probe.state_state != 0 || probe.state_NewState != 2 || probe.state_energy != 3.0 || probe.state_NewEnergy != 5.0 || probe.state_NewEnergyWithOutlier != 7.0) {
      std::fprintf(stderr, "a linearised residual was touched\n"); return 1; }
  }
  dump(dir, "m_newState", o_ns); dump(dir, "m_state", o_st); dump(dir, "m_act", o_act); dump(dir, "m_newEnergy", o_ne); dump(dir, "m_newEnergyWO", o_nw); dump(dir, "m_energy", o_en);
  // ---- setAdjointsF / setDeltaF
  ba.setAdjointsF(&HC); ba.setDeltaF(&HC);
  std::vector<double> o_ad((size_t)nf * nf * 128), o_fd(nf * 16);
  std::vector<float> o_adf((size_t)nf * nf * 128), o_htd((size_t)nf * nf * 8), o_pd(np), o_cd(ef.cDeltaF, ef.cDeltaF + 4);
  for (int k = 0; k < nf * nf; k++) {
    for (int e = 0; e < 64; e++) { o_ad[(size_t)k * 128 + e] = ef.adHost[k].m[e]; o_ad[(size_t)k * 128 + 64 + e] = ef.adTarget[k].m[e];
                                   o_adf[(size_t)k * 128 + e] = ef.adHostF[k].m[e]; o_adf[(size_t)k * 128 + 64 + e] = ef.adTargetF[k].m[e]; }
    for (int j = 0; j < 8; j++) o_htd[(size_t)k * 8 + j] = ef.adHTdeltaF[k].m[j];
  }
  for (int f = 0; f < nf; f++) for (int i = 0; i < 8; i++) { o_fd[f * 16 + i] = ef.frames[f]->delta[i]; o_fd[f * 16 + 8 + i] = ef.frames[f]->delta_prior[i]; }
  for (int p = 0; p < np; p++) o_pd[p] = G.efps[p]->deltaF;
  dump(dir, "m_adjoints", o_ad); dump(dir, "m_adjointsF", o_adf); dump(dir, "m_adHTdeltaF", o_htd); dump(dir, "m_frame_delta", o_fd); dump(dir, "m_point_delta", o_pd); dump(dir, "m_cDeltaF", o_cd);
  // ---- the three accumulations through the accumulator classes (bodies of accumulateAF_MT / LF_MT / SCF_MT)
  sdso_shim::AccumulatedTopHessianSSE<BA> accSSE_top_A(ba), accSSE_top_L(ba);
  sdso_shim::AccumulatedSCHessianSSE<BA> accSSE_bot(ba);
  DynMat H3[3]; DynVec b3[3];
  const int nAll = (int)ef.allPoints.size();
  accSSE_top_A.setZero(nf);
  for (EFFrame* f : ef.frames) for (EFPoint* p : f->points) accSSE_top_A.addPoint<0>(p, &ef);            // the MT == false branch (:225-230)
  accSSE_top_A.stitchDoubleMT((void*)nullptr, H3[0], b3[0], &ef, false, false);
  ef.resInA = accSSE_top_A.nres[0];
  accSSE_top_L.setZero(nf);
  accSSE_top_L.addPointsInternal<1>(&ef.allPoints, &ef, 0, nAll);                                            // the MT == true branch (:238-241)
  accSSE_top_L.stitchDoubleMT((void*)nullptr, H3[1], b3[1], &ef, true, true);
  ef.resInL = accSSE_top_L.nres[0];
  accSSE_bot.setZero(nf);
  accSSE_bot.addPointsInternal(&ef.allPoints, true, 0, nAll);
  accSSE_bot.stitchDoubleMT((void*)nullptr, H3[2], b3[2], &ef, true);
  std::vector<double> o_st3;
  for (int k = 0; k < 3; k++) { o_st3.insert(o_st3.end(), H3[k].d.begin(), H3[k].d.end()); o_st3.insert(o_st3.end(), b3[k].d.begin(), b3[k].d.end()); }
  dump(dir, "m_stitched", o_st3);
  // ---- solveSystemF(iteration, lambda, &HCalib) and the energies
  ba.solveSystemF(0, 1e-1, &HC);
  std::vector<double> o_x = ef.lastX, o_bS = ef.lastbS, o_HS = ef.lastHS.d, o_fs(nf * 10), o_cs(HC.step.v, HC.step.v + 4);
  std::vector<float> o_ps(np * 3);
  for (int f = 0; f < nf; f++) for (int i = 0; i < 10; i++) o_fs[f * 10 + i] = G.fhs[f]->step[i];
  for (int p = 0; p < np; p++) { o_ps[p * 3] = G.phs[p]->step; o_ps[p * 3 + 1] = G.efps[p]->HdiF; o_ps[p * 3 + 2] = G.efps[p]->bdSumF; }
  dump(dir, "m_lastX", o_x); dump(dir, "m_lastbS", o_bS); dump(dir, "m_lastHS", o_HS); dump(dir, "m_frame_step", o_fs); dump(dir, "m_calib_step", o_cs); dump(dir, "m_point_step", o_ps);
  std::vector<double> o_e = {E, ba.calcLEnergyF_MT(), ba.calcMEnergyF(), (double)ef.resInA, (double)ef.resInL};
  // ---- marginalizePointsF's accumulator calls for the points hosted in the oldest keyframe (EnergyFunctional.cpp:680-717)
  DynMat M, Msc; DynVec Mb, Mbsc;
  accSSE_bot.setZero(nf); accSSE_top_A.setZero(nf);
  int nflag = 0;
  for (int p = 0; p < np; p++)
    if (G.host[p] == 0) { accSSE_top_A.addPoint<2>(G.efps[p].get(), &ef); accSSE_bot.addPoint(G.efps[p].get(), false); nflag++; }
  accSSE_top_A.stitchDouble(M, Mb, &ef, false, false);
  accSSE_bot.stitchDouble(Msc, Mbsc, &ef);
  ef.resInM += accSSE_top_A.nres[0];
  std::vector<double> o_marg;
  o_marg.insert(o_marg.end(), M.d.begin(), M.d.end()); o_marg.insert(o_marg.end(), Mb.d.begin(), Mb.d.end());
  o_marg.insert(o_marg.end(), Msc.d.begin(), Msc.d.end()); o_marg.insert(o_marg.end(), Mbsc.d.begin(), Mbsc.d.end());
  o_marg.insert(o_marg.end(), ba.HM().begin(), ba.HM().end()); o_marg.insert(o_marg.end(), ba.bM().begin(), ba.bM().end());
  dump(dir, "m_marg", o_marg);
  o_e.push_back(ba.calcLEnergyF_MT()); o_e.push_back(ba.calcMEnergyF()); o_e.push_back((double)ef.resInM); o_e.push_back((double)nflag);
  dump(dir, "m_energies", o_e);
  std::printf("members ok: E %.9g resInA %d resInL %d resInM %d flagged %d n %d\n", E, ef.resInA, ef.resInL, ef.resInM, nflag, n);
  return 0;
}

static int run_selector(const std::string& dir) {
  auto meta = load<int>(dir, "meta");             // w, h, levels, potential, recursions
  auto par = load<float>(dir, "par");             // density, thFactor
  FrameHessian fr; load_pyramid(dir, "img", fr, meta[2]);
  sdso_shim::Device dev(0);
  int w[SDSO_PYR_LEVELS], h[SDSO_PYR_LEVELS];
  for (int l = 0; l < meta[2]; l++) { w[l] = meta[0] >> l; h[l] = meta[1] >> l; }
  dev.uploadFrame(7, &fr, meta[2], w, h);
  sdso_shim::PixelSelector sel(dev);
  sel.currentPotential = meta[3];
  std::vector<float> map((size_t)meta[0] * meta[1]);
  const int n = sel.makeMaps(7, map.data(), par[0], meta[4], false, par[1]);
  unsigned long long sum = 0; long cnt[5] = {0, 0, 0, 0, 0};
  for (size_t i = 0; i < map.size(); i++) { const int v = (int)map[i]; cnt[v]++; sum = sum * 1000003ull + (unsigned long long)(v * 7 + 1) * (i + 1); }
  std::printf("n %d potential %d c1 %ld c2 %ld c4 %ld hash %llu\n", n, sel.currentPotential, cnt[1], cnt[2], cnt[4], sum);
  // marginalizeFrame through the shim on a small SPD prior
  std::vector<double> HM(20 * 20, 0.0), bM(20);
  for (int i = 0; i < 20; i++) { bM[i] = 0.1 * (i + 1); for (int j = 0; j < 20; j++) HM[i * 20 + j] = (i == j ? 50.0 + i : 1.0 / (1 + i + j)); }
  const double prior[8] = {1e3, 0, 1e3, 0, 1e2, 0, 1e6, 1e6}, dprior[8] = {1e-3, 0, -2e-3, 0, 1e-3, 0, 1e-4, -1e-4};
  sdso_shim::marginalizeFrame(2, 0, prior, dprior, HM, bM);
  std::printf("marg %zu %zu %.17g %.17g\n", HM.size(), bM.size(), HM[0], bM[11]);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: test_shim <dir> tracker|stereo|ba\n"); return 2; }
  try {
    const std::string what = argv[2];
    if (what == "tracker") return run_tracker(argv[1], false);
    if (what == "tracker_g2o") return run_tracker(argv[1], true);
    if (what == "stereo") return run_stereo(argv[1], false);
    if (what == "stereo_g2o") return run_stereo(argv[1], true);
    if (what == "ba") return run_ba(argv[1]);
    if (what == "ba_members") return run_ba_members(argv[1]);
    if (what == "tracker_ref") return run_tracker_ref(argv[1]);
    if (what == "selector") return run_selector(argv[1]);
  } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
  return 2;
}
