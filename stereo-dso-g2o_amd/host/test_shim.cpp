// Exercises host/sdso_shim.h the way the reference's classes would call it, with small stand-ins
// for the Eigen / Sophus / DSO types (none of which exist in this image).  tests/test_host_shim.py
// writes a problem as raw arrays into a directory, runs this program on the GPU box and compares
// what it prints with the same problem pushed through the C-ABI from Python.
//
//   test_shim <dir> tracker|tracker_g2o|stereo|stereo_g2o|ba|selector
#include <cstdio>
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
struct CalibHessian {
  double value_scaled[4], value_zero[4];
  float fxl() const { return (float)value_scaled[0]; } float fyl() const { return (float)value_scaled[1]; }
  float cxl() const { return (float)value_scaled[2]; } float cyl() const { return (float)value_scaled[3]; }
};
struct FrameHessian {
  Vec3f* dIp[SDSO_PYR_LEVELS];
  std::vector<std::vector<float>> store;
  SE3 worldToCam_evalPT; double state[10], state_zero[10];
  float ab_exposure = 1, frameEnergyTH = 0; int frameID = 0, idx = 0, slot = 0;
  const SE3& get_worldToCam_evalPT() const { return worldToCam_evalPT; }
  const double* get_state() const { return state; }
  const double* get_state_zero() const { return state_zero; }
};
struct EFFrame; struct EFPoint; struct EFResidual;
struct PointFrameResidual { int state_state = 0; };
struct PointHessian { float u, v, idepth, idepth_zero, color[8], weights[8]; bool hasDepthPrior = false; };
struct EFResidual { PointFrameResidual* data; EFFrame* target; };
struct EFPoint { PointHessian* data; std::vector<EFResidual*> residualsAll; int stateFlag = 0; };
struct EFFrame { FrameHessian* data; std::vector<EFPoint*> points; int idx; };
struct DynMat {
  int n = 0; std::vector<double> d;
  double& operator()(int i, int j) { return d[(size_t)i * n + j]; }
};
struct EnergyFunctional { std::vector<EFFrame*> frames; DynMat HM; std::vector<double> bM; };
struct ImmaturePoint {
  float u, v, idepth_max;
  float u_stereo, v_stereo, idepth_min, idepth_min_stereo, idepth_max_stereo, idepth_stereo, energyTH, quality, color[8], weights[8];
  Mat22f gradH; int lastTraceStatus; float lastTraceUV[2]; float lastTracePixelInterval;
};
struct Vec5 { double v[5]; double operator[](int i) const { return v[i]; } };
// compile the temporal-trace wrapper against the stand-in as well (it is exercised through the C-ABI in tests/test_stereo.py)
template void sdso_shim::traceOnAll<ImmaturePoint>(sdso_shim::Device&, std::vector<ImmaturePoint*>&, const std::vector<int>&, const std::vector<sdso_trace_geom_t>&, int,
                                                   std::vector<uint8_t>&);

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
  sdso_shim::traceStereoAll(dev, pts, 3, K, kf[4], meta[3] != 0, status);
  for (int i = 0; i < n; i++)
    std::printf("%d %d %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", (int)status[i], store[i].lastTraceStatus, store[i].idepth_min_stereo, store[i].idepth_max_stereo,
                store[i].idepth_stereo, store[i].quality, store[i].lastTraceUV[0], store[i].lastTraceUV[1], store[i].lastTracePixelInterval);
  return 0;
}

static int run_ba(const std::string& dir) {
  auto meta = load<int>(dir, "meta");             // nf np nr w h its
  const int nf = meta[0], np = meta[1], nr = meta[2], w = meta[3], h = meta[4];
  auto calib = load<double>(dir, "calib");        // value_scaled(4) value_zero(4)
  auto evalPT = load<double>(dir, "evalPT"), state = load<double>(dir, "state"), state_zero = load<double>(dir, "state_zero"),
       HM = load<double>(dir, "HM"), bM = load<double>(dir, "bM");
  auto exposure = load<float>(dir, "ab_exposure"), eTH = load<float>(dir, "frameEnergyTH");
  auto frameID = load<int>(dir, "frameID"), host = load<int>(dir, "host"), res_point = load<int>(dir, "res_point"), res_target = load<int>(dir, "res_target");
  auto u = load<float>(dir, "u"), v = load<float>(dir, "v"), idepth = load<float>(dir, "idepth"), idz = load<float>(dir, "idepth_zero"),
       color = load<float>(dir, "color"), weights = load<float>(dir, "weights");
  auto prior = load<uint8_t>(dir, "hasDepthPrior"), res_state = load<uint8_t>(dir, "res_state");
  sdso_shim::Device dev(0);
  std::vector<std::unique_ptr<FrameHessian>> fhs;
  std::vector<std::unique_ptr<EFFrame>> effs;
  EnergyFunctional ef;
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
  std::vector<std::unique_ptr<PointHessian>> phs;
  std::vector<std::unique_ptr<EFPoint>> efps;
  std::vector<std::unique_ptr<PointFrameResidual>> pfrs;
  std::vector<std::unique_ptr<EFResidual>> efrs;
  int r = 0;
  for (int p = 0; p < np; p++) {
    phs.emplace_back(new PointHessian);
    PointHessian& ph = *phs.back();
    ph.u = u[p]; ph.v = v[p]; ph.idepth = idepth[p]; ph.idepth_zero = idz[p]; ph.hasDepthPrior = prior[p] != 0;
    for (int k = 0; k < 8; k++) { ph.color[k] = color[p * 8 + k]; ph.weights[k] = weights[p * 8 + k]; }
    efps.emplace_back(new EFPoint{&ph, {}, 0});
    for (; r < nr && res_point[r] == p; r++) {
      pfrs.emplace_back(new PointFrameResidual{(int)res_state[r]});
      efrs.emplace_back(new EFResidual{pfrs.back().get(), ef.frames[res_target[r]]});
      efps.back()->residualsAll.push_back(efrs.back().get());
    }
    ef.frames[host[p]]->points.push_back(efps.back().get());   // points arrive grouped by host (makeIDX order)
  }
  const int n = 8 * nf + 4;
  ef.HM.n = n; ef.HM.d = HM; ef.bM = bM;
  CalibHessian HC; for (int i = 0; i < 4; i++) { HC.value_scaled[i] = calib[i]; HC.value_zero[i] = calib[4 + i]; }
  sdso_shim::WindowedBA<EnergyFunctional, CalibHessian> ba(dev, 0);
  ba.upload(&ef, &HC, w, h, /*solverMode=*/meta[6], 1e12, 1e8, true, [](FrameHessian* fh) { return fh->slot; });
  const float rmse = ba.optimize(meta[5], [&](int f, const double* st) { for (int i = 0; i < 10; i++) fhs[f]->state[i] = st[i]; },
                                 [](EFPoint* p, float idp) { p->data->idepth = idp; }, [](EFResidual* rr, uint8_t s) { rr->data->state_state = s; });
  std::printf("rmse %.9g iterations %d resInA %d energy %.17g\n", rmse, ba.lastResult.iterations, ba.lastResult.resInA, ba.lastResult.lastEnergy);
  for (int f = 0; f < nf; f++) { std::printf("state"); for (int i = 0; i < 10; i++) std::printf(" %.17g", fhs[f]->state[i]); std::printf("\n"); }
  std::printf("idepth"); for (int p = 0; p < np; p++) std::printf(" %.9g", phs[p]->idepth); std::printf("\n");
  std::printf("rstate"); for (auto& q : pfrs) std::printf(" %d", q->state_state); std::printf("\n");
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
    if (what == "selector") return run_selector(argv[1]);
  } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
  return 2;
}
