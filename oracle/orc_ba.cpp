// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_api.h).
// Windowed bundle adjustment, DSO-native arithmetic.  Follows (paths under /root/reference):
//   src/FullSystem/Residuals.cpp:83-385                 PointFrameResidual::linearize / applyRes
//   src/FullSystem/ResidualProjections.h:45-96          projectPoint (both overloads)
//   src/FullSystem/HessianBlocks.{h,cpp}                setState, setStateZero, FrameFramePrecalc::set, getPrior
//   src/OptimizationBackend/EnergyFunctionalStructs.cpp:37-123   takeDataF, takeData, fixLinearizationF
//   src/OptimizationBackend/AccumulatedTopHessian.{h,cpp}        addPoint<mode>, stitchDouble(Internal/MT)
//   src/OptimizationBackend/AccumulatedSCHessian.{h,cpp}         addPoint, stitchDouble(Internal/MT)
//   src/OptimizationBackend/EnergyFunctional.cpp:41-119, 173-207, 212-442, 663-736, 775-995, 1021-1032
//   src/FullSystem/FullSystemOptimize.cpp:52-370, 871-1041, 1087-1147   GN driver, nullspaces
//   src/FullSystem/FullSystem.cpp:1004-1021, 1633-1644            flagPointsForRemoval core, setPrecalcValues
#include "orc_api.h"
#include <limits>
#include <pthread.h>
#include <sched.h>
#include "orc_math.h"
#include "orc_acc.h"
#include "orc_common.h"
#include <cmath>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

using namespace orc;

namespace {

enum ResState { RS_IN = 0, RS_OOB = 1, RS_OUTLIER = 2 };

// src/util/IndexThreadReduce.h:34-196 with the number of workers a run-time value (the reference fixes NUM_THREADS = 6,
// util/NumType.h:38): persistent workers take index chunks of `stepSize` from a shared counter under one mutex (stepSize 0 =
// ceil(n / workers)); a worker that got no chunk is still called once with (0, 0) so that per-thread state is initialised
// everywhere (:179-187); the caller sleeps on a condition variable until every worker reports done.  Used by the timed CPU
// baseline only (orc_ba_set_threads); the parity path of the tests stays single-threaded (tid 0).
class Reducer {
 public:
  typedef std::function<void(int, int, double*, int)> Fn;
  explicit Reducer(int n) : nthreads(n), isDone(n, 0), gotOne(n, 1) {
    for (int i = 0; i < n; i++) workers.emplace_back(&Reducer::workerLoop, this, i);
    // timed baseline hygiene: every worker on a core of its own (the i-th CPU this process may run on), so that a leg's time does
    // not depend on where the scheduler happens to put the threads
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0) {
      std::vector<int> cpus;
      for (int c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &allowed)) cpus.push_back(c);
      if ((int)cpus.size() >= n)
        for (int i = 0; i < n; i++) {
          cpu_set_t one;
          CPU_ZERO(&one);
          CPU_SET(cpus[i], &one);
          pinned += pthread_setaffinity_np(workers[i].native_handle(), sizeof(one), &one) == 0 ? 1 : 0;
        }
    }
  }
  int pinned = 0;     // workers that were bound to a core
  ~Reducer() {
    { std::unique_lock<std::mutex> lock(exMutex); running = false; todo_signal.notify_all(); }
    for (auto& t : workers) t.join();
  }
  double reduce(const Fn& f, int first, int end, int step = 0) {
    if (step == 0) step = ((end - first) + nthreads - 1) / nthreads;
    std::unique_lock<std::mutex> lock(exMutex);
    stats = 0;
    callPerIndex = f; nextIndex = first; maxIndex = end; stepSize = step;
    for (int i = 0; i < nthreads; i++) { isDone[i] = 0; gotOne[i] = 0; }
    todo_signal.notify_all();
    while (true) {
      done_signal.wait(lock);
      bool allDone = true;
      for (int i = 0; i < nthreads; i++) allDone = allDone && isDone[i];
      if (allDone) break;
    }
    nextIndex = 0; maxIndex = 0;
    return stats;
  }
  const int nthreads;

 private:
  void workerLoop(int idx) {
    std::unique_lock<std::mutex> lock(exMutex);
    while (running) {
      int todo = 0;
      bool gotSomething = false;
      if (nextIndex < maxIndex) { todo = nextIndex; nextIndex += stepSize; gotSomething = true; }
      if (gotSomething) {
        const int hi = std::min(todo + stepSize, maxIndex);
        lock.unlock();
        double sres = 0;
        callPerIndex(todo, hi, &sres, idx);
        lock.lock();
        gotOne[idx] = 1;
        stats += sres;
      } else {
        if (!gotOne[idx]) {
          lock.unlock();
          double sres = 0;
          callPerIndex(0, 0, &sres, idx);
          lock.lock();
          gotOne[idx] = 1;
          stats += sres;
        }
        isDone[idx] = 1;
        done_signal.notify_all();
        todo_signal.wait(lock);
      }
    }
  }
  std::vector<std::thread> workers;
  std::vector<char> isDone, gotOne;
  std::mutex exMutex;
  std::condition_variable todo_signal, done_signal;
  int nextIndex = 0, maxIndex = 0, stepSize = 1;
  bool running = true;
  double stats = 0;
  Fn callPerIndex;
};

// one thread's private accumulators (AccumulatedTopHessianSSE::acc[tid], AccumulatedSCHessianSSE::accD/E/EB/Hcc/bc[tid])
struct AccSet {
  std::vector<AccumulatorApprox> topA, topL;
  std::vector<AccumulatorXX<8, 8>> D;
  std::vector<AccumulatorXX<8, 4>> E;
  std::vector<AccumulatorX<8>> EB;
  AccumulatorXX<4, 4> Hcc;
  AccumulatorX<4> bc;
  int nresA = 0, nresL = 0;
};

struct RawJ {  // RawResidualJacobian.h:32-65, flattened in the ABI's field order (74 floats)
  float resF[8];
  float Jpdxi[2][6];
  float Jpdc[2][4];
  float Jpdd[2];
  float JIdx[2][8];
  float JabF[2][8];
  float JIdx2[4];    // (0,0) (0,1) (1,0) (1,1)
  float JabJIdx[4];
  float Jab2[4];
};
static_assert(sizeof(RawJ) == 74 * 4, "RawJ must be 74 floats");

struct Calib {
  double value_zero[4], value_scaled[4], value[4], step[4], value_backup[4], value_minus_value_zero[4], step_backup[4] = {0, 0, 0, 0};
  float value_scaledf[4], value_scaledi[4];
  float fxl() const { return value_scaledf[0]; }
  float fyl() const { return value_scaledf[1]; }
  float cxl() const { return value_scaledf[2]; }
  float cyl() const { return value_scaledf[3]; }
  float fxli() const { return value_scaledi[0]; }
  float fyli() const { return value_scaledi[1]; }
  void finishScaled() {
    for (int i = 0; i < 4; i++) value_scaledf[i] = (float)value_scaled[i];
    value_scaledi[0] = 1.0f / value_scaledf[0];
    value_scaledi[1] = 1.0f / value_scaledf[1];
    value_scaledi[2] = -value_scaledf[2] / value_scaledf[0];
    value_scaledi[3] = -value_scaledf[3] / value_scaledf[1];
    for (int i = 0; i < 4; i++) value_minus_value_zero[i] = value[i] - value_zero[i];
  }
  void setValue(const double* v) {  // HessianBlocks.h:318-333
    for (int i = 0; i < 4; i++) value[i] = v[i];
    value_scaled[0] = SCALE_F * value[0];
    value_scaled[1] = SCALE_F * value[1];
    value_scaled[2] = SCALE_C * value[2];
    value_scaled[3] = SCALE_C * value[3];
    finishScaled();
  }
  void setValueScaled(const double* vs) {  // HessianBlocks.h:335-349
    for (int i = 0; i < 4; i++) value_scaled[i] = vs[i];
    value[0] = SCALE_F_INVERSE * value_scaled[0];
    value[1] = SCALE_F_INVERSE * value_scaled[1];
    value[2] = SCALE_C_INVERSE * value_scaled[2];
    value[3] = SCALE_C_INVERSE * value_scaled[3];
    finishScaled();
  }
};

struct Frame {
  SE3 worldToCam_evalPT;
  double state_zero[10], state_scaled[10], state[10], step[10], state_backup[10], step_backup[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  SE3 PRE_worldToCam, PRE_camToWorld;
  float ab_exposure, frameEnergyTH;
  int frameID;
  const float* dI;
  double nullspaces_pose[6][6];  // [row][col]
  double nullspaces_scale[6];
  double nullspaces_affine[4][2];
  // EFFrame
  double prior[8], delta_prior[8], delta[8];

  double aff_a() const { return state_scaled[6]; }
  double aff_b() const { return state_scaled[7]; }
  double aff0_a() const { return state_zero[6] * SCALE_A; }
  double aff0_b() const { return state_zero[7] * SCALE_B; }

  void setState(const double* s) {  // HessianBlocks.h:161-181
    for (int i = 0; i < 10; i++) state[i] = s[i];
    for (int i = 0; i < 3; i++) state_scaled[i] = SCALE_XI_TRANS * state[i];
    for (int i = 3; i < 6; i++) state_scaled[i] = SCALE_XI_ROT * state[i];
    state_scaled[6] = SCALE_A * state[6];
    state_scaled[7] = SCALE_B * state[7];
    state_scaled[8] = SCALE_A * state[8];
    state_scaled[9] = SCALE_B * state[9];
    PRE_worldToCam = se3_mul(se3_exp(state_scaled), worldToCam_evalPT);
    PRE_camToWorld = se3_inv(PRE_worldToCam);
  }
  void setStateZero(const double* sz) {  // HessianBlocks.cpp:78-123
    for (int i = 0; i < 10; i++) state_zero[i] = sz[i];
    SE3 Tinv = se3_inv(worldToCam_evalPT);
    for (int i = 0; i < 6; i++) {
      double eps[6] = {0, 0, 0, 0, 0, 0};
      eps[i] = 1e-3;
      SE3 EepsP = se3_exp(eps);
      eps[i] = -1e-3;
      SE3 EepsM = se3_exp(eps);
      SE3 P = se3_mul(se3_mul(worldToCam_evalPT, EepsP), Tinv);
      SE3 M = se3_mul(se3_mul(worldToCam_evalPT, EepsM), Tinv);
      double lp[6], lm[6];
      se3_log(P, lp); se3_log(M, lm);
      for (int r = 0; r < 6; r++) nullspaces_pose[r][i] = (lp[r] - lm[r]) / (2e-3);
    }
    SE3 P = worldToCam_evalPT;
    for (int i = 0; i < 3; i++) P.t[i] *= 1.00001;
    P = se3_mul(P, Tinv);
    SE3 M = worldToCam_evalPT;
    for (int i = 0; i < 3; i++) M.t[i] /= 1.00001;
    M = se3_mul(M, Tinv);
    double lp[6], lm[6];
    se3_log(P, lp); se3_log(M, lm);
    for (int r = 0; r < 6; r++) nullspaces_scale[r] = (lp[r] - lm[r]) / (2e-3);
    for (int r = 0; r < 4; r++) nullspaces_affine[r][0] = nullspaces_affine[r][1] = 0;
    nullspaces_affine[0][0] = 1; nullspaces_affine[1][0] = 0;
    nullspaces_affine[0][1] = 0; nullspaces_affine[1][1] = expf((float)aff0_a()) * ab_exposure;
  }
  void setEvalPT(const SE3& T, const double* s) { worldToCam_evalPT = T; setState(s); setStateZero(s); }
  void getPrior(double* p, double optA, double optB, int solverMode) const {  // HessianBlocks.h:239-265
    for (int i = 0; i < 10; i++) p[i] = 0;
    if (frameID == 0) {
      for (int i = 0; i < 3; i++) p[i] = setting_initialTransPrior;
      for (int i = 3; i < 6; i++) p[i] = setting_initialRotPrior;
      if (solverMode & SOLVER_REMOVE_POSEPRIOR) for (int i = 0; i < 6; i++) p[i] = 0;
      p[6] = setting_initialAffAPrior;
      p[7] = setting_initialAffBPrior;
    } else {
      p[6] = (optA < 0) ? (double)setting_initialAffAPrior : optA;
      p[7] = (optB < 0) ? (double)setting_initialAffBPrior : optB;
    }
    p[8] = setting_initialAffAPrior;
    p[9] = setting_initialAffBPrior;
  }
};

struct Precalc {  // HessianBlocks.h:72-97
  float PRE_RTll[9], PRE_KRKiTll[9], PRE_RKiTll[9], PRE_RTll_0[9];
  float PRE_aff_mode[2], PRE_b0_mode;
  float PRE_tTll[3], PRE_KtTll[3], PRE_tTll_0[3];
};

struct Point {
  float u, v, idepth, idepth_zero, idepth_scaled, idepth_zero_scaled;
  float color[8], weights[8];
  int host;
  bool hasDepthPrior;
  float step, idepth_backup, step_backup = 0;
  float idepth_hessian, maxRelBaseline;
  int numGoodResiduals;
  int rbeg, rend;
  // EFPoint
  float priorF, deltaF, bdSumF, HdiF;
  float Hdd_accLF, Hcd_accLF[4], bd_accLF;
  float Hdd_accAF, Hcd_accAF[4], bd_accAF;
  void setIdepth(float id) { idepth = id; idepth_scaled = SCALE_IDEPTH * id; }
  void setIdepthZero(float id) { idepth_zero = id; idepth_zero_scaled = SCALE_IDEPTH * id; }
};

struct Residual {
  int point, host, target;
  ResState state_state, state_NewState;
  double state_energy, state_NewEnergy, state_NewEnergyWithOutlier;
  RawJ Jnew;  // PointFrameResidual::J
  RawJ Jef;   // EFResidual::J
  float res_toZeroF[8], JpJdF[8];
  bool isLinearized, isActive;
  bool isNew, toRemove;
  float projectedTo[8][2], centerProjectedTo[3];
  void resetOOB() { state_NewEnergy = state_energy = 0; state_NewState = RS_OUTLIER; state_state = RS_IN; }
};

}  // namespace

struct orc_ba {
  int nf, np, nr, w, h;
  float wM3G, hM3G;
  Calib HCalib;
  std::vector<Frame> frames;
  std::vector<Point> points;
  std::vector<Residual> res;
  std::vector<Precalc> precalc;  // [host*nf + target]  (host->targetPrecalc[target->idx])
  std::vector<double> adHost, adTarget;  // [h + t*nf][64]
  std::vector<float> adHostF, adTargetF;
  std::vector<float> adHTdeltaF;  // [h+t*nf][8]
  float cDeltaF[4];
  double cPrior[4];
  float cPriorF[4];
  MatX HM;
  VecX bM;
  int solverMode;
  double affineOptModeA, affineOptModeB;
  bool forceAcceptStep;
  // accumulators: T[tid]; T[0] alone on the single-thread (parity) path
  std::vector<AccSet> T = std::vector<AccSet>(1);
  Reducer* red = nullptr;   // null: everything runs on the calling thread
  int nresA, nresL, resInM;
  orc_ba_opt_result_t lastResult{0, 0, 0, 0};
  ~orc_ba() { delete red; }
  MatX lastHS;
  VecX lastbS, lastX;
  std::vector<float> stepTrace;  // the loop's stepsize of every iteration (SOLVER_STEPMOMENTUM; 1 otherwise)
  std::vector<VecX> xTrace;   // lastX of every solveSystemF of the latest optimize() (test infrastructure: per-iteration parity of the pose updates)
  std::vector<VecX> lastNullspaces_pose, lastNullspaces_scale;

  // ---------------------------------------------------------------- host tables
  void setPrecalc(int hi, int ti) {  // HessianBlocks.cpp:206-242
    Frame& host = frames[hi];
    Frame& target = frames[ti];
    Precalc& P = precalc[hi * nf + ti];
    SE3 leftToLeft_0 = se3_mul(target.worldToCam_evalPT, se3_inv(host.worldToCam_evalPT));
    for (int i = 0; i < 9; i++) P.PRE_RTll_0[i] = (float)leftToLeft_0.R[i];
    for (int i = 0; i < 3; i++) P.PRE_tTll_0[i] = (float)leftToLeft_0.t[i];
    SE3 leftToLeft = se3_mul(target.PRE_worldToCam, host.PRE_camToWorld);
    for (int i = 0; i < 9; i++) P.PRE_RTll[i] = (float)leftToLeft.R[i];
    for (int i = 0; i < 3; i++) P.PRE_tTll[i] = (float)leftToLeft.t[i];
    float K[9] = {HCalib.fxl(), 0, HCalib.cxl(), 0, HCalib.fyl(), HCalib.cyl(), 0, 0, 1};
    float Ki[9];
    mat3_inv<float>(K, Ki);
    float KR[9];
    mat3_mul<float>(K, P.PRE_RTll, KR);
    mat3_mul<float>(KR, Ki, P.PRE_KRKiTll);
    mat3_mul<float>(P.PRE_RTll, Ki, P.PRE_RKiTll);
    mat3_vec<float>(K, P.PRE_tTll, P.PRE_KtTll);
    double aff[2];
    fromToVecExposure(host.ab_exposure, target.ab_exposure, host.aff_a(), host.aff_b(), target.aff_a(), target.aff_b(), aff);
    P.PRE_aff_mode[0] = (float)aff[0];
    P.PRE_aff_mode[1] = (float)aff[1];
    P.PRE_b0_mode = (float)host.aff0_b();
  }
  void setAdjointsF() {  // EnergyFunctional.cpp:41-119
    adHost.assign((size_t)nf * nf * 64, 0.0);
    adTarget.assign((size_t)nf * nf * 64, 0.0);
    for (int h = 0; h < nf; h++)
      for (int t = 0; t < nf; t++) {
        SE3 hostToTarget = se3_mul(frames[t].worldToCam_evalPT, se3_inv(frames[h].worldToCam_evalPT));
        double AH[8][8] = {{0}}, AT[8][8] = {{0}};
        for (int i = 0; i < 8; i++) AH[i][i] = AT[i][i] = 1;
        double Adj[36];
        se3_adj(hostToTarget, Adj);
        for (int i = 0; i < 6; i++)
          for (int j = 0; j < 6; j++) AH[i][j] = -Adj[j * 6 + i];
        double aff[2];
        fromToVecExposure(frames[h].ab_exposure, frames[t].ab_exposure, frames[h].aff0_a(), frames[h].aff0_b(),
                          frames[t].aff0_a(), frames[t].aff0_b(), aff);
        float affLL0 = (float)aff[0];
        AT[6][6] = -affLL0;
        AT[7][7] = -1;
        AH[6][6] = affLL0;
        AH[7][7] = affLL0;
        for (int j = 0; j < 8; j++) {
          for (int i = 0; i < 3; i++) { AH[i][j] *= SCALE_XI_TRANS; AT[i][j] *= SCALE_XI_TRANS; }
          for (int i = 3; i < 6; i++) { AH[i][j] *= SCALE_XI_ROT; AT[i][j] *= SCALE_XI_ROT; }
          AH[6][j] *= SCALE_A; AT[6][j] *= SCALE_A;
          AH[7][j] *= SCALE_B; AT[7][j] *= SCALE_B;
        }
        for (int i = 0; i < 8; i++)
          for (int j = 0; j < 8; j++) {
            adHost[(size_t)(h + t * nf) * 64 + i * 8 + j] = AH[i][j];
            adTarget[(size_t)(h + t * nf) * 64 + i * 8 + j] = AT[i][j];
          }
      }
    for (int i = 0; i < 4; i++) { cPrior[i] = setting_initialCalibHessian; cPriorF[i] = (float)cPrior[i]; }
    adHostF.resize(adHost.size()); adTargetF.resize(adTarget.size());
    for (size_t i = 0; i < adHost.size(); i++) { adHostF[i] = (float)adHost[i]; adTargetF[i] = (float)adTarget[i]; }
  }
  void setDeltaF() {  // EnergyFunctional.cpp:173-207
    adHTdeltaF.assign((size_t)nf * nf * 8, 0.f);
    for (int h = 0; h < nf; h++)
      for (int t = 0; t < nf; t++) {
        int idx = h + t * nf;
        float dh[8], dt[8];
        for (int i = 0; i < 8; i++) {
          dh[i] = (float)(frames[h].state[i] - frames[h].state_zero[i]);
          dt[i] = (float)(frames[t].state[i] - frames[t].state_zero[i]);
        }
        // row-vector * matrix, each a float dot product evaluated left to right; then summed
        for (int j = 0; j < 8; j++) {
          float sh = 0, st = 0;
          for (int i = 0; i < 8; i++) { sh += dh[i] * adHostF[(size_t)idx * 64 + i * 8 + j]; st += dt[i] * adTargetF[(size_t)idx * 64 + i * 8 + j]; }
          adHTdeltaF[(size_t)idx * 8 + j] = sh + st;
        }
      }
    for (int i = 0; i < 4; i++) cDeltaF[i] = (float)HCalib.value_minus_value_zero[i];
    for (Frame& f : frames) {
      for (int i = 0; i < 8; i++) { f.delta[i] = f.state[i] - f.state_zero[i]; f.delta_prior[i] = f.state[i]; }
    }
    for (Point& p : points) p.deltaF = p.idepth - p.idepth_zero;
  }
  void setPrecalcValues() {  // FullSystem.cpp:1633-1644
    for (int h = 0; h < nf; h++)
      for (int t = 0; t < nf; t++) setPrecalc(h, t);
    setDeltaF();
  }

  // ---------------------------------------------------------------- linearize
  // ResidualProjections.h:45-58
  bool projectPoint2(float u_pt, float v_pt, float idepth, const float* KRKi, const float* Kt, float& Ku, float& Kv) const {
    float ptp[3];
    for (int r = 0; r < 3; r++) ptp[r] = ((KRKi[r * 3 + 0] * u_pt + KRKi[r * 3 + 1] * v_pt) + KRKi[r * 3 + 2] * 1.0f) + Kt[r] * idepth;
    Ku = ptp[0] / ptp[2];
    Kv = ptp[1] / ptp[2];
    return Ku > 1.1f && Kv > 1.1f && Ku < wM3G && Kv < hM3G;
  }
  // ResidualProjections.h:64-96
  bool projectPointFull(float u_pt, float v_pt, float idepth, int dx, int dy, const float* R, const float* t,
                        float& drescale, float& u, float& v, float& Ku, float& Kv, float* KliP, float& new_idepth) const {
    KliP[0] = (u_pt + dx - HCalib.cxl()) * HCalib.fxli();
    KliP[1] = (v_pt + dy - HCalib.cyl()) * HCalib.fyli();
    KliP[2] = 1;
    float ptp[3];
    for (int r = 0; r < 3; r++) ptp[r] = ((R[r * 3 + 0] * KliP[0] + R[r * 3 + 1] * KliP[1]) + R[r * 3 + 2] * KliP[2]) + t[r] * idepth;
    drescale = 1.0f / ptp[2];
    new_idepth = idepth * drescale;
    if (!(drescale > 0)) return false;
    u = ptp[0] * drescale;
    v = ptp[1] * drescale;
    Ku = u * HCalib.fxl() + HCalib.cxl();
    Kv = v * HCalib.fyl() + HCalib.cyl();
    return Ku > 1.1f && Kv > 1.1f && Ku < wM3G && Kv < hM3G;
  }

  // Residuals.cpp:83-336
  double linearize(Residual& r) {
    r.state_NewEnergyWithOutlier = -1;
    if (r.state_state == RS_OOB) { r.state_NewState = RS_OOB; return r.state_energy; }
    const Point& point = points[r.point];
    const Precalc& precalc_ = precalc[r.host * nf + r.target];
    const Frame& host = frames[r.host];
    const Frame& target = frames[r.target];
    float energyLeft = 0;
    const float* dIl = target.dI;
    const float* PRE_KRKiTll = precalc_.PRE_KRKiTll;
    const float* PRE_KtTll = precalc_.PRE_KtTll;
    const float* PRE_RTll_0 = precalc_.PRE_RTll_0;
    const float* PRE_tTll_0 = precalc_.PRE_tTll_0;
    const float* color = point.color;
    const float* weights = point.weights;
    float affLL0 = precalc_.PRE_aff_mode[0], affLL1 = precalc_.PRE_aff_mode[1];
    float b0 = precalc_.PRE_b0_mode;
    RawJ* J = &r.Jnew;

    float d_xi_x[6], d_xi_y[6], d_C_x[4], d_C_y[4], d_d_x, d_d_y;
    {
      float drescale, u, v, new_idepth, Ku, Kv, KliP[3];
      if (!projectPointFull(point.u, point.v, point.idepth_zero_scaled, 0, 0, PRE_RTll_0, PRE_tTll_0, drescale, u, v, Ku, Kv, KliP, new_idepth)) {
        r.state_NewState = RS_OOB;
        return r.state_energy;
      }
      r.centerProjectedTo[0] = Ku; r.centerProjectedTo[1] = Kv; r.centerProjectedTo[2] = new_idepth;
      const float fxl = HCalib.fxl(), fyl = HCalib.fyl(), fxli = HCalib.fxli(), fyli = HCalib.fyli();
#define R0(i, j) PRE_RTll_0[(i) * 3 + (j)]
      d_d_x = drescale * (PRE_tTll_0[0] - PRE_tTll_0[2] * u) * SCALE_IDEPTH * fxl;
      d_d_y = drescale * (PRE_tTll_0[1] - PRE_tTll_0[2] * v) * SCALE_IDEPTH * fyl;
      d_C_x[2] = drescale * (R0(2, 0) * u - R0(0, 0));
      d_C_x[3] = fxl * drescale * (R0(2, 1) * u - R0(0, 1)) * fyli;
      d_C_x[0] = KliP[0] * d_C_x[2];
      d_C_x[1] = KliP[1] * d_C_x[3];
      d_C_y[2] = fyl * drescale * (R0(2, 0) * v - R0(1, 0)) * fxli;
      d_C_y[3] = drescale * (R0(2, 1) * v - R0(1, 1));
      d_C_y[0] = KliP[0] * d_C_y[2];
      d_C_y[1] = KliP[1] * d_C_y[3];
#undef R0
      d_C_x[0] = (d_C_x[0] + u) * SCALE_F;
      d_C_x[1] *= SCALE_F;
      d_C_x[2] = (d_C_x[2] + 1) * SCALE_C;
      d_C_x[3] *= SCALE_C;
      d_C_y[0] *= SCALE_F;
      d_C_y[1] = (d_C_y[1] + v) * SCALE_F;
      d_C_y[2] *= SCALE_C;
      d_C_y[3] = (d_C_y[3] + 1) * SCALE_C;
      d_xi_x[0] = new_idepth * fxl;
      d_xi_x[1] = 0;
      d_xi_x[2] = -new_idepth * u * fxl;
      d_xi_x[3] = -u * v * fxl;
      d_xi_x[4] = (1 + u * u) * fxl;
      d_xi_x[5] = -v * fxl;
      d_xi_y[0] = 0;
      d_xi_y[1] = new_idepth * fyl;
      d_xi_y[2] = -new_idepth * v * fyl;
      d_xi_y[3] = -(1 + v * v) * fyl;
      d_xi_y[4] = u * v * fyl;
      d_xi_y[5] = u * fyl;
    }
    for (int i = 0; i < 6; i++) { J->Jpdxi[0][i] = d_xi_x[i]; J->Jpdxi[1][i] = d_xi_y[i]; }
    for (int i = 0; i < 4; i++) { J->Jpdc[0][i] = d_C_x[i]; J->Jpdc[1][i] = d_C_y[i]; }
    J->Jpdd[0] = d_d_x; J->Jpdd[1] = d_d_y;

    float JIdxJIdx_00 = 0, JIdxJIdx_11 = 0, JIdxJIdx_10 = 0;
    float JabJIdx_00 = 0, JabJIdx_01 = 0, JabJIdx_10 = 0, JabJIdx_11 = 0;
    float JabJab_00 = 0, JabJab_01 = 0, JabJab_11 = 0;
    float wJI2_sum = 0;
    for (int idx = 0; idx < patternNum; idx++) {
      float Ku, Kv;
      if (!projectPoint2(point.u + patternP[idx][0], point.v + patternP[idx][1], point.idepth_scaled, PRE_KRKiTll, PRE_KtTll, Ku, Kv)) {
        r.state_NewState = RS_OOB;
        return r.state_energy;
      }
      r.projectedTo[idx][0] = Ku;
      r.projectedTo[idx][1] = Kv;
      float hitColor[3];
      interp33(dIl, Ku, Kv, w, hitColor);
      float residual = hitColor[0] - (float)(affLL0 * color[idx] + affLL1);
      float drdA = (color[idx] - b0);
      if (!std::isfinite((float)hitColor[0])) { r.state_NewState = RS_OOB; return r.state_energy; }
      float wgt = sqrtf(setting_outlierTHSumComponent / (setting_outlierTHSumComponent + (hitColor[1] * hitColor[1] + hitColor[2] * hitColor[2])));
      wgt = 0.5f * (wgt + weights[idx]);
      float hw = fabsf(residual) < setting_huberTH ? 1 : setting_huberTH / fabsf(residual);
      energyLeft += wgt * wgt * hw * residual * residual * (2 - hw);
      {
        if (hw < 1) hw = sqrtf(hw);
        hw = hw * wgt;
        hitColor[1] *= hw;
        hitColor[2] *= hw;
        J->resF[idx] = residual * hw;
        J->JIdx[0][idx] = hitColor[1];
        J->JIdx[1][idx] = hitColor[2];
        J->JabF[0][idx] = drdA * hw;
        J->JabF[1][idx] = hw;
        JIdxJIdx_00 += hitColor[1] * hitColor[1];
        JIdxJIdx_11 += hitColor[2] * hitColor[2];
        JIdxJIdx_10 += hitColor[1] * hitColor[2];
        JabJIdx_00 += drdA * hw * hitColor[1];
        JabJIdx_01 += drdA * hw * hitColor[2];
        JabJIdx_10 += hw * hitColor[1];
        JabJIdx_11 += hw * hitColor[2];
        JabJab_00 += drdA * drdA * hw * hw;
        JabJab_01 += drdA * hw * hw;
        JabJab_11 += hw * hw;
        wJI2_sum += hw * hw * (hitColor[1] * hitColor[1] + hitColor[2] * hitColor[2]);
        if (affineOptModeA < 0) J->JabF[0][idx] = 0;
        if (affineOptModeB < 0) J->JabF[1][idx] = 0;
      }
    }
    J->JIdx2[0] = JIdxJIdx_00; J->JIdx2[1] = JIdxJIdx_10; J->JIdx2[2] = JIdxJIdx_10; J->JIdx2[3] = JIdxJIdx_11;
    J->JabJIdx[0] = JabJIdx_00; J->JabJIdx[1] = JabJIdx_01; J->JabJIdx[2] = JabJIdx_10; J->JabJIdx[3] = JabJIdx_11;
    J->Jab2[0] = JabJab_00; J->Jab2[1] = JabJab_01; J->Jab2[2] = JabJab_01; J->Jab2[3] = JabJab_11;

    r.state_NewEnergyWithOutlier = energyLeft;
    float th = std::max<float>(host.frameEnergyTH, target.frameEnergyTH);
    if (energyLeft > th || wJI2_sum < 2) {
      energyLeft = th;
      r.state_NewState = RS_OUTLIER;
    } else {
      r.state_NewState = RS_IN;
    }
    r.state_NewEnergy = energyLeft;
    return energyLeft;
  }

  void takeDataF(Residual& r) {  // EnergyFunctionalStructs.cpp:37-51
    std::swap(r.Jef, r.Jnew);
    const RawJ* J = &r.Jef;
    float JI_JI_Jd[2];
    JI_JI_Jd[0] = J->JIdx2[0] * J->Jpdd[0] + J->JIdx2[1] * J->Jpdd[1];
    JI_JI_Jd[1] = J->JIdx2[2] * J->Jpdd[0] + J->JIdx2[3] * J->Jpdd[1];
    for (int i = 0; i < 6; i++) r.JpJdF[i] = J->Jpdxi[0][i] * JI_JI_Jd[0] + J->Jpdxi[1][i] * JI_JI_Jd[1];
    r.JpJdF[6] = J->JabJIdx[0] * J->Jpdd[0] + J->JabJIdx[1] * J->Jpdd[1];
    r.JpJdF[7] = J->JabJIdx[2] * J->Jpdd[0] + J->JabJIdx[3] * J->Jpdd[1];
  }
  void applyRes(Residual& r) {  // Residuals.cpp:367-385 (copyJacobians = true)
    if (r.state_state == RS_OOB) return;
    if (r.state_NewState == RS_IN) { r.isActive = true; takeDataF(r); }
    else r.isActive = false;
    r.state_state = r.state_NewState;
    r.state_energy = r.state_NewEnergy;
  }
  void fixLinearizationF(Residual& r) {  // EnergyFunctionalStructs.cpp:96-123
    const float* dp = &adHTdeltaF[(size_t)(r.host + nf * r.target) * 8];
    const RawJ* J = &r.Jef;
    float dx = dot6(J->Jpdxi[0], dp) + dot4(J->Jpdc[0], cDeltaF) + J->Jpdd[0] * points[r.point].deltaF;
    float dy = dot6(J->Jpdxi[1], dp) + dot4(J->Jpdc[1], cDeltaF) + J->Jpdd[1] * points[r.point].deltaF;
    for (int i = 0; i < 8; i++) {
      float rtz = J->resF[i];
      rtz = rtz - J->JIdx[0][i] * dx;
      rtz = rtz - J->JIdx[1][i] * dy;
      rtz = rtz - J->JabF[0][i] * dp[6];
      rtz = rtz - J->JabF[1][i] * dp[7];
      r.res_toZeroF[i] = rtz;
    }
    r.isLinearized = true;
  }
  static float dot6(const float* a, const float* b) { float s = 0; for (int i = 0; i < 6; i++) s += a[i] * b[i]; return s; }
  static float dot4(const float* a, const float* b) { float s = 0; for (int i = 0; i < 4; i++) s += a[i] * b[i]; return s; }

  void setNewFrameEnergyTH() {  // FullSystemOptimize.cpp:98-139
    std::vector<float> allResVec;
    int newFrame = nf - 1;
    for (Residual& r : res)
      if (!r.isLinearized && r.state_NewEnergyWithOutlier >= 0 && r.target == newFrame) allResVec.push_back((float)r.state_NewEnergyWithOutlier);
    if (allResVec.empty()) { frames[newFrame].frameEnergyTH = 12 * 12 * patternNum; return; }
    int nthIdx = setting_frameEnergyTHN * allResVec.size();
    std::nth_element(allResVec.begin(), allResVec.begin() + nthIdx, allResVec.end());
    float nthElement = sqrtf(allResVec[nthIdx]);
    float th = nthElement * setting_frameEnergyTHFacMedian;
    th = 26.0f * setting_frameEnergyTHConstWeight + th * (1 - setting_frameEnergyTHConstWeight);
    th = th * th;
    th *= setting_overallEnergyTHWeight * setting_overallEnergyTHWeight;
    frames[newFrame].frameEnergyTH = th;
  }
  // FullSystemOptimize.cpp:142-203 (activeResiduals = residuals that are not linearized)
  double linearizeAll(bool fixLinearization) {
    double lastEnergyP = 0;
    if (red) {  // treadReduce.reduce(linearizeAll_Reductor, 0, activeResiduals.size(), 0)  (:142-160)
      lastEnergyP = red->reduce([&](int lo, int hi, double* stats, int) {
        for (int k = lo; k < hi; k++) {
          Residual& r = res[k];
          if (r.isLinearized) continue;
          *stats += linearize(r);
          if (fixLinearization) applyRes(r);
        }
      }, 0, nr, 0);
      if (fixLinearization) for (Residual& r : res) if (!r.isLinearized) afterFixedLinearization(r);   // (the timed legs: bookkeeping after the reduce)
    } else {
      for (Residual& r : res) {
        if (r.isLinearized) continue;
        lastEnergyP += linearize(r);
        if (fixLinearization) { applyRes(r); afterFixedLinearization(r); }
      }
    }
    setNewFrameEnergyTH();
    return lastEnergyP;
  }
  // linearizeAll_Reductor with fixLinearization, after applyRes (FullSystemOptimize.cpp:62-84)
  void afterFixedLinearization(Residual& r) {
    r.toRemove = false;
    if (r.isActive) {
      if (r.isNew) {
        Point& p = points[r.point];
        const Precalc& pc = precalc[r.host * nf + r.target];
        float ptp_inf[3], ptp[3];
        for (int i = 0; i < 3; i++) ptp_inf[i] = (pc.PRE_KRKiTll[i * 3 + 0] * p.u + pc.PRE_KRKiTll[i * 3 + 1] * p.v) + pc.PRE_KRKiTll[i * 3 + 2] * 1.0f;
        for (int i = 0; i < 3; i++) ptp[i] = ptp_inf[i] + pc.PRE_KtTll[i] * p.idepth_scaled;
        const float dx = ptp_inf[0] / ptp_inf[2] - ptp[0] / ptp[2], dy = ptp_inf[1] / ptp_inf[2] - ptp[1] / ptp[2];
        const float relBS = 0.01 * sqrtf(dx * dx + dy * dy);
        if (relBS > p.maxRelBaseline) p.maxRelBaseline = relBS;
        p.numGoodResiduals++;
      }
    } else r.toRemove = true;
  }

  // ---------------------------------------------------------------- accumulate
  template <int mode>
  void addPointTop(Point& p, std::vector<AccumulatorApprox>& acc, int& nres) {  // AccumulatedTopHessian.cpp:36-198
    const float* dc = cDeltaF;
    float dd = p.deltaF;
    float bd_acc = 0, Hdd_acc = 0, Hcd_acc[4] = {0, 0, 0, 0};
    for (int ri = p.rbeg; ri < p.rend; ri++) {
      Residual& r = res[ri];
      if (mode == 0) { if (r.isLinearized || !r.isActive) continue; }
      if (mode == 1) { if (!r.isLinearized || !r.isActive) continue; }
      if (mode == 2) { if (!r.isActive) continue; }
      const RawJ* rJ = &r.Jef;
      int htIDX = r.host + r.target * nf;
      const float* dp = &adHTdeltaF[(size_t)htIDX * 8];
      float resApprox[8];
      if (mode == 0) for (int i = 0; i < 8; i++) resApprox[i] = rJ->resF[i];
      if (mode == 1) {
        float dx = dot6(rJ->Jpdxi[0], dp) + dot4(rJ->Jpdc[0], dc) + rJ->Jpdd[0] * dd;
        float dy = dot6(rJ->Jpdxi[1], dp) + dot4(rJ->Jpdc[1], dc) + rJ->Jpdd[1] * dd;
        for (int i = 0; i < 8; i++) {
          float rtz = r.res_toZeroF[i];
          rtz = rtz + rJ->JIdx[0][i] * dx;
          rtz = rtz + rJ->JIdx[1][i] * dy;
          rtz = rtz + rJ->JabF[0][i] * dp[6];
          rtz = rtz + rJ->JabF[1][i] * dp[7];
          resApprox[i] = rtz;
        }
      }
      if (mode == 2) for (int i = 0; i < 8; i++) resApprox[i] = r.res_toZeroF[i];

      float JI_r[2] = {0, 0}, Jab_r[2] = {0, 0}, rr = 0;
      for (int i = 0; i < patternNum; i++) {
        JI_r[0] += resApprox[i] * rJ->JIdx[0][i];
        JI_r[1] += resApprox[i] * rJ->JIdx[1][i];
        Jab_r[0] += resApprox[i] * rJ->JabF[0][i];
        Jab_r[1] += resApprox[i] * rJ->JabF[1][i];
        rr += resApprox[i] * resApprox[i];
      }
      acc[htIDX].update(rJ->Jpdc[0], rJ->Jpdxi[0], rJ->Jpdc[1], rJ->Jpdxi[1], rJ->JIdx2[0], rJ->JIdx2[1], rJ->JIdx2[3]);
      acc[htIDX].updateBotRight(rJ->Jab2[0], rJ->Jab2[1], Jab_r[0], rJ->Jab2[3], Jab_r[1], rr);
      acc[htIDX].updateTopRight(rJ->Jpdc[0], rJ->Jpdxi[0], rJ->Jpdc[1], rJ->Jpdxi[1], rJ->JabJIdx[0], rJ->JabJIdx[1],
                                rJ->JabJIdx[2], rJ->JabJIdx[3], JI_r[0], JI_r[1]);
      float Ji2_Jpdd[2];
      Ji2_Jpdd[0] = rJ->JIdx2[0] * rJ->Jpdd[0] + rJ->JIdx2[1] * rJ->Jpdd[1];
      Ji2_Jpdd[1] = rJ->JIdx2[2] * rJ->Jpdd[0] + rJ->JIdx2[3] * rJ->Jpdd[1];
      bd_acc += JI_r[0] * rJ->Jpdd[0] + JI_r[1] * rJ->Jpdd[1];
      Hdd_acc += Ji2_Jpdd[0] * rJ->Jpdd[0] + Ji2_Jpdd[1] * rJ->Jpdd[1];
      for (int i = 0; i < 4; i++) Hcd_acc[i] += rJ->Jpdc[0][i] * Ji2_Jpdd[0] + rJ->Jpdc[1][i] * Ji2_Jpdd[1];
      nres++;
    }
    if (mode == 0) {
      p.Hdd_accAF = Hdd_acc; p.bd_accAF = bd_acc;
      for (int i = 0; i < 4; i++) p.Hcd_accAF[i] = Hcd_acc[i];
    }
    if (mode == 1 || mode == 2) {
      p.Hdd_accLF = Hdd_acc; p.bd_accLF = bd_acc;
      for (int i = 0; i < 4; i++) p.Hcd_accLF[i] = Hcd_acc[i];
    }
    if (mode == 2) {
      for (int i = 0; i < 4; i++) p.Hcd_accAF[i] = 0;
      p.Hdd_accAF = 0; p.bd_accAF = 0;
    }
  }
  void addPointSC(Point& p, bool shiftPriorToZero, AccSet& A) {  // AccumulatedSCHessian.cpp:34-103
    int ngoodres = 0;
    for (int ri = p.rbeg; ri < p.rend; ri++) if (res[ri].isActive) ngoodres++;
    if (ngoodres == 0) { p.HdiF = 0; p.bdSumF = 0; p.idepth_hessian = 0; p.maxRelBaseline = 0; return; }
    float H = p.Hdd_accAF + p.Hdd_accLF + p.priorF;
    if (H < 1e-10) H = 1e-10;
    p.idepth_hessian = H;
    p.HdiF = 1.0 / H;
    p.bdSumF = p.bd_accAF + p.bd_accLF;
    if (shiftPriorToZero) p.bdSumF += p.priorF * p.deltaF;
    float Hcd[4];
    for (int i = 0; i < 4; i++) Hcd[i] = p.Hcd_accAF[i] + p.Hcd_accLF[i];
    A.Hcc.update(Hcd, Hcd, p.HdiF);
    A.bc.update(Hcd, p.bdSumF * p.HdiF);
    int nFrames2 = nf * nf;
    for (int r1i = p.rbeg; r1i < p.rend; r1i++) {
      Residual& r1 = res[r1i];
      if (!r1.isActive) continue;
      int r1ht = r1.host + r1.target * nf;
      for (int r2i = p.rbeg; r2i < p.rend; r2i++) {
        Residual& r2 = res[r2i];
        if (!r2.isActive) continue;
        A.D[r1ht + r2.target * nFrames2].update(r1.JpJdF, r2.JpJdF, p.HdiF);
      }
      A.E[r1ht].update(r1.JpJdF, Hcd, p.HdiF);
      A.EB[r1ht].update(r1.JpJdF, p.HdiF * p.bdSumF);
    }
  }
  void zeroTop(std::vector<AccumulatorApprox>& a) { a.resize((size_t)nf * nf); for (auto& x : a) x.initialize(); }
  void zeroSC(AccSet& A) {
    A.D.resize((size_t)nf * nf * nf); A.E.resize((size_t)nf * nf); A.EB.resize((size_t)nf * nf);
    for (auto& x : A.D) x.initialize();
    for (auto& x : A.E) x.initialize();
    for (auto& x : A.EB) x.initialize();
    A.Hcc.initialize(); A.bc.initialize();
  }
  void zeroAll() { for (AccSet& A : T) { zeroTop(A.topA); zeroTop(A.topL); zeroSC(A); A.nresA = A.nresL = 0; } }
  void accumulateAll() {  // EnergyFunctional.cpp:212-269
    if (red) {
      // setZero on every worker (reduce over an empty range), then addPointsInternal in chunks of 50 points with acc[tid]
      red->reduce([&](int, int, double*, int tid) { zeroTop(T[tid].topA); T[tid].nresA = 0; }, 0, 0, 0);
      red->reduce([&](int lo, int hi, double*, int tid) { for (int i = lo; i < hi; i++) addPointTop<0>(points[i], T[tid].topA, T[tid].nresA); }, 0, np, 50);
      red->reduce([&](int, int, double*, int tid) { zeroTop(T[tid].topL); T[tid].nresL = 0; }, 0, 0, 0);
      red->reduce([&](int lo, int hi, double*, int tid) { for (int i = lo; i < hi; i++) addPointTop<1>(points[i], T[tid].topL, T[tid].nresL); }, 0, np, 50);
      red->reduce([&](int, int, double*, int tid) { zeroSC(T[tid]); }, 0, 0, 0);
      red->reduce([&](int lo, int hi, double*, int tid) { for (int i = lo; i < hi; i++) addPointSC(points[i], true, T[tid]); }, 0, np, 50);
      nresA = nresL = 0;
      for (AccSet& A : T) { nresA += A.nresA; nresL += A.nresL; }
      return;
    }
    // single accumulator copy, tid 0
    AccSet& A = T[0];
    zeroTop(A.topA); nresA = 0;
    for (Point& p : points) addPointTop<0>(p, A.topA, nresA);
    zeroTop(A.topL); nresL = 0;
    for (Point& p : points) addPointTop<1>(p, A.topL, nresL);
    zeroSC(A);
    for (Point& p : points) addPointSC(p, true, A);
  }

  // ---------------------------------------------------------------- stitch
  static void mul88(const double* A, const double* B, double* C, bool transB) {  // 8x8
    for (int i = 0; i < 8; i++)
      for (int j = 0; j < 8; j++) {
        double s = 0;
        for (int k = 0; k < 8; k++) s += A[i * 8 + k] * (transB ? B[j * 8 + k] : B[k * 8 + j]);
        C[i * 8 + j] = s;
      }
  }
  // AccumulatedTopHessian.cpp:265-337 (stitchDoubleInternal for pairs [kmin, kmax)) — the copies of all threads are summed in double (:299-308)
  void stitchTopBlocks(bool linearized, MatX& H, VecX& b, int kmin, int kmax) {
    for (int k = kmin; k < kmax; k++) {
      int h = k % nf, t = k / nf;
      int hIdx = 4 + h * 8, tIdx = 4 + t * 8, aidx = h + nf * t;
      double accH[13][13];
      for (int i = 0; i < 13; i++) for (int j = 0; j < 13; j++) accH[i][j] = 0;
      for (AccSet& A : T) {
        AccumulatorApprox& acc = linearized ? A.topL[aidx] : A.topA[aidx];
        acc.finish();
        if (acc.num == 0) continue;
        for (int i = 0; i < 13; i++) for (int j = 0; j < 13; j++) accH[i][j] += acc.h(i, j);
      }
      const double* AH = &adHost[(size_t)aidx * 64];
      const double* AT = &adTarget[(size_t)aidx * 64];
      double A88[64], tmp[64], out[64];
      for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) A88[i * 8 + j] = accH[4 + i][4 + j];
      mul88(AH, A88, tmp, false); mul88(tmp, AH, out, true);
      for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) H(hIdx + i, hIdx + j) += out[i * 8 + j];
      mul88(AT, A88, tmp, false); mul88(tmp, AT, out, true);
      for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) H(tIdx + i, tIdx + j) += out[i * 8 + j];
      mul88(AH, A88, tmp, false); mul88(tmp, AT, out, true);
      for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) H(hIdx + i, tIdx + j) += out[i * 8 + j];
      for (int i = 0; i < 8; i++)
        for (int c = 0; c < 4; c++) {
          double sh = 0, st = 0;
          for (int kk = 0; kk < 8; kk++) { sh += AH[i * 8 + kk] * accH[4 + kk][c]; st += AT[i * 8 + kk] * accH[4 + kk][c]; }
          H(hIdx + i, c) += sh;
          H(tIdx + i, c) += st;
        }
      for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) H(i, j) += accH[i][j];
      for (int i = 0; i < 8; i++) {
        double sh = 0, st = 0;
        for (int kk = 0; kk < 8; kk++) { sh += AH[i * 8 + kk] * accH[4 + kk][12]; st += AT[i * 8 + kk] * accH[4 + kk][12]; }
        b[hIdx + i] += sh;
        b[tIdx + i] += st;
      }
      for (int i = 0; i < 4; i++) b[i] += accH[i][12];
    }
  }
  // stitchDoubleMT (AccumulatedTopHessian.h:95-148): pairs split over the workers into private H/b copies that are then added, or
  // one pass on the calling thread; priors (:324-336) and the symmetrisation (:133-147) afterwards
  void stitchTop(bool linearized, MatX& H, VecX& b, bool usePrior) {
    int n = nf * 8 + 4;
    H = MatX(n, n); b.assign(n, 0.0);
    if (red) {
      std::vector<MatX> Hs(red->nthreads, MatX(n, n));
      std::vector<VecX> bs(red->nthreads, VecX(n, 0.0));
      red->reduce([&](int lo, int hi, double*, int tid) { stitchTopBlocks(linearized, Hs[tid], bs[tid], lo, hi); }, 0, nf * nf, 0);
      for (int t = 0; t < red->nthreads; t++) {
        for (size_t i = 0; i < H.d.size(); i++) H.d[i] += Hs[t].d[i];
        for (int i = 0; i < n; i++) b[i] += bs[t][i];
      }
    } else {
      stitchTopBlocks(linearized, H, b, 0, nf * nf);
    }
    if (usePrior) {
      for (int i = 0; i < 4; i++) { H(i, i) += cPrior[i]; b[i] += cPrior[i] * (double)cDeltaF[i]; }
      for (int h = 0; h < nf; h++)
        for (int i = 0; i < 8; i++) {
          H(4 + h * 8 + i, 4 + h * 8 + i) += frames[h].prior[i];
          b[4 + h * 8 + i] += frames[h].prior[i] * frames[h].delta_prior[i];
        }
    }
    for (int h = 0; h < nf; h++) {
      int hIdx = 4 + h * 8;
      for (int i = 0; i < 8; i++) for (int c = 0; c < 4; c++) H(c, hIdx + i) = H(hIdx + i, c);
      for (int t = h + 1; t < nf; t++) {
        int tIdx = 4 + t * 8;
        for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) H(hIdx + i, tIdx + j) += H(tIdx + j, hIdx + i);
        for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) H(tIdx + j, hIdx + i) = H(hIdx + i, tIdx + j);
      }
    }
  }
  // AccumulatedSCHessian.cpp:106-195 (stitchDoubleInternal for pairs [kmin, kmax)); thread copies summed in double (:136-141, :158-164)
  void stitchSCBlocks(MatX& H, VecX& b, int kmin, int kmax) {
    int nframes2 = nf * nf;
    for (int k0 = kmin; k0 < kmax; k0++) {
      int i = k0 % nf, j = k0 / nf;
      int iIdx = 4 + i * 8, jIdx = 4 + j * 8, ijIdx = i + nf * j;
      double Hpc[8][4], bp[8];
      for (int a = 0; a < 8; a++) { bp[a] = 0; for (int c = 0; c < 4; c++) Hpc[a][c] = 0; }
      for (AccSet& A : T) {
        A.E[ijIdx].finish(); A.EB[ijIdx].finish();
        for (int a = 0; a < 8; a++) { bp[a] += A.EB[ijIdx].a(a); for (int c = 0; c < 4; c++) Hpc[a][c] += A.E[ijIdx].a(a, c); }
      }
      const double* AHij = &adHost[(size_t)ijIdx * 64];
      const double* ATij = &adTarget[(size_t)ijIdx * 64];
      for (int a = 0; a < 8; a++)
        for (int c = 0; c < 4; c++) {
          double sh = 0, st = 0;
          for (int kk = 0; kk < 8; kk++) { sh += AHij[a * 8 + kk] * Hpc[kk][c]; st += ATij[a * 8 + kk] * Hpc[kk][c]; }
          H(iIdx + a, c) += sh;
          H(jIdx + a, c) += st;
        }
      for (int a = 0; a < 8; a++) {
        double sh = 0, st = 0;
        for (int kk = 0; kk < 8; kk++) { sh += AHij[a * 8 + kk] * bp[kk]; st += ATij[a * 8 + kk] * bp[kk]; }
        b[iIdx + a] += sh;
        b[jIdx + a] += st;
      }
      for (int k = 0; k < nf; k++) {
        int kIdx = 4 + k * 8, ijkIdx = ijIdx + k * nframes2, ikIdx = i + nf * k;
        double D[64], tmp[64], out[64];
        for (int a = 0; a < 64; a++) D[a] = 0;
        bool any = false;
        for (AccSet& A : T) {
          A.D[ijkIdx].finish();
          if (A.D[ijkIdx].num == 0) continue;
          any = true;
          for (int a = 0; a < 8; a++) for (int c = 0; c < 8; c++) D[a * 8 + c] += A.D[ijkIdx].a(a, c);
        }
        if (!any) continue;   // (the reference multiplies the zero block through: same result)
        const double* AHik = &adHost[(size_t)ikIdx * 64];
        const double* ATik = &adTarget[(size_t)ikIdx * 64];
        mul88(AHij, D, tmp, false); mul88(tmp, AHik, out, true);
        for (int a = 0; a < 8; a++) for (int c = 0; c < 8; c++) H(iIdx + a, iIdx + c) += out[a * 8 + c];
        mul88(ATij, D, tmp, false); mul88(tmp, ATik, out, true);
        for (int a = 0; a < 8; a++) for (int c = 0; c < 8; c++) H(jIdx + a, kIdx + c) += out[a * 8 + c];
        mul88(ATij, D, tmp, false); mul88(tmp, AHik, out, true);
        for (int a = 0; a < 8; a++) for (int c = 0; c < 8; c++) H(jIdx + a, iIdx + c) += out[a * 8 + c];
        mul88(AHij, D, tmp, false); mul88(tmp, ATik, out, true);
        for (int a = 0; a < 8; a++) for (int c = 0; c < 8; c++) H(iIdx + a, kIdx + c) += out[a * 8 + c];
      }
    }
    if (kmin == 0 && kmax > 0) {
      for (AccSet& A : T) {
        A.Hcc.finish(); A.bc.finish();
        for (int a = 0; a < 4; a++) { for (int c = 0; c < 4; c++) H(a, c) += A.Hcc.a(a, c); b[a] += A.bc.a(a); }
      }
    }
  }
  // stitchDoubleMT (AccumulatedSCHessian.h:96-135)
  void stitchSC(MatX& H, VecX& b) {
    int n = nf * 8 + 4;
    H = MatX(n, n); b.assign(n, 0.0);
    if (red) {
      std::vector<MatX> Hs(red->nthreads, MatX(n, n));
      std::vector<VecX> bs(red->nthreads, VecX(n, 0.0));
      red->reduce([&](int lo, int hi, double*, int tid) { stitchSCBlocks(Hs[tid], bs[tid], lo, hi); }, 0, nf * nf, 0);
      for (int t = 0; t < red->nthreads; t++) {
        for (size_t i = 0; i < H.d.size(); i++) H.d[i] += Hs[t].d[i];
        for (int i = 0; i < n; i++) b[i] += bs[t][i];
      }
    } else {
      stitchSCBlocks(H, b, 0, nf * nf);
    }
    for (int h = 0; h < nf; h++) {
      int hIdx = 4 + h * 8;
      for (int a = 0; a < 8; a++) for (int c = 0; c < 4; c++) H(c, hIdx + a) = H(hIdx + a, c);
    }
  }

  // ---------------------------------------------------------------- solve
  void getNullspaces() {  // FullSystemOptimize.cpp:1087-1147
    int n = 4 + nf * 8;
    lastNullspaces_pose.clear(); lastNullspaces_scale.clear();
    for (int i = 0; i < 6; i++) {
      VecX ns(n, 0.0);
      for (int f = 0; f < nf; f++) {
        for (int r = 0; r < 6; r++) ns[4 + f * 8 + r] = frames[f].nullspaces_pose[r][i];
        for (int r = 0; r < 3; r++) ns[4 + f * 8 + r] *= SCALE_XI_TRANS_INVERSE;
        for (int r = 3; r < 6; r++) ns[4 + f * 8 + r] *= SCALE_XI_ROT_INVERSE;
      }
      lastNullspaces_pose.push_back(ns);
    }
    VecX ns(n, 0.0);
    for (int f = 0; f < nf; f++) {
      for (int r = 0; r < 6; r++) ns[4 + f * 8 + r] = frames[f].nullspaces_scale[r];
      for (int r = 0; r < 3; r++) ns[4 + f * 8 + r] *= SCALE_XI_TRANS_INVERSE;
      for (int r = 3; r < 6; r++) ns[4 + f * 8 + r] *= SCALE_XI_ROT_INVERSE;
    }
    lastNullspaces_scale.push_back(ns);
  }
  void orthogonalize(VecX* b, MatX* H) {  // EnergyFunctional.cpp:775-835
    std::vector<VecX> ns;
    ns.insert(ns.end(), lastNullspaces_pose.begin(), lastNullspaces_pose.end());
    ns.insert(ns.end(), lastNullspaces_scale.begin(), lastNullspaces_scale.end());
    int dim = (int)ns[0].size(), m = (int)ns.size();
    MatX N(dim, m);
    for (int i = 0; i < m; i++) {
      double nrm = 0;
      for (int k = 0; k < dim; k++) nrm += ns[i][k] * ns[i][k];
      nrm = std::sqrt(nrm);
      for (int k = 0; k < dim; k++) N(k, i) = ns[i][k] / nrm;
    }
    MatX P;
    span_projector(N, setting_solverModeDelta, P);  // = 0.5*(NNpiT + NNpiT^T)
    if (b) {
      VecX Pb(dim, 0.0);
      for (int i = 0; i < dim; i++) { double s = 0; for (int k = 0; k < dim; k++) s += P(i, k) * (*b)[k]; Pb[i] = s; }
      for (int i = 0; i < dim; i++) (*b)[i] -= Pb[i];
    }
    if (H) {
      MatX PH(dim, dim), PHP(dim, dim);
      for (int i = 0; i < dim; i++) for (int j = 0; j < dim; j++) { double s = 0; for (int k = 0; k < dim; k++) s += P(i, k) * (*H)(k, j); PH(i, j) = s; }
      for (int i = 0; i < dim; i++) for (int j = 0; j < dim; j++) { double s = 0; for (int k = 0; k < dim; k++) s += PH(i, k) * P(k, j); PHP(i, j) = s; }
      for (int i = 0; i < dim; i++) for (int j = 0; j < dim; j++) (*H)(i, j) -= PHP(i, j);
    }
  }
  VecX getStitchedDeltaF() const {  // EnergyFunctional.cpp:1021-1032
    VecX d(4 + nf * 8);
    for (int i = 0; i < 4; i++) d[i] = (double)cDeltaF[i];
    for (int h = 0; h < nf; h++) for (int i = 0; i < 8; i++) d[4 + 8 * h + i] = frames[h].delta[i];
    return d;
  }
  // EnergyFunctional.cpp:272-341
  void resubstituteF(const VecX& x) {
    int n = 4 + nf * 8;
    std::vector<float> xF(n);
    for (int i = 0; i < n; i++) xF[i] = (float)x[i];
    for (int i = 0; i < 4; i++) HCalib.step[i] = -x[i];
    std::vector<float> xAd((size_t)nf * nf * 8);
    for (int h = 0; h < nf; h++) {
      for (int i = 0; i < 8; i++) frames[h].step[i] = -x[4 + 8 * h + i];
      frames[h].step[8] = frames[h].step[9] = 0;
      for (int t = 0; t < nf; t++)
        for (int j = 0; j < 8; j++) {
          float sh = 0, st = 0;
          for (int i = 0; i < 8; i++) {
            sh += xF[4 + 8 * h + i] * adHostF[(size_t)(h + nf * t) * 64 + i * 8 + j];
            st += xF[4 + 8 * t + i] * adTargetF[(size_t)(h + nf * t) * 64 + i * 8 + j];
          }
          xAd[(size_t)(nf * h + t) * 8 + j] = sh + st;
        }
    }
    const float* xc = xF.data();
    auto perPoint = [&](int lo, int hi, double*, int) { for (int pi = lo; pi < hi; pi++) {
      Point& p = points[pi];
      int ngoodres = 0;
      for (int ri = p.rbeg; ri < p.rend; ri++) if (res[ri].isActive) ngoodres++;
      if (ngoodres == 0) { p.step = 0; continue; }
      float b = p.bdSumF;
      float hs[4];
      for (int i = 0; i < 4; i++) hs[i] = p.Hcd_accAF[i] + p.Hcd_accLF[i];
      b -= dot4(xc, hs);
      for (int ri = p.rbeg; ri < p.rend; ri++) {
        Residual& r = res[ri];
        if (!r.isActive) continue;
        const float* xa = &xAd[(size_t)(r.host * nf + r.target) * 8];
        float s = 0;
        for (int i = 0; i < 8; i++) s += xa[i] * r.JpJdF[i];
        b -= s;
      }
      p.step = -b * p.HdiF;
    } };
    if (red) red->reduce(perPoint, 0, np, 50);   // resubstituteF_MT (:294-295)
    else perPoint(0, np, nullptr, 0);
  }
  // EnergyFunctional.cpp:838-995
  void solveSystemF(int iteration, double lambda) {
    if (solverMode & SOLVER_USE_GN) lambda = 0;
    if (solverMode & SOLVER_FIX_LAMBDA) lambda = 1e-5;
    int n = 4 + nf * 8;
    accumulateAll();
    MatX HL_top, HA_top, H_sc;
    VecX bL_top, bA_top, b_sc;
    stitchTop(false, HA_top, bA_top, false);
    stitchTop(true, HL_top, bL_top, true);
    stitchSC(H_sc, b_sc);
    VecX delta = getStitchedDeltaF();
    VecX bM_top(n);
    for (int i = 0; i < n; i++) { double s = 0; for (int k = 0; k < n; k++) s += HM(i, k) * delta[k]; bM_top[i] = bM[i] + s; }
    MatX HFinal_top(n, n);
    VecX bFinal_top(n);
    if (solverMode & SOLVER_ORTHOGONALIZE_SYSTEM) {
      bool haveFirstFrame = false;
      for (Frame& f : frames) if (f.frameID == 0) haveFirstFrame = true;
      MatX HT_act(n, n); VecX bT_act(n);
      for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) HT_act(i, j) = HL_top(i, j) + HA_top(i, j) - H_sc(i, j); bT_act[i] = bL_top[i] + bA_top[i] - b_sc[i]; }
      if (!haveFirstFrame) orthogonalize(&bT_act, &HT_act);
      for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) HFinal_top(i, j) = HT_act(i, j) + HM(i, j); bFinal_top[i] = bT_act[i] + bM_top[i]; }
      lastHS = HFinal_top; lastbS = bFinal_top;
      for (int i = 0; i < n; i++) HFinal_top(i, i) *= (1 + lambda);
    } else {
      for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) HFinal_top(i, j) = HL_top(i, j) + HM(i, j) + HA_top(i, j);
        bFinal_top[i] = bL_top[i] + bM_top[i] + bA_top[i] - b_sc[i];
      }
      lastHS = MatX(n, n);
      for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) lastHS(i, j) = HFinal_top(i, j) - H_sc(i, j);
      lastbS = bFinal_top;
      for (int i = 0; i < n; i++) HFinal_top(i, i) *= (1 + lambda);
      double f = (double)(1.0f / (1 + lambda));
      for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) HFinal_top(i, j) -= H_sc(i, j) * f;
    }
    VecX x;
    if (solverMode & SOLVER_SVD) {  // :924-965.  JacobiSVD of the symmetric scaled system = its eigen-decomposition: S = |w| sorted
      VecX SVecI(n);                // decreasing, U_i = sign(w_i) V_i (Eigen is external and unpinned: restated, not copied)
      for (int i = 0; i < n; i++) SVecI[i] = 1.0 / std::sqrt(HFinal_top(i, i));
      MatX HFinalScaled(n, n);
      VecX bs(n);
      for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) HFinalScaled(i, j) = SVecI[i] * HFinal_top(i, j) * SVecI[j]; bs[i] = SVecI[i] * bFinal_top[i]; }
      VecX w; MatX V;
      sym_eigen(HFinalScaled, w, V);
      std::vector<int> ord(n);
      for (int i = 0; i < n; i++) ord[i] = i;
      std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return std::fabs(w[a]) > std::fabs(w[b]); });
      double maxSv = 0;
      for (int i = 0; i < n; i++) maxSv = std::max(maxSv, std::fabs(w[i]));
      VecX Ub(n, 0.0);
      for (int i = 0; i < n; i++) {
        const int c = ord[i];
        const double S = std::fabs(w[c]);
        double ub = 0;
        for (int k = 0; k < n; k++) ub += V(k, c) * bs[k];
        if (w[c] < 0) ub = -ub;
        if (S < setting_solverModeDelta * maxSv) ub = 0;
        if ((solverMode & SOLVER_SVD_CUT7) && (i >= n - 7)) ub = 0;
        else ub /= S;
        Ub[i] = ub;
      }
      x.assign(n, 0.0);
      for (int i = 0; i < n; i++) {
        const int c = ord[i];
        for (int k = 0; k < n; k++) x[k] += V(k, c) * Ub[i];
      }
      for (int k = 0; k < n; k++) x[k] *= SVecI[k];
    } else {  // LDLT branch (:966-977)
      VecX SVecI(n);
      for (int i = 0; i < n; i++) SVecI[i] = 1.0 / std::sqrt(HFinal_top(i, i) + 10);
      MatX HFinalScaled(n, n);
      VecX bs(n);
      for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) HFinalScaled(i, j) = SVecI[i] * HFinal_top(i, j) * SVecI[j]; bs[i] = SVecI[i] * bFinal_top[i]; }
      VecX y;
      ldlt_solve(HFinalScaled, bs, y);
      x.resize(n);
      for (int i = 0; i < n; i++) x[i] = SVecI[i] * y[i];
    }
    if ((solverMode & SOLVER_ORTHOGONALIZE_X) || (iteration >= 2 && (solverMode & SOLVER_ORTHOGONALIZE_X_LATER))) orthogonalize(&x, 0);
    lastX = x;
    resubstituteF(x);
  }

  // ---------------------------------------------------------------- energies (EnergyFunctional.cpp:344-442)
  double calcMEnergyF() {
    VecX delta = getStitchedDeltaF();
    int n = (int)delta.size();
    double e = 0;
    for (int i = 0; i < n; i++) { double s = 0; for (int k = 0; k < n; k++) s += HM(i, k) * delta[k]; e += delta[i] * (2 * bM[i] + s); }
    return e;
  }
  double calcLEnergyF() {
    double E = 0;
    for (Frame& f : frames) for (int i = 0; i < 8; i++) E += f.delta_prior[i] * f.prior[i] * f.delta_prior[i];
    { float s = 0; for (int i = 0; i < 4; i++) s += cDeltaF[i] * cPriorF[i] * cDeltaF[i]; E += s; }
    float Ept = 0;  // single Accumulator11 over all points (one thread)
    for (Point& p : points) {
      float dd = p.deltaF;
      for (int ri = p.rbeg; ri < p.rend; ri++) {
        Residual& r = res[ri];
        if (!r.isLinearized || !r.isActive) continue;
        const float* dp = &adHTdeltaF[(size_t)(r.host + nf * r.target) * 8];
        const RawJ* rJ = &r.Jef;
        float dx = dot6(rJ->Jpdxi[0], dp) + dot4(rJ->Jpdc[0], cDeltaF) + rJ->Jpdd[0] * dd;
        float dy = dot6(rJ->Jpdxi[1], dp) + dot4(rJ->Jpdc[1], cDeltaF) + rJ->Jpdd[1] * dd;
        for (int i = 0; i < 8; i++) {
          float Jdelta = rJ->JIdx[0][i] * dx;
          Jdelta = Jdelta + rJ->JIdx[1][i] * dy;
          Jdelta = Jdelta + rJ->JabF[0][i] * dp[6];
          Jdelta = Jdelta + rJ->JabF[1][i] * dp[7];
          float r0 = r.res_toZeroF[i];
          r0 = r0 + r0;
          r0 = r0 + Jdelta;
          Ept += Jdelta * r0;
        }
      }
      Ept += p.deltaF * p.deltaF * p.priorF;
    }
    return E + Ept;
  }

  // ---------------------------------------------------------------- GN driver (FullSystemOptimize.cpp)
  void backupState(bool backupLastStep) {  // :309-351
    if (solverMode & SOLVER_MOMENTUM) {    // :311-345: the previous iteration's steps are kept (zeros before the first one)
      for (int i = 0; i < 4; i++) { HCalib.step_backup[i] = backupLastStep ? HCalib.step[i] : 0.0; HCalib.value_backup[i] = HCalib.value[i]; }
      for (Frame& f : frames) for (int i = 0; i < 10; i++) { f.step_backup[i] = backupLastStep ? f.step[i] : 0.0; f.state_backup[i] = f.state[i]; }
      for (Point& p : points) { p.idepth_backup = p.idepth; p.step_backup = backupLastStep ? p.step : 0.f; }
      return;
    }
    for (int i = 0; i < 4; i++) HCalib.value_backup[i] = HCalib.value[i];
    for (Frame& f : frames) for (int i = 0; i < 10; i++) f.state_backup[i] = f.state[i];
    for (Point& p : points) p.idepth_backup = p.idepth;
  }
  bool doStepFromBackup(float stepfacC, float stepfacT, float stepfacR, float stepfacA, float stepfacD) {  // :207-305
    double pstepfac[10];
    for (int i = 0; i < 3; i++) pstepfac[i] = stepfacT;
    for (int i = 3; i < 6; i++) pstepfac[i] = stepfacR;
    for (int i = 6; i < 10; i++) pstepfac[i] = stepfacA;
    float sumA = 0, sumB = 0, sumT = 0, sumR = 0, sumID = 0, numID = 0, sumNID = 0;
    double nv[4];
    if (solverMode & SOLVER_MOMENTUM) {   // :225-251: the whole new step plus half of the previous one (poses and points), no step factors
      for (int i = 0; i < 4; i++) nv[i] = HCalib.value_backup[i] + HCalib.step[i];
      HCalib.setValue(nv);
      for (int fi = 0; fi < nf; fi++) {
        Frame& fh = frames[fi];
        double step[10], ns[10];
        for (int i = 0; i < 10; i++) step[i] = fh.step[i];
        for (int i = 0; i < 6; i++) step[i] += 0.5f * fh.step_backup[i];
        for (int i = 0; i < 10; i++) ns[i] = fh.state_backup[i] + step[i];
        fh.setState(ns);
        sumA += step[6] * step[6];
        sumB += step[7] * step[7];
        sumT += step[0] * step[0] + step[1] * step[1] + step[2] * step[2];
        sumR += step[3] * step[3] + step[4] * step[4] + step[5] * step[5];
        for (Point& ph : points) {
          if (ph.host != fi) continue;
          const float pstep = ph.step + 0.5f * (ph.step_backup);
          ph.setIdepth(ph.idepth_backup + pstep);
          sumID += pstep * pstep;
          sumNID += fabsf(ph.idepth_backup);
          numID++;
          ph.setIdepthZero(ph.idepth_backup + pstep);
        }
      }
      sumA /= nf; sumB /= nf; sumR /= nf; sumT /= nf;
      sumID /= numID; sumNID /= numID;
      setPrecalcValues();
      return sqrtf(sumA) < 0.0005 * setting_thOptIterations && sqrtf(sumB) < 0.00005 * setting_thOptIterations &&
             sqrtf(sumR) < 0.00005 * setting_thOptIterations && sqrtf(sumT) * sumNID < 0.00005 * setting_thOptIterations;
    }
    for (int i = 0; i < 4; i++) nv[i] = HCalib.value_backup[i] + stepfacC * HCalib.step[i];
    HCalib.setValue(nv);
    for (int fi = 0; fi < nf; fi++) {
      Frame& fh = frames[fi];
      double ns[10];
      for (int i = 0; i < 10; i++) ns[i] = fh.state_backup[i] + pstepfac[i] * fh.step[i];
      fh.setState(ns);
      sumA += fh.step[6] * fh.step[6];
      sumB += fh.step[7] * fh.step[7];
      sumT += fh.step[0] * fh.step[0] + fh.step[1] * fh.step[1] + fh.step[2] * fh.step[2];
      sumR += fh.step[3] * fh.step[3] + fh.step[4] * fh.step[4] + fh.step[5] * fh.step[5];
      for (Point& ph : points) {
        if (ph.host != fi) continue;
        ph.setIdepth(ph.idepth_backup + stepfacD * ph.step);
        sumID += ph.step * ph.step;
        sumNID += fabsf(ph.idepth_backup);
        numID++;
        ph.setIdepthZero(ph.idepth_backup + stepfacD * ph.step);
      }
    }
    sumA /= nf; sumB /= nf; sumR /= nf; sumT /= nf;
    sumID /= numID; sumNID /= numID;
    setPrecalcValues();
    return sqrtf(sumA) < 0.0005 * setting_thOptIterations && sqrtf(sumB) < 0.00005 * setting_thOptIterations &&
           sqrtf(sumR) < 0.00005 * setting_thOptIterations && sqrtf(sumT) * sumNID < 0.00005 * setting_thOptIterations;
  }
  void loadStateBackup() {  // :355-370
    HCalib.setValue(HCalib.value_backup);
    for (Frame& f : frames) f.setState(f.state_backup);
    for (Point& p : points) { p.setIdepth(p.idepth_backup); p.setIdepthZero(p.idepth_backup); }
    setPrecalcValues();
  }
  double calcLEnergy() { return forceAcceptStep ? 0 : calcLEnergyF(); }
  double calcMEnergy() { return forceAcceptStep ? 0 : calcMEnergyF(); }
  void applyAll() {   // applyRes_Reductor (FullSystemOptimize.cpp:90-96), chunks of 50
    auto f = [&](int lo, int hi, double*, int) { for (int k = lo; k < hi; k++) if (!res[k].isLinearized) applyRes(res[k]); };
    if (red) red->reduce(f, 0, nr, 50); else f(0, nr, nullptr, 0);
  }

  float optimize(int mnumOptIts, orc_ba_opt_result_t* out) {  // :871-1041
    out->iterations = 0; out->lastEnergy = 0; out->rmse = 0; out->resInA = 0;
    if (nf < 2) return 0;
    if (nf < 3) mnumOptIts = 20;
    if (nf < 4) mnumOptIts = 15;
    for (Residual& r : res) if (!r.isLinearized) r.resetOOB();
    xTrace.clear();
    double lastEnergy = linearizeAll(false);
    double lastEnergyL = calcLEnergy();
    double lastEnergyM = calcMEnergy();
    applyAll();
    double lambda = 1e-1;
    float stepsize = 1;
    const int nx = 4 + 8 * nf;
    VecX previousX(nx, std::numeric_limits<double>::quiet_NaN());
    stepTrace.clear();
    for (int iteration = 0; iteration < mnumOptIts; iteration++) {
      out->iterations++;
      backupState(iteration != 0);
      getNullspaces();
      solveSystemF(iteration, lambda);
      xTrace.push_back(lastX);
      {  // :933-948
        double dot = 0, n0 = 0, n1 = 0;
        for (int i = 0; i < nx; i++) { dot += previousX[i] * lastX[i]; n0 += previousX[i] * previousX[i]; n1 += lastX[i] * lastX[i]; }
        const double incDirChange = (1e-20 + dot) / (1e-20 + std::sqrt(n0) * std::sqrt(n1));
        previousX = lastX;
        if (std::isfinite(incDirChange) && (solverMode & SOLVER_STEPMOMENTUM)) {
          float newStepsize = exp(incDirChange * 1.4);
          if (incDirChange < 0 && stepsize > 1) stepsize = 1;
          stepsize = sqrtf(sqrtf(newStepsize * stepsize * stepsize * stepsize));
          if (stepsize > 2) stepsize = 2;
          if (stepsize < 0.25) stepsize = 0.25;
        }
        stepTrace.push_back(stepsize);
      }
      bool canbreak = doStepFromBackup(stepsize, stepsize, stepsize, stepsize, stepsize);
      double newEnergy = linearizeAll(false);
      double newEnergyL = calcLEnergy();
      double newEnergyM = calcMEnergy();
      if (forceAcceptStep || (newEnergy + newEnergyL + newEnergyM < lastEnergy + lastEnergyL + lastEnergyM)) {
        applyAll();
        lastEnergy = newEnergy; lastEnergyL = newEnergyL; lastEnergyM = newEnergyM;
        lambda *= 0.25;
      } else {
        loadStateBackup();
        lastEnergy = linearizeAll(false);
        lastEnergyL = calcLEnergy();
        lastEnergyM = calcMEnergy();
        lambda *= 1e2;
      }
      if (canbreak && iteration >= setting_minOptIterations) break;
    }
    double newStateZero[10] = {0};
    newStateZero[6] = frames[nf - 1].state[6];
    newStateZero[7] = frames[nf - 1].state[7];
    frames[nf - 1].setEvalPT(frames[nf - 1].PRE_worldToCam, newStateZero);
    setAdjointsF();
    setPrecalcValues();
    lastEnergy = linearizeAll(true);
    out->lastEnergy = lastEnergy;
    out->resInA = nresA;
    out->rmse = sqrtf((float)(lastEnergy / (patternNum * nresA)));
    lastResult = *out;
    return (float)out->rmse;
  }

  // ---------------------------------------------------------------- marginalisation of points
  // FullSystem.cpp:1004-1021 (per flagged point) then EnergyFunctional.cpp:663-736.
  void marginalizePoints(const uint8_t* marg_flag) {
    for (int pi = 0; pi < np; pi++) {
      if (!marg_flag[pi]) continue;
      Point& p = points[pi];
      for (int ri = p.rbeg; ri < p.rend; ri++) {
        Residual& r = res[ri];
        r.resetOOB();
        linearize(r);
        r.isLinearized = false;
        applyRes(r);
        if (r.isActive) fixLinearizationF(r);
      }
    }
    zeroAll(); nresA = 0;
    for (int pi = 0; pi < np; pi++) {
      if (!marg_flag[pi]) continue;
      Point& p = points[pi];
      p.priorF *= setting_idepthFixPriorMargFac;
      addPointTop<2>(p, T[0].topA, nresA);
      addPointSC(p, false, T[0]);
    }
    // stitchDouble (single-thread variants, AccumulatedTopHessian.cpp:201-262 / AccumulatedSCHessian.cpp:198-256):
    // identical block arithmetic to stitchTop/stitchSC above with usePrior=false; the SC variant
    // assigns (not adds) Hcc/bc, which is the same on a zeroed matrix.
    MatX M, Msc; VecX Mb, Mbsc;
    stitchTop(false, M, Mb, false);
    stitchSC(Msc, Mbsc);
    resInM += nresA;
    int n = 4 + nf * 8;
    MatX H(n, n); VecX b(n);
    for (int i = 0; i < n; i++) {
      for (int j = 0; j < n; j++) H(i, j) = M(i, j) - Msc(i, j);
      b[i] = Mb[i] - Mbsc[i];
    }
    if (solverMode & (SOLVER_ORTHOGONALIZE_POINTMARG | SOLVER_ORTHOGONALIZE_FULL)) getNullspaces();   // FullSystem.cpp:1453-1456, right before marginalizePointsF
    if (solverMode & SOLVER_ORTHOGONALIZE_POINTMARG) {   // EnergyFunctional.cpp:711-723
      bool haveFirstFrame = false;
      for (Frame& f : frames) if (f.frameID == 0) haveFirstFrame = true;
      if (!haveFirstFrame) orthogonalize(&b, &H);
    }
    for (int i = 0; i < n; i++) {
      for (int j = 0; j < n; j++) HM(i, j) += setting_margWeightFac * H(i, j);
      bM[i] += setting_margWeightFac * b[i];
    }
    if (solverMode & SOLVER_ORTHOGONALIZE_FULL) orthogonalize(&bM, &HM);   // :730-731
  }
};

// =================================================================== C API
extern "C" orc_ba* orc_ba_create(const orc_ba_window_t* W) {
  orc_ba* h = new orc_ba();
  h->nf = W->nf; h->np = W->np; h->nr = W->nr; h->w = W->w; h->h = W->h;
  h->wM3G = W->w - 3; h->hM3G = W->h - 3;
  h->solverMode = W->solverMode;
  h->affineOptModeA = W->affineOptModeA; h->affineOptModeB = W->affineOptModeB;
  h->forceAcceptStep = W->forceAcceptStep != 0;
  for (int i = 0; i < 4; i++) h->HCalib.value_zero[i] = W->calib_value_zero[i];
  h->HCalib.setValueScaled(W->calib_value_scaled);
  for (int i = 0; i < 4; i++) h->HCalib.step[i] = 0;
  h->frames.resize(h->nf);
  for (int f = 0; f < h->nf; f++) {
    Frame& F = h->frames[f];
    std::memcpy(F.worldToCam_evalPT.R, W->evalPT + f * 12, 9 * sizeof(double));
    std::memcpy(F.worldToCam_evalPT.t, W->evalPT + f * 12 + 9, 3 * sizeof(double));
    F.ab_exposure = W->ab_exposure[f];
    F.frameEnergyTH = W->frameEnergyTH[f];
    F.frameID = W->frameID[f];
    F.dI = W->dI ? W->dI[f] : nullptr;
    F.setState(W->state + f * 10);
    F.setStateZero(W->state_zero + f * 10);
    for (int i = 0; i < 10; i++) F.step[i] = 0;
    double p10[10];
    F.getPrior(p10, h->affineOptModeA, h->affineOptModeB, h->solverMode);
    for (int i = 0; i < 8; i++) F.prior[i] = p10[i];
  }
  h->points.resize(h->np);
  for (int i = 0; i < h->np; i++) {
    Point& p = h->points[i];
    p.u = W->u[i]; p.v = W->v[i];
    p.setIdepth(W->idepth[i]); p.setIdepthZero(W->idepth_zero[i]);
    for (int k = 0; k < 8; k++) { p.color[k] = W->color[i * 8 + k]; p.weights[k] = W->weights[i * 8 + k]; }
    p.host = W->host[i];
    p.hasDepthPrior = W->hasDepthPrior[i] != 0;
    p.step = 0; p.idepth_backup = p.idepth; p.idepth_hessian = 0;
    p.maxRelBaseline = W->maxRelBaseline ? W->maxRelBaseline[i] : 0.f;
    p.numGoodResiduals = W->numGoodResiduals ? W->numGoodResiduals[i] : 0;
    p.rbeg = p.rend = 0;
    p.priorF = p.hasDepthPrior ? setting_idepthFixPrior * SCALE_IDEPTH * SCALE_IDEPTH : 0;  // EFPoint::takeData
    if (h->solverMode & SOLVER_REMOVE_POSEPRIOR) p.priorF = 0;
    p.deltaF = p.idepth - p.idepth_zero;
    p.bdSumF = p.HdiF = 0;
    p.Hdd_accLF = p.bd_accLF = p.Hdd_accAF = p.bd_accAF = 0;
    for (int k = 0; k < 4; k++) p.Hcd_accLF[k] = p.Hcd_accAF[k] = 0;
  }
  h->res.resize(h->nr);
  int prev = -1;
  for (int i = 0; i < h->nr; i++) {
    Residual& r = h->res[i];
    std::memset(&r, 0, sizeof(r));
    r.point = W->res_point[i];
    r.host = h->points[r.point].host;
    r.target = W->res_target[i];
    r.state_state = (ResState)W->res_state[i];
    r.state_NewState = RS_OUTLIER;
    r.state_NewEnergyWithOutlier = -1;
    r.isLinearized = false; r.isActive = false;
    r.isNew = W->res_isNew ? W->res_isNew[i] != 0 : true; r.toRemove = false;
    if (r.point != prev) { h->points[r.point].rbeg = i; prev = r.point; }
    h->points[r.point].rend = i + 1;
  }
  int n = 4 + 8 * h->nf;
  h->HM = MatX(n, n); h->bM.assign(n, 0.0);
  if (W->HM) for (int i = 0; i < n * n; i++) h->HM.d[i] = W->HM[i];
  if (W->bM) for (int i = 0; i < n; i++) h->bM[i] = W->bM[i];
  h->precalc.resize((size_t)h->nf * h->nf);
  h->setAdjointsF();
  h->setPrecalcValues();
  h->nresA = h->nresL = h->resInM = 0;
  h->zeroAll();
  h->getNullspaces();
  return h;
}
extern "C" void orc_ba_destroy(orc_ba* h) { delete h; }
// CPU-baseline threading (IndexThreadReduce model): n <= 1 restores the single-thread path.  The packed-accumulator getters read
// thread 0's copy only, so they are meaningful on the single-thread path (the one the parity tests use).
extern "C" int orc_ba_set_threads(orc_ba* h, int n) {
  delete h->red;
  h->red = nullptr;
  h->T.assign(n > 1 ? n : 1, AccSet());
  h->zeroAll();
  if (n > 1) h->red = new Reducer(n);
  return h->red ? h->red->pinned : 0;     // workers bound to a core of their own (timed-baseline hygiene)
}
extern "C" int orc_ba_linearize(orc_ba* h, double* energy) { double e = h->linearizeAll(false); if (energy) *energy = e; return 0; }
extern "C" int orc_ba_get_linearization(orc_ba* h, float* J, uint8_t* newState, float* newEnergy, float* newEnergyWithOutlier,
                                        float* projectedTo, float* centerProjectedTo) {
  for (int i = 0; i < h->nr; i++) {
    const Residual& r = h->res[i];
    if (J) std::memcpy(J + (size_t)i * 74, &r.Jnew, sizeof(RawJ));
    if (newState) newState[i] = (uint8_t)r.state_NewState;
    if (newEnergy) newEnergy[i] = (float)r.state_NewEnergy;
    if (newEnergyWithOutlier) newEnergyWithOutlier[i] = (float)r.state_NewEnergyWithOutlier;
    if (projectedTo) std::memcpy(projectedTo + (size_t)i * 16, r.projectedTo, 16 * sizeof(float));
    if (centerProjectedTo) std::memcpy(centerProjectedTo + (size_t)i * 3, r.centerProjectedTo, 3 * sizeof(float));
  }
  return 0;
}
extern "C" int orc_ba_get_ef_jacobians(orc_ba* h, float* J) {   // EFResidual::J
  for (int i = 0; i < h->nr; i++) std::memcpy(J + (size_t)i * 74, &h->res[i].Jef, sizeof(RawJ));
  return 0;
}
extern "C" int orc_ba_apply_res(orc_ba* h) { h->applyAll(); return 0; }
extern "C" int orc_ba_get_residual_state(orc_ba* h, uint8_t* state, uint8_t* isActive, float* JpJdF) {
  for (int i = 0; i < h->nr; i++) {
    if (state) state[i] = (uint8_t)h->res[i].state_state;
    if (isActive) isActive[i] = h->res[i].isActive ? 1 : 0;
    if (JpJdF) std::memcpy(JpJdF + (size_t)i * 8, h->res[i].JpJdF, 8 * sizeof(float));
  }
  return 0;
}
extern "C" int orc_ba_accumulate(orc_ba* h) { h->accumulateAll(); return 0; }
extern "C" int orc_ba_accum_floats(int nf) { return nf * nf * 91 * 2 + nf * nf * nf * 64 + nf * nf * 32 + nf * nf * 8 + 16 + 4 + 2; }

static void packTop(AccumulatorApprox& a, float* out) {
  a.finish();
  int k = 0;
  for (int r = 0; r < 10; r++) for (int c = r; c < 10; c++) out[k++] = a.H[r][c];
  for (int r = 0; r < 10; r++) for (int c = 0; c < 3; c++) out[k++] = a.H[r][10 + c];
  out[k++] = a.H[10][10]; out[k++] = a.H[10][11]; out[k++] = a.H[10][12];
  out[k++] = a.H[11][11]; out[k++] = a.H[11][12]; out[k++] = a.H[12][12];
}
extern "C" int orc_ba_get_accumulators(orc_ba* h, float* packed) {
  int nf = h->nf;
  float* p = packed;
  for (int i = 0; i < nf * nf; i++, p += 91) packTop(h->T[0].topA[i], p);
  for (int i = 0; i < nf * nf; i++, p += 91) packTop(h->T[0].topL[i], p);
  for (int i = 0; i < nf * nf * nf; i++) { h->T[0].D[i].finish(); for (int a = 0; a < 8; a++) for (int c = 0; c < 8; c++) *p++ = h->T[0].D[i].A1m[a][c]; }
  for (int i = 0; i < nf * nf; i++) { h->T[0].E[i].finish(); for (int a = 0; a < 8; a++) for (int c = 0; c < 4; c++) *p++ = h->T[0].E[i].A1m[a][c]; }
  for (int i = 0; i < nf * nf; i++) { h->T[0].EB[i].finish(); for (int a = 0; a < 8; a++) *p++ = h->T[0].EB[i].A1m[a]; }
  h->T[0].Hcc.finish(); h->T[0].bc.finish();
  for (int a = 0; a < 4; a++) for (int c = 0; c < 4; c++) *p++ = h->T[0].Hcc.A1m[a][c];
  for (int a = 0; a < 4; a++) *p++ = h->T[0].bc.A1m[a];
  *p++ = (float)h->nresA; *p++ = (float)h->nresL;
  return 0;
}
// truth mode (orc_acc.h): the same packed layout in double, read through the accumulators' h() / a()
extern "C" void orc_set_acc64(int on) { acc64_mode() = on; }
extern "C" int orc_ba_get_accumulators_f64(orc_ba* h, double* packed) {
  int nf = h->nf;
  double* p = packed;
  for (int which = 0; which < 2; which++)
    for (int i = 0; i < nf * nf; i++) {
      AccumulatorApprox& a = which ? h->T[0].topL[i] : h->T[0].topA[i];
      a.finish();
      for (int r = 0; r < 10; r++) for (int c = r; c < 10; c++) *p++ = a.h(r, c);
      for (int r = 0; r < 10; r++) for (int c = 0; c < 3; c++) *p++ = a.h(r, 10 + c);
      *p++ = a.h(10, 10); *p++ = a.h(10, 11); *p++ = a.h(10, 12); *p++ = a.h(11, 11); *p++ = a.h(11, 12); *p++ = a.h(12, 12);
    }
  for (int i = 0; i < nf * nf * nf; i++) { h->T[0].D[i].finish(); for (int a = 0; a < 8; a++) for (int c = 0; c < 8; c++) *p++ = h->T[0].D[i].a(a, c); }
  for (int i = 0; i < nf * nf; i++) { h->T[0].E[i].finish(); for (int a = 0; a < 8; a++) for (int c = 0; c < 4; c++) *p++ = h->T[0].E[i].a(a, c); }
  for (int i = 0; i < nf * nf; i++) { h->T[0].EB[i].finish(); for (int a = 0; a < 8; a++) *p++ = h->T[0].EB[i].a(a); }
  h->T[0].Hcc.finish(); h->T[0].bc.finish();
  for (int a = 0; a < 4; a++) for (int c = 0; c < 4; c++) *p++ = h->T[0].Hcc.a(a, c);
  for (int a = 0; a < 4; a++) *p++ = h->T[0].bc.a(a);
  *p++ = (double)h->nresA; *p++ = (double)h->nresL;
  return 0;
}
extern "C" int orc_ba_get_point_terms(orc_ba* h, float* HdiF, float* bdSumF, float* Hdd_accAF, float* bd_accAF, float* Hcd_accAF) {
  for (int i = 0; i < h->np; i++) {
    const Point& p = h->points[i];
    if (HdiF) HdiF[i] = p.HdiF;
    if (bdSumF) bdSumF[i] = p.bdSumF;
    if (Hdd_accAF) Hdd_accAF[i] = p.Hdd_accAF;
    if (bd_accAF) bd_accAF[i] = p.bd_accAF;
    if (Hcd_accAF) for (int k = 0; k < 4; k++) Hcd_accAF[i * 4 + k] = p.Hcd_accAF[k];
  }
  return 0;
}
extern "C" int orc_ba_solve(orc_ba* h, int iteration, double lambda, double* x, double* HS, double* bS, double* frame_step, double* calib_step) {
  h->getNullspaces();
  h->solveSystemF(iteration, lambda);
  int n = 4 + 8 * h->nf;
  if (x) for (int i = 0; i < n; i++) x[i] = h->lastX[i];
  if (HS) for (int i = 0; i < n * n; i++) HS[i] = h->lastHS.d[i];
  if (bS) for (int i = 0; i < n; i++) bS[i] = h->lastbS[i];
  if (frame_step) for (int f = 0; f < h->nf; f++) for (int i = 0; i < 8; i++) frame_step[f * 8 + i] = h->frames[f].step[i];
  if (calib_step) for (int i = 0; i < 4; i++) calib_step[i] = h->HCalib.step[i];
  return 0;
}
// EnergyFunctional::accumulateAF_MT / accumulateLF_MT / accumulateSCF_MT (EnergyFunctional.cpp:212-269): the three stitched systems
// solveSystemF adds up (:856-868) — top A without priors, top L with priors, Schur complement — of the accumulators as they stand
extern "C" int orc_ba_get_stitched(orc_ba* h, double* HA, double* bA, double* HL, double* bL, double* Hsc, double* bsc) {
  const int n = 4 + h->nf * 8;
  MatX A, L, S;
  VecX a, l, sv;
  h->stitchTop(false, A, a, false);
  h->stitchTop(true, L, l, true);
  h->stitchSC(S, sv);
  auto put = [n](const MatX& M, const VecX& v, double* Ho, double* bo) {
    if (Ho) for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) Ho[(size_t)i * n + j] = M(i, j);
    if (bo) for (int i = 0; i < n; i++) bo[i] = v[i];
  };
  put(A, a, HA, bA); put(L, l, HL, bL); put(S, sv, Hsc, bsc);
  return 0;
}
extern "C" int orc_ba_get_point_steps(orc_ba* h, float* step) { for (int i = 0; i < h->np; i++) step[i] = h->points[i].step; return 0; }
extern "C" int orc_ba_optimize(orc_ba* h, int mnumOptIts, double* state_out, float* idepth_out, uint8_t* res_state_out, orc_ba_opt_result_t* out) {
  orc_ba_opt_result_t tmp;
  h->optimize(mnumOptIts, out ? out : &tmp);
  if (state_out) for (int f = 0; f < h->nf; f++) for (int i = 0; i < 10; i++) state_out[f * 10 + i] = h->frames[f].state[i];
  if (idepth_out) for (int i = 0; i < h->np; i++) idepth_out[i] = h->points[i].idepth;
  if (res_state_out) for (int i = 0; i < h->nr; i++) res_state_out[i] = (uint8_t)h->res[i].state_state;
  return 0;
}
// lastX (8nf+4 doubles) of every Gauss-Newton iteration of the latest orc_ba_optimize, iteration-major; returns the number of iterations
extern "C" int orc_ba_get_x_trace(orc_ba* h, double* x, int cap_iterations) {
  const int n = 4 + 8 * h->nf, its = (int)h->xTrace.size();
  for (int it = 0; it < its && it < cap_iterations; it++) for (int i = 0; i < n; i++) x[(size_t)it * n + i] = h->xTrace[it][i];
  return its;
}
// the loop's stepsize of every iteration of the latest orc_ba_optimize (FullSystemOptimize.cpp:936-948; 1 without SOLVER_STEPMOMENTUM)
extern "C" int orc_ba_get_step_trace(orc_ba* h, float* stepsize, int cap_iterations) {
  const int its = (int)h->stepTrace.size();
  for (int it = 0; it < its && it < cap_iterations; it++) stepsize[it] = h->stepTrace[it];
  return its;
}
extern "C" int orc_ba_get_post_state(orc_ba* h, orc_ba_post_state_t* o) {
  const int nf = h->nf, n = 4 + 8 * nf;
  for (int i = 0; i < h->np; i++) {
    const Point& p = h->points[i];
    if (o->idepth) o->idepth[i] = p.idepth;
    if (o->step) o->step[i] = p.step;
    if (o->HdiF) o->HdiF[i] = p.HdiF;
    if (o->bdSumF) o->bdSumF[i] = p.bdSumF;
    if (o->idepth_hessian) o->idepth_hessian[i] = p.idepth_hessian;
    if (o->maxRelBaseline) o->maxRelBaseline[i] = p.maxRelBaseline;
    if (o->numGoodResiduals) o->numGoodResiduals[i] = p.numGoodResiduals;
  }
  o->n_toRemove = 0;
  for (int i = 0; i < h->nr; i++) {
    const Residual& r = h->res[i];
    if (o->state_state) o->state_state[i] = (uint8_t)r.state_state;
    if (o->isActiveAndIsGoodNEW) o->isActiveAndIsGoodNEW[i] = r.isActive ? 1 : 0;
    if (o->state_energy) o->state_energy[i] = (float)r.state_energy;
    // (meaningful where the residual is active: the closing linearisation wrote them; zeros elsewhere, like the product)
    if (o->centerProjectedTo) for (int k = 0; k < 3; k++) o->centerProjectedTo[i * 3 + k] = (r.isActive && !r.isLinearized) ? r.centerProjectedTo[k] : 0.f;
    if (o->projectedTo) for (int k = 0; k < 16; k++) o->projectedTo[i * 16 + k] = (r.isActive && !r.isLinearized) ? r.projectedTo[k / 2][k % 2] : 0.f;
    if (o->toRemove) o->toRemove[i] = r.toRemove ? 1 : 0;
    o->n_toRemove += r.toRemove ? 1 : 0;
  }
  for (int f = 0; f < nf; f++) {
    const Frame& F = h->frames[f];
    for (int i = 0; i < 10; i++) {
      if (o->state) o->state[f * 10 + i] = F.state[i];
      if (o->state_zero) o->state_zero[f * 10 + i] = F.state_zero[i];
      if (o->frame_step) o->frame_step[f * 10 + i] = F.step[i];
    }
    if (o->evalPT) { std::memcpy(o->evalPT + f * 12, F.worldToCam_evalPT.R, 72); std::memcpy(o->evalPT + f * 12 + 9, F.worldToCam_evalPT.t, 24); }
    if (o->PRE_worldToCam) { std::memcpy(o->PRE_worldToCam + f * 12, F.PRE_worldToCam.R, 72); std::memcpy(o->PRE_worldToCam + f * 12 + 9, F.PRE_worldToCam.t, 24); }
    if (o->frameEnergyTH) o->frameEnergyTH[f] = F.frameEnergyTH;
  }
  for (int i = 0; i < 4; i++) { o->calib_value[i] = h->HCalib.value[i]; o->calib_value_scaled[i] = h->HCalib.value_scaled[i]; o->calib_step[i] = h->HCalib.step[i]; }
  if (o->lastX) for (int i = 0; i < n; i++) o->lastX[i] = h->lastX[i];
  if (o->lastHS) for (int i = 0; i < n * n; i++) o->lastHS[i] = h->lastHS.d[i];
  if (o->lastbS) for (int i = 0; i < n; i++) o->lastbS[i] = h->lastbS[i];
  o->resInA = h->nresA; o->resInL = h->nresL; o->resInM = h->resInM;
  o->result = h->lastResult;
  return 0;
}
extern "C" int orc_ba_marginalize_points(orc_ba* h, const uint8_t* marg_flag, double* HM_out, double* bM_out) {
  h->marginalizePoints(marg_flag);
  int n = 4 + 8 * h->nf;
  if (HM_out) for (int i = 0; i < n * n; i++) HM_out[i] = h->HM.d[i];
  if (bM_out) for (int i = 0; i < n; i++) bM_out[i] = h->bM[i];
  return 0;
}
// EnergyFunctional::calcLEnergyF_MT / calcMEnergyF (EnergyFunctional.cpp:420-442, :344-351) at the current state, whatever forceAcceptStep says
extern "C" int orc_ba_calc_energies(orc_ba* h, double* EL, double* EM) {
  if (EL) *EL = h->calcLEnergyF();
  if (EM) *EM = h->calcMEnergyF();
  return 0;
}
// what setDeltaF leaves behind (EnergyFunctional.cpp:173-207): cDeltaF, EFFrame::delta / delta_prior, EFPoint::deltaF
extern "C" int orc_ba_get_deltas(orc_ba* h, float* cDeltaF, double* frame_delta, double* frame_delta_prior, float* point_deltaF) {
  if (cDeltaF) for (int i = 0; i < 4; i++) cDeltaF[i] = h->cDeltaF[i];
  for (int f = 0; f < h->nf; f++)
    for (int i = 0; i < 8; i++) {
      if (frame_delta) frame_delta[f * 8 + i] = h->frames[f].delta[i];
      if (frame_delta_prior) frame_delta_prior[f * 8 + i] = h->frames[f].delta_prior[i];
    }
  if (point_deltaF) for (int i = 0; i < h->np; i++) point_deltaF[i] = h->points[i].deltaF;
  return 0;
}
extern "C" int orc_ba_get_tables(orc_ba* h, float* precalc, double* adHost, double* adTarget, float* adHTdeltaF) {
  int nf = h->nf;
  if (precalc)
    for (int i = 0; i < nf * nf; i++) {
      const Precalc& P = h->precalc[i];
      float* o = precalc + (size_t)i * 27;
      std::memcpy(o, P.PRE_KRKiTll, 36); std::memcpy(o + 9, P.PRE_KtTll, 12); std::memcpy(o + 12, P.PRE_RTll_0, 36);
      std::memcpy(o + 21, P.PRE_tTll_0, 12); o[24] = P.PRE_aff_mode[0]; o[25] = P.PRE_aff_mode[1]; o[26] = P.PRE_b0_mode;
    }
  if (adHost) std::memcpy(adHost, h->adHost.data(), h->adHost.size() * 8);
  if (adTarget) std::memcpy(adTarget, h->adTarget.data(), h->adTarget.size() * 8);
  if (adHTdeltaF) std::memcpy(adHTdeltaF, h->adHTdeltaF.data(), h->adHTdeltaF.size() * 4);
  return 0;
}

// EnergyFunctional::marginalizeFrame (EnergyFunctional.cpp:554-660) on plain arrays
extern "C" int orc_marginalize_frame(int nf, int idx, const double* prior8, const double* delta_prior8, const double* HM_in, const double* bM_in,
                                     double* HM_out, double* bM_out) {
  const int odim = nf * 8 + 4, ndim = odim - 8;
  MatX HM(odim, odim); VecX bM(odim);
  for (int i = 0; i < odim; i++) { bM[i] = bM_in[i]; for (int j = 0; j < odim; j++) HM(i, j) = HM_in[(size_t)i * odim + j]; }
  if (idx != nf - 1) {                                  // [step 1] :569-591
    const int io = idx * 8 + 4, ntail = 8 * (nf - idx - 1);
    VecX b2 = bM;
    for (int i = 0; i < ntail; i++) b2[io + i] = bM[io + 8 + i];
    for (int i = 0; i < 8; i++) b2[odim - 8 + i] = bM[io + i];
    bM = b2;
    MatX H2 = HM;
    for (int r = 0; r < odim; r++) { for (int i = 0; i < ntail; i++) H2(r, io + i) = HM(r, io + 8 + i); for (int i = 0; i < 8; i++) H2(r, odim - 8 + i) = HM(r, io + i); }
    MatX H3 = H2;
    for (int c = 0; c < odim; c++) { for (int i = 0; i < ntail; i++) H3(io + i, c) = H2(io + 8 + i, c); for (int i = 0; i < 8; i++) H3(odim - 8 + i, c) = H2(io + i, c); }
    HM = H3;
  }
  for (int i = 0; i < 8; i++) { HM(ndim + i, ndim + i) += prior8[i]; bM[ndim + i] += prior8[i] * delta_prior8[i]; }   // [step 2] :596-597
  VecX SVec(odim), SVecI(odim);                         // [step 3] :601-632
  for (int i = 0; i < odim; i++) { SVec[i] = std::sqrt(std::fabs(HM(i, i)) + 10); SVecI[i] = 1.0 / SVec[i]; }
  MatX HS(odim, odim); VecX bS(odim);
  for (int i = 0; i < odim; i++) { bS[i] = SVecI[i] * bM[i]; for (int j = 0; j < odim; j++) HS(i, j) = SVecI[i] * HM(i, j) * SVecI[j]; }
  MatX hpi(8, 8), hinv;
  for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) hpi(i, j) = 0.5f * (HS(ndim + i, ndim + j) + HS(ndim + i, ndim + j));
  mat_inverse(hpi, hinv);
  for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) hinv(i, j) = 0.5f * (hinv(i, j) + hinv(i, j));
  MatX bli(ndim, 8);
  for (int r = 0; r < ndim; r++) for (int c = 0; c < 8; c++) { double s = 0; for (int k = 0; k < 8; k++) s += HS(ndim + k, r) * hinv(k, c); bli(r, c) = s; }
  MatX HT = HS; VecX bT = bS;
  for (int r = 0; r < ndim; r++) {
    for (int c = 0; c < ndim; c++) { double s = 0; for (int k = 0; k < 8; k++) s += bli(r, k) * HS(ndim + k, c); HT(r, c) = HS(r, c) - s; }
    double s = 0;
    for (int k = 0; k < 8; k++) s += bli(r, k) * bS[ndim + k];
    bT[r] = bS[r] - s;
  }
  for (int r = 0; r < ndim; r++) {
    bM_out[r] = SVec[r] * bT[r];
    for (int c = 0; c < ndim; c++) HM_out[(size_t)r * ndim + c] = 0.5 * (SVec[r] * HT(r, c) * SVec[c] + SVec[c] * HT(c, r) * SVec[r]);
  }
  return 0;
}
