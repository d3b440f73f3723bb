// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
// Restates the float accumulators of src/OptimizationBackend/MatrixAccumulators.h with the
// reference's summation order: 4 SSE-lane partials where the reference uses __m128, and the
// three-tier carry (flush level 0 into "1k" when more than 1000 updates are pending, "1k" into
// "1m" likewise; finish() forces both) of :64-79, :137-152, :872-903, :1260-1276.
#pragma once
#include <cstring>
#include <cstddef>

namespace orc {

// TRUTH MODE (not in the reference): when on, the windowed-BA accumulators below also keep every sum in double, with
// double products of the same float inputs, and their readers (h() / a()) return those instead of the float tiers.
// The float path is unchanged; the mode exists so that tests can measure how far the float summation order of the CPU
// path and of the device path each sit from an order-independent value.
inline int& acc64_mode() { static int on = 0; return on; }

// MatrixAccumulators.h:907-1277 (Accumulator9).  Variable order [J0..J7, J8=r].
struct Accumulator9 {
  float H[9][9];
  size_t num;
  float S[45][4], S1k[45][4], S1m[45][4];
  float numIn1, numIn1k, numIn1m;

  void initialize() {
    std::memset(H, 0, sizeof(H));
    std::memset(S, 0, sizeof(S)); std::memset(S1k, 0, sizeof(S1k)); std::memset(S1m, 0, sizeof(S1m));
    num = 0; numIn1 = numIn1k = numIn1m = 0;
  }
  void shiftUp(bool force) {  // :1260-1276
    if (numIn1 > 1000 || force) {
      for (int i = 0; i < 45; i++) for (int l = 0; l < 4; l++) S1k[i][l] = S[i][l] + S1k[i][l];
      numIn1k += numIn1; numIn1 = 0;
      std::memset(S, 0, sizeof(S));
    }
    if (numIn1k > 1000 || force) {
      for (int i = 0; i < 45; i++) for (int l = 0; l < 4; l++) S1m[i][l] = S1k[i][l] + S1m[i][l];
      numIn1m += numIn1k; numIn1k = 0;
      std::memset(S1k, 0, sizeof(S1k));
    }
  }
  // :1025-1100 — four points at once, lane l holds point l.  J[k][l].
  void updateSSE_weighted(const float J[9][4], const float w[4]) {
    int idx = 0;
    for (int r = 0; r < 9; r++) {
      float Jw[4];
      for (int l = 0; l < 4; l++) Jw[l] = J[r][l] * w[l];
      for (int c = r; c < 9; c++) {
        for (int l = 0; l < 4; l++) S[idx][l] = S[idx][l] + Jw[l] * J[c][l];
        idx++;
      }
    }
    num += 4; numIn1++;
    shiftUp(false);
  }
  void finish() {  // :927-947
    std::memset(H, 0, sizeof(H));
    shiftUp(true);
    int idx = 0;
    for (int r = 0; r < 9; r++)
      for (int c = r; c < 9; c++) {
        float d = S1m[idx][0] + S1m[idx][1] + S1m[idx][2] + S1m[idx][3];
        H[r][c] = H[c][r] = d;
        idx++;
      }
  }
};

// MatrixAccumulators.h:564-904 (AccumulatorApprox). 13x13: [C0..C3 | xi0..xi5 | a b | r].
struct AccumulatorApprox {
  float H[13][13];
  size_t num;
  float Data[60], Data1k[60], Data1m[60];
  float TR[32], TR1k[32], TR1m[32];
  float BR[8], BR1k[8], BR1m[8];
  float numIn1, numIn1k, numIn1m;
  double Dd[60], TRd[32], BRd[8], Hd[13][13];   // truth mode
  double h(int r, int c) const { return acc64_mode() ? Hd[r][c] : (double)H[r][c]; }

  void initialize() {
    std::memset(Dd, 0, sizeof(Dd)); std::memset(TRd, 0, sizeof(TRd)); std::memset(BRd, 0, sizeof(BRd));
    std::memset(Data, 0, sizeof(Data)); std::memset(Data1k, 0, sizeof(Data)); std::memset(Data1m, 0, sizeof(Data));
    std::memset(TR, 0, sizeof(TR)); std::memset(TR1k, 0, sizeof(TR)); std::memset(TR1m, 0, sizeof(TR));
    std::memset(BR, 0, sizeof(BR)); std::memset(BR1k, 0, sizeof(BR)); std::memset(BR1m, 0, sizeof(BR));
    num = 0; numIn1 = numIn1k = numIn1m = 0;
  }
  void shiftUp(bool force) {  // :872-903
    if (numIn1 > 1000 || force) {
      for (int i = 0; i < 60; i++) Data1k[i] = Data[i] + Data1k[i];
      for (int i = 0; i < 32; i++) TR1k[i] = TR[i] + TR1k[i];
      for (int i = 0; i < 8; i++) BR1k[i] = BR[i] + BR1k[i];
      numIn1k += numIn1; numIn1 = 0;
      std::memset(Data, 0, sizeof(Data)); std::memset(TR, 0, sizeof(TR)); std::memset(BR, 0, sizeof(BR));
    }
    if (numIn1k > 1000 || force) {
      for (int i = 0; i < 60; i++) Data1m[i] = Data1k[i] + Data1m[i];
      for (int i = 0; i < 32; i++) TR1m[i] = TR1k[i] + TR1m[i];
      for (int i = 0; i < 8; i++) BR1m[i] = BR1k[i] + BR1m[i];
      numIn1m += numIn1k; numIn1k = 0;
      std::memset(Data1k, 0, sizeof(Data)); std::memset(TR1k, 0, sizeof(TR)); std::memset(BR1k, 0, sizeof(BR));
    }
  }
  // :714-790. x = [x4|x6], y = [y4|y6]
  void update(const float* x4, const float* x6, const float* y4, const float* y6, float a, float b, float c) {
    float x[10], y[10];
    for (int i = 0; i < 4; i++) { x[i] = x4[i]; y[i] = y4[i]; }
    for (int i = 0; i < 6; i++) { x[4 + i] = x6[i]; y[4 + i] = y6[i]; }
    int idx = 0;
    for (int r = 0; r < 10; r++)
      for (int cc = r; cc < 10; cc++) {
        // Data[idx] += a*x[cc]*x[r] + c*y[cc]*y[r] + b*(x[cc]*y[r] + y[cc]*x[r]);
        Data[idx] += a * x[cc] * x[r] + c * y[cc] * y[r] + b * (x[cc] * y[r] + y[cc] * x[r]);
        idx++;
      }
    if (acc64_mode()) {   // truth mode only: the timed CPU baseline must not pay for it
      idx = 0;
      for (int r = 0; r < 10; r++)
        for (int cc = r; cc < 10; cc++) {
          Dd[idx] += (double)a * x[cc] * x[r] + (double)c * y[cc] * y[r] + (double)b * ((double)x[cc] * y[r] + (double)y[cc] * x[r]);
          idx++;
        }
    }
    num++; numIn1++;
    shiftUp(false);
  }
  // :793-840
  void updateTopRight(const float* x4, const float* x6, const float* y4, const float* y6,
                      float TR00, float TR10, float TR01, float TR11, float TR02, float TR12) {
    float x[10], y[10];
    for (int i = 0; i < 4; i++) { x[i] = x4[i]; y[i] = y4[i]; }
    for (int i = 0; i < 6; i++) { x[4 + i] = x6[i]; y[4 + i] = y6[i]; }
    for (int r = 0; r < 10; r++) {
      TR[3 * r + 0] += x[r] * TR00 + y[r] * TR10;
      TR[3 * r + 1] += x[r] * TR01 + y[r] * TR11;
      TR[3 * r + 2] += x[r] * TR02 + y[r] * TR12;
    }
    if (acc64_mode())
      for (int r = 0; r < 10; r++) {
        TRd[3 * r + 0] += (double)x[r] * TR00 + (double)y[r] * TR10;
        TRd[3 * r + 1] += (double)x[r] * TR01 + (double)y[r] * TR11;
        TRd[3 * r + 2] += (double)x[r] * TR02 + (double)y[r] * TR12;
      }
  }
  // :842-855
  void updateBotRight(float a00, float a01, float a02, float a11, float a12, float a22) {
    BR[0] += a00; BR[1] += a01; BR[2] += a02; BR[3] += a11; BR[4] += a12; BR[5] += a22;
    if (acc64_mode()) { BRd[0] += a00; BRd[1] += a01; BRd[2] += a02; BRd[3] += a11; BRd[4] += a12; BRd[5] += a22; }
  }
  void finish() {  // :589-618
    std::memset(H, 0, sizeof(H));
    shiftUp(true);
    int idx = 0;
    for (int r = 0; r < 10; r++)
      for (int c = r; c < 10; c++) { H[r][c] = H[c][r] = Data1m[idx]; idx++; }
    idx = 0;
    for (int r = 0; r < 10; r++)
      for (int c = 0; c < 3; c++) { H[r][c + 10] = H[c + 10][r] = TR1m[idx]; idx++; }
    H[10][10] = BR1m[0];
    H[10][11] = H[11][10] = BR1m[1];
    H[10][12] = H[12][10] = BR1m[2];
    H[11][11] = BR1m[3];
    H[11][12] = H[12][11] = BR1m[4];
    H[12][12] = BR1m[5];
    num = (size_t)(numIn1 + numIn1k + numIn1m);
    std::memset(Hd, 0, sizeof(Hd));
    idx = 0;
    for (int r = 0; r < 10; r++)
      for (int c = r; c < 10; c++) { Hd[r][c] = Hd[c][r] = Dd[idx]; idx++; }
    idx = 0;
    for (int r = 0; r < 10; r++)
      for (int c = 0; c < 3; c++) { Hd[r][c + 10] = Hd[c + 10][r] = TRd[idx]; idx++; }
    Hd[10][10] = BRd[0]; Hd[10][11] = Hd[11][10] = BRd[1]; Hd[10][12] = Hd[12][10] = BRd[2];
    Hd[11][11] = BRd[3]; Hd[11][12] = Hd[12][11] = BRd[4]; Hd[12][12] = BRd[5];
  }
};

// MatrixAccumulators.h:31-80 (AccumulatorXX<i,j>) and :155-214 (AccumulatorX<i>)
template <int I, int J>
struct AccumulatorXX {
  float A[I][J], A1k[I][J], A1m[I][J];
  size_t num;
  float numIn1, numIn1k, numIn1m;
  double Ad[I][J];   // truth mode
  double a(int i, int j) const { return acc64_mode() ? Ad[i][j] : (double)A1m[i][j]; }
  void initialize() {
    std::memset(Ad, 0, sizeof(Ad));
    std::memset(A, 0, sizeof(A)); std::memset(A1k, 0, sizeof(A)); std::memset(A1m, 0, sizeof(A));
    num = 0; numIn1 = numIn1k = numIn1m = 0;
  }
  void shiftUp(bool force) {
    if (numIn1 > 1000 || force) {
      for (int i = 0; i < I; i++) for (int j = 0; j < J; j++) { A1k[i][j] += A[i][j]; A[i][j] = 0; }
      numIn1k += numIn1; numIn1 = 0;
    }
    if (numIn1k > 1000 || force) {
      for (int i = 0; i < I; i++) for (int j = 0; j < J; j++) { A1m[i][j] += A1k[i][j]; A1k[i][j] = 0; }
      numIn1m += numIn1k; numIn1k = 0;
    }
  }
  void update(const float* L, const float* R, float w) {  // A += w*L*R^T  ((w*L)*R^T in Eigen)
    for (int i = 0; i < I; i++) {
      float wl = w * L[i];
      for (int j = 0; j < J; j++) A[i][j] += wl * R[j];
    }
    if (acc64_mode())
      for (int i = 0; i < I; i++) for (int j = 0; j < J; j++) Ad[i][j] += (double)w * L[i] * R[j];
    numIn1++;
    shiftUp(false);
  }
  void finish() { shiftUp(true); num = (size_t)(numIn1 + numIn1k + numIn1m); }
};
template <int I>
struct AccumulatorX {
  float A[I], A1k[I], A1m[I];
  size_t num;
  float numIn1, numIn1k, numIn1m;
  double Ad[I];   // truth mode
  double a(int i) const { return acc64_mode() ? Ad[i] : (double)A1m[i]; }
  void initialize() {
    std::memset(Ad, 0, sizeof(Ad));
    std::memset(A, 0, sizeof(A)); std::memset(A1k, 0, sizeof(A)); std::memset(A1m, 0, sizeof(A));
    num = 0; numIn1 = numIn1k = numIn1m = 0;
  }
  void shiftUp(bool force) {
    if (numIn1 > 1000 || force) {
      for (int i = 0; i < I; i++) { A1k[i] += A[i]; A[i] = 0; }
      numIn1k += numIn1; numIn1 = 0;
    }
    if (numIn1k > 1000 || force) {
      for (int i = 0; i < I; i++) { A1m[i] += A1k[i]; A1k[i] = 0; }
      numIn1m += numIn1k; numIn1k = 0;
    }
  }
  void update(const float* L, float w) {
    for (int i = 0; i < I; i++) A[i] += w * L[i];
    if (acc64_mode()) for (int i = 0; i < I; i++) Ad[i] += (double)w * L[i];
    numIn1++;
    shiftUp(false);
  }
  void finish() { shiftUp(true); num = (size_t)(numIn1 + numIn1k + numIn1m); }
};

}  // namespace orc
