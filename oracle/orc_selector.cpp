// ORACLE — TEST INFRASTRUCTURE ONLY.  parity unpinned (no reference fixtures exist for this path).
//
// CPU restatement of PixelSelector (src/FullSystem/PixelSelector2.cpp in /root/reference):
//   constructor :40-56 (randomPattern from srand(3141592) / rand() & 0xFF), computeHistQuantil :67-81,
//   makeHists :84-189, makeMaps :193-300 (without the display block), select :330-540.
// absSquaredGrad[lvl] = dx*dx + dy*dy as FrameHessian::makeImages writes it (HessianBlocks.cpp:192; the B-gradient
// weight of :194-198 is 1 for the identity response, the only case covered).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "orc_api.h"

namespace {
const float setting_minGradHistCut = 0.5f, setting_minGradHistAdd = 7, setting_gradDownweightPerLevel = 0.75f;   // settings.cpp:105-107

struct Selector {
  int w, h, w1, w2;
  std::vector<unsigned char> randomPattern;
  std::vector<float> ths, thsSmoothed;
  int thsStep = 0;
  int currentPotential = 3;
  const float* dI0;                       // AoS float3, level 0
  std::vector<float> abs0, abs1, abs2;    // absSquaredGrad[0..2]

  int computeHistQuantil(const int* hist, float below) {
    int th = hist[0] * below + 0.5f;
    for (int i = 0; i < 90; i++) {
      th -= hist[i + 1];
      if (th < 0) return i;
    }
    return 90;
  }
  void makeHists() {
    const int w32 = w / 32, h32 = h / 32;
    thsStep = w32;
    ths.assign((size_t)w32 * h32 + 100, 0.f);
    thsSmoothed.assign((size_t)w32 * h32 + 100, 0.f);
    int hist0[100] = {0};   // (the reference clears 50 of its 100 ints; the quantile walk stops long before index 50)
    for (int y = 0; y < h32; y++)
      for (int x = 0; x < w32; x++) {
        const float* map0 = abs0.data() + 32 * x + 32 * y * w;
        std::memset(hist0, 0, sizeof(int) * 50);
        for (int j = 0; j < 32; j++)
          for (int i = 0; i < 32; i++) {
            int it = i + 32 * x, jt = j + 32 * y;
            if (it > w - 2 || jt > h - 2 || it < 1 || jt < 1) continue;
            int g = sqrtf(map0[i + j * w]);
            if (g > 48) g = 48;
            hist0[g + 1]++;
            hist0[0]++;
          }
        ths[x + y * w32] = computeHistQuantil(hist0, setting_minGradHistCut) + setting_minGradHistAdd;
      }
    for (int y = 0; y < h32; y++)
      for (int x = 0; x < w32; x++) {
        float sum = 0, num = 0;
        if (x > 0) {
          if (y > 0) { num++; sum += ths[x - 1 + (y - 1) * w32]; }
          if (y < h32 - 1) { num++; sum += ths[x - 1 + (y + 1) * w32]; }
          num++; sum += ths[x - 1 + y * w32];
        }
        if (x < w32 - 1) {
          if (y > 0) { num++; sum += ths[x + 1 + (y - 1) * w32]; }
          if (y < h32 - 1) { num++; sum += ths[x + 1 + (y + 1) * w32]; }
          num++; sum += ths[x + 1 + y * w32];
        }
        if (y > 0) { num++; sum += ths[x + (y - 1) * w32]; }
        if (y < h32 - 1) { num++; sum += ths[x + (y + 1) * w32]; }
        num++; sum += ths[x + y * w32];
        thsSmoothed[x + y * w32] = (sum / num) * (sum / num);
      }
  }
  void select(float* map_out, int pot, float thFactor, int n[3]) {
    static const float directions[16][2] = {{0, 1.0000f}, {0.3827f, 0.9239f}, {0.1951f, 0.9808f}, {0.9239f, 0.3827f}, {0.7071f, 0.7071f},
                                            {0.3827f, -0.9239f}, {0.8315f, 0.5556f}, {0.8315f, -0.5556f}, {0.5556f, -0.8315f}, {0.9808f, 0.1951f},
                                            {0.9239f, -0.3827f}, {0.7071f, -0.7071f}, {0.5556f, 0.8315f}, {0.9808f, -0.1951f}, {1.0000f, 0.0000f},
                                            {0.1951f, -0.9808f}};
    std::memset(map_out, 0, sizeof(float) * w * h);
    const float dw1 = setting_gradDownweightPerLevel, dw2 = dw1 * dw1;
    int n3 = 0, n2 = 0, n4 = 0;
    for (int y4 = 0; y4 < h; y4 += (4 * pot))
      for (int x4 = 0; x4 < w; x4 += (4 * pot)) {
        int my3 = std::min((4 * pot), h - y4), mx3 = std::min((4 * pot), w - x4);
        int bestIdx4 = -1; float bestVal4 = 0;
        const float* dir4 = directions[randomPattern[n2] & 0xF];
        for (int y3 = 0; y3 < my3; y3 += (2 * pot))
          for (int x3 = 0; x3 < mx3; x3 += (2 * pot)) {
            int x34 = x3 + x4, y34 = y3 + y4;
            int my2 = std::min((2 * pot), h - y34), mx2 = std::min((2 * pot), w - x34);
            int bestIdx3 = -1; float bestVal3 = 0;
            const float* dir3 = directions[randomPattern[n2] & 0xF];
            for (int y2 = 0; y2 < my2; y2 += pot)
              for (int x2 = 0; x2 < mx2; x2 += pot) {
                int x234 = x2 + x34, y234 = y2 + y34;
                int my1 = std::min(pot, h - y234), mx1 = std::min(pot, w - x234);
                int bestIdx2 = -1; float bestVal2 = 0;
                const float* dir2 = directions[randomPattern[n2] & 0xF];
                for (int y1 = 0; y1 < my1; y1 += 1)
                  for (int x1 = 0; x1 < mx1; x1 += 1) {
                    int idx = x1 + x234 + w * (y1 + y234);
                    int xf = x1 + x234, yf = y1 + y234;
                    if (xf < 4 || xf >= w - 5 || yf < 4 || yf > h - 4) continue;
                    float pixelTH0 = thsSmoothed[(xf >> 5) + (yf >> 5) * thsStep];
                    float pixelTH1 = pixelTH0 * dw1;
                    float pixelTH2 = pixelTH1 * dw2;
                    float ag0 = abs0[idx];
                    const float gx = dI0[idx * 3 + 1], gy = dI0[idx * 3 + 2];
                    if (ag0 > pixelTH0 * thFactor) {
                      float dirNorm = fabsf((float)(gx * dir2[0] + gy * dir2[1]));
                      if (dirNorm > bestVal2) { bestVal2 = dirNorm; bestIdx2 = idx; bestIdx3 = -2; bestIdx4 = -2; }
                    }
                    if (bestIdx3 == -2) continue;
                    float ag1 = abs1[(int)(xf * 0.5f + 0.25f) + (int)(yf * 0.5f + 0.25f) * w1];
                    if (ag1 > pixelTH1 * thFactor) {
                      float dirNorm = fabsf((float)(gx * dir3[0] + gy * dir3[1]));
                      if (dirNorm > bestVal3) { bestVal3 = dirNorm; bestIdx3 = idx; bestIdx4 = -2; }
                    }
                    if (bestIdx4 == -2) continue;
                    float ag2 = abs2[(int)(xf * 0.25f + 0.125) + (int)(yf * 0.25f + 0.125) * w2];
                    if (ag2 > pixelTH2 * thFactor) {
                      float dirNorm = fabsf((float)(gx * dir4[0] + gy * dir4[1]));
                      if (dirNorm > bestVal4) { bestVal4 = dirNorm; bestIdx4 = idx; }
                    }
                  }
                if (bestIdx2 > 0) { map_out[bestIdx2] = 1; bestVal3 = 1e10; n2++; }
              }
            if (bestIdx3 > 0) { map_out[bestIdx3] = 2; bestVal4 = 1e10; n3++; }
          }
        if (bestIdx4 > 0) { map_out[bestIdx4] = 4; n4++; }
      }
    n[0] = n2; n[1] = n3; n[2] = n4;
  }
  int makeMaps(float* map_out, float density, int recursionsLeft, float thFactor) {
    float numHave = 0, numWant = density, quotia;
    int idealPotential = currentPotential;
    {
      int n[3];
      select(map_out, currentPotential, thFactor, n);
      numHave = n[0] + n[1] + n[2];
      quotia = numWant / numHave;
      float K = numHave * (currentPotential + 1) * (currentPotential + 1);
      idealPotential = sqrtf(K / numWant) - 1;
      if (idealPotential < 1) idealPotential = 1;
      if (recursionsLeft > 0 && quotia > 1.25 && currentPotential > 1) {
        if (idealPotential >= currentPotential) idealPotential = currentPotential - 1;
        currentPotential = idealPotential;
        return makeMaps(map_out, density, recursionsLeft - 1, thFactor);
      } else if (recursionsLeft > 0 && quotia < 0.25) {
        if (idealPotential <= currentPotential) idealPotential = currentPotential + 1;
        currentPotential = idealPotential;
        return makeMaps(map_out, density, recursionsLeft - 1, thFactor);
      }
    }
    int numHaveSub = numHave;
    if (quotia < 0.95) {
      int wh = w * h, rn = 0;
      unsigned char charTH = 255 * quotia;
      for (int i = 0; i < wh; i++)
        if (map_out[i] != 0) {
          if (randomPattern[rn] > charTH) { map_out[i] = 0; numHaveSub--; }
          rn++;
        }
    }
    currentPotential = idealPotential;
    return numHaveSub;
  }
};
}  // namespace

// dIp[0..2]: AoS float3 levels 0..2 (sizes w>>l x h>>l).  potential: PixelSelector::currentPotential, in/out.
static const float* g_gammaB = nullptr;
static float g_gammaB_store[256];
// CalibHessian::B for the gamma-weighted absSquaredGrad (setting_gammaWeightsPixelSelect == 1); NULL = identity response
extern "C" void orc_set_gamma(const float* B) {
  if (!B) { g_gammaB = nullptr; return; }
  for (int i = 0; i < 256; i++) g_gammaB_store[i] = B[i];
  g_gammaB = g_gammaB_store;
}
// FullSystem::setGammaFunction (FullSystem.cpp:210-234)
extern "C" void orc_gamma_from_binv(const float* BInv, float* B) {
  for (int i = 0; i < 256; i++) B[i] = 0;
  for (int i = 1; i < 255; i++)
    for (int s = 1; s < 255; s++)
      if (BInv[s] <= i && BInv[s + 1] >= i) { B[i] = s + (i - BInv[s]) / (BInv[s + 1] - BInv[s]); break; }
  B[0] = 0;
  B[255] = 255;
}
extern "C" int orc_pixel_select(const float* const* dIp, int w, int h, float density, int recursionsLeft, float thFactor, int* potential,
                                float* map_out) {
  Selector S;
  S.w = w; S.h = h; S.w1 = w >> 1; S.w2 = w >> 2;
  S.randomPattern.resize((size_t)w * h);
  std::srand(3141592);
  for (int i = 0; i < w * h; i++) S.randomPattern[i] = rand() & 0xFF;
  S.dI0 = dIp[0];
  std::vector<float>* ab[3] = {&S.abs0, &S.abs1, &S.abs2};
  for (int l = 0; l < 3; l++) {
    const int wl = w >> l, hl = h >> l;
    ab[l]->assign((size_t)wl * hl, 0.f);
    for (int i = 0; i < wl * hl; i++) {
      const float dx = dIp[l][i * 3 + 1], dy = dIp[l][i * 3 + 2];
      float a = dx * dx + dy * dy;                                  // HessianBlocks.cpp:192
      if (g_gammaB) {                                               // :194-198 with CalibHessian::getBGradOnly (HessianBlocks.h:356-362)
        int c = dIp[l][i * 3] + 0.5f;
        if (c < 5) c = 5;
        if (c > 250) c = 250;
        const float gw = g_gammaB[c + 1] - g_gammaB[c];
        a *= gw * gw;
      }
      (*ab[l])[i] = a;
    }
  }
  S.currentPotential = *potential;
  S.makeHists();
  const int nsel = S.makeMaps(map_out, density, recursionsLeft, thFactor);
  *potential = S.currentPotential;
  return nsel;
}
// the first n outputs of rand() & 0xFF after srand(3141592) (PixelSelector2.cpp:43-44): pins the library's own generator
extern "C" void orc_selector_random_pattern(int n, unsigned char* out) {
  std::srand(3141592);
  for (int i = 0; i < n; i++) out[i] = rand() & 0xFF;
}
