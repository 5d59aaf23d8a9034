// jxlsynth — forward transforms / quantisation tables of the synthesiser (encoder side).  Independent of oracle/.
// Conventions follow JPEG XL's VarDCT: DCT scaled so that DC = block mean, coefficient blocks in
// the "cols >= rows" stored layout, 27 strategies numbered as in the codestream.
#pragma once
#include "synth_entropy.h"

namespace synth {

enum { S_DCT = 0, S_IDENTITY, S_DCT2X2, S_DCT4X4, S_DCT16X16, S_DCT32X32, S_DCT16X8, S_DCT8X16, S_DCT32X8, S_DCT8X32,
       S_DCT32X16, S_DCT16X32, S_DCT4X8, S_DCT8X4, S_AFV0, S_AFV1, S_AFV2, S_AFV3, S_DCT64X64, S_DCT64X32, S_DCT32X64 };
static const uint8_t kCovX[27] = {1, 1, 1, 1, 2, 4, 1, 2, 1, 4, 2, 4, 1, 1, 1, 1, 1, 1, 8, 4, 8, 16, 8, 16, 32, 16, 32};
static const uint8_t kCovY[27] = {1, 1, 1, 1, 2, 4, 2, 1, 4, 1, 4, 2, 1, 1, 1, 1, 1, 1, 8, 8, 4, 16, 16, 8, 32, 32, 16};
static const uint8_t kBucket[27] = {0, 1, 1, 1, 2, 3, 4, 4, 5, 5, 6, 6, 1, 1, 1, 1, 1, 1, 7, 8, 8, 9, 10, 10, 11, 12, 12};
static const uint8_t kKind[27] = {0, 1, 2, 3, 4, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 10, 10, 11, 12, 12, 13, 14, 14, 15, 16, 16};
static const uint8_t kKindR[17] = {1, 1, 1, 1, 2, 4, 1, 1, 2, 1, 1, 8, 4, 16, 8, 32, 16};
static const uint8_t kKindC[17] = {1, 1, 1, 1, 2, 4, 2, 4, 4, 1, 1, 8, 8, 16, 16, 32, 32};

inline int ILog2(int v) { int r = 0; while ((1 << r) < v) r++; return r; }

struct Wc { std::vector<float> t[9]; Wc() { for (int l = 1; l <= 8; l++) { int N = 1 << l; t[l].resize(N / 2); for (int i = 0; i < N / 2; i++) t[l][i] = (float)(1.0 / (2.0 * std::cos((i + 0.5) * M_PI / N))); } } };
inline const Wc& wc() { static Wc w; return w; }
static const float kS2 = 1.41421356237309504880f;

inline void FDCT1D(float* v, int N, float* tmp) {  // unscaled forward
  if (N == 1) return;
  if (N == 2) { float a = v[0], b = v[1]; v[0] = a + b; v[1] = a - b; return; }
  const int H = N / 2;
  for (int i = 0; i < H; i++) tmp[i] = v[i] + v[N - 1 - i];
  FDCT1D(tmp, H, tmp + N);
  const float* w = wc().t[ILog2(N)].data();
  for (int i = 0; i < H; i++) tmp[H + i] = (v[i] - v[N - 1 - i]) * w[i];
  FDCT1D(tmp + H, H, tmp + N);
  tmp[H] = tmp[H] * kS2 + tmp[H + 1];
  for (int i = 1; i + 1 < H; i++) tmp[H + i] += tmp[H + i + 1];
  for (int i = 0; i < H; i++) { v[2 * i] = tmp[i]; v[2 * i + 1] = tmp[H + i]; }
}
inline void IDCT1D(float* v, int N, float* tmp) {
  if (N == 1) return;
  if (N == 2) { float a = v[0], b = v[1]; v[0] = a + b; v[1] = a - b; return; }
  const int H = N / 2;
  for (int i = 0; i < H; i++) { tmp[i] = v[2 * i]; tmp[H + i] = v[2 * i + 1]; }
  IDCT1D(tmp, H, tmp + N);
  for (int i = H - 1; i > 0; i--) tmp[H + i] += tmp[H + i - 1];
  tmp[H] *= kS2;
  IDCT1D(tmp + H, H, tmp + N);
  const float* w = wc().t[ILog2(N)].data();
  for (int i = 0; i < H; i++) { float a = tmp[i], b = tmp[H + i] * w[i]; v[i] = a + b; v[N - 1 - i] = a - b; }
}
// pixels in[y*stride+x] (R rows, C cols) -> semantic coefficients c[v*C+u], DC = mean
inline void FDCT2D(const float* in, int stride, int R, int C, float* c) {
  std::vector<float> col(R), tmp(4 * std::max(R, C));
  for (int x = 0; x < C; x++) {
    for (int y = 0; y < R; y++) col[y] = in[(size_t)y * stride + x];
    FDCT1D(col.data(), R, tmp.data());
    for (int v = 0; v < R; v++) c[(size_t)v * C + x] = col[v] / R;
  }
  for (int v = 0; v < R; v++) {
    float* row = c + (size_t)v * C;
    FDCT1D(row, C, tmp.data());
    for (int u = 0; u < C; u++) row[u] /= C;
  }
}
inline void IDCT2D(const float* c, int R, int C, float* out, int stride) {
  std::vector<float> buf((size_t)R * C), col(R), tmp(4 * std::max(R, C));
  for (int v = 0; v < R; v++) { float* row = &buf[(size_t)v * C]; memcpy(row, c + (size_t)v * C, sizeof(float) * C); IDCT1D(row, C, tmp.data()); }
  for (int x = 0; x < C; x++) {
    for (int v = 0; v < R; v++) col[v] = buf[(size_t)v * C + x];
    IDCT1D(col.data(), R, tmp.data());
    for (int y = 0; y < R; y++) out[(size_t)y * stride + x] = col[y];
  }
}
inline size_t StoredIdx(int R, int C, int v, int u) { return R >= C ? (size_t)u * R + v : (size_t)v * C + u; }
inline float Resample(int N, int k) { return k == 0 ? 1.0f : (float)(std::sin(k * M_PI / (2.0 * N)) / std::sin(k * M_PI / (16.0 * N)) / 8.0); }

// Forward transform of one varblock: pixels (stride) -> stored-layout coefficients (size covered*64)
// AFV basis (dec_transforms-inl.h k4x4AFVBasis), orthonormal: the forward transform is its transpose
static const float kAFVBasis[16][16] = {
    {0.25f, 0.25f, 0.25f, 0.25f, 0.25f, 0.25f, 0.25f, 0.25f, 0.25f, 0.25f, 0.25f, 0.25f, 0.25f, 0.25f, 0.25f, 0.25f},
    {0.876902929799142f, 0.2206518106944235f, -0.10140050393753763f, -0.1014005039375375f, 0.2206518106944236f, -0.10140050393753777f, -0.10140050393753772f, -0.10140050393753763f, -0.10140050393753758f, -0.10140050393753769f, -0.1014005039375375f, -0.10140050393753768f, -0.10140050393753768f, -0.10140050393753759f, -0.10140050393753763f, -0.10140050393753741f},
    {0.0f, 0.0f, 0.40670075830260755f, 0.44444816619734445f, 0.0f, 0.0f, 0.19574399372042936f, 0.2929100136981264f, -0.40670075830260716f, -0.19574399372042872f, 0.0f, 0.11379074460448091f, -0.44444816619734384f, -0.29291001369812636f, -0.1137907446044814f, 0.0f},
    {0.0f, 0.0f, -0.21255748058288748f, 0.3085497062849767f, 0.0f, 0.4706702258572536f, -0.1621205195722993f, 0.0f, -0.21255748058287047f, -0.16212051957228327f, -0.47067022585725277f, -0.1464291867126764f, 0.3085497062849487f, 0.0f, -0.14642918671266536f, 0.4251149611657548f},
    {0.0f, -0.7071067811865474f, 0.0f, 0.0f, 0.7071067811865476f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f},
    {-0.4105377591765233f, 0.6235485373547691f, -0.06435071657946274f, -0.06435071657946266f, 0.6235485373547694f, -0.06435071657946284f, -0.0643507165794628f, -0.06435071657946274f, -0.06435071657946272f, -0.06435071657946279f, -0.06435071657946266f, -0.06435071657946277f, -0.06435071657946277f, -0.06435071657946273f, -0.06435071657946274f, -0.0643507165794626f},
    {0.0f, 0.0f, -0.4517556589999482f, 0.15854503551840063f, 0.0f, -0.04038515160822202f, 0.0074182263792423875f, 0.39351034269210167f, -0.45175565899994635f, 0.007418226379244351f, 0.1107416575309343f, 0.08298163094882051f, 0.15854503551839705f, 0.3935103426921022f, 0.0829816309488214f, -0.45175565899994796f},
    {0.0f, 0.0f, -0.304684750724869f, 0.5112616136591823f, 0.0f, 0.0f, -0.290480129728998f, -0.06578701549142804f, 0.304684750724884f, 0.2904801297290076f, 0.0f, -0.23889773523344604f, -0.5112616136592012f, 0.06578701549142545f, 0.23889773523345467f, 0.0f},
    {0.0f, 0.0f, 0.3017929516615495f, 0.25792362796341184f, 0.0f, 0.16272340142866204f, 0.09520022653475037f, 0.0f, 0.3017929516615503f, 0.09520022653475055f, -0.16272340142866173f, -0.35312385449816297f, 0.25792362796341295f, 0.0f, -0.3531238544981624f, -0.6035859033230976f},
    {0.0f, 0.0f, 0.40824829046386274f, 0.0f, 0.0f, 0.0f, 0.0f, -0.4082482904638628f, -0.4082482904638635f, 0.0f, 0.0f, -0.40824829046386296f, 0.0f, 0.4082482904638634f, 0.408248290463863f, 0.0f},
    {0.0f, 0.0f, 0.1747866975480809f, 0.0812611176717539f, 0.0f, 0.0f, -0.3675398009862027f, -0.307882213957909f, -0.17478669754808135f, 0.3675398009862011f, 0.0f, 0.4826689115059883f, -0.08126111767175039f, 0.30788221395790305f, -0.48266891150598584f, 0.0f},
    {0.0f, 0.0f, -0.21105601049335784f, 0.18567180916109802f, 0.0f, 0.0f, 0.49215859013738733f, -0.38525013709251915f, 0.21105601049335806f, -0.49215859013738905f, 0.0f, 0.17419412659916217f, -0.18567180916109904f, 0.3852501370925211f, -0.1741941265991621f, 0.0f},
    {0.0f, 0.0f, -0.14266084808807264f, -0.3416446842253372f, 0.0f, 0.7367497537172237f, 0.24627107722075148f, -0.08574019035519306f, -0.14266084808807344f, 0.24627107722075137f, 0.14883399227113567f, -0.04768680350229251f, -0.3416446842253373f, -0.08574019035519267f, -0.047686803502292804f, -0.14266084808807242f},
    {0.0f, 0.0f, -0.13813540350758585f, 0.3302282550303788f, 0.0f, 0.08755115000587084f, -0.07946706605909573f, -0.4613374887461511f, -0.13813540350758294f, -0.07946706605910261f, 0.49724647109535086f, 0.12538059448563663f, 0.3302282550303805f, -0.4613374887461554f, 0.12538059448564315f, -0.13813540350758452f},
    {0.0f, 0.0f, -0.17437602599651067f, 0.0702790691196284f, 0.0f, -0.2921026642334881f, 0.3623817333531167f, 0.0f, -0.1743760259965108f, 0.36238173335311646f, 0.29210266423348785f, -0.4326608024727445f, 0.07027906911962818f, 0.0f, -0.4326608024727457f, 0.34875205199302267f},
    {0.0f, 0.0f, 0.11354987314994337f, -0.07417504595810355f, 0.0f, 0.19402893032594343f, -0.435190496523228f, 0.21918684838857466f, 0.11354987314994257f, -0.4351904965232251f, 0.5550443808910661f, -0.25468277124066463f, -0.07417504595810233f, 0.2191868483885728f, -0.25468277124066413f, 0.1135498731499429f},
};

inline void ForwardTransform(int s, const float* px, int stride, float* coef) {
  const int cx = kCovX[s], cy = kCovY[s], R = 8 * cy, C = 8 * cx;
  switch (s) {
    case S_IDENTITY: {
      float dcs[4];
      for (int y = 0; y < 2; y++) for (int x = 0; x < 2; x++) {
        const float* p = px + (y * 4) * stride + x * 4;
        float p11 = p[stride + 1];
        float rs = 0;
        for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++) {
          if ((iy == 0 && ix == 0)) continue;
          float v = (iy == 1 && ix == 1) ? p[0] - p11 : p[iy * stride + ix] - p11;
          coef[(y + iy * 2) * 8 + x + ix * 2] = v;
          rs += v;
        }
        dcs[y * 2 + x] = p11 + rs / 16.0f;
      }
      coef[0] = (dcs[0] + dcs[1] + dcs[2] + dcs[3]) / 4; coef[1] = (dcs[0] + dcs[1] - dcs[2] - dcs[3]) / 4;
      coef[8] = (dcs[0] - dcs[1] + dcs[2] - dcs[3]) / 4; coef[9] = (dcs[0] - dcs[1] - dcs[2] + dcs[3]) / 4;
      return;
    }
    case S_DCT2X2: {
      float a[64], b[64];
      for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) a[y * 8 + x] = px[y * stride + x];
      for (int S = 8; S >= 2; S /= 2) {
        const int n = S / 2;
        memcpy(b, a, sizeof(a));
        for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) {
          float r00 = a[2 * y * 8 + 2 * x], r01 = a[2 * y * 8 + 2 * x + 1], r10 = a[(2 * y + 1) * 8 + 2 * x], r11 = a[(2 * y + 1) * 8 + 2 * x + 1];
          b[y * 8 + x] = (r00 + r01 + r10 + r11) / 4; b[y * 8 + n + x] = (r00 + r01 - r10 - r11) / 4;
          b[(y + n) * 8 + x] = (r00 - r01 + r10 - r11) / 4; b[(y + n) * 8 + n + x] = (r00 - r01 - r10 + r11) / 4;
        }
        memcpy(a, b, sizeof(a));
      }
      memcpy(coef, a, sizeof(a));
      return;
    }
    case S_DCT4X4: {
      float dcs[4];
      for (int y = 0; y < 2; y++) for (int x = 0; x < 2; x++) {
        float sem[16];
        FDCT2D(px + y * 4 * stride + x * 4, stride, 4, 4, sem);
        for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++) {
          // stored block index iy*4+ix corresponds to (u=iy, v=ix)
          float v = sem[ix * 4 + iy];
          if (iy == 0 && ix == 0) dcs[y * 2 + x] = v; else coef[(y + iy * 2) * 8 + x + ix * 2] = v;
        }
      }
      coef[0] = (dcs[0] + dcs[1] + dcs[2] + dcs[3]) / 4; coef[1] = (dcs[0] + dcs[1] - dcs[2] - dcs[3]) / 4;
      coef[8] = (dcs[0] - dcs[1] + dcs[2] - dcs[3]) / 4; coef[9] = (dcs[0] - dcs[1] - dcs[2] + dcs[3]) / 4;
      return;
    }
    case S_DCT4X8: {
      float dcs[2];
      for (int y = 0; y < 2; y++) {
        float sem[32];
        FDCT2D(px + y * 4 * stride, stride, 4, 8, sem);  // [v*8+u] == stored layout for R<C
        for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 8; ix++) {
          if (iy == 0 && ix == 0) dcs[y] = sem[0]; else coef[(y + iy * 2) * 8 + ix] = sem[iy * 8 + ix];
        }
      }
      coef[0] = (dcs[0] + dcs[1]) / 2; coef[8] = (dcs[0] - dcs[1]) / 2;
      return;
    }
    case S_DCT8X4: {
      float dcs[2];
      for (int x = 0; x < 2; x++) {
        float sem[32];
        FDCT2D(px + x * 4, stride, 8, 4, sem);  // [v*4+u]; stored = [u*8+v]
        for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 8; ix++) {
          float v = sem[ix * 4 + iy];
          if (iy == 0 && ix == 0) dcs[x] = v; else coef[(x + iy * 2) * 8 + ix] = v;
        }
      }
      coef[0] = (dcs[0] + dcs[1]) / 2; coef[8] = (dcs[0] - dcs[1]) / 2;
      return;
    }
    case S_AFV0: case S_AFV1: case S_AFV2: case S_AFV3: {
      // forward of dec_transforms-inl.h AFVTransformToPixels: 4x4 corner in the (orthonormal) AFV basis, 4x4 DCT beside it,
      // 4x8 DCT of the other half; the three means are mixed into coefficients (0,0), (0,1), (1,0)
      const int afv_x = (s - S_AFV0) & 1, afv_y = (s - S_AFV0) >> 1;
      float pix[16], cf[16];
      for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++)
        pix[(afv_y == 1 ? 3 - iy : iy) * 4 + (afv_x == 1 ? 3 - ix : ix)] = px[(iy + afv_y * 4) * stride + afv_x * 4 + ix];
      for (int j = 0; j < 16; j++) { float a = 0; for (int i = 0; i < 16; i++) a += kAFVBasis[j][i] * pix[i]; cf[j] = a; }
      for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++) if (iy || ix) coef[iy * 2 * 8 + ix * 2] = cf[iy * 4 + ix];
      const float a = cf[0] / 4.0f;
      float sem[32];
      FDCT2D(px + afv_y * 4 * stride + (afv_x == 1 ? 0 : 4), stride, 4, 4, sem);
      const float m1 = sem[0];
      for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++) if (iy || ix) coef[iy * 2 * 8 + ix * 2 + 1] = sem[ix * 4 + iy];
      FDCT2D(px + (afv_y == 1 ? 0 : 4) * stride, stride, 4, 8, sem);
      const float m2 = sem[0];
      for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 8; ix++) if (iy || ix) coef[(1 + iy * 2) * 8 + ix] = sem[iy * 8 + ix];
      const float sum = (a + m1) / 2;             // = b00 + b10
      coef[0] = (sum + m2) / 2; coef[8] = (sum - m2) / 2; coef[1] = (a - m1) / 2;
      return;
    }
    default: {
      std::vector<float> sem((size_t)R * C);
      FDCT2D(px, stride, R, C, sem.data());
      for (int v = 0; v < R; v++) for (int u = 0; u < C; u++) coef[StoredIdx(R, C, v, u)] = sem[(size_t)v * C + u];
      return;
    }
  }
}

// LF samples (cy x cx) of a varblock from its lowest-frequency coefficients (enc side of LowestFrequenciesFromDC)
inline void LFFromLowestFrequencies(int s, const float* coef, float* lf, int lf_stride) {
  const int cx = kCovX[s], cy = kCovY[s], R = 8 * cy, C = 8 * cx;
  if (cx == 1 && cy == 1) { lf[0] = coef[0]; return; }
  std::vector<float> c((size_t)cx * cy);
  for (int v = 0; v < cy; v++) for (int u = 0; u < cx; u++) c[(size_t)v * cx + u] = coef[StoredIdx(R, C, v, u)] / (Resample(cy, v) * Resample(cx, u));
  IDCT2D(c.data(), cy, cx, lf, lf_stride);
}

// natural coefficient order (same definition as the codestream's)
inline std::vector<uint32_t> NaturalOrder(int s) {
  int cx = kCovX[s], cy = kCovY[s];
  if (cy > cx) std::swap(cx, cy);
  const int xs = cx * 8, ratio = cx / cy, lr = ILog2(ratio), mask = ratio - 1;
  std::vector<uint32_t> out((size_t)cx * cy * 64);
  size_t cur = (size_t)cx * cy;
  for (int i = 0; i < xs; i++) for (int j = 0; j <= i; j++) {
    int x = j, y = i - j;
    if (i & 1) std::swap(x, y);
    if (y & mask) continue;
    y >>= lr;
    size_t val = (x < cx && y < cy) ? (size_t)y * cx + x : cur++;
    out[val] = (uint32_t)(y * xs + x);
  }
  for (int ip = xs - 1; ip > 0; ip--) {
    int i = ip - 1;
    for (int j = 0; j <= i; j++) {
      int x = xs - 1 - (i - j), y = xs - 1 - j;
      if (i & 1) std::swap(x, y);
      if (y & mask) continue;
      y >>= lr;
      out[cur++] = (uint32_t)(y * xs + x);
    }
  }
  return out;
}

// ---- quantisation tables signalled explicitly in the stream ----------------------------------------------------------
struct Bands { int n; float v[3][17]; };
struct QuantSpec {
  int mode = 0;  // 0 = library default (not used by streams we emit for used kinds)
  Bands dct;
  float idw[3][3];
  float dct2w[3][6];
  float dct4mul[3][2];
  float dct4x8mul[3];
  float afvw[3][9];
  Bands dct4x4;
};

inline float BandMul(float v) { return v > 0 ? 1.0f + v : 1.0f / (1.0f - v); }
inline void BandWeights(const Bands& p, int c, int ROWS, int COLS, float* out) {
  float bands[17];
  bands[0] = p.v[c][0];
  for (int i = 1; i < p.n; i++) bands[i] = bands[i - 1] * BandMul(p.v[c][i]);
  float scale = (p.n - 1) / (kS2 + 1e-6f);
  float rcpcol = scale / (COLS - 1), rcprow = scale / (ROWS - 1);
  for (int y = 0; y < ROWS; y++) for (int x = 0; x < COLS; x++) {
    float dx = x * rcpcol, dy = y * rcprow;
    float dist = std::sqrt(dx * dx + dy * dy);
    float w;
    if (p.n == 1) w = bands[0];
    else { int idx = (int)dist; if (idx + 1 >= p.n) idx = p.n - 2; float frac = dist - idx; w = bands[idx] * std::pow(bands[idx + 1] / bands[idx], frac); }
    out[y * COLS + x] = w;
  }
}

// Parameters (rounded to F16 so the stream carries them exactly).  First band is stored /64 in the stream.
inline Bands MakeBands(int n, const float* x, const float* y, const float* b) {
  Bands r; r.n = n;
  const float* src[3] = {x, y, b};
  for (int c = 0; c < 3; c++) for (int i = 0; i < n; i++) r.v[c][i] = i == 0 ? RoundToHalf(src[c][i] / 64.0f) * 64.0f : RoundToHalf(src[c][i]);
  return r;
}

inline QuantSpec DefaultSpec(int kind) {
  QuantSpec q;
  auto B = [&](int n, std::initializer_list<float> x, std::initializer_list<float> y, std::initializer_list<float> b) {
    std::vector<float> vx(x), vy(y), vb(b);
    q.dct = MakeBands(n, vx.data(), vy.data(), vb.data());
  };
  switch (kind) {
    case 0: q.mode = 6; B(6, {3150.0f, 0.0f, -0.4f, -0.4f, -0.4f, -2.0f}, {560.0f, 0.0f, -0.3f, -0.3f, -0.3f, -0.3f}, {512.0f, -2.0f, -1.0f, 0.0f, -1.0f, -2.0f}); break;
    case 1: {
      q.mode = 1;
      float w[3][3] = {{280.0f, 3160.0f, 3160.0f}, {60.0f, 864.0f, 864.0f}, {18.0f, 200.0f, 200.0f}};
      for (int c = 0; c < 3; c++) for (int i = 0; i < 3; i++) q.idw[c][i] = RoundToHalf(w[c][i] / 64.0f) * 64.0f;
      break;
    }
    case 2: {
      q.mode = 2;
      float w[3][6] = {{3840.0f, 2560.0f, 1280.0f, 640.0f, 480.0f, 300.0f}, {960.0f, 640.0f, 320.0f, 180.0f, 140.0f, 120.0f}, {640.0f, 320.0f, 128.0f, 64.0f, 32.0f, 16.0f}};
      for (int c = 0; c < 3; c++) for (int i = 0; i < 6; i++) q.dct2w[c][i] = RoundToHalf(w[c][i] / 64.0f) * 64.0f;
      break;
    }
    case 3:
      q.mode = 3; B(4, {2200.0f, 0.0f, 0.0f, 0.0f}, {392.0f, 0.0f, 0.0f, 0.0f}, {112.0f, -0.25f, -0.25f, -0.5f});
      for (int c = 0; c < 3; c++) { q.dct4mul[c][0] = 1.0f; q.dct4mul[c][1] = 1.0f; }
      break;
    case 4: q.mode = 6; B(7, {8996.87f, -1.3f, -0.494f, -0.439f, -0.635f, -0.9018f, -1.616f}, {3191.48f, -0.674f, -0.8075f, -0.4493f, -0.3587f, -0.3132f, -0.3762f}, {1157.5f, -2.053f, -1.4f, -0.5069f, -0.4271f, -1.4857f, -4.921f}); break;
    case 5: q.mode = 6; B(8, {15718.4f, -1.025f, -0.98f, -0.9012f, -0.4f, -0.4882f, -0.4211f, -0.27f}, {7305.76f, -0.8042f, -0.7633f, -0.5566f, -0.4979f, -0.437f, -0.4018f, -0.2732f}, {3803.53f, -3.0607f, -2.0413f, -2.0236f, -0.5495f, -0.4f, -0.4f, -0.3f}); break;
    case 6: q.mode = 6; B(7, {7240.77f, -0.7f, -0.7f, -0.2f, -0.2f, -0.2f, -0.5f}, {1448.15f, -0.5f, -0.5f, -0.5f, -0.2f, -0.2f, -0.2f}, {506.854f, -1.4f, -0.2f, -0.5f, -0.5f, -1.5f, -3.6f}); break;
    case 7: q.mode = 6; B(8, {16283.25f, -1.7813f, -1.6309f, -1.0382f, -0.85f, -0.7f, -0.9f, -1.2361f}, {5089.16f, -0.32f, -0.3536f, -0.3034f, -0.61f, -0.5f, -0.5f, -0.6f}, {3397.78f, -0.3213f, -0.3451f, -0.7034f, -0.9f, -1.0f, -1.0f, -1.1755f}); break;
    case 8: q.mode = 6; B(8, {13844.97f, -0.9711f, -0.658f, -0.4203f, -0.2271f, -0.2206f, -0.226f, -0.6f}, {4798.96f, -0.6113f, -0.8377f, -0.7901f, -0.2693f, -0.3827f, -0.2292f, -0.2072f}, {1807.24f, -1.2f, -1.2f, -0.7f, -0.7f, -0.7f, -0.4f, -0.5f}); break;
    case 9:
      q.mode = 4; B(4, {2198.05f, -0.9627f, -0.7619f, -0.6551f}, {764.366f, -0.9263f, -0.9675f, -0.2785f}, {527.108f, -1.4594f, -1.4501f, -1.5844f});
      for (int c = 0; c < 3; c++) q.dct4x8mul[c] = 1.0f;
      break;
    case 10: {  // AFV: corner weights, DCT4X8 bands, DCT4X4 bands
      q.mode = 5;
      const float w[3][9] = {{3072.0f, 3072.0f, 256.0f, 256.0f, 256.0f, 414.0f, 0.0f, 0.0f, 0.0f}, {1024.0f, 1024.0f, 50.0f, 50.0f, 50.0f, 58.0f, 0.0f, 0.0f, 0.0f},
                             {384.0f, 384.0f, 12.0f, 12.0f, 12.0f, 22.0f, -0.25f, -0.25f, -0.25f}};
      for (int c = 0; c < 3; c++) for (int i = 0; i < 9; i++) q.afvw[c][i] = i < 6 ? RoundToHalf(w[c][i] / 64.0f) * 64.0f : RoundToHalf(w[c][i]);
      B(4, {2200.0f, 0.0f, 0.0f, 0.0f}, {392.0f, 0.0f, 0.0f, 0.0f}, {112.0f, -0.25f, -0.25f, -0.5f});
      q.dct4x4 = q.dct;
      B(4, {2198.05f, -0.9627f, -0.7619f, -0.6551f}, {764.366f, -0.9263f, -0.9675f, -0.2785f}, {527.108f, -1.4594f, -1.4501f, -1.5844f});
      break;
    }
    case 11: q.mode = 6; B(8, {0.9f * 26629.07f, -1.025f, -0.78f, -0.6501f, -0.1904f, -0.2082f, -0.4211f, -0.3273f}, {0.9f * 9311.32f, -0.3042f, -0.3633f, -0.3566f, -0.3443f, -0.337f, -0.3018f, -0.2732f}, {0.9f * 4992.25f, -1.2f, -1.2f, -0.8f, -0.7f, -0.7f, -0.4f, -0.5f}); break;
    case 12: q.mode = 6; B(8, {0.65f * 23629.07f, -1.025f, -0.78f, -0.6501f, -0.1904f, -0.2082f, -0.4211f, -0.3273f}, {0.65f * 8611.32f, -0.3042f, -0.3633f, -0.3566f, -0.3443f, -0.337f, -0.3018f, -0.2732f}, {0.65f * 4492.25f, -1.2f, -1.2f, -0.8f, -0.7f, -0.7f, -0.4f, -0.5f}); break;
    case 13: q.mode = 6; B(8, {1.8f * 23966.17f, -1.025f, -0.78f, -0.6501f, -0.1904f, -0.2082f, -0.4211f, -0.3273f}, {1.8f * 8380.19f, -0.3042f, -0.3633f, -0.3566f, -0.3443f, -0.337f, -0.3018f, -0.2732f}, {1.8f * 4493.02f, -1.2f, -1.2f, -0.8f, -0.7f, -0.7f, -0.4f, -0.5f}); break;
    case 14: q.mode = 6; B(8, {1.3f * 15358.9f, -1.025f, -0.78f, -0.6501f, -0.1904f, -0.2082f, -0.4211f, -0.3273f}, {1.3f * 5597.36f, -0.3042f, -0.3633f, -0.3566f, -0.3443f, -0.337f, -0.3018f, -0.2732f}, {1.3f * 2919.96f, -1.2f, -1.2f, -0.8f, -0.7f, -0.7f, -0.4f, -0.5f}); break;
    case 15: q.mode = 6; B(8, {3.6f * 23966.17f, -1.025f, -0.78f, -0.6501f, -0.1904f, -0.2082f, -0.4211f, -0.3273f}, {3.6f * 8380.19f, -0.3042f, -0.3633f, -0.3566f, -0.3443f, -0.337f, -0.3018f, -0.2732f}, {3.6f * 4493.02f, -1.2f, -1.2f, -0.8f, -0.7f, -0.7f, -0.4f, -0.5f}); break;
    case 16: q.mode = 6; B(8, {2.6f * 15358.9f, -1.025f, -0.78f, -0.6501f, -0.1904f, -0.2082f, -0.4211f, -0.3273f}, {2.6f * 5597.36f, -0.3042f, -0.3633f, -0.3566f, -0.3443f, -0.337f, -0.3018f, -0.2732f}, {2.6f * 2919.96f, -1.2f, -1.2f, -0.8f, -0.7f, -0.7f, -0.4f, -0.5f}); break;
    default: q.mode = 0; break;
  }
  return q;
}

// dequant table (1/weight) for channel c of kind, stored layout
inline void ComputeTable(const QuantSpec& q, int kind, int c, std::vector<float>& out) {
  const int ROWS = 8 * kKindR[kind], COLS = 8 * kKindC[kind];
  std::vector<float> w((size_t)ROWS * COLS, 1.0f);
  switch (q.mode) {
    case 6: BandWeights(q.dct, c, ROWS, COLS, w.data()); break;
    case 1: for (int i = 0; i < 64; i++) w[i] = q.idw[c][0]; w[1] = w[8] = q.idw[c][1]; w[9] = q.idw[c][2]; break;
    case 2: {
      const float* d = q.dct2w[c];
      w[0] = 1e6f; w[1] = w[8] = d[0]; w[9] = d[1];
      for (int y = 0; y < 2; y++) for (int x = 0; x < 2; x++) { w[y * 8 + x + 2] = d[2]; w[(y + 2) * 8 + x] = d[2]; w[(y + 2) * 8 + x + 2] = d[3]; }
      for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) { w[y * 8 + x + 4] = d[4]; w[(y + 4) * 8 + x] = d[4]; w[(y + 4) * 8 + x + 4] = d[5]; }
      break;
    }
    case 3: {
      float w4[16]; BandWeights(q.dct, c, 4, 4, w4);
      for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) w[y * 8 + x] = w4[(y / 2) * 4 + x / 2];
      w[1] /= q.dct4mul[c][0]; w[8] /= q.dct4mul[c][0]; w[9] /= q.dct4mul[c][1];
      break;
    }
    case 4: {
      float w48[32]; BandWeights(q.dct, c, 4, 8, w48);
      for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) w[y * 8 + x] = w48[(y / 2) * 8 + x];
      w[8] /= q.dct4x8mul[c];
      break;
    }
    case 5: {
      static const float kFreqs[16] = {0, 0, 0.8517778890324296f, 5.37778436506804f, 0, 0, 4.734747904497923f, 5.449245381693219f, 1.6598270267479331f, 4.0f,
                                       7.275749096817861f, 10.423227632456525f, 2.662932286148962f, 7.630657783650829f, 8.962388608184032f, 12.97166202570235f};
      float w48[32], w44[16], bands[4];
      BandWeights(q.dct, c, 4, 8, w48);
      BandWeights(q.dct4x4, c, 4, 4, w44);
      const float lo = 0.8517778890324296f, hi = 12.97166202570235f - lo + 1e-6f;
      bands[0] = q.afvw[c][5];
      for (int i = 1; i < 4; i++) bands[i] = bands[i - 1] * BandMul(q.afvw[c][i + 5]);
      w[0] = 1.0f; w[8] = q.afvw[c][0]; w[1] = q.afvw[c][1]; w[16] = q.afvw[c][2]; w[2] = q.afvw[c][3]; w[18] = q.afvw[c][4];
      for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
        if (x < 2 && y < 2) continue;
        const float sp = (kFreqs[y * 4 + x] - lo) * 3 / hi;
        const int idx = (int)sp;
        w[2 * y * 8 + 2 * x] = bands[idx] * std::pow(bands[idx + 1] / bands[idx], sp - idx);
      }
      for (int y = 0; y < 4; y++) for (int x = 0; x < 8; x++) if (x || y) w[(2 * y + 1) * 8 + x] = w48[y * 8 + x];
      for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) if (x || y) w[2 * y * 8 + 2 * x + 1] = w44[y * 4 + x];
      break;
    }
    default: throw std::runtime_error("quant kind without explicit spec used");
  }
  out.resize(w.size());
  for (size_t i = 0; i < w.size(); i++) out[i] = 1.0f / w[i];
}

}  // namespace synth
