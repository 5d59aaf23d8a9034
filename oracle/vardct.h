// ORACLE — TEST INFRASTRUCTURE ONLY (see bits.h).
// VarDCT numerics: restates libjxl v0.11.2 lib/jxl/{ac_strategy.{h,cc},coeff_order.cc,quant_weights.cc,
// dct-inl.h,dct_scales.h,dec_transforms-inl.h,dec_group.cc(dequant),compressed_dc.cc}.
// context model / orders / RAW tables are [V]; the float pipeline (tables, op order) is [R]
// — PARITY UNPINNED against libjxl for pixels; this file is the self-consistent definition both the
// synthesiser and the HIP kernels are checked against.  Round 2: AFV transforms + AFV weights, FastPowf band interpolation.
#pragma once
#include "bits.h"
#include <algorithm>
#include <cmath>

namespace jxlo {

// ---- AC strategies (ac_strategy.h) --------------------------------------------------------------------------------
enum Strategy {
  DCT = 0, IDENTITY, DCT2X2, DCT4X4, DCT16X16, DCT32X32, DCT16X8, DCT8X16, DCT32X8, DCT8X32, DCT32X16, DCT16X32,
  DCT4X8, DCT8X4, AFV0, AFV1, AFV2, AFV3, DCT64X64, DCT64X32, DCT32X64, DCT128X128, DCT128X64, DCT64X128,
  DCT256X256, DCT256X128, DCT128X256, kNumStrategies
};
// names are rows x cols; covered blocks
static const uint8_t kCoveredX[27] = {1, 1, 1, 1, 2, 4, 1, 2, 1, 4, 2, 4, 1, 1, 1, 1, 1, 1, 8, 4, 8, 16, 8, 16, 32, 16, 32};
static const uint8_t kCoveredY[27] = {1, 1, 1, 1, 2, 4, 2, 1, 4, 1, 4, 2, 1, 1, 1, 1, 1, 1, 8, 8, 4, 16, 16, 8, 32, 32, 16};
static const uint8_t kOrderBucket[27] = {0, 1, 1, 1, 2, 3, 4, 4, 5, 5, 6, 6, 1, 1, 1, 1, 1, 1, 7, 8, 8, 9, 10, 10, 11, 12, 12};
// quant table kind per strategy (quant_weights.h kQuantTable)
enum QuantKind { QDCT = 0, QIDENTITY, QDCT2X2, QDCT4X4, QDCT16X16, QDCT32X32, QDCT8X16, QDCT8X32, QDCT16X32, QDCT4X8, QAFV,
                 QDCT64X64, QDCT32X64, QDCT128X128, QDCT64X128, QDCT256X256, QDCT128X256, kNumQuantKinds };
static const uint8_t kQuantKind[27] = {QDCT, QIDENTITY, QDCT2X2, QDCT4X4, QDCT16X16, QDCT32X32, QDCT8X16, QDCT8X16, QDCT8X32, QDCT8X32,
                                       QDCT16X32, QDCT16X32, QDCT4X8, QDCT4X8, QAFV, QAFV, QAFV, QAFV, QDCT64X64, QDCT32X64, QDCT32X64,
                                       QDCT128X128, QDCT64X128, QDCT64X128, QDCT256X256, QDCT128X256, QDCT128X256};
// table dims in blocks (rows<=cols layout)
static const uint8_t kKindRows[17] = {1, 1, 1, 1, 2, 4, 1, 1, 2, 1, 1, 8, 4, 16, 8, 32, 16};
static const uint8_t kKindCols[17] = {1, 1, 1, 1, 2, 4, 2, 4, 4, 1, 1, 8, 8, 16, 16, 32, 32};
// order bucket sizes in blocks
static const uint16_t kOrderSizeBlocks[13] = {1, 1, 4, 16, 2, 4, 8, 64, 32, 256, 128, 1024, 512};

inline int Log2Int(int v) { int r = 0; while ((1 << r) < v) r++; return r; }

// coeff_order.cc / ac_strategy.cc natural order for a strategy (8x8 case [V], general [R])
inline std::vector<uint32_t> NaturalCoeffOrder(int strategy) {
  int cx = kCoveredX[strategy], cy = kCoveredY[strategy];
  if (cy > cx) std::swap(cx, cy);  // cols >= rows layout
  const int xs = cx * 8;
  const int ratio = cx / cy, lr = Log2Int(ratio), mask = ratio - 1;
  std::vector<uint32_t> out((size_t)cx * cy * 64);
  size_t cur = (size_t)cx * cy;
  for (int i = 0; i < xs; i++) {
    for (int j = 0; j <= i; j++) {
      int x = j, y = i - j;
      if (i & 1) std::swap(x, y);
      if (y & mask) continue;
      y >>= lr;
      size_t val = (x < cx && y < cy) ? (size_t)y * cx + x : cur++;
      out[val] = (uint32_t)(y * xs + x);
    }
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
  JXLO_CHECK(cur == out.size());
  return out;
}
// representative strategy of each of the 13 order buckets (all strategies of a bucket share the layout)
static const uint8_t kBucketStrategy[13] = {DCT, IDENTITY, DCT16X16, DCT32X32, DCT16X8, DCT32X8, DCT32X16, DCT64X64, DCT64X32,
                                            DCT128X128, DCT128X64, DCT256X256, DCT256X128};

// ---- DCT (dct-inl.h, dct_scales.h) ---------------------------------------------------------------------------------
// libjxl convention: forward F(0) = mean, F(k) = sqrt2/N * sum f(n) cos((2n+1)k pi/2N); inverse unscaled.
struct DctConsts {
  // WcMultipliers<N>[i] = 1 / (2 cos((i + 0.5) pi / N)), N = 2..256
  std::vector<float> wc[9];  // index log2(N)
  DctConsts() {
    for (int l = 1; l <= 8; l++) {
      int N = 1 << l;
      wc[l].resize(N / 2);
      for (int i = 0; i < N / 2; i++) wc[l][i] = (float)(1.0 / (2.0 * std::cos((i + 0.5) * M_PI / N)));
    }
  }
};
inline const DctConsts& dctc() { static DctConsts c; return c; }
static const float kSqrt2f = 1.41421356237309504880f;

// in-place inverse DCT of length N (power of two) on contiguous v; tmp needs 2N floats of scratch
inline void IDCT1D(float* v, int N, float* tmp) {
  if (N == 1) return;
  if (N == 2) { float a = v[0], b = v[1]; v[0] = a + b; v[1] = a - b; return; }
  const int H = N / 2;
  for (int i = 0; i < H; i++) { tmp[i] = v[2 * i]; tmp[H + i] = v[2 * i + 1]; }
  IDCT1D(tmp, H, tmp + N);
  for (int i = H - 1; i > 0; i--) tmp[H + i] = tmp[H + i] + tmp[H + i - 1];
  tmp[H] = tmp[H] * kSqrt2f;
  IDCT1D(tmp + H, H, tmp + N);
  const float* wc = dctc().wc[Log2Int(N)].data();
  for (int i = 0; i < H; i++) {
    float mul = wc[i], in1 = tmp[i], in2 = tmp[H + i];
    v[i] = std::fmaf(mul, in2, in1);
    v[N - 1 - i] = std::fmaf(-mul, in2, in1);
  }
}
// in-place forward DCT of length N without the 1/N scale
inline void DCT1DUnscaled(float* v, int N, float* tmp) {
  if (N == 1) return;
  if (N == 2) { float a = v[0], b = v[1]; v[0] = a + b; v[1] = a - b; return; }
  const int H = N / 2;
  for (int i = 0; i < H; i++) tmp[i] = v[i] + v[N - 1 - i];
  DCT1DUnscaled(tmp, H, tmp + N);
  const float* wc = dctc().wc[Log2Int(N)].data();
  for (int i = 0; i < H; i++) tmp[H + i] = (v[i] - v[N - 1 - i]) * wc[i];
  DCT1DUnscaled(tmp + H, H, tmp + N);
  tmp[H] = std::fmaf(tmp[H], kSqrt2f, tmp[H + 1]);
  for (int i = 1; i + 1 < H; i++) tmp[H + i] = tmp[H + i] + tmp[H + i + 1];
  for (int i = 0; i < H; i++) { v[2 * i] = tmp[i]; v[2 * i + 1] = tmp[H + i]; }
}

// 2-D inverse: coefficient block in semantic layout c[v*C+u] (v vertical freq, R rows; u horizontal, C cols),
// output pixels out[y*stride+x]. Horizontal pass first, then vertical (ComputeScaledIDCT).
inline void IDCT2D(const float* c, int R, int C, float* out, int stride) {
  std::vector<float> buf((size_t)R * C), col(R), tmp(4 * std::max(R, C));
  for (int v = 0; v < R; v++) {
    float* row = &buf[(size_t)v * C];
    for (int u = 0; u < C; u++) row[u] = c[(size_t)v * C + u];
    IDCT1D(row, C, tmp.data());
  }
  for (int x = 0; x < C; x++) {
    for (int v = 0; v < R; v++) col[v] = buf[(size_t)v * C + x];
    IDCT1D(col.data(), R, tmp.data());
    for (int y = 0; y < R; y++) out[(size_t)y * stride + x] = col[y];
  }
}
// 2-D forward scaled DCT: pixels in[y*stride+x] -> c[v*C+u]; vertical pass first then horizontal (ComputeScaledDCT)
inline void DCT2D(const float* in, int stride, int R, int C, float* c) {
  std::vector<float> buf((size_t)R * C), col(R), tmp(4 * std::max(R, C));
  const float sr = 1.0f / R, sc = 1.0f / C;
  for (int x = 0; x < C; x++) {
    for (int y = 0; y < R; y++) col[y] = in[(size_t)y * stride + x];
    DCT1DUnscaled(col.data(), R, tmp.data());
    for (int v = 0; v < R; v++) buf[(size_t)v * C + x] = col[v] * sr;
  }
  for (int v = 0; v < R; v++) {
    float* row = &buf[(size_t)v * C];
    DCT1DUnscaled(row, C, tmp.data());
    for (int u = 0; u < C; u++) c[(size_t)v * C + u] = row[u] * sc;
  }
}

// stored ("cols >= rows") layout index of semantic coefficient (v,u) of an R x C transform
inline size_t StoredIndex(int R, int C, int v, int u) { return R >= C ? (size_t)u * R + v : (size_t)v * C + u; }

// dct_scales.h DCTTotalResampleScale<N, 8N>(k) = (1/8) sin(k pi / 2N) / sin(k pi / 16N)
inline float ResampleScale(int N, int k) {
  if (k == 0) return 1.0f;
  return (float)(std::sin(k * M_PI / (2.0 * N)) / std::sin(k * M_PI / (16.0 * N)) / 8.0);
}

// dec_transforms-inl.h LowestFrequenciesFromDC: fills the LLF slots of a stored-layout block from the cy x cx LF samples
inline void LowestFrequenciesFromLF(int strategy, const float* lf, int lf_stride, float* block) {
  int cx = kCoveredX[strategy], cy = kCoveredY[strategy];
  if (cx == 1 && cy == 1) { block[0] = lf[0]; return; }
  std::vector<float> c((size_t)cx * cy);
  DCT2D(lf, lf_stride, cy, cx, c.data());
  const int R = 8 * cy, C = 8 * cx;
  for (int v = 0; v < cy; v++)
    for (int u = 0; u < cx; u++)
      block[StoredIndex(R, C, v, u)] = c[(size_t)v * cx + u] * ResampleScale(cy, v) * ResampleScale(cx, u);
}


// dec_transforms-inl.h AFVIDCT4x4: the 16 basis functions of the 4x4 "corner" transform (ISO/IEC 18181-1 AFV basis).
// Recalled digits, verified: the 16x16 matrix is orthonormal to 1.5e-14 (tests/test_oracle_goldens.py), row 0 is the
// constant function, every row is symmetric or antisymmetric under transposition of the 4x4 block.
static const float k4x4AFVBasis[16][16] = {
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
inline void AFVIDCT4x4(const float* coeffs, float* pixels) {
  for (int i = 0; i < 16; i++) {
    float px = 0.0f;
    for (int j = 0; j < 16; j++) px = std::fmaf(coeffs[j], k4x4AFVBasis[j][i], px);
    pixels[i] = px;
  }
}

// dec_transforms-inl.h TransformToPixels. coeffs: stored layout, size covered*64; out: pixel block.
inline void InverseTransform(int strategy, const float* coeffs, float* out, int stride) {
  const int cx = kCoveredX[strategy], cy = kCoveredY[strategy];
  const int R = 8 * cy, C = 8 * cx;
  switch (strategy) {
    case IDENTITY: {
      float dcs[4];
      float b00 = coeffs[0], b01 = coeffs[1], b10 = coeffs[8], b11 = coeffs[9];
      dcs[0] = b00 + b01 + b10 + b11; dcs[1] = b00 + b01 - b10 - b11;
      dcs[2] = b00 - b01 + b10 - b11; dcs[3] = b00 - b01 - b10 + b11;
      for (int y = 0; y < 2; y++) for (int x = 0; x < 2; x++) {
        float block_dc = dcs[y * 2 + x];
        float residual_sum = 0;
        for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++) {
          if (ix == 0 && iy == 0) continue;
          residual_sum += coeffs[(y + iy * 2) * 8 + x + ix * 2];
        }
        out[(4 * y + 1) * stride + 4 * x + 1] = block_dc - residual_sum * (1.0f / 16);
        for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++) {
          if (ix == 1 && iy == 1) continue;
          out[(y * 4 + iy) * stride + x * 4 + ix] = coeffs[(y + iy * 2) * 8 + x + ix * 2] + out[(4 * y + 1) * stride + 4 * x + 1];
        }
        out[y * 4 * stride + x * 4] = coeffs[(y + 2) * 8 + x + 2] + out[(4 * y + 1) * stride + 4 * x + 1];
      }
      return;
    }
    case DCT2X2: {
      float a[64], b[64];
      for (int i = 0; i < 64; i++) a[i] = coeffs[i];
      for (int S = 2; S <= 8; S *= 2) {
        const int n = S / 2;
        for (int i = 0; i < 64; i++) b[i] = a[i];
        for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) {
          float c00 = a[y * 8 + x], c01 = a[y * 8 + n + x], c10 = a[(y + n) * 8 + x], c11 = a[(y + n) * 8 + n + x];
          float r00 = c00 + c01 + c10 + c11, r01 = c00 + c01 - c10 - c11, r10 = c00 - c01 + c10 - c11, r11 = c00 - c01 - c10 + c11;
          b[y * 2 * 8 + x * 2] = r00; b[y * 2 * 8 + x * 2 + 1] = r01;
          b[(y * 2 + 1) * 8 + x * 2] = r10; b[(y * 2 + 1) * 8 + x * 2 + 1] = r11;
        }
        for (int i = 0; i < 64; i++) a[i] = b[i];
      }
      for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) out[y * stride + x] = a[y * 8 + x];
      return;
    }
    case DCT4X4: {
      float dcs[4];
      float b00 = coeffs[0], b01 = coeffs[1], b10 = coeffs[8], b11 = coeffs[9];
      dcs[0] = b00 + b01 + b10 + b11; dcs[1] = b00 + b01 - b10 - b11;
      dcs[2] = b00 - b01 + b10 - b11; dcs[3] = b00 - b01 - b10 + b11;
      for (int y = 0; y < 2; y++) for (int x = 0; x < 2; x++) {
        float blk[16];  // stored layout of a 4x4 (R>=C): blk[u*4+v]
        blk[0] = dcs[y * 2 + x];
        for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++) {
          if (ix == 0 && iy == 0) continue;
          blk[iy * 4 + ix] = coeffs[(y + iy * 2) * 8 + x + ix * 2];
        }
        float sem[16];
        for (int v = 0; v < 4; v++) for (int u = 0; u < 4; u++) sem[v * 4 + u] = blk[StoredIndex(4, 4, v, u)];
        IDCT2D(sem, 4, 4, out + y * 4 * stride + x * 4, stride);
      }
      return;
    }
    case DCT4X8: {  // two 4-row x 8-col halves stacked vertically
      float b0 = coeffs[0], b1 = coeffs[8];
      float dcs[2] = {b0 + b1, b0 - b1};
      for (int y = 0; y < 2; y++) {
        float blk[32];  // 4x8: R<C stored [v*8+u]
        blk[0] = dcs[y];
        for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 8; ix++) {
          if (ix == 0 && iy == 0) continue;
          blk[iy * 8 + ix] = coeffs[(y + iy * 2) * 8 + ix];
        }
        IDCT2D(blk, 4, 8, out + y * 4 * stride, stride);
      }
      return;
    }
    case DCT8X4: {  // two 8-row x 4-col halves side by side
      float b0 = coeffs[0], b1 = coeffs[8];
      float dcs[2] = {b0 + b1, b0 - b1};
      for (int x = 0; x < 2; x++) {
        float blk[32];  // 8x4: R>=C stored as [u*8+v] (4 rows of 8)
        blk[0] = dcs[x];
        for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 8; ix++) {
          if (ix == 0 && iy == 0) continue;
          blk[iy * 8 + ix] = coeffs[(x + iy * 2) * 8 + ix];
        }
        float sem[32];
        for (int v = 0; v < 8; v++) for (int u = 0; u < 4; u++) sem[v * 4 + u] = blk[StoredIndex(8, 4, v, u)];
        IDCT2D(sem, 8, 4, out + x * 4, stride);
      }
      return;
    }
    case AFV0: case AFV1: case AFV2: case AFV3: {  // dec_transforms-inl.h AFVTransformToPixels<afv_kind>
      const int afv_kind = strategy - AFV0, afv_x = afv_kind & 1, afv_y = afv_kind / 2;
      const float block00 = coeffs[0], block01 = coeffs[1], block10 = coeffs[8];
      const float dcs[3] = {(block00 + block10 + block01) * 4.0f, (block00 + block10 - block01), block00 - block10};
      float coeff[16], block[32];
      coeff[0] = dcs[0];
      for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++) { if (ix == 0 && iy == 0) continue; coeff[iy * 4 + ix] = coeffs[iy * 2 * 8 + ix * 2]; }
      AFVIDCT4x4(coeff, block);
      for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++)
        out[(iy + afv_y * 4) * stride + afv_x * 4 + ix] = block[(afv_y == 1 ? 3 - iy : iy) * 4 + (afv_x == 1 ? 3 - ix : ix)];
      // 4x4 DCT of the (even row, odd column) coefficients next to the corner
      block[0] = dcs[1];
      for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++) { if (ix == 0 && iy == 0) continue; block[iy * 4 + ix] = coeffs[iy * 2 * 8 + ix * 2 + 1]; }
      {
        float sem[16];
        for (int v = 0; v < 4; v++) for (int u = 0; u < 4; u++) sem[v * 4 + u] = block[StoredIndex(4, 4, v, u)];
        IDCT2D(sem, 4, 4, out + afv_y * 4 * stride + (afv_x == 1 ? 0 : 4), stride);
      }
      // 4x8 DCT of the odd rows in the other half
      block[0] = dcs[2];
      for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 8; ix++) { if (ix == 0 && iy == 0) continue; block[iy * 8 + ix] = coeffs[(1 + iy * 2) * 8 + ix]; }
      IDCT2D(block, 4, 8, out + (afv_y == 1 ? 0 : 4) * stride, stride);
      return;
    }
    default: {
      std::vector<float> sem((size_t)R * C);
      for (int v = 0; v < R; v++) for (int u = 0; u < C; u++) sem[(size_t)v * C + u] = coeffs[StoredIndex(R, C, v, u)];
      IDCT2D(sem.data(), R, C, out, stride);
      return;
    }
  }
}

// ---- dequantisation matrices (quant_weights.cc) ---------------------------------------------------------------------
struct DctBandParams { int num_bands = 0; float bands[3][17] = {{0}}; };
struct QuantEncoding {
  int mode = 0;  // 0 library ... 7 raw
  float idweights[3][3];
  float dct2weights[3][6];
  float dct4multipliers[3][2];
  float dct4x8multipliers[3];
  float afv_weights[3][9];
  DctBandParams dct, dct4x4;
  float raw_den = 0;
  std::vector<int32_t> raw[3];  // X,Y,B
};

inline DctBandParams MakeBands(int n, std::initializer_list<float> x, std::initializer_list<float> y, std::initializer_list<float> b) {
  DctBandParams p; p.num_bands = n;
  int i = 0; for (float v : x) p.bands[0][i++] = v;
  i = 0; for (float v : y) p.bands[1][i++] = v;
  i = 0; for (float v : b) p.bands[2][i++] = v;
  return p;
}

// library defaults [R] (quant_weights.cc DequantMatricesLibraryDef)
inline QuantEncoding LibraryQuant(int kind) {
  QuantEncoding q;
  auto seq = [](float first, std::initializer_list<float> rest) { std::vector<float> v{first}; v.insert(v.end(), rest); return v; };
  (void)seq;
  switch (kind) {
    case QDCT:
      q.mode = 6;
      q.dct = MakeBands(6, {3150.0f, 0.0f, -0.4f, -0.4f, -0.4f, -2.0f}, {560.0f, 0.0f, -0.3f, -0.3f, -0.3f, -0.3f}, {512.0f, -2.0f, -1.0f, 0.0f, -1.0f, -2.0f});
      break;
    case QIDENTITY: {
      q.mode = 1;
      float w[3][3] = {{280.0f, 3160.0f, 3160.0f}, {60.0f, 864.0f, 864.0f}, {18.0f, 200.0f, 200.0f}};
      memcpy(q.idweights, w, sizeof(w));
      break;
    }
    case QDCT2X2: {
      q.mode = 2;
      float w[3][6] = {{3840.0f, 2560.0f, 1280.0f, 640.0f, 480.0f, 300.0f}, {960.0f, 640.0f, 320.0f, 180.0f, 140.0f, 120.0f}, {640.0f, 320.0f, 128.0f, 64.0f, 32.0f, 16.0f}};
      memcpy(q.dct2weights, w, sizeof(w));
      break;
    }
    case QDCT4X4: {
      q.mode = 3;
      q.dct = MakeBands(4, {2200.0f, 0.0f, 0.0f, 0.0f}, {392.0f, 0.0f, 0.0f, 0.0f}, {112.0f, -0.25f, -0.25f, -0.5f});
      for (int c = 0; c < 3; c++) { q.dct4multipliers[c][0] = 1.0f; q.dct4multipliers[c][1] = 1.0f; }
      break;
    }
    case QDCT16X16:
      q.mode = 6;
      q.dct = MakeBands(7, {8996.8725711814115328f, -1.3000777393353804f, -0.49424529824571225f, -0.439093774457103443f, -0.6350101832695744f, -0.90177264050827612f, -1.6162099239887414f},
                        {3191.48366296844234752f, -0.67424582104194355f, -0.80745813428471001f, -0.44925837484843441f, -0.35865440981033403f, -0.31322389111877305f, -0.37615025315725483f},
                        {1157.50408145487200256f, -2.0531423165804414f, -1.4f, -0.50687130033378396f, -0.42708730624733904f, -1.4856834539296244f, -4.9209142884401604f});
      break;
    case QDCT32X32:
      q.mode = 6;
      q.dct = MakeBands(8, {15718.40830982518931456f, -1.025f, -0.98f, -0.9012f, -0.4f, -0.48819395464f, -0.421064f, -0.27f},
                        {7305.7636810695983104f, -0.8041958212306401f, -0.7633036457487539f, -0.55660379990111464f, -0.49785304658857626f, -0.43699592683512467f, -0.40180866526242109f, -0.27321683125358037f},
                        {3803.53173721215041536f, -3.060733579805728f, -2.0413270132490346f, -2.0235650159727417f, -0.5495389509954993f, -0.4f, -0.4f, -0.3f});
      break;
    case QDCT8X16:
      q.mode = 6;
      q.dct = MakeBands(7, {7240.7734393502f, -0.7f, -0.7f, -0.2f, -0.2f, -0.2f, -0.5f}, {1448.15468787004f, -0.5f, -0.5f, -0.5f, -0.2f, -0.2f, -0.2f},
                        {506.854140754517f, -1.4f, -0.2f, -0.5f, -0.5f, -1.5f, -3.6f});
      break;
    case QDCT8X32:
      q.mode = 6;
      q.dct = MakeBands(8, {16283.2494710648897f, -1.7812845336559429f, -1.6309059012653515f, -1.0382179034313539f, -0.85f, -0.7f, -0.9f, -1.2360638576849587f},
                        {5089.15750884921511936f, -0.320049391452786891f, -0.35362849922161446f, -0.30340000000000003f, -0.61f, -0.5f, -0.5f, -0.6f},
                        {3397.77603275308720128f, -0.321327362693153371f, -0.34507619223117997f, -0.70340000000000003f, -0.9f, -1.0f, -1.0f, -1.1754605576265209f});
      break;
    case QDCT16X32:
      q.mode = 6;
      q.dct = MakeBands(8, {13844.97076442300573f, -0.97113799999999995f, -0.658f, -0.42026f, -0.22712f, -0.2206f, -0.226f, -0.6f},
                        {4798.964084220744293f, -0.61125308982767057f, -0.83770786552491361f, -0.79014862079498627f, -0.2692727459704829f, -0.38272769465388551f, -0.22924222653091453f, -0.20719098826199578f},
                        {1807.236946760964614f, -1.2f, -1.2f, -0.7f, -0.7f, -0.7f, -0.4f, -0.5f});
      break;
    case QDCT4X8:
      q.mode = 4;
      q.dct = MakeBands(4, {2198.050556016380522f, -0.96269623020744692f, -0.76194253026666783f, -0.6551140670773547f},
                        {764.3655248643528689f, -0.92630200888366945f, -0.9675229603596517f, -0.27845290869168118f},
                        {527.107573587542228f, -1.4594385811273854f, -1.450082094097871593f, -1.5843722511996204f});
      for (int c = 0; c < 3; c++) q.dct4x8multipliers[c] = 1.0f;
      break;
    case QAFV: {
      q.mode = 5;
      float w[3][9] = {{3072.0f, 3072.0f, 256.0f, 256.0f, 256.0f, 414.0f, 0.0f, 0.0f, 0.0f},
                       {1024.0f, 1024.0f, 50.0f, 50.0f, 50.0f, 58.0f, 0.0f, 0.0f, 0.0f},
                       {384.0f, 384.0f, 12.0f, 12.0f, 12.0f, 22.0f, -0.25f, -0.25f, -0.25f}};
      memcpy(q.afv_weights, w, sizeof(w));
      q.dct = LibraryQuant(QDCT4X8).dct;       // the 4x8 part uses the DCT4X8 bands
      q.dct4x4 = LibraryQuant(QDCT4X4).dct;    // the 4x4 part uses the DCT4X4 bands
      break;
    }
    default: {
      // 64x64 family: base bands scaled per size [R]
      static const float k64[3] = {26629.073922049845f, 9311.3238710010046f, 4992.2486445538634f};
      static const float k32x64[3] = {23629.073922049845f, 8611.3238710010046f, 4492.2486445538634f};
      float mul; const float* base;
      switch (kind) {
        case QDCT64X64: mul = 0.9f; base = k64; break;
        case QDCT32X64: mul = 0.65f; base = k32x64; break;
        case QDCT128X128: mul = 1.8f; base = k64; break;
        case QDCT64X128: mul = 1.3f; base = k32x64; break;
        case QDCT256X256: mul = 3.6f; base = k64; break;
        default: mul = 2.6f; base = k32x64; break;
      }
      q.mode = 6;
      q.dct = MakeBands(8, {mul * base[0], -1.025f, -0.78f, -0.65012f, -0.19041574084286472f, -0.20819395464f, -0.421064f, -0.32733845535848671f},
                        {mul * base[1], -0.3041958212306401f, -0.3633036457487539f, -0.35660379990111464f, -0.3443074455424403f, -0.33699592683512467f, -0.30180866526242109f, -0.27321683125358037f},
                        {mul * base[2], -1.2f, -1.2f, -0.8f, -0.7f, -0.7f, -0.4f, -0.5f});
    }
  }
  return q;
}

inline float BandMult(float v) { return v > 0 ? 1.0f + v : 1.0f / (1.0f - v); }

// base/fast_math-inl.h FastLog2f / FastPow2f / FastPowf: the rational approximations quant_weights.cc interpolates the
// band weights with (NOT std::pow: the tables differ from an exact power by up to 3e-5 relative).  Constants recalled and
// checked: |FastLog2f - log2| < 2.8e-6 on [0.01, 100], FastPow2f within 2.1e-7 relative on [-20, 20].
inline float FastLog2f(float x) {
  const float p[3] = {-1.8503833400518310E-06f, 1.4287160470083755E+00f, 7.4245873327820566E-01f};
  const float q[3] = {9.9032814277590719E-01f, 1.0096718572241148E+00f, 1.7409343003366853E-01f};
  int32_t x_bits; memcpy(&x_bits, &x, 4);
  const int32_t exp_bits = x_bits - 0x3f2aaaab;   // = 2/3
  const int32_t exp_shifted = exp_bits >> 23;
  const int32_t m_bits = x_bits - (int32_t)((uint32_t)exp_shifted << 23);
  float mantissa; memcpy(&mantissa, &m_bits, 4);
  const float exp_val = (float)exp_shifted;
  const float t = mantissa - 1.0f;
  float yp = std::fmaf(p[2], t, p[1]); yp = std::fmaf(yp, t, p[0]);
  float yq = std::fmaf(q[2], t, q[1]); yq = std::fmaf(yq, t, q[0]);
  return yp / yq + exp_val;
}
inline float FastPow2f(float x) {
  const float floorx = std::floor(x);
  const int32_t e_bits = (int32_t)((uint32_t)((int32_t)floorx + 127) << 23);
  float exp; memcpy(&exp, &e_bits, 4);
  const float frac = x - floorx;
  float num = frac + 1.01749063e+01f;
  num = std::fmaf(num, frac, 4.88687798e+01f);
  num = std::fmaf(num, frac, 9.85506591e+01f);
  num = num * exp;
  float den = std::fmaf(frac, 2.10242958e-01f, -2.22328856e-02f);
  den = std::fmaf(den, frac, -1.94414990e+01f);
  den = std::fmaf(den, frac, 9.85506633e+01f);
  return num / den;
}
inline float FastPowf(float base, float exponent) { return FastPow2f(FastLog2f(base) * exponent); }
// quant_weights.cc Interpolate
inline float InterpolateBands(float pos, float max, const float* array, int len) {
  const float scaled_pos = pos * (len - 1) / max;
  const int idx = (int)scaled_pos;
  const float a = array[idx], b = array[idx + 1];
  return a * FastPowf(b / a, scaled_pos - idx);
}

// quant_weights.cc GetQuantWeights: weights (NOT inverted) for a ROWS x COLS table from band parameters
inline void BandWeights(const DctBandParams& p, int c, int ROWS, int COLS, float* out) {
  float bands[17];
  bands[0] = p.bands[c][0];
  if (bands[0] < 1e-8f) JXLO_FAIL("bad quant band");
  for (int i = 1; i < p.num_bands; i++) { bands[i] = bands[i - 1] * BandMult(p.bands[c][i]); if (bands[i] < 1e-8f) JXLO_FAIL("bad quant band"); }
  float scale = (p.num_bands - 1) / (kSqrt2f + 1e-6f);
  float rcpcol = scale / (COLS - 1), rcprow = scale / (ROWS - 1);
  for (int y = 0; y < ROWS; y++) {
    float dy = y * rcprow, dy2 = dy * dy;
    for (int x = 0; x < COLS; x++) {
      float dx = x * rcpcol;
      float dist = std::sqrt(std::fmaf(dx, dx, dy2));
      float w;
      if (p.num_bands == 1) w = bands[0];
      else {
        int idx = (int)dist;
        if (idx + 1 >= p.num_bands) idx = p.num_bands - 2;
        float frac = dist - idx;
        float a = bands[idx], b = bands[idx + 1];
        w = a * FastPowf(b / a, frac);
      }
      out[y * COLS + x] = w;
    }
  }
}

// Computes the dequant table (1/weight) of `kind` for channel c into out (rows*cols*64 floats, stored layout)
inline void ComputeQuantTable(const QuantEncoding& q0, int kind, int c, std::vector<float>& out) {
  QuantEncoding lib;
  const QuantEncoding* q = &q0;
  if (q0.mode == 0) { lib = LibraryQuant(kind); q = &lib; }
  const int ROWS = 8 * kKindRows[kind], COLS = 8 * kKindCols[kind];
  const size_t n = (size_t)ROWS * COLS;
  std::vector<float> w(n, 0.f);
  switch (q->mode) {
    case 7: {
      if (q->raw[c].size() != n) JXLO_FAIL("raw quant table size");
      out.resize(n);
      for (size_t i = 0; i < n; i++) {
        if (q->raw[c][i] <= 0) JXLO_FAIL("raw quant table value <= 0");
        // weights = 1/(den*v); table = 1/weights
        out[i] = 1.0f / (1.0f / (q->raw_den * (float)q->raw[c][i]));
      }
      return;
    }
    case 6: BandWeights(q->dct, c, ROWS, COLS, w.data()); break;
    case 1:
      JXLO_CHECK(n == 64);
      for (int i = 0; i < 64; i++) w[i] = q->idweights[c][0];
      w[1] = q->idweights[c][1]; w[8] = q->idweights[c][1]; w[9] = q->idweights[c][2];
      break;
    case 2: {
      JXLO_CHECK(n == 64);
      const float* d = q->dct2weights[c];
      w[0] = 1e6f;  // unused slot (LLF comes from LF)
      w[1] = w[8] = d[0]; w[9] = d[1];
      for (int y = 0; y < 2; y++) for (int x = 0; x < 2; x++) { w[y * 8 + x + 2] = d[2]; w[(y + 2) * 8 + x] = d[2]; }
      for (int y = 0; y < 2; y++) for (int x = 0; x < 2; x++) w[(y + 2) * 8 + x + 2] = d[3];
      for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) { w[y * 8 + x + 4] = d[4]; w[(y + 4) * 8 + x] = d[4]; }
      for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) w[(y + 4) * 8 + x + 4] = d[5];
      break;
    }
    case 3: {
      JXLO_CHECK(n == 64);
      float w4[16];
      BandWeights(q->dct, c, 4, 4, w4);
      for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) w[y * 8 + x] = w4[(y / 2) * 4 + x / 2];
      w[1] /= q->dct4multipliers[c][0]; w[8] /= q->dct4multipliers[c][0]; w[9] /= q->dct4multipliers[c][1];
      break;
    }
    case 4: {
      JXLO_CHECK(n == 64);
      float w48[32];
      BandWeights(q->dct, c, 4, 8, w48);
      for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) w[y * 8 + x] = w48[(y / 2) * 8 + x];
      w[8] /= q->dct4x8multipliers[c];
      break;
    }
    case 5: {  // quant_weights.cc kQuantModeAFV
      JXLO_CHECK(n == 64);
      static const float kFreqs[16] = {0xBAD, 0xBAD, 0.8517778890324296f, 5.37778436506804f, 0xBAD, 0xBAD, 4.734747904497923f, 5.449245381693219f,
                                       1.6598270267479331f, 4.0f, 7.275749096817861f, 10.423227632456525f, 2.662932286148962f, 7.630657783650829f,
                                       8.962388608184032f, 12.97166202570235f};
      float w48[32], w44[16];
      BandWeights(q->dct, c, 4, 8, w48);
      BandWeights(q->dct4x4, c, 4, 4, w44);
      const float lo = 0.8517778890324296f, hi = 12.97166202570235f - lo + 1e-6f;
      float bands[4];
      bands[0] = q->afv_weights[c][5];
      if (bands[0] < 1e-8f) JXLO_FAIL("bad AFV band");
      for (int i = 1; i < 4; i++) { bands[i] = bands[i - 1] * BandMult(q->afv_weights[c][i + 5]); if (bands[i] < 1e-8f) JXLO_FAIL("bad AFV band"); }
      auto set = [&](int x, int y, float v) { w[y * 8 + x] = v; };
      w[0] = 1.0f;   // unused (LLF comes from the LF image)
      set(0, 1, q->afv_weights[c][0]); set(1, 0, q->afv_weights[c][1]);
      set(0, 2, q->afv_weights[c][2]); set(2, 0, q->afv_weights[c][3]); set(2, 2, q->afv_weights[c][4]);
      for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
        if (x < 2 && y < 2) continue;
        set(2 * x, 2 * y, InterpolateBands(kFreqs[y * 4 + x] - lo, hi, bands, 4));
      }
      for (int y = 0; y < 4; y++) for (int x = 0; x < 8; x++) { if (x == 0 && y == 0) continue; w[(2 * y + 1) * 8 + x] = w48[y * 8 + x]; }
      for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) { if (x == 0 && y == 0) continue; w[(2 * y) * 8 + 2 * x + 1] = w44[y * 4 + x]; }
      break;
    }
    default: JXLO_FAIL("bad quant mode");
  }
  out.resize(n);
  for (size_t i = 0; i < n; i++) {
    if (!(w[i] > 0) || !std::isfinite(w[i])) JXLO_FAIL("bad quant weight");
    out[i] = 1.0f / w[i];
  }
}

// dec_group.cc AdjustQuantBias [R]
inline float AdjustQuantBias(int c, int32_t q, const float* biases) {
  if (q == 0) return 0.0f;
  if (q == 1) return biases[c];
  if (q == -1) return -biases[c];
  float fq = (float)q;
  return fq - biases[3] / fq;
}

}  // namespace jxlo
