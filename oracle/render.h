// ORACLE — TEST INFRASTRUCTURE ONLY (see bits.h).
// Render pipeline stages: restates libjxl v0.11.2 lib/jxl/{compressed_dc.cc (adaptive LF smoothing), epf.cc,
// render_pipeline/stage_{gaborish,epf,xyb,from_linear,write}.cc, dec_xyb-inl.h, cms/transfer_functions-inl.h}.
// SURVEY.md App. B.6 "Render constants [R]" — constants self-consistency-checked, operation order recalled:
// PARITY UNPINNED against libjxl.  The integer write stage is [V] through the PNG goldens.
#pragma once
#include "headers.h"
#include <cmath>

namespace jxlo {

struct Plane {
  int w = 0, h = 0;
  std::vector<float> d;
  Plane() {}
  Plane(int w_, int h_) : w(w_), h(h_), d((size_t)w_ * h_, 0.f) {}
  float* row(int y) { return d.data() + (size_t)y * w; }
  const float* row(int y) const { return d.data() + (size_t)y * w; }
};
struct Image3 { Plane p[3]; int w() const { return p[0].w; } int h() const { return p[0].h; } };

inline int Mirror(int x, int size) {
  while (x < 0 || x >= size) x = x < 0 ? -x - 1 : 2 * size - 1 - x;
  return x;
}

// ---- non-separable upsampling 2x / 4x / 8x (stage_upsampling.cc [R]; ISO/IEC 18181-1 "upsampling") -------------------
// weights: the 15 / 55 / 210 stored coefficients of the symmetric (5N x 5N) matrix, N = up / 2.  Sub-pixel (sy, sx) of an
// input sample uses the 5x5 kernel K[ky][kx][jy][jx] = M[5 ky + jy][5 kx + jx] with ky = min(sy, up-1-sy) and the window
// mirrored (jy = 4 - iy) for the lower / right half; the result is clamped to the range of the 25 input samples.
// Accumulation order (this restatement's choice, shared with the HIP kernel): row-major over the window, fused multiply-add.
// Default 2x weights: recalled from the standard (they sum to 1 over the kernel, see tests); the 4x / 8x default tables are
// not reproducible here — streams relying on them are rejected, streams carrying custom weights decode.
static const float kDefaultUp2Weights[15] = {-0.01716200f, -0.03452303f, -0.04022174f, -0.02921014f, -0.00624645f, 0.14111091f, 0.28896755f, 0.00278718f,
                                             -0.01610267f, 0.56661550f,  0.03777607f,  -0.01986694f, -0.03144731f, -0.01185068f, -0.00213539f};
inline Plane UpsamplePlane(const Plane& in, int up, const float* weights, int out_w, int out_h) {
  const int N = up / 2;
  auto M = [&](int i, int j) { const int y = i < j ? i : j, x = i < j ? j : i; return weights[5 * N * y - y * (y - 1) / 2 + x - y]; };
  Plane out(out_w, out_h);
  for (int oy = 0; oy < out_h; oy++) {
    const int y = oy / up, sy = oy % up, ky = sy < N ? sy : up - 1 - sy;
    const bool fy = sy >= N;
    for (int ox = 0; ox < out_w; ox++) {
      const int x = ox / up, sx = ox % up, kx = sx < N ? sx : up - 1 - sx;
      const bool fx = sx >= N;
      float sum = 0.0f, mn = 0.0f, mx = 0.0f;
      for (int iy = 0; iy < 5; iy++) {
        const float* row = in.row(Mirror(y + iy - 2, in.h));
        for (int ix = 0; ix < 5; ix++) {
          const float v = row[Mirror(x + ix - 2, in.w)];
          const float k = M(5 * ky + (fy ? 4 - iy : iy), 5 * kx + (fx ? 4 - ix : ix));
          sum = std::fmaf(k, v, sum);
          if (iy == 0 && ix == 0) { mn = v; mx = v; } else { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
        }
      }
      out.row(oy)[ox] = sum < mn ? mn : (sum > mx ? mx : sum);
    }
  }
  return out;
}

// compressed_dc.cc AdaptiveDCSmoothing [R]
inline void AdaptiveLFSmoothing(const float* lf_factors, Image3& lf) {
  const int w = lf.w(), h = lf.h();
  if (w <= 2 || h <= 2) return;
  const float kW0 = 0.05226273532324128f, kW1 = 0.20345139757231578f, kW2 = 0.0334829185968739f;
  Image3 out = lf;
  for (int y = 1; y + 1 < h; y++) {
    for (int x = 1; x + 1 < w; x++) {
      float gap = 0.5f;
      float mc[3], sm[3];
      for (int c = 0; c < 3; c++) {
        const float* t = lf.p[c].row(y - 1); const float* m = lf.p[c].row(y); const float* b = lf.p[c].row(y + 1);
        float corner = (t[x - 1] + t[x + 1]) + (b[x - 1] + b[x + 1]);
        float edge = (t[x] + m[x - 1]) + (m[x + 1] + b[x]);
        mc[c] = m[x];
        sm[c] = std::fmaf(corner, kW2, std::fmaf(edge, kW1, mc[c] * kW0));
        gap = std::max(gap, std::fabs((mc[c] - sm[c]) / lf_factors[c]));
      }
      float factor = std::max(0.0f, std::fmaf(-4.0f, gap, 3.0f));
      for (int c = 0; c < 3; c++) out.p[c].row(y)[x] = std::fmaf(sm[c] - mc[c], factor, mc[c]);
    }
  }
  lf = out;
}

// stage_gaborish.cc [R]: 3x3 symmetric convolution, mirrored image borders
inline void Gaborish(const LoopFilter& lf, Image3& img) {
  const int w = img.w(), h = img.h();
  for (int c = 0; c < 3; c++) {
    float w1 = lf.gab_w[2 * c], w2 = lf.gab_w[2 * c + 1];
    float div = 1.0f + 4.0f * (w1 + w2);
    float n0 = 1.0f / div, n1 = w1 / div, n2 = w2 / div;
    Plane out(w, h);
    for (int y = 0; y < h; y++) {
      const float* t = img.p[c].row(Mirror(y - 1, h));
      const float* m = img.p[c].row(y);
      const float* b = img.p[c].row(Mirror(y + 1, h));
      float* o = out.row(y);
      for (int x = 0; x < w; x++) {
        int xl = Mirror(x - 1, w), xr = Mirror(x + 1, w);
        float sum0 = m[x];
        float sum1 = (m[xl] + m[xr]) + (t[x] + b[x]);
        float sum2 = (t[xl] + t[xr]) + (b[xl] + b[xr]);
        o[x] = std::fmaf(sum2, n2, std::fmaf(sum1, n1, sum0 * n0));
      }
    }
    img.p[c] = out;
  }
}

static const float kInvSigmaNum = -1.1715728752538099024f;
static const float kMinSigma = -3.90524291751269967465540850526868f;

// epf.cc ComputeSigma [R]: per-8x8-block inverse sigma. hf_mul/sharpness are per 8x8 block maps of the frame.
inline void ComputeInvSigma(const LoopFilter& lf, float quant_scale, const std::vector<int32_t>& hf_mul,
                            const std::vector<uint8_t>& sharpness, int bw, int bh, std::vector<float>& inv_sigma) {
  inv_sigma.assign((size_t)bw * bh, 0.f);
  for (int i = 0; i < bw * bh; i++) {
    float sigma_quant = lf.quant_mul / (quant_scale * (float)hf_mul[i] * kInvSigmaNum);
    float sigma = sigma_quant * lf.sharp_lut[sharpness[i]];
    sigma = std::min(-1e-4f, sigma);
    inv_sigma[i] = 1.0f / sigma;
  }
}

// stage_epf.cc [R]. pass: 0 (12 taps, plus-SAD), 1 (4 taps, plus-SAD), 2 (4 taps, point-SAD)
inline void EPFPass(const LoopFilter& lf, int pass, const std::vector<float>& inv_sigma, int bw, Image3& img) {
  const int w = img.w(), h = img.h();
  Image3 out = img;
  float sigma_scale = pass == 0 ? lf.pass0_sigma_scale : pass == 2 ? lf.pass2_sigma_scale : 1.0f;
  const float sm = sigma_scale * 1.65f;
  const float bsm = sm * lf.border_sad_mul;
  static const int taps0[12][2] = {{0, -2}, {-1, -1}, {0, -1}, {1, -1}, {-2, 0}, {-1, 0}, {1, 0}, {2, 0}, {-1, 1}, {0, 1}, {1, 1}, {0, 2}};
  static const int taps1[4][2] = {{0, -1}, {-1, 0}, {1, 0}, {0, 1}};
  static const int plus[5][2] = {{0, 0}, {0, -1}, {-1, 0}, {1, 0}, {0, 1}};
  auto px = [&](int c, int x, int y) -> float { return img.p[c].row(Mirror(y, h))[Mirror(x, w)]; };
  for (int y = 0; y < h; y++) {
    for (int x = 0; x < w; x++) {
      float is = inv_sigma[(size_t)(y / 8) * bw + x / 8];
      if (is < kMinSigma) continue;  // copy
      bool border = (x % 8 == 0) || (x % 8 == 7) || (y % 8 == 0) || (y % 8 == 7);
      float sad_mul = border ? bsm : sm;
      float vmul = is * sad_mul;
      float wsum = 1.0f;
      float acc[3] = {px(0, x, y), px(1, x, y), px(2, x, y)};
      const int ntaps = pass == 0 ? 12 : 4;
      for (int t = 0; t < ntaps; t++) {
        int dx = pass == 0 ? taps0[t][0] : taps1[t][0];
        int dy = pass == 0 ? taps0[t][1] : taps1[t][1];
        float sad = 0.f;
        if (pass == 2) {
          for (int c = 0; c < 3; c++) sad = std::fmaf(std::fabs(px(c, x + dx, y + dy) - px(c, x, y)), lf.channel_scale[c], sad);
        } else {
          for (int c = 0; c < 3; c++) {
            float s = 0.f;
            for (int k = 0; k < 5; k++) s += std::fabs(px(c, x + dx + plus[k][0], y + dy + plus[k][1]) - px(c, x + plus[k][0], y + plus[k][1]));
            sad = std::fmaf(s, lf.channel_scale[c], sad);
          }
        }
        float wgt = std::max(0.0f, std::fmaf(sad, vmul, 1.0f));
        wsum += wgt;
        for (int c = 0; c < 3; c++) acc[c] = std::fmaf(wgt, px(c, x + dx, y + dy), acc[c]);
      }
      float inv = 1.0f / wsum;
      for (int c = 0; c < 3; c++) out.p[c].row(y)[x] = acc[c] * inv;
    }
  }
  img = out;
}

// dec_xyb-inl.h XybToRgb + opsin_params.cc [R]
struct OpsinParams {
  float inv[9];
  float neg_bias[3];       // -bias
  float neg_bias_cbrt[3];  // cbrt(-bias)
};
// dec_cache.cc / dec_xyb.cc OutputEncodingInfo::SetColorEncoding: for a grey-scale image the three rows of the inverse
// opsin matrix are replaced by their luminance-weighted sum (kSRGBLuminances), so R = G = B = luminance by construction;
// Mul3x3Matrix accumulates each element in double.
// ---- output primaries / white point (dec_xyb.cc OutputEncodingInfo::SetColorEncoding; cms/jxl_cms_internal.h PrimariesToXYZ,
// AdaptToXYZD50 — [R]: formulas restated, intermediate precision (double here) not checkable against libjxl).  XYB decodes to linear
// sRGB; an image whose header names other primaries or another white point gets the inverse opsin matrix multiplied by
// (linear sRGB -> its own primaries), going through XYZ D50 with linear Bradford adaptation on both sides.
struct Mat3d { double m[3][3]; };
inline Mat3d Mul3(const Mat3d& a, const Mat3d& b) {
  Mat3d r;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double e = 0; for (int k = 0; k < 3; k++) e += a.m[i][k] * b.m[k][j]; r.m[i][j] = e; }
  return r;
}
inline Mat3d Inv3(const Mat3d& a) {
  const double (*m)[3] = a.m;
  const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
  JXLO_CHECK(std::fabs(det) > 1e-12);
  Mat3d r;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    r.m[j][i] = (m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1]) / det;
  }
  return r;
}
inline Mat3d PrimariesToXYZ(const double p[6], const double w[2]) {
  JXLO_CHECK(w[1] > 1e-9);
  Mat3d prim;
  for (int c = 0; c < 3; c++) { prim.m[0][c] = p[2 * c]; prim.m[1][c] = p[2 * c + 1]; prim.m[2][c] = 1.0 - p[2 * c] - p[2 * c + 1]; }
  const Mat3d pinv = Inv3(prim);
  const double wxyz[3] = {w[0] / w[1], 1.0, (1.0 - w[0] - w[1]) / w[1]};
  Mat3d r;
  for (int c = 0; c < 3; c++) {
    double scale = 0;
    for (int k = 0; k < 3; k++) scale += pinv.m[c][k] * wxyz[k];
    for (int i = 0; i < 3; i++) r.m[i][c] = prim.m[i][c] * scale;
  }
  return r;
}
inline Mat3d AdaptToXYZD50(const double w[2]) {
  JXLO_CHECK(w[1] > 1e-9);
  const Mat3d brad = {{{0.8951, 0.2664, -0.1614}, {-0.7502, 1.7135, 0.0367}, {0.0389, -0.0685, 1.0296}}};
  const Mat3d brad_inv = {{{0.9869929, -0.1470543, 0.1599627}, {0.4323053, 0.5183603, 0.0492912}, {-0.0085287, 0.0400428, 0.9684867}}};
  const double wxyz[3] = {w[0] / w[1], 1.0, (1.0 - w[0] - w[1]) / w[1]}, w50[3] = {0.96422, 1.0, 0.82521};
  Mat3d scaled = brad;
  for (int i = 0; i < 3; i++) {
    double lms = 0, lms50 = 0;
    for (int k = 0; k < 3; k++) { lms += brad.m[i][k] * wxyz[k]; lms50 += brad.m[i][k] * w50[k]; }
    for (int k = 0; k < 3; k++) scaled.m[i][k] = brad.m[i][k] * (lms50 / lms);
  }
  return Mul3(brad_inv, scaled);
}
// xy of the enumerated white points / primaries (color_encoding_internal.h)
inline void WhiteXY(const ColorEncoding& c, double w[2]) {
  switch (c.white_point) {
    case 1: w[0] = 0.3127; w[1] = 0.3290; break;
    case 2: w[0] = c.custom_xy[0] * 1e-6; w[1] = c.custom_xy[1] * 1e-6; break;
    case 10: w[0] = w[1] = 1.0 / 3; break;
    case 11: w[0] = 0.314; w[1] = 0.351; break;
    default: JXLO_FAIL("white point enum");
  }
}
inline void PrimariesXY(const ColorEncoding& c, double p[6]) {
  static const double srgb[6] = {0.639998686, 0.330010138, 0.300003784, 0.600003357, 0.150002046, 0.059997204};
  static const double bt2100[6] = {0.708, 0.292, 0.170, 0.797, 0.131, 0.046}, p3[6] = {0.680, 0.320, 0.265, 0.690, 0.150, 0.060};
  switch (c.primaries) {
    case 1: for (int i = 0; i < 6; i++) p[i] = srgb[i]; break;
    case 2: for (int i = 0; i < 6; i++) p[i] = c.custom_xy[2 + i] * 1e-6; break;
    case 9: for (int i = 0; i < 6; i++) p[i] = bt2100[i]; break;
    case 11: for (int i = 0; i < 6; i++) p[i] = p3[i]; break;
    default: JXLO_FAIL("primaries enum");
  }
}

inline OpsinParams MakeOpsin(const ImageMetadata& m, float intensity_target, float luminances[3] = nullptr) {
  OpsinParams o;
  float s = 255.0f / intensity_target;
  float inv[9];
  for (int i = 0; i < 9; i++) inv[i] = m.opsin_inv[i];
  if (luminances) { luminances[0] = 0.2126f; luminances[1] = 0.7152f; luminances[2] = 0.0722f; }
  if (!m.color.all_default && !m.color.want_icc && m.color.color_space == 0 && (m.color.primaries != 1 || m.color.white_point != 1)) {
    double w[2], p[6], w65[2] = {0.3127, 0.3290}, psrgb[6];
    ColorEncoding srgb; srgb.white_point = 1; srgb.primaries = 1;
    WhiteXY(m.color, w); PrimariesXY(m.color, p); PrimariesXY(srgb, psrgb);
    const Mat3d srgb_to_xyzd50 = Mul3(AdaptToXYZD50(w65), PrimariesToXYZ(psrgb, w65));
    const Mat3d original_to_xyz = PrimariesToXYZ(p, w);
    if (luminances) for (int i = 0; i < 3; i++) luminances[i] = (float)original_to_xyz.m[1][i];
    const Mat3d srgb_to_original = Mul3(Inv3(Mul3(AdaptToXYZD50(w), original_to_xyz)), srgb_to_xyzd50);
    Mat3d oi;
    for (int i = 0; i < 9; i++) oi.m[i / 3][i % 3] = inv[i];
    const Mat3d adapted = Mul3(srgb_to_original, oi);
    for (int i = 0; i < 9; i++) inv[i] = (float)adapted.m[i / 3][i % 3];
  }
  if (m.color.color_space == 1) {
    const float lum[3] = {0.2126f, 0.7152f, 0.0722f};
    float folded[9];
    for (int x = 0; x < 3; x++) for (int y = 0; y < 3; y++) {
      double e = 0;
      for (int z = 0; z < 3; z++) e += lum[z] * inv[z * 3 + x];
      folded[y * 3 + x] = (float)e;
    }
    for (int i = 0; i < 9; i++) inv[i] = folded[i];
  }
  for (int i = 0; i < 9; i++) o.inv[i] = inv[i] * s;
  for (int i = 0; i < 3; i++) { o.neg_bias[i] = m.opsin_bias[i]; o.neg_bias_cbrt[i] = std::cbrt(m.opsin_bias[i]); }
  return o;
}
inline void XybToLinear(const OpsinParams& o, float X, float Y, float B, float* r, float* g, float* b) {
  float gr = (Y + X) - o.neg_bias_cbrt[0];
  float gg = (Y - X) - o.neg_bias_cbrt[1];
  float gb = B - o.neg_bias_cbrt[2];
  float mr = std::fmaf(gr * gr, gr, o.neg_bias[0]);
  float mg = std::fmaf(gg * gg, gg, o.neg_bias[1]);
  float mb = std::fmaf(gb * gb, gb, o.neg_bias[2]);
  *r = std::fmaf(o.inv[2], mb, std::fmaf(o.inv[1], mg, o.inv[0] * mr));
  *g = std::fmaf(o.inv[5], mb, std::fmaf(o.inv[4], mg, o.inv[3] * mr));
  *b = std::fmaf(o.inv[8], mb, std::fmaf(o.inv[7], mg, o.inv[6] * mr));
}

// transfer_functions-inl.h TF_SRGB::EncodedFromDisplay [R] (rational polynomial in sqrt(x))
inline float LinearToSRGB(float v) {
  static const float p[5] = {-5.135152395e-4f, 5.287254571e-3f, 3.903842876e-1f, 1.474205315f, 7.352629620e-1f};
  static const float q[5] = {1.004519624e-2f, 3.036675394e-1f, 1.340816930f, 9.258482155e-1f, 2.424867759e-2f};
  float x = std::fabs(v);
  float lin = x * 12.92f;
  float s = std::sqrt(x);
  float yp = p[4], yq = q[4];
  for (int i = 3; i >= 0; i--) { yp = std::fmaf(yp, s, p[i]); yq = std::fmaf(yq, s, q[i]); }
  float poly = yp / yq;
  float r = x > 0.0031308f ? poly : lin;
  return std::copysign(r, v);
}

// transfer_functions-inl.h TF_PQ::EncodedFromDisplay [R]: 4-over-4 rational polynomials in x^(1/4), one for small values; x = linear
// value (1.0 = intensity target), scale = intensity_target / 10000.  The coefficients reproduce SMPTE ST 2084 to 7e-7 (1.7e-6 below
// 1e-4), which is how their recollection was checked (tests/test_oracle_goldens.py).
inline float PqFromLinear(float v, float scale) {
  static const float p[5] = {1.351392e-02f, -1.095778e+00f, 5.522776e+01f, 1.492516e+02f, 4.838434e+01f};
  static const float q[5] = {1.012416e+00f, 2.016708e+01f, 9.263710e+01f, 1.120607e+02f, 2.590418e+01f};
  static const float plo[5] = {9.863406e-06f, 3.881234e-01f, 1.352821e+02f, 6.889862e+04f, -2.864824e+05f};
  static const float qlo[5] = {3.371868e+01f, 1.477719e+03f, 1.608477e+04f, -4.389884e+04f, -2.072546e+05f};
  const float xs = std::fabs(v) * scale;
  const float t = std::sqrt(std::sqrt(xs));
  const bool small = xs < 1e-4f;
  const float* pp = small ? plo : p;
  const float* qq = small ? qlo : q;
  float yp = pp[4], yq = qq[4];
  for (int i = 3; i >= 0; i--) { yp = std::fmaf(yp, t, pp[i]); yq = std::fmaf(yq, t, qq[i]); }
  return std::copysign(yp / yq, v);
}
// transfer_functions-inl.h TF_HLG_Base::EncodedFromDisplay (scalar, double; stage_from_linear.cc OpHlg applies it lane by lane)
inline float HlgFromLinear(float v) {
  const double kA = 0.17883277, kB = 1 - 4 * kA, kC = 0.5599107295, kDiv12 = 1.0 / 12;
  double s = std::fabs((double)v);
  if (s == 0.0) return 0.0f;
  const double e = s <= kDiv12 ? std::sqrt(3.0 * s) : kA * std::log(12 * s - kB) + kC;
  return (float)std::copysign(e, (double)v);
}
// cms/tone_mapping-inl.h HlgOOTF::ToSceneLight(display_luminance = intensity target, luminances of the output primaries): display
// light -> scene light before the HLG OETF; skipped when the exponent is within 0.01 of zero
struct HlgOotf {
  float exponent = 0, lum[3] = {0, 0, 0}; bool apply = false;
  HlgOotf() {}
  HlgOotf(float display_luminance, const float luminances[3]) {
    const float gamma = (1 / 1.2f) * std::pow(1.111f, -std::log2(display_luminance / 1000.f));
    exponent = gamma - 1;
    apply = exponent < -0.01f || 0.01f < exponent;
    for (int i = 0; i < 3; i++) lum[i] = luminances[i];
  }
  void Apply(float* r, float* g, float* b) const {
    if (!apply) return;
    const float luminance = std::fmaf(lum[0], *r, std::fmaf(lum[1], *g, lum[2] * *b));
    const float ratio = std::min(FastPowf(luminance, exponent), 1e9f);
    *r *= ratio; *g *= ratio; *b *= ratio;
  }
};

inline uint16_t FloatToHalf(float f) {
  uint32_t x; memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000;
  int32_t exp = (int32_t)((x >> 23) & 0xFF) - 127 + 15;
  uint32_t mant = x & 0x7FFFFF;
  if (((x >> 23) & 0xFF) == 0xFF) return (uint16_t)(sign | 0x7C00 | (mant ? 0x200 : 0));
  if (exp >= 31) return (uint16_t)(sign | 0x7C00);
  if (exp <= 0) {
    if (exp < -10) return (uint16_t)sign;
    mant |= 0x800000;
    int shift = 14 - exp;
    uint32_t m = mant >> shift;
    uint32_t rem = mant & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (m & 1))) m++;
    return (uint16_t)(sign | m);
  }
  uint32_t m = mant >> 13, rem = mant & 0x1FFF;
  uint32_t r = (uint32_t)(exp << 10) | m;
  if (rem > 0x1000 || (rem == 0x1000 && (m & 1))) r++;
  return (uint16_t)(sign | r);
}

}  // namespace jxlo
