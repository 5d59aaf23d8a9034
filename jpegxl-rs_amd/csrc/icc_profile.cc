// jxl-hip: ICC v4.4 profile synthesis from the codestream's enumerated ColorEncoding (host only; a few hundred bytes).
//
// Replaces what libjxl's JxlDecoderGetICCProfileSize / JxlDecoderGetColorAsICCProfile return to
// jpegxl-rs/src/decode.rs:368-388 when the image carries no embedded ICC stream: a display-class matrix/TRC profile
// (tags desc, cprt, wtpt, chad, r/g/bXYZ, r/g/bTRC — or kTRC for grey) with the white point adapted to the D50 PCS by
// the linear Bradford transform and the profile ID set to the MD5 of the profile (ICC.1:2010 §7.2.18).
// libjxl's own byte layout is not reproducible here (v0.11.2 sources are absent), so the contract is colorimetric:
// the reference's test only requires that lcms2 accepts the profile (tests/decode.rs:64); ours checks that too and that
// lcms2 reads back the encoded primaries / curve.
#include "host_parse.h"
#include <cmath>
#include <cstring>

namespace jxlhip {

namespace {

struct Md5 {
  uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
  static uint32_t Rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
  void Block(const uint8_t* p) {
    static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20,
                              4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
    uint32_t m[16];
    for (int i = 0; i < 16; i++) m[i] = p[4 * i] | p[4 * i + 1] << 8 | p[4 * i + 2] << 16 | (uint32_t)p[4 * i + 3] << 24;
    uint32_t A = a, B = b, C = c, D = d;
    for (int i = 0; i < 64; i++) {
      uint32_t f; int g;
      if (i < 16) { f = (B & C) | (~B & D); g = i; }
      else if (i < 32) { f = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
      else if (i < 48) { f = B ^ C ^ D; g = (3 * i + 5) & 15; }
      else { f = C ^ (B | ~D); g = (7 * i) & 15; }
      const uint32_t k = (uint32_t)(int64_t)std::floor(std::fabs(std::sin((double)(i + 1))) * 4294967296.0);
      const uint32_t t = D; D = C; C = B;
      B = B + Rol(A + f + k + m[g], S[i]);
      A = t;
    }
    a += A; b += B; c += C; d += D;
  }
  void Digest(const vec<uint8_t>& msg, uint8_t out[16]) {
    vec<uint8_t> v(msg);
    const uint64_t bits = (uint64_t)msg.size() * 8;
    v.push_back(0x80);
    while (v.size() % 64 != 56) v.push_back(0);
    for (int i = 0; i < 8; i++) v.push_back((uint8_t)(bits >> (8 * i)));
    for (size_t i = 0; i < v.size(); i += 64) Block(&v[i]);
    const uint32_t w[4] = {a, b, c, d};
    for (int i = 0; i < 16; i++) out[i] = (uint8_t)(w[i / 4] >> (8 * (i % 4)));
  }
};

void Put32(vec<uint8_t>& v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s)); }
void Put16(vec<uint8_t>& v, uint32_t x) { v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x); }
void PutTag(vec<uint8_t>& v, const char* t) { for (int i = 0; i < 4; i++) v.push_back((uint8_t)t[i]); }
void PutS15(vec<uint8_t>& v, double x) { Put32(v, (uint32_t)(int32_t)std::lround(x * 65536.0)); }
void Set32(vec<uint8_t>& v, size_t pos, uint32_t x) { for (int i = 0; i < 4; i++) v[pos + i] = (uint8_t)(x >> (24 - 8 * i)); }

struct Mat3 { double m[3][3]; };
Mat3 Mul(const Mat3& a, const Mat3& b) {
  Mat3 r{};
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) r.m[i][j] += a.m[i][k] * b.m[k][j];
  return r;
}
Mat3 Inv(const Mat3& a) {
  const double (*m)[3] = a.m;
  const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
  if (std::fabs(det) < 1e-12) throw ParseError("colour encoding: singular primaries", false);
  Mat3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      r.m[j][i] = (m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1]) / det;
    }
  return r;
}
void XyToXyz(double x, double y, double out[3]) {
  if (!(y > 1e-9)) throw ParseError("colour encoding: white point / primary with y <= 0", false);
  out[0] = x / y; out[1] = 1.0; out[2] = (1.0 - x - y) / y;
}

constexpr double kD50[3] = {0.96420288, 1.0, 0.82490540};

// linear Bradford adaptation from `white` (XYZ, Y=1) to the D50 PCS
Mat3 AdaptToD50(const double white[3]) {
  const Mat3 brad = {{{0.8951, 0.2664, -0.1614}, {-0.7502, 1.7135, 0.0367}, {0.0389, -0.0685, 1.0296}}};
  double src[3] = {0, 0, 0}, dst[3] = {0, 0, 0};
  for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) { src[i] += brad.m[i][k] * white[k]; dst[i] += brad.m[i][k] * kD50[k]; }
  Mat3 scale{};
  for (int i = 0; i < 3; i++) scale.m[i][i] = dst[i] / src[i];
  return Mul(Inv(brad), Mul(scale, brad));
}

void Mluc(vec<uint8_t>& v, const std::string& text) {
  PutTag(v, "mluc"); Put32(v, 0); Put32(v, 1); Put32(v, 12); PutTag(v, "enUS"); Put32(v, (uint32_t)text.size() * 2); Put32(v, 28);
  for (char ch : text) Put16(v, (uint8_t)ch);
}
void Xyz(vec<uint8_t>& v, const double p[3]) { PutTag(v, "XYZ "); Put32(v, 0); for (int i = 0; i < 3; i++) PutS15(v, p[i]); }
void Para(vec<uint8_t>& v, int type, std::initializer_list<double> params) {
  PutTag(v, "para"); Put32(v, 0); Put16(v, (uint32_t)type); Put16(v, 0);
  for (double p : params) PutS15(v, p);
}

}  // namespace

std::string ColorDescription(const ImageHeader& ih) {
  static const char* kSpace[] = {"RGB", "Gra", "XYB", "CS?"};
  static const char* kIntent[] = {"Per", "Rel", "Sat", "Abs"};
  std::string s = kSpace[ih.color_space < 3 ? ih.color_space : 3];
  s += ih.white_point == 1 ? "_D65" : ih.white_point == 10 ? "_EER" : ih.white_point == 11 ? "_DCI" : "_Cst";
  if (ih.color_space == 0) s += ih.primaries == 1 ? "_SRG" : ih.primaries == 9 ? "_202" : ih.primaries == 11 ? "_DCI" : "_Cst";
  s += std::string("_") + kIntent[ih.rendering_intent & 3];
  if (ih.have_gamma) { char b[32]; snprintf(b, sizeof b, "_g%.7f", ih.gamma * 1e-7); s += b; }
  else s += ih.tf == 13 ? "_SRG" : ih.tf == 8 ? "_Lin" : ih.tf == 1 ? "_709" : ih.tf == 17 ? "_DCI" : ih.tf == 16 ? "_PeQ" : ih.tf == 18 ? "_HLG" : "_TF?";
  return s;
}

namespace {
void WhiteXy(const ImageHeader& ih, double wxy[2]) {
  switch (ih.white_point) {
    case 1: wxy[0] = 0.3127; wxy[1] = 0.3290; break;
    case 2: wxy[0] = ih.white_xy[0]; wxy[1] = ih.white_xy[1]; break;
    case 10: wxy[0] = wxy[1] = 1.0 / 3; break;
    case 11: wxy[0] = 0.314; wxy[1] = 0.351; break;
    default: throw ParseError("colour encoding: white point enum", false);
  }
}
const double kSrgbPrimaries[6] = {0.639998686, 0.330010138, 0.300003784, 0.600003357, 0.150002046, 0.059997204};
void PrimariesXy(const ImageHeader& ih, double pxy[6]) {
  switch (ih.primaries) {
    case 1: memcpy(pxy, kSrgbPrimaries, sizeof kSrgbPrimaries); break;
    case 2: for (int i = 0; i < 6; i++) pxy[i] = ih.prim_xy[i]; break;
    case 9: { const double p[6] = {0.708, 0.292, 0.170, 0.797, 0.131, 0.046}; memcpy(pxy, p, sizeof p); break; }
    case 11: { const double p[6] = {0.680, 0.320, 0.265, 0.690, 0.150, 0.060}; memcpy(pxy, p, sizeof p); break; }
    default: throw ParseError("colour encoding: primaries enum", false);
  }
}
// cms PrimariesToXYZ: columns = the primaries' XYZ scaled so that they add up to the white point
Mat3 PrimariesToXyz(const double pxy[6], const double wxy[2]) {
  if (!(wxy[1] > 1e-9)) throw ParseError("colour encoding: white point with y <= 0", false);
  Mat3 prim;
  for (int c = 0; c < 3; c++) { prim.m[0][c] = pxy[2 * c]; prim.m[1][c] = pxy[2 * c + 1]; prim.m[2][c] = 1.0 - pxy[2 * c] - pxy[2 * c + 1]; }
  const Mat3 pinv = Inv(prim);
  const double wxyz[3] = {wxy[0] / wxy[1], 1.0, (1.0 - wxy[0] - wxy[1]) / wxy[1]};
  Mat3 r;
  for (int c = 0; c < 3; c++) {
    double scale = 0;
    for (int k = 0; k < 3; k++) scale += pinv.m[c][k] * wxyz[k];
    for (int i = 0; i < 3; i++) r.m[i][c] = prim.m[i][c] * scale;
  }
  return r;
}
// cms AdaptToXYZD50 (its own D50 and inverse Bradford constants, not the ICC header's)
Mat3 AdaptToXyzD50(const double wxy[2]) {
  if (!(wxy[1] > 1e-9)) throw ParseError("colour encoding: white point with y <= 0", false);
  const Mat3 brad = {{{0.8951, 0.2664, -0.1614}, {-0.7502, 1.7135, 0.0367}, {0.0389, -0.0685, 1.0296}}};
  const Mat3 brad_inv = {{{0.9869929, -0.1470543, 0.1599627}, {0.4323053, 0.5183603, 0.0492912}, {-0.0085287, 0.0400428, 0.9684867}}};
  const double wxyz[3] = {wxy[0] / wxy[1], 1.0, (1.0 - wxy[0] - wxy[1]) / wxy[1]}, w50[3] = {0.96422, 1.0, 0.82521};
  Mat3 scaled = brad;
  for (int i = 0; i < 3; i++) {
    double lms = 0, lms50 = 0;
    for (int k = 0; k < 3; k++) { lms += brad.m[i][k] * wxyz[k]; lms50 += brad.m[i][k] * w50[k]; }
    for (int k = 0; k < 3; k++) scaled.m[i][k] = brad.m[i][k] * (lms50 / lms);
  }
  return Mul(brad_inv, scaled);
}
}  // namespace

bool SrgbToOriginalPrimaries(const ImageHeader& ih, double m[9], float luminances[3]) {
  for (int i = 0; i < 9; i++) m[i] = i % 4 == 0 ? 1.0 : 0.0;
  luminances[0] = 0.2126f; luminances[1] = 0.7152f; luminances[2] = 0.0722f;
  if (ih.color_default || ih.want_icc || ih.color_space != 0 || (ih.primaries == 1 && ih.white_point == 1)) return false;
  double w[2], p[6];
  const double w65[2] = {0.3127, 0.3290};
  WhiteXy(ih, w); PrimariesXy(ih, p);
  const Mat3 srgb_to_xyzd50 = Mul(AdaptToXyzD50(w65), PrimariesToXyz(kSrgbPrimaries, w65));
  const Mat3 original_to_xyz = PrimariesToXyz(p, w);
  for (int i = 0; i < 3; i++) luminances[i] = (float)original_to_xyz.m[1][i];
  const Mat3 srgb_to_original = Mul(Inv(Mul(AdaptToXyzD50(w), original_to_xyz)), srgb_to_xyzd50);
  for (int i = 0; i < 9; i++) m[i] = srgb_to_original.m[i / 3][i % 3];
  return true;
}

void ColorChromaticities(const ImageHeader& ih, double white_xy[2], double primaries_xy[6]) {
  ImageHeader h = ih;
  if (h.color_default) { h.white_point = 1; h.primaries = 1; }
  WhiteXy(h, white_xy);
  if (h.color_space == 1) memcpy(primaries_xy, kSrgbPrimaries, sizeof kSrgbPrimaries); else PrimariesXy(h, primaries_xy);
}

vec<uint8_t> SynthesizeIcc(const ImageHeader& ih) {
  if (ih.want_icc) throw ParseError("unsupported: embedded ICC profile", true);
  if (ih.color_space > 1) throw ParseError("unsupported: ICC profile for XYB / unknown colour space", true);
  const bool grey = ih.color_space == 1;
  double wxy[2];
  WhiteXy(ih, wxy);
  double pxy[6];
  PrimariesXy(ih, pxy);
  double white[3];
  XyToXyz(wxy[0], wxy[1], white);
  const Mat3 chad = AdaptToD50(white);
  double col[3][3] = {};  // adapted r, g, b colorants
  if (!grey) {
    Mat3 p;
    for (int c = 0; c < 3; c++) { double t[3]; XyToXyz(pxy[2 * c], pxy[2 * c + 1], t); for (int i = 0; i < 3; i++) p.m[i][c] = t[i]; }
    const Mat3 pi = Inv(p);
    double s[3] = {0, 0, 0};
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) s[i] += pi.m[i][k] * white[k];
    for (int i = 0; i < 3; i++) for (int c = 0; c < 3; c++) p.m[i][c] *= s[c];
    const Mat3 a = Mul(chad, p);
    for (int c = 0; c < 3; c++) for (int i = 0; i < 3; i++) col[c][i] = a.m[i][c];
  }

  vec<uint8_t> trc;
  if (ih.have_gamma) {
    if (ih.gamma == 0 || ih.gamma > 10000000) throw ParseError("colour encoding: gamma", false);
    Para(trc, 0, {1.0 / (ih.gamma * 1e-7)});
  } else {
    switch (ih.tf) {
      case 13: Para(trc, 3, {2.4, 1.0 / 1.055, 0.055 / 1.055, 1.0 / 12.92, 0.04045}); break;
      case 8: Para(trc, 0, {1.0}); break;
      case 1: Para(trc, 3, {1.0 / 0.45, 1.0 / 1.099, 0.099 / 1.099, 1.0 / 4.5, 0.081}); break;
      case 17: Para(trc, 0, {2.6}); break;
      case 16: case 18: {
        // HDR transfer functions have no parametric ICC form: a sampled curve (encoded value -> display / scene light, both normalised to
        // [0, 1]) carries what a v4 CMM can use, the `cicp` tag below what an HDR-aware one wants (ITU-T H.273 code points)
        const int n = 4096;
        PutTag(trc, "curv"); Put32(trc, 0); Put32(trc, (uint32_t)n);
        for (int i = 0; i < n; i++) {
          const double e = (double)i / (n - 1);
          double l;
          if (ih.tf == 16) {          // SMPTE ST 2084 EOTF, 1.0 = 10000 cd/m2
            const double m1 = 2610.0 / 16384, m2 = 2523.0 / 4096 * 128, c1 = 3424.0 / 4096, c2 = 2413.0 / 4096 * 32, c3 = 2392.0 / 4096 * 32;
            const double p = std::pow(e, 1.0 / m2);
            l = std::pow(std::max(p - c1, 0.0) / (c2 - c3 * p), 1.0 / m1);
          } else {                    // ARIB STD-B67 inverse OETF (scene light, 1.0 at signal 1.0)
            const double a = 0.17883277, b = 1 - 4 * a, c = 0.5 - a * std::log(4 * a);
            l = e <= 0.5 ? e * e / 3.0 : (std::exp((e - c) / a) + b) / 12.0;
          }
          Put16(trc, (uint32_t)std::lround(std::min(1.0, std::max(0.0, l)) * 65535.0));
        }
        break;
      }
      default: throw ParseError("colour encoding: transfer function enum", false);
    }
  }

  struct Tag { const char* sig; vec<uint8_t> data; int alias; };
  vec<Tag> tags;
  { Tag t{"desc", {}, -1}; Mluc(t.data, ColorDescription(ih)); tags.push_back(std::move(t)); }
  { Tag t{"cprt", {}, -1}; Mluc(t.data, "CC0"); tags.push_back(std::move(t)); }
  { Tag t{"wtpt", {}, -1}; Xyz(t.data, kD50); tags.push_back(std::move(t)); }
  { Tag t{"chad", {}, -1}; PutTag(t.data, "sf32"); Put32(t.data, 0); for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) PutS15(t.data, chad.m[i][j]); tags.push_back(std::move(t)); }
  if (!grey) {
    const char* names[3] = {"rXYZ", "gXYZ", "bXYZ"};
    for (int c = 0; c < 3; c++) { Tag t{names[c], {}, -1}; Xyz(t.data, col[c]); tags.push_back(std::move(t)); }
    const int first = (int)tags.size();
    tags.push_back(Tag{"rTRC", trc, -1});
    tags.push_back(Tag{"gTRC", {}, first});
    tags.push_back(Tag{"bTRC", {}, first});
  } else {
    tags.push_back(Tag{"kTRC", trc, -1});
  }
  if (!ih.have_gamma && (ih.tf == 16 || ih.tf == 18) && !grey) {
    // cicp (ICC v4.4): colour primaries, transfer characteristics, matrix coefficients (0: RGB), full range — where H.273 names the primaries
    int prim = -1;
    if (ih.primaries == 1 && ih.white_point == 1) prim = 1;            // BT.709 / sRGB
    else if (ih.primaries == 9 && ih.white_point == 1) prim = 9;       // BT.2020 / BT.2100
    else if (ih.primaries == 11 && ih.white_point == 11) prim = 11;    // SMPTE RP 431-2 (DCI white)
    else if (ih.primaries == 11 && ih.white_point == 1) prim = 12;     // SMPTE EG 432-1 (P3 D65)
    if (prim >= 0) { Tag t{"cicp", {}, -1}; PutTag(t.data, "cicp"); Put32(t.data, 0); t.data.push_back((uint8_t)prim); t.data.push_back((uint8_t)ih.tf); t.data.push_back(0); t.data.push_back(1); tags.push_back(std::move(t)); }
  }

  vec<uint8_t> out;
  Put32(out, 0);                      // size, patched below
  PutTag(out, "jxl ");                // preferred CMM
  Put32(out, 0x04400000);             // version 4.4
  PutTag(out, "mntr");
  PutTag(out, grey ? "GRAY" : "RGB ");
  PutTag(out, "XYZ ");
  Put16(out, 2019); Put16(out, 12); Put16(out, 1); Put16(out, 0); Put16(out, 0); Put16(out, 0);
  PutTag(out, "acsp");
  PutTag(out, "APPL");
  Put32(out, 0);                      // flags
  Put32(out, 0); Put32(out, 0);       // manufacturer, model
  Put32(out, 0); Put32(out, 0);       // attributes
  Put32(out, ih.rendering_intent & 3);
  for (int i = 0; i < 3; i++) PutS15(out, kD50[i]);
  PutTag(out, "jxl ");                // creator
  out.resize(128, 0);                 // profile ID + reserved
  Put32(out, (uint32_t)tags.size());
  const size_t table = out.size();
  out.resize(table + 12 * tags.size(), 0);
  vec<std::pair<uint32_t, uint32_t>> where(tags.size());
  for (size_t i = 0; i < tags.size(); i++) {
    if (tags[i].alias >= 0) { where[i] = where[tags[i].alias]; continue; }
    where[i] = {(uint32_t)out.size(), (uint32_t)tags[i].data.size()};
    out.insert(out.end(), tags[i].data.begin(), tags[i].data.end());
    while (out.size() % 4) out.push_back(0);
  }
  for (size_t i = 0; i < tags.size(); i++) {
    memcpy(&out[table + 12 * i], tags[i].sig, 4);
    Set32(out, table + 12 * i + 4, where[i].first);
    Set32(out, table + 12 * i + 8, where[i].second);
  }
  Set32(out, 0, (uint32_t)out.size());
  // profile ID: MD5 with flags, rendering intent and the ID field zeroed
  vec<uint8_t> z(out);
  memset(&z[44], 0, 4); memset(&z[64], 0, 4); memset(&z[84], 0, 16);
  Md5().Digest(z, &out[84]);
  return out;
}

}  // namespace jxlhip
