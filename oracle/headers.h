// ORACLE — TEST INFRASTRUCTURE ONLY (see bits.h).
// Image / frame headers and TOC: restates libjxl v0.11.2 lib/jxl/{headers.cc,image_metadata.cc,
// color_encoding_internal.cc,frame_header.cc,loop_filter.cc,toc.cc,coeff_order.cc(ReadPermutation)}.
// SURVEY.md App. B.1-B.3 ([V] for every field value that occurs in the reference fixtures).
#pragma once
#include "entropy.h"

namespace jxlo {

struct BitDepth { bool float_sample = false; uint32_t bits = 8, exp_bits = 0; };
struct ExtraChannelInfo {
  uint32_t type = 0;  // 0 = alpha
  BitDepth depth;
  uint32_t dim_shift = 0;
  std::string name;
  bool alpha_associated = false;
  float spot[4] = {0, 0, 0, 0};
  uint32_t cfa = 1;
};
struct ColorEncoding {
  bool all_default = true, want_icc = false;
  uint32_t color_space = 0;  // 0 RGB 1 Gray 2 XYB 3 Unknown
  uint32_t white_point = 1, primaries = 1;
  bool have_gamma = false;
  uint32_t gamma = 0;
  uint32_t tf = 13;  // sRGB
  uint32_t rendering_intent = 1;
  int32_t custom_xy[8] = {0};
};
struct ImageMetadata {
  uint32_t xsize = 0, ysize = 0;
  bool all_default = true;
  uint32_t orientation = 1;
  bool have_intrinsic = false; uint32_t intrinsic_x = 0, intrinsic_y = 0;
  bool have_preview = false, have_animation = false;
  uint32_t preview_x = 0, preview_y = 0;
  uint32_t tps_num = 0, tps_den = 0, num_loops = 0; bool have_timecodes = false;
  BitDepth depth;
  bool modular_16bit = true;
  std::vector<ExtraChannelInfo> extra;
  bool xyb_encoded = true;
  ColorEncoding color;
  float intensity_target = 255.f, min_nits = 0.f, linear_below = 0.f;
  bool relative_to_max_display = false;
  // CustomTransformData
  bool default_m = true;
  float opsin_inv[9];
  float opsin_bias[3];
  float quant_bias[4];
  uint32_t cw_mask = 0;
  std::vector<float> up2, up4, up8;
  std::vector<uint8_t> icc;   // embedded ICC profile (color.want_icc), decoded
  int num_color_channels() const { return color.color_space == 1 ? 1 : 3; }
};

// ---- embedded ICC profile: icc_codec.cc (ICCReader + UnpredictICC) [R] -----------------------------------------------
// Byte stream under 41 contexts (kind of the previous byte x kind of the one before), then the "predicted ICC" container:
// varint(size of the profile) varint(size of the command stream) commands data.
namespace icc_detail {
inline int ByteKind1(int b) {
  if (('a' <= b && b <= 'z') || ('A' <= b && b <= 'Z')) return 0;
  if (('0' <= b && b <= '9') || b == '.' || b == ',') return 1;
  if (b <= 1) return 2 + b;
  if (b < 16) return 4;
  if (b > 240 && b < 255) return 5;
  if (b == 255) return 6;
  return 7;
}
inline int ByteKind2(int b) {
  if (('a' <= b && b <= 'z') || ('A' <= b && b <= 'Z')) return 0;
  if (('0' <= b && b <= '9') || b == '.' || b == ',') return 1;
  if (b < 16) return 2;
  if (b > 240) return 3;
  return 4;
}
inline int Context(size_t i, int b1, int b2) { return i <= 128 ? 0 : 1 + ByteKind1(b1) + 8 * ByteKind2(b2); }

struct Cursor {
  const std::vector<uint8_t>& v;
  uint64_t varint(size_t& at) const {
    uint64_t r = 0;
    for (int k = 0; k < 10; k++) {
      if (at >= v.size()) JXLO_FAIL("ICC: varint out of bounds");
      uint8_t b = v[at++];
      r |= (uint64_t)(b & 0x7F) << (7 * k);
      if (b < 0x80) return r;
    }
    JXLO_FAIL("ICC: varint too long");
  }
};
inline void be32(std::vector<uint8_t>& o, uint64_t x) { for (int s = 24; s >= 0; s -= 8) o.push_back((uint8_t)(x >> s)); }
inline void word(std::vector<uint8_t>& o, const std::string& w) { o.insert(o.end(), w.begin(), w.end()); }
// transposes a (width x ceil(n / width)) layout back: output i takes input i / width + (i % width) * rows, ragged last column
inline std::vector<uint8_t> Unshuffled(const uint8_t* in, size_t n, size_t width) {
  std::vector<uint8_t> out(n);
  size_t rows = (n + width - 1) / width, start = 0, j = 0;
  for (size_t i = 0; i < n; i++) { out[i] = in[j]; j += rows; if (j >= n) j = ++start; }
  return out;
}
inline uint64_t Extrapolate(int order, uint64_t a, uint64_t b, uint64_t c) { return order == 0 ? a : order == 1 ? 2 * a - b : 3 * a - 3 * b + c; }
}  // namespace icc_detail

inline std::vector<uint8_t> UnpredictICC(const std::vector<uint8_t>& enc) {
  using namespace icc_detail;
  Cursor cur{enc};
  const size_t n = enc.size();
  size_t data = 0;
  const uint64_t out_size = cur.varint(data);
  const uint64_t cmd_size = cur.varint(data);
  if (out_size > 0xFFFFFFFFull || cmd_size > 0xFFFFFFFFull || cmd_size > n - data) JXLO_FAIL("ICC: sizes");
  size_t cmd = data;
  const size_t cmd_end = data + (size_t)cmd_size;
  data = cmd_end;
  std::vector<uint8_t> icc;
  // 128-byte header as differences from a guess
  std::vector<uint8_t> guess(128, 0);
  guess[0] = (uint8_t)(out_size >> 24); guess[1] = (uint8_t)(out_size >> 16); guess[2] = (uint8_t)(out_size >> 8); guess[3] = (uint8_t)out_size;
  guess[8] = 4;
  const char* fixed[4] = {"mntr", "RGB ", "XYZ ", "acsp"};
  const int fixed_at[4] = {12, 16, 20, 36};
  for (int k = 0; k < 4; k++) memcpy(&guess[fixed_at[k]], fixed[k], 4);
  const uint8_t d50[12] = {0, 0, 246, 214, 0, 1, 0, 0, 0, 0, 211, 45};
  memcpy(&guess[68], d50, 12);
  for (size_t i = 0;; i++) {
    if (icc.size() == out_size) { if (cmd != cmd_end || data != n) JXLO_FAIL("ICC: trailing data"); return icc; }
    if (i == 128) break;
    if (i == 8) memcpy(&guess[80], &icc[4], 4);                                     // creator guess = preferred CMM
    if (i == 41) { if (icc[40] == 'A') memcpy(&guess[41], "PPL", 3); else if (icc[40] == 'M') memcpy(&guess[41], "SFT", 3); }
    if (i == 42) { if (icc[40] == 'S' && icc[41] == 'G') memcpy(&guess[42], "I ", 2); else if (icc[40] == 'S' && icc[41] == 'U') memcpy(&guess[42], "NW", 2); }
    if (data >= n) JXLO_FAIL("ICC: header data missing");
    icc.push_back((uint8_t)(enc[data++] + guess[i]));
  }
  if (cmd >= cmd_end) JXLO_FAIL("ICC: commands missing");
  // tag table
  static const char* kKnownTags[17] = {"cprt", "wtpt", "bkpt", "rXYZ", "gXYZ", "bXYZ", "kXYZ", "rTRC", "gTRC", "bTRC", "kTRC", "chad", "desc", "chrm", "dmnd", "dmdd", "lumi"};
  uint64_t ntags_plus1 = cur.varint(cmd);
  if (ntags_plus1) {
    const uint64_t ntags = ntags_plus1 - 1;
    if (ntags > 0xFFFFFFFFull) JXLO_FAIL("ICC: tag count");
    be32(icc, ntags);
    uint64_t last_start = 128 + 12 * ntags, last_size = 0;
    while (true) {
      if (icc.size() > out_size || cmd > cmd_end) JXLO_FAIL("ICC: tag table overrun");
      if (cmd == cmd_end) break;
      const int c = enc[cmd++], code = c & 63;
      if (code == 0) break;
      std::string name;
      if (code == 1) { if (n - data < 4) JXLO_FAIL("ICC: tag name missing"); name.assign((const char*)&enc[data], 4); data += 4; }
      else if (code == 2) name = "rTRC";
      else if (code == 3) name = "rXYZ";
      else if (code - 4 < 17) name = kKnownTags[code - 4];
      else JXLO_FAIL("ICC: tag code");
      word(icc, name);
      uint64_t size = last_size;
      if (name == "rXYZ" || name == "gXYZ" || name == "bXYZ" || name == "kXYZ" || name == "wtpt" || name == "bkpt" || name == "lumi") size = 20;
      uint64_t start = last_start + last_size;
      if (c & 64) { if (cmd >= cmd_end) JXLO_FAIL("ICC: tag offset missing"); start = cur.varint(cmd); }
      if (start > 0xFFFFFFFFull) JXLO_FAIL("ICC: tag offset");
      be32(icc, start);
      if (c & 128) { if (cmd >= cmd_end) JXLO_FAIL("ICC: tag size missing"); size = cur.varint(cmd); }
      if (size > 0xFFFFFFFFull) JXLO_FAIL("ICC: tag size");
      be32(icc, size);
      last_start = start; last_size = size;
      if (code == 2) for (const char* t : {"gTRC", "bTRC"}) { word(icc, t); be32(icc, start); be32(icc, size); }
      if (code == 3) {
        if (start + 2 * size > 0xFFFFFFFFull) JXLO_FAIL("ICC: tag offset");
        word(icc, "gXYZ"); be32(icc, start + size); be32(icc, size);
        word(icc, "bXYZ"); be32(icc, start + 2 * size); be32(icc, size);
      }
    }
  }
  // tag data
  static const char* kKnownTypes[8] = {"XYZ ", "desc", "text", "mluc", "para", "curv", "sf32", "gbd "};
  while (true) {
    if (icc.size() > out_size || cmd > cmd_end) JXLO_FAIL("ICC: content overrun");
    if (cmd == cmd_end) break;
    const int c = enc[cmd++];
    if (c == 1 || c == 2 || c == 3) {
      if (cmd >= cmd_end) JXLO_FAIL("ICC: count missing");
      const uint64_t count = cur.varint(cmd);
      if (count > n - data) JXLO_FAIL("ICC: data missing");
      if (c == 1) icc.insert(icc.end(), enc.begin() + data, enc.begin() + data + count);
      else { auto u = Unshuffled(&enc[data], (size_t)count, c == 2 ? 2 : 4); icc.insert(icc.end(), u.begin(), u.end()); }
      data += (size_t)count;
    } else if (c == 4) {
      if (cmd_end - cmd < 2) JXLO_FAIL("ICC: predictor flags missing");
      const int flags = enc[cmd++];
      const size_t width = (flags & 3) + 1;
      const int order = (flags >> 2) & 3;
      if (width == 3 || order == 3) JXLO_FAIL("ICC: predictor parameters");
      uint64_t stride = width;
      if (flags & 16) { if (cmd >= cmd_end) JXLO_FAIL("ICC: stride missing"); stride = cur.varint(cmd); if (stride < width) JXLO_FAIL("ICC: stride"); }
      if (icc.empty() || ((icc.size() - 1) >> 2) < stride) JXLO_FAIL("ICC: stride too large");
      if (cmd >= cmd_end) JXLO_FAIL("ICC: count missing");
      const uint64_t count = cur.varint(cmd);
      if (count > n - data) JXLO_FAIL("ICC: data missing");
      std::vector<uint8_t> res(enc.begin() + data, enc.begin() + data + count);
      if (width > 1) res = Unshuffled(res.data(), res.size(), width);
      const size_t base = icc.size();
      for (size_t i = 0; i < res.size(); i++) {
        const size_t unit = base + i - i % width;       // first byte of the big-endian value this byte belongs to
        uint64_t past[3];
        for (int k = 0; k < 3; k++) { uint64_t v = 0; for (size_t b = 0; b < width; b++) v = (v << 8) | icc[unit - (size_t)stride * (k + 1) + b]; past[k] = v; }
        const uint64_t pred = Extrapolate(order, past[0], past[1], past[2]);
        const int shift = (int)(8 * (width - 1 - i % width));
        icc.push_back((uint8_t)((pred >> shift) + res[i]));
      }
      data += (size_t)count;
    } else if (c == 10) {
      word(icc, "XYZ "); be32(icc, 0);
      if (n - data < 12) JXLO_FAIL("ICC: XYZ data missing");
      icc.insert(icc.end(), enc.begin() + data, enc.begin() + data + 12); data += 12;
    } else if (c >= 16 && c < 24) { word(icc, kKnownTypes[c - 16]); be32(icc, 0); }
    else JXLO_FAIL("ICC: unknown command");
  }
  if (data != n || icc.size() != out_size) JXLO_FAIL("ICC: size mismatch");
  return icc;
}

inline void ReadEmbeddedICC(BitReader& br, std::vector<uint8_t>& icc) {
  const uint64_t enc_size = U64(br);
  if (enc_size > (1ull << 28)) JXLO_FAIL("ICC: encoded size");
  EntropyCode ec;
  ReadEntropyCode(br, 41, ec);
  SymbolReader sr;
  sr.Init(&ec, br);
  std::vector<uint8_t> enc((size_t)enc_size);
  for (size_t i = 0; i < enc.size(); i++) {
    const uint32_t v = sr.Read(br, icc_detail::Context(i, i ? enc[i - 1] : 0, i > 1 ? enc[i - 2] : 0));
    if (v > 255) JXLO_FAIL("ICC: byte out of range");
    enc[i] = (uint8_t)v;
  }
  if (!sr.CheckFinal()) JXLO_FAIL("ICC: ANS final state");
  icc = UnpredictICC(enc);
}

inline void ReadSize(BitReader& br, uint32_t& xs, uint32_t& ys) {
  bool small = br.Bool();
  if (small) ys = (br.u(5) + 1) * 8;
  else ys = U32(br, BitsOffset(9, 1), BitsOffset(13, 1), BitsOffset(18, 1), BitsOffset(30, 1));
  uint32_t ratio = br.u(3);
  if (ratio == 0) {
    if (small) xs = (br.u(5) + 1) * 8;
    else xs = U32(br, BitsOffset(9, 1), BitsOffset(13, 1), BitsOffset(18, 1), BitsOffset(30, 1));
  } else {
    static const uint32_t num[8] = {0, 1, 12, 4, 3, 16, 5, 2}, den[8] = {1, 1, 10, 3, 2, 9, 4, 1};
    xs = (uint32_t)((uint64_t)ys * num[ratio] / den[ratio]);
  }
}

// headers.cc PreviewHeader: the size of the preview frame that precedes the frames of the image (at most 4096 x 4096)
inline void ReadPreviewSize(BitReader& br, uint32_t& xs, uint32_t& ys) {
  const bool div8 = br.Bool();
  if (div8) ys = 8 * U32(br, Val(16), Val(32), BitsOffset(5, 1), BitsOffset(9, 33));
  else ys = U32(br, BitsOffset(6, 1), BitsOffset(8, 65), BitsOffset(10, 321), BitsOffset(12, 1345));
  const uint32_t ratio = br.u(3);
  if (ratio == 0) {
    if (div8) xs = 8 * U32(br, Val(16), Val(32), BitsOffset(5, 1), BitsOffset(9, 33));
    else xs = U32(br, BitsOffset(6, 1), BitsOffset(8, 65), BitsOffset(10, 321), BitsOffset(12, 1345));
  } else {
    static const uint32_t num[8] = {0, 1, 12, 4, 3, 16, 5, 2}, den[8] = {1, 1, 10, 3, 2, 9, 4, 1};
    xs = (uint32_t)((uint64_t)ys * num[ratio] / den[ratio]);
  }
  if (xs > 4096 || ys > 4096) JXLO_FAIL("preview too large");
}

inline void ReadBitDepth(BitReader& br, BitDepth& d) {
  d.float_sample = br.Bool();
  if (!d.float_sample) { d.bits = U32(br, Val(8), Val(10), Val(12), BitsOffset(6, 1)); d.exp_bits = 0; }
  else { d.bits = U32(br, Val(32), Val(16), Val(24), BitsOffset(6, 1)); d.exp_bits = br.u(4) + 1; }
}

inline std::string ReadName(BitReader& br) {
  uint32_t n = U32(br, Val(0), Bits(4), BitsOffset(5, 16), BitsOffset(10, 48));
  std::string s;
  for (uint32_t i = 0; i < n; i++) s.push_back((char)br.u(8));
  return s;
}

inline int32_t ReadCustomXY(BitReader& br) {
  return UnpackSigned(U32(br, Bits(19), BitsOffset(19, 524288), BitsOffset(20, 1048576), BitsOffset(21, 2097152)));
}

inline void ReadColorEncoding(BitReader& br, ColorEncoding& c) {
  c = ColorEncoding();
  c.all_default = br.Bool();
  if (c.all_default) return;
  c.want_icc = br.Bool();
  c.color_space = Enum(br);
  if (c.want_icc) return;
  if (c.color_space != 2) {
    c.white_point = Enum(br);
    if (c.white_point == 2) { c.custom_xy[0] = ReadCustomXY(br); c.custom_xy[1] = ReadCustomXY(br); }
  }
  if (c.color_space != 2 && c.color_space != 1) {
    c.primaries = Enum(br);
    if (c.primaries == 2) for (int i = 2; i < 8; i++) c.custom_xy[i] = ReadCustomXY(br);
  }
  if (c.color_space != 2) {  // XYB has an implicit transfer function
    c.have_gamma = br.Bool();
    if (c.have_gamma) c.gamma = br.u(24);
    else c.tf = Enum(br);
  }
  c.rendering_intent = Enum(br);
}

static const float kDefaultInverseOpsin[9] = {11.031566901960783f,  -9.866943921568629f, -0.16462299647058826f,
                                              -3.254147380392157f,  4.418770392156863f,  -0.16462299647058826f,
                                              -3.6588512862745097f, 2.7129230470588235f, 1.9459282392156863f};
static const float kOpsinBias = 0.0037930732552754493f;

// returns position of the first frame (byte aligned). Container must have been stripped already.
inline void ReadImageHeaders(BitReader& br, ImageMetadata& m) {
  if (br.u(16) != 0x0AFF) JXLO_FAIL("not a JXL codestream");
  ReadSize(br, m.xsize, m.ysize);
  m.all_default = br.Bool();
  bool extra_fields = false;
  if (!m.all_default) {
    extra_fields = br.Bool();
    if (extra_fields) {
      m.orientation = br.u(3) + 1;
      m.have_intrinsic = br.Bool();
      if (m.have_intrinsic) ReadSize(br, m.intrinsic_x, m.intrinsic_y);
      m.have_preview = br.Bool();
      if (m.have_preview) ReadPreviewSize(br, m.preview_x, m.preview_y);
      m.have_animation = br.Bool();
      if (m.have_animation) {
        m.tps_num = U32(br, Val(100), Val(1000), BitsOffset(10, 1), BitsOffset(30, 1));
        m.tps_den = U32(br, Val(1), Val(1001), BitsOffset(8, 1), BitsOffset(10, 1));
        m.num_loops = U32(br, Val(0), Bits(3), Bits(16), Bits(32));
        m.have_timecodes = br.Bool();
      }
    }
    ReadBitDepth(br, m.depth);
    m.modular_16bit = br.Bool();
    uint32_t nextra = U32(br, Val(0), Val(1), BitsOffset(4, 2), BitsOffset(12, 1));
    m.extra.resize(nextra);
    for (auto& e : m.extra) {
      bool d_alpha = br.Bool();
      if (d_alpha) continue;
      e.type = Enum(br);
      ReadBitDepth(br, e.depth);
      e.dim_shift = U32(br, Val(0), Val(3), Val(4), BitsOffset(3, 1));
      e.name = ReadName(br);
      if (e.type == 0) e.alpha_associated = br.Bool();
      if (e.type == 2) for (int i = 0; i < 4; i++) e.spot[i] = F16(br);
      if (e.type == 5) e.cfa = U32(br, Val(1), Bits(2), BitsOffset(4, 3), BitsOffset(8, 19));
    }
    m.xyb_encoded = br.Bool();
    ReadColorEncoding(br, m.color);
    if (extra_fields) {
      bool tm_default = br.Bool();
      if (!tm_default) {
        m.intensity_target = F16(br);
        m.min_nits = F16(br);
        m.relative_to_max_display = br.Bool();
        m.linear_below = F16(br);
      }
    }
    SkipExtensions(br);
  }
  m.default_m = br.Bool();
  for (int i = 0; i < 9; i++) m.opsin_inv[i] = kDefaultInverseOpsin[i];
  for (int i = 0; i < 3; i++) m.opsin_bias[i] = -kOpsinBias;
  m.quant_bias[0] = 1.0f - 0.05465007330715401f; m.quant_bias[1] = 1.0f - 0.07005449891748593f;
  m.quant_bias[2] = 1.0f - 0.049935103337343655f; m.quant_bias[3] = 0.145f;
  if (!m.default_m) {
    if (m.xyb_encoded) {
      bool opsin_default = br.Bool();
      if (!opsin_default) {
        for (int i = 0; i < 9; i++) m.opsin_inv[i] = F16(br);
        for (int i = 0; i < 3; i++) m.opsin_bias[i] = F16(br);
        for (int i = 0; i < 4; i++) m.quant_bias[i] = F16(br);
      }
    }
    m.cw_mask = br.u(3);
    if (m.cw_mask & 1) { m.up2.resize(15); for (auto& v : m.up2) v = F16(br); }
    if (m.cw_mask & 2) { m.up4.resize(55); for (auto& v : m.up4) v = F16(br); }
    if (m.cw_mask & 4) { m.up8.resize(210); for (auto& v : m.up8) v = F16(br); }
  }
  if (m.color.want_icc) ReadEmbeddedICC(br, m.icc);
  br.byte_align();
}

// ---- frame header ------------------------------------------------------------------------------------------------
struct BlendInfo { uint32_t mode = 0, alpha_channel = 0, source = 0; bool clamp = false; };
struct Passes {
  uint32_t num_passes = 1, num_ds = 0;
  uint32_t shift[11] = {0}, downsample[4] = {0}, last_pass[4] = {0};
};
struct LoopFilter {
  bool gab = true;
  float gab_w[6] = {0.115169525f, 0.061248592f, 0.115169525f, 0.061248592f, 0.115169525f, 0.061248592f};
  uint32_t epf_iters = 2;
  float sharp_lut[8] = {0.f, 1.f / 7, 2.f / 7, 3.f / 7, 4.f / 7, 5.f / 7, 6.f / 7, 1.f};
  float channel_scale[3] = {40.0f, 5.0f, 3.5f};
  float pass1_zeroflush = 0.45f, pass2_zeroflush = 0.6f;
  float quant_mul = 0.46f, pass0_sigma_scale = 0.9f, pass2_sigma_scale = 6.5f, border_sad_mul = 2.0f / 3.0f;
  float sigma_for_modular = 1.0f;
};
enum FrameType { kRegular = 0, kLFFrame = 1, kReferenceOnly = 2, kSkipProgressive = 3 };
enum FrameFlags { kNoise = 1, kPatches = 2, kSplines = 16, kUseLfFrame = 32, kSkipAdaptiveLFSmoothing = 128 };
struct FrameHeader {
  uint32_t type = 0;
  bool modular = false;
  uint64_t flags = 0;
  bool do_ycbcr = false;
  uint32_t jpeg_upsampling[3] = {0, 0, 0};
  uint32_t upsampling = 1;
  std::vector<uint32_t> ec_upsampling;
  uint32_t group_size_shift = 1;
  uint32_t x_qm_scale = 3, b_qm_scale = 2;
  Passes passes;
  uint32_t lf_level = 0;
  bool have_crop = false;
  int32_t x0 = 0, y0 = 0;
  uint32_t xsize = 0, ysize = 0;  // frame size in image pixels (before upsampling division)
  BlendInfo blend;
  std::vector<BlendInfo> ec_blend;
  uint32_t duration = 0, timecode = 0;
  bool is_last = true;
  uint32_t save_as_reference = 0;
  bool save_before_ct = false;
  std::string name;
  LoopFilter lf;
  // derived
  uint32_t width = 0, height = 0;  // decoded size (after dividing by upsampling / lf_level)
  uint32_t group_dim = 256;
  uint32_t num_groups = 0, num_lf_groups = 0, xgroups = 0, ygroups = 0, xlfgroups = 0, ylfgroups = 0;
  size_t toc_entries() const {
    if (num_groups == 1 && passes.num_passes == 1) return 1;
    return 1 + num_lf_groups + 1 + (size_t)num_groups * passes.num_passes;
  }
};

inline void ReadBlend(BitReader& br, BlendInfo& b, size_t num_extra, bool partial) {
  b.mode = U32(br, Val(0), Val(1), Val(2), BitsOffset(2, 3));
  if (num_extra > 0 && (b.mode == 2 || b.mode == 3)) b.alpha_channel = U32(br, Val(0), Val(1), Val(2), BitsOffset(3, 3));
  if (num_extra > 0 && (b.mode == 2 || b.mode == 3 || b.mode == 4)) b.clamp = br.Bool();
  if (b.mode != 0 || partial) b.source = br.u(2);
}

inline void ReadFrameHeader(BitReader& br, const ImageMetadata& m, FrameHeader& f) {
  f = FrameHeader();
  const size_t num_extra = m.extra.size();
  f.ec_upsampling.assign(num_extra, 1);
  f.ec_blend.assign(num_extra, BlendInfo());
  bool all_default = br.Bool();
  f.xsize = m.xsize; f.ysize = m.ysize;
  bool xyb = m.xyb_encoded;
  if (!all_default) {
    f.type = br.u(2);
    f.modular = br.u(1) != 0;
    f.flags = U64(br);
    if (!m.xyb_encoded) f.do_ycbcr = br.Bool();
    if (f.do_ycbcr && !(f.flags & kUseLfFrame)) for (int i = 0; i < 3; i++) f.jpeg_upsampling[i] = br.u(2);
    if (!(f.flags & kUseLfFrame)) {
      f.upsampling = U32(br, Val(1), Val(2), Val(4), Val(8));
      for (size_t i = 0; i < num_extra; i++) f.ec_upsampling[i] = U32(br, Val(1), Val(2), Val(4), Val(8));
    }
    if (f.modular) f.group_size_shift = br.u(2);
    if (!f.modular && xyb) { f.x_qm_scale = br.u(3); f.b_qm_scale = br.u(3); }
    else if (!xyb) { f.x_qm_scale = 2; f.b_qm_scale = 2; }
    if (f.type != kReferenceOnly) {
      Passes& p = f.passes;
      p.num_passes = U32(br, Val(1), Val(2), Val(3), BitsOffset(3, 4));
      if (p.num_passes != 1) {
        p.num_ds = U32(br, Val(0), Val(1), Val(2), BitsOffset(1, 3));
        for (uint32_t i = 0; i + 1 < p.num_passes; i++) p.shift[i] = br.u(2);
        for (uint32_t i = 0; i < p.num_ds; i++) p.downsample[i] = U32(br, Val(1), Val(2), Val(4), Val(8));
        for (uint32_t i = 0; i < p.num_ds; i++) p.last_pass[i] = U32(br, Val(0), Val(1), Val(2), Bits(3));
      }
    }
    bool partial = false;
    if (f.type == kLFFrame) {
      f.lf_level = U32(br, Val(1), Val(2), Val(3), Val(4));
    } else {
      f.have_crop = br.Bool();
      if (f.have_crop) {
        if (f.type != kReferenceOnly) {
          f.x0 = UnpackSigned(U32(br, Bits(8), BitsOffset(11, 256), BitsOffset(14, 2304), BitsOffset(30, 18688)));
          f.y0 = UnpackSigned(U32(br, Bits(8), BitsOffset(11, 256), BitsOffset(14, 2304), BitsOffset(30, 18688)));
        }
        f.xsize = U32(br, Bits(8), BitsOffset(11, 256), BitsOffset(14, 2304), BitsOffset(30, 18688));
        f.ysize = U32(br, Bits(8), BitsOffset(11, 256), BitsOffset(14, 2304), BitsOffset(30, 18688));
        partial = f.x0 > 0 || f.y0 > 0 || (int64_t)f.xsize + f.x0 < (int64_t)m.xsize || (int64_t)f.ysize + f.y0 < (int64_t)m.ysize;
      }
    }
    if (f.type == kRegular || f.type == kSkipProgressive) {
      ReadBlend(br, f.blend, num_extra, partial);
      for (size_t i = 0; i < num_extra; i++) ReadBlend(br, f.ec_blend[i], num_extra, partial);
      if (m.have_animation) {
        f.duration = U32(br, Val(0), Val(1), Bits(8), Bits(32));
        if (m.have_timecodes) f.timecode = br.u(32);
      }
      f.is_last = br.Bool();
    } else {
      f.is_last = false;
    }
    if (f.type != kLFFrame && !f.is_last) f.save_as_reference = br.u(2);
    bool can_ref = !f.is_last && f.type != kLFFrame && (f.duration == 0 || f.save_as_reference != 0);
    bool full_replace = (f.type == kRegular || f.type == kSkipProgressive) && f.blend.mode == 0 && !partial;
    if (f.type == kReferenceOnly || (can_ref && full_replace)) f.save_before_ct = br.Bool();
    else f.save_before_ct = false;
    f.name = ReadName(br);
    // RestorationFilter
    LoopFilter& lf = f.lf;
    bool lf_default = br.Bool();
    if (!lf_default) {
      lf.gab = br.Bool();
      if (lf.gab) {
        bool custom = br.Bool();
        if (custom) for (int i = 0; i < 6; i++) lf.gab_w[i] = F16(br);
      }
      lf.epf_iters = br.u(2);
      if (lf.epf_iters > 0) {
        if (!f.modular) {
          bool sharp_custom = br.Bool();
          if (sharp_custom) for (int i = 0; i < 8; i++) lf.sharp_lut[i] = F16(br);
        }
        bool weight_custom = br.Bool();
        if (weight_custom) {
          for (int i = 0; i < 3; i++) lf.channel_scale[i] = F16(br);
          lf.pass1_zeroflush = F16(br);
          lf.pass2_zeroflush = F16(br);
        }
        bool sigma_custom = br.Bool();
        if (sigma_custom) {
          if (!f.modular) lf.quant_mul = F16(br);
          lf.pass0_sigma_scale = F16(br);
          lf.pass2_sigma_scale = F16(br);
          lf.border_sad_mul = F16(br);
        }
        if (f.modular) lf.sigma_for_modular = F16(br);
      }
      SkipExtensions(br);
    }
    SkipExtensions(br);
  } else {
    if (!xyb) { f.x_qm_scale = 2; f.b_qm_scale = 2; }
  }
  // derived geometry
  uint32_t w = f.xsize, h = f.ysize;
  if (f.upsampling > 1) { w = (w + f.upsampling - 1) / f.upsampling; h = (h + f.upsampling - 1) / f.upsampling; }
  if (f.type == kLFFrame) { uint32_t d = 1u << (3 * f.lf_level); w = (w + d - 1) / d; h = (h + d - 1) / d; }
  f.width = w; f.height = h;
  f.group_dim = f.modular ? (128u << f.group_size_shift) : 256u;
  // VarDCT frames always use 256 (group_size_shift only read for modular; but modular sub-images of a VarDCT
  // frame use the same group_dim = 256)
  f.xgroups = (w + f.group_dim - 1) / f.group_dim;
  f.ygroups = (h + f.group_dim - 1) / f.group_dim;
  f.num_groups = f.xgroups * f.ygroups;
  f.xlfgroups = (w + f.group_dim * 8 - 1) / (f.group_dim * 8);
  f.ylfgroups = (h + f.group_dim * 8 - 1) / (f.group_dim * 8);
  f.num_lf_groups = f.xlfgroups * f.ylfgroups;
}

// coeff_order.cc ReadPermutation / DecodeLehmerCode
inline uint32_t CoeffOrderContext(uint32_t v) {
  uint32_t t = v == 0 ? 0 : 1 + FloorLog2(v);
  return std::min<uint32_t>(t, 7);
}
inline void ReadPermutation(BitReader& br, SymbolReader& sr, size_t skip, size_t size, std::vector<uint32_t>& perm) {
  std::vector<uint32_t> lehmer(size, 0);
  uint32_t end = sr.Read(br, CoeffOrderContext((uint32_t)size)) + (uint32_t)skip;
  if (end > size) JXLO_FAIL("bad permutation size");
  uint32_t last = 0;
  for (size_t i = skip; i < end; i++) {
    lehmer[i] = sr.Read(br, CoeffOrderContext(last));
    last = lehmer[i];
    if (lehmer[i] >= size - i) JXLO_FAIL("bad lehmer code");
  }
  std::vector<uint32_t> temp(size);
  for (size_t i = 0; i < size; i++) temp[i] = (uint32_t)i;
  perm.resize(size);
  for (size_t i = 0; i < size; i++) {
    perm[i] = temp[lehmer[i]];
    temp.erase(temp.begin() + lehmer[i]);
  }
}

struct Section { size_t offset, size; };
// toc.cc ReadToc + ReadGroupOffsets; on return br is at the start of the first section (byte aligned)
inline void ReadTOC(BitReader& br, size_t n, std::vector<Section>& sec) {
  bool permuted = br.Bool();
  std::vector<uint32_t> perm;
  if (permuted) {
    EntropyCode ec;
    ReadEntropyCode(br, 8, ec);
    SymbolReader sr;
    sr.Init(&ec, br);
    ReadPermutation(br, sr, 0, n, perm);
    if (!sr.CheckFinal()) JXLO_FAIL("toc permutation ANS final state");
  }
  br.byte_align();
  std::vector<size_t> sizes(n);
  for (size_t i = 0; i < n; i++) sizes[i] = U32(br, Bits(10), BitsOffset(14, 1024), BitsOffset(22, 17408), BitsOffset(30, 4211712));
  br.byte_align();
  size_t base = br.pos / 8;
  std::vector<Section> phys(n);
  size_t off = base;
  for (size_t i = 0; i < n; i++) { phys[i] = {off, sizes[i]}; off += sizes[i]; }
  sec.resize(n);
  for (size_t i = 0; i < n; i++) sec[i] = permuted ? phys[perm[i]] : phys[i];
  // sections end at `off`
  sec.push_back({off, 0});
}

// container (ISO-BMFF boxes): returns the concatenated codestream (decode.cc box scan). SURVEY B.1.
inline std::vector<uint8_t> ExtractCodestream(const uint8_t* data, size_t size, bool* had_container, bool* has_jbrd) {
  if (had_container) *had_container = false;
  if (has_jbrd) *has_jbrd = false;
  if (size >= 2 && data[0] == 0xFF && data[1] == 0x0A) return std::vector<uint8_t>(data, data + size);
  static const uint8_t sig[12] = {0, 0, 0, 0xC, 'J', 'X', 'L', ' ', 0xD, 0xA, 0x87, 0xA};
  if (size < 12 || memcmp(data, sig, 12) != 0) JXLO_FAIL("bad signature");
  if (had_container) *had_container = true;
  std::vector<uint8_t> out;
  size_t pos = 0;
  while (pos + 8 <= size) {
    uint64_t bs = ((uint64_t)data[pos] << 24) | (data[pos + 1] << 16) | (data[pos + 2] << 8) | data[pos + 3];
    const uint8_t* type = data + pos + 4;
    size_t hdr = 8;
    if (bs == 1) {
      if (pos + 16 > size) JXLO_FAIL("truncated box");
      bs = 0;
      for (int i = 0; i < 8; i++) bs = (bs << 8) | data[pos + 8 + i];
      hdr = 16;
    }
    size_t end = bs == 0 ? size : pos + bs;
    if (end > size || end < pos + hdr) JXLO_FAIL("truncated box");
    if (!memcmp(type, "jxlc", 4)) out.insert(out.end(), data + pos + hdr, data + end);
    else if (!memcmp(type, "jxlp", 4)) { if (end < pos + hdr + 4) JXLO_FAIL("bad jxlp"); out.insert(out.end(), data + pos + hdr + 4, data + end); }
    else if (!memcmp(type, "jbrd", 4)) { if (has_jbrd) *has_jbrd = true; }
    pos = end;
  }
  if (out.empty()) JXLO_FAIL("no codestream in container");
  return out;
}

}  // namespace jxlo
