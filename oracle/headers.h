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
  int num_color_channels() const { return color.color_space == 1 ? 1 : 3; }
};

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
      if (m.have_preview) JXLO_FAIL("unsupported: preview frame");
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
  if (m.color.want_icc) JXLO_FAIL("unsupported: embedded ICC profile");
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
