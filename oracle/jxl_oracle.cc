// ORACLE — TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library; the product (jpegxl-rs_amd/) never links, imports or executes it.
//
// Whole-image CPU decoder restating libjxl v0.11.2's decode path (JxlDecoderProcessInput →
// lib/jxl/decode.cc → dec_frame.cc …), the dependency the reference calls at jpegxl-rs/src/decode.rs:231-238.
// libjxl is an un-vendored submodule (/root/reference/.gitmodules:1-3), so this restates the published
// algorithm (ISO/IEC 18181-1 + upstream v0.11.2) as digested in SURVEY.md App. B and is pinned on the
// reference's own fixtures: samples/sample.jxl ≡ sample.png (jpegxl-rs/src/image.rs:169), bench.jxl ≡ bench.png.
// VarDCT float pixel pipeline: PARITY UNPINNED against libjxl (no golden exists in the reference).
#include "frame.h"
#include <chrono>
#include <memory>

using namespace jxlo;

namespace {

struct Decoded {
  ImageMetadata meta;
  bool have_container = false, has_jbrd = false;
  int w = 0, h = 0;
  int num_color = 3;
  bool has_alpha = false;
  // final float channels (display-referred, nominal range [0,1]); 1 or 3 colour + optional alpha
  std::vector<Plane> color;
  Plane alpha;
  Dump dump;
  size_t tokens_lf = 0, tokens_hf = 0, tokens_modular = 0;
  double seconds = 0;
  std::string error;
};

void InitFrame(Frame& f, const ImageMetadata& m) {
  f.m = &m;
  f.w = (int)f.fh.width; f.h = (int)f.fh.height;
  f.bw = (f.w + 7) / 8; f.bh = (f.h + 7) / 8;
  f.cw = (f.bw + 7) / 8; f.chh = (f.bh + 7) / 8;
  if (!f.fh.modular) {
    size_t nb = (size_t)f.bw * f.bh;
    for (int c = 0; c < 3; c++) { f.lf.p[c] = Plane(f.bw, f.bh); f.lfq[c].assign(nb, 0); }
    f.strategy.assign(nb, 0); f.is_first.assign(nb, 0); f.hf_mul.assign(nb, 1); f.sharpness.assign(nb, 0);
    f.ytox_map.assign((size_t)f.cw * f.chh, 0); f.ytob_map.assign((size_t)f.cw * f.chh, 0);
    for (int c = 0; c < 3; c++) {
      f.coeffs[c].resize(f.fh.num_groups);
      for (auto& v : f.coeffs[c]) v.assign(65536, 0);
    }
  }
}

void DecodeFrameSections(const uint8_t* data, size_t size, BitReader& br, Frame& f) {
  size_t n = f.fh.toc_entries();
  std::vector<Section> sec;
  ReadTOC(br, n, sec);
  if (sec.back().offset > size) JXLO_FAIL("truncated frame");
  auto reader = [&](size_t i) { BitReader r(data + sec[i].offset, sec[i].size); return r; };
  if (n == 1) {
    BitReader r = reader(0);
    ReadLfGlobal(r, f);
    ReadLfGroup(r, f, 0);
    if (!f.fh.modular) ReadHfGlobal(r, f);
    ReadPassGroup(r, f, 0, 0);
    if (r.pos > r.size * 8) JXLO_FAIL("section overrun");
  } else {
    { BitReader r = reader(0); ReadLfGlobal(r, f); if (r.pos > r.size * 8) JXLO_FAIL("LfGlobal overrun"); }
    for (uint32_t g = 0; g < f.fh.num_lf_groups; g++) { BitReader r = reader(1 + g); ReadLfGroup(r, f, g); if (r.pos > r.size * 8) JXLO_FAIL("LfGroup overrun"); }
    if (!f.fh.modular) { BitReader r = reader(1 + f.fh.num_lf_groups); ReadHfGlobal(r, f); if (r.pos > r.size * 8) JXLO_FAIL("HfGlobal overrun"); }
    for (uint32_t p = 0; p < f.fh.passes.num_passes; p++)
      for (uint32_t g = 0; g < f.fh.num_groups; g++) {
        BitReader r = reader(2 + f.fh.num_lf_groups + p * f.fh.num_groups + g);
        ReadPassGroup(r, f, p, g);
        if (r.pos > r.size * 8) JXLO_FAIL("PassGroup overrun");
      }
  }
  br.pos = sec.back().offset * 8;
}

void StorePlanes(Dump& d, const char* prefix, const Image3& img) {
  static const char* n[3] = {"0", "1", "2"};
  for (int c = 0; c < 3; c++) d.planes[std::string(prefix) + n[c]] = img.p[c];
}

void DecodeImage(const uint8_t* data, size_t size, Decoded& out, bool want_dump) {
  std::vector<uint8_t> cs = ExtractCodestream(data, size, &out.have_container, &out.has_jbrd);
  BitReader br(cs.data(), cs.size());
  ImageMetadata& m = out.meta;
  ReadImageHeaders(br, m);
  out.w = (int)m.xsize; out.h = (int)m.ysize;
  for (;;) {
    Frame f;
    ReadFrameHeader(br, m, f.fh);
    if (f.fh.type != kRegular) JXLO_FAIL("unsupported: non-regular frame (reference / LF / skip-progressive frames)");
    const int up = (int)f.fh.upsampling;
    const float* up_weights = nullptr;
    if (up > 1) {
      if (f.fh.modular) JXLO_FAIL("unsupported: upsampling of a Modular frame");
      const std::vector<float>& cw = up == 2 ? m.up2 : up == 4 ? m.up4 : m.up8;
      if (!cw.empty()) up_weights = cw.data();
      else if (up == 2) up_weights = kDefaultUp2Weights;
      else JXLO_FAIL("unsupported: default 4x / 8x upsampling weights (tables not reproducible offline)");
    }
    if (f.fh.have_crop && (f.fh.x0 != 0 || f.fh.y0 != 0 || f.fh.xsize != m.xsize || f.fh.ysize != m.ysize)) JXLO_FAIL("unsupported: cropped frame");
    if (!f.fh.is_last) JXLO_FAIL("unsupported: multi-frame image");
    if (f.fh.do_ycbcr) for (int i = 0; i < 3; i++) if (f.fh.jpeg_upsampling[i]) JXLO_FAIL("unsupported: chroma subsampling");
    InitFrame(f, m);
    f.dump = want_dump ? &out.dump : nullptr;
    DecodeFrameSections(cs.data(), cs.size(), br, f);
    out.tokens_lf = f.tokens_lf; out.tokens_hf = f.tokens_hf; out.tokens_modular = f.tokens_modular;
    // undo global modular transforms
    if (!f.gimg.channel.empty()) UndoTransforms(f.gimg, f.gimg_header.wp);
    const int cw_ = f.w, ch_ = f.h;                         // coded size
    const int w = up > 1 ? out.w : f.w, h = up > 1 ? out.h : f.h;   // size after upsampling
    const bool gray = m.color.color_space == 1;
    out.num_color = gray ? 1 : 3;
    Image3 rgb;
    size_t first_extra = 0;
    if (!f.fh.modular) {
      // ---- VarDCT ----
      if (!(f.fh.flags & kSkipAdaptiveLFSmoothing) && !(f.fh.flags & kUseLfFrame)) {
        float fac[3];
        const float inv_quant_lf = InvGlobalScale(f) / (float)f.quant_lf;
        for (int c = 0; c < 3; c++) fac[c] = f.m_lf[c] * inv_quant_lf;
        if (want_dump) StorePlanes(out.dump, "lf_raw", f.lf);
        AdaptiveLFSmoothing(fac, f.lf);
      }
      if (want_dump) {
        StorePlanes(out.dump, "lf", f.lf);
        for (int c = 0; c < 3; c++) {
          std::vector<int32_t>& v = out.dump.ints[std::string("coeff") + char('0' + c)];
          for (auto& g : f.coeffs[c]) v.insert(v.end(), g.begin(), g.end());
        }
        for (int c = 0; c < 3; c++) out.dump.ints[std::string("lfq") + char('0' + c)] = f.lfq[c];
        std::vector<int32_t>& st = out.dump.ints["strategy"];
        std::vector<int32_t>& hm = out.dump.ints["hf_mul"];
        std::vector<int32_t>& sh = out.dump.ints["sharpness"];
        for (size_t i = 0; i < f.strategy.size(); i++) { st.push_back(f.is_first[i] ? f.strategy[i] : -1 - f.strategy[i]); hm.push_back(f.hf_mul[i]); sh.push_back(f.sharpness[i]); }
        std::vector<int32_t>& cf = out.dump.ints["cfl"];
        for (size_t i = 0; i < f.ytox_map.size(); i++) { cf.push_back(f.ytox_map[i]); cf.push_back(f.ytob_map[i]); }
      }
      DequantAndIDCT(f);
      if (want_dump) StorePlanes(out.dump, "idct", f.xyb);
      Image3 img = CropImage(f.xyb, cw_, ch_);
      if (f.fh.lf.gab) Gaborish(f.fh.lf, img);
      if (want_dump) StorePlanes(out.dump, "gab", img);
      if (f.fh.lf.epf_iters > 0) {
        std::vector<float> inv_sigma;
        ComputeInvSigma(f.fh.lf, (float)f.global_scale / 65536.0f, f.hf_mul, f.sharpness, f.bw, f.bh, inv_sigma);
        if (f.fh.lf.epf_iters >= 3) EPFPass(f.fh.lf, 0, inv_sigma, f.bw, img);
        EPFPass(f.fh.lf, 1, inv_sigma, f.bw, img);
        if (f.fh.lf.epf_iters >= 2) EPFPass(f.fh.lf, 2, inv_sigma, f.bw, img);
      }
      if (want_dump) StorePlanes(out.dump, "epf", img);
      if (up > 1) {   // stage_upsampling.cc: XYB planes, before the colour transform
        for (int c = 0; c < 3; c++) img.p[c] = UpsamplePlane(img.p[c], up, up_weights, out.w, out.h);
        if (want_dump) StorePlanes(out.dump, "ups", img);
      }
      rgb = img;
      if (m.xyb_encoded) {
        OpsinParams op = MakeOpsin(m, m.intensity_target);
        const bool linear_out = !m.color.all_default && !m.color.have_gamma && m.color.tf == 8;
        if (!m.color.all_default && !linear_out && !(m.color.tf == 13 && !m.color.have_gamma)) JXLO_FAIL("unsupported: output transfer function");
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
          float r, g, b;
          XybToLinear(op, img.p[0].row(y)[x], img.p[1].row(y)[x], img.p[2].row(y)[x], &r, &g, &b);
          if (!linear_out) { r = LinearToSRGB(r); g = LinearToSRGB(g); b = LinearToSRGB(b); }
          rgb.p[0].row(y)[x] = r; rgb.p[1].row(y)[x] = g; rgb.p[2].row(y)[x] = b;
        }
      } else if (f.fh.do_ycbcr) {
        // stage_ycbcr.cc: planes are Cb, Y, Cr
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
          float cb = img.p[0].row(y)[x], yy = img.p[1].row(y)[x], cr = img.p[2].row(y)[x];
          const float c128 = 128.0f / 255, crcr = 1.402f, cgcb = -0.114f * 1.772f / 0.587f, cgcr = -0.299f * 1.402f / 0.587f, cbcb = 1.772f;
          float yb = yy + c128;
          rgb.p[0].row(y)[x] = std::fmaf(crcr, cr, yb);
          rgb.p[1].row(y)[x] = std::fmaf(cgcr, cr, std::fmaf(cgcb, cb, yb));
          rgb.p[2].row(y)[x] = std::fmaf(cbcb, cb, yb);
        }
      }
      first_extra = 0;
      out.color.clear();
      if (gray) out.color.push_back(rgb.p[1]);  // all channels equal for gray
      else for (int c = 0; c < 3; c++) out.color.push_back(rgb.p[c]);
    } else {
      // ---- Modular ---- (dec_modular.cc ModularImageToDecodedRect)
      if (m.xyb_encoded) JXLO_FAIL("unsupported: XYB modular frame");
      if (m.depth.float_sample) JXLO_FAIL("unsupported: float modular samples");
      int nb = gray ? 1 : 3;
      if ((int)f.gimg.channel.size() < nb) JXLO_FAIL("missing colour channels");
      const float factor = 1.0f / (float)((1u << m.depth.bits) - 1);
      out.color.clear();
      for (int c = 0; c < nb; c++) {
        const Channel& ch = f.gimg.channel[c];
        JXLO_CHECK(ch.w == w && ch.h == h);
        Plane p(w, h);
        for (size_t i = 0; i < (size_t)w * h; i++) p.d[i] = (float)ch.data[i] * factor;
        out.color.push_back(p);
      }
      if (want_dump) for (int c = 0; c < nb; c++) out.dump.ints[std::string("modular") + char('0' + c)].assign(f.gimg.channel[c].data.begin(), f.gimg.channel[c].data.end());
      first_extra = nb;
    }
    // extra channels: first alpha channel
    out.has_alpha = false;
    for (size_t e = 0; e < m.extra.size(); e++) {
      if (m.extra[e].type != 0) continue;
      const Channel& ch = f.gimg.channel[first_extra + e];
      JXLO_CHECK(ch.w == cw_ && ch.h == ch_);
      const float factor = 1.0f / (float)((1u << m.extra[e].depth.bits) - 1);
      out.alpha = Plane(cw_, ch_);
      for (size_t i = 0; i < (size_t)cw_ * ch_; i++) out.alpha.d[i] = (float)ch.data[i] * factor;
      if (up > 1) out.alpha = UpsamplePlane(out.alpha, up, up_weights, out.w, out.h);   // ec_upsampling == upsampling (checked at parse)
      if (want_dump) out.dump.ints["alpha"].assign(ch.data.begin(), ch.data.end());
      out.has_alpha = true;
      break;
    }
    break;
  }
}

// stage_write.cc: clamp, scale, round-to-nearest-even, interleave (SURVEY b17 [V])
size_t WritePixels(const Decoded& d, int type /*0 u8,1 u16,2 f32,3 f16*/, int num_channels, int big_endian, size_t align, std::vector<uint8_t>& out) {
  const int w = d.w, h = d.h;
  const size_t bps = type == 0 ? 1 : type == 2 ? 4 : 2;
  size_t stride = (size_t)w * num_channels * bps;
  if (align > 1) stride = (stride + align - 1) / align * align;
  size_t total = stride * (h - 1) + (size_t)w * num_channels * bps;
  out.assign(total, 0);
  for (int y = 0; y < h; y++) {
    uint8_t* row = out.data() + stride * y;
    for (int x = 0; x < w; x++) {
      for (int c = 0; c < num_channels; c++) {
        float v;
        bool is_alpha = (num_channels == 2 && c == 1) || (num_channels == 4 && c == 3);
        if (is_alpha) v = d.has_alpha ? d.alpha.row(y)[x] : 1.0f;
        else if (num_channels <= 2) v = d.color.size() == 1 ? d.color[0].row(y)[x] : d.color[1].row(y)[x];
        else v = d.color.size() == 1 ? d.color[0].row(y)[x] : d.color[c].row(y)[x];
        uint8_t* p = row + ((size_t)x * num_channels + c) * bps;
        if (type == 0) {
          float s = std::min(1.0f, std::max(0.0f, v)) * 255.0f;
          p[0] = (uint8_t)std::nearbyintf(s);
        } else if (type == 1) {
          float s = std::min(1.0f, std::max(0.0f, v)) * 65535.0f;
          uint16_t u = (uint16_t)std::nearbyintf(s);
          if (big_endian) { p[0] = u >> 8; p[1] = u & 255; } else { p[0] = u & 255; p[1] = u >> 8; }
        } else if (type == 2) {
          uint32_t u; memcpy(&u, &v, 4);
          if (big_endian) { p[0] = u >> 24; p[1] = u >> 16; p[2] = u >> 8; p[3] = u; } else memcpy(p, &u, 4);
        } else {
          uint16_t u = FloatToHalf(v);
          if (big_endian) { p[0] = u >> 8; p[1] = u & 255; } else { p[0] = u & 255; p[1] = u >> 8; }
        }
      }
    }
  }
  return total;
}

}  // namespace

// ---- C API for ctypes ----------------------------------------------------------------------------------------------
extern "C" {

struct jxlo_info {
  uint32_t xsize, ysize, bits_per_sample, exponent_bits, num_color_channels, num_extra_channels, alpha_bits, orientation;
  uint32_t have_container, xyb_encoded, has_jbrd, reserved;
  float intensity_target, min_nits;
  uint64_t tokens_lf, tokens_hf, tokens_modular;
  double seconds;
};

struct jxlo_handle {
  Decoded d;
  std::vector<uint8_t> pixels;
  std::string err;
};

jxlo_handle* jxlo_decode(const uint8_t* data, size_t size, int want_dump) {
  jxlo_handle* h = new jxlo_handle();
  try {
    auto t0 = std::chrono::steady_clock::now();
    DecodeImage(data, size, h->d, want_dump != 0);
    h->d.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  } catch (const std::exception& e) {
    h->err = e.what();
    if (h->err.empty()) h->err = "error";
  }
  return h;
}
const char* jxlo_error(jxlo_handle* h) { return h->err.empty() ? nullptr : h->err.c_str(); }
void jxlo_free(jxlo_handle* h) { delete h; }
void jxlo_get_info(jxlo_handle* h, jxlo_info* i) {
  const ImageMetadata& m = h->d.meta;
  memset(i, 0, sizeof(*i));
  i->xsize = m.xsize; i->ysize = m.ysize; i->bits_per_sample = m.depth.bits; i->exponent_bits = m.depth.exp_bits;
  i->num_color_channels = m.color.color_space == 1 ? 1 : 3;
  i->num_extra_channels = (uint32_t)m.extra.size();
  for (auto& e : m.extra) if (e.type == 0) { i->alpha_bits = e.depth.bits; break; }
  i->orientation = m.orientation;
  i->have_container = h->d.have_container; i->xyb_encoded = m.xyb_encoded; i->has_jbrd = h->d.has_jbrd;
  i->intensity_target = m.intensity_target; i->min_nits = m.min_nits;
  i->tokens_lf = h->d.tokens_lf; i->tokens_hf = h->d.tokens_hf; i->tokens_modular = h->d.tokens_modular;
  i->seconds = h->d.seconds;
}
// renders pixels; returns byte size (0 on error); pointer valid until next call / free
size_t jxlo_render(jxlo_handle* h, int type, int num_channels, int big_endian, size_t align, const uint8_t** out) {
  if (!h->err.empty()) return 0;
  size_t n = WritePixels(h->d, type, num_channels, big_endian, align, h->pixels);
  *out = h->pixels.data();
  return n;
}
int jxlo_get_plane(jxlo_handle* h, const char* name, const float** p, int* w, int* hh) {
  auto it = h->d.dump.planes.find(name);
  if (it == h->d.dump.planes.end()) return 0;
  *p = it->second.d.data(); *w = it->second.w; *hh = it->second.h;
  return 1;
}
int jxlo_get_ints(jxlo_handle* h, const char* name, const int32_t** p, size_t* n) {
  auto it = h->d.dump.ints.find(name);
  if (it == h->d.dump.ints.end()) return 0;
  *p = it->second.data(); *n = it->second.size();
  return 1;
}
// standalone stage entry points for kernel-level parity tests
void jxlo_idct(int strategy, const float* coeffs, float* out, int stride) { InverseTransform(strategy, coeffs, out, stride); }
void jxlo_natural_order(int strategy, uint32_t* out) { auto v = NaturalCoeffOrder(strategy); memcpy(out, v.data(), v.size() * 4); }
float jxlo_srgb(float v) { return LinearToSRGB(v); }
}
