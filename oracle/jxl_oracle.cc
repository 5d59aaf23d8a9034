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
  bool has_alpha = false, alpha_premultiplied = false;
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
  if (f.fh.do_ycbcr && !f.fh.modular) {
    // frame_header.h YCbCrChromaSubsampling: sampling-factor modes per channel -> shifts relative to the largest factor; the block
    // grid is padded to whole cells of the coarsest channel (frame_dimensions: xsize_blocks = DivCeil(xsize, 8 << maxhs) << maxhs)
    static const int kH[4] = {0, 1, 1, 0}, kV[4] = {0, 1, 0, 1};
    int maxhs = 0, maxvs = 0;
    for (int c = 0; c < 3; c++) { maxhs = std::max(maxhs, kH[f.fh.jpeg_upsampling[c]]); maxvs = std::max(maxvs, kV[f.fh.jpeg_upsampling[c]]); }
    for (int c = 0; c < 3; c++) { f.hs[c] = maxhs - kH[f.fh.jpeg_upsampling[c]]; f.vs[c] = maxvs - kV[f.fh.jpeg_upsampling[c]]; f.subsampled |= f.hs[c] || f.vs[c]; }
    f.bw = ((f.w + (8 << maxhs) - 1) / (8 << maxhs)) << maxhs; f.bh = ((f.h + (8 << maxvs) - 1) / (8 << maxvs)) << maxvs;
  }
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

// Progressive preview (decode.cc JxlDecoderFlushImage at the kDC step: the frame as it stands when the LF image and the HF metadata are there and no AC group has
// been decoded — every AC coefficient still zero): set for the decodes started afterwards on this thread; the PassGroup sections are not looked at (they may be cut off).
static thread_local bool g_dc_only = false;
// Later progression steps (kLastPasses / kPasses) and truncated input: at most g_max_passes passes of every group are decoded (-1: all), and with g_allow_truncated a
// PassGroup section that is not completely there — and every later pass of that group — is left out: dec_frame.cc Flush draws each group with the passes that have arrived.
static thread_local int g_max_passes = -1;
static thread_local bool g_allow_truncated = false;
void DecodeFrameSections(const uint8_t* data, size_t size, BitReader& br, Frame& f) {
  size_t n = f.fh.toc_entries();
  std::vector<Section> sec;
  ReadTOC(br, n, sec);
  if ((g_dc_only || g_allow_truncated) && n > 1 && !f.fh.modular) {
    if (sec[1 + f.fh.num_lf_groups].offset + sec[1 + f.fh.num_lf_groups].size > size) JXLO_FAIL("truncated frame (LF part incomplete)");
  } else if (sec.back().offset > size) JXLO_FAIL("truncated frame");
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
    std::vector<uint8_t> gone(f.fh.num_groups, 0);       // (truncated input) the group's stream of an earlier pass was not there
    for (uint32_t p = 0; p < f.fh.passes.num_passes && !(g_dc_only && !f.fh.modular); p++)
      for (uint32_t g = 0; g < f.fh.num_groups; g++) {
        if (!f.fh.modular && g_max_passes >= 0 && (int)p >= g_max_passes) continue;
        const size_t si = 2 + f.fh.num_lf_groups + p * f.fh.num_groups + g;
        if (g_allow_truncated && !f.fh.modular && (gone[g] || sec[si].offset + sec[si].size > size)) { gone[g] = 1; continue; }
        BitReader r = reader(si);
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

// Default 4x upsampling weights (image_metadata.cc kWeights4): recalled; the kernel they define is a partition of unity
// for every sub-pixel (tests/test_oracle_goldens.py).
static const float kDefaultUp4Weights[55] = {
    -0.02419067f, -0.03491987f, -0.03693351f, -0.03094285f, -0.00529785f, -0.01663432f, -0.03556863f, -0.03888905f, -0.03516850f, -0.00989469f, 0.23651958f,
    0.33392945f,  -0.01073543f, -0.01313181f, -0.03556694f, 0.13048175f,  0.40103025f,  0.03951150f,  -0.02077584f, 0.46914198f,  -0.00209270f, -0.01484589f,
    -0.04064806f, 0.18942530f,  0.56279892f,  0.06674400f,  -0.02335494f, -0.03551682f, -0.00754830f, -0.02267919f, -0.02363578f, 0.00315804f,  -0.03399098f,
    -0.01359519f, -0.00091653f, -0.00335467f, -0.01163294f, -0.01610294f, -0.00974088f, -0.00191622f, -0.01095446f, -0.03198464f, -0.04455121f, -0.02799790f,
    -0.00645912f, 0.06390599f,  0.22963888f,  0.00630981f,  -0.01897349f, 0.67537268f,  0.08483369f,  -0.02534994f, -0.02205197f, -0.01667999f, -0.00384443f};

// Default 8x upsampling weights (image_metadata.cc kWeights8): recalled; all 16 sub-pixel kernels sum to 1 to 7 digits.
static const float kDefaultUp8Weights[210] = {
    -0.02928613f, -0.03706353f, -0.03783812f, -0.03324558f, -0.00447632f, -0.02519406f, -0.03752601f, -0.03901508f, -0.03663285f, -0.00646649f,
    -0.02066407f, -0.03838633f, -0.04002101f, -0.03900035f, -0.00901973f, -0.01626393f, -0.03954148f, -0.04046620f, -0.03979621f, -0.01224485f,
    0.29895328f, 0.35757708f, -0.02447552f, -0.01081748f, -0.04314594f, 0.23903219f, 0.41119301f, -0.00573046f, -0.01450239f, -0.04246845f,
    0.17567618f, 0.45220643f, 0.02287757f, -0.01936783f, -0.03583255f, 0.11572472f, 0.47416733f, 0.06284440f, -0.02685066f, 0.42720050f,
    -0.02248939f, -0.01155273f, -0.04562755f, 0.28689496f, 0.49093869f, -0.00007891f, -0.01545926f, -0.04562659f, 0.21238920f, 0.53980934f,
    0.03369474f, -0.02070211f, -0.03866988f, 0.14229550f, 0.56593398f, 0.08045181f, -0.02888298f, -0.03680918f, -0.00542229f, -0.02920477f,
    -0.02788574f, -0.02118180f, -0.03942402f, -0.00775547f, -0.02433614f, -0.03193943f, -0.02030828f, -0.04044014f, -0.01074016f, -0.01930822f,
    -0.03620399f, -0.01974125f, -0.03919545f, -0.01456093f, -0.00045072f, -0.00360110f, -0.01020207f, -0.01231907f, -0.00638988f, -0.00071592f,
    -0.00279122f, -0.00957115f, -0.01288327f, -0.00730937f, -0.00107783f, -0.00210156f, -0.00890705f, -0.01317668f, -0.00813895f, -0.00153491f,
    -0.02128481f, -0.04173044f, -0.04831487f, -0.03293190f, -0.00525260f, -0.01720322f, -0.04052736f, -0.05045706f, -0.03607317f, -0.00738030f,
    -0.01341764f, -0.03965629f, -0.05151616f, -0.03814886f, -0.01005819f, 0.18968273f, 0.33063684f, -0.01300105f, -0.01372950f, -0.04017465f,
    0.13727832f, 0.36402234f, 0.01027890f, -0.01832107f, -0.03365072f, 0.08734506f, 0.38194295f, 0.04338228f, -0.02525993f, 0.56408126f,
    0.00458352f, -0.01648227f, -0.04887868f, 0.24585519f, 0.62026135f, 0.04314807f, -0.02213737f, -0.04158014f, 0.16637289f, 0.65027023f,
    0.09621636f, -0.03101388f, -0.04082742f, -0.00904519f, -0.02790922f, -0.02117818f, 0.00798662f, -0.03995711f, -0.01243427f, -0.02231705f,
    -0.02946266f, 0.00992055f, -0.03600283f, -0.01684920f, -0.00111684f, -0.00411204f, -0.01297130f, -0.01723725f, -0.01022545f, -0.00165306f,
    -0.00313110f, -0.01218016f, -0.01763266f, -0.01125620f, -0.00231663f, -0.01374149f, -0.03797620f, -0.05142937f, -0.03117307f, -0.00581914f,
    -0.01064003f, -0.03608089f, -0.05272168f, -0.03375670f, -0.00795586f, 0.09628104f, 0.27129991f, -0.00353779f, -0.01734151f, -0.03153981f,
    0.05686230f, 0.28500998f, 0.02230594f, -0.02374955f, 0.68214326f, 0.05018048f, -0.02320852f, -0.04383616f, 0.18459474f, 0.71517975f,
    0.10805613f, -0.03263677f, -0.03637639f, -0.01394373f, -0.02511203f, -0.01728636f, 0.05407331f, -0.02867568f, -0.01893131f, -0.00240854f,
    -0.00446511f, -0.01636187f, -0.02377053f, -0.01522848f, -0.00333334f, -0.00819975f, -0.02964169f, -0.04499287f, -0.02745350f, -0.00612408f,
    0.02727416f, 0.19446600f, 0.00159832f, -0.02232473f, 0.74982506f, 0.11452620f, -0.03348048f, -0.01605681f, -0.02070339f, -0.00458223f,
};

static thread_local bool g_render_spot = true;   // JxlDecoderSetRenderSpotcolors (default on)

// stage_spot.cc: every spot-colour extra channel in turn, p = mix * spot + (1 - mix) * p with mix = solidity * channel value
static void SpotMix(const ImageMetadata& m, const std::vector<Plane>& extra, int x, int y, float* r, float* g, float* b) {
  for (size_t e = 0; e < extra.size() && e < m.extra.size(); e++) {
    if (m.extra[e].type != 2) continue;
    const float mix = m.extra[e].spot[3] * extra[e].row(y)[x];
    *r = mix * m.extra[e].spot[0] + (1.0f - mix) * *r;
    *g = mix * m.extra[e].spot[1] + (1.0f - mix) * *g;
    *b = mix * m.extra[e].spot[2] + (1.0f - mix) * *b;
  }
}

// dec_modular.cc int_to_float: a float sample's bit pattern (sign, exp_bits of exponent, the rest mantissa) -> binary32
static float IntToFloatSample(int32_t in, uint32_t bits, uint32_t exp_bits) {
  uint32_t f = (uint32_t)in;
  float out;
  if (bits == 32) { memcpy(&out, &f, 4); return out; }
  const int exp_bias = (1 << (exp_bits - 1)) - 1;
  const int sign_shift = (int)bits - 1, mant_bits = (int)bits - (int)exp_bits - 1, mant_shift = 23 - mant_bits;
  const int signbit = (int)((f >> sign_shift) & 1u);
  f &= (1u << sign_shift) - 1;
  if (f == 0) return signbit ? -0.f : 0.f;
  int exp = (int)(f >> mant_bits);
  int mantissa = (int)(f & ((1u << mant_bits) - 1));
  mantissa <<= mant_shift;
  if (exp == 0 && exp_bits < 8) {          // subnormal number: normalise, then drop the leading 1 (implicit from now on)
    while ((mantissa & 0x800000) == 0) { mantissa <<= 1; exp--; }
    exp++;
    mantissa &= 0x7fffff;
  }
  exp -= exp_bias;
  exp += 127;
  JXLO_CHECK(exp >= 0);
  f = (signbit ? 0x80000000u : 0u) | ((uint32_t)exp << 23) | (uint32_t)mantissa;
  memcpy(&out, &f, 4);
  return out;
}

// dec_modular.cc ModularImageToDecodedRect: integer channels of the frame's Modular image -> float planes
void ModularToFloat(const Frame& f, const ImageMetadata& m, Image3& img) {
  const bool gray = m.color.color_space == 1;
  const int w = f.w, h = f.h;
  for (int c = 0; c < 3; c++) img.p[c] = Plane(w, h);
  if (m.depth.float_sample && !m.xyb_encoded) {
    const uint32_t b = m.depth.bits, eb = m.depth.exp_bits;
    if (b > 32 || eb < 2 || eb > 8 || b < eb + 2 || b - eb - 1 > 23 || (b == 32 && eb != 8)) JXLO_FAIL("unsupported: float sample layout");
  }
  if (m.xyb_encoded) {
    // XYB is coded as Y, X, B - Y and scaled by the LF dequantisation factors (DequantMatrices::DCQuants)
    if (f.gimg.channel.size() < 3) JXLO_FAIL("missing colour channels");
    const Channel &cy = f.gimg.channel[0], &cx = f.gimg.channel[1], &cb = f.gimg.channel[2];
    JXLO_CHECK(cy.w == w && cy.h == h && cx.w == w && cx.h == h && cb.w == w && cb.h == h);
    for (size_t i = 0; i < (size_t)w * h; i++) {
      img.p[0].d[i] = (float)cx.data[i] * f.m_lf[0];
      img.p[1].d[i] = (float)cy.data[i] * f.m_lf[1];
      img.p[2].d[i] = (float)(cb.data[i] + cy.data[i]) * f.m_lf[2];
    }
    return;
  }
  const int nb = (gray && !f.fh.do_ycbcr) ? 1 : 3;
  if ((int)f.gimg.channel.size() < nb) JXLO_FAIL("missing colour channels");
  const bool fl = m.depth.float_sample;
  const float factor = fl ? 1.0f : (float)(1.0 / (double)((1u << m.depth.bits) - 1));
  for (int c = 0; c < 3; c++) {
    const Channel& ch = f.gimg.channel[nb == 1 ? 0 : c];
    JXLO_CHECK(ch.w == w && ch.h == h);
    if (fl) for (size_t i = 0; i < (size_t)w * h; i++) img.p[c].d[i] = IntToFloatSample(ch.data[i], m.depth.bits, m.depth.exp_bits);
    else for (size_t i = 0; i < (size_t)w * h; i++) img.p[c].d[i] = (float)ch.data[i] * factor;
  }
}

void DecodeImage(const uint8_t* data, size_t size, Decoded& out, bool want_dump) {
  std::vector<uint8_t> cs = ExtractCodestream(data, size, &out.have_container, &out.has_jbrd);
  BitReader br(cs.data(), cs.size());
  ImageMetadata& m = out.meta;
  ReadImageHeaders(br, m);
  out.w = (int)m.xsize; out.h = (int)m.ysize;
  const bool gray = m.color.color_space == 1;
  out.num_color = gray ? 1 : 3;
  const size_t num_extra = m.extra.size();
  std::vector<bool> ec_premul(num_extra, false);
  for (size_t e = 0; e < num_extra; e++) ec_premul[e] = m.extra[e].alpha_associated;
  RefFrame refs[4];
  struct LfFrame { bool valid = false; Image3 img; } lf_frames[4];     // dec_cache.h PassesSharedState::dc_frames
  uint32_t visible_frame_index = 0, nonvisible_frame_index = 0;
  if (m.have_preview) {
    // decode.cc: the preview is a frame of its own in front of the image's frames (frame_header.cc: its default size is the PreviewHeader's).  Only a caller
    // that subscribes to JXL_DEC_PREVIEW_IMAGE gets it decoded (jpegxl-rs never does, decode.rs:334-347): header and TOC are read to find where it ends.
    ImageMetadata pm = m;
    pm.xsize = m.preview_x; pm.ysize = m.preview_y;
    FrameHeader ph;
    ReadFrameHeader(br, pm, ph);
    if (ph.type != kRegular) JXLO_FAIL("the preview must be a regular frame");
    std::vector<Section> sec;
    ReadTOC(br, ph.toc_entries(), sec);
    if (sec.back().offset > cs.size()) JXLO_FAIL("truncated preview frame");
    br.pos = sec.back().offset * 8;
  }
  for (;;) {
    Frame f;
    ReadFrameHeader(br, m, f.fh);
    const FrameHeader& fh = f.fh;
    if (fh.type == kLFFrame && (fh.upsampling != 1 || (fh.flags & (kPatches | kSplines | kNoise)))) JXLO_FAIL("LF frame with upsampling / image features");
    if (fh.type == kRegular || fh.type == kSkipProgressive) { visible_frame_index++; nonvisible_frame_index = 0; } else nonvisible_frame_index++;
    const int up = (int)fh.upsampling;
    const float* up_weights = nullptr;
    if (up > 1) {
      const std::vector<float>& cw = up == 2 ? m.up2 : up == 4 ? m.up4 : m.up8;
      if (!cw.empty()) up_weights = cw.data();
      else if (up == 2) up_weights = kDefaultUp2Weights;
      else if (up == 4) up_weights = kDefaultUp4Weights;
      else up_weights = kDefaultUp8Weights;
    }
    InitFrame(f, m);
    if (f.subsampled) {
      if (!(fh.flags & kSkipAdaptiveLFSmoothing) || fh.lf.gab || fh.lf.epf_iters || fh.upsampling != 1 || fh.passes.num_passes != 1)
        JXLO_FAIL("unsupported: chroma subsampling together with LF smoothing / restoration filters / upsampling / passes");
    }
    if (fh.flags & kUseLfFrame) {
      // dec_cache.cc InitializePassesSharedState: the LF image is the one the LF frame of the next level left (dc_frames[lf_level]; a regular frame is level 0),
      // no LF coefficients in the LfGroups, no dequantisation, no adaptive smoothing (dec_frame.cc FinalizeDC)
      if (fh.modular) JXLO_FAIL("use_lf_frame on a Modular frame");
      if (fh.lf_level >= 4 || !lf_frames[fh.lf_level].valid) JXLO_FAIL("the LF frame this frame refers to has not been decoded");
      const Image3& src = lf_frames[fh.lf_level].img;
      if (src.w() != f.bw || src.h() != f.bh || f.subsampled) JXLO_FAIL("LF frame of the wrong size");
      f.lf = src;
    }
    f.dump = want_dump ? &out.dump : nullptr;
    DecodeFrameSections(cs.data(), cs.size(), br, f);
    out.tokens_lf += f.tokens_lf; out.tokens_hf += f.tokens_hf; out.tokens_modular += f.tokens_modular;
    // undo global modular transforms
    if (!f.gimg.channel.empty()) UndoTransforms(f.gimg, f.gimg_header.wp);
    const int cw_ = f.w, ch_ = f.h;                          // coded size
    const int fw = (int)fh.xsize, fhh = (int)fh.ysize;       // frame size after upsampling
    Image3 img;
    size_t first_extra = 0;
    std::vector<float> inv_sigma;
    if (!fh.modular) {
      // ---- VarDCT ----
      if (!(fh.flags & kSkipAdaptiveLFSmoothing) && !(fh.flags & kUseLfFrame)) {
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
          v.clear();
          for (auto& g : f.coeffs[c]) v.insert(v.end(), g.begin(), g.end());
        }
        for (int c = 0; c < 3; c++) out.dump.ints[std::string("lfq") + char('0' + c)] = f.lfq[c];
        std::vector<int32_t>& st = out.dump.ints["strategy"]; st.clear();
        std::vector<int32_t>& hm = out.dump.ints["hf_mul"]; hm.clear();
        std::vector<int32_t>& sh = out.dump.ints["sharpness"]; sh.clear();
        for (size_t i = 0; i < f.strategy.size(); i++) { st.push_back(f.is_first[i] ? f.strategy[i] : -1 - f.strategy[i]); hm.push_back(f.hf_mul[i]); sh.push_back(f.sharpness[i]); }
        std::vector<int32_t>& cf = out.dump.ints["cfl"]; cf.clear();
        for (size_t i = 0; i < f.ytox_map.size(); i++) { cf.push_back(f.ytox_map[i]); cf.push_back(f.ytob_map[i]); }
      }
      DequantAndIDCT(f);
      if (f.subsampled) UpsampleChroma(f);
      if (want_dump) StorePlanes(out.dump, "idct", f.xyb);
      img = CropImage(f.xyb, cw_, ch_);
      if (fh.lf.epf_iters > 0) ComputeInvSigma(fh.lf, (float)f.global_scale / 65536.0f, f.hf_mul, f.sharpness, f.bw, f.bh, inv_sigma);
      first_extra = 0;
    } else {
      // ---- Modular ----
      ModularToFloat(f, m, img);
      if (want_dump) {
        const int nb = (gray && !m.xyb_encoded) ? 1 : 3;
        for (int c = 0; c < nb; c++) out.dump.ints[std::string("modular") + char('0' + c)].assign(f.gimg.channel[c].data.begin(), f.gimg.channel[c].data.end());
      }
      first_extra = (gray && !m.xyb_encoded && !fh.do_ycbcr) ? 1 : 3;
      if (fh.lf.epf_iters > 0) inv_sigma.assign((size_t)f.bw * f.bh, kInvSigmaNum / fh.lf.sigma_for_modular);   // epf.cc: constant sigma image
    }
    // loop filters (both encodings: stage_gaborish.cc, stage_epf.cc)
    if (fh.lf.gab) Gaborish(fh.lf, img);
    if (want_dump) StorePlanes(out.dump, "gab", img);
    if (fh.lf.epf_iters > 0) {
      if (fh.lf.epf_iters >= 3) EPFPass(fh.lf, 0, inv_sigma, f.bw, img);
      EPFPass(fh.lf, 1, inv_sigma, f.bw, img);
      if (fh.lf.epf_iters >= 2) EPFPass(fh.lf, 2, inv_sigma, f.bw, img);
    }
    if (want_dump) StorePlanes(out.dump, "epf", img);
    // extra channels as float planes (coded size)
    std::vector<Plane> extra(num_extra);
    for (size_t e = 0; e < num_extra; e++) {
      if (m.extra[e].dim_shift != 0) JXLO_FAIL("unsupported: subsampled extra channel");
      const bool efl = m.extra[e].depth.float_sample;
      if (efl) {
        const uint32_t b = m.extra[e].depth.bits, eb = m.extra[e].depth.exp_bits;
        if (b > 32 || eb < 2 || eb > 8 || b < eb + 2 || b - eb - 1 > 23 || (b == 32 && eb != 8)) JXLO_FAIL("unsupported: float sample layout");
      }
      if (first_extra + e >= f.gimg.channel.size()) JXLO_FAIL("missing extra channel");
      const Channel& ch = f.gimg.channel[first_extra + e];
      JXLO_CHECK(ch.w == cw_ && ch.h == ch_);
      const float factor = efl ? 1.0f : 1.0f / (float)((1u << m.extra[e].depth.bits) - 1);
      extra[e] = Plane(cw_, ch_);
      if (efl) for (size_t i = 0; i < (size_t)cw_ * ch_; i++) extra[e].d[i] = IntToFloatSample(ch.data[i], m.extra[e].depth.bits, m.extra[e].depth.exp_bits);
      else for (size_t i = 0; i < (size_t)cw_ * ch_; i++) extra[e].d[i] = (float)ch.data[i] * factor;
      if (want_dump && m.extra[e].type == 0 && !out.dump.ints.count("alpha")) out.dump.ints["alpha"].assign(ch.data.begin(), ch.data.end());
    }
    // image features (dec_cache.cc PreparePipeline order): patches, splines, upsampling, noise
    const float y_to_x = f.base_x, y_to_b = f.base_b;     // ColorCorrelationMap::YtoXRatio(0) / YtoBRatio(0)
    if (fh.flags & kPatches) ApplyPatches(f.patches, refs, true, ec_premul, img, extra);
    if (want_dump && (fh.flags & kPatches)) StorePlanes(out.dump, "patches", img);
    if (fh.flags & kSplines) {
      BuildSplineSegments(f.splines, y_to_x, y_to_b);
      DrawSplines(f.splines, img);
      if (want_dump) StorePlanes(out.dump, "splines", img);
    }
    if (up > 1) {   // stage_upsampling.cc: before the colour transform
      for (int c = 0; c < 3; c++) img.p[c] = UpsamplePlane(img.p[c], up, up_weights, fw, fhh);
      for (size_t e = 0; e < num_extra; e++) extra[e] = UpsamplePlane(extra[e], up, up_weights, fw, fhh);   // ec_upsampling == upsampling (checked at parse)
      if (want_dump) StorePlanes(out.dump, "ups", img);
    }
    if (fh.flags & kNoise) {
      if (!m.xyb_encoded) JXLO_FAIL("noise on a non-XYB frame");
      AddNoise(f.noise, visible_frame_index, nonvisible_frame_index, (int)fh.group_dim, y_to_x, y_to_b, img);
      if (want_dump) StorePlanes(out.dump, "noise", img);
    }
    if (fh.type == kLFFrame) {
      // dec_frame.cc: an LF frame ends here — its samples (XYB, or whatever the image is coded in) are the LF image of the frames one level down
      if (fh.lf_level < 1 || fh.lf_level > 4) JXLO_FAIL("LF level");
      lf_frames[fh.lf_level - 1].valid = true;
      lf_frames[fh.lf_level - 1].img = img;
      continue;
    }
    JXLO_CHECK(img.w() == fw && img.h() == fhh);
    const bool can_ref = !fh.is_last && fh.type != kLFFrame && (fh.duration == 0 || fh.save_as_reference != 0);
    if (can_ref && fh.save_before_ct) {
      RefFrame& r = refs[fh.save_as_reference];
      r.valid = true; r.is_xyb = true; r.w = fw; r.h = fhh; r.color = img; r.extra = extra;
    }
    if (fh.type == kReferenceOnly) continue;
    // ---- colour transform to the output space (stage_xyb.cc, stage_from_linear.cc, stage_ycbcr.cc)
    // spot colours (dec_cache.cc PreparePipeline: XYB stage, [from-linear + blending], spot stage, from-linear): mixed in linear light
    // when an XYB frame goes straight to the output, in the output space otherwise; only what is handed out gets them
    bool has_spot = false;
    for (size_t e = 0; e < num_extra; e++) if (m.extra[e].type == 2) has_spot = true;
    has_spot = has_spot && g_render_spot && fh.is_last;
    if (has_spot && gray) JXLO_FAIL("unsupported: spot colours on a grey image");
    bool frame_blends = fh.have_crop || fh.blend.mode != 0;
    for (auto& b : fh.ec_blend) if (b.mode != 0) frame_blends = true;
    const bool spot_in_linear = has_spot && m.xyb_encoded && !frame_blends;
    Image3 rgb = img;
    if (m.xyb_encoded) {
      float luminances[3];
      OpsinParams op = MakeOpsin(m, m.intensity_target, luminances);
      // stage_from_linear.cc: 0 sRGB, 1 linear, 2 gamma (OpGamma: FastPowf, zero below 1e-5), 3 Rec.709, 4 PQ, 5 HLG (inverse OOTF first)
      int tf_kind = 0; float inverse_gamma = 1.0f;
      HlgOotf ootf;
      if (!m.color.all_default) {
        if (m.color.have_gamma) { tf_kind = 2; inverse_gamma = (float)m.color.gamma * 1e-7f; }
        else if (m.color.tf == 13) tf_kind = 0;
        else if (m.color.tf == 8) tf_kind = 1;
        else if (m.color.tf == 17) { tf_kind = 2; inverse_gamma = 1.0f / 2.6f; }   // DCI
        else if (m.color.tf == 1) tf_kind = 3;
        else if (m.color.tf == 16) tf_kind = 4;
        else if (m.color.tf == 18) { tf_kind = 5; ootf = HlgOotf(m.intensity_target, luminances); }
        else JXLO_FAIL("unsupported: output transfer function");
      }
      const float pq_scale = m.intensity_target * (1.0f / 10000.0f);
      auto tf = [&](float v) -> float {
        switch (tf_kind) {
          case 0: return LinearToSRGB(v);
          case 1: return v;
          case 2: return v <= 1e-5f ? 0.0f : FastPowf(v, inverse_gamma);
          case 4: return PqFromLinear(v, pq_scale);
          case 5: return HlgFromLinear(v);
          default: return v <= 0.018f ? 4.5f * v : std::fmaf(1.099f, FastPowf(v, 0.45f), -0.099f);
        }
      };
      for (int y = 0; y < fhh; y++) for (int x = 0; x < fw; x++) {
        float r, g, b;
        XybToLinear(op, img.p[0].row(y)[x], img.p[1].row(y)[x], img.p[2].row(y)[x], &r, &g, &b);
        if (spot_in_linear) SpotMix(m, extra, x, y, &r, &g, &b);       // stage_spot.cc between the XYB stage and the transfer function
        if (tf_kind == 5) ootf.Apply(&r, &g, &b);
        r = tf(r); g = tf(g); b = tf(b);
        rgb.p[0].row(y)[x] = r; rgb.p[1].row(y)[x] = g; rgb.p[2].row(y)[x] = b;
      }
    } else if (fh.do_ycbcr) {
      // stage_ycbcr.cc: planes are Cb, Y, Cr
      for (int y = 0; y < fhh; y++) for (int x = 0; x < fw; x++) {
        float cb = img.p[0].row(y)[x], yy = img.p[1].row(y)[x], cr = img.p[2].row(y)[x];
        const float c128 = 128.0f / 255, crcr = 1.402f, cgcb = -0.114f * 1.772f / 0.587f, cgcr = -0.299f * 1.402f / 0.587f, cbcb = 1.772f;
        float yb = yy + c128;
        rgb.p[0].row(y)[x] = std::fmaf(crcr, cr, yb);
        rgb.p[1].row(y)[x] = std::fmaf(cgcr, cr, std::fmaf(cgcb, cb, yb));
        rgb.p[2].row(y)[x] = std::fmaf(cbcb, cb, yb);
      }
    }
    // ---- blending onto the canvas (stage_blending.cc; blending.cc PerformBlending)
    bool replace_all = fh.blend.mode == 0;
    for (auto& b : fh.ec_blend) if (b.mode != 0) replace_all = false;
    const bool needs_blending = fh.have_crop || !replace_all;
    Image3 canvas; std::vector<Plane> canvas_extra(num_extra);
    if (!needs_blending) {
      JXLO_CHECK(fw == out.w && fhh == out.h);
      canvas = rgb; canvas_extra = extra;
    } else {
      auto bg_of = [&](uint32_t source) -> const RefFrame* {
        const RefFrame& r = refs[source];
        if (!r.valid) return nullptr;
        if (r.is_xyb) JXLO_FAIL("blending source was saved before the colour transform");
        if (r.w != out.w || r.h != out.h) JXLO_FAIL("blending source has the wrong size");
        return &r;
      };
      const RefFrame* bg = bg_of(fh.blend.source);
      for (int c = 0; c < 3; c++) canvas.p[c] = Plane(out.w, out.h);
      for (size_t e = 0; e < num_extra; e++) canvas_extra[e] = Plane(out.w, out.h);
      std::vector<const RefFrame*> ebg(num_extra);
      for (size_t e = 0; e < num_extra; e++) ebg[e] = bg_of(fh.ec_blend[e].source);
      const uint32_t mode = fh.blend.mode;
      const bool uses_alpha = mode == 2 || mode == 3;
      if (uses_alpha && fh.blend.alpha_channel >= num_extra && num_extra > 0) JXLO_FAIL("bad blend alpha channel");
      if (uses_alpha && num_extra == 0) JXLO_FAIL("alpha blending without extra channels");
      for (int Y = 0; Y < out.h; Y++) for (int X = 0; X < out.w; X++) {
        const int fx = X - fh.x0, fy = Y - fh.y0;
        const bool inside = fx >= 0 && fy >= 0 && fx < fw && fy < fhh;
        if (!inside) {
          for (int c = 0; c < 3; c++) canvas.p[c].row(Y)[X] = bg ? bg->color.p[c].row(Y)[X] : 0.0f;
          for (size_t e = 0; e < num_extra; e++) canvas_extra[e].row(Y)[X] = ebg[e] ? ebg[e]->extra[e].row(Y)[X] : 0.0f;
          continue;
        }
        float fga = 1.0f, bga = 1.0f; bool premul = false;
        if (uses_alpha) {
          const uint32_t a = fh.blend.alpha_channel;
          fga = extra[a].row(fy)[fx]; bga = bg ? bg->extra[a].row(Y)[X] : 0.0f; premul = ec_premul[a];
        }
        for (int c = 0; c < 3; c++) {
          const float b = bg ? bg->color.p[c].row(Y)[X] : 0.0f;
          canvas.p[c].row(Y)[X] = FrameBlendSample(mode, fh.blend.clamp, premul, b, rgb.p[c].row(fy)[fx], bga, fga);
        }
        for (size_t e = 0; e < num_extra; e++) {
          const BlendInfo& bi = fh.ec_blend[e];
          const float b = ebg[e] ? ebg[e]->extra[e].row(Y)[X] : 0.0f;
          const float fv = extra[e].row(fy)[fx];
          float o;
          if (bi.mode == 2 || bi.mode == 3) {
            const uint32_t a = bi.alpha_channel;
            const float efga = extra[a].row(fy)[fx], ebga = ebg[e] ? ebg[e]->extra[a].row(Y)[X] : 0.0f;
            if (a == e) { const float fa = bi.clamp ? Clamp01(efga) : efga; o = bi.mode == 2 ? 1.0f - (1.0f - fa) * (1.0f - ebga) : ebga; }
            else o = FrameBlendSample(bi.mode, bi.clamp, ec_premul[a], b, fv, ebga, efga);
          } else o = FrameBlendSample(bi.mode, bi.clamp, false, b, fv, 1.0f, 1.0f);
          canvas_extra[e].row(Y)[X] = o;
        }
      }
    }
    if (can_ref && !fh.save_before_ct) {
      RefFrame& r = refs[fh.save_as_reference];
      r.valid = true; r.is_xyb = false; r.w = out.w; r.h = out.h; r.color = canvas; r.extra = canvas_extra;
    }
    if (!fh.is_last) continue;   // coalescing: only the composite of the last frame is handed out (animation frames: last wins)
    if (has_spot && !spot_in_linear)
      for (int y = 0; y < out.h; y++) for (int x = 0; x < out.w; x++) SpotMix(m, canvas_extra, x, y, &canvas.p[0].row(y)[x], &canvas.p[1].row(y)[x], &canvas.p[2].row(y)[x]);
    out.color.clear();
    if (gray) out.color.push_back(canvas.p[0]);
    else for (int c = 0; c < 3; c++) out.color.push_back(canvas.p[c]);
    out.has_alpha = false;
    for (size_t e = 0; e < num_extra; e++) {
      if (m.extra[e].type != 0) continue;
      out.alpha = canvas_extra[e];
      out.has_alpha = true;
      out.alpha_premultiplied = m.extra[e].alpha_associated;
      break;
    }
    break;
  }
}

// stage_write.cc: clamp, scale, round-to-nearest-even, interleave (SURVEY b17 [V])
size_t WritePixels(const Decoded& d, int type /*0 u8,1 u16,2 f32,3 f16*/, int num_channels, int big_endian, size_t align, std::vector<uint8_t>& out, bool unpremul = false) {
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
        // alpha.cc UnpremultiplyAlpha (stage_write.cc, JxlDecoderSetUnpremultiplyAlpha): only when alpha is associated and written out
        if (!is_alpha && unpremul && d.has_alpha && d.alpha_premultiplied && (num_channels == 2 || num_channels == 4))
          v *= 1.0f / std::max(1.0f / (float)(1u << 26), d.alpha.row(y)[x]);
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
  bool unpremul = false;
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
void jxlo_set_unpremultiply_alpha(jxlo_handle* h, int v) { h->unpremul = v != 0; }
void jxlo_set_render_spotcolors(int v) { g_render_spot = v != 0; }   // applies to the decodes started afterwards on this thread
void jxlo_set_progress(int max_passes, int allow_truncated) { g_max_passes = max_passes; g_allow_truncated = allow_truncated != 0; }   // later progression steps / input cut off inside the AC groups
void jxlo_set_dc_only(int v) { g_dc_only = v != 0; }                  // likewise: JxlDecoderFlushImage at the kDC step (no AC group decoded)
// embedded ICC profile of the image (empty when the colour encoding is enumerated)
size_t jxlo_icc(jxlo_handle* h, uint8_t* out, size_t cap) { const auto& v = h->d.meta.icc; if (out && cap >= v.size() && !v.empty()) memcpy(out, v.data(), v.size()); return v.size(); }
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
  size_t n = WritePixels(h->d, type, num_channels, big_endian, align, h->pixels, h->unpremul);
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
// recalled tables / approximations exposed to their self-consistency tests
int jxlo_table(const char* name, const float** p, size_t* n) {
  std::string s(name);
  if (s == "afv_basis") { *p = &k4x4AFVBasis[0][0]; *n = 256; return 1; }
  if (s == "up2") { *p = kDefaultUp2Weights; *n = 15; return 1; }
  if (s == "up4") { *p = kDefaultUp4Weights; *n = 55; return 1; }
  if (s == "up8") { *p = kDefaultUp8Weights; *n = 210; return 1; }
  return 0;
}
float jxlo_fastmath(int kind, float x, float y) {
  switch (kind) { case 0: return FastLog2f(x); case 1: return FastPow2f(x); case 2: return FastPowf(x, y); case 3: return FastErff(x); default: return FastCosf(x); }
}
// dequantisation table (1 / weight) of a library-default quant kind, channel c; returns the number of entries
size_t jxlo_library_qtable(int kind, int c, float* out, size_t cap) {
  std::vector<float> t;
  ComputeQuantTable(QuantEncoding(), kind, c, t);
  for (size_t i = 0; i < t.size() && i < cap; i++) out[i] = t[i];
  return t.size();
}
}
