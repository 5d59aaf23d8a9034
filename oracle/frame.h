// ORACLE — TEST INFRASTRUCTURE ONLY (see bits.h).
// Frame decoder: restates libjxl v0.11.2 lib/jxl/{dec_frame.cc,dec_group.cc,dec_modular.cc,dec_cache.cc,
// compressed_dc.cc,chroma_from_luma.cc,quantizer.cc,ac_context.h,coeff_order.cc}.  SURVEY.md App. B.3-B.6.
// Entry points the reference reaches this through: JxlDecoderProcessInput (jpegxl-rs/src/decode.rs:238).
#pragma once
#include "headers.h"
#include "modular.h"
#include "vardct.h"
#include "render.h"
#include "image_features.h"
#include <map>

namespace jxlo {

struct BlockCtxMap {
  std::vector<int32_t> lf_thresholds[3];
  std::vector<uint32_t> qf_thresholds;
  std::vector<uint8_t> ctx_map;
  int num_ctxs = 15, num_lf_ctxs = 1;
  static const uint8_t* DefaultMap() {
    static const uint8_t m[39] = {0, 1, 2, 2, 3, 3, 4, 5, 6, 6, 6, 6, 6, 7, 8, 9, 9, 10, 11, 12, 13, 14, 14, 14, 14, 14,
                                  7, 8, 9, 9, 10, 11, 12, 13, 14, 14, 14, 14, 14};
    return m;
  }
  void SetDefault() {
    for (auto& t : lf_thresholds) t.clear();
    qf_thresholds.clear();
    ctx_map.assign(DefaultMap(), DefaultMap() + 39);
    num_ctxs = 15; num_lf_ctxs = 1;
  }
  // ac_context.h BlockCtxMap::Context — c is the XYB channel index (0=X,1=Y,2=B)
  int Context(int lf_idx, uint32_t qf, int ord, int c) const {
    size_t qf_idx = 0;
    for (uint32_t t : qf_thresholds) if (qf > t) qf_idx++;
    size_t idx = c < 2 ? (c ^ 1) : 2;
    idx = idx * 13 + ord;
    idx = idx * (qf_thresholds.size() + 1) + qf_idx;
    idx = idx * num_lf_ctxs + lf_idx;
    return ctx_map[idx];
  }
};

struct Dump {  // intermediate results exposed to the parity tests
  std::map<std::string, Plane> planes;
  std::map<std::string, std::vector<int32_t>> ints;
};

struct PassInfo {
  std::vector<std::vector<uint32_t>> order;  // [bucket*3 + c]
  EntropyCode code;
};

struct Frame {
  const ImageMetadata* m = nullptr;
  FrameHeader fh;
  // geometry
  int w = 0, h = 0, bw = 0, bh = 0;  // pixels, 8x8 blocks
  // chroma subsampling (frame_header.h YCbCrChromaSubsampling): channel c lives on a grid of (bw >> hs[c]) x (bh >> vs[c]) blocks, kept
  // in the top-left corner of the full-size LF / coefficient / pixel arrays
  int hs[3] = {0, 0, 0}, vs[3] = {0, 0, 0};
  bool subsampled = false;
  // LfGlobal
  float m_lf[3] = {1.0f / 4096, 1.0f / 512, 1.0f / 256};
  uint32_t global_scale = 1, quant_lf = 1;
  BlockCtxMap bcm;
  uint32_t color_factor = 84; float base_x = 0.f, base_b = 1.0f; int32_t ytox_lf = 0, ytob_lf = 0;
  GlobalTree gtree;
  ModularImage gimg;  // full-frame modular image
  size_t gimg_global_decoded = 0;
  GroupHeader gimg_header;
  // LF / HF metadata (frame-wide maps)
  Image3 lf;                         // dequantised LF in X,Y,B
  std::vector<int32_t> lfq[3];       // quantised LF (X,Y,B) for context thresholds
  std::vector<uint8_t> strategy;     // per 8x8 block: strategy of covering varblock
  std::vector<uint8_t> is_first;     // top-left block of a varblock
  std::vector<int32_t> hf_mul;       // per block
  std::vector<uint8_t> sharpness;    // per block
  std::vector<int8_t> ytox_map, ytob_map; int cw = 0, chh = 0;  // per 64x64 tile
  // HfGlobal
  QuantEncoding qenc[17];
  std::vector<float> qtable[17][3];
  uint32_t num_hf_presets = 1;
  std::vector<PassInfo> pass;
  // coefficients per group
  std::vector<std::vector<int32_t>> coeffs[3];  // [c][group] -> 65536 ints
  // output
  Image3 xyb;     // float planes (XYB, or RGB / YCbCr for non-XYB VarDCT)
  size_t tokens_lf = 0, tokens_hf = 0, tokens_modular = 0;
  Dump* dump = nullptr;
  // image features (LfGlobal)
  PatchDictionary patches;
  Splines splines;
  NoiseParams noise;
};

inline float InvGlobalScale(const Frame& f) { return 65536.0f / (float)f.global_scale; }

// ---- LfGlobal ------------------------------------------------------------------------------------------------------
inline void ReadBlockCtxMap(BitReader& br, BlockCtxMap& b) {
  b.SetDefault();
  if (br.Bool()) return;
  for (int j = 0; j < 3; j++) {
    uint32_t n = br.u(4);
    b.lf_thresholds[j].resize(n);
    for (auto& t : b.lf_thresholds[j]) t = UnpackSigned(U32(br, Bits(4), BitsOffset(8, 16), BitsOffset(16, 272), BitsOffset(32, 65808)));
  }
  uint32_t nq = br.u(4);
  b.qf_thresholds.resize(nq);
  for (auto& t : b.qf_thresholds) t = U32(br, Bits(2), BitsOffset(3, 4), BitsOffset(5, 12), BitsOffset(8, 44)) + 1;
  b.num_lf_ctxs = (int)((b.lf_thresholds[0].size() + 1) * (b.lf_thresholds[1].size() + 1) * (b.lf_thresholds[2].size() + 1));
  size_t n = 3 * 13 * (size_t)b.num_lf_ctxs * (nq + 1);
  if (n > 39 * 64) JXLO_FAIL("block context map too large");
  int nc = 0;
  ReadContextMap(br, (int)n, b.ctx_map, nc);
  if (nc > 16) JXLO_FAIL("too many block contexts");
  b.num_ctxs = nc;
}

// dec_modular.cc DecodeGlobalInfo
inline void ReadGlobalModular(BitReader& br, Frame& f) {
  const ImageMetadata& m = *f.m;
  bool has_tree = br.Bool();
  int nb_chans = 0;
  if (f.fh.modular) nb_chans = (m.color.color_space == 1 && !m.xyb_encoded) ? 1 : 3;  // gray & no colour transform
  if (f.fh.modular && f.fh.do_ycbcr) nb_chans = 3;
  size_t nb_extra = m.extra.size();
  if (has_tree) {
    size_t limit = std::min<size_t>(1 << 22, 1024 + (size_t)f.w * f.h * (nb_chans + nb_extra) / 16);
    ReadTree(br, f.gtree.tree, limit);
    ReadEntropyCode(br, f.gtree.tree.num_leaves, f.gtree.code);
    f.gtree.present = true;
  }
  ModularImage& gi = f.gimg;
  gi.w = f.w; gi.h = f.h;
  gi.bitdepth = (int)m.depth.bits;
  for (int c = 0; c < nb_chans; c++) gi.channel.emplace_back(f.w, f.h);
  for (size_t e = 0; e < nb_extra; e++) {
    uint32_t ups = f.fh.ec_upsampling[e];
    if (ups != f.fh.upsampling) JXLO_FAIL("unsupported: extra channel upsampling differs from colour upsampling");
    gi.channel.emplace_back(f.w, f.h);
  }
  if (gi.channel.empty()) return;
  // GroupHeader + transforms + globally-decodable channels (without undoing transforms)
  GroupHeader& gh = f.gimg_header;
  ReadGroupHeader(br, gh);
  for (auto& t : gh.transforms) { MetaApply(gi, t); gi.transforms.push_back(t); }
  Tree local_tree; EntropyCode local_code;
  const Tree* tree; const EntropyCode* code;
  if (!gh.use_global_tree) {
    size_t npix = 0; for (auto& c : gi.channel) npix += (size_t)c.w * c.h;
    ReadTree(br, local_tree, std::min<size_t>(1 << 22, 1024 + npix));
    ReadEntropyCode(br, local_tree.num_leaves, local_code);
    tree = &local_tree; code = &local_code;
  } else {
    if (!f.gtree.present) JXLO_FAIL("global tree missing");
    tree = &f.gtree.tree; code = &f.gtree.code;
  }
  size_t end = gi.channel.size();
  uint32_t dist_mult = 0;
  const int maxsz = (int)f.fh.group_dim;
  for (size_t i = 0; i < gi.channel.size(); i++) {
    const Channel& c = gi.channel[i];
    if ((int)i >= gi.nb_meta_channels && (c.w > maxsz || c.h > maxsz)) { end = i; break; }
    dist_mult = std::max<uint32_t>(dist_mult, c.w);
  }
  // libjxl reads no symbols (not even the ANS state) when nothing is decodable?  It always initialises the reader.
  SymbolReader sr;
  sr.Init(code, br, dist_mult);
  for (size_t i = 0; i < end; i++) DecodeChannel(br, sr, gi, (int)i, *tree, gh.wp, 0);
  if (!sr.CheckFinal()) JXLO_FAIL("global modular ANS final state");
  f.tokens_modular += sr.tokens;
  f.gimg_global_decoded = end;
}

inline void ReadLfGlobal(BitReader& br, Frame& f) {
  // dec_frame.cc ProcessDCGlobal: image features first (patches, splines, noise), in this order
  if (f.fh.flags & kPatches) ReadPatches(br, f.m->extra.size(), (size_t)f.w * f.h, f.patches);
  if (f.fh.flags & kSplines) ReadSplines(br, (size_t)f.w * f.h, f.splines);
  if (f.fh.flags & kNoise) ReadNoise(br, f.noise);
  // LfChannelDequantization
  if (!br.Bool()) for (int c = 0; c < 3; c++) f.m_lf[c] = F16(br) * (1.0f / 128.0f);
  if (!f.fh.modular) {
    f.global_scale = U32(br, BitsOffset(11, 1), BitsOffset(11, 2049), BitsOffset(12, 4097), BitsOffset(16, 8193));
    f.quant_lf = U32(br, Val(16), BitsOffset(5, 1), BitsOffset(8, 1), BitsOffset(16, 1));
    ReadBlockCtxMap(br, f.bcm);
    if (!br.Bool()) {
      f.color_factor = U32(br, Val(84), Val(256), BitsOffset(8, 2), BitsOffset(16, 258));
      f.base_x = F16(br);
      f.base_b = F16(br);
      f.ytox_lf = (int32_t)br.u(8) - 128;
      f.ytob_lf = (int32_t)br.u(8) - 128;
    }
  }
  ReadGlobalModular(br, f);
}

// ---- modular group decode into the full image (dec_modular.cc DecodeGroup) ---------------------------------------
inline void DecodeModularGroup(BitReader& br, Frame& f, int x0, int y0, int xs, int ys, int min_shift, int max_shift, uint32_t stream_id) {
  ModularImage& full = f.gimg;
  size_t c = full.nb_meta_channels;
  const int gd = (int)f.fh.group_dim;
  for (; c < full.channel.size(); c++) {
    const Channel& fc = full.channel[c];
    if (fc.w > gd || fc.h > gd) break;
  }
  size_t beginc = c;
  ModularImage gi;
  gi.bitdepth = full.bitdepth;
  struct Pos { size_t c; int x, y, w, h; };
  std::vector<Pos> pos;
  for (c = beginc; c < full.channel.size(); c++) {
    const Channel& fc = full.channel[c];
    int shift = std::min(fc.hshift, fc.vshift);
    if (shift > max_shift || shift < min_shift) continue;
    int rx = x0 >> fc.hshift, ry = y0 >> fc.vshift, rw = xs >> fc.hshift, rh = ys >> fc.vshift;
    if (rx >= fc.w || ry >= fc.h) continue;
    rw = std::min(rw, fc.w - rx); rh = std::min(rh, fc.h - ry);
    if (rw <= 0 || rh <= 0) continue;
    gi.channel.emplace_back(rw, rh, fc.hshift, fc.vshift);
    pos.push_back({c, rx, ry, rw, rh});
  }
  if (gi.channel.empty()) return;
  gi.w = xs; gi.h = ys;
  ModularDecode(br, gi, stream_id, &f.gtree, 0, /*undo=*/true, &f.tokens_modular);
  JXLO_CHECK(gi.channel.size() == pos.size());
  for (size_t i = 0; i < pos.size(); i++) {
    Channel& fc = full.channel[pos[i].c];
    const Channel& g = gi.channel[i];
    JXLO_CHECK(g.w == pos[i].w && g.h == pos[i].h);
    for (int y = 0; y < g.h; y++) memcpy(fc.row(pos[i].y + y) + pos[i].x, g.row(y), sizeof(pixel_t) * g.w);
  }
}

// ---- LfGroup -------------------------------------------------------------------------------------------------------
inline void ReadLfGroup(BitReader& br, Frame& f, int g) {
  const int nlf = (int)f.fh.num_lf_groups;
  const int gx = g % (int)f.fh.xlfgroups, gy = g / (int)f.fh.xlfgroups;
  const int bx0 = gx * 256, by0 = gy * 256;
  const int gbw = std::min(256, f.bw - bx0), gbh = std::min(256, f.bh - by0);
  if (!f.fh.modular && !(f.fh.flags & kUseLfFrame)) {
    uint32_t extra_precision = br.u(2);
    ModularImage img;
    img.bitdepth = 16;
    static const int kChanOfStream[3] = {1, 0, 2};        // the stream's channels are Y, X, B
    for (int i = 0; i < 3; i++) img.channel.emplace_back(gbw >> f.hs[kChanOfStream[i]], gbh >> f.vs[kChanOfStream[i]]);   // (dec_modular.cc DecodeVarDCTDC)
    img.w = gbw; img.h = gbh;
    ModularDecode(br, img, 1 + g, &f.gtree, 0, true, &f.tokens_lf);
    if (f.subsampled) {
      // compressed_dc.cc DequantDC, the branch without chroma-from-luma: every channel on its own grid
      const float mul = 1.0f / (float)(1 << extra_precision);
      const float inv_quant_lf = InvGlobalScale(f) / (float)f.quant_lf;
      for (int i = 0; i < 3; i++) {
        const int c = kChanOfStream[i];
        const float fac = (f.m_lf[c] * inv_quant_lf) * mul;
        const Channel& ch = img.channel[i];
        for (int y = 0; y < ch.h; y++) for (int x = 0; x < ch.w; x++) {
          const size_t o = (size_t)((by0 >> f.vs[c]) + y) * f.bw + (bx0 >> f.hs[c]) + x;
          f.lfq[c][o] = ch.row(y)[x];
          f.lf.p[c].d[o] = (float)ch.row(y)[x] * fac;
        }
      }
    } else {
    // dequant: compressed_dc.cc DequantDC; modular channel order is Y, X, B
    const float mul = 1.0f / (float)(1 << extra_precision);
    const float inv_quant_lf = InvGlobalScale(f) / (float)f.quant_lf;
    float fac[3];
    for (int c = 0; c < 3; c++) fac[c] = (f.m_lf[c] * inv_quant_lf) * mul;
    const float cfl_x = f.base_x + (float)f.ytox_lf * (1.0f / (float)f.color_factor);
    const float cfl_b = f.base_b + (float)f.ytob_lf * (1.0f / (float)f.color_factor);
    for (int y = 0; y < gbh; y++) {
      const pixel_t* qy = img.channel[0].row(y);
      const pixel_t* qx = img.channel[1].row(y);
      const pixel_t* qb = img.channel[2].row(y);
      for (int x = 0; x < gbw; x++) {
        size_t o = (size_t)(by0 + y) * f.bw + bx0 + x;
        f.lfq[0][o] = qx[x]; f.lfq[1][o] = qy[x]; f.lfq[2][o] = qb[x];
        float vy = (float)qy[x] * fac[1];
        float vx = (float)qx[x] * fac[0];
        float vb = (float)qb[x] * fac[2];
        f.lf.p[1].d[o] = vy;
        f.lf.p[0].d[o] = std::fmaf(vy, cfl_x, vx);
        f.lf.p[2].d[o] = std::fmaf(vy, cfl_b, vb);
      }
    }
    }
  }
  // ModularLfGroup
  const int lfd = (int)f.fh.group_dim * 8;
  DecodeModularGroup(br, f, gx * lfd, gy * lfd, lfd, lfd, 3, 1000, 1 + nlf + g);
  if (f.fh.modular) return;
  // HfMetadata (dec_frame.cc / ac_strategy / DecodeAcMetadata)
  uint32_t nb_blocks = 1 + br.u(CeilLog2((uint32_t)(gbw * gbh)));
  const int cw = (gbw + 7) / 8, ch = (gbh + 7) / 8;
  ModularImage img;
  img.bitdepth = 8;
  img.channel.emplace_back(cw, ch);
  img.channel.emplace_back(cw, ch);
  img.channel.emplace_back((int)nb_blocks, 2);
  img.channel.emplace_back(gbw, gbh);
  ModularDecode(br, img, 1 + 2 * nlf + g, &f.gtree, 0, true, &f.tokens_lf);
  for (int y = 0; y < ch; y++) for (int x = 0; x < cw; x++) {
    size_t o = (size_t)(gy * 32 + y) * f.cw + gx * 32 + x;
    int a = img.channel[0].row(y)[x], b = img.channel[1].row(y)[x];
    if (a < -128 || a > 127 || b < -128 || b > 127) JXLO_FAIL("cfl factor out of range");
    f.ytox_map[o] = (int8_t)a; f.ytob_map[o] = (int8_t)b;
  }
  for (int y = 0; y < gbh; y++) for (int x = 0; x < gbw; x++) {
    int s = img.channel[3].row(y)[x];
    if (s < 0 || s > 7) JXLO_FAIL("sharpness out of range");
    f.sharpness[(size_t)(by0 + y) * f.bw + bx0 + x] = (uint8_t)s;
  }
  uint32_t num = 0;
  std::vector<uint8_t> covered((size_t)gbw * gbh, 0);
  for (int y = 0; y < gbh; y++) {
    for (int x = 0; x < gbw; x++) {
      if (covered[(size_t)y * gbw + x]) continue;
      if (num >= nb_blocks) JXLO_FAIL("not enough varblocks");
      int s = img.channel[2].row(0)[num];
      int q = img.channel[2].row(1)[num];
      num++;
      if (s < 0 || s >= 27) JXLO_FAIL("bad strategy");
      if (q < 0 || q > 255) JXLO_FAIL("bad hf_mul");
      int cx = kCoveredX[s], cy = kCoveredY[s];
      if (x + cx > gbw || y + cy > gbh) JXLO_FAIL("varblock exceeds LF group");
      // must not cross a 256x256 px group (32 blocks) boundary
      if ((x % 32) + cx > 32 || (y % 32) + cy > 32) JXLO_FAIL("varblock crosses group");
      for (int iy = 0; iy < cy; iy++) for (int ix = 0; ix < cx; ix++) {
        size_t lo = (size_t)(y + iy) * gbw + x + ix;
        if (covered[lo]) JXLO_FAIL("overlapping varblocks");
        covered[lo] = 1;
        size_t o = (size_t)(by0 + y + iy) * f.bw + bx0 + x + ix;
        f.strategy[o] = (uint8_t)s;
        f.is_first[o] = (ix == 0 && iy == 0);
        f.hf_mul[o] = q + 1;
      }
    }
  }
}

// ---- HfGlobal -------------------------------------------------------------------------------------------------------
inline void ReadDctParams(BitReader& br, DctBandParams& p) {
  p.num_bands = br.u(4) + 1;
  for (int c = 0; c < 3; c++) for (int i = 0; i < p.num_bands; i++) p.bands[c][i] = F16(br);
  for (int c = 0; c < 3; c++) p.bands[c][0] *= 64.0f;
}

inline void ReadHfGlobal(BitReader& br, Frame& f) {
  const int nlf = (int)f.fh.num_lf_groups;
  // quant_weights.cc DequantMatrices::Decode
  bool all_default = br.Bool();
  for (int k = 0; k < 17; k++) f.qenc[k] = QuantEncoding();
  if (!all_default) {
    for (int k = 0; k < 17; k++) {
      QuantEncoding& q = f.qenc[k];
      q.mode = br.u(3);
      const int rows = 8 * kKindRows[k], cols = 8 * kKindCols[k];
      switch (q.mode) {
        case 0: break;
        case 1:
          if (k != QIDENTITY) JXLO_FAIL("identity mode on wrong table");
          for (int c = 0; c < 3; c++) for (int i = 0; i < 3; i++) q.idweights[c][i] = F16(br) * 64.0f;
          break;
        case 2:
          if (k != QDCT2X2) JXLO_FAIL("dct2 mode on wrong table");
          for (int c = 0; c < 3; c++) for (int i = 0; i < 6; i++) q.dct2weights[c][i] = F16(br) * 64.0f;
          break;
        case 3:
          if (k != QDCT4X4) JXLO_FAIL("dct4 mode on wrong table");
          for (int c = 0; c < 3; c++) for (int i = 0; i < 2; i++) q.dct4multipliers[c][i] = F16(br);
          ReadDctParams(br, q.dct);
          break;
        case 4:
          if (k != QDCT4X8) JXLO_FAIL("dct4x8 mode on wrong table");
          for (int c = 0; c < 3; c++) q.dct4x8multipliers[c] = F16(br);
          ReadDctParams(br, q.dct);
          break;
        case 5:
          if (k != QAFV) JXLO_FAIL("afv mode on wrong table");
          for (int c = 0; c < 3; c++) for (int i = 0; i < 9; i++) { q.afv_weights[c][i] = F16(br); if (i < 6) q.afv_weights[c][i] *= 64.0f; }
          ReadDctParams(br, q.dct);
          ReadDctParams(br, q.dct4x4);
          break;
        case 6: ReadDctParams(br, q.dct); break;
        case 7: {
          q.raw_den = F16(br);
          ModularImage img;
          img.bitdepth = 8;
          for (int c = 0; c < 3; c++) img.channel.emplace_back(cols, rows);
          ModularDecode(br, img, 1 + 3 * nlf + k, &f.gtree, 0, true, &f.tokens_modular);
          for (int c = 0; c < 3; c++) q.raw[c].assign(img.channel[c].data.begin(), img.channel[c].data.end());
          break;
        }
      }
    }
  }
  // tables are computed lazily per used kind (see EnsureQuantTable)
  f.num_hf_presets = 1 + br.u(CeilLog2(f.fh.num_groups));
  // per pass: coefficient orders + histograms (dec_frame.cc ProcessACGlobal, coeff_order.cc DecodeCoeffOrders)
  f.pass.resize(f.fh.passes.num_passes);
  for (auto& ps : f.pass) {
    uint32_t used_orders = U32(br, Val(0x5F), Val(0x13), Val(0), Bits(13));
    ps.order.assign(13 * 3, {});
    EntropyCode oc; SymbolReader osr;
    if (used_orders) { ReadEntropyCode(br, 8, oc); osr.Init(&oc, br); }
    for (int b = 0; b < 13; b++) {
      std::vector<uint32_t> natural = NaturalCoeffOrder(kBucketStrategy[b]);
      for (int c = 0; c < 3; c++) {
        if (used_orders & (1u << b)) {
          size_t size = natural.size();
          std::vector<uint32_t> perm;
          ReadPermutation(br, osr, size / 64, size, perm);
          std::vector<uint32_t>& o = ps.order[b * 3 + c];
          o.resize(size);
          for (size_t i = 0; i < size; i++) o[i] = natural[perm[i]];
        } else {
          ps.order[b * 3 + c] = natural;
        }
      }
    }
    if (used_orders && !osr.CheckFinal()) JXLO_FAIL("coefficient order ANS final state");
    ReadEntropyCode(br, (int)(495 * f.bcm.num_ctxs * f.num_hf_presets), ps.code);
  }
}

inline const std::vector<float>& EnsureQuantTable(Frame& f, int kind, int c) {
  if (f.qtable[kind][c].empty()) ComputeQuantTable(f.qenc[kind], kind, c, f.qtable[kind][c]);
  return f.qtable[kind][c];
}

// ---- PassGroup ------------------------------------------------------------------------------------------------------
static const uint16_t kCoeffFreqContext[64] = {0xBAD, 0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 15, 16, 16, 17, 17,
                                               18,    18, 19, 19, 20, 20, 21, 21, 22, 22, 23, 23, 23, 23, 24, 24, 24, 24, 25, 25, 25, 25,
                                               26,    26, 26, 26, 27, 27, 27, 27, 28, 28, 28, 28, 29, 29, 29, 29, 30, 30, 30, 30};
static const uint16_t kCoeffNumNonzeroContext[64] = {0xBAD, 0,   31,  62,  62,  93,  93,  93,  93,  123, 123, 123, 123, 152, 152, 152,
                                                     152,   152, 152, 152, 152, 180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 180,
                                                     180,   206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206,
                                                     206,   206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206};

inline void ReadPassGroup(BitReader& br, Frame& f, int pass_idx, int g) {
  const int nlf = (int)f.fh.num_lf_groups;
  const int gx = g % (int)f.fh.xgroups, gy = g / (int)f.fh.xgroups;
  if (!f.fh.modular) {
    const int bx0 = gx * 32, by0 = gy * 32;
    const int gbw = std::min(32, f.bw - bx0), gbh = std::min(32, f.bh - by0);
    PassInfo& ps = f.pass[pass_idx];
    uint32_t preset = br.u(CeilLog2(f.num_hf_presets));
    if (preset >= f.num_hf_presets) JXLO_FAIL("bad hf preset");
    const int nctx = f.bcm.num_ctxs;
    const size_t ctx_offset = (size_t)495 * nctx * preset;
    SymbolReader sr;
    sr.Init(&ps.code, br);
    const uint32_t shift = (pass_idx + 1 < (int)f.fh.passes.num_passes) ? f.fh.passes.shift[pass_idx] : 0;
    std::vector<uint8_t> nzmap[3];
    for (auto& v : nzmap) v.assign(32 * 32, 0);
    size_t offset = 0;
    for (int by = 0; by < gbh; by++) {
      for (int bx = 0; bx < gbw; bx++) {
        size_t o = (size_t)(by0 + by) * f.bw + bx0 + bx;
        if (!f.is_first[o]) continue;
        const int s = f.strategy[o];
        const int cx = kCoveredX[s], cy = kCoveredY[s];
        const int covered = cx * cy, log2cov = Log2Int(covered);
        const int size = covered * 64;
        const int ord = kOrderBucket[s];
        // LF context index from quantised LF
        int lf_idx = 0;
        if (f.bcm.num_lf_ctxs > 1 && !(f.fh.flags & kUseLfFrame)) {     // (dec_cache.cc: a frame that takes its LF from an LF frame has quant_dc zero-filled)
          int bX = 0, bY = 0, bB = 0;
          auto at = [&](int c) { return f.lfq[c][(size_t)((by0 + by) >> f.vs[c]) * f.bw + ((bx0 + bx) >> f.hs[c])]; };   // (quant_dc is kept at full resolution)
          for (int32_t t : f.bcm.lf_thresholds[0]) if (at(0) > t) bX++;
          for (int32_t t : f.bcm.lf_thresholds[1]) if (at(1) > t) bY++;
          for (int32_t t : f.bcm.lf_thresholds[2]) if (at(2) > t) bB++;
          lf_idx = (bX * ((int)f.bcm.lf_thresholds[2].size() + 1) + bB) * ((int)f.bcm.lf_thresholds[1].size() + 1) + bY;
        }
        static const int chan_order[3] = {1, 0, 2};
        for (int ci = 0; ci < 3; ci++) {
          const int c = chan_order[ci];
          // dec_group.cc: a subsampled channel only has a block where the block starts one of its cells; its "non-zeros"
          // neighbourhood lives on its own grid
          const int sbx = bx >> f.hs[c], sby = by >> f.vs[c];
          if ((sbx << f.hs[c]) != bx || (sby << f.vs[c]) != by) continue;
          if (f.subsampled && s != 0) JXLO_FAIL("unsupported: chroma subsampling with a transform other than DCT8");
          const int block_ctx = f.bcm.Context(lf_idx, (uint32_t)f.hf_mul[o], ord, c);
          // predicted nzeros
          int pred;
          if (f.subsampled) {
            if (sbx == 0) pred = sby == 0 ? 32 : nzmap[c][(sby - 1) * 32 + sbx];
            else if (sby == 0) pred = nzmap[c][sby * 32 + sbx - 1];
            else pred = (nzmap[c][(sby - 1) * 32 + sbx] + nzmap[c][sby * 32 + sbx - 1] + 1) / 2;
          } else
          if (bx == 0) pred = by == 0 ? 32 : nzmap[c][(by - 1) * 32 + bx];
          else if (by == 0) pred = nzmap[c][by * 32 + bx - 1];
          else pred = (nzmap[c][(by - 1) * 32 + bx] + nzmap[c][by * 32 + bx - 1] + 1) / 2;
          int pc = pred > 64 ? 64 : pred;
          size_t nz_ctx = ctx_offset + (pc < 8 ? (size_t)block_ctx + (size_t)nctx * pc : (size_t)block_ctx + (size_t)nctx * (4 + pc / 2));
          uint32_t nzeros = sr.Read(br, (int)nz_ctx);
          if (nzeros + covered > (uint32_t)size) JXLO_FAIL("nzeros too large");
          uint8_t nzm = (uint8_t)((nzeros + covered - 1) >> log2cov);
          if (f.subsampled) nzmap[c][sby * 32 + sbx] = nzm;
          else for (int iy = 0; iy < cy; iy++) for (int ix = 0; ix < cx; ix++) nzmap[c][(by + iy) * 32 + bx + ix] = nzm;
          const size_t histo_offset = ctx_offset + (size_t)37 * nctx + (size_t)458 * block_ctx;
          const std::vector<uint32_t>& order = ps.order[ord * 3 + c];
          int32_t* blk = f.coeffs[c][g].data() + offset;
          uint32_t prev = nzeros > (uint32_t)size / 16 ? 0 : 1;
          for (int k = covered; k < size && nzeros != 0; k++) {
            uint32_t nzl = (nzeros + covered - 1) >> log2cov;
            uint32_t kk = (uint32_t)k >> log2cov;
            size_t ctx = histo_offset + (kCoeffNumNonzeroContext[nzl] + kCoeffFreqContext[kk]) * 2 + prev;
            uint32_t u = sr.Read(br, (int)ctx);
            int32_t v = UnpackSigned(u);
            prev = u != 0;
            nzeros -= prev;
            blk[order[k]] += v * (1 << shift);
          }
          if (nzeros != 0) JXLO_FAIL("nzeros != 0 at end of block");
        }
        offset += size;
      }
    }
    if (!sr.CheckFinal()) JXLO_FAIL("AC group ANS final state");
    f.tokens_hf += sr.tokens;
  }
  // modular channels of this group
  const int gd = (int)f.fh.group_dim;
  int min_shift = 0, max_shift = 2;
  if (f.fh.passes.num_passes > 1) {
    // dec_frame.cc: per-pass shift ranges derived from downsampling
    int maxs = 2, mins = 3;   // passes.h GetDownsamplingBracket: maxShift = 2, minShift = 3 to start with
    uint32_t np = f.fh.passes.num_passes;
    // passes.GetDownsamplingBracket
    for (uint32_t i = 0;; i++) {
      for (uint32_t j = 0; j < f.fh.passes.num_ds; j++) if (i == f.fh.passes.last_pass[j]) mins = Log2Int(f.fh.passes.downsample[j]);
      if (i + 1 == np) mins = 0;
      if (i == (uint32_t)pass_idx) break;
      maxs = mins - 1;
    }
    min_shift = mins; max_shift = maxs;
  }
  DecodeModularGroup(br, f, gx * gd, gy * gd, gd, gd, min_shift, max_shift, 1 + 3 * nlf + 17 + f.fh.num_groups * pass_idx + g);
}

// ---- reconstruction -----------------------------------------------------------------------------------------------
inline void DequantAndIDCT(Frame& f) {
  const ImageMetadata& m = *f.m;
  f.xyb.p[0] = Plane(f.bw * 8, f.bh * 8); f.xyb.p[1] = Plane(f.bw * 8, f.bh * 8); f.xyb.p[2] = Plane(f.bw * 8, f.bh * 8);
  const float inv_gs = InvGlobalScale(f);
  const float x_dm = std::pow(0.8f, (float)f.fh.x_qm_scale - 2.0f);
  const float b_dm = std::pow(0.8f, (float)f.fh.b_qm_scale - 2.0f);
  const float cscale = 1.0f / (float)f.color_factor;
  std::vector<float> blk[3];
  for (uint32_t g = 0; g < f.fh.num_groups; g++) {
    const int gx = g % f.fh.xgroups, gy = g / f.fh.xgroups;
    const int bx0 = gx * 32, by0 = gy * 32;
    const int gbw = std::min(32, f.bw - bx0), gbh = std::min(32, f.bh - by0);
    size_t offset = 0;
    for (int by = 0; by < gbh; by++) for (int bx = 0; bx < gbw; bx++) {
      size_t o = (size_t)(by0 + by) * f.bw + bx0 + bx;
      if (!f.is_first[o]) continue;
      const int s = f.strategy[o];
      const int cx = kCoveredX[s], cy = kCoveredY[s];
      const int size = cx * cy * 64;
      const int kind = kQuantKind[s];
      const float sd = inv_gs / (float)f.hf_mul[o];
      const float sdc[3] = {sd * x_dm, sd, sd * b_dm};
      if (f.subsampled) {
        // every channel on its own grid, 8x8 DCT only, no chroma-from-luma (JPEG transcodes never carry any)
        const size_t tile = (size_t)((by0 + by) / 8) * f.cw + (bx0 + bx) / 8;
        if (f.base_x != 0.f || f.base_b != 0.f || f.ytox_map[tile] != 0 || f.ytob_map[tile] != 0) JXLO_FAIL("unsupported: chroma from luma in a chroma-subsampled frame");
        for (int c = 0; c < 3; c++) {
          const int sbx = (bx0 + bx) >> f.hs[c], sby = (by0 + by) >> f.vs[c];
          if ((sbx << f.hs[c]) != bx0 + bx || (sby << f.vs[c]) != by0 + by) continue;
          blk[c].assign(size, 0.f);
          const std::vector<float>& table = EnsureQuantTable(f, kind, c);
          const int32_t* q = f.coeffs[c][g].data() + offset;
          for (int k = 0; k < size; k++) blk[c][k] = AdjustQuantBias(c, q[k], m.quant_bias) * (table[k] * sdc[c]);
          LowestFrequenciesFromLF(s, &f.lf.p[c].d[(size_t)sby * f.bw + sbx], f.bw, blk[c].data());
          InverseTransform(s, blk[c].data(), f.xyb.p[c].row(sby * 8) + sbx * 8, f.xyb.p[c].w);
        }
        offset += size;
        continue;
      }
      for (int c = 0; c < 3; c++) {
        blk[c].assign(size, 0.f);
        const std::vector<float>& table = EnsureQuantTable(f, kind, c);
        const int32_t* q = f.coeffs[c][g].data() + offset;
        for (int k = 0; k < size; k++) blk[c][k] = AdjustQuantBias(c, q[k], m.quant_bias) * (table[k] * sdc[c]);
      }
      // chroma from luma (only for XYB / non-subsampled)
      size_t tile = (size_t)((by0 + by) / 8) * f.cw + (bx0 + bx) / 8;
      const float kx = f.base_x + (float)f.ytox_map[tile] * cscale;
      const float kb = f.base_b + (float)f.ytob_map[tile] * cscale;
      for (int k = 0; k < size; k++) {
        blk[0][k] = std::fmaf(kx, blk[1][k], blk[0][k]);
        blk[2][k] = std::fmaf(kb, blk[1][k], blk[2][k]);
      }
      for (int c = 0; c < 3; c++) {
        LowestFrequenciesFromLF(s, &f.lf.p[c].d[o], f.bw, blk[c].data());
        float* out = f.xyb.p[c].row((by0 + by) * 8) + (bx0 + bx) * 8;
        InverseTransform(s, blk[c].data(), out, f.xyb.p[c].w);
      }
      offset += size;
    }
  }
}

// stage_chroma_upsampling.cc: horizontal, then vertical 2x upsampling of a subsampled channel with the (1/4, 3/4) kernel,
// out[2x] = 0.25 in[x-1] + 0.75 in[x], out[2x+1] = 0.25 in[x+1] + 0.75 in[x], neighbours mirrored at the channel's own edges
// (the channel covers ceil(size / 2) samples of the image) [R]
inline void UpsampleChroma(Frame& f) {
  for (int c = 0; c < 3; c++) {
    if (f.hs[c] == 0 && f.vs[c] == 0) continue;
    int cw = (f.w + (1 << f.hs[c]) - 1) >> f.hs[c], ch = (f.h + (1 << f.vs[c]) - 1) >> f.vs[c];
    Plane cur(cw, ch);
    for (int y = 0; y < ch; y++) memcpy(cur.row(y), f.xyb.p[c].row(y), sizeof(float) * cw);
    if (f.hs[c]) {
      Plane out(2 * cw, ch);
      for (int y = 0; y < ch; y++) {
        const float* in = cur.row(y);
        float* o = out.row(y);
        for (int x = 0; x < cw; x++) {
          const float mid = in[x] * 0.75f, prev = in[x ? x - 1 : 0], next = in[x + 1 < cw ? x + 1 : cw - 1];
          o[2 * x] = std::fmaf(0.25f, prev, mid);
          o[2 * x + 1] = std::fmaf(0.25f, next, mid);
        }
      }
      cur = out; cw *= 2;
    }
    if (f.vs[c]) {
      Plane out(cw, 2 * ch);
      for (int y = 0; y < ch; y++) {
        const float* in = cur.row(y);
        const float* up = cur.row(y ? y - 1 : 0);
        const float* down = cur.row(y + 1 < ch ? y + 1 : ch - 1);
        float* o0 = out.row(2 * y);
        float* o1 = out.row(2 * y + 1);
        for (int x = 0; x < cw; x++) {
          const float mid = in[x] * 0.75f;
          o0[x] = std::fmaf(0.25f, up[x], mid);
          o1[x] = std::fmaf(0.25f, down[x], mid);
        }
      }
      cur = out; ch *= 2;
    }
    Plane full(f.bw * 8, f.bh * 8);
    for (int y = 0; y < std::min(ch, f.bh * 8); y++) memcpy(full.row(y), cur.row(y), sizeof(float) * std::min(cw, f.bw * 8));
    f.xyb.p[c] = full;
  }
}

inline Image3 CropImage(const Image3& in, int w, int h) {
  Image3 out;
  for (int c = 0; c < 3; c++) {
    out.p[c] = Plane(w, h);
    for (int y = 0; y < h; y++) memcpy(out.p[c].row(y), in.p[c].row(y), sizeof(float) * w);
  }
  return out;
}

}  // namespace jxlo
