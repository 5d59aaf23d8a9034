// jxlsynth — YCbCr VarDCT frames with chroma subsampling (fixture generator; NOT on the product decode path, independent of oracle/).
// The shape `cjxl photo.jpg` gives a 4:2:0 / 4:2:2 / 4:4:0 JPEG: a non-XYB image, do_YCbCr frame, 8x8 DCT only, no restoration
// filters, no chroma-from-luma, adaptive LF smoothing off, every channel on its own block grid (frame_header.h
// YCbCrChromaSubsampling, dec_group.cc).  Lossy and not JPEG-faithful (library default quantisation table, own quantiser): the
// streams exist so that decoders can be compared on the subsampled geometry.
#pragma once

namespace synth {

struct YCbCrQuantised {
  int w, h; int mode[3];
  std::vector<int32_t> lfq[3], qc[3];     // per channel: (bh >> vs) x (bw >> hs) LF values / x 64 coefficients in libjxl's (transposed) layout
  std::vector<int32_t> hf_mul;            // bw x bh
  uint32_t global_scale = 1, quant_lf = 16;
  bool custom_m_lf = false; float m_lf[3] = {0, 0, 0};
  const int32_t* raw_table = nullptr; float raw_den = 0;   // 3 x 64 (channel, libjxl layout) when the 8x8 DCT uses a RAW table
  bool gray = false;                      // grey image header (a one-component JPEG: the frame still has three channels, Cb = Cr = 0)
};

static void YCbCrGeometry(int w, int h, const int mode[3], int hs[3], int vs[3], int* bw, int* bh) {
  static const int kH[4] = {0, 1, 1, 0}, kV[4] = {0, 1, 0, 1};
  int maxhs = 0, maxvs = 0;
  for (int c = 0; c < 3; c++) { maxhs = std::max(maxhs, kH[mode[c]]); maxvs = std::max(maxvs, kV[mode[c]]); }
  for (int c = 0; c < 3; c++) { hs[c] = maxhs - kH[mode[c]]; vs[c] = maxvs - kV[mode[c]]; }
  *bw = ((w + (8 << maxhs) - 1) / (8 << maxhs)) << maxhs; *bh = ((h + (8 << maxvs) - 1) / (8 << maxvs)) << maxvs;
}

static std::vector<uint8_t> WriteYCbCrFrame(const YCbCrQuantised& in, const Params& p) {
  const int w = in.w, h = in.h;
  const int* mode = in.mode;
  int hs[3], vs[3], bw, bh;
  YCbCrGeometry(w, h, mode, hs, vs, &bw, &bh);
  const int xg = (w + 255) / 256, yg = (h + 255) / 256, ngroups = xg * yg;
  const int xlg = (w + 2047) / 2048, ylg = (h + 2047) / 2048, nlf = xlg * ylg;
  int bwc[3], bhc[3];
  for (int c = 0; c < 3; c++) { bwc[c] = bw >> hs[c]; bhc[c] = bh >> vs[c]; }
  const std::vector<int32_t>* lfq = in.lfq;
  const std::vector<int32_t>* qc = in.qc;
  const std::vector<int32_t>& hf_mul = in.hf_mul;
  const uint32_t global_scale = in.global_scale, quant_lf = in.quant_lf;
  // ---- Modular streams: LF coefficients (per-channel size) + HF metadata under the fixed global tree
  std::vector<int> bfs;
  GTree gt = MakeGlobalTree(nlf, &bfs);
  const int root = bfs[0];
  std::vector<Token> tree_tokens;
  TreeTokens(gt, bfs, tree_tokens);
  struct LfData { std::vector<int32_t> ch[3], m[4]; int gbw, gbh, nb; std::vector<Token> lf_tok, meta_tok; };
  std::vector<LfData> lgd(nlf);
  for (int g = 0; g < nlf; g++) {
    LfData& d = lgd[g];
    const int gx = g % xlg, gy = g / xlg, bx0 = gx * 256, by0 = gy * 256;
    d.gbw = std::min(256, bw - bx0); d.gbh = std::min(256, bh - by0);
    static const int order[3] = {1, 0, 2};   // Y, Cb, Cr
    std::vector<ChanRef> cr;
    for (int i = 0; i < 3; i++) {
      const int c = order[i], cwid = d.gbw >> hs[c], chei = d.gbh >> vs[c];
      d.ch[i].resize((size_t)cwid * chei);
      for (int y = 0; y < chei; y++) for (int x = 0; x < cwid; x++) d.ch[i][(size_t)y * cwid + x] = lfq[c][(size_t)((by0 >> vs[c]) + y) * bwc[c] + (bx0 >> hs[c]) + x];
      cr.push_back({d.ch[i].data(), cwid, chei});
    }
    ModularTokens(gt, root, cr, 1 + g, d.lf_tok);
    const int mcw = (d.gbw + 7) / 8, mch = (d.gbh + 7) / 8;
    d.m[0].assign((size_t)mcw * mch, 0); d.m[1].assign((size_t)mcw * mch, 0);      // no chroma-from-luma
    d.nb = d.gbw * d.gbh;
    d.m[2].assign((size_t)d.nb, 0);                                                // strategy DCT8 everywhere
    for (int y = 0; y < d.gbh; y++) for (int x = 0; x < d.gbw; x++) d.m[2].push_back(hf_mul[(size_t)(by0 + y) * bw + bx0 + x] - 1);
    d.m[3].assign((size_t)d.gbw * d.gbh, 0);
    std::vector<ChanRef> mr{{d.m[0].data(), mcw, mch}, {d.m[1].data(), mcw, mch}, {d.m[2].data(), d.nb, 2}, {d.m[3].data(), d.gbw, d.gbh}};
    ModularTokens(gt, root, mr, 1 + 2 * nlf + g, d.meta_tok);
  }
  // ---- AC tokens per group: a channel takes part in a block only where the block starts one of its (larger) cells
  const int nctx = 15;
  const std::vector<uint32_t> order = NaturalOrder(S_DCT);
  std::vector<std::vector<Token>> ac_tok(ngroups);
  for (int g = 0; g < ngroups; g++) {
    const int gx = g % xg, gy = g / xg, bx0 = gx * 32, by0 = gy * 32;
    const int gbw = std::min(32, bw - bx0), gbh = std::min(32, bh - by0);
    uint8_t nzmap[3][1024];
    memset(nzmap, 0, sizeof(nzmap));
    std::vector<Token>& tk = ac_tok[g];
    for (int by = 0; by < gbh; by++) for (int bx = 0; bx < gbw; bx++) {
      static const int chan[3] = {1, 0, 2};
      for (int ci = 0; ci < 3; ci++) {
        const int c = chan[ci];
        const int sbx = bx >> hs[c], sby = by >> vs[c];
        if ((sbx << hs[c]) != bx || (sby << vs[c]) != by) continue;
        const int block_ctx = kDefaultBlockCtx[(c < 2 ? (c ^ 1) : 2) * 13 + 0];
        const int32_t* q = qc[c].data() + ((size_t)((by0 >> vs[c]) + sby) * bwc[c] + (bx0 >> hs[c]) + sbx) * 64;
        int nz = 0;
        for (int k = 1; k < 64; k++) nz += q[order[k]] != 0;
        int pred;
        if (sbx == 0) pred = sby == 0 ? 32 : nzmap[c][(sby - 1) * 32 + sbx];
        else if (sby == 0) pred = nzmap[c][sby * 32 + sbx - 1];
        else pred = (nzmap[c][(sby - 1) * 32 + sbx] + nzmap[c][sby * 32 + sbx - 1] + 1) / 2;
        const int pc = std::min(pred, 64);
        tk.push_back({(uint32_t)(pc < 8 ? block_ctx + nctx * pc : block_ctx + nctx * (4 + pc / 2)), (uint32_t)nz});
        nzmap[c][sby * 32 + sbx] = (uint8_t)nz;
        const uint32_t histo = 37 * nctx + 458 * block_ctx;
        int left = nz;
        uint32_t prev = nz > 4 ? 0 : 1;
        for (int k = 1; k < 64 && left != 0; k++) {
          const uint32_t ctx = histo + (kNzCtx[left] + kFreqCtx[k]) * 2 + prev;
          const uint32_t u = PackSigned(q[order[k]]);
          tk.push_back({ctx, u});
          prev = u != 0;
          left -= prev;
        }
      }
    }
  }
  // RAW quantisation table of the 8x8 DCT (quant_weights.cc kQuantModeRAW): a 3-channel 8x8 Modular image under the global tree,
  // stream id 1 + 3 * nlf + kind
  std::vector<Token> raw_tok;
  if (in.raw_table) {
    std::vector<ChanRef> cr;
    for (int c = 0; c < 3; c++) cr.push_back({in.raw_table + 64 * c, 8, 8});
    ModularTokens(gt, root, cr, 1 + 3 * nlf + 0, raw_tok);
  }
  EntropyCoder tree_code, mod_code, ac_code;
  { std::vector<const std::vector<Token>*> s{&tree_tokens}; BuildEntropyCoder(s, 6, UintConfig{4, 2, 0}, 6, tree_code); }
  { std::vector<const std::vector<Token>*> s; for (auto& d : lgd) { s.push_back(&d.lf_tok); s.push_back(&d.meta_tok); } s.push_back(&raw_tok); BuildEntropyCoder(s, gt.num_leaves, UintConfig{4, 2, 0}, 32, mod_code); }
  { std::vector<const std::vector<Token>*> s; for (auto& t : ac_tok) s.push_back(&t); BuildEntropyCoder(s, 495 * nctx, UintConfig{4, 2, 0}, 96, ac_code); }
  std::vector<BitWriter> sections;
  {  // LfGlobal
    BitWriter s;
    if (!in.custom_m_lf) s.put(1, 1);  // LfChannelDequantization all_default
    else { s.put(0, 1); for (int c = 0; c < 3; c++) WriteF16(s, in.m_lf[c] * 128.0f); }
    WriteU32(s, global_scale, {11, 1}, {11, 2049}, {12, 4097}, {16, 8193});
    WriteU32(s, quant_lf, {0, 16}, {5, 1}, {8, 1}, {16, 1});
    s.put(1, 1);  // default BlockCtxMap
    // LfChannelCorrelation without any chroma-from-luma (the default has base_correlation_b = 1, an XYB habit): colour factor 84, bases 0
    s.put(0, 1); WriteU32(s, 84, {0, 84}, {0, 256}, {8, 2}, {16, 258}); WriteF16(s, 0.0f); WriteF16(s, 0.0f); s.put(128, 8); s.put(128, 8);
    s.put(1, 1);  // GlobalModular: has_tree
    WriteEntropyCode(s, tree_code);
    EncodeTokens(s, tree_code, tree_tokens);
    WriteEntropyCode(s, mod_code);
    sections.push_back(s);
  }
  for (int g = 0; g < nlf; g++) {  // LfGroup
    BitWriter s;
    LfData& d = lgd[g];
    s.put(0, 2);  // extra_precision
    s.put(1, 1); s.put(1, 1); s.put(0, 2);
    EncodeTokens(s, mod_code, d.lf_tok);
    s.put(d.nb - 1, CeilLog2((uint32_t)(d.gbw * d.gbh)));
    s.put(1, 1); s.put(1, 1); s.put(0, 2);
    EncodeTokens(s, mod_code, d.meta_tok);
    sections.push_back(s);
  }
  {  // HfGlobal: library default dequantisation matrices (or a RAW table for the 8x8 DCT: JPEG transcodes), one preset, natural orders
    BitWriter s;
    if (!in.raw_table) s.put(1, 1);
    else {
      s.put(0, 1);
      for (int k = 0; k < 17; k++) {
        if (k != 0) { s.put(0, 3); continue; }                     // library default for everything but kind 0
        s.put(7, 3);                                                // kQuantModeRAW
        WriteF16(s, in.raw_den);
        s.put(1, 1); s.put(1, 1); s.put(0, 2);                      // GroupHeader: global tree, default WP, no transforms
        EncodeTokens(s, mod_code, raw_tok);
      }
    }
    s.put(0, CeilLog2((uint32_t)ngroups));
    s.put(2, 2);
    WriteEntropyCode(s, ac_code);
    sections.push_back(s);
  }
  for (int g = 0; g < ngroups; g++) { BitWriter s; EncodeTokens(s, ac_code, ac_tok[g]); sections.push_back(s); }
  BitWriter out;
  Params q = p;
  q.gab = 0; q.epf_iters = 0; q.noise = 0; q.upsampling = 1; q.num_passes = 1; q.skip_lf_smoothing = 1; q.out_bits = 8; q.hdr = 0;
  q.do_ycbcr = 1; for (int c = 0; c < 3; c++) q.jpeg_upsampling[c] = mode[c];
  WriteImageHeader(out, w, h, q, false, 8, false, in.gray);
  WriteFrameHeader(out, q, false, false, 0, 1, false, w, h);
  WriteTOCAndSections(out, sections, ngroups == 1, 0);
  out.align();
  return out.bytes;
}


// sampling-factor modes per jxl channel (0 = Cb, 1 = Y, 2 = Cr): 0 = 1x1, 1 = 2x2, 2 = 2x1 (horizontal factor 2), 3 = 1x2
static std::vector<uint8_t> EncodeYCbCr(const uint8_t* rgb8, int w, int h, const int mode[3], const Params& p) {
  static const int kH[4] = {0, 1, 1, 0}, kV[4] = {0, 1, 0, 1};
  int maxhs = 0, maxvs = 0, hs[3], vs[3];
  for (int c = 0; c < 3; c++) { maxhs = std::max(maxhs, kH[mode[c]]); maxvs = std::max(maxvs, kV[mode[c]]); }
  for (int c = 0; c < 3; c++) { hs[c] = maxhs - kH[mode[c]]; vs[c] = maxvs - kV[mode[c]]; }
  const int bw = ((w + (8 << maxhs) - 1) / (8 << maxhs)) << maxhs, bh = ((h + (8 << maxvs) - 1) / (8 << maxvs)) << maxvs;
  const int xg = (w + 255) / 256, yg = (h + 255) / 256, ngroups = xg * yg;
  const int xlg = (w + 2047) / 2048, ylg = (h + 2047) / 2048, nlf = xlg * ylg;
  int bwc[3], bhc[3];
  for (int c = 0; c < 3; c++) { bwc[c] = bw >> hs[c]; bhc[c] = bh >> vs[c]; }
  Pcg32 rng(p.seed * 613 + 11);
  // ---- planes: YCbCr (JPEG matrix, Y centred like the decoder expects), chroma box-averaged, edge-replicated to whole blocks
  std::vector<float> full[3];
  for (auto& v : full) v.resize((size_t)w * h);
  for (size_t i = 0; i < (size_t)w * h; i++) {
    const float R = rgb8[3 * i] / 255.0f, G = rgb8[3 * i + 1] / 255.0f, B = rgb8[3 * i + 2] / 255.0f;
    full[1][i] = 0.299f * R + 0.587f * G + 0.114f * B - 128.0f / 255.0f;
    full[0][i] = -0.168736f * R - 0.331264f * G + 0.5f * B;
    full[2][i] = 0.5f * R - 0.418688f * G - 0.081312f * B;
  }
  std::vector<float> pl[3];
  for (int c = 0; c < 3; c++) {
    const int pw = bwc[c] * 8, ph = bhc[c] * 8, fx = 1 << hs[c], fy = 1 << vs[c];
    pl[c].resize((size_t)pw * ph);
    for (int y = 0; y < ph; y++) for (int x = 0; x < pw; x++) {
      float acc = 0;
      for (int dy = 0; dy < fy; dy++) for (int dx = 0; dx < fx; dx++) acc += full[c][(size_t)std::min(y * fy + dy, h - 1) * w + std::min(x * fx + dx, w - 1)];
      pl[c][(size_t)y * pw + x] = acc / (float)(fx * fy);
    }
  }
  // ---- quantisation
  const uint32_t global_scale = (uint32_t)std::min(65535.0f, std::max(1.0f, 4587.0f / p.distance));
  const uint32_t quant_lf = 16;
  const float inv_gs = 65536.0f / (float)global_scale;
  const float m_lf[3] = {1.0f / 4096, 1.0f / 512, 1.0f / 256};
  std::vector<int32_t> hf_mul((size_t)bw * bh);
  for (auto& v : hf_mul) v = 3 + (int)(rng.next() % 6);
  std::vector<float> table[3];
  { const QuantSpec spec = DefaultSpec(0); for (int c = 0; c < 3; c++) ComputeTable(spec, 0, c, table[c]); }
  std::vector<int32_t> qc[3], lfq[3];
  auto quant = [](float v) -> int32_t { const float a = std::fabs(v); if (a < 0.58f) return 0; const int32_t q = (int32_t)(a + 0.5f); return v < 0 ? -q : q; };
  for (int c = 0; c < 3; c++) {
    qc[c].assign((size_t)bwc[c] * bhc[c] * 64, 0);
    lfq[c].assign((size_t)bwc[c] * bhc[c], 0);
    const int pw = bwc[c] * 8;
    for (int by = 0; by < bhc[c]; by++) for (int bx = 0; bx < bwc[c]; bx++) {
      float cf[64], lf = 0;
      ForwardTransform(S_DCT, pl[c].data() + (size_t)by * 8 * pw + bx * 8, pw, cf);
      LFFromLowestFrequencies(S_DCT, cf, &lf, 1);
      lfq[c][(size_t)by * bwc[c] + bx] = (int32_t)std::lrintf(lf / (m_lf[c] * inv_gs / (float)quant_lf));
      // the block's quantisation multiplier is the one of the full-resolution block it starts at
      const float sd = inv_gs / (float)hf_mul[(size_t)(by << vs[c]) * bw + (bx << hs[c])];
      int32_t* q = qc[c].data() + ((size_t)by * bwc[c] + bx) * 64;
      for (int k = 1; k < 64; k++) q[k] = quant(cf[k] / (table[c][k] * sd));
    }
  }
  YCbCrQuantised in;
  in.w = w; in.h = h; for (int c = 0; c < 3; c++) { in.mode[c] = mode[c]; in.lfq[c] = lfq[c]; in.qc[c] = qc[c]; }
  in.hf_mul = hf_mul; in.global_scale = global_scale; in.quant_lf = quant_lf;
  return WriteYCbCrFrame(in, p);
}

// JPEG transcode (what libjxl's lossless JPEG recompression writes, enc_frame.cc with jpeg_data): the JPEG's quantised coefficients as
// they are — DC as the LF image, AC transposed into libjxl's layout — quantiser at unity (global_scale 65536, quant_lf 1, hf_mul 1),
// LF factors qt[0] / (8 * 255) per channel, the JPEG quantisation tables as a RAW table with denominator 1 / (8 * 255).
// planes: per jxl channel (Cb, Y, Cr) the component's blocks x 64 coefficients in JPEG natural (row-major) order; qt: 3 x 64, same order.
static std::vector<uint8_t> EncodeJpegTranscode(int w, int h, const int mode[3], const int16_t* const planes[3], const int32_t* qt, const Params& p, bool gray = false) {
  YCbCrQuantised in;
  in.w = w; in.h = h; in.gray = gray;
  int hs[3], vs[3], bw, bh;
  YCbCrGeometry(w, h, mode, hs, vs, &bw, &bh);
  static std::vector<int32_t> raw;
  raw.assign(3 * 64, 1);
  for (int c = 0; c < 3; c++) {
    in.mode[c] = mode[c];
    const int bwc = bw >> hs[c], bhc = bh >> vs[c];
    in.lfq[c].resize((size_t)bwc * bhc); in.qc[c].assign((size_t)bwc * bhc * 64, 0);
    for (size_t b = 0; b < (size_t)bwc * bhc; b++) {
      const int16_t* src = planes[c] + b * 64;
      in.lfq[c][b] = src[0];
      for (int v = 0; v < 8; v++) for (int u = 0; u < 8; u++) if (u | v) in.qc[c][b * 64 + u * 8 + v] = src[v * 8 + u];
    }
    for (int v = 0; v < 8; v++) for (int u = 0; u < 8; u++) raw[c * 64 + u * 8 + v] = qt[c * 64 + v * 8 + u];
    in.m_lf[c] = (float)qt[c * 64] / 2040.0f;
  }
  in.custom_m_lf = true;
  in.hf_mul.assign((size_t)bw * bh, 1);
  in.global_scale = 65536; in.quant_lf = 1;
  in.raw_table = raw.data(); in.raw_den = 1.0f / 2040.0f;
  return WriteYCbCrFrame(in, p);
}

}  // namespace synth
