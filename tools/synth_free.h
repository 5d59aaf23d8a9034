// jxlsynth — "free-running" Modular streams (fixture generator; NOT on the product decode path, independent of oracle/).
// The fixed-tree encoder in jxl_synth.cc predicts like a decoder and therefore only speaks the few tree shapes it can
// simulate.  This one does not simulate anything: every context of a stream maps to ONE histogram, so a token's code does
// not depend on the context the decoder will derive for it, and any sequence of residual tokens is a valid stream whatever
// the MA tree, the predictors or the properties are.  The decoded image is whatever the decoder makes of it — exactly what a
// decoder-vs-decoder parity test needs to reach features no real encoder run is available for here:
//   random MA trees over all properties (incl. the weighted predictor's error, property 15, and the previous-channel
//   properties >= 16), all 14 predictors, leaf offsets and multipliers, custom weighted-predictor parameters,
//   local trees (per-section tree + code instead of the global one), LZ77 copies (incl. the special-distance table),
//   palettes with delta entries and a predictor.
// Channels whose values must be meaningful (the palette itself, the index channel) sit under a Zero-predictor leaf, where the
// decoded value is the token.
#pragma once

namespace synth {

struct FreeParams {
  uint32_t seed = 1;
  int w = 64, h = 64, nchan = 3, has_alpha = 0, bits = 8;
  int tree_flags = 0;     // 1: weighted predictor (predictor 6, property 15)  2: properties >= 16  4: leaf offsets / multipliers
                          // 8: all 14 predictors (else zero / W / gradient)  16: custom WP header
  int tree_depth = 5;
  int local_trees = 0;    // 1: every section stream carries its own tree + code; 2: the global stream too (no global tree at all)
  int lz77 = 0;
  int palette = 0;        // 1: global palette over the colour channels
  int nb_colors = 16, nb_deltas = 0, pal_pred = 0;
};

struct FreeChan { int w, h; int kind; };   // kind 0: free-running, 1: palette entries, 2: palette indices

namespace free_detail {

struct Range { int64_t lo, hi; };

struct TreeGen {
  Pcg32 rng;
  const FreeParams& p;
  GTree t;
  TreeGen(const FreeParams& pp, uint32_t seed) : rng(seed), p(pp) {}
  int RandPred() {
    if (p.tree_flags & 8) { int k = (int)(rng.next() % 14); if (k == 6 && !(p.tree_flags & 1)) k = 5; return k; }
    static const int base[4] = {0, 1, 5, 5};
    int k = base[rng.next() % 4];
    if ((p.tree_flags & 1) && rng.next() % 3 == 0) k = 6;
    return k;
  }
  int Leaf(int pred) {
    int id = t.add_leaf(pred);
    if ((p.tree_flags & 4) && pred != 0) {
      t.nodes[id].off = (int)(rng.next() % 7) - 3;
      t.nodes[id].mul_log = (int)(rng.next() % 2);
      t.nodes[id].mul_bits = (int)(rng.next() % 3);
    }
    return id;
  }
  // random subtree; `ranges` = the interval each property is known to lie in on this path (dec_ma.cc rejects splits outside it)
  int Random(int depth, std::vector<Range> ranges, int w, int h) {
    if (depth <= 0 || (depth < p.tree_depth && rng.next() % 5 == 0)) return Leaf(RandPred());   // (never a leaf at the root of a random subtree)
    std::vector<int> props = {2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14};
    if (p.tree_flags & 1) { props.push_back(15); props.push_back(15); }
    if (p.tree_flags & 2) for (int k = 16; k < 24; k++) props.push_back(k);
    for (int attempt = 0; attempt < 8; attempt++) {
      const int prop = props[rng.next() % props.size()];
      int64_t centre = 0, spread = 12;
      if (prop == 2) { centre = h / 2; spread = h / 2 + 1; }
      else if (prop == 3) { centre = w / 2; spread = w / 2 + 1; }
      else if (prop == 4 || prop == 5 || prop == 16 || prop == 18 || prop == 20 || prop == 22) { centre = 10; spread = 10; }
      const int64_t lo = std::max(ranges[prop].lo, centre - spread), hi = std::min(ranges[prop].hi - 1, centre + spread);
      if (lo > hi) continue;
      const int64_t val = lo + (int64_t)(rng.next() % (uint32_t)(hi - lo + 1));
      std::vector<Range> left = ranges, right = ranges;
      left[prop].lo = val + 1; right[prop].hi = val;
      const int l = Random(depth - 1, left, w, h), r = Random(depth - 1, right, w, h);
      return t.add_inner(prop, (int)val, l, r);
    }
    return Leaf(RandPred());
  }
};

inline std::vector<Range> FullRanges() { return std::vector<Range>(64, Range{-2147483648LL, 2147483647LL}); }

// Tree for one stream, or for all of them when global.  The controlled channels of the global stream are a prefix of its
// channel list (palette entries, then the index channel if it fits a group): one static split on the channel index sends them
// to a Zero-predictor leaf.  Section streams get two random subtrees split on the channel index (so that pruning by static
// properties has something to do); with a palette whose index channel lives in the sections, their channel 0 is controlled.
inline int BuildTree(TreeGen& g, const std::vector<FreeChan>& global_chans, int n_group_chans, bool group_first_controlled,
                     bool for_global, bool for_groups, int w, int h) {
  const auto ranges = FullRanges();
  int groups_root = -1, global_root = -1;
  if (for_groups) {
    const int a = g.Random(g.p.tree_depth, ranges, 256, 256);
    const int b = group_first_controlled ? g.t.add_leaf(0) : g.Random(g.p.tree_depth, ranges, 256, 256);
    groups_root = n_group_chans > 1 ? g.t.add_inner(0, 0, a, b) : b;      // channel > 0 ? a : b
  }
  if (for_global) {
    int ncontrolled = 0;
    while (ncontrolled < (int)global_chans.size() && global_chans[ncontrolled].kind != 0) ncontrolled++;
    if (ncontrolled == (int)global_chans.size() && ncontrolled > 0) global_root = g.t.add_leaf(0);
    else {
      const int f = g.Random(g.p.tree_depth, ranges, w, h);
      global_root = ncontrolled ? g.t.add_inner(0, ncontrolled - 1, f, g.t.add_leaf(0)) : f;
    }
  }
  if (for_global && for_groups) return g.t.add_inner(1, 0, groups_root, global_root);   // stream id > 0 ? sections : global
  return for_global ? global_root : groups_root;
}

inline void FinishTree(GTree& t, int root, std::vector<int>* bfs) {
  bfs->assign(1, root);
  for (size_t i = 0; i < bfs->size(); i++) { const TNode& n = t.nodes[(*bfs)[i]]; if (n.prop >= 0) { bfs->push_back(n.l); bfs->push_back(n.r); } }
  int leaf = 0;
  for (int id : *bfs) if (t.nodes[id].prop < 0) t.nodes[id].ctx = leaf++;
  t.num_leaves = leaf;
}

// Token sequence of one stream: `count` symbols in decode order.  value_of(i) gives the controlled value of symbol i or
// INT32_MIN for a free one.
template <typename F>
inline void StreamTokens(Pcg32& rng, size_t count, bool lz77, const EntropyCoder& proto, uint32_t dist_ctx, F value_of, std::vector<Token>& out) {
  size_t i = 0;
  while (i < count) {
    if (lz77 && i > 0 && rng.next() % 9 == 0) {
      uint32_t len = proto.lz_min_length + (rng.next() % 4 == 0 ? rng.next() % 300 : rng.next() % 12);
      len = (uint32_t)std::min<size_t>(len, count - i);
      if (len >= proto.lz_min_length) {
        Token t; t.ctx = 0; t.raw = 1;
        uint32_t tok, nb, bits;
        EncodeHybrid(proto.lz_len_cfg, len - proto.lz_min_length, &tok, &nb, &bits);
        t.value = proto.lz_min_symbol + tok; t.nb = (uint8_t)nb; t.bits = bits;
        out.push_back(t);
        // distance: values < 120 go through the special-distance table (needs a non-zero multiplier = widest channel), larger
        // ones are distance + 119
        const uint32_t dv = rng.next() % 3 == 0 ? rng.next() % 120 : 120 + rng.next() % (uint32_t)std::min<size_t>(i + 40, 5000);
        out.push_back(Token{dist_ctx, dv});
        i += len;
        continue;
      }
    }
    const int32_t v = value_of(i);
    if (v != INT32_MIN) out.push_back(Token{0, PackSigned(v)});
    else {
      // skewed small residuals, now and then a large one (hybrid-uint extra bits)
      const uint32_t r = rng.next();
      int32_t d = (r & 3) == 0 ? 0 : (int32_t)((r >> 2) % 5) - 2;
      if ((r >> 8) % 37 == 0) d = (int32_t)((r >> 14) % 600) - 300;
      out.push_back(Token{0, PackSigned(d)});
    }
    i++;
  }
}

}  // namespace free_detail

static std::vector<uint8_t> EncodeModularFree(const FreeParams& fp) {
  using namespace free_detail;
  const int gd = 256, lfd = gd * 8, w = fp.w, h = fp.h;
  const int xg = (w + gd - 1) / gd, yg = (h + gd - 1) / gd, ngroups = xg * yg;
  const int xlg = (w + lfd - 1) / lfd, ylg = (h + lfd - 1) / lfd, nlf = xlg * ylg;
  const int ncolor = fp.nchan, ntot = ncolor + (fp.has_alpha ? 1 : 0);
  Pcg32 rng(fp.seed * 2654435761u + 17);
  // channel list after the (forward) transforms, as the decoder's MetaApply will build it
  std::vector<FreeChan> chans;
  int nmeta = 0;
  if (fp.palette) {
    chans.push_back({fp.nb_colors, ncolor, 1}); nmeta = 1;
    chans.push_back({w, h, 2});
    for (int c = ncolor; c < ntot; c++) chans.push_back({w, h, 0});
  } else for (int c = 0; c < ntot; c++) chans.push_back({w, h, 0});
  int nglobal = 0;
  while (nglobal < (int)chans.size() && (nglobal < nmeta || (chans[nglobal].w <= gd && chans[nglobal].h <= gd))) nglobal++;
  std::vector<FreeChan> gch(chans.begin(), chans.begin() + nglobal);
  const int n_group_chans = (int)chans.size() - nglobal;
  const bool single = ngroups == 1;
  const bool any_group_stream = n_group_chans > 0;
  const bool first_is_index = fp.palette && nglobal == nmeta;     // the index channel is the first channel of every section stream

  EntropyCoder proto;
  proto.lz77 = fp.lz77 != 0; proto.lz_min_symbol = 224; proto.lz_min_length = 3; proto.lz_len_cfg = UintConfig{3, 0, 0};

  // palette content and index values (controlled)
  std::vector<int32_t> pal_vals, idx_vals;
  if (fp.palette) {
    pal_vals.resize((size_t)fp.nb_colors * ncolor);
    for (auto& v : pal_vals) v = (int32_t)(rng.next() % (1u << fp.bits));
    for (int i = 0; i < fp.nb_deltas && i < fp.nb_colors; i++) for (int c = 0; c < ncolor; c++) pal_vals[(size_t)c * fp.nb_colors + i] = (int32_t)(rng.next() % 9) - 4;
    idx_vals.resize((size_t)w * h);
    for (auto& v : idx_vals) {
      const uint32_t r = rng.next() % 100;
      if (r < 70) v = (int32_t)(rng.next() % (uint32_t)fp.nb_colors);
      else if (r < 80) v = -(int32_t)(1 + rng.next() % 150);                              // implicit delta palette (only when nb_deltas > 0: adds the prediction)
      else if (r < 90) v = fp.nb_colors + (int32_t)(rng.next() % 64);                     // implicit 4x4x4 cube
      else v = fp.nb_colors + 64 + (int32_t)(rng.next() % 125);                           // implicit 5x5x5 cube
      if (v < 0 && fp.nb_deltas == 0 && fp.pal_pred == 0 && rng.next() % 2) v = 0;
    }
  }

  struct Stream { GTree tree; std::vector<int> bfs; int root = 0; std::vector<Token> tok; bool present = false; bool local = false; };
  Stream global;
  std::vector<Stream> groups(ngroups);
  const bool global_tree_exists = fp.local_trees < 2;
  GTree gtree; std::vector<int> gbfs; int groot = 0;
  if (global_tree_exists) {
    TreeGen g(fp, fp.seed * 977 + 5);
    groot = BuildTree(g, gch, n_group_chans, first_is_index, true, any_group_stream && fp.local_trees == 0, w, h);
    gtree = g.t;
    FinishTree(gtree, groot, &gbfs);
  }
  auto local_tree = [&](Stream& s, uint32_t seed, bool is_global) {
    TreeGen g(fp, seed);
    s.root = BuildTree(g, gch, n_group_chans, first_is_index, is_global, !is_global, w, h);
    s.tree = g.t;
    FinishTree(s.tree, s.root, &s.bfs);
    s.local = true;
  };
  if (!global_tree_exists) local_tree(global, fp.seed * 31 + 1, true);
  // token streams
  {
    size_t count = 0;
    std::vector<size_t> start;
    for (auto& c : gch) { start.push_back(count); count += (size_t)c.w * c.h; }
    const uint32_t dist_ctx = (uint32_t)((global.local ? global.tree.num_leaves : gtree.num_leaves));
    StreamTokens(rng, count, fp.lz77 != 0, proto, dist_ctx, [&](size_t i) -> int32_t {
      for (size_t c = gch.size(); c-- > 0;) if (i >= start[c]) {
        if (gch[c].kind == 1) return pal_vals[i - start[c]];
        if (gch[c].kind == 2) return idx_vals[i - start[c]];
        return INT32_MIN;
      }
      return INT32_MIN;
    }, global.tok);
    global.present = true;
  }
  for (int g = 0; g < ngroups && any_group_stream; g++) {
    Stream& s = groups[g];
    const int x0 = (g % xg) * gd, y0 = (g / xg) * gd;
    const int rw = std::min(gd, w - x0), rh = std::min(gd, h - y0);
    if (fp.local_trees >= 1) local_tree(s, fp.seed * 131 + 7 * g + 3, false);
    const size_t per = (size_t)rw * rh, count = per * n_group_chans;
    const uint32_t dist_ctx = (uint32_t)(s.local ? s.tree.num_leaves : gtree.num_leaves);
    StreamTokens(rng, count, fp.lz77 != 0, proto, dist_ctx, [&](size_t i) -> int32_t {
      if (first_is_index && i < per) return idx_vals[(size_t)(y0 + i / rw) * w + x0 + i % rw];
      return INT32_MIN;
    }, s.tok);
    s.present = true;
  }
  // entropy codes: one cluster each; the global code covers every stream that uses the global tree
  auto make_code = [&](const std::vector<const std::vector<Token>*>& ss, int leaves, EntropyCoder& ec) {
    BuildEntropyCoder(ss, leaves + (fp.lz77 ? 1 : 0), UintConfig{4, 1, 0}, 1, ec);
    ec.lz77 = proto.lz77; ec.lz_min_symbol = proto.lz_min_symbol; ec.lz_min_length = proto.lz_min_length; ec.lz_len_cfg = proto.lz_len_cfg;
  };
  auto write_tree_and_code = [&](BitWriter& s, const GTree& t, const std::vector<int>& bfs, const EntropyCoder& code) {
    std::vector<Token> tt;
    TreeTokens(t, bfs, tt);
    EntropyCoder tc;
    { std::vector<const std::vector<Token>*> ss{&tt}; BuildEntropyCoder(ss, 6, UintConfig{4, 2, 0}, 6, tc); }
    WriteEntropyCode(s, tc);
    EncodeTokens(s, tc, tt);
    WriteEntropyCode(s, code);
  };
  auto write_wp = [&](BitWriter& s, Pcg32& r) {
    if (!(fp.tree_flags & 16)) { s.put(1, 1); return; }
    s.put(0, 1);
    for (int i = 0; i < 7; i++) s.put(r.next() % 32, 5);     // p1C, p2C, p3Ca..p3Ce
    for (int i = 0; i < 4; i++) s.put(r.next() % 16, 4);     // w0..w3
  };
  EntropyCoder gcode;
  if (global_tree_exists) {
    std::vector<const std::vector<Token>*> ss;
    if (!global.local) ss.push_back(&global.tok);
    for (auto& g : groups) if (g.present && !g.local) ss.push_back(&g.tok);
    make_code(ss, gtree.num_leaves, gcode);
  }
  std::vector<BitWriter> sections;
  {
    BitWriter s;
    s.put(1, 1);  // LfChannelDequantization default
    s.put(global_tree_exists ? 1 : 0, 1);
    if (global_tree_exists) write_tree_and_code(s, gtree, gbfs, gcode);
    s.put(global.local ? 0 : 1, 1);   // use_global_tree
    write_wp(s, rng);
    WriteU32(s, fp.palette ? 1 : 0, {0, 0}, {0, 1}, {4, 2}, {8, 18});
    if (fp.palette) {
      s.put(1, 2);
      WriteU32(s, 0, {3, 0}, {6, 8}, {10, 72}, {13, 1096});                                  // begin_c
      WriteU32(s, (uint32_t)ncolor, {0, 1}, {0, 3}, {0, 4}, {13, 1});                        // num_c
      WriteU32(s, (uint32_t)fp.nb_colors, {8, 0}, {10, 256}, {12, 1280}, {16, 5376});
      WriteU32(s, (uint32_t)fp.nb_deltas, {0, 0}, {8, 1}, {10, 257}, {16, 1281});
      s.put((uint32_t)fp.pal_pred, 4);
    }
    if (global.local) {
      EntropyCoder lc;
      std::vector<const std::vector<Token>*> ss{&global.tok};
      make_code(ss, global.tree.num_leaves, lc);
      write_tree_and_code(s, global.tree, global.bfs, lc);
      EncodeTokens(s, lc, global.tok);
    } else EncodeTokens(s, gcode, global.tok);
    sections.push_back(s);
  }
  for (int g = 0; g < nlf; g++) sections.push_back(BitWriter());   // no squeezed channels: ModularLfGroup is empty
  sections.push_back(BitWriter());                                  // HfGlobal slot
  for (int g = 0; g < ngroups; g++) {
    BitWriter s;
    Stream& st = groups[g];
    if (st.present) {
      s.put(st.local ? 0 : 1, 1);
      write_wp(s, rng);
      s.put(0, 2);   // no local transforms
      if (st.local) {
        EntropyCoder lc;
        std::vector<const std::vector<Token>*> ss{&st.tok};
        make_code(ss, st.tree.num_leaves, lc);
        write_tree_and_code(s, st.tree, st.bfs, lc);
        EncodeTokens(s, lc, st.tok);
      } else EncodeTokens(s, gcode, st.tok);
    }
    sections.push_back(s);
  }
  BitWriter out;
  Params p;
  p.out_bits = fp.bits; p.gab = 0; p.epf_iters = 0; p.noise = 0; p.upsampling = 1; p.num_passes = 1; p.skip_lf_smoothing = 0;
  WriteImageHeader(out, w, h, p, false, fp.bits, fp.has_alpha != 0, ncolor == 1);
  WriteFrameHeader(out, p, true, false, fp.has_alpha ? 1 : 0, 1, false, w, h);
  WriteTOCAndSections(out, sections, single);
  out.align();
  return out.bytes;
}

}  // namespace synth
