// ORACLE — TEST INFRASTRUCTURE ONLY (see bits.h).
// Modular sub-bitstream decoder: restates libjxl v0.11.2 lib/jxl/modular/encoding/{encoding.cc,dec_ma.cc,
// context_predict.h} and lib/jxl/modular/transform/{rct,palette,squeeze}.cc.
// MA tree, properties {0,1,5,9-13,15}, predictors {0,1,5,6}, weighted predictor, RCT and
// plain palettes are [V] (sample.jxl → sample.png, bench.jxl → bench.png bit-exact); the rest is [R].
#pragma once
#include "entropy.h"
#include <array>
#include <cstdlib>

namespace jxlo {

typedef int32_t pixel_t;

struct Channel {
  int w = 0, h = 0;
  int hshift = 0, vshift = 0;
  std::vector<pixel_t> data;
  Channel() {}
  Channel(int w_, int h_, int hs = 0, int vs = 0) : w(w_), h(h_), hshift(hs), vshift(vs), data((size_t)w_ * h_, 0) {}
  pixel_t* row(int y) { return data.data() + (size_t)y * w; }
  const pixel_t* row(int y) const { return data.data() + (size_t)y * w; }
};

struct WPHeader {
  int p1 = 16, p2 = 10, p3a = 7, p3b = 7, p3c = 7, p3d = 0, p3e = 0;
  int w[4] = {13, 12, 12, 12};
};

struct TreeNode {
  int property;  // -1: leaf
  int32_t splitval;
  int lchild, rchild;  // property > splitval ? lchild : rchild
  int predictor;
  int64_t offset;
  uint32_t multiplier;
  int ctx;  // leaf context id
};

struct Tree {
  std::vector<TreeNode> nodes;
  int num_leaves = 0;
  bool uses_wp = false;
  int max_property = 0;
};

enum TransformId { kRCT = 0, kPalette = 1, kSqueeze = 2 };
struct SqueezeParams { bool horizontal, in_place; uint32_t begin_c, num_c; };
struct Transform {
  int id = 0;
  uint32_t begin_c = 0, rct_type = 0;
  uint32_t num_c = 0, nb_colors = 0, nb_deltas = 0, predictor = 0;
  std::vector<SqueezeParams> squeezes;
  bool meta_palette = false;  // palette applied to meta channels (bookkeeping for the inverse)
};

struct ModularImage {
  std::vector<Channel> channel;
  int nb_meta_channels = 0;
  int bitdepth = 8;
  int w = 0, h = 0;
  std::vector<Transform> transforms;
};

// dec_ma.cc DecodeTree (SURVEY B.5 "MA tree" [V])
inline void ReadTree(BitReader& br, Tree& tree, size_t tree_size_limit) {
  EntropyCode ec;
  ReadEntropyCode(br, 6, ec);
  SymbolReader sr;
  sr.Init(&ec, br);
  tree.nodes.clear();
  tree.num_leaves = 0;
  size_t to_decode = 1;
  while (to_decode > 0) {
    if (tree.nodes.size() > tree_size_limit) JXLO_FAIL("MA tree too large");
    to_decode--;
    int property = (int)sr.Read(br, 1) - 1;
    TreeNode n{};
    if (property == -1) {
      n.property = -1;
      n.predictor = (int)sr.Read(br, 2);
      if (n.predictor >= 14) JXLO_FAIL("bad predictor");
      n.offset = UnpackSigned(sr.Read(br, 3));
      uint32_t mul_log = sr.Read(br, 4);
      if (mul_log >= 31) JXLO_FAIL("bad mul_log");
      uint32_t mul_bits = sr.Read(br, 5);
      if (mul_bits + 1 >= (1u << (31 - mul_log))) JXLO_FAIL("bad mul_bits");
      n.multiplier = (mul_bits + 1u) << mul_log;
      n.ctx = tree.num_leaves++;
      if (n.predictor == 6) tree.uses_wp = true;
      tree.nodes.push_back(n);
      continue;
    }
    if (property > 255) JXLO_FAIL("bad tree property");
    n.property = property;
    if (property == 15) tree.uses_wp = true;
    tree.max_property = std::max(tree.max_property, property);
    n.splitval = UnpackSigned(sr.Read(br, 0));
    n.lchild = (int)(tree.nodes.size() + to_decode + 1);
    n.rchild = (int)(tree.nodes.size() + to_decode + 2);
    tree.nodes.push_back(n);
    to_decode += 2;
  }
  if (!sr.CheckFinal()) JXLO_FAIL("tree ANS final state");
}

// context_predict.h weighted predictor (SURVEY B.5 "Weighted predictor [V]")
struct WPState {
  WPHeader hdr;
  int xsize = 0;
  std::vector<int32_t> pred_errors[4];
  std::vector<int32_t> error;
  int64_t prediction[4];
  int64_t pred = 0;
  static uint32_t divlookup(int i) { return (1u << 24) / (uint32_t)(i + 1); }
  void Init(const WPHeader& h, int xs) {
    hdr = h; xsize = xs;
    for (int i = 0; i < 4; i++) pred_errors[i].assign((size_t)(xs + 2) * 2, 0);
    error.assign((size_t)(xs + 2) * 2, 0);
  }
  static inline int FloorLog2u(uint64_t x) { int r = 0; while (x >>= 1) r++; return r; }
  static inline uint32_t ErrorWeight(uint64_t x, uint32_t maxweight) {
    int shift = FloorLog2u(x + 1) - 5;
    if (shift < 0) shift = 0;
    return 4 + ((maxweight * divlookup((int)(x >> shift))) >> shift);
  }
  // returns prediction (with 3 extra bits); *max_err receives property 15
  inline int64_t Predict(int x, int y, int64_t N, int64_t W, int64_t NE, int64_t NW, int64_t NN, int32_t* max_err) {
    size_t cur_row = (y & 1) ? 0 : (size_t)(xsize + 2);
    size_t prev_row = (y & 1) ? (size_t)(xsize + 2) : 0;
    size_t pos_N = prev_row + x;
    size_t pos_NE = x < xsize - 1 ? pos_N + 1 : pos_N;
    size_t pos_NW = x > 0 ? pos_N - 1 : pos_N;
    uint32_t weights[4];
    for (int i = 0; i < 4; i++)
      weights[i] = ErrorWeight((uint64_t)pred_errors[i][pos_N] + pred_errors[i][pos_NE] + pred_errors[i][pos_NW], hdr.w[i]);
    N *= 8; W *= 8; NE *= 8; NW *= 8; NN *= 8;
    int64_t teW = x == 0 ? 0 : error[cur_row + x - 1];
    int64_t teN = error[pos_N];
    int64_t teNW = error[pos_NW];
    int64_t sumWN = teN + teW;
    int64_t teNE = error[pos_NE];
    if (max_err) {
      int64_t p = teW;
      if (std::llabs(teN) > std::llabs(p)) p = teN;
      if (std::llabs(teNW) > std::llabs(p)) p = teNW;
      if (std::llabs(teNE) > std::llabs(p)) p = teNE;
      *max_err = (int32_t)p;
    }
    prediction[0] = W + NE - N;
    prediction[1] = N - (((sumWN + teNE) * hdr.p1) >> 5);
    prediction[2] = W - (((sumWN + teNW) * hdr.p2) >> 5);
    prediction[3] = N - ((teNW * hdr.p3a + teN * hdr.p3b + teNE * hdr.p3c + (NN - N) * hdr.p3d + (NW - W) * hdr.p3e) >> 5);
    // weighted average
    uint32_t wsum = 0;
    for (int i = 0; i < 4; i++) wsum += weights[i];
    int lw = FloorLog2u(wsum);
    wsum = 0;
    for (int i = 0; i < 4; i++) { weights[i] >>= lw - 4; wsum += weights[i]; }
    int64_t sum = (wsum >> 1) - 1;
    for (int i = 0; i < 4; i++) sum += prediction[i] * (int64_t)weights[i];
    pred = (sum * (int64_t)divlookup((int)wsum - 1)) >> 24;
    if (((teN ^ teW) | (teN ^ teNW)) > 0) return pred;
    int64_t mx = std::max(W, std::max(NE, N)), mn = std::min(W, std::min(NE, N));
    pred = std::max(mn, std::min(mx, pred));
    return pred;
  }
  inline void Update(int64_t val, int x, int y) {
    size_t cur_row = (y & 1) ? 0 : (size_t)(xsize + 2);
    size_t prev_row = (y & 1) ? (size_t)(xsize + 2) : 0;
    val *= 8;
    error[cur_row + x] = (int32_t)(pred - val);
    for (int i = 0; i < 4; i++) {
      int32_t err = (int32_t)((std::llabs(prediction[i] - val) + 3) >> 3);
      pred_errors[i][cur_row + x] = err;
      pred_errors[i][prev_row + x + 1] += err;
    }
  }
};

inline int64_t ClampedGradient(int64_t n, int64_t w, int64_t l) {
  int64_t m = std::min(n, w), M = std::max(n, w);
  int64_t grad = n + w - l;
  return l < m ? M : (l > M ? m : grad);   // NW<min → max ; NW>max → min
}

// encoding.cc DecodeModularChannelMAANS (generic path; special-cased fast paths give identical results)
inline void DecodeChannel(BitReader& br, SymbolReader& sr, ModularImage& img, int chan, const Tree& tree,
                          const WPHeader& wph, uint32_t stream_id) {
  Channel& ch = img.channel[chan];
  if (ch.w == 0 || ch.h == 0) return;
  const int w = ch.w, h = ch.h;
  WPState wp;
  if (tree.uses_wp) wp.Init(wph, w);
  // reference channels for properties >= 16
  std::vector<int> refs;
  if (tree.max_property >= 16) {
    for (int j = chan - 1; j >= 0; j--) {
      const Channel& rc = img.channel[j];
      if (rc.w != ch.w || rc.h != ch.h || rc.hshift != ch.hshift || rc.vshift != ch.vshift) continue;
      refs.push_back(j);
    }
  }
  std::vector<int32_t> props(16 + 4 * refs.size() + 4, 0);
  for (int y = 0; y < h; y++) {
    pixel_t* p = ch.row(y);
    const pixel_t* pn = y > 0 ? ch.row(y - 1) : nullptr;
    const pixel_t* pnn = y > 1 ? ch.row(y - 2) : nullptr;
    props[0] = chan; props[1] = (int32_t)stream_id; props[2] = y;
    props[9] = 0;
    for (int x = 0; x < w; x++) {
      int64_t W = x ? p[x - 1] : (y ? pn[x] : 0);
      int64_t N = y ? pn[x] : W;
      int64_t NW = (x && y) ? pn[x - 1] : W;
      int64_t NE = (x + 1 < w && y) ? pn[x + 1] : N;
      int64_t WW = x > 1 ? p[x - 2] : W;
      int64_t NN = y > 1 ? pnn[x] : N;
      int64_t NEE = (x + 2 < w && y) ? pn[x + 2] : NE;
      props[3] = x;
      props[4] = (int32_t)std::llabs(N);
      props[5] = (int32_t)std::llabs(W);
      props[6] = (int32_t)N;
      props[7] = (int32_t)W;
      props[8] = (int32_t)(W - props[9]);
      props[9] = (int32_t)(W + N - NW);
      props[10] = (int32_t)(W - NW);
      props[11] = (int32_t)(NW - N);
      props[12] = (int32_t)(N - NE);
      props[13] = (int32_t)(N - NN);
      props[14] = (int32_t)(W - WW);
      int64_t wp_pred = 0;
      if (tree.uses_wp) wp_pred = wp.Predict(x, y, N, W, NE, NW, NN, &props[15]);
      for (size_t r = 0; r < refs.size(); r++) {
        const Channel& rc = img.channel[refs[r]];
        const pixel_t* rp = rc.row(y);
        int64_t v = rp[x];
        int64_t rl = x ? rp[x - 1] : 0;
        int64_t rt = y ? rc.row(y - 1)[x] : rl;
        int64_t rtl = (x && y) ? rc.row(y - 1)[x - 1] : rl;
        int64_t g = ClampedGradient(rt, rl, rtl);
        props[16 + 4 * r + 0] = (int32_t)std::llabs(v);
        props[16 + 4 * r + 1] = (int32_t)v;
        props[16 + 4 * r + 2] = (int32_t)std::llabs(v - g);
        props[16 + 4 * r + 3] = (int32_t)(v - g);
      }
      // tree walk
      int pos = 0;
      while (tree.nodes[pos].property >= 0) {
        const TreeNode& n = tree.nodes[pos];
        int32_t pv = n.property < (int)props.size() ? props[n.property] : 0;
        pos = pv > n.splitval ? n.lchild : n.rchild;
      }
      const TreeNode& leaf = tree.nodes[pos];
      int64_t guess;
      switch (leaf.predictor) {
        case 0: guess = 0; break;
        case 1: guess = W; break;
        case 2: guess = N; break;
        case 3: guess = (W + N) / 2; break;
        case 4: { int64_t pp = W + N - NW; int64_t pa = std::llabs(pp - W), pb = std::llabs(pp - N); guess = pa < pb ? W : N; } break;
        case 5: guess = ClampedGradient(N, W, NW); break;
        case 6: guess = (wp_pred + 3) >> 3; break;
        case 7: guess = NE; break;
        case 8: guess = NW; break;
        case 9: guess = WW; break;
        case 10: guess = (W + NW) / 2; break;
        case 11: guess = (N + NW) / 2; break;
        case 12: guess = (N + NE) / 2; break;
        case 13: guess = (6 * N - 2 * NN + 7 * W + WW + NEE + 3 * NE + 8) / 16; break;
        default: JXLO_FAIL("bad predictor");
      }
      uint32_t tok = sr.Read(br, leaf.ctx);
      int64_t val = (int64_t)UnpackSigned(tok) * (int64_t)leaf.multiplier + leaf.offset + guess;
      p[x] = (pixel_t)val;
      if (tree.uses_wp) wp.Update(p[x], x, y);
    }
  }
}

// ---- transforms -------------------------------------------------------------------------------------------------
inline void ReadTransform(BitReader& br, Transform& t) {
  t.id = br.u(2);
  if (t.id == 3) JXLO_FAIL("bad transform id");
  if (t.id != kSqueeze) t.begin_c = U32(br, Bits(3), BitsOffset(6, 8), BitsOffset(10, 72), BitsOffset(13, 1096));
  if (t.id == kRCT) {
    t.rct_type = U32(br, Val(6), Bits(2), BitsOffset(4, 2), BitsOffset(6, 10));
    if (t.rct_type >= 42) JXLO_FAIL("bad rct type");
  } else if (t.id == kPalette) {
    t.num_c = U32(br, Val(1), Val(3), Val(4), BitsOffset(13, 1));
    t.nb_colors = U32(br, Bits(8), BitsOffset(10, 256), BitsOffset(12, 1280), BitsOffset(16, 5376));
    t.nb_deltas = U32(br, Val(0), BitsOffset(8, 1), BitsOffset(10, 257), BitsOffset(16, 1281));
    t.predictor = br.u(4);
    if (t.predictor >= 14) JXLO_FAIL("bad palette predictor");
  } else {
    uint32_t num = U32(br, Val(0), BitsOffset(4, 1), BitsOffset(6, 9), BitsOffset(8, 41));
    t.squeezes.resize(num);
    for (auto& s : t.squeezes) {
      s.horizontal = br.Bool();
      s.in_place = br.Bool();
      s.begin_c = U32(br, Bits(3), BitsOffset(6, 8), BitsOffset(10, 72), BitsOffset(13, 1096));
      s.num_c = U32(br, Val(1), Val(2), Val(3), BitsOffset(4, 4));
    }
  }
}

// squeeze.cc DefaultSqueezeParameters [R]
inline void DefaultSqueeze(std::vector<SqueezeParams>& p, const ModularImage& img) {
  int nb = (int)img.channel.size() - img.nb_meta_channels;
  p.clear();
  int w = img.channel[img.nb_meta_channels].w, h = img.channel[img.nb_meta_channels].h;
  if (nb > 2 && img.channel[img.nb_meta_channels + 1].w == w && img.channel[img.nb_meta_channels + 1].h == h) {
    SqueezeParams s{true, false, (uint32_t)img.nb_meta_channels + 1, 2};
    p.push_back(s);
    s.horizontal = false;
    p.push_back(s);
  }
  SqueezeParams s{false, true, (uint32_t)img.nb_meta_channels, (uint32_t)nb};
  bool wide = w > h;
  if (!wide) {
    if (h > 8) { s.horizontal = false; p.push_back(s); h = (h + 1) / 2; }
  }
  while (w > 8 || h > 8) {
    if (w > 8) { s.horizontal = true; p.push_back(s); w = (w + 1) / 2; }
    if (h > 8) { s.horizontal = false; p.push_back(s); h = (h + 1) / 2; }
  }
}

// transform.cc MetaApply — adjusts the channel list before decoding
inline void MetaApply(ModularImage& img, Transform& t) {
  if (t.id == kRCT) {
    if (t.begin_c + 3 > img.channel.size()) JXLO_FAIL("rct out of range");
    return;
  }
  if (t.id == kPalette) {
    uint32_t endc = t.begin_c + t.num_c - 1;
    if (endc >= img.channel.size()) JXLO_FAIL("palette out of range");
    if ((int)t.begin_c < img.nb_meta_channels) {
      if ((int)endc >= img.nb_meta_channels) JXLO_FAIL("palette spans meta and non-meta channels");
      img.nb_meta_channels += 2 - (int)t.num_c;
      t.meta_palette = true;
    } else {
      img.nb_meta_channels += 1;
    }
    img.channel.erase(img.channel.begin() + t.begin_c + 1, img.channel.begin() + endc + 1);
    Channel pch((int)t.nb_colors, (int)t.num_c);
    pch.hshift = -1;
    img.channel.insert(img.channel.begin(), pch);
    return;
  }
  // squeeze
  if (t.squeezes.empty()) DefaultSqueeze(t.squeezes, img);
  for (auto& s : t.squeezes) {
    uint32_t beginc = s.begin_c, endc = s.begin_c + s.num_c - 1;
    if (endc >= img.channel.size()) JXLO_FAIL("squeeze out of range");
    uint32_t offset = s.in_place ? endc + 1 : (uint32_t)img.channel.size();
    if ((int)beginc < img.nb_meta_channels) {
      if (!s.in_place) JXLO_FAIL("squeeze of meta channels must be in place");
      if ((int)endc >= img.nb_meta_channels) JXLO_FAIL("squeeze spans meta");
      img.nb_meta_channels += s.num_c;
    }
    for (uint32_t c = beginc; c <= endc; c++) {
      Channel& ch = img.channel[c];
      int w = ch.w, h = ch.h;
      Channel res;
      if (s.horizontal) {
        ch.w = (w + 1) / 2; ch.hshift++;
        res = Channel(w - (w + 1) / 2, h, ch.hshift, ch.vshift);
      } else {
        ch.h = (h + 1) / 2; ch.vshift++;
        res = Channel(w, h - (h + 1) / 2, ch.hshift, ch.vshift);
      }
      ch.data.assign((size_t)ch.w * ch.h, 0);
      img.channel.insert(img.channel.begin() + offset + (c - beginc), res);
    }
  }
}

// rct.cc InvRCT [V]
inline void InvRCT(ModularImage& img, const Transform& t) {
  uint32_t m = t.begin_c;
  int perm = t.rct_type / 7, kind = t.rct_type % 7;
  Channel& c0 = img.channel[m];
  Channel& c1 = img.channel[m + 1];
  Channel& c2 = img.channel[m + 2];
  JXLO_CHECK(c0.w == c1.w && c0.w == c2.w && c0.h == c1.h && c0.h == c2.h);
  size_t n = (size_t)c0.w * c0.h;
  for (size_t i = 0; i < n; i++) {
    pixel_t a = c0.data[i], b = c1.data[i], c = c2.data[i];
    pixel_t o0, o1, o2;
    if (kind == 6) {  // YCgCo
      pixel_t tmp = (pixel_t)((uint32_t)a - (uint32_t)(c >> 1));
      o1 = (pixel_t)((uint32_t)c + (uint32_t)tmp);              // G
      o2 = (pixel_t)((uint32_t)tmp - (uint32_t)(b >> 1));       // B
      o0 = (pixel_t)((uint32_t)o2 + (uint32_t)b);               // R
    } else {
      pixel_t first = a, second = b, third = c;
      if (kind & 1) third = (pixel_t)((uint32_t)third + (uint32_t)first);
      if ((kind >> 1) == 1) second = (pixel_t)((uint32_t)second + (uint32_t)first);
      else if ((kind >> 1) == 2) second = (pixel_t)((uint32_t)second + (uint32_t)(((int64_t)first + third) >> 1));
      o0 = first; o1 = second; o2 = third;
    }
    pixel_t out[3] = {o0, o1, o2};
    // outputs to perm%3, (perm+1+perm/3)%3, (perm+2-perm/3)%3
    pixel_t res[3];
    res[perm % 3] = out[0];
    res[(perm + 1 + perm / 3) % 3] = out[1];
    res[(perm + 2 - perm / 3) % 3] = out[2];
    c0.data[i] = res[0]; c1.data[i] = res[1]; c2.data[i] = res[2];
  }
}

// palette.h palette lookup incl. delta / implicit entries ([V] plain lookup; the rest [R])
static const int16_t kDeltaPalette[72][3] = {
    {0, 0, 0},       {4, 4, 4},       {11, 0, 0},      {0, 0, -13},     {0, -12, 0},     {-10, -10, -10}, {-18, -18, -18}, {-27, -27, -27},
    {-18, -18, 0},   {0, 0, -32},     {-32, 0, 0},     {-37, -37, -37}, {0, -32, -32},   {24, 24, 45},    {50, 50, 50},    {-45, -24, -24},
    {-24, -45, -45}, {0, -24, -24},   {-34, -34, 0},   {-24, 0, -24},   {-45, -45, -24}, {64, 64, 64},    {-32, 0, -32},   {0, -32, 0},
    {-32, 0, 32},    {-24, -45, -24}, {45, 24, 45},    {24, -24, -45},  {-45, -24, 24},  {80, 80, 80},    {64, 0, 0},      {0, 0, -64},
    {0, -64, -64},   {-24, -24, 45},  {96, 96, 96},    {64, 64, 0},     {45, -24, -24},  {34, -34, 0},    {112, 112, 112}, {24, -45, -45},
    {45, 45, -24},   {0, -32, 32},    {24, -24, 45},   {0, 96, 96},     {45, -24, 24},   {24, -45, -24},  {-24, -45, 24},  {0, -64, 0},
    {96, 0, 0},      {128, 128, 128}, {64, 0, 64},     {144, 144, 144}, {96, 96, 0},     {-36, -36, 36},  {45, -24, -45},  {45, -45, -24},
    {0, 0, -96},     {0, 128, 128},   {0, 96, 0},      {45, 24, -45},   {-128, 0, 0},    {24, -45, 24},   {-45, 24, -45},  {64, 0, -64},
    {64, -64, -64},  {96, 0, 96},     {45, -45, 24},   {24, 45, -45},   {64, 64, -64},   {128, 128, 0},   {0, 0, -128},    {-24, 45, -45}};

inline pixel_t PaletteGetValue(const Channel& pal, int index, int c, int palette_size, int bit_depth) {
  if (index < 0) {
    if (c >= 3) return 0;
    index = -(index + 1);
    index %= 1 + 2 * (72 - 1);
    static const int kMul[2] = {-1, 1};
    pixel_t r = kDeltaPalette[(index + 1) >> 1][c] * kMul[index & 1];
    if (bit_depth > 8) r *= 1 << (bit_depth - 8);
    return r;
  } else if (palette_size <= index && index < palette_size + 64) {
    if (c >= 3) return 0;
    index -= palette_size;
    index >>= c * 2;
    return (pixel_t)(((int64_t)(index % 4) * ((1 << bit_depth) - 1)) / 4 + (1 << std::max(0, bit_depth - 3)));
  } else if (palette_size + 64 <= index) {
    if (c >= 3) return 0;
    index -= palette_size + 64;
    for (int i = 0; i < c; i++) index /= 5;
    return (pixel_t)(((int64_t)(index % 5) * ((1 << bit_depth) - 1)) / 4);
  }
  return pal.row(c)[index];
}

inline void InvPalette(ModularImage& img, const Transform& t, const WPHeader& wph) {
  int nb = (int)t.nb_colors;
  uint32_t c0 = t.begin_c + 1;
  Channel pal = img.channel[0];
  int num_c = (int)t.num_c;
  int w = img.channel[c0].w, h = img.channel[c0].h;
  for (int i = 1; i < num_c; i++) img.channel.insert(img.channel.begin() + c0 + 1, Channel(w, h, img.channel[c0].hshift, img.channel[c0].vshift));
  int bit_depth = std::min(img.bitdepth, 24);
  if (t.nb_deltas == 0 && t.predictor == 0) {
    Channel idx = img.channel[c0];
    for (int c = 0; c < num_c; c++) {
      Channel& out = img.channel[c0 + c];
      for (size_t i = 0; i < (size_t)w * h; i++) {
        out.data[i] = PaletteGetValue(pal, idx.data[i], c, nb, bit_depth);
      }
    }
  } else {
    // delta palette with prediction [R]
    Channel idx = img.channel[c0];
    for (int c = 0; c < num_c; c++) {
      Channel& out = img.channel[c0 + c];
      WPState wp;
      if (t.predictor == 6) wp.Init(wph, w);
      for (int y = 0; y < h; y++) {
        pixel_t* p = out.row(y);
        const pixel_t* pn = y ? out.row(y - 1) : nullptr;
        const pixel_t* pnn = y > 1 ? out.row(y - 2) : nullptr;
        for (int x = 0; x < w; x++) {
          int index = idx.row(y)[x];
          pixel_t val = PaletteGetValue(pal, index, c, nb, bit_depth);
          int64_t W = x ? p[x - 1] : (y ? pn[x] : 0);
          int64_t N = y ? pn[x] : W;
          int64_t NW = (x && y) ? pn[x - 1] : W;
          int64_t NE = (x + 1 < w && y) ? pn[x + 1] : N;
          int64_t WW = x > 1 ? p[x - 2] : W;
          int64_t NN = y > 1 ? pnn[x] : N;
          int64_t NEE = (x + 2 < w && y) ? pn[x + 2] : NE;
          int64_t wp_pred = 0;
          if (t.predictor == 6) wp_pred = wp.Predict(x, y, N, W, NE, NW, NN, nullptr);
          if (index < (int)t.nb_deltas) {
            int64_t guess;
            switch (t.predictor) {
              case 0: guess = 0; break; case 1: guess = W; break; case 2: guess = N; break; case 3: guess = (W + N) / 2; break;
              case 4: { int64_t pp = W + N - NW; guess = std::llabs(pp - W) < std::llabs(pp - N) ? W : N; } break;
              case 5: guess = ClampedGradient(N, W, NW); break;
              case 6: guess = (wp_pred + 3) >> 3; break;
              case 7: guess = NE; break; case 8: guess = NW; break; case 9: guess = WW; break;
              case 10: guess = (W + NW) / 2; break; case 11: guess = (N + NW) / 2; break; case 12: guess = (N + NE) / 2; break;
              default: guess = (6 * N - 2 * NN + 7 * W + WW + NEE + 3 * NE + 8) / 16; break;
            }
            val = (pixel_t)(val + guess);
          }
          p[x] = val;
          if (t.predictor == 6) wp.Update(p[x], x, y);
        }
      }
    }
  }
  img.channel.erase(img.channel.begin());
  // nb_meta_channels bookkeeping is done by the caller (mirrors MetaApply)
}

// squeeze.cc SmoothTendency / InvHSqueeze / InvVSqueeze [R]
inline int64_t SmoothTendency(int64_t B, int64_t a, int64_t n) {
  int64_t diff = 0;
  if (B >= a && a >= n) {
    diff = (4 * B - 3 * n - a + 6) / 12;
    if (diff - (diff & 1) > 2 * (B - a)) diff = 2 * (B - a) + 1;
    if (diff + (diff & 1) > 2 * (a - n)) diff = 2 * (a - n);
  } else if (B <= a && a <= n) {
    diff = (4 * B - 3 * n - a - 6) / 12;
    if (diff + (diff & 1) < 2 * (B - a)) diff = 2 * (B - a) - 1;
    if (diff - (diff & 1) < 2 * (a - n)) diff = 2 * (a - n);
  }
  return diff;
}

inline void InvSqueeze(ModularImage& img, const Transform& t) {
  for (int i = (int)t.squeezes.size() - 1; i >= 0; i--) {
    const SqueezeParams& s = t.squeezes[i];
    uint32_t beginc = s.begin_c, endc = s.begin_c + s.num_c - 1;
    uint32_t offset = s.in_place ? endc + 1 : (uint32_t)(img.channel.size() + beginc - endc - 1);
    if ((int)beginc < img.nb_meta_channels) img.nb_meta_channels -= s.num_c;
    for (uint32_t c = beginc; c <= endc; c++) {
      uint32_t rc = offset + c - beginc;
      Channel& avg = img.channel[c];
      Channel& res = img.channel[rc];
      if (s.horizontal) {
        JXLO_CHECK(avg.h == res.h);
        Channel out(avg.w + res.w, avg.h, avg.hshift - 1, avg.vshift);
        for (int y = 0; y < avg.h; y++) {
          const pixel_t* pa = avg.row(y);
          const pixel_t* pr = res.row(y);
          pixel_t* po = out.row(y);
          for (int x = 0; x < res.w; x++) {
            int64_t dmt = pr[x];
            int64_t a = pa[x];
            int64_t next_avg = x + 1 < avg.w ? pa[x + 1] : a;
            int64_t left = x ? po[2 * x - 1] : a;
            int64_t tendency = SmoothTendency(left, a, next_avg);
            int64_t diff = dmt + tendency;
            int64_t A = ((a * 2) + diff + (diff > 0 ? -(diff & 1) : (diff & 1))) >> 1;
            po[2 * x] = (pixel_t)A;
            po[2 * x + 1] = (pixel_t)(A - diff);
          }
          if (avg.w > res.w) po[2 * res.w] = pa[res.w];
        }
        img.channel[c] = out;
      } else {
        JXLO_CHECK(avg.w == res.w);
        Channel out(avg.w, avg.h + res.h, avg.hshift, avg.vshift - 1);
        for (int y = 0; y < res.h; y++) {
          const pixel_t* pa = avg.row(y);
          const pixel_t* pna = y + 1 < avg.h ? avg.row(y + 1) : pa;
          const pixel_t* pr = res.row(y);
          pixel_t* po0 = out.row(2 * y);
          pixel_t* po1 = out.row(2 * y + 1);
          const pixel_t* ptop = y ? out.row(2 * y - 1) : pa;
          for (int x = 0; x < avg.w; x++) {
            int64_t dmt = pr[x];
            int64_t a = pa[x];
            int64_t next_avg = pna[x];
            int64_t top = ptop[x];
            int64_t tendency = SmoothTendency(top, a, next_avg);
            int64_t diff = dmt + tendency;
            int64_t A = ((a * 2) + diff + (diff > 0 ? -(diff & 1) : (diff & 1))) >> 1;
            po0[x] = (pixel_t)A;
            po1[x] = (pixel_t)(A - diff);
          }
        }
        if (avg.h > res.h) {
          const pixel_t* pa = avg.row(res.h);
          pixel_t* po = out.row(2 * res.h);
          for (int x = 0; x < avg.w; x++) po[x] = pa[x];
        }
        img.channel[c] = out;
      }
    }
    img.channel.erase(img.channel.begin() + offset, img.channel.begin() + offset + (endc - beginc + 1));
  }
}

inline void UndoTransforms(ModularImage& img, const WPHeader& wph) {
  for (int i = (int)img.transforms.size() - 1; i >= 0; i--) {
    const Transform& t = img.transforms[i];
    if (t.id == kRCT) InvRCT(img, t);
    else if (t.id == kPalette) {
      InvPalette(img, t, wph);
      img.nb_meta_channels -= t.meta_palette ? 2 - (int)t.num_c : 1;
    } else InvSqueeze(img, t);
  }
  img.transforms.clear();
}

// encoding.cc ModularDecode: GroupHeader + (tree) + channels
struct GroupHeader {
  bool use_global_tree = false;
  WPHeader wp;
  std::vector<Transform> transforms;
};

inline void ReadGroupHeader(BitReader& br, GroupHeader& gh) {
  gh.use_global_tree = br.Bool();
  bool wp_default = br.Bool();
  if (!wp_default) {
    gh.wp.p1 = br.u(5); gh.wp.p2 = br.u(5);
    gh.wp.p3a = br.u(5); gh.wp.p3b = br.u(5); gh.wp.p3c = br.u(5); gh.wp.p3d = br.u(5); gh.wp.p3e = br.u(5);
    for (int i = 0; i < 4; i++) gh.wp.w[i] = br.u(4);
  }
  uint32_t nb = U32(br, Val(0), Val(1), BitsOffset(4, 2), BitsOffset(8, 18));
  gh.transforms.resize(nb);
  for (auto& t : gh.transforms) ReadTransform(br, t);
}

struct GlobalTree {
  bool present = false;
  Tree tree;
  EntropyCode code;
};

// Decodes one modular sub-stream into img (whose channels are pre-sized). max_chan_size: channels (after meta)
// larger than this stop the decode (GlobalModular); <=0 means decode all. Returns number of channels decoded.
// If undo is true, transforms are undone after decoding.
struct ModularDecodeResult { size_t first_undecoded = 0; WPHeader wp; };

inline ModularDecodeResult ModularDecode(BitReader& br, ModularImage& img, uint32_t stream_id, const GlobalTree* global,
                                         int max_chan_size, bool undo, size_t* tokens = nullptr) {
  ModularDecodeResult res;
  if (img.channel.empty()) return res;
  GroupHeader gh;
  ReadGroupHeader(br, gh);
  res.wp = gh.wp;
  for (auto& t : gh.transforms) { MetaApply(img, t); img.transforms.push_back(t); }
  size_t nb_channels = img.channel.size();
  Tree local_tree;
  EntropyCode local_code;
  const Tree* tree;
  const EntropyCode* code;
  if (!gh.use_global_tree) {
    size_t npix = 0;
    for (auto& c : img.channel) npix += (size_t)c.w * c.h;
    size_t limit = std::min<size_t>(1 << 22, 1024 + npix);
    ReadTree(br, local_tree, limit);
    ReadEntropyCode(br, local_tree.num_leaves, local_code);
    tree = &local_tree; code = &local_code;
  } else {
    if (!global || !global->present) JXLO_FAIL("global tree requested but absent");
    tree = &global->tree; code = &global->code;
  }
  // distance multiplier = max channel width among decoded channels
  uint32_t dist_mult = 0;
  size_t end = nb_channels;
  for (size_t i = 0; i < nb_channels; i++) {
    const Channel& c = img.channel[i];
    if ((int)i >= img.nb_meta_channels && max_chan_size > 0 && (c.w > max_chan_size || c.h > max_chan_size)) { end = i; break; }
    dist_mult = std::max<uint32_t>(dist_mult, c.w);
  }
  SymbolReader sr;
  sr.Init(code, br, dist_mult);
  for (size_t i = 0; i < end; i++) DecodeChannel(br, sr, img, (int)i, *tree, gh.wp, stream_id);
  if (!sr.CheckFinal()) JXLO_FAIL("modular stream ANS final state");
  if (tokens) *tokens += sr.tokens;
  res.first_undecoded = end;
  if (undo) UndoTransforms(img, gh.wp);
  return res;
}

}  // namespace jxlo
