// jxlsynth — deterministic JPEG XL bit-stream synthesiser.
// Produces conformant VarDCT (XYB, ANS, variable block sizes, gaborish/EPF flags) and Modular-lossless streams from
// seeded synthetic images, because the reference ships no `cjxl -d 1` fixture and no encoder exists in this image.
// Fixture/bench input generator only: NOT on the product decode path, independent of oracle/.
#include "synth_dct.h"
#include <map>

namespace synth {

// ---- PRNG + synthetic image (SURVEY §8d "Concrete synthetic inputs") ------------------------------------------------
struct Pcg32 {
  uint64_t state, inc;
  explicit Pcg32(uint64_t seed, uint64_t seq = 54) { state = 0; inc = (seq << 1) | 1; next(); state += seed; next(); }
  uint32_t next() {
    uint64_t old = state;
    state = old * 6364136223846793005ULL + inc;
    uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u), rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((32 - rot) & 31));
  }
  float uniform() { return (next() >> 8) * (1.0f / 16777216.0f); }
  float gauss() { float u1 = std::max(uniform(), 1e-7f), u2 = uniform(); return std::sqrt(-2.0f * std::log(u1)) * std::cos(6.2831853f * u2); }
};

static void SyntheticImage(uint32_t seed, int w, int h, uint8_t* rgb) {
  Pcg32 r1(seed * 4 + 1), r2(seed * 4 + 2), r3(seed * 4 + 3);
  struct Cosine { float fx, fy, ph, amp[3]; } cs[8];
  for (auto& c : cs) {
    c.fx = (r1.uniform() * 2 - 1) * 6.0f / w * 6.2831853f; c.fy = (r1.uniform() * 2 - 1) * 6.0f / h * 6.2831853f;
    c.ph = r1.uniform() * 6.2831853f;
    for (float& a : c.amp) a = (r1.uniform() * 2 - 1) * 0.12f;
  }
  struct Rect { float x0, y0, x1, y1, soft, col[3], alpha; } rs[64];
  for (auto& r : rs) {
    float cx = r2.uniform() * w, cy = r2.uniform() * h, rw = (0.02f + r2.uniform() * 0.2f) * w, rh = (0.02f + r2.uniform() * 0.2f) * h;
    r.x0 = cx - rw / 2; r.x1 = cx + rw / 2; r.y0 = cy - rh / 2; r.y1 = cy + rh / 2;
    r.soft = 0.5f + r2.uniform() * 6.0f;
    for (float& c : r.col) c = r2.uniform();
    r.alpha = 0.3f + 0.7f * r2.uniform();
  }
  std::vector<float> fx(8 * w);
  for (int y = 0; y < h; y++) {
    for (int x = 0; x < w; x++) {
      float v[3] = {0.5f, 0.5f, 0.5f};
      for (auto& c : cs) { float s = std::cos(c.fx * x + c.fy * y + c.ph); for (int k = 0; k < 3; k++) v[k] += c.amp[k] * s; }
      for (auto& r : rs) {
        if (x < r.x0 - 3 * r.soft || x > r.x1 + 3 * r.soft || y < r.y0 - 3 * r.soft || y > r.y1 + 3 * r.soft) continue;
        float dx = std::min(x - r.x0, r.x1 - x), dy = std::min(y - r.y0, r.y1 - y);
        float d = std::min(dx, dy) / r.soft;
        float a = r.alpha / (1.0f + std::exp(-2.0f * d));
        for (int k = 0; k < 3; k++) v[k] = v[k] * (1 - a) + r.col[k] * a;
      }
      for (int k = 0; k < 3; k++) {
        float t = v[k] + r3.gauss() * (2.0f / 255.0f);
        int q = (int)std::lrintf(std::min(1.0f, std::max(0.0f, t)) * 255.0f);
        rgb[((size_t)y * w + x) * 3 + k] = (uint8_t)q;
      }
    }
  }
}

// ---- colour ----------------------------------------------------------------------------------------------------
static inline float SrgbToLinear(float v) { return v <= 0.04045f ? v / 12.92f : std::pow((v + 0.055f) / 1.055f, 2.4f); }
static void LinearToXYB(float r, float g, float b, float* X, float* Y, float* B) {
  const float bias = 0.0037930732552754493f;
  float mr = 0.30f * r + 0.622f * g + 0.078f * b + bias;
  float mg = 0.23f * r + 0.692f * g + 0.078f * b + bias;
  float mb = 0.24342268924547819f * r + 0.20476744424496821f * g + 0.55180986650955360f * b + bias;
  float cb = std::cbrt(bias);
  float gr = std::cbrt(std::max(0.0f, mr)) - cb, gg = std::cbrt(std::max(0.0f, mg)) - cb, gb = std::cbrt(std::max(0.0f, mb)) - cb;
  *X = 0.5f * (gr - gg); *Y = 0.5f * (gr + gg); *B = gb;
}

struct Params {
  uint32_t seed = 1;
  float distance = 1.0f;
  int epf_iters = 1;
  int gab = 1;
  int strategy_mix = 1;   // 0 DCT8 only, 1 SURVEY mix, 2 + large blocks / exotic small ones, 3 = 2 unaligned, 4 = 2 + DCT128/256, 5 = 4 unaligned
  int out_bits = 8;       // 8 | 16 | 32 (float, linear, intensity_target 1000 when hdr)
  int hdr = 0;
  int skip_lf_smoothing = 0;
  int custom_orders = 0;  // reserved
  int orientation = 1;    // EXIF-style 1..8, written to the image header
  int upsampling = 1;     // 1 | 2 | 4 | 8: the frame is coded at 1/upsampling of the image size
  int custom_up_weights = 0;  // 1: the image header carries explicit upsampling weights (required for 4x / 8x here)
  int num_passes = 1;     // 1..3: coefficients split into bit planes (pass p carries value >> shift[p], the last pass the remainder)
  int pass_ds = 0;        // VarDCT, num_passes > 1: the frame header lists every pass but the last as the last pass of a downsampling ratio (4, 2 / 2): the kLastPasses progression steps
  int permute_toc = 0;    // != 0: sections stored in a shuffled order (seed), TOC carries the permutation
  int reserved[3] = {0};
  // ---- frame control (multi-frame streams are assembled by concatenating the pieces): noise, frame type, crop, blending, slots
  int noise = 0; uint32_t noise_lut[8] = {0};     // flag kNoise + 8 x u(10)
  int frame_type = 0;                             // 0 regular, 2 reference only, 3 skip progressive
  int have_crop = 0, crop_x0 = 0, crop_y0 = 0;    // frame size = the planes' size; image size = canvas_w x canvas_h
  int canvas_w = 0, canvas_h = 0;                 // image size when it differs from the frame (crop) — 0: same
  int blend_mode = 0, blend_source = 0, blend_clamp = 0;   // colour and every extra channel use the same blending info
  int is_last = 1, save_as_reference = 0, save_before_ct = 0;
  int emit = 0;                                   // 0 image header + frame, 1 frame only, 2 image header only
  int use_lf_frame = 0;                           // VarDCT: the LF image comes from the LF frame written before (flag 32: no LF coefficients in the LfGroups)
  int lf_level = 0;                               // frame_type 1 (LF frame): its level (1: the LF image of the regular frames)
  int duration = 0;                               // animation (jxlsynth_set_animation): ticks this frame is shown
  int mod_passes = 1;                             // Modular frames: passes (1 .. 3); the squeezed channels are spread over them by shift
  int mod_ds = 1;                                 // ... with downsampling entries (passes.h GetDownsamplingBracket): pass i carries shift np - 1 - i (the first also 2);
                                                  // 0: no entries — everything rides in the last pass, the others are empty
  int num_extra_hdr = -1;                         // extra channels announced by the image header (-1: as the frame has)
  int alpha_premultiplied = 0;                    // image header: alpha_associated
  int xyb_image = 0;                              // Modular frames: the image is XYB encoded (samples are Y, X, B - Y scaled by the LF factors)
  int do_ycbcr = 0; int jpeg_upsampling[3] = {0, 0, 0};   // non-XYB VarDCT frames: YCbCr with per-channel sampling-factor modes (tools/synth_ycbcr.h)
};

// ---- modular sub-stream tokenisation with the fixed global tree --------------------------------------------------
// Tree (decode rule: property > split ? left : right):
//  root: stream_id(prop1) > nLF ? HFMETA : LFCOEF
//  LFCOEF: channel(prop0) > 0 ? (channel > 1 ? B : X) : Y ; each a balanced tree over prop 9 (W+N-NW) cutoffs,
//          leaves = gradient predictor (5)
//  HFMETA: channel > 1 ? (channel > 2 ? SHARP(pred W) : (y > 0 ? HFMUL(pred W) : STRATEGY(pred W))) : CFL (pred zero)
struct TNode { int prop; int split; int l, r; int pred; int ctx; int off = 0, mul_log = 0, mul_bits = 0; };
struct GTree {
  std::vector<TNode> nodes;
  int num_leaves = 0;
  int add_leaf(int pred) { nodes.push_back({-1, 0, -1, -1, pred, 0}); return (int)nodes.size() - 1; }
  int add_inner(int prop, int split, int l, int r) { nodes.push_back({prop, split, l, r, 0, 0}); return (int)nodes.size() - 1; }
};
static int BuildCutoffTree(GTree& t, int prop, const std::vector<int>& cut, int lo, int hi, int pred) {
  // values v with cut[i-1] < v <= cut[i] ... ; build balanced tree over cut[lo..hi)
  if (lo >= hi) return t.add_leaf(pred);
  int mid = (lo + hi) / 2;
  int l = BuildCutoffTree(t, prop, cut, mid + 1, hi, pred);  // prop > cut[mid]
  int r = BuildCutoffTree(t, prop, cut, lo, mid, pred);
  return t.add_inner(prop, cut[mid], l, r);
}
static GTree MakeGlobalTree(int nlf, std::vector<int>* bfs_order) {
  GTree t;
  int root;
  if (LfTreeShape() >= 1) {
    // The tree shape of a default-effort encode (libjxl enc_modular.cc: tree kinds "WP fixed DC" for the LF coefficients, "AC meta" for the HF
    // metadata), restated from the format's point of view: the LF channels look at the weighted predictor's largest neighbouring error
    // (property 15) through 33 cut-offs, every leaf predicts with the weighted predictor (6); the HF-metadata channels: chroma-from-luma maps
    // -> clamped gradient, the (strategy, quantiser) rows -> zero / W under splits on the row and on W, the sharpness map -> zero under N > 0, W > 0.
    static const int wcuts[] = {-500, -392, -255, -191, -127, -95, -63, -47, -31, -23, -15, -11, -7, -4, -3, -1, 0, 1, 3, 5, 7, 11, 15, 23, 31, 47, 63, 95, 127, 191, 255, 392, 500};
    std::vector<int> cut(wcuts, wcuts + sizeof(wcuts) / sizeof(wcuts[0]));
    const int lf = BuildCutoffTree(t, 15, cut, 0, (int)cut.size(), 6);        // one subtree for the three channels (the real encoder does not split on the channel either)
    auto w_split3 = [&](int pred) {   // W > 5 ? (W > 11 ? a : b) : (W > 3 ? c : d)
      return t.add_inner(7, 5, t.add_inner(7, 11, t.add_leaf(pred), t.add_leaf(pred)), t.add_inner(7, 3, t.add_leaf(pred), t.add_leaf(pred)));
    };
    const int qf = w_split3(1), acs = w_split3(0);
    const int blk = t.add_inner(2, 0, qf, acs);                                 // row 1: quantiser, row 0: strategy
    const int epf = t.add_inner(6, 0, t.add_inner(7, 0, t.add_leaf(0), t.add_leaf(0)), t.add_inner(7, 0, t.add_leaf(0), t.add_leaf(0)));
    const int c23 = t.add_inner(0, 2, epf, blk);
    const int cfl = t.add_inner(0, 0, t.add_leaf(5), t.add_leaf(5));
    const int meta = t.add_inner(0, 1, c23, cfl);
    root = t.add_inner(1, nlf, meta, lf);
  } else {
  static const int cuts[] = {-255, -127, -63, -31, -15, -7, -3, -1, 0, 1, 3, 7, 15, 31, 63, 127, 255};
  std::vector<int> cut(cuts, cuts + sizeof(cuts) / sizeof(cuts[0]));
  int ty = BuildCutoffTree(t, 9, cut, 0, (int)cut.size(), 5);
  int tx = BuildCutoffTree(t, 9, cut, 0, (int)cut.size(), 5);
  int tb = BuildCutoffTree(t, 9, cut, 0, (int)cut.size(), 5);
  if (UsePrevChannelProps()) {
    // X looks at |Y| here (property 16: nearest previous channel), B at the sign of X minus its gradient prediction (19) and at |Y| (20: the channel before that)
    tx = t.add_inner(16, 2, BuildCutoffTree(t, 9, cut, 0, (int)cut.size(), 5), tx);
    const int tb2 = t.add_inner(20, 4, BuildCutoffTree(t, 9, cut, 0, (int)cut.size(), 1), BuildCutoffTree(t, 9, cut, 0, (int)cut.size(), 5));
    tb = t.add_inner(19, 0, tb2, tb);
  }
  int lf_c = t.add_inner(0, 1, tb, tx);
  int lf = t.add_inner(0, 0, lf_c, ty);
  int sharp = t.add_leaf(1), hfmul = t.add_leaf(1), strat = t.add_leaf(1), cfl = t.add_leaf(0);
  if (UsePrevChannelProps()) cfl = t.add_inner(17, 0, t.add_leaf(0), t.add_inner(17, -1, t.add_leaf(0), t.add_leaf(1)));   // ytob looks at ytox's value (ytox itself: no previous channel, property 0)
  int blk = t.add_inner(2, 0, hfmul, strat);
  int c23 = t.add_inner(0, 2, sharp, blk);
  int meta = t.add_inner(0, 1, c23, cfl);
  root = t.add_inner(1, nlf, meta, lf);
  }
  // BFS numbering (that is the order nodes are written / leaf contexts are assigned)
  std::vector<int> order{root};
  for (size_t i = 0; i < order.size(); i++) {
    const TNode& n = t.nodes[order[i]];
    if (n.prop >= 0) { order.push_back(n.l); order.push_back(n.r); }
  }
  int leaf = 0;
  for (int id : order) if (t.nodes[id].prop < 0) t.nodes[id].ctx = leaf++;
  t.num_leaves = leaf;
  *bfs_order = order;
  // move root to a known place: remember it as last node
  return t;
}

static void TreeTokens(const GTree& t, const std::vector<int>& bfs, std::vector<Token>& tok) {
  for (int id : bfs) {
    const TNode& n = t.nodes[id];
    if (n.prop < 0) {
      tok.push_back({1, 0});
      tok.push_back({2, (uint32_t)n.pred});
      tok.push_back({3, PackSigned(n.off)});        // offset
      tok.push_back({4, (uint32_t)n.mul_log});      // multiplier = (mul_bits + 1) << mul_log
      tok.push_back({5, (uint32_t)n.mul_bits});
    } else {
      tok.push_back({1, (uint32_t)n.prop + 1});
      tok.push_back({0, PackSigned(n.split)});
    }
  }
}

// The weighted predictor as the format defines it (ISO/IEC 18181-1 "self-correcting predictor"; default parameters): four sub-predictors whose
// recent errors around the sample weight their average.  Encoder-side simulation for the fixed trees above: Predict() gives the prediction (x 8)
// and the largest neighbouring error (property 15), Update() records what the sample turned out to be.
struct WpParams { int p1 = 16, p2 = 10, p3[5] = {7, 7, 7, 0, 0}; uint32_t wmax[4] = {13, 12, 12, 12}; };
// LfTreeShape() 2: the LF-group streams of VarDCT frames spell the predictor's parameters out in their group headers — these, not the defaults
static WpParams LfWpParams() {
  WpParams q;
  if (LfTreeShape() == 2) { q.p1 = 20; q.p2 = 8; const int p3[5] = {5, 9, 6, 3, 2}; const uint32_t wm[4] = {10, 14, 9, 13}; for (int i = 0; i < 5; i++) q.p3[i] = p3[i]; for (int i = 0; i < 4; i++) q.wmax[i] = wm[i]; }
  return q;
}
static void WriteGroupHeaderLf(BitWriter& s) {     // GroupHeader of an LfGroup sub-stream: global tree, predictor parameters, no transforms
  s.put(1, 1);
  if (LfTreeShape() == 2) {
    const WpParams q = LfWpParams();
    s.put(0, 1); s.put(q.p1, 5); s.put(q.p2, 5);
    for (int i = 0; i < 5; i++) s.put(q.p3[i], 5);
    for (int i = 0; i < 4; i++) s.put(q.wmax[i], 4);
  } else s.put(1, 1);
  s.put(0, 2);
}
struct WpSim {
  WpParams par;
  int w = 0;
  std::vector<int32_t> err[2];            // true errors of the two rows in use
  std::vector<uint32_t> sub[4][2];        // per sub-predictor: error magnitudes, (w + 2) entries per row
  int64_t pred[4] = {0, 0, 0, 0}, avg = 0;
  void Init(int width) { w = width; for (int r = 0; r < 2; r++) { err[r].assign((size_t)w + 2, 0); for (int i = 0; i < 4; i++) sub[i][r].assign((size_t)w + 2, 0); } }
  static uint32_t Recip(uint32_t i) { return (1u << 24) / (i + 1); }
  static int Log2Floor(uint64_t v) { int n = 0; while (v >>= 1) n++; return n; }
  int64_t Predict(int x, int y, int64_t N, int64_t W, int64_t NE, int64_t NW, int64_t NN, int32_t* max_error) {
    const uint32_t* kMaxWeight = par.wmax;
    const int cur = y & 1, up = cur ^ 1;
    const int n = x, ne = x + 1 < w ? x + 1 : x, nw = x > 0 ? x - 1 : x;
    uint32_t wt[4];
    for (int i = 0; i < 4; i++) {
      const uint64_t e = (uint64_t)sub[i][up][n] + sub[i][up][ne] + sub[i][up][nw];
      int shift = Log2Floor(e + 1) - 5;
      if (shift < 0) shift = 0;
      wt[i] = 4 + (uint32_t)(((uint64_t)kMaxWeight[i] * Recip((uint32_t)(e >> shift))) >> shift);
    }
    N *= 8; W *= 8; NE *= 8; NW *= 8; NN *= 8;
    const int64_t eW = x ? err[cur][x - 1] : 0, eN = err[up][n], eNW = err[up][nw], eNE = err[up][ne];
    int64_t m = eW;
    if (std::llabs(eN) > std::llabs(m)) m = eN;
    if (std::llabs(eNW) > std::llabs(m)) m = eNW;
    if (std::llabs(eNE) > std::llabs(m)) m = eNE;
    *max_error = (int32_t)m;
    pred[0] = W + NE - N;
    pred[1] = N - (((eW + eN + eNE) * par.p1) >> 5);
    pred[2] = W - (((eW + eN + eNW) * par.p2) >> 5);
    pred[3] = N - ((eNW * par.p3[0] + eN * par.p3[1] + eNE * par.p3[2] + (NN - N) * par.p3[3] + (NW - W) * par.p3[4]) >> 5);
    uint32_t total = wt[0] + wt[1] + wt[2] + wt[3];
    const int lg = Log2Floor(total);
    total = 0;
    for (int i = 0; i < 4; i++) { wt[i] >>= lg - 4; total += wt[i]; }
    int64_t acc = (int64_t)(total >> 1) - 1;
    for (int i = 0; i < 4; i++) acc += pred[i] * (int64_t)wt[i];
    avg = (acc * (int64_t)Recip(total - 1)) >> 24;
    if (((eN ^ eW) | (eN ^ eNW)) <= 0) {
      const int64_t hi = std::max(W, std::max(NE, N)), lo = std::min(W, std::min(NE, N));
      avg = std::max(lo, std::min(hi, avg));
    }
    return avg;
  }
  void Update(int64_t sample, int x, int y) {
    const int cur = y & 1, up = cur ^ 1;
    sample *= 8;
    err[cur][x] = (int32_t)(avg - sample);
    for (int i = 0; i < 4; i++) {
      const uint32_t e = (uint32_t)((std::llabs(pred[i] - sample) + 3) >> 3);
      sub[i][cur][x] = e;
      sub[i][up][x + 1] += e;
    }
  }
};

// tokenises channels (each w x h ints) of one modular sub-stream under the global tree
struct ChanRef { const int32_t* d; int w, h; };
static void ModularTokens(const GTree& t, int root, const std::vector<ChanRef>& chans, int stream_id, std::vector<Token>& tok, const WpParams& wpar = WpParams()) {
  bool tree_wp = false;
  for (const TNode& n : t.nodes) tree_wp |= n.prop == 15 || (n.prop < 0 && n.pred == 6);
  WpSim wp;
  wp.par = wpar;
  for (size_t ci = 0; ci < chans.size(); ci++) {
    const ChanRef& ch = chans[ci];
    if (tree_wp) wp.Init(ch.w);
    for (int y = 0; y < ch.h; y++) {
      const int32_t* p = ch.d + (size_t)y * ch.w;
      const int32_t* pn = y ? p - ch.w : nullptr;
      for (int x = 0; x < ch.w; x++) {
        int64_t W = x ? p[x - 1] : (y ? pn[x] : 0);
        int64_t N = y ? pn[x] : W;
        int64_t NW = (x && y) ? pn[x - 1] : W;
        int props[16 + 4 * 4] = {(int)ci, stream_id, y, x, 0, 0, (int)N, (int)W, 0, (int)(W + N - NW)};
        int64_t wp_guess = 0;
        if (tree_wp) {
          const int64_t NE = (y && x + 1 < ch.w) ? pn[x + 1] : N, NN = y > 1 ? pn[x - ch.w] : N;
          int32_t max_error = 0;
          wp_guess = (wp.Predict(x, y, N, W, NE, NW, NN, &max_error) + 3) >> 3;
          props[15] = max_error;
        }
        {  // encoding.cc PrecomputeReferences: earlier channels of this stream with the same size, nearest first
          int r = 0;
          for (int cj = (int)ci - 1; cj >= 0 && r < 4; cj--) {
            const ChanRef& rc = chans[(size_t)cj];
            if (rc.w != ch.w || rc.h != ch.h) continue;
            const int32_t* rp = rc.d + (size_t)y * rc.w;
            const int64_t v = rp[x], rl = x ? rp[x - 1] : 0, rt = y ? rp[x - rc.w] : rl, rtl = (x && y) ? rp[x - 1 - rc.w] : rl;
            const int64_t m = std::min(rt, rl), M = std::max(rt, rl), g = rtl < m ? M : (rtl > M ? m : rt + rl - rtl);
            props[16 + 4 * r] = (int)std::llabs(v); props[17 + 4 * r] = (int)v; props[18 + 4 * r] = (int)std::llabs(v - g); props[19 + 4 * r] = (int)(v - g);
            r++;
          }
        }
        int pos = root;
        while (t.nodes[pos].prop >= 0) pos = props[t.nodes[pos].prop] > t.nodes[pos].split ? t.nodes[pos].l : t.nodes[pos].r;
        const TNode& leaf = t.nodes[pos];
        int64_t guess;
        if (leaf.pred == 0) guess = 0;
        else if (leaf.pred == 1) guess = W;
        else if (leaf.pred == 6) guess = wp_guess;
        else { int64_t m = std::min(N, W), M = std::max(N, W), g = N + W - NW; guess = NW < m ? M : (NW > M ? m : g); }
        tok.push_back({(uint32_t)leaf.ctx, PackSigned((int32_t)(p[x] - guess))});
        if (tree_wp) wp.Update(p[x], x, y);
      }
    }
  }
}

// LZ77 over a finished token stream (dec_ans.h ANSSymbolReader: the window holds the decoded values of the whole stream, whatever their contexts): runs
// that repeat the value before (distance 1: special distance code 1) or the value one row up (distance `row`: special distance code 0 = the stream's
// distance multiplier, its widest channel) become a length symbol in the context of the run's first value + a distance token in the extra context
// special = false: a stream without a distance multiplier (AC coefficients): no special distance codes, the distance token is distance - 1; copies at
// distance 1 (runs), 2 and 3 (repeating pairs / triples)
static void ApplyLz77(std::vector<Token>& tok, uint32_t dist_ctx, const EntropyCoder& proto, size_t row, bool special = true) {
  std::vector<Token> out;
  const size_t n = tok.size();
  size_t i = 0;
  while (i < n) {
    size_t l1 = 0, lr = 0, l2 = 0, l3 = 0;
    if (i >= 1) while (i + l1 < n && tok[i + l1].value == tok[i + l1 - 1].value) l1++;
    if (row > 1 && i >= row) while (i + lr < n && tok[i + lr].value == tok[i + lr - row].value) lr++;
    if (!special && i >= 2) while (i + l2 < n && tok[i + l2].value == tok[i + l2 - 2].value) l2++;
    if (!special && i >= 3) while (i + l3 < n && tok[i + l3].value == tok[i + l3 - 3].value) l3++;
    const size_t len = std::max(std::max(l1, lr), std::max(l2, l3));
    if (len >= std::max<size_t>(proto.lz_min_length, 6)) {
      Token t; t.ctx = tok[i].ctx; t.raw = 1;
      uint32_t sym, nb, bits;
      EncodeHybrid(proto.lz_len_cfg, (uint32_t)len - proto.lz_min_length, &sym, &nb, &bits);
      t.value = proto.lz_min_symbol + sym; t.nb = (uint8_t)nb; t.bits = bits;
      out.push_back(t);
      if (special) out.push_back(Token{dist_ctx, lr >= l1 ? 0u : 1u});      // (ties go to the row copy: flat areas then exercise both distance codes)
      else out.push_back(Token{dist_ctx, l3 == len ? 2u : l2 == len ? 1u : 0u});
      i += len;
    } else out.push_back(tok[i++]);
  }
  tok.swap(out);
}

// ---- image / frame headers -------------------------------------------------------------------------------------
static void WriteSize(BitWriter& w, uint32_t xs, uint32_t ys) {
  w.put(0, 1);  // small = 0
  WriteU32(w, ys, {9, 1}, {13, 1}, {18, 1}, {30, 1});
  w.put(0, 3);  // ratio 0
  WriteU32(w, xs, {9, 1}, {13, 1}, {18, 1}, {30, 1});
}

// Explicit upsampling weights for factor `up`: products of a Catmull-Rom kernel sampled at the sub-pixel centres, stored as
// the upper triangle of the symmetric (5N x 5N) matrix (N = up / 2), every value rounded to F16 like the stream stores it.
static std::vector<float> CustomUpWeights(int up) {
  const int N = up / 2;
  std::vector<double> a(5 * N);
  auto cr = [](double t) { t = std::fabs(t); return t < 1 ? 1.5 * t * t * t - 2.5 * t * t + 1 : (t < 2 ? -0.5 * t * t * t + 2.5 * t * t - 4 * t + 2 : 0.0); };
  for (int k = 0; k < N; k++) {
    const double d = (k + 0.5) / up - 0.5;   // sub-pixel centre relative to the input sample
    double sum = 0;
    for (int j = 0; j < 5; j++) { a[5 * k + j] = cr((j - 2) - d); sum += a[5 * k + j]; }
    for (int j = 0; j < 5; j++) a[5 * k + j] /= sum;
  }
  std::vector<float> wts;
  for (int y = 0; y < 5 * N; y++) for (int x = y; x < 5 * N; x++) wts.push_back(RoundToHalf((float)(a[y] * a[x])));
  return wts;
}

// ---- image features of the next frame(s): patch dictionary and splines, written at the head of LfGlobal (dec_patch_dictionary.cc,
// splines.cc).  Set through jxlsynth_set_features as flat integer scripts:
//   patches: { ref, x0, y0, xsize, ysize, count, count x { x, y, (1 + num_extra) x { mode, alpha_channel, clamp } } } ...
//   splines: { quant_adjust, num, num x { start_x, start_y, ncp, ncp x { dx, dy } (double deltas), 3 x 32 colour DCT, 32 sigma DCT } }
static thread_local std::vector<int32_t> g_patches, g_splines;
static thread_local int g_feature_extra = 0;      // number of extra channels the patch blending entries cover
static void WriteFeatures(BitWriter& s) {
  if (!g_patches.empty()) {
    std::vector<Token> tok;
    size_t i = 0;
    std::vector<Token> body;
    uint32_t nrefs = 0;
    while (i < g_patches.size()) {
      const int32_t* r = &g_patches[i];
      nrefs++;
      body.push_back({1, (uint32_t)r[0]});
      body.push_back({3, (uint32_t)r[1]}); body.push_back({3, (uint32_t)r[2]});
      body.push_back({2, (uint32_t)r[3] - 1}); body.push_back({2, (uint32_t)r[4] - 1});
      const int count = r[5];
      body.push_back({7, (uint32_t)count - 1});
      i += 6;
      int px = 0, py = 0;
      for (int k = 0; k < count; k++) {
        const int x = g_patches[i], y = g_patches[i + 1];
        i += 2;
        if (k == 0) { body.push_back({4, (uint32_t)x}); body.push_back({4, (uint32_t)y}); }
        else { body.push_back({6, PackSigned(x - px)}); body.push_back({6, PackSigned(y - py)}); }
        px = x; py = y;
        for (int e = 0; e < 1 + g_feature_extra; e++) {
          const int mode = g_patches[i], alpha = g_patches[i + 1], clamp = g_patches[i + 2];
          i += 3;
          body.push_back({5, (uint32_t)mode});
          if (mode >= 4 && g_feature_extra > 1) body.push_back({8, (uint32_t)alpha});
          if (mode >= 3) body.push_back({9, (uint32_t)clamp});
        }
      }
    }
    tok.push_back({0, nrefs});
    tok.insert(tok.end(), body.begin(), body.end());
    EntropyCoder code;
    std::vector<const std::vector<Token>*> ss{&tok};
    BuildEntropyCoder(ss, 10, UintConfig{4, 2, 0}, 4, code);
    WriteEntropyCode(s, code);
    EncodeTokens(s, code, tok);
  }
  if (!g_splines.empty()) {
    std::vector<Token> tok;
    const int32_t* v = g_splines.data();
    const int adjust = v[0], num = v[1];
    tok.push_back({2, (uint32_t)num - 1});
    // starting points first, then the quantisation adjustment, then the splines
    std::vector<size_t> at;
    size_t i = 2;
    for (int k = 0; k < num; k++) { at.push_back(i); const int ncp = g_splines[i + 2]; i += 3 + 2 * (size_t)ncp + 128; }
    int lx = 0, ly = 0;
    for (int k = 0; k < num; k++) {
      const int x = g_splines[at[k]], y = g_splines[at[k] + 1];
      if (k == 0) { tok.push_back({1, (uint32_t)x}); tok.push_back({1, (uint32_t)y}); }
      else { tok.push_back({1, PackSigned(x - lx)}); tok.push_back({1, PackSigned(y - ly)}); }
      lx = x; ly = y;
    }
    tok.push_back({0, PackSigned(adjust)});
    for (int k = 0; k < num; k++) {
      const int32_t* q = &g_splines[at[k]];
      const int ncp = q[2];
      tok.push_back({3, (uint32_t)ncp});
      for (int c = 0; c < 2 * ncp; c++) tok.push_back({4, PackSigned(q[3 + c])});
      for (int c = 0; c < 128; c++) tok.push_back({5, PackSigned(q[3 + 2 * ncp + c])});
    }
    EntropyCoder code;
    std::vector<const std::vector<Token>*> ss{&tok};
    BuildEntropyCoder(ss, 6, UintConfig{4, 2, 0}, 4, code);
    WriteEntropyCode(s, code);
    EncodeTokens(s, code, tok);
  }
}

// ---- embedded ICC profile, encoder side (the inverse of icc_codec.cc UnpredictICC): header as differences from the predicted
// header, one command per tag (known names, implicit offsets / sizes, TRC and XYZ triples where they apply), tag data as a mix
// of insert / shuffle / predict commands chosen to exercise the decoder rather than to compress.
static thread_local std::vector<uint8_t> g_icc;
// enumerated colour encoding of the image headers written from now on (jxlsynth_set_color): white point / primaries / transfer function
// enums of color_encoding_internal.h, gamma in 1e-7 units (0 = use tf), intensity target in nits
struct ColorOverride { bool set = false; int white_point = 1, primaries = 1, tf = 13; uint32_t gamma = 0; float intensity_target = 255.0f; };
static thread_local ColorOverride g_color;
static thread_local bool g_spot_set = false;          // the image's extra channel is a spot colour (jxlsynth_set_spot) instead of alpha
static thread_local float g_spot[4] = {0, 0, 0, 0};   // its colour and solidity
static thread_local int g_float_exp_bits = 0;     // != 0: the image's samples are floats with this many exponent bits (jxlsynth_set_float)
static void IccVarint(std::vector<uint8_t>& v, uint64_t x) { while (x > 127) { v.push_back((uint8_t)(x | 128)); x >>= 7; } v.push_back((uint8_t)x); }
static std::vector<uint8_t> IccShuffleFwd(const std::vector<uint8_t>& in, size_t width) {   // decoder: out[i] = in[j], j walking columns
  const size_t n = in.size(), rows = (n + width - 1) / width;
  std::vector<uint8_t> out(n);
  size_t start = 0, j = 0;
  for (size_t i = 0; i < n; i++) { out[j] = in[i]; j += rows; if (j >= n) j = ++start; }
  return out;
}
static std::vector<uint8_t> EncodeIccBytes(const std::vector<uint8_t>& icc) {
  std::vector<uint8_t> cmds, data;
  const size_t n = icc.size();
  // header
  uint8_t guess[128] = {0};
  guess[0] = (uint8_t)(n >> 24); guess[1] = (uint8_t)(n >> 16); guess[2] = (uint8_t)(n >> 8); guess[3] = (uint8_t)n;
  guess[8] = 4; memcpy(guess + 12, "mntr", 4); memcpy(guess + 16, "RGB ", 4); memcpy(guess + 20, "XYZ ", 4); memcpy(guess + 36, "acsp", 4);
  guess[70] = 246; guess[71] = 214; guess[73] = 1; guess[78] = 211; guess[79] = 45;
  for (size_t i = 0; i < 128 && i < n; i++) {
    if (i == 8) memcpy(guess + 80, &icc[4], 4);
    if (i == 41) { if (icc[40] == 'A') memcpy(guess + 41, "PPL", 3); else if (icc[40] == 'M') memcpy(guess + 41, "SFT", 3); }
    if (i == 42) { if (icc[40] == 'S' && icc[41] == 'G') memcpy(guess + 42, "I ", 2); else if (icc[40] == 'S' && icc[41] == 'U') memcpy(guess + 42, "NW", 2); }
    data.push_back((uint8_t)(icc[i] - guess[i]));
  }
  size_t pos = std::min<size_t>(128, n);
  if (n >= 132) {
    auto rd32 = [&](size_t o) { return ((uint32_t)icc[o] << 24) | ((uint32_t)icc[o + 1] << 16) | ((uint32_t)icc[o + 2] << 8) | icc[o + 3]; };
    const uint32_t ntags = rd32(128);
    if (132 + (size_t)ntags * 12 > n) throw std::runtime_error("ICC tag table out of bounds");
    IccVarint(cmds, (uint64_t)ntags + 1);
    static const char* kKnown[17] = {"cprt", "wtpt", "bkpt", "rXYZ", "gXYZ", "bXYZ", "kXYZ", "rTRC", "gTRC", "bTRC", "kTRC", "chad", "desc", "chrm", "dmnd", "dmdd", "lumi"};
    uint64_t last_start = 128 + 12ull * ntags, last_size = 0;
    for (uint32_t t = 0; t < ntags;) {
      const size_t o = 132 + 12 * (size_t)t;
      const std::string name((const char*)&icc[o], 4);
      const uint32_t start = rd32(o + 4), size = rd32(o + 8);
      auto same = [&](uint32_t k, const char* nm, uint32_t st, uint32_t sz) {
        return t + k < ntags && memcmp(&icc[132 + 12 * (size_t)(t + k)], nm, 4) == 0 && rd32(132 + 12 * (size_t)(t + k) + 4) == st && rd32(132 + 12 * (size_t)(t + k) + 8) == sz;
      };
      int code = 1, span = 1;
      if (name == "rTRC" && same(1, "gTRC", start, size) && same(2, "bTRC", start, size)) { code = 2; span = 3; }
      else if (name == "rXYZ" && same(1, "gXYZ", start + size, size) && same(2, "bXYZ", start + 2 * size, size)) { code = 3; span = 3; }
      else for (int k = 0; k < 17; k++) if (name == kKnown[k]) code = 4 + k;
      uint64_t implied_size = last_size;
      if (name == "rXYZ" || name == "gXYZ" || name == "bXYZ" || name == "kXYZ" || name == "wtpt" || name == "bkpt" || name == "lumi") implied_size = 20;
      int c = code;
      if (start != last_start + last_size) c |= 64;
      if (size != implied_size) c |= 128;
      cmds.push_back((uint8_t)c);
      if (code == 1) data.insert(data.end(), name.begin(), name.end());
      if (c & 64) IccVarint(cmds, start);
      if (c & 128) IccVarint(cmds, size);
      last_start = start; last_size = size;
      t += span;
    }
    cmds.push_back(0);     // end of the tag list
    pos = 132 + 12 * (size_t)ntags;
  }
  // tag data: cycle through the content commands
  int turn = 0;
  while (pos < n) {
    size_t len = std::min<size_t>(n - pos, 24 + 37 * (size_t)(turn % 5));
    const int kind = turn++ % 6;
    if (kind == 5 && n - pos >= 20 && memcmp(&icc[pos], "XYZ \0\0\0\0", 8) == 0) { cmds.push_back(10); data.insert(data.end(), icc.begin() + pos + 8, icc.begin() + pos + 20); pos += 20; continue; }
    if (kind == 0 || kind == 5) { cmds.push_back(1); IccVarint(cmds, len); data.insert(data.end(), icc.begin() + pos, icc.begin() + pos + len); }
    else if (kind == 1 || kind == 2) {
      const size_t width = kind == 1 ? 2 : 4;
      cmds.push_back((uint8_t)(kind == 1 ? 2 : 3)); IccVarint(cmds, len);
      const std::vector<uint8_t> sh = IccShuffleFwd(std::vector<uint8_t>(icc.begin() + pos, icc.begin() + pos + len), width);
      data.insert(data.end(), sh.begin(), sh.end());
    } else {
      // predict: width 1 / 2 / 4, order 0..2, explicit stride now and then
      const size_t width = kind == 3 ? 1 : (turn % 2 ? 2 : 4);
      const int order = turn % 3;
      const bool explicit_stride = turn % 4 == 0;
      const uint64_t stride = explicit_stride ? width * 2 : width;
      if (pos == 0 || ((pos - 1) >> 2) < stride) { cmds.push_back(1); IccVarint(cmds, len); data.insert(data.end(), icc.begin() + pos, icc.begin() + pos + len); pos += len; continue; }
      cmds.push_back(4);
      cmds.push_back((uint8_t)((width - 1) | (order << 2) | (explicit_stride ? 16 : 0)));
      if (explicit_stride) IccVarint(cmds, stride);
      IccVarint(cmds, len);
      std::vector<uint8_t> res(len);
      for (size_t i = 0; i < len; i++) {
        const size_t unit = pos + i - i % width;
        uint64_t past[3];
        for (int k = 0; k < 3; k++) { uint64_t v = 0; for (size_t b = 0; b < width; b++) v = (v << 8) | icc[unit - (size_t)stride * (k + 1) + b]; past[k] = v; }
        const uint64_t pred = order == 0 ? past[0] : order == 1 ? 2 * past[0] - past[1] : 3 * past[0] - 3 * past[1] + past[2];
        res[i] = (uint8_t)(icc[pos + i] - (uint8_t)(pred >> (8 * (width - 1 - i % width))));
      }
      if (width > 1) res = IccShuffleFwd(res, width);
      data.insert(data.end(), res.begin(), res.end());
    }
    pos += len;
  }
  std::vector<uint8_t> out;
  IccVarint(out, n);
  IccVarint(out, cmds.size());
  out.insert(out.end(), cmds.begin(), cmds.end());
  out.insert(out.end(), data.begin(), data.end());
  return out;
}
static void WriteIccStream(BitWriter& w, const std::vector<uint8_t>& icc) {
  const std::vector<uint8_t> enc = EncodeIccBytes(icc);
  auto kind1 = [](int b) { if ((b >= 'a' && b <= 'z') || (b >= 'A' && b <= 'Z')) return 0; if ((b >= '0' && b <= '9') || b == '.' || b == ',') return 1; if (b <= 1) return 2 + b; if (b < 16) return 4; if (b > 240 && b < 255) return 5; if (b == 255) return 6; return 7; };
  auto kind2 = [](int b) { if ((b >= 'a' && b <= 'z') || (b >= 'A' && b <= 'Z')) return 0; if ((b >= '0' && b <= '9') || b == '.' || b == ',') return 1; if (b < 16) return 2; if (b > 240) return 3; return 4; };
  std::vector<Token> tok;
  for (size_t i = 0; i < enc.size(); i++) {
    const int b1 = i ? enc[i - 1] : 0, b2 = i > 1 ? enc[i - 2] : 0;
    tok.push_back({(uint32_t)(i <= 128 ? 0 : 1 + kind1(b1) + 8 * kind2(b2)), enc[i]});
  }
  WriteU64(w, enc.size());
  EntropyCoder code;
  std::vector<const std::vector<Token>*> ss{&tok};
  BuildEntropyCoder(ss, 41, UintConfig{4, 2, 0}, 6, code);
  WriteEntropyCode(w, code);
  EncodeTokens(w, code, tok);
}

static int g_anim_num = 0, g_anim_den = 1, g_anim_loops = 0;   // jxlsynth_set_animation: ticks per second (0: no animation), loop count
static int g_preview_w = 0, g_preview_h = 0;   // jxlsynth_set_preview: the image header announces a preview frame of this size (the caller emits it first)
// headers.cc PreviewHeader
static void WritePreviewSize(BitWriter& w, int xs, int ys) {
  const bool div8 = xs % 8 == 0 && ys % 8 == 0;
  auto dim = [&](int v) {
    if (div8) WriteU32(w, (uint32_t)(v / 8), {0, 16}, {0, 32}, {5, 1}, {9, 33});
    else WriteU32(w, (uint32_t)v, {6, 1}, {8, 65}, {10, 321}, {12, 1345});
  };
  w.put(div8 ? 1 : 0, 1);
  dim(ys);
  w.put(0, 3);     // ratio 0: the width follows
  dim(xs);
}
// XYB images written from now on (this thread) carry their own OpsinInverseMatrix bundle: the library's matrix, opsin biases and quantisation biases as binary16 values
// (so a decoder that reads them computes with slightly different numbers than one that falls back to its defaults)
static bool& CustomOpsin() { static thread_local bool v = false; return v; }
static void WriteImageHeader(BitWriter& w, int xs, int ys, const Params& p, bool xyb, int bits, bool has_alpha, bool gray) {
  w.put(0xFF, 8); w.put(0x0A, 8);
  WriteSize(w, xs, ys);
  const bool custom_opsin = xyb && CustomOpsin();
  const bool custom_up = (p.upsampling > 1 && p.custom_up_weights) || custom_opsin;      // (either makes the transform-data bundle non-default)
  const bool preview = g_preview_w > 0 && g_preview_h > 0, anim = g_anim_num > 0;
  bool all_default = xyb && bits == 8 && !has_alpha && !p.hdr && p.out_bits != 32 && !gray && p.orientation == 1 && !custom_up && g_icc.empty() && !g_color.set && !g_float_exp_bits && !preview && !anim;
  w.put(all_default, 1);
  if (!all_default) {
    const bool custom_target = g_color.set && g_color.intensity_target != 255.0f;
    bool extra_fields = p.hdr || p.orientation != 1 || custom_target || preview || anim;
    w.put(extra_fields, 1);
    if (extra_fields) {
      w.put((uint32_t)(p.orientation - 1), 3);
      w.put(0, 1);                                  // no intrinsic size
      w.put(preview ? 1 : 0, 1);
      if (preview) WritePreviewSize(w, g_preview_w, g_preview_h);
      w.put(anim ? 1 : 0, 1);
      if (anim) {                                   // headers.cc AnimationHeader
        WriteU32(w, (uint32_t)g_anim_num, {0, 100}, {0, 1000}, {10, 1}, {30, 1});
        WriteU32(w, (uint32_t)g_anim_den, {0, 1}, {0, 1001}, {8, 1}, {10, 1});
        WriteU32(w, (uint32_t)g_anim_loops, {0, 0}, {3, 0}, {16, 0}, {32, 0});
        w.put(0, 1);                                // have_timecodes
      }
    }
    // BitDepth
    if (p.out_bits == 32) { w.put(1, 1); WriteU32(w, 32, {0, 32}, {0, 16}, {0, 24}, {6, 1}); w.put(8 - 1, 4); }
    else if (g_float_exp_bits) { w.put(1, 1); WriteU32(w, bits, {0, 32}, {0, 16}, {0, 24}, {6, 1}); w.put((uint32_t)g_float_exp_bits - 1, 4); }
    else { w.put(0, 1); WriteU32(w, bits, {0, 8}, {0, 10}, {0, 12}, {6, 1}); }
    w.put(1, 1);  // modular_16bit_buffers
    WriteU32(w, has_alpha ? 1 : 0, {0, 0}, {0, 1}, {4, 2}, {12, 1});
    if (has_alpha && g_spot_set) {
      w.put(0, 1);                                         // not the default 8-bit alpha
      WriteU32(w, 2, {0, 0}, {0, 1}, {4, 2}, {6, 18});     // type kSpotColor
      w.put(0, 1); WriteU32(w, bits, {0, 8}, {0, 10}, {0, 12}, {6, 1});
      WriteU32(w, 0, {0, 0}, {0, 3}, {0, 4}, {3, 1});      // dim_shift
      WriteU32(w, 0, {0, 0}, {4, 0}, {5, 16}, {10, 48});   // name
      for (int i = 0; i < 4; i++) WriteF16(w, g_spot[i]);  // spot colour, solidity
    } else if (has_alpha) {
      if (bits == 8 && !p.alpha_premultiplied) w.put(1, 1);  // d_alpha
      else {
        w.put(0, 1);
        WriteU32(w, 0, {0, 0}, {0, 1}, {4, 2}, {6, 18});  // type alpha
        if (g_float_exp_bits) { w.put(1, 1); WriteU32(w, bits, {0, 32}, {0, 16}, {0, 24}, {6, 1}); w.put((uint32_t)g_float_exp_bits - 1, 4); }   // float samples: alpha as well
        else { w.put(0, 1); WriteU32(w, bits, {0, 8}, {0, 10}, {0, 12}, {6, 1}); }
        WriteU32(w, 0, {0, 0}, {0, 3}, {0, 4}, {3, 1});  // dim_shift
        WriteU32(w, 0, {0, 0}, {4, 0}, {5, 16}, {10, 48});  // name
        w.put(p.alpha_premultiplied ? 1 : 0, 1);  // alpha_associated
      }
    }
    w.put(xyb, 1);
    // ColorEncoding
    bool ce_default = !p.hdr && !gray && g_icc.empty() && !g_color.set;
    w.put(ce_default, 1);
    if (!g_icc.empty()) {
      w.put(1, 1);                                                  // want_icc: only the colour space follows
      WriteU32(w, gray ? 1 : 0, {0, 0}, {0, 1}, {4, 2}, {6, 18});
    } else if (g_color.set) {
      w.put(0, 1);  // want_icc
      WriteU32(w, gray ? 1 : 0, {0, 0}, {0, 1}, {4, 2}, {6, 18});
      WriteU32(w, (uint32_t)g_color.white_point, {0, 0}, {0, 1}, {4, 2}, {6, 18});
      if (!gray) WriteU32(w, (uint32_t)g_color.primaries, {0, 0}, {0, 1}, {4, 2}, {6, 18});
      if (g_color.gamma) { w.put(1, 1); w.put(g_color.gamma, 24); }
      else { w.put(0, 1); WriteU32(w, (uint32_t)g_color.tf, {0, 0}, {0, 1}, {4, 2}, {6, 18}); }
      WriteU32(w, 1, {0, 0}, {0, 1}, {4, 2}, {6, 18});              // rendering intent relative
    } else if (!ce_default) {
      w.put(0, 1);  // want_icc
      WriteU32(w, gray ? 1 : 0, {0, 0}, {0, 1}, {4, 2}, {6, 18});   // colour space
      WriteU32(w, 1, {0, 0}, {0, 1}, {4, 2}, {6, 18});              // white point D65
      if (!gray) WriteU32(w, 1, {0, 0}, {0, 1}, {4, 2}, {6, 18});   // primaries sRGB
      w.put(0, 1);                                                  // have_gamma = 0
      WriteU32(w, p.hdr ? 8 : 13, {0, 0}, {0, 1}, {4, 2}, {6, 18}); // tf linear | sRGB
      WriteU32(w, 1, {0, 0}, {0, 1}, {4, 2}, {6, 18});              // rendering intent relative
    }
    if (extra_fields) {
      if (p.hdr || custom_target) { w.put(0, 1); WriteF16(w, custom_target ? g_color.intensity_target : 1000.0f); WriteF16(w, 0.0f); w.put(0, 1); WriteF16(w, 0.0f); }   // tone mapping: intensity_target
      else w.put(1, 1);                                                                                      // tone mapping all default
    }
    WriteU64(w, 0);  // extensions
  }
  if (!custom_up) w.put(1, 1);  // default_m
  else {
    w.put(0, 1);
    if (xyb) {
      w.put(custom_opsin ? 0 : 1, 1);   // OpsinInverseMatrix all_default
      if (custom_opsin) {
        for (float v : {11.031566901960783f, -9.866943921568629f, -0.16462299647058826f, -3.254147380392157f, 4.418770392156863f, -0.16462299647058826f,
                        -3.6588512862745097f, 2.7129230470588235f, 1.9459282392156863f}) WriteF16(w, v);
        for (int i = 0; i < 3; i++) WriteF16(w, -0.0037930732552754493f);
        for (float v : {1.0f - 0.05465007330715401f, 1.0f - 0.07005449891748593f, 1.0f - 0.049935103337343655f, 0.145f}) WriteF16(w, v);
      }
    }
    const bool up_w = p.upsampling > 1 && p.custom_up_weights;
    w.put(!up_w ? 0 : p.upsampling == 2 ? 1 : p.upsampling == 4 ? 2 : 4, 3);   // cw_mask
    if (up_w) for (float v : CustomUpWeights(p.upsampling)) WriteF16(w, v);
  }
  if (!g_icc.empty()) WriteIccStream(w, g_icc);
  w.align();
}

// bit-plane split of the passes: 2 passes -> shifts {2, 0}; 3 passes -> {3, 1, 0}
static int PassShift(int num_passes, int pass) { return pass + 1 == num_passes ? 0 : (num_passes == 2 ? 2 : (pass == 0 ? 3 : 1)); }

// VarDCT frames written from now on (this thread) carry a RestorationFilter bundle with every custom field set: gaborish weights, EPF sharpness LUT, channel scales,
// sigma parameters (loop_filter.cc) — the encoder side does not look at them
static bool& CustomFilters() { static thread_local bool v = false; return v; }
// x_qm_scale / b_qm_scale of the VarDCT frames written from now on (this thread): the X / B quantisation steps are scaled by 0.8^(scale - 2) (frame_header.cc; 3 / 2 by default)
static int* QmScales() { static thread_local int v[2] = {3, 2}; return v; }
// quant_lf (LfGlobal Quantizer, 1..65536; 16 by default) of the VarDCT frames written from now on (this thread): the LF steps are the channel's LF factor * 65536 / global_scale / quant_lf
static int& QuantLf() { static thread_local int v = 16; return v; }
// extra_precision (0..3) of the LF groups of the VarDCT frames written from now on (this thread): the LF coefficients are coded in steps 2^extra_precision times finer
static int& LfExtraPrecision() { static thread_local int v = 0; return v; }
static void WriteFrameHeader(BitWriter& w, const Params& p, bool modular, bool xyb, int num_extra, int group_shift, bool lf_default, int frame_w = 0, int frame_h = 0) {
  w.put(0, 1);  // all_default
  w.put((uint32_t)p.frame_type, 2);
  w.put(modular ? 1 : 0, 1);
  WriteU64(w, ((!modular && p.skip_lf_smoothing) ? 0x80 : 0) | (p.noise ? 1 : 0) | (g_patches.empty() ? 0 : 2) | (g_splines.empty() ? 0 : 16) | ((!modular && p.use_lf_frame) ? 32 : 0));
  if (!xyb) {
    w.put(p.do_ycbcr ? 1 : 0, 1);
    if (p.do_ycbcr) for (int c = 0; c < 3; c++) w.put((uint32_t)p.jpeg_upsampling[c], 2);   // YCbCrChromaSubsampling, channels Cb, Y, Cr
  }
  const uint32_t ups_sel = p.upsampling == 2 ? 1 : p.upsampling == 4 ? 2 : p.upsampling == 8 ? 3 : 0;
  if (modular || !p.use_lf_frame) {
    w.put(ups_sel, 2);      // upsampling
    for (int i = 0; i < num_extra; i++) w.put(ups_sel, 2);   // ec_upsampling: same factor
  }
  if (modular) w.put(group_shift, 2);
  if (!modular && xyb) { w.put((uint32_t)QmScales()[0], 3); w.put((uint32_t)QmScales()[1], 3); }
  if (p.frame_type != 2) {
    const int np = modular ? p.mod_passes : p.num_passes;
    w.put((uint32_t)(np - 1), 2);  // num_passes (1, 2, 3)
    if (np != 1) {
      const int nds = (modular ? p.mod_ds : p.pass_ds) ? np - 1 : 0;
      w.put((uint32_t)nds, 2);                             // num_downsample (0 .. 2)
      for (int i = 0; i + 1 < np; i++) w.put(modular ? 0u : (uint32_t)PassShift(np, i), 2);   // shift of every pass but the last
      // entry i: "pass i completes the image at 1 / (2 << (nds - 1 - i))" -> downsample 4, 2 (np = 3) or 2 (np = 2); U32(Val 1, 2, 4, 8)
      for (int i = 0; i < nds; i++) w.put((uint32_t)(nds - i), 2);
      for (int i = 0; i < nds; i++) w.put((uint32_t)i, 2);    // last_pass: U32(Val 0, 1, 2, Bits(3))
    }
  }
  bool partial = false;
  if (p.frame_type == 1) w.put((uint32_t)(p.lf_level - 1), 2);   // lf_level 1..4 (an LF frame has no crop: its size is the image's / 8^level)
  else w.put(p.have_crop ? 1 : 0, 1);  // have_crop
  if (p.frame_type != 1 && p.have_crop) {
    auto pack = [](int32_t v) { return v >= 0 ? (uint32_t)v * 2 : (uint32_t)(-v) * 2 - 1; };
    if (p.frame_type != 2) {
      WriteU32(w, pack(p.crop_x0), {8, 0}, {11, 256}, {14, 2304}, {30, 18688});
      WriteU32(w, pack(p.crop_y0), {8, 0}, {11, 256}, {14, 2304}, {30, 18688});
    }
    WriteU32(w, (uint32_t)frame_w, {8, 0}, {11, 256}, {14, 2304}, {30, 18688});
    WriteU32(w, (uint32_t)frame_h, {8, 0}, {11, 256}, {14, 2304}, {30, 18688});
    const int cw = p.canvas_w ? p.canvas_w : frame_w, ch = p.canvas_h ? p.canvas_h : frame_h;
    partial = p.crop_x0 > 0 || p.crop_y0 > 0 || frame_w + p.crop_x0 < cw || frame_h + p.crop_y0 < ch;
  }
  if (p.frame_type == 0 || p.frame_type == 3) {
    // blending info (+ one per extra channel)
    for (int i = 0; i < 1 + num_extra; i++) {
      if (p.blend_mode < 3) w.put((uint32_t)p.blend_mode, 2); else { w.put(3, 2); w.put((uint32_t)p.blend_mode - 3, 2); }
      if (num_extra > 0 && (p.blend_mode == 2 || p.blend_mode == 3)) w.put(0, 2);   // alpha channel 0
      if (num_extra > 0 && (p.blend_mode == 2 || p.blend_mode == 3 || p.blend_mode == 4)) w.put(p.blend_clamp ? 1 : 0, 1);
      if (p.blend_mode != 0 || partial) w.put((uint32_t)p.blend_source, 2);
    }
    if (g_anim_num > 0) WriteU32(w, (uint32_t)p.duration, {0, 0}, {0, 1}, {8, 0}, {32, 0});   // duration in ticks (no timecodes)
    w.put(p.is_last ? 1 : 0, 1);  // is_last
  }
  const bool is_last = (p.frame_type == 0 || p.frame_type == 3) ? p.is_last != 0 : false;
  if (!is_last && p.frame_type != 1) w.put((uint32_t)p.save_as_reference, 2);
  {
    const int dur = g_anim_num > 0 && (p.frame_type == 0 || p.frame_type == 3) ? p.duration : 0;
    const bool can_ref = !is_last && p.frame_type != 1 && (dur == 0 || p.save_as_reference != 0);
    const bool full_replace = (p.frame_type == 0 || p.frame_type == 3) && p.blend_mode == 0 && !partial;
    if (p.frame_type == 2 || (can_ref && full_replace)) w.put(p.save_before_ct ? 1 : 0, 1);
  }
  w.put(0, 2);  // name length 0
  // RestorationFilter
  const bool custom_lf = CustomFilters() && !modular;
  if (lf_default && !custom_lf) w.put(1, 1);
  else {
    w.put(0, 1);
    w.put(p.gab ? 1 : 0, 1);
    if (p.gab) {
      w.put(custom_lf ? 1 : 0, 1);  // gab_custom
      if (custom_lf) for (float v : {0.15f, 0.08f, 0.12f, 0.05f, 0.1f, 0.07f}) WriteF16(w, v);       // (x, y, b) x (weight of the 4 edge neighbours, of the 4 corners)
    }
    w.put(p.epf_iters, 2);
    if (p.epf_iters > 0) {
      if (!modular) { w.put(custom_lf ? 1 : 0, 1); if (custom_lf) for (float v : {0.0f, 0.1f, 0.3f, 0.4f, 0.6f, 0.7f, 0.9f, 1.1f}) WriteF16(w, v); }   // sharp_custom + LUT
      w.put(custom_lf ? 1 : 0, 1);                // weight_custom: channel scales, two reserved values
      if (custom_lf) for (float v : {35.0f, 6.0f, 3.0f, 0.4f, 0.35f}) WriteF16(w, v);
      w.put(custom_lf ? 1 : 0, 1);                // sigma_custom: quant_mul, pass0 / pass2 sigma scale, border SAD multiplier
      if (custom_lf) for (float v : {0.5f, 0.8f, 7.0f, 0.7f}) WriteF16(w, v);
      if (modular) WriteF16(w, 1.0f);
    }
    WriteU64(w, 0);
  }
  WriteU64(w, 0);  // frame extensions
}

static void WriteTOCAndSections(BitWriter& out, const std::vector<BitWriter>& sections, bool single, uint32_t permute_seed = 0) {
  if (!single && permute_seed) {
    // permuted TOC (toc.cc): logical section i is stored at position perm[i]; here the storage order is a seeded shuffle
    const size_t n = sections.size();
    std::vector<uint32_t> store(n);                 // store[j] = logical index kept at storage position j
    for (size_t i = 0; i < n; i++) store[i] = (uint32_t)i;
    Pcg32 rng(permute_seed);
    for (size_t i = n - 1; i > 0; i--) std::swap(store[i], store[rng.next() % (i + 1)]);
    std::vector<uint32_t> perm(n);
    for (size_t j = 0; j < n; j++) perm[store[j]] = (uint32_t)j;
    std::vector<uint32_t> temp(n), lehmer(n);
    for (size_t i = 0; i < n; i++) temp[i] = (uint32_t)i;
    for (size_t i = 0; i < n; i++) { const auto it = std::find(temp.begin(), temp.end(), perm[i]); lehmer[i] = (uint32_t)(it - temp.begin()); temp.erase(it); }
    size_t end = n;
    while (end > 0 && lehmer[end - 1] == 0) end--;
    auto ctxof = [](uint32_t v) { uint32_t t = 0; while (v) { t++; v >>= 1; } return std::min<uint32_t>(t, 7); };
    std::vector<Token> tok;
    tok.push_back({ctxof((uint32_t)n), (uint32_t)end});
    uint32_t last = 0;
    for (size_t i = 0; i < end; i++) { tok.push_back({ctxof(last), lehmer[i]}); last = lehmer[i]; }
    EntropyCoder code;
    { std::vector<const std::vector<Token>*> ts{&tok}; BuildEntropyCoder(ts, 8, UintConfig{4, 2, 0}, 8, code); }
    out.put(1, 1);
    WriteEntropyCode(out, code);
    EncodeTokens(out, code, tok);
    out.align();
    std::vector<BitWriter> al(n);
    for (size_t j = 0; j < n; j++) { al[j] = sections[store[j]]; al[j].align(); }
    for (auto& s : al) WriteU32(out, (uint32_t)s.bytes.size(), {10, 0}, {14, 1024}, {22, 17408}, {30, 4211712});
    out.align();
    for (auto& s : al) out.bytes.insert(out.bytes.end(), s.bytes.begin(), s.bytes.end());
    return;
  }
  out.put(0, 1);  // not permuted
  out.align();
  if (single) {
    BitWriter all;
    for (auto& s : sections) all.append(s);
    all.align();
    WriteU32(out, (uint32_t)all.bytes.size(), {10, 0}, {14, 1024}, {22, 17408}, {30, 4211712});
    out.align();
    for (uint8_t b : all.bytes) out.bytes.push_back(b);
    return;
  }
  std::vector<BitWriter> al = sections;
  for (auto& s : al) s.align();
  for (auto& s : al) WriteU32(out, (uint32_t)s.bytes.size(), {10, 0}, {14, 1024}, {22, 17408}, {30, 4211712});
  out.align();
  for (auto& s : al) out.bytes.insert(out.bytes.end(), s.bytes.begin(), s.bytes.end());
}

// ---- VarDCT encoder ----------------------------------------------------------------------------------------------
static const uint16_t kFreqCtx[64] = {0xBAD, 0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 15, 16, 16, 17, 17,
                                      18,    18, 19, 19, 20, 20, 21, 21, 22, 22, 23, 23, 23, 23, 24, 24, 24, 24, 25, 25, 25, 25,
                                      26,    26, 26, 26, 27, 27, 27, 27, 28, 28, 28, 28, 29, 29, 29, 29, 30, 30, 30, 30};
static const uint16_t kNzCtx[64] = {0xBAD, 0,   31,  62,  62,  93,  93,  93,  93,  123, 123, 123, 123, 152, 152, 152,
                                    152,   152, 152, 152, 152, 180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 180,
                                    180,   206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206,
                                    206,   206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206};
static const uint8_t kDefaultBlockCtx[39] = {0, 1, 2, 2, 3, 3, 4, 5, 6, 6, 6, 6, 6, 7, 8, 9, 9, 10, 11, 12, 13, 14, 14, 14, 14, 14,
                                             7, 8, 9, 9, 10, 11, 12, 13, 14, 14, 14, 14, 14};

struct SChan { std::vector<int32_t> d; int w, h, hs, vs; };
struct SqStep { bool horizontal, in_place; int begin_c, num_c; };
static std::vector<SqStep> DefaultSqueezeSteps(const std::vector<SChan>& ch);
static void ApplySqueeze(std::vector<SChan>& ch, const std::vector<SqStep>& steps);
// the extra channel of the VarDCT frames written from now on (this thread) goes through the default Squeeze chain — what a cjxl encode of an RGBA picture with a progressive or lossy alpha does
// (its alpha is coded "lossy": squeezed, residuals quantised through the tree's multipliers): sub-channels squeezed by >= 3 ride in the LfGroup sections between the
// LF coefficients and the HF metadata, the others in the PassGroup sections of the last pass, the small ones in GlobalModular
static bool& AlphaSqueeze() { static thread_local bool v = false; return v; }
// number of histogram sets ("HF presets", HfGlobal num_hf_presets) of the VarDCT frames written from now on (this thread): group g of every pass codes its coefficients
// with set g % n — libjxl's encoder clusters the groups of a large picture into several sets
static int& HfPresets() { static thread_local int v = 1; return v; }
// VarDCT frames written from now on (this thread) carry a BlockCtxMap of their own: thresholds on the quantised LF of X / Y / B (1 / 2 / 1 of them, at quantiles of the
// frame) and two on the quantiser field, 39 x 12 x 3 entries onto 16 block contexts (ac_context.h; what libjxl's encoder fits at default effort)
static bool& CustomBlockCtx() { static thread_local bool v = false; return v; }
// VarDCT frames written from now on (this thread) carry their own LfChannelDequantization (1 / 2048, 1 / 256, 1 / 128) and LfChannelCorrelation (colour factor 64, base
// correlations 0.125 / 0.75, LF factors +6 / -10): the encoder side quantises the LF with these steps and predicts X / B from Y with these factors
static bool& CustomLfGlobal() { static thread_local bool v = false; return v; }
// group_size_shift of the Modular frames written from now on (this thread): groups of 128 << shift samples a side (frame_header.cc; 1 = 256 is what VarDCT frames always use)
static int& ModularGroupShift() { static thread_local int v = 1; return v; }
static std::vector<uint8_t> EncodeVarDCT(const float* xyb_planes[3], int w, int h, const Params& p, const uint8_t* alpha = nullptr, int img_w = 0, int img_h = 0) {
  if (img_w == 0) { img_w = w; img_h = h; }   // (w, h) = coded size; (img_w, img_h) = image size when the frame is upsampled
  const int bw = (w + 7) / 8, bh = (h + 7) / 8;
  const int pw = bw * 8, ph = bh * 8;
  const int xg = (w + 255) / 256, yg = (h + 255) / 256, ngroups = xg * yg;
  const int xlg = (w + 2047) / 2048, ylg = (h + 2047) / 2048, nlf = xlg * ylg;
  const int cw = (bw + 7) / 8, chh = (bh + 7) / 8;
  Pcg32 rng(p.seed * 977 + 4);
  // padded planes (edge replicate)
  std::vector<float> pl[3];
  for (int c = 0; c < 3; c++) {
    pl[c].resize((size_t)pw * ph);
    for (int y = 0; y < ph; y++) for (int x = 0; x < pw; x++) pl[c][(size_t)y * pw + x] = xyb_planes[c][(size_t)std::min(y, h - 1) * w + std::min(x, w - 1)];
  }
  // --- strategy map
  std::vector<int8_t> strat((size_t)bw * bh, -1);
  std::vector<uint8_t> first((size_t)bw * bh, 0);
  struct Cand { int s; float prob; };
  std::vector<Cand> cands;
  if (p.strategy_mix == 1) cands = {{S_DCT32X32, 0.05f}, {S_DCT16X16, 0.10f}, {S_DCT16X8, 0.075f}, {S_DCT8X16, 0.075f}, {S_DCT4X8, 0.03f}, {S_DCT8X4, 0.03f}, {S_DCT4X4, 0.04f},
                                     {S_AFV0, 0.006f}, {S_AFV1, 0.006f}, {S_AFV2, 0.006f}, {S_AFV3, 0.006f}};
  else if (p.strategy_mix >= 2 && p.strategy_mix < 100) cands = {{S_DCT64X64, 0.02f}, {S_DCT64X32, 0.01f}, {S_DCT32X64, 0.01f}, {S_DCT32X32, 0.04f}, {S_DCT32X16, 0.02f}, {S_DCT16X32, 0.02f}, {S_DCT32X8, 0.02f},
                                         {S_DCT8X32, 0.02f}, {S_DCT16X16, 0.08f}, {S_DCT16X8, 0.06f}, {S_DCT8X16, 0.06f}, {S_DCT4X8, 0.03f}, {S_DCT8X4, 0.03f}, {S_DCT4X4, 0.03f},
                                         {S_DCT2X2, 0.02f}, {S_IDENTITY, 0.02f}, {S_AFV0, 0.01f}, {S_AFV1, 0.01f}, {S_AFV2, 0.01f}, {S_AFV3, 0.01f}};
  if (p.strategy_mix == 4 || p.strategy_mix == 5) {  // + the DCT128/256 family (5: also at unaligned positions)
    const std::vector<Cand> big = {{24, 0.15f}, {25, 0.1f}, {26, 0.1f}, {21, 0.1f}, {22, 0.05f}, {23, 0.05f}};
    cands.insert(cands.begin(), big.begin(), big.end());
  }
  const bool unaligned = p.strategy_mix == 3 || p.strategy_mix == 5;
  if (p.strategy_mix >= 100) cands = {{p.strategy_mix - 100, 1.0f}};  // force one strategy wherever it fits
  for (int by = 0; by < bh; by++) for (int bx = 0; bx < bw; bx++) {
    if (strat[(size_t)by * bw + bx] >= 0) continue;
    int chosen = S_DCT;
    for (auto& c : cands) {   // largest first; a candidate that fits here is taken with its target area share
      int cx = kCovX[c.s], cy = kCovY[c.s];
      bool ok = (unaligned || ((bx % cx == 0) && (by % cy == 0))) && bx + cx <= bw && by + cy <= bh && (bx % 32) + cx <= 32 && (by % 32) + cy <= 32;
      for (int iy = 0; ok && iy < cy; iy++) for (int ix = 0; ix < cx; ix++) if (strat[(size_t)(by + iy) * bw + bx + ix] >= 0) ok = false;
      if (!ok) continue;
      float q = unaligned ? std::max(c.prob / (float)(cx * cy) * 2.0f, cx * cy >= 128 ? 0.002f : 0.0f) : c.prob;
      if (p.strategy_mix >= 100) q = 1.0f;
      if (rng.uniform() < q) { chosen = c.s; break; }
    }
    int cx = kCovX[chosen], cy = kCovY[chosen];
    for (int iy = 0; iy < cy; iy++) for (int ix = 0; ix < cx; ix++) strat[(size_t)(by + iy) * bw + bx + ix] = (int8_t)chosen;
    first[(size_t)by * bw + bx] = 1;
  }
  // --- quantisation parameters
  const uint32_t global_scale = (uint32_t)std::min(65535.0f, std::max(1.0f, 4587.0f / p.distance));
  const uint32_t quant_lf = (uint32_t)QuantLf();
  const float inv_gs = 65536.0f / (float)global_scale;
  const bool custom_lfg = CustomLfGlobal();
  const float m_lf[3] = {custom_lfg ? 1.0f / 2048 : 1.0f / 4096, custom_lfg ? 1.0f / 256 : 1.0f / 512, custom_lfg ? 1.0f / 128 : 1.0f / 256};
  const float cfl_factor = custom_lfg ? 64.0f : 84.0f, cfl_base_x = custom_lfg ? 0.125f : 0.0f, cfl_base_b = custom_lfg ? 0.75f : 1.0f;
  const int cfl_x_lf = custom_lfg ? 6 : 0, cfl_b_lf = custom_lfg ? -10 : 0;
  const float x_dm = std::pow(0.8f, (float)(QmScales()[0] - 2)), b_dm = std::pow(0.8f, (float)(QmScales()[1] - 2));  // x_qm_scale 3, b_qm_scale 2 by default
  std::vector<int32_t> hf_mul((size_t)bw * bh, 1);
  std::vector<int32_t> sharp((size_t)bw * bh, 0);
  for (int by = 0; by < bh; by++) for (int bx = 0; bx < bw; bx++) {
    size_t o = (size_t)by * bw + bx;
    if (first[o]) {
      int q = 14 + (int)(rng.next() % 12);
      int s = strat[o];
      for (int iy = 0; iy < kCovY[s]; iy++) for (int ix = 0; ix < kCovX[s]; ix++) hf_mul[o + (size_t)iy * bw + ix] = q;
    }
    uint32_t r = rng.next() % 100;
    sharp[o] = r < 10 ? 0 : r < 70 ? 4 : (int)(r % 8);
  }
  // quant tables for used kinds
  bool used_kind[17] = {false};
  for (size_t o = 0; o < strat.size(); o++) used_kind[kKind[strat[o]]] = true;
  QuantSpec specs[17];
  std::vector<float> table[17][3];
  for (int k = 0; k < 17; k++) {
    if (!used_kind[k]) { specs[k].mode = 0; continue; }
    specs[k] = DefaultSpec(k);
    for (int c = 0; c < 3; c++) ComputeTable(specs[k], k, c, table[k][c]);
  }
  // --- forward transforms, LF extraction
  std::vector<float> lf[3];
  for (int c = 0; c < 3; c++) lf[c].assign((size_t)bw * bh, 0.f);
  // coefficient storage per varblock: offset table
  std::vector<size_t> coff((size_t)bw * bh, 0);
  size_t total_coef = 0;
  for (size_t o = 0; o < strat.size(); o++) if (first[o]) { coff[o] = total_coef; total_coef += (size_t)kCovX[strat[o]] * kCovY[strat[o]] * 64; }
  std::vector<float> coef[3];
  for (int c = 0; c < 3; c++) coef[c].assign(total_coef, 0.f);
  for (int by = 0; by < bh; by++) for (int bx = 0; bx < bw; bx++) {
    size_t o = (size_t)by * bw + bx;
    if (!first[o]) continue;
    int s = strat[o];
    for (int c = 0; c < 3; c++) {
      float* cf = coef[c].data() + coff[o];
      ForwardTransform(s, pl[c].data() + (size_t)by * 8 * pw + bx * 8, pw, cf);
      LFFromLowestFrequencies(s, cf, lf[c].data() + o, bw);
    }
  }
  // --- quantise LF (Y first; X and B coded relative to dequantised Y: default cfl for LF = (0, 1))
  std::vector<int32_t> lfq[3];
  for (int c = 0; c < 3; c++) lfq[c].assign((size_t)bw * bh, 0);
  float lfstep[3];
  for (int c = 0; c < 3; c++) lfstep[c] = m_lf[c] * inv_gs / (float)quant_lf / (float)(1 << LfExtraPrecision());
  for (size_t o = 0; o < (size_t)bw * bh; o++) {
    int32_t qy = (int32_t)std::lrintf(lf[1][o] / lfstep[1]);
    float dy = qy * lfstep[1];
    lfq[1][o] = qy;
    lfq[0][o] = (int32_t)std::lrintf((lf[0][o] - (cfl_base_x + (float)cfl_x_lf / cfl_factor) * dy) / lfstep[0]);
    lfq[2][o] = (int32_t)std::lrintf((lf[2][o] - (cfl_base_b + (float)cfl_b_lf / cfl_factor) * dy) / lfstep[2]);
  }
  // --- chroma-from-luma factors per 64x64 tile: X uses 0, B least-squares around base 1.0
  std::vector<int32_t> ytox((size_t)cw * chh, 0), ytob((size_t)cw * chh, 0);
  {
    std::vector<double> num((size_t)cw * chh, 0.0), den((size_t)cw * chh, 0.0), numx((size_t)cw * chh, 0.0);
    for (int by = 0; by < bh; by++) for (int bx = 0; bx < bw; bx++) {
      size_t o = (size_t)by * bw + bx;
      if (!first[o]) continue;
      int s = strat[o];
      size_t n = (size_t)kCovX[s] * kCovY[s] * 64;
      size_t tile = (size_t)(by / 8) * cw + bx / 8;
      const float* y = coef[1].data() + coff[o]; const float* b = coef[2].data() + coff[o];
      const float* xc = coef[0].data() + coff[o];
      for (size_t k = 1; k < n; k++) { num[tile] += (double)y[k] * (b[k] - cfl_base_b * y[k]); numx[tile] += (double)y[k] * (xc[k] - cfl_base_x * y[k]); den[tile] += (double)y[k] * y[k]; }
    }
    for (size_t t = 0; t < num.size(); t++) {
      double f = den[t] > 1e-12 ? num[t] / den[t] : 0.0;
      int v = (int)std::lrint(f * (double)cfl_factor);
      ytob[t] = std::max(-128, std::min(127, v));
      // (X map: zero unless the frame carries its own colour-correlation parameters; then fitted like the B map, and never left at zero so that the term is exercised)
      ytox[t] = 0;
      if (custom_lfg) { const int vx = (int)std::lrint((den[t] > 1e-12 ? numx[t] / den[t] : 0.0) * (double)cfl_factor); ytox[t] = std::max(-128, std::min(127, vx == 0 ? (t % 2 ? 3 : -2) : vx)); }
    }
  }
  // --- quantise AC
  std::vector<int32_t> qc[3];
  for (int c = 0; c < 3; c++) qc[c].assign(total_coef, 0);
  const float bias[4] = {1.0f - 0.05465007330715401f, 1.0f - 0.07005449891748593f, 1.0f - 0.049935103337343655f, 0.145f};
  auto quant = [](float v) -> int32_t {
    float a = std::fabs(v);
    if (a < 0.58f) return 0;
    int32_t q = (int32_t)(a + 0.5f);
    return v < 0 ? -q : q;
  };
  for (int by = 0; by < bh; by++) for (int bx = 0; bx < bw; bx++) {
    size_t o = (size_t)by * bw + bx;
    if (!first[o]) continue;
    int s = strat[o], kind = kKind[s];
    int cx = kCovX[s], cy = kCovY[s];
    size_t n = (size_t)cx * cy * 64;
    float sd = inv_gs / (float)hf_mul[o];
    float sdc[3] = {sd * x_dm, sd, sd * b_dm};
    size_t tile = (size_t)(by / 8) * cw + bx / 8;
    float kx = cfl_base_x + ytox[tile] / cfl_factor, kb = cfl_base_b + ytob[tile] / cfl_factor;
    const float* ty = table[kind][1].data(); const float* tx = table[kind][0].data(); const float* tb = table[kind][2].data();
    float* fy = coef[1].data() + coff[o]; float* fx = coef[0].data() + coff[o]; float* fb = coef[2].data() + coff[o];
    int32_t* qy = qc[1].data() + coff[o]; int32_t* qx = qc[0].data() + coff[o]; int32_t* qb = qc[2].data() + coff[o];
    // LLF slots are not coded: zero them (they are positions (v<cy,u<cx) of the stored layout)
    std::vector<uint8_t> is_llf(n, 0);
    { int R = 8 * cy, C = 8 * cx; for (int v = 0; v < cy; v++) for (int u = 0; u < cx; u++) is_llf[StoredIdx(R, C, v, u)] = 1; }
    for (size_t k = 0; k < n; k++) {
      if (is_llf[k]) continue;
      int32_t q = quant(fy[k] / (ty[k] * sdc[1]));
      qy[k] = q;
      float adj = q == 0 ? 0.f : (q == 1 ? bias[1] : q == -1 ? -bias[1] : (float)q - bias[3] / (float)q);
      float dy = adj * (ty[k] * sdc[1]);
      qx[k] = quant((fx[k] - kx * dy) / (tx[k] * sdc[0]));
      qb[k] = quant((fb[k] - kb * dy) / (tb[k] * sdc[2]));
    }
  }
  // --- tokens: modular streams (LF coefficients + HF metadata) under the global tree
  std::vector<int> bfs;
  GTree gt = MakeGlobalTree(nlf, &bfs);
  const int root = bfs[0];
  std::vector<Token> tree_tokens;
  TreeTokens(gt, bfs, tree_tokens);
  struct LfGroupData { std::vector<int32_t> ch[3]; std::vector<int32_t> m[4]; int gbw, gbh, nb; std::vector<Token> lf_tok, meta_tok; };
  std::vector<LfGroupData> lgd(nlf);
  for (int g = 0; g < nlf; g++) {
    LfGroupData& d = lgd[g];
    int gx = g % xlg, gy = g / xlg, bx0 = gx * 256, by0 = gy * 256;
    d.gbw = std::min(256, bw - bx0); d.gbh = std::min(256, bh - by0);
    static const int order[3] = {1, 0, 2};  // Y, X, B
    for (int i = 0; i < 3; i++) {
      d.ch[i].resize((size_t)d.gbw * d.gbh);
      for (int y = 0; y < d.gbh; y++) for (int x = 0; x < d.gbw; x++) d.ch[i][(size_t)y * d.gbw + x] = lfq[order[i]][(size_t)(by0 + y) * bw + bx0 + x];
    }
    std::vector<ChanRef> cr;
    for (int i = 0; i < 3; i++) cr.push_back({d.ch[i].data(), d.gbw, d.gbh});
    ModularTokens(gt, root, cr, 1 + g, d.lf_tok, LfWpParams());
    // HF metadata
    int mcw = (d.gbw + 7) / 8, mch = (d.gbh + 7) / 8;
    d.m[0].resize((size_t)mcw * mch); d.m[1].resize((size_t)mcw * mch);
    for (int y = 0; y < mch; y++) for (int x = 0; x < mcw; x++) {
      d.m[0][(size_t)y * mcw + x] = ytox[(size_t)(gy * 32 + y) * cw + gx * 32 + x];
      d.m[1][(size_t)y * mcw + x] = ytob[(size_t)(gy * 32 + y) * cw + gx * 32 + x];
    }
    std::vector<int32_t> st, hm;
    for (int y = 0; y < d.gbh; y++) for (int x = 0; x < d.gbw; x++) {
      size_t o = (size_t)(by0 + y) * bw + bx0 + x;
      if (first[o]) { st.push_back(strat[o]); hm.push_back(hf_mul[o] - 1); }
    }
    d.nb = (int)st.size();
    d.m[2] = st; d.m[2].insert(d.m[2].end(), hm.begin(), hm.end());
    d.m[3].resize((size_t)d.gbw * d.gbh);
    for (int y = 0; y < d.gbh; y++) for (int x = 0; x < d.gbw; x++) d.m[3][(size_t)y * d.gbw + x] = sharp[(size_t)(by0 + y) * bw + bx0 + x];
    std::vector<ChanRef> mr{{d.m[0].data(), mcw, mch}, {d.m[1].data(), mcw, mch}, {d.m[2].data(), d.nb, 2}, {d.m[3].data(), d.gbw, d.gbh}};
    ModularTokens(gt, root, mr, 1 + 2 * nlf + g, d.meta_tok, LfWpParams());
  }
  // --- tokens: AC per group
  const bool custom_bcm = CustomBlockCtx();
  std::vector<int32_t> lf_thr[3];
  std::vector<uint32_t> qf_thr;
  std::vector<uint8_t> bcm_map;
  int num_lf_ctxs = 1;
  if (custom_bcm) {
    auto quantile = [&](int c, double q) { std::vector<int32_t> v(lfq[c]); std::sort(v.begin(), v.end()); return v[(size_t)(q * (double)(v.size() - 1))]; };
    lf_thr[0] = {quantile(0, 0.5)}; lf_thr[1] = {quantile(1, 0.33), quantile(1, 0.66)}; lf_thr[2] = {quantile(2, 0.5)};
    if (lf_thr[1][1] <= lf_thr[1][0]) lf_thr[1][1] = lf_thr[1][0] + 1;       // (strictly increasing is not required by the format; kept tidy)
    qf_thr = {17, 21};
    num_lf_ctxs = 2 * 3 * 2;
    bcm_map.resize((size_t)39 * num_lf_ctxs * 3);
    for (size_t i = 0; i < bcm_map.size(); i++) {
      const int lf_idx = (int)(i % num_lf_ctxs), qf_idx = (int)((i / num_lf_ctxs) % 3), co = (int)(i / num_lf_ctxs / 3);
      bcm_map[i] = (uint8_t)((kDefaultBlockCtx[co] + 5 * lf_idx + 3 * qf_idx) % 16);
    }
  }
  const int nctx = custom_bcm ? 16 : 15;
  const int np = p.num_passes;
  // per-pass coefficient planes: pass p carries (remainder >> shift[p]); the decoder adds value << shift
  std::vector<std::vector<int32_t>> qpass[3];
  for (int c = 0; c < 3; c++) {
    qpass[c].assign(np, std::vector<int32_t>());
    std::vector<int32_t> rem = qc[c];
    for (int ps = 0; ps < np; ps++) {
      const int sh = PassShift(np, ps);
      qpass[c][ps].resize(rem.size());
      for (size_t i = 0; i < rem.size(); i++) { const int32_t a = rem[i] >> sh; qpass[c][ps][i] = a; rem[i] -= a * (1 << sh); }
    }
  }
  std::vector<std::vector<Token>> ac_tok_all((size_t)ngroups * np);
  for (int ps = 0; ps < np; ps++) {
  std::vector<Token>* ac_tok = &ac_tok_all[(size_t)ps * ngroups];
  std::vector<uint32_t> natural[13];
  static const int bucket_rep[13] = {S_DCT, S_IDENTITY, S_DCT16X16, S_DCT32X32, S_DCT16X8, S_DCT32X8, S_DCT32X16, S_DCT64X64, S_DCT64X32, 21, 22, 24, 25};
  for (int b = 0; b < 13; b++) natural[b] = NaturalOrder(bucket_rep[b]);
  for (int g = 0; g < ngroups; g++) {
    int gx = g % xg, gy = g / xg, bx0 = gx * 32, by0 = gy * 32;
    int gbw = std::min(32, bw - bx0), gbh = std::min(32, bh - by0);
    uint8_t nzmap[3][1024];
    memset(nzmap, 0, sizeof(nzmap));
    std::vector<Token>& tk = ac_tok[g];
    for (int by = 0; by < gbh; by++) for (int bx = 0; bx < gbw; bx++) {
      size_t o = (size_t)(by0 + by) * bw + bx0 + bx;
      if (!first[o]) continue;
      int s = strat[o], cx = kCovX[s], cy = kCovY[s], covered = cx * cy, l2 = ILog2(covered), size = covered * 64, ord = kBucket[s];
      const std::vector<uint32_t>& order = natural[ord];
      static const int chan[3] = {1, 0, 2};
      for (int ci = 0; ci < 3; ci++) {
        int c = chan[ci];
        int idx = (c < 2 ? (c ^ 1) : 2) * 13 + ord;
        int block_ctx = kDefaultBlockCtx[idx];
        if (custom_bcm) {
          int lf_idx = 0;
          if (!p.use_lf_frame) {       // (a frame whose LF comes from an LF frame has no quantised LF of its own: index 0)
            int b3[3] = {0, 0, 0};
            for (int cc = 0; cc < 3; cc++) for (int32_t t : lf_thr[cc]) if (lfq[cc][o] > t) b3[cc]++;
            lf_idx = (b3[0] * ((int)lf_thr[2].size() + 1) + b3[2]) * ((int)lf_thr[1].size() + 1) + b3[1];
          }
          int qf_idx = 0;
          for (uint32_t t : qf_thr) if ((uint32_t)hf_mul[o] > t) qf_idx++;
          block_ctx = bcm_map[((size_t)idx * (qf_thr.size() + 1) + qf_idx) * num_lf_ctxs + lf_idx];
        }
        const int32_t* q = qpass[c][ps].data() + coff[o];
        int nz = 0;
        for (int k = covered; k < size; k++) nz += q[order[k]] != 0;
        int pred;
        if (bx == 0) pred = by == 0 ? 32 : nzmap[c][(by - 1) * 32 + bx];
        else if (by == 0) pred = nzmap[c][by * 32 + bx - 1];
        else pred = (nzmap[c][(by - 1) * 32 + bx] + nzmap[c][by * 32 + bx - 1] + 1) / 2;
        int pc = std::min(pred, 64);
        uint32_t nzctx = pc < 8 ? block_ctx + nctx * pc : block_ctx + nctx * (4 + pc / 2);
        tk.push_back({nzctx, (uint32_t)nz});
        uint8_t nzm = (uint8_t)((nz + covered - 1) >> l2);
        for (int iy = 0; iy < cy; iy++) for (int ix = 0; ix < cx; ix++) nzmap[c][(by + iy) * 32 + bx + ix] = nzm;
        uint32_t histo = 37 * nctx + 458 * block_ctx;
        int left = nz;
        uint32_t prev = nz > size / 16 ? 0 : 1;
        for (int k = covered; k < size && left != 0; k++) {
          uint32_t nzl = (left + covered - 1) >> l2, kk = (uint32_t)k >> l2;
          uint32_t ctx = histo + (kNzCtx[nzl] + kFreqCtx[kk]) * 2 + prev;
          uint32_t u = PackSigned(q[order[k]]);
          tk.push_back({ctx, u});
          prev = u != 0;
          left -= prev;
        }
      }
    }
  }
  }  // passes
  // --- optional alpha: one 8-bit extra channel coded losslessly by the frame's Modular sub-streams (GlobalModular when
  // the image fits one group, else the modular part of every PassGroup), under the same global tree
  std::vector<Token> alpha_global_tok;
  std::vector<std::vector<Token>> alpha_tok(ngroups), alpha_lf_tok;
  std::vector<char> alpha_lf_has, alpha_has;          // squeezed alpha: which LfGroup / PassGroup sections carry a Modular sub-stream
  const bool alpha_sq = alpha && AlphaSqueeze();
  const bool alpha_global = alpha && w <= 256 && h <= 256;
  if (alpha) {
    std::vector<int32_t> a32((size_t)w * h);
    for (size_t i = 0; i < a32.size(); i++) a32[i] = alpha[i];
    if (AlphaSqueeze()) {
      std::vector<SChan> ach(1);
      ach[0].d = a32; ach[0].w = w; ach[0].h = h; ach[0].hs = ach[0].vs = 0;
      ApplySqueeze(ach, DefaultSqueezeSteps(ach));
      const int nfinal = (int)ach.size();
      int nglobal = 0;
      while (nglobal < nfinal && ach[nglobal].w <= 256 && ach[nglobal].h <= 256) nglobal++;
      {
        std::vector<ChanRef> cr;
        for (int c = 0; c < nglobal; c++) if (ach[c].w && ach[c].h) cr.push_back({ach[c].d.data(), ach[c].w, ach[c].h});
        ModularTokens(gt, root, cr, 0, alpha_global_tok);
      }
      auto group_stream = [&](int x0, int y0, int dim, int min_shift, int max_shift, int stream_id, std::vector<Token>& out) -> bool {
        std::vector<std::vector<int32_t>> store;
        std::vector<ChanRef> cr;
        for (int c = nglobal; c < nfinal; c++) {
          const SChan& sc = ach[c];
          if (!sc.w || !sc.h) continue;
          const int shift = std::min(sc.hs, sc.vs);
          if (shift < min_shift || shift > max_shift) continue;
          const int rx = x0 >> sc.hs, ry = y0 >> sc.vs;
          if (rx >= sc.w || ry >= sc.h) continue;
          const int rw = std::min(dim >> sc.hs, sc.w - rx), rh = std::min(dim >> sc.vs, sc.h - ry);
          if (rw <= 0 || rh <= 0) continue;
          store.emplace_back((size_t)rw * rh);
          for (int y = 0; y < rh; y++) memcpy(&store.back()[(size_t)y * rw], &sc.d[(size_t)(ry + y) * sc.w + rx], sizeof(int32_t) * rw);
          cr.push_back({nullptr, rw, rh});
        }
        for (size_t i = 0; i < cr.size(); i++) cr[i].d = store[i].data();
        if (cr.empty()) return false;
        ModularTokens(gt, root, cr, stream_id, out);
        return true;
      };
      alpha_lf_tok.resize(nlf); alpha_lf_has.assign(nlf, 0); alpha_has.assign(ngroups, 0);
      for (int g = 0; g < nlf; g++) alpha_lf_has[g] = group_stream((g % xlg) * 2048, (g / xlg) * 2048, 2048, 3, 1000, 1 + nlf + g, alpha_lf_tok[g]);
      for (int g = 0; g < ngroups; g++) alpha_has[g] = group_stream((g % xg) * 256, (g / xg) * 256, 256, 0, 2, 1 + 3 * nlf + 17 + ngroups * (p.num_passes - 1) + g, alpha_tok[g]);
    } else if (alpha_global) {
      std::vector<ChanRef> cr{{a32.data(), w, h}};
      ModularTokens(gt, root, cr, 0, alpha_global_tok);
    } else {
      for (int g = 0; g < ngroups; g++) {
        const int x0 = (g % xg) * 256, y0 = (g / xg) * 256, gw = std::min(256, w - x0), gh = std::min(256, h - y0);
        std::vector<int32_t> rect((size_t)gw * gh);
        for (int y = 0; y < gh; y++) memcpy(&rect[(size_t)y * gw], &a32[(size_t)(y0 + y) * w + x0], sizeof(int32_t) * gw);
        std::vector<ChanRef> cr{{rect.data(), gw, gh}};
        ModularTokens(gt, root, cr, 1 + 3 * nlf + 17 + ngroups * (p.num_passes - 1) + g, alpha_tok[g]);
      }
      alpha_has.assign(ngroups, 1);
    }
  }
  // --- entropy codes
  EntropyCoder tree_code, mod_code;
  std::vector<EntropyCoder> ac_codes(np);
  { std::vector<const std::vector<Token>*> s{&tree_tokens}; BuildEntropyCoder(s, 6, UintConfig{4, 2, 0}, 6, tree_code); }
  const bool lz77_lf = UseLz77Lf();
  if (lz77_lf) {
    EntropyCoder proto;
    proto.lz_min_symbol = 224; proto.lz_min_length = 3; proto.lz_len_cfg = UintConfig{3, 0, 0};
    for (auto& d : lgd) { ApplyLz77(d.lf_tok, (uint32_t)gt.num_leaves, proto, (size_t)d.gbw); ApplyLz77(d.meta_tok, (uint32_t)gt.num_leaves, proto, 0); }
  }
  { std::vector<const std::vector<Token>*> s; for (auto& d : lgd) { s.push_back(&d.lf_tok); s.push_back(&d.meta_tok); }
    s.push_back(&alpha_global_tok); for (auto& t : alpha_tok) s.push_back(&t); for (auto& t : alpha_lf_tok) s.push_back(&t);
    BuildEntropyCoder(s, gt.num_leaves + (lz77_lf ? 1 : 0), UintConfig{4, 2, 0}, 32, mod_code);
    if (lz77_lf) { mod_code.lz77 = true; mod_code.lz_min_symbol = 224; mod_code.lz_min_length = 3; mod_code.lz_len_cfg = UintConfig{3, 0, 0}; } }
  const bool lz77_ac = UseLz77Ac();
  const int npresets = std::max(1, std::min(HfPresets(), ngroups));
  for (int ps = 0; ps < np; ps++) {
    if (npresets > 1)      // histogram set g % npresets: its contexts follow those of the sets before it (dec_group.cc: context offset = histo_selector * num AC contexts)
      for (int g = 0; g < ngroups; g++) for (Token& t : ac_tok_all[(size_t)ps * ngroups + g]) t.ctx += (uint32_t)((g % npresets) * 495 * nctx);
    if (lz77_ac) {   // LZ77 over every group's coefficient stream of the pass (dec_group.cc reads them with a reader without distance multiplier)
      EntropyCoder proto;
      proto.lz_min_symbol = 224; proto.lz_min_length = 3; proto.lz_len_cfg = UintConfig{3, 0, 0};
      for (int g = 0; g < ngroups; g++) ApplyLz77(ac_tok_all[(size_t)ps * ngroups + g], (uint32_t)(495 * nctx * npresets), proto, 0, /*special=*/false);
    }
    std::vector<const std::vector<Token>*> s; for (int g = 0; g < ngroups; g++) s.push_back(&ac_tok_all[(size_t)ps * ngroups + g]);
    BuildEntropyCoder(s, 495 * nctx * npresets + (lz77_ac ? 1 : 0), UintConfig{4, 2, 0}, 96, ac_codes[ps]);
    if (lz77_ac) { ac_codes[ps].lz77 = true; ac_codes[ps].lz_min_symbol = 224; ac_codes[ps].lz_min_length = 3; ac_codes[ps].lz_len_cfg = UintConfig{3, 0, 0}; }
  }
  // --- sections
  std::vector<BitWriter> sections;
  {  // LfGlobal
    BitWriter s;
    WriteFeatures(s);                                                             // patch dictionary, splines
    if (p.noise) for (int i = 0; i < 8; i++) s.put(p.noise_lut[i] & 1023, 10);   // NoiseParams
    if (!custom_lfg) s.put(1, 1);  // LfChannelDequantization all_default
    else { s.put(0, 1); for (int c = 0; c < 3; c++) WriteF16(s, m_lf[c] * 128.0f); }
    WriteU32(s, global_scale, {11, 1}, {11, 2049}, {12, 4097}, {16, 8193});
    WriteU32(s, quant_lf, {0, 16}, {5, 1}, {8, 1}, {16, 1});
    if (!custom_bcm) s.put(1, 1);  // default BlockCtxMap
    else {
      s.put(0, 1);
      for (int c = 0; c < 3; c++) {
        s.put((uint32_t)lf_thr[c].size(), 4);
        for (int32_t t : lf_thr[c]) WriteU32(s, PackSigned(t), {4, 0}, {8, 16}, {16, 272}, {32, 65808});
      }
      s.put((uint32_t)qf_thr.size(), 4);
      for (uint32_t t : qf_thr) WriteU32(s, t - 1, {2, 0}, {3, 4}, {5, 12}, {8, 44});
      WriteContextMap(s, bcm_map, 16);
    }
    if (!custom_lfg) s.put(1, 1);  // default LfChannelCorrelation
    else {
      s.put(0, 1);
      WriteU32(s, (uint32_t)cfl_factor, {0, 84}, {0, 256}, {8, 2}, {16, 258});
      WriteF16(s, cfl_base_x); WriteF16(s, cfl_base_b);
      s.put((uint32_t)(cfl_x_lf + 128), 8); s.put((uint32_t)(cfl_b_lf + 128), 8);
    }
    s.put(1, 1);  // GlobalModular: has_tree
    WriteEntropyCode(s, tree_code);
    EncodeTokens(s, tree_code, tree_tokens);
    WriteEntropyCode(s, mod_code);
    if (alpha) {  // the global Modular image has a channel: GroupHeader + whatever is decodable globally
      s.put(1, 1); s.put(1, 1);
      if (alpha_sq) { s.put(1, 2); s.put(2, 2); WriteU32(s, 0, {0, 0}, {4, 1}, {6, 9}, {8, 41}); }    // one transform: Squeeze with the default chain (zero explicit steps)
      else s.put(0, 2);
      EncodeTokens(s, mod_code, alpha_global_tok);
    }
    sections.push_back(s);
  }
  for (int g = 0; g < nlf; g++) {  // LfGroup
    BitWriter s;
    LfGroupData& d = lgd[g];
    if (!p.use_lf_frame) {
      s.put((uint32_t)LfExtraPrecision(), 2);  // extra_precision
      WriteGroupHeaderLf(s);
      EncodeTokens(s, mod_code, d.lf_tok);
    }
    // ModularLfGroup: the extra channel's sub-channels squeezed by >= 3 in both directions (none without Squeeze: nothing is written then)
    if (alpha_sq && alpha_lf_has[g]) { s.put(1, 1); s.put(1, 1); s.put(0, 2); EncodeTokens(s, mod_code, alpha_lf_tok[g]); }
    s.put(d.nb - 1, CeilLog2((uint32_t)(d.gbw * d.gbh)));
    WriteGroupHeaderLf(s);
    EncodeTokens(s, mod_code, d.meta_tok);
    sections.push_back(s);
  }
  {  // HfGlobal
    BitWriter s;
    s.put(0, 1);  // dequant matrices not all default
    for (int k = 0; k < 17; k++) {
      const QuantSpec& q = specs[k];
      s.put(q.mode, 3);
      auto write_bands = [&](const Bands& b) {
        s.put(b.n - 1, 4);
        for (int c = 0; c < 3; c++) for (int i = 0; i < b.n; i++) WriteF16(s, i == 0 ? b.v[c][i] / 64.0f : b.v[c][i]);
      };
      switch (q.mode) {
        case 0: break;
        case 1: for (int c = 0; c < 3; c++) for (int i = 0; i < 3; i++) WriteF16(s, q.idw[c][i] / 64.0f); break;
        case 2: for (int c = 0; c < 3; c++) for (int i = 0; i < 6; i++) WriteF16(s, q.dct2w[c][i] / 64.0f); break;
        case 3: for (int c = 0; c < 3; c++) for (int i = 0; i < 2; i++) WriteF16(s, q.dct4mul[c][i]); write_bands(q.dct); break;
        case 4: for (int c = 0; c < 3; c++) WriteF16(s, q.dct4x8mul[c]); write_bands(q.dct); break;
        case 5: for (int c = 0; c < 3; c++) for (int i = 0; i < 9; i++) WriteF16(s, i < 6 ? q.afvw[c][i] / 64.0f : q.afvw[c][i]); write_bands(q.dct); write_bands(q.dct4x4); break;
        case 6: write_bands(q.dct); break;
      }
    }
    s.put((uint32_t)(npresets - 1), CeilLog2((uint32_t)ngroups));  // num_hf_presets - 1
    for (int ps = 0; ps < np; ps++) {       // HfPass: natural orders, one entropy code per pass
      s.put(2, 2);                          // used_orders = Val(0)
      WriteEntropyCode(s, ac_codes[ps]);
    }
    sections.push_back(s);
  }
  for (int ps = 0; ps < np; ps++) for (int g = 0; g < ngroups; g++) {  // PassGroup, pass-major
    BitWriter s;
    s.put((uint32_t)(g % npresets), CeilLog2((uint32_t)npresets));   // which histogram set (0 bits when there is one)
    EncodeTokens(s, ac_codes[ps], ac_tok_all[(size_t)ps * ngroups + g]);
    // extra channels (shift 0..2) ride in the last pass (Passes::GetDownsamplingBracket without downsampling entries)
    if (alpha && (alpha_sq || !alpha_global) && ps == np - 1 && alpha_has[g]) { s.put(1, 1); s.put(1, 1); s.put(0, 2); EncodeTokens(s, mod_code, alpha_tok[g]); }
    sections.push_back(s);
  }
  BitWriter out;
  const bool hdr_alpha = p.num_extra_hdr >= 0 ? p.num_extra_hdr > 0 : alpha != nullptr;
  if (p.emit != 1) WriteImageHeader(out, p.canvas_w ? p.canvas_w : img_w, p.canvas_h ? p.canvas_h : img_h, p, true, p.out_bits == 16 ? 16 : 8, hdr_alpha, false);
  if (p.emit == 2) return out.bytes;
  bool lf_default = p.gab == 1 && p.epf_iters == 2;
  WriteFrameHeader(out, p, false, true, alpha ? 1 : 0, 1, lf_default, img_w, img_h);
  WriteTOCAndSections(out, sections, ngroups == 1 && np == 1, (uint32_t)p.permute_toc);
  out.align();
  return out.bytes;
}

// ---- Modular lossless encoder (gradient predictor, fixed global tree, 256x256 groups, optional RCT + Squeeze) -----

// ISO/IEC 18181-1 squeeze "smooth tendency" (the decoder adds it back, so both sides must agree exactly)
static int64_t SqTendency(int64_t B, int64_t a, int64_t n) {
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
static void FwdSqueeze(const SChan& in, bool horizontal, SChan& avg, SChan& res) {
  const int w = in.w, h = in.h;
  if (horizontal) {
    avg = SChan{{}, (w + 1) / 2, h, in.hs + 1, in.vs};
    res = SChan{{}, w - (w + 1) / 2, h, in.hs + 1, in.vs};
    avg.d.resize((size_t)avg.w * h); res.d.resize((size_t)res.w * h);
    for (int y = 0; y < h; y++) {
      const int32_t* p = &in.d[(size_t)y * w];
      int32_t* pa = avg.d.data() + (size_t)y * avg.w;
      int32_t* pr = res.d.data() + (size_t)y * res.w;
      for (int x = 0; x < res.w; x++) { int64_t A = p[2 * x], B = p[2 * x + 1]; pa[x] = (int32_t)((A + B + (A > B)) >> 1); }
      if (avg.w > res.w) pa[res.w] = p[w - 1];
      for (int x = 0; x < res.w; x++) {
        int64_t a = pa[x], next = x + 1 < avg.w ? pa[x + 1] : a, left = x ? p[2 * x - 1] : a;
        pr[x] = (int32_t)((int64_t)p[2 * x] - p[2 * x + 1] - SqTendency(left, a, next));
      }
    }
  } else {
    avg = SChan{{}, w, (h + 1) / 2, in.hs, in.vs + 1};
    res = SChan{{}, w, h - (h + 1) / 2, in.hs, in.vs + 1};
    avg.d.resize((size_t)w * avg.h); res.d.resize((size_t)w * res.h);
    for (int y = 0; y < res.h; y++)
      for (int x = 0; x < w; x++) { int64_t A = in.d[(size_t)(2 * y) * w + x], B = in.d[(size_t)(2 * y + 1) * w + x]; avg.d[(size_t)y * w + x] = (int32_t)((A + B + (A > B)) >> 1); }
    if (avg.h > res.h) for (int x = 0; x < w; x++) avg.d[(size_t)res.h * w + x] = in.d[(size_t)(h - 1) * w + x];
    for (int y = 0; y < res.h; y++)
      for (int x = 0; x < w; x++) {
        int64_t a = avg.d[(size_t)y * w + x], next = y + 1 < avg.h ? avg.d[(size_t)(y + 1) * w + x] : a, top = y ? in.d[(size_t)(2 * y - 1) * w + x] : a;
        res.d[(size_t)y * w + x] = (int32_t)((int64_t)in.d[(size_t)(2 * y) * w + x] - in.d[(size_t)(2 * y + 1) * w + x] - SqTendency(top, a, next));
      }
  }
}
// the default chain a decoder derives when the stream carries zero explicit steps (no meta channels here)
static std::vector<SqStep> DefaultSqueezeSteps(const std::vector<SChan>& ch) {
  std::vector<SqStep> p;
  const int nb = (int)ch.size();
  int w = ch[0].w, h = ch[0].h;
  if (nb > 2 && ch[1].w == w && ch[1].h == h) { p.push_back({true, false, 1, 2}); p.push_back({false, false, 1, 2}); }
  if (w <= h && h > 8) { p.push_back({false, true, 0, nb}); h = (h + 1) / 2; }
  while (w > 8 || h > 8) {
    if (w > 8) { p.push_back({true, true, 0, nb}); w = (w + 1) / 2; }
    if (h > 8) { p.push_back({false, true, 0, nb}); h = (h + 1) / 2; }
  }
  return p;
}
static void ApplySqueeze(std::vector<SChan>& ch, const std::vector<SqStep>& steps) {
  for (const SqStep& s : steps) {
    const int endc = s.begin_c + s.num_c - 1;
    if (endc >= (int)ch.size()) throw std::runtime_error("squeeze step out of range");
    const int offset = s.in_place ? endc + 1 : (int)ch.size();
    for (int c = s.begin_c; c <= endc; c++) {
      SChan avg, res;
      FwdSqueeze(ch[c], s.horizontal, avg, res);
      ch[c] = std::move(avg);
      ch.insert(ch.begin() + offset + (c - s.begin_c), std::move(res));
    }
  }
}

// squeeze: 0 = none, 1 = default chain (signalled with zero explicit steps), 2 = short explicit chain mixing in-place and appended residuals
static std::vector<uint8_t> EncodeModular(const int32_t* const* planes, int nchan, int w, int h, int bits, bool has_alpha, bool rct, int squeeze, const Params* fx = nullptr) {
  // channels: nchan colour (1 or 3) [+1 alpha].  Optional RCT type 6 (YCgCo) signalled as a global transform.
  const int group_shift = ModularGroupShift(), gd = 128 << group_shift, lfd = gd * 8;
  const int xg = (w + gd - 1) / gd, yg = (h + gd - 1) / gd, ngroups = xg * yg;
  const int xlg = (w + lfd - 1) / lfd, ylg = (h + lfd - 1) / lfd, nlf = xlg * ylg;
  const int ntot = nchan + (has_alpha ? 1 : 0);
  std::vector<SChan> ch(ntot);
  for (int c = 0; c < ntot; c++) { ch[c].d.assign(planes[c], planes[c] + (size_t)w * h); ch[c].w = w; ch[c].h = h; ch[c].hs = ch[c].vs = 0; }
  if (rct && nchan == 3) {
    for (size_t i = 0; i < (size_t)w * h; i++) {
      int32_t R = ch[0].d[i], G = ch[1].d[i], B = ch[2].d[i];
      int32_t co = R - B, tmp = B + (co >> 1), cg = G - tmp, y = tmp + (cg >> 1);
      ch[0].d[i] = y; ch[1].d[i] = co; ch[2].d[i] = cg;
    }
  }
  std::vector<SqStep> steps;
  if (squeeze == 1) steps = DefaultSqueezeSteps(ch);
  else if (squeeze == 2) {
    steps.push_back({true, true, 0, ntot});
    steps.push_back({false, true, 0, ntot});
    steps.push_back({true, false, 0, 1});
    if (ntot > 1) steps.push_back({false, false, ntot - 1, 1});
  }
  if (squeeze) ApplySqueeze(ch, steps);
  const int nfinal = (int)ch.size();
  // tree: channel split (unsqueezed only: squeezed sub-streams have varying channel lists), then prop 9 cutoffs, gradient predictor
  GTree t;
  static const int cuts[] = {-1023, -255, -63, -15, -3, 0, 3, 15, 63, 255, 1023};
  std::vector<int> cut(cuts, cuts + 11);
  int root;
  if (squeeze) root = BuildCutoffTree(t, 9, cut, 0, (int)cut.size(), 5);
  else {
    std::vector<int> sub(ntot);
    for (int c = 0; c < ntot; c++) sub[c] = BuildCutoffTree(t, 9, cut, 0, (int)cut.size(), 5);
    root = sub[ntot - 1];
    for (int c = ntot - 2; c >= 0; c--) root = t.add_inner(0, c, root, sub[c]);  // channel > c ? (higher channels) : channel c
  }
  std::vector<int> bfs{root};
  for (size_t i = 0; i < bfs.size(); i++) { const TNode& n = t.nodes[bfs[i]]; if (n.prop >= 0) { bfs.push_back(n.l); bfs.push_back(n.r); } }
  int leaf = 0;
  for (int id : bfs) if (t.nodes[id].prop < 0) t.nodes[id].ctx = leaf++;
  t.num_leaves = leaf;
  std::vector<Token> tree_tokens;
  TreeTokens(t, bfs, tree_tokens);
  const int np = fx ? std::max(1, std::min(3, fx->mod_passes)) : 1;
  const bool with_ds = fx ? fx->mod_ds != 0 : true;
  bool single = ngroups == 1 && np == 1;
  // passes.h GetDownsamplingBracket for the pass layout WriteFrameHeader announces
  auto bracket = [&](int pass, int* mins, int* maxs) {
    int mn = 3, mx = 2;
    for (int i = 0;; i++) {
      if (with_ds && i + 1 < np) mn = np - 1 - i;      // downsample 2^(np - 1 - i) completes at pass i
      if (i == np - 1) mn = 0;
      if (i == pass) break;
      mx = mn - 1;
    }
    *mins = mn; *maxs = mx;
  };
  // GlobalModular takes the leading channels that fit a group; the others go to LfGroups (shift >= 3) or PassGroups
  int nglobal = 0;
  while (nglobal < nfinal && ch[nglobal].w <= gd && ch[nglobal].h <= gd) nglobal++;
  std::vector<Token> global_tok;
  std::vector<std::vector<Token>> lftok(nlf), gtok((size_t)ngroups * np);
  std::vector<char> lf_has(nlf, 0), g_has((size_t)ngroups * np, 0);
  {
    std::vector<ChanRef> cr;
    for (int c = 0; c < nglobal; c++) if (ch[c].w && ch[c].h) cr.push_back({ch[c].d.data(), ch[c].w, ch[c].h});
    ModularTokens(t, root, cr, 0, global_tok);
  }
  auto group_stream = [&](int x0, int y0, int dim, int min_shift, int max_shift, int stream_id, std::vector<Token>& out) -> bool {
    std::vector<std::vector<int32_t>> store;
    std::vector<ChanRef> cr;
    for (int c = nglobal; c < nfinal; c++) {
      const SChan& sc = ch[c];
      if (!sc.w || !sc.h) continue;
      const int shift = std::min(sc.hs, sc.vs);
      if (shift < min_shift || shift > max_shift) continue;
      const int rx = x0 >> sc.hs, ry = y0 >> sc.vs;
      if (rx >= sc.w || ry >= sc.h) continue;
      const int rw = std::min(dim >> sc.hs, sc.w - rx), rh = std::min(dim >> sc.vs, sc.h - ry);
      if (rw <= 0 || rh <= 0) continue;
      store.emplace_back((size_t)rw * rh);
      for (int y = 0; y < rh; y++) memcpy(&store.back()[(size_t)y * rw], &sc.d[(size_t)(ry + y) * sc.w + rx], sizeof(int32_t) * rw);
      cr.push_back({nullptr, rw, rh});
    }
    for (size_t i = 0; i < cr.size(); i++) cr[i].d = store[i].data();
    if (cr.empty()) return false;
    ModularTokens(t, root, cr, stream_id, out);
    return true;
  };
  if (nglobal < nfinal) {
    for (int g = 0; g < nlf; g++) lf_has[g] = group_stream((g % xlg) * lfd, (g / xlg) * lfd, lfd, 3, 1000, 1 + nlf + g, lftok[g]);
    for (int ps = 0; ps < np; ps++) {
      int mins, maxs;
      bracket(ps, &mins, &maxs);
      for (int g = 0; g < ngroups; g++) g_has[(size_t)ps * ngroups + g] = group_stream((g % xg) * gd, (g / xg) * gd, gd, mins, maxs, 1 + 3 * nlf + 17 + ps * ngroups + g, gtok[(size_t)ps * ngroups + g]);
    }
  }
  EntropyCoder tree_code, code;
  { std::vector<const std::vector<Token>*> s{&tree_tokens}; BuildEntropyCoder(s, 6, UintConfig{4, 2, 0}, 6, tree_code); }
  { std::vector<const std::vector<Token>*> s{&global_tok}; for (auto& g : lftok) s.push_back(&g); for (auto& g : gtok) s.push_back(&g); BuildEntropyCoder(s, t.num_leaves, UintConfig{4, 2, 0}, 64, code); }
  std::vector<BitWriter> sections;
  {
    BitWriter s;
    WriteFeatures(s);
    s.put(1, 1);  // LfChannelDequantization default
    s.put(1, 1);  // has_tree
    WriteEntropyCode(s, tree_code);
    EncodeTokens(s, tree_code, tree_tokens);
    WriteEntropyCode(s, code);
    // global GroupHeader
    s.put(1, 1); s.put(1, 1);
    const bool has_rct = rct && nchan == 3;
    WriteU32(s, (has_rct ? 1 : 0) + (squeeze ? 1 : 0), {0, 0}, {0, 1}, {4, 2}, {8, 18});
    if (has_rct) { s.put(0, 2); WriteU32(s, 0, {3, 0}, {6, 8}, {10, 72}, {13, 1096}); WriteU32(s, 6, {0, 6}, {2, 0}, {4, 2}, {6, 10}); }
    if (squeeze) {
      s.put(2, 2);
      const std::vector<SqStep> none;
      const std::vector<SqStep>& sig = squeeze == 1 ? none : steps;
      WriteU32(s, (uint32_t)sig.size(), {0, 0}, {4, 1}, {6, 9}, {8, 41});
      for (const SqStep& q : sig) {
        s.put(q.horizontal, 1); s.put(q.in_place, 1);
        WriteU32(s, (uint32_t)q.begin_c, {3, 0}, {6, 8}, {10, 72}, {13, 1096});
        WriteU32(s, (uint32_t)q.num_c, {0, 1}, {0, 2}, {0, 3}, {4, 4});
      }
    }
    EncodeTokens(s, code, global_tok);  // (ANS state is written even when no channel is decodable globally)
    sections.push_back(s);
  }
  for (int g = 0; g < nlf; g++) {  // LfGroups: the squeezed channels with shift >= 3
    BitWriter s;
    if (lf_has[g]) { s.put(1, 1); s.put(1, 1); s.put(0, 2); EncodeTokens(s, code, lftok[g]); }
    sections.push_back(s);
  }
  // HfGlobal is absent for modular frames but still has a TOC slot
  sections.push_back(BitWriter());
  for (int k = 0; k < ngroups * np; k++) {      // PassGroups, pass-major
    BitWriter s;
    if (g_has[k]) { s.put(1, 1); s.put(1, 1); s.put(0, 2); EncodeTokens(s, code, gtok[k]); }
    sections.push_back(s);
  }
  BitWriter out;
  Params p;
  if (fx) p = *fx;
  p.mod_passes = np;
  p.out_bits = bits; p.gab = 0; p.epf_iters = 0; p.noise = 0; p.upsampling = 1; p.num_passes = 1; p.skip_lf_smoothing = 0;  // lossless: no restoration filters
  if (p.emit != 1) WriteImageHeader(out, p.canvas_w ? p.canvas_w : w, p.canvas_h ? p.canvas_h : h, p, p.xyb_image != 0, bits, has_alpha, nchan == 1);
  if (p.emit == 2) return out.bytes;
  WriteFrameHeader(out, p, true, p.xyb_image != 0, has_alpha ? 1 : 0, group_shift, false, w, h);
  WriteTOCAndSections(out, sections, single);
  out.align();
  return out.bytes;
}

}  // namespace synth

#include "synth_free.h"
#include "synth_ycbcr.h"

// ---- C API ---------------------------------------------------------------------------------------------------------
extern "C" {
struct jxlsynth_params {
  uint32_t seed; float distance; int32_t epf_iters, gab, strategy_mix, out_bits, hdr, skip_lf_smoothing, custom_orders, orientation, upsampling, custom_up_weights, num_passes, permute_toc, pass_ds; int32_t reserved[2];
};
static thread_local std::string g_err;
const char* jxlsynth_last_error() { return g_err.c_str(); }
void jxlsynth_free(uint8_t* p) { free(p); }
void jxlsynth_image(uint32_t seed, int w, int h, uint8_t* rgb) { synth::SyntheticImage(seed, w, h, rgb); }
// ICC profile embedded by the image headers written from now on in this thread (size 0: none, enumerated colour encoding)
void jxlsynth_set_icc(const uint8_t* icc, size_t size) { synth::g_icc.assign(icc, icc + size); }
void jxlsynth_set_float(int exp_bits) { synth::g_float_exp_bits = exp_bits; }
// entropy-coded streams written from now on in this thread use prefix (Huffman) codes instead of ANS — what cjxl's fast efforts emit
void jxlsynth_set_animation(int tps_num, int tps_den, int loops) { synth::g_anim_num = tps_num; synth::g_anim_den = tps_den > 0 ? tps_den : 1; synth::g_anim_loops = loops; }
void jxlsynth_set_preview(int w, int h) { synth::g_preview_w = w; synth::g_preview_h = h; }
void jxlsynth_set_prefix(int on) { synth::UsePrefixCodes() = on != 0; }
void jxlsynth_set_lz77_lf(int on) { synth::UseLz77Lf() = on != 0; }
void jxlsynth_set_lz77_ac(int on) { synth::UseLz77Ac() = on != 0; }
void jxlsynth_set_alpha_squeeze(int on) { synth::AlphaSqueeze() = on != 0; }
void jxlsynth_set_hf_presets(int n) { synth::HfPresets() = n < 1 ? 1 : n; }
void jxlsynth_set_lf_extra_precision(int e) { synth::LfExtraPrecision() = e < 0 || e > 3 ? 0 : e; }
void jxlsynth_set_quant_lf(int q) { synth::QuantLf() = q < 1 || q > 65536 ? 16 : q; }
void jxlsynth_set_qm_scales(int x, int b) { synth::QmScales()[0] = x < 0 || x > 7 ? 3 : x; synth::QmScales()[1] = b < 0 || b > 7 ? 2 : b; }
void jxlsynth_set_custom_opsin(int on) { synth::CustomOpsin() = on != 0; }
void jxlsynth_set_custom_lf_global(int on) { synth::CustomLfGlobal() = on != 0; }
void jxlsynth_set_custom_block_ctx(int on) { synth::CustomBlockCtx() = on != 0; }
void jxlsynth_set_custom_filters(int on) { synth::CustomFilters() = on != 0; }
void jxlsynth_set_modular_group_shift(int shift) { synth::ModularGroupShift() = shift < 0 || shift > 3 ? 1 : shift; }
void jxlsynth_set_prev_channel_props(int on) { synth::UsePrevChannelProps() = on != 0; }
void jxlsynth_set_lf_tree_shape(int shape) { synth::LfTreeShape() = shape; }
// rgba == NULL: the extra channel is alpha again
void jxlsynth_set_spot(const float* rgba) { synth::g_spot_set = rgba != nullptr; if (rgba) for (int i = 0; i < 4; i++) synth::g_spot[i] = rgba[i]; }
// white_point < 0 clears the override
void jxlsynth_set_color(int white_point, int primaries, int tf, uint32_t gamma_1e7, float intensity_target) {
  synth::g_color = synth::ColorOverride();
  if (white_point < 0) return;
  synth::g_color.set = true; synth::g_color.white_point = white_point; synth::g_color.primaries = primaries; synth::g_color.tf = tf;
  synth::g_color.gamma = gamma_1e7; synth::g_color.intensity_target = intensity_target;
}
// patch dictionary / splines of the frames written from now on in this thread (see WriteFeatures; n = 0 clears)
void jxlsynth_set_features(const int32_t* patches, size_t npatch, const int32_t* splines, size_t nspline, int num_extra) {
  synth::g_patches.assign(patches, patches + npatch); synth::g_splines.assign(splines, splines + nspline); synth::g_feature_extra = num_extra;
}

static int finish(const std::vector<uint8_t>& v, uint8_t** out, size_t* n) {
  *out = (uint8_t*)malloc(v.size());
  memcpy(*out, v.data(), v.size());
  *n = v.size();
  return 0;
}
// rgb8: interleaved sRGB u8 (w*h*3).  For hdr: rgb_lin (float, linear, w*h*3) is used instead.
int jxlsynth_vardct2(const uint8_t* rgb8, const float* rgb_lin, const uint8_t* alpha8, int w, int h, const jxlsynth_params* pp, uint8_t** out, size_t* n);
int jxlsynth_vardct(const uint8_t* rgb8, const float* rgb_lin, int w, int h, const jxlsynth_params* pp, uint8_t** out, size_t* n) {
  return jxlsynth_vardct2(rgb8, rgb_lin, nullptr, w, h, pp, out, n);
}
// alpha8: optional w*h 8-bit alpha plane carried as an extra channel of the VarDCT frame
int jxlsynth_vardct2(const uint8_t* rgb8, const float* rgb_lin, const uint8_t* alpha8, int w, int h, const jxlsynth_params* pp, uint8_t** out, size_t* n) {
  try {
    synth::Params p;
    p.seed = pp->seed; p.distance = pp->distance; p.epf_iters = pp->epf_iters; p.gab = pp->gab; p.strategy_mix = pp->strategy_mix;
    p.out_bits = pp->out_bits; p.hdr = pp->hdr; p.skip_lf_smoothing = pp->skip_lf_smoothing;
    p.orientation = pp->orientation >= 1 && pp->orientation <= 8 ? pp->orientation : 1;
    p.upsampling = (pp->upsampling == 2 || pp->upsampling == 4 || pp->upsampling == 8) ? pp->upsampling : 1;
    p.custom_up_weights = pp->custom_up_weights;
    p.num_passes = pp->num_passes >= 1 && pp->num_passes <= 3 ? pp->num_passes : 1;
    p.permute_toc = pp->permute_toc; p.pass_ds = pp->pass_ds;
    std::vector<float> pl[3];
    for (auto& v : pl) v.resize((size_t)w * h);
    const float scale = p.hdr ? 255.0f / 1000.0f : 1.0f;  // intensity_target 1000: linear 1.0 == 1000 nits
    (void)scale;
    for (size_t i = 0; i < (size_t)w * h; i++) {
      float r, g, b;
      if (rgb_lin) { r = rgb_lin[3 * i]; g = rgb_lin[3 * i + 1]; b = rgb_lin[3 * i + 2]; if (p.hdr) { r *= 1000.f / 255.f; g *= 1000.f / 255.f; b *= 1000.f / 255.f; } }
      else { r = synth::SrgbToLinear(rgb8[3 * i] / 255.0f); g = synth::SrgbToLinear(rgb8[3 * i + 1] / 255.0f); b = synth::SrgbToLinear(rgb8[3 * i + 2] / 255.0f); }
      synth::LinearToXYB(r, g, b, &pl[0][i], &pl[1][i], &pl[2][i]);
    }
    if (p.upsampling > 1) {
      // code the frame at 1/upsampling of the image size: box-filtered XYB, alpha point-sampled
      const int up = p.upsampling, cw = (w + up - 1) / up, ch = (h + up - 1) / up;
      std::vector<float> small[3];
      for (int c = 0; c < 3; c++) {
        small[c].resize((size_t)cw * ch);
        for (int y = 0; y < ch; y++) for (int x = 0; x < cw; x++) {
          double acc = 0; int cnt = 0;
          for (int yy = y * up; yy < std::min(h, (y + 1) * up); yy++) for (int xx = x * up; xx < std::min(w, (x + 1) * up); xx++) { acc += pl[c][(size_t)yy * w + xx]; cnt++; }
          small[c][(size_t)y * cw + x] = (float)(acc / cnt);
        }
      }
      std::vector<uint8_t> small_a;
      if (alpha8) { small_a.resize((size_t)cw * ch); for (int y = 0; y < ch; y++) for (int x = 0; x < cw; x++) small_a[(size_t)y * cw + x] = alpha8[(size_t)std::min(h - 1, y * up + up / 2) * w + std::min(w - 1, x * up + up / 2)]; }
      const float* planes[3] = {small[0].data(), small[1].data(), small[2].data()};
      return finish(synth::EncodeVarDCT(planes, cw, ch, p, alpha8 ? small_a.data() : nullptr, w, h), out, n);
    }
    const float* planes[3] = {pl[0].data(), pl[1].data(), pl[2].data()};
    return finish(synth::EncodeVarDCT(planes, w, h, p, alpha8), out, n);
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
// Frame control for multi-frame / feature streams: a stream is the concatenation of one image header (emit = 2, or the first
// frame emitted with emit = 0) and any number of frames (emit = 1), the last one with is_last = 1.
struct jxlsynth_frame {
  int32_t noise; uint32_t noise_lut[8];
  int32_t frame_type, have_crop, crop_x0, crop_y0, canvas_w, canvas_h, blend_mode, blend_source, blend_clamp, is_last, save_as_reference, save_before_ct, emit, num_extra_hdr, xyb_image, alpha_premultiplied, use_lf_frame, lf_level, mod_passes, mod_ds, duration;
};
static void ApplyFrame(synth::Params& p, const jxlsynth_frame* fx) {
  if (!fx) return;
  p.noise = fx->noise; for (int i = 0; i < 8; i++) p.noise_lut[i] = fx->noise_lut[i];
  p.frame_type = fx->frame_type; p.have_crop = fx->have_crop; p.crop_x0 = fx->crop_x0; p.crop_y0 = fx->crop_y0; p.canvas_w = fx->canvas_w; p.canvas_h = fx->canvas_h;
  p.blend_mode = fx->blend_mode; p.blend_source = fx->blend_source; p.blend_clamp = fx->blend_clamp; p.is_last = fx->is_last; p.save_as_reference = fx->save_as_reference;
  p.save_before_ct = fx->save_before_ct; p.emit = fx->emit; p.num_extra_hdr = fx->num_extra_hdr; p.xyb_image = fx->xyb_image; p.alpha_premultiplied = fx->alpha_premultiplied;
  p.use_lf_frame = fx->use_lf_frame; p.lf_level = fx->lf_level;
  p.mod_passes = fx->mod_passes > 0 ? fx->mod_passes : 1; p.mod_ds = fx->mod_ds; p.duration = fx->duration;
}
int jxlsynth_vardct3(const uint8_t* rgb8, const uint8_t* alpha8, int w, int h, const jxlsynth_params* pp, const jxlsynth_frame* fx, uint8_t** out, size_t* n) {
  try {
    synth::Params p;
    p.seed = pp->seed; p.distance = pp->distance; p.epf_iters = pp->epf_iters; p.gab = pp->gab; p.strategy_mix = pp->strategy_mix;
    p.out_bits = pp->out_bits; p.skip_lf_smoothing = pp->skip_lf_smoothing;
    ApplyFrame(p, fx);
    std::vector<float> pl[3];
    for (auto& v : pl) v.resize((size_t)w * h);
    for (size_t i = 0; i < (size_t)w * h; i++)
      synth::LinearToXYB(synth::SrgbToLinear(rgb8[3 * i] / 255.0f), synth::SrgbToLinear(rgb8[3 * i + 1] / 255.0f), synth::SrgbToLinear(rgb8[3 * i + 2] / 255.0f), &pl[0][i], &pl[1][i], &pl[2][i]);
    const float* planes[3] = {pl[0].data(), pl[1].data(), pl[2].data()};
    return finish(synth::EncodeVarDCT(planes, w, h, p, alpha8), out, n);
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
int jxlsynth_modular3(const int32_t* const* planes, int nchan, int has_alpha, int w, int h, int bits, int rct, int squeeze, const jxlsynth_frame* fx, uint8_t** out, size_t* n) {
  try {
    synth::Params p;
    ApplyFrame(p, fx);
    return finish(synth::EncodeModular(planes, nchan, w, h, bits, has_alpha != 0, rct != 0, squeeze, &p), out, n);
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
// planes: nchan (+alpha) pointers to w*h int32 samples
int jxlsynth_modular2(const int32_t* const* planes, int nchan, int has_alpha, int w, int h, int bits, int rct, int squeeze, uint8_t** out, size_t* n) {
  try { return finish(synth::EncodeModular(planes, nchan, w, h, bits, has_alpha != 0, rct != 0, squeeze), out, n); }
  catch (const std::exception& e) { g_err = e.what(); return 1; }
}
int jxlsynth_modular(const int32_t* const* planes, int nchan, int has_alpha, int w, int h, int bits, int rct, uint8_t** out, size_t* n) {
  try { return finish(synth::EncodeModular(planes, nchan, w, h, bits, has_alpha != 0, rct != 0, 0), out, n); }
  catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// YCbCr VarDCT frame with chroma subsampling (tools/synth_ycbcr.h); modes = sampling-factor modes of Cb, Y, Cr (0 1x1, 1 2x2, 2 2x1, 3 1x2)
int jxlsynth_ycbcr(const uint8_t* rgb8, int w, int h, const int32_t* modes, uint32_t seed, float distance, uint8_t** out, size_t* n) {
  try {
    synth::Params p; p.seed = seed; p.distance = distance;
    const int m[3] = {modes[0], modes[1], modes[2]};
    return finish(synth::EncodeYCbCr(rgb8, w, h, m, p), out, n);
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
// Codestream of a JPEG transcode (tools/synth_ycbcr.h EncodeJpegTranscode): planes / qt per jxl channel (Cb, Y, Cr), JPEG natural order
int jxlsynth_jpeg_transcode(int w, int h, const int32_t* modes, const int16_t* cb, const int16_t* y, const int16_t* cr, const int32_t* qt, uint8_t** out, size_t* n) {
  try {
    synth::Params p;
    const int m[3] = {modes[0], modes[1], modes[2]};
    const int16_t* planes[3] = {cb, y, cr};
    std::vector<int16_t> zero;
    const bool gray = !cb && !cr;                       // one-component JPEG: grey image header, empty chroma channels
    if (gray) { zero.assign((size_t)((w + 7) / 8) * ((h + 7) / 8) * 64, 0); planes[0] = planes[2] = zero.data(); }
    return finish(synth::EncodeJpegTranscode(w, h, m, planes, qt, p, gray), out, n);
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
// Free-running Modular stream (tools/synth_free.h): feature coverage without an encoder-side simulation of the decoder.
struct jxlsynth_free_params { uint32_t seed; int w, h, nchan, has_alpha, bits, tree_flags, tree_depth, local_trees, lz77, palette, nb_colors, nb_deltas, pal_pred; };
int jxlsynth_modular_free(const jxlsynth_free_params* pp, uint8_t** out, size_t* n) {
  try {
    synth::FreeParams p;
    p.seed = pp->seed; p.w = pp->w; p.h = pp->h; p.nchan = pp->nchan; p.has_alpha = pp->has_alpha; p.bits = pp->bits;
    p.tree_flags = pp->tree_flags; p.tree_depth = pp->tree_depth; p.local_trees = pp->local_trees; p.lz77 = pp->lz77;
    p.palette = pp->palette; p.nb_colors = pp->nb_colors; p.nb_deltas = pp->nb_deltas; p.pal_pred = pp->pal_pred;
    return finish(synth::EncodeModularFree(p), out, n);
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
}
