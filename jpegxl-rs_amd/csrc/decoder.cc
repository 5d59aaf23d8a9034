// jxl-hip: batch decoder (see decoder.h).
#include "decoder.h"
#include <atomic>
#include <exception>
#include <mutex>
#include <thread>
#include <chrono>
#include <map>
#include <queue>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstring>

namespace jxlhip {

#define HIP_CHECK(expr)                                                                                    \
  do {                                                                                                     \
    hipError_t e_ = (expr);                                                                                \
    if (e_ != hipSuccess) throw ParseError(std::string("HIP error: ") + hipGetErrorString(e_) + " in " #expr, false); \
  } while (0)

// Growable byte buffer in pinned host memory — what Prepare assembles the constant arena in and uploads from.  Pinned: the upload is a
// DMA at link speed (57 GB/s) instead of a staged copy, and can overlap the GPU's work.  The capacity survives clear(): a batch object that is
// refilled step after step (JxlHipBatchReset) allocates once.  Falls back to pageable memory when pinning fails (no GPU: host-only describe).
HostStage::~HostStage() {
  Release();
  for (auto& r : retired_) { if (r.second) (void)hipHostFree(r.first); else std::free(r.first); }
}
void HostStage::Release() {
  if (!p_) return;
  if (pinned_) (void)hipHostFree(p_); else std::free(p_);
  p_ = nullptr; n_ = cap_ = 0;
}
void HostStage::Reserve(size_t n) {
  if (n <= cap_) return;
  const size_t keep = n_;
  Resize(n);
  n_ = keep;
}
void HostStage::Resize(size_t n) {
  if (n > cap_) {
    // (pinned allocations cost milliseconds each and serialise with the rest of the runtime: capacities double, so that an arena assembled piece by piece
    // — or a batch object refilled with jobs of changing size — settles after a few)
    size_t cap = std::max<size_t>(cap_ * 2, 1 << 20);
    while (cap < n) cap *= 2;
    uint8_t* q = nullptr;
    bool pinned = hipHostMalloc((void**)&q, cap, hipHostMallocDefault) == hipSuccess && q;
    if (!pinned) { (void)hipGetLastError(); q = (uint8_t*)std::malloc(cap); if (!q) throw std::bad_alloc(); }
    const size_t keep = n_;
    if (keep) memcpy(q, p_, keep);
    // (hipHostFree waits for the whole device — hundreds of milliseconds under a busy pipeline: the outgrown block is kept until the object dies; capacities
    // double, so all of them together are smaller than the live one)
    if (p_) retired_.push_back({p_, pinned_});
    p_ = q; cap_ = cap; pinned_ = pinned; n_ = keep;
  }
  if (n > n_) {   // zero what Put does not overwrite: alignment gaps (< 256 B) — the payload is copied over right after
    const size_t gap_end = std::min(n, n_ + 512);
    memset(p_ + n_, 0, gap_end - n_);
  }
  n_ = n;
}

namespace {
size_t Align(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// library-default kernels (image_metadata.cc kWeights2 / 4 / 8): each sub-pixel kernel sums to 1
static const float kUp2[15] = {-0.01716200f, -0.03452303f, -0.04022174f, -0.02921014f, -0.00624645f, 0.14111091f, 0.28896755f, 0.00278718f,
                               -0.01610267f, 0.56661550f,  0.03777607f,  -0.01986694f, -0.03144731f, -0.01185068f, -0.00213539f};
static const float kUp4[55] = {
    -0.02419067f, -0.03491987f, -0.03693351f, -0.03094285f, -0.00529785f, -0.01663432f, -0.03556863f, -0.03888905f, -0.03516850f, -0.00989469f, 0.23651958f,
    0.33392945f,  -0.01073543f, -0.01313181f, -0.03556694f, 0.13048175f,  0.40103025f,  0.03951150f,  -0.02077584f, 0.46914198f,  -0.00209270f, -0.01484589f,
    -0.04064806f, 0.18942530f,  0.56279892f,  0.06674400f,  -0.02335494f, -0.03551682f, -0.00754830f, -0.02267919f, -0.02363578f, 0.00315804f,  -0.03399098f,
    -0.01359519f, -0.00091653f, -0.00335467f, -0.01163294f, -0.01610294f, -0.00974088f, -0.00191622f, -0.01095446f, -0.03198464f, -0.04455121f, -0.02799790f,
    -0.00645912f, 0.06390599f,  0.22963888f,  0.00630981f,  -0.01897349f, 0.67537268f,  0.08483369f,  -0.02534994f, -0.02205197f, -0.01667999f, -0.00384443f};
static const float kUp8[210] = {
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


struct ConstOffsets {
  size_t cs = 0, sec_off = 0, sec_size = 0, tree = 0, bcm = 0;
  size_t mod_ctx = 0, mod_cfg = 0, mod_alias = 0, mod_pc = 0, mod_po = 0, mod_ps = 0, mod_chan = 0, up_weights = 0;
  struct Pass { size_t ac_ctx = 0, ac_cfg = 0, ac_alias = 0, ac_pc = 0, ac_po = 0, ac_ps = 0; size_t orders[39] = {0}; };
  vec<Pass> pass;
  struct Local { uint32_t unit = 0; size_t tree = 0, ctx = 0, cfg = 0, alias = 0, pc = 0, po = 0, ps = 0; };   // sub-streams with their own tree / code
  vec<Local> local;
  size_t qtable[17 * 3] = {0};
  bool has_qtable[17] = {false};
};

struct Arena {
  HostStage& buf;
  explicit Arena(HostStage& b) : buf(b) {}
  size_t Put(const void* src, size_t n, size_t align = 256) {
    const size_t off = Align(buf.size(), align), end = off + std::max<size_t>(n, 4);
    buf.Resize(end);                                   // (the gap before `off` and a short tail are zeroed)
    if (n) memcpy(buf.data() + off, src, n);
    return off;
  }
};

void PutCode(Arena& a, const HostCode& c, size_t* ctx, size_t* cfg, size_t* alias, size_t* pc, size_t* po, size_t* ps) {
  *ctx = a.Put(c.ctx_map.data(), c.ctx_map.size());
  *cfg = a.Put(c.cfg.data(), c.cfg.size() * 4);
  *alias = a.Put(c.alias.data(), c.alias.size() * 8);
  *pc = a.Put(c.pfx_count.data(), c.pfx_count.size() * 2);
  *po = a.Put(c.pfx_sym_off.data(), c.pfx_sym_off.size() * 4);
  *ps = a.Put(c.pfx_syms.data(), c.pfx_syms.size() * 2);
}
// ---- SIMT LF decode: channel classes (kernels.h LfSimtPlan) -----------------------------------------------------------------------------
// Classifies the part of the MA tree channel `chan` of sub-stream `stream_id` can reach (static splits on the channel index / stream id are
// followed wherever they occur; splits on the row — property 2 — partition the channel's rows into classes).  Within a row class the
// subtree may test at most two of {W + N - NW (9), W (7), N (6), the weighted predictor's largest error (15)}: one property through a
// 1024-entry value -> cluster table (splits inside [-512, 510]), two through a 32 x 32 table (splits inside [-16, 14]); its leaves share one
// predictor out of zero / W / N / clamped gradient / weighted / (W + N) / 2 / select / NE, with offset 0 and multiplier 1.  That covers the
// fixed trees libjxl's encoder writes for LF-group streams at its default and faster efforts (enc_modular.cc: "WP fixed DC", "gradient
// fixed DC", "AC meta" and its variants); learned trees (slower efforts) usually test more properties and keep LfDecodeKernel.
struct LfRowClass { uint32_t kind = 0, pred = 0, sel_a = 0, sel_b = 0, cluster = 0; vec<uint8_t> lut; };
struct LfChanClass { bool uses_wp = false; vec<LfRowClass> rows; uint8_t row_to_class[512]; };
namespace {
struct LfClassifier {
  const vec<TreeNode>& nd;
  const HostCode& code;
  int chan; int32_t stream_id;
  LfClassifier(const HostTree& t, const HostCode& c, int ch, uint32_t sid) : nd(t.nodes), code(c), chan(ch), stream_id((int32_t)sid) {}
  static int SelOf(int prop) { return prop == 9 ? 0 : prop == 7 ? 1 : prop == 6 ? 2 : prop == 15 ? 3 : -1; }
  static int PredCode(int p) { static const int k[8] = {0, 1, 2, 5, 6, 3, 4, 7}; for (int i = 0; i < 8; i++) if (k[i] == p) return i; return -1; }
  // child taken at a static node, or -1 for a dynamic one; y < 0: the row is not fixed (its splits count as dynamic)
  int Static(const TreeNode& n, int y) const {
    if (n.prop == 0) return chan > n.val ? (int)n.a : (int)n.b;
    if (n.prop == 1) return stream_id > n.val ? (int)n.a : (int)n.b;
    if (n.prop == 2 && y >= 0) return y > n.val ? (int)n.a : (int)n.b;
    return -1;
  }
  // row thresholds and weighted-predictor use of the whole reachable subtree
  bool Survey(vec<int32_t>* ysplits, bool* uses_wp) const {
    vec<uint32_t> stack{0};
    size_t visited = 0;
    while (!stack.empty()) {
      const uint32_t pos = stack.back(); stack.pop_back();
      if (pos >= nd.size() || ++visited > 8192) return false;
      const TreeNode& n = nd[pos];
      if (n.prop < 0) { if ((n.a & 0xFF) == 6) *uses_wp = true; continue; }
      const int st = Static(n, -1);
      if (st >= 0) { stack.push_back((uint32_t)st); continue; }
      if (n.prop == 2) ysplits->push_back(n.val);
      if (n.prop == 15) *uses_wp = true;
      stack.push_back(n.a); stack.push_back(n.b);
    }
    return true;
  }
  bool ClassifyRow(int y, LfRowClass* out) const {
    int props[2] = {-1, -1}, np = 0, pred = -1;
    int32_t lo_split = 0, hi_split = 0;
    {
      vec<uint32_t> stack{0};
      size_t visited = 0;
      while (!stack.empty()) {
        const uint32_t pos = stack.back(); stack.pop_back();
        if (pos >= nd.size() || ++visited > 8192) return false;
        const TreeNode& n = nd[pos];
        if (n.prop < 0) {
          const int pc = PredCode((int)(n.a & 0xFF));
          if (pc < 0 || n.val != 0 || n.b != 1) return false;
          if (pred < 0) pred = pc; else if (pred != pc) return false;
          if ((n.a >> 8) >= code.ctx_map.size()) return false;
          continue;
        }
        const int st = Static(n, y);
        if (st >= 0) { stack.push_back((uint32_t)st); continue; }
        if (SelOf(n.prop) < 0) return false;
        int k = 0;
        while (k < np && props[k] != n.prop) k++;
        if (k == np) { if (np == 2) return false; props[np++] = n.prop; }
        lo_split = std::min(lo_split, n.val); hi_split = std::max(hi_split, n.val);
        stack.push_back(n.a); stack.push_back(n.b);
      }
    }
    if (pred < 0) return false;
    const int32_t lo = np == 2 ? -16 : -512, hi = np == 2 ? 15 : 511;
    if (np && (lo_split < lo || hi_split > hi - 1)) return false;
    out->kind = (uint32_t)np; out->pred = (uint32_t)pred;
    out->sel_a = np > 0 ? (uint32_t)SelOf(props[0]) : 0; out->sel_b = np > 1 ? (uint32_t)SelOf(props[1]) : 0;
    const int32_t span = hi - lo + 1;
    out->lut.assign(np == 0 ? 1 : np == 1 ? (size_t)span : (size_t)span * span, 0);
    // the table, by pushing the value box down the tree (a split at `val` sends (val, hi] to child a, [lo, val] to child b): every
    // reachable node once instead of a walk from the root per value — 7 tables per LF group of every frame add up in a streaming loop
    struct Box { uint32_t pos; int32_t lo[2], hi[2]; };
    vec<Box> stack{Box{0, {lo, lo}, {hi, hi}}};
    while (!stack.empty()) {
      const Box r = stack.back(); stack.pop_back();
      const TreeNode& n = nd[r.pos];
      if (n.prop < 0) {
        const uint8_t cl = code.ctx_map[n.a >> 8];
        if (np == 0) out->lut[0] = cl;
        else if (np == 1) memset(out->lut.data() + (r.lo[0] - lo), cl, (size_t)(r.hi[0] - r.lo[0] + 1));
        else for (int32_t a = r.lo[0]; a <= r.hi[0]; a++) memset(out->lut.data() + (size_t)(a - lo) * span + (r.lo[1] - lo), cl, (size_t)(r.hi[1] - r.lo[1] + 1));
        continue;
      }
      const int st = Static(n, y);
      if (st >= 0) { Box b = r; b.pos = (uint32_t)st; stack.push_back(b); continue; }
      const int k = n.prop == props[0] ? 0 : 1;
      if (r.hi[k] > n.val) { Box b = r; b.pos = n.a; b.lo[k] = std::max(r.lo[k], n.val + 1); stack.push_back(b); }
      if (r.lo[k] <= n.val) { Box b = r; b.pos = n.b; b.hi[k] = std::min(r.hi[k], n.val); stack.push_back(b); }
    }
    out->cluster = out->lut[0];
    return true;
  }
};
}  // namespace
// whether channel `chan` of sub-stream `stream_id` can reach a weighted-predictor leaf or a split on its error (a damaged tree counts as yes)
bool LfChannelUsesWp(const HostTree& tree, int chan, uint32_t stream_id) {
  static const HostCode no_code;
  vec<int32_t> ysplits;
  bool uses = false;
  return tree.nodes.empty() || !LfClassifier(tree, no_code, chan, stream_id).Survey(&ysplits, &uses) || uses;
}
bool ClassifyLfChannel(const HostTree& tree, const HostCode& code, int chan, uint32_t stream_id, LfChanClass* out) {
  if (tree.nodes.empty()) return false;
  LfClassifier cl(tree, code, chan, stream_id);
  vec<int32_t> ysplits;
  out->uses_wp = false;
  if (!cl.Survey(&ysplits, &out->uses_wp)) return false;
  // rows 0 ... 511 (the table's reach; LF-group channels have at most 256 rows): a class per interval between row thresholds
  vec<int32_t> starts{0};
  for (int32_t v : ysplits) if (v >= 0 && v < 511) starts.push_back(v + 1);
  std::sort(starts.begin(), starts.end());
  starts.erase(std::unique(starts.begin(), starts.end()), starts.end());
  if (starts.size() > 64) return false;
  out->rows.clear();
  for (size_t k = 0; k < starts.size(); k++) {
    LfRowClass rc;
    if (!cl.ClassifyRow(starts[k], &rc)) return false;
    size_t idx = 0;
    while (idx < out->rows.size() && !(out->rows[idx].kind == rc.kind && out->rows[idx].pred == rc.pred && out->rows[idx].sel_a == rc.sel_a && out->rows[idx].sel_b == rc.sel_b && out->rows[idx].lut == rc.lut)) idx++;
    if (idx == out->rows.size()) out->rows.push_back(std::move(rc));
    const int32_t end = k + 1 < starts.size() ? starts[k + 1] : 512;
    for (int32_t y = starts[k]; y < end; y++) out->row_to_class[y] = (uint8_t)idx;
  }
  return true;
}

DevCode ViewCode(const HostCode& c, const uint8_t* base, size_t ctx, size_t cfg, size_t alias, size_t pc, size_t po, size_t ps) {
  DevCode d;
  d.ctx_map = base + ctx; d.cfg = (const uint32_t*)(base + cfg); d.alias = (const uint64_t*)(base + alias);
  d.pfx_count = (const uint16_t*)(base + pc); d.pfx_sym_off = (const uint32_t*)(base + po); d.pfx_syms = (const uint16_t*)(base + ps);
  d.num_ctx = c.num_ctx; d.num_clusters = c.num_clusters; d.log_alpha = c.log_alpha; d.use_prefix = c.use_prefix;
  d.lz77 = c.lz77; d.lz_min_symbol = c.lz_min_symbol; d.lz_min_length = c.lz_min_length; d.lz_len_cfg = c.lz_len_cfg;
  return d;
}
}  // namespace

// Colour-transform parameters of an image (stage_xyb.cc OpsinParams, dec_xyb.cc OutputEncodingInfo::SetColorEncoding):
// inverse opsin matrix scaled to the intensity target — for grey-scale images its rows are replaced by their luminance-weighted
// sum (kSRGBLuminances; Mul3x3Matrix accumulates in double), so R = G = B — and the output transfer function.
// sRGB or linear output (colour modes 0 / 1): what the fused gaborish + EPF + output kernel computes; other transfer functions take the
// unfused filter kernels and OutputKernel
static bool SimpleTransfer(const ImageHeader& ih) { return ih.color_default || (!ih.have_gamma && (ih.tf == 13 || ih.tf == 8)); }

static void FillColor(const ImageHeader& ih, bool do_ycbcr, FrameDev& f) {
  float inv[9];
  for (int k = 0; k < 9; k++) inv[k] = ih.opsin_inv[k];
  // images whose header names other primaries / another white point than sRGB / D65: XYB decodes to linear sRGB, the matrix takes it on
  // to the image's own primaries (icc_profile.cc SrgbToOriginalPrimaries)
  float luminances[3] = {0.2126f, 0.7152f, 0.0722f};
  double to_original[9];
  if (ih.xyb_encoded && SrgbToOriginalPrimaries(ih, to_original, luminances)) {   // (not for other images: their samples are in that space already)
    float adapted[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
      double e = 0;
      for (int k = 0; k < 3; k++) e += to_original[i * 3 + k] * (double)inv[k * 3 + j];
      adapted[i * 3 + j] = (float)e;
    }
    for (int k = 0; k < 9; k++) inv[k] = adapted[k];
  }
  if (ih.color_space == 1) {
    const float lum[3] = {0.2126f, 0.7152f, 0.0722f};
    float folded[9];
    for (int x = 0; x < 3; x++) for (int y = 0; y < 3; y++) {
      double e = 0;
      for (int z = 0; z < 3; z++) e += lum[z] * inv[z * 3 + x];
      folded[y * 3 + x] = (float)e;
    }
    for (int k = 0; k < 9; k++) inv[k] = folded[k];
  }
  const float s = 255.0f / ih.intensity_target;
  for (int k = 0; k < 9; k++) f.opsin_inv[k] = inv[k] * s;
  for (int k = 0; k < 3; k++) { f.neg_bias[k] = ih.opsin_bias[k]; f.neg_bias_cbrt[k] = std::cbrt(ih.opsin_bias[k]); }
  f.inverse_gamma = 1.0f;
  if (ih.xyb_encoded) {
    f.color_mode = 0;
    if (!ih.color_default) {
      if (ih.have_gamma) { f.color_mode = 4; f.inverse_gamma = (float)ih.gamma * 1e-7f; }
      else if (ih.tf == 8) f.color_mode = 1;
      else if (ih.tf == 17) { f.color_mode = 4; f.inverse_gamma = 1.0f / 2.6f; }
      else if (ih.tf == 1) f.color_mode = 5;
      else if (ih.tf == 16) { f.color_mode = 6; f.hdr_par[0] = ih.intensity_target * (1.0f / 10000.0f); }   // TF_PQ(intensity_target)
      else if (ih.tf == 18) {
        // stage_from_linear.cc OpHlg: HlgOOTF::ToSceneLight(display_luminance = intensity target, luminances of the output primaries)
        f.color_mode = 7;
        const float gamma = (1 / 1.2f) * std::pow(1.111f, -std::log2(ih.intensity_target / 1000.f));
        f.hdr_par[0] = gamma - 1;
        f.hdr_par[1] = (f.hdr_par[0] < -0.01f || 0.01f < f.hdr_par[0]) ? 1.0f : 0.0f;
        for (int k = 0; k < 3; k++) f.hdr_par[2 + k] = luminances[k];
      }
    }
  } else f.color_mode = do_ycbcr ? 2 : 3;
}

// ---- device arena pool.  hipMalloc / hipFree of the multi-gigabyte arenas cost 0.5-3.5 s per batch object on the boxes measured (page tables of tens of GB set up and
// torn down), which is what a "fresh batch" paid each time (bench.py one_pass_128: 92 ms here, 3.6 s on the driver's box in round 3).  Arenas that a batch lets go of are kept
// in a process-wide free list (bounded: JXL_HIP_ARENA_POOL_MB, default 60 % of the device's memory; 0 turns the pool off) and handed to the next batch that asks for that much memory on that device.
namespace {
struct ArenaPool {
  struct Block { void* p; size_t cap; int dev; };
  std::mutex mu;
  std::vector<Block> blocks;
  size_t held = 0;
  static size_t Limit() {   // default: 60 % of the device's memory — what is handed back with hipFree is what the next hipMalloc (of any size) waits for: 2-3.5 s after a pipeline's 130 GB
    static const size_t v = [] {
      if (const char* e = getenv("JXL_HIP_ARENA_POOL_MB")) return (size_t)atoll(e) << 20;
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); total_b = (size_t)64 << 30; }
      return total_b / 10 * 6;
    }();
    return v;
  }
  static constexpr size_t kMinBytes = (size_t)8 << 20;   // small allocations are cheap: not pooled
  void* Take(size_t want, size_t* cap, int dev) {
    if (want < kMinBytes || Limit() == 0) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    int best = -1;
    for (size_t i = 0; i < blocks.size(); i++)
      if (blocks[i].dev == dev && blocks[i].cap >= want && blocks[i].cap <= 4 * want + ((size_t)64 << 20) && (best < 0 || blocks[i].cap < blocks[(size_t)best].cap)) best = (int)i;
    if (best < 0) return nullptr;
    void* p = blocks[(size_t)best].p; *cap = blocks[(size_t)best].cap; held -= *cap;
    blocks.erase(blocks.begin() + best);
    return p;
  }
  // `dev`: the device the block was allocated on (the owner's, not the calling thread's current one).  The device-wide wait — what hipFree does implicitly: nothing in
  // flight may still touch the block when somebody else gets it — happens outside the pool's lock.
  // idle: the caller knows that nothing on the device refers to the block any more (a batch object is only refilled after its last decode has left the GPU) — no
  // device-wide wait, which under a busy pipeline takes as long as everything in flight (seen: 400 ms stalls of a prepare thread)
  void Give(void* p, size_t cap, int dev, bool idle = false) {
    if (!p) return;
    if (dev >= 0 && cap >= kMinBytes && Limit() != 0) {
      bool room;
      { std::lock_guard<std::mutex> lock(mu); room = held + cap <= Limit() && blocks.size() < 96; }
      if (room && idle) {
        std::lock_guard<std::mutex> lock(mu);
        if (held + cap <= Limit() && blocks.size() < 96) { blocks.push_back(Block{p, cap, dev}); held += cap; return; }
      } else if (room) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = -1; }
        if (cur != dev) (void)hipSetDevice(dev);
        (void)hipDeviceSynchronize();
        if (cur >= 0 && cur != dev) (void)hipSetDevice(cur);
        std::lock_guard<std::mutex> lock(mu);
        if (held + cap <= Limit() && blocks.size() < 96) { blocks.push_back(Block{p, cap, dev}); held += cap; return; }
      }
    }
    (void)hipFree(p);
  }
  size_t Trim() {   // gives everything back to the runtime (an allocation failed: the pool may be what is in the way; or the caller asked: JxlHipArenaPoolTrim)
    std::vector<Block> mine;
    size_t bytes;
    { std::lock_guard<std::mutex> lock(mu); mine.swap(blocks); bytes = held; held = 0; }
    for (auto& b : mine) (void)hipFree(b.p);
    return bytes;
  }
  size_t Held() { std::lock_guard<std::mutex> lock(mu); return held; }
};
ArenaPool& Pool() { static ArenaPool* pool = new ArenaPool(); return *pool; }     // (never destroyed: the runtime may be gone by then)
}  // namespace
size_t DeviceArenaPoolTrim() { return Pool().Trim(); }
size_t DeviceArenaPoolHeld() { return Pool().Held(); }
void* DeviceArenaTake(size_t want, size_t* cap, int device) { return Pool().Take(want, cap, device); }
void DeviceArenaGive(void* p, size_t cap, int device, bool idle) { Pool().Give(p, cap, device, idle); }

Batch::Batch(int device) : device_(device) {
  if (device_ >= 0) HIP_CHECK(hipSetDevice(device_));      // (-1: host-side parsing only, JxlHipDebugDescribe)
}
Batch::~Batch() {
  // (the calling thread's current device is left as it was found)
  int cur = -1;
  if (device_ >= 0 && hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = -1; }
  if (device_ >= 0 && cur != device_) (void)hipSetDevice(device_);
  if (clear_stream_) { (void)hipStreamSynchronize((hipStream_t)clear_stream_); (void)hipStreamDestroy((hipStream_t)clear_stream_); }
  if (clear_event_) (void)hipEventDestroy((hipEvent_t)clear_event_);
  for (void* ev : mod_join_events_) (void)hipEventDestroy((hipEvent_t)ev);
  if (mod_fork_event_) (void)hipEventDestroy((hipEvent_t)mod_fork_event_);
  if (idct_event_) (void)hipEventDestroy((hipEvent_t)idct_event_);
  if (flags_event_) (void)hipEventDestroy((hipEvent_t)flags_event_);
  if (flags_pinned_) (void)hipHostFree(flags_pinned_);
  if (status_event_) (void)hipEventDestroy((hipEvent_t)status_event_);
  if (status_pinned_) (void)hipHostFree(status_pinned_);
  Pool().Give(dconst_, const_cap_, device_);
  Pool().Give(dwork_, work_cap_, device_);
  if (dcoef_ && !coef_owner_ && !coef_is_ext_) Pool().Give(dcoef_, coef_cap_, device_);
  if (dbig_ && !big_owner_ && !big_is_ext_) Pool().Give(dbig_, big_cap_, device_);
  if (big_owner_) big_owner_->big_sharers_--;
  if (dframes_) (void)hipFree(dframes_);
  if (dpasses_) (void)hipFree(dpasses_);
  if (dlocal_) (void)hipFree(dlocal_);
  if (device_ >= 0 && cur >= 0 && cur != device_) (void)hipSetDevice(cur);
}

// The coefficient and pixel planes are only touched by the "rest" half of a decode (HF decode ... write), which a caller
// that pipelines two batches runs strictly one after the other on one stream: the second batch may therefore use the
// first one's buffers.  Must be called before Prepare(); `owner` must already be prepared and stay alive.
void Batch::ShareBigArena(Batch* owner) {
  if (prepared_) throw ParseError("ShareBigArena after Prepare", false);
  if (big_owner_ == owner) return;
  if (big_owner_) { big_owner_->big_sharers_--; dbig_ = nullptr; big_cap_ = 0; }       // (the pointer aliased the old owner's planes: not ours to free)
  if (dbig_ && !big_is_ext_) Pool().Give(dbig_, big_cap_, device_);
  dbig_ = nullptr; big_cap_ = 0; big_is_ext_ = false;
  big_owner_ = owner;
  if (owner) owner->big_sharers_++;
}
// Likewise for the quantised-coefficient planes (written by the HF stage, consumed — and zeroed again — by the IDCT of the same decode):
// batches whose [HF ... IDCT] intervals never overlap may use one set.  A deep pipeline alternates between two owners so that the HF
// stage of batch k + 1 can run beside the IDCT of batch k.
void Batch::ShareCoefArena(Batch* owner) {
  if (prepared_) throw ParseError("ShareCoefArena after Prepare", false);
  if (coef_owner_ == owner) return;
  if (coef_owner_) { dcoef_ = nullptr; coef_cap_ = 0; coef_laid_out_ = 0; }            // (aliased the old owner's planes: not ours to free)
  if (dcoef_ && !coef_is_ext_) Pool().Give(dcoef_, coef_cap_, device_);
  dcoef_ = nullptr; coef_cap_ = 0; coef_is_ext_ = false;
  coef_owner_ = owner;
}

void Batch::UseSharedPlanes(SharedPlanes* big, SharedPlanes* coef) {
  if (big_owner_ || coef_owner_) throw ParseError("UseSharedPlanes: the batch shares another batch's arenas already", false);
  ext_big_ = big; ext_coef_ = coef;
  prepared_ = false;
}

// ---- stream-ordered completion (pipelined callers): the status words travel to pinned memory behind the decode, an event says when
void Batch::EnqueueStatusReadback(void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  const size_t n = images_.size();
  if (!status_pinned_ || status_pinned_n_ < 2 * n) {
    if (status_pinned_) (void)hipHostFree(status_pinned_);
    status_pinned_ = nullptr;
    const size_t cap = std::max<size_t>(8192, 4 * n);             // (hipHostFree waits for the device: generous, so that a refilled batch object rarely gets here)
    HIP_CHECK(hipHostMalloc((void**)&status_pinned_, cap * 4));
    status_pinned_n_ = cap;
  }
  if (!status_event_) { hipEvent_t ev; HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); status_event_ = ev; }
  if (n) {
    HIP_CHECK(hipMemcpyAsync(status_pinned_, dwork_ + status_off_, n * 4, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(status_pinned_ + n, dwork_ + hfw_off_, n * 4, hipMemcpyDeviceToHost, stream));
  }
  HIP_CHECK(hipEventRecord((hipEvent_t)status_event_, stream));
  status_pending_ = true;
  if (lf_batch_) lf_batch_->EnqueueStatusReadback(stream_v);
}
void Batch::HarvestStatus(vec<uint32_t>* per_unit) {
  const size_t n = images_.size();
  per_unit->assign(n, 0);
  if (!status_pending_) throw ParseError("HarvestStatus without EnqueueStatusReadback", false);
  HIP_CHECK(hipEventSynchronize((hipEvent_t)status_event_));
  status_pending_ = false;
  bool any = false;
  for (size_t i = 0; i < n; i++) { (*per_unit)[i] = status_pinned_[i]; any |= status_pinned_[i] != 0; }
  if (any_vardct_ && decodes_since_finish_ > 0) {
    for (size_t i = 0; i < n; i++) hf_written_[i] = status_pinned_[n + i] / decodes_since_finish_;
    decodes_since_finish_ = 0;
  }
  if (lf_batch_) { vec<uint32_t> lf; if (lf_batch_->status_pending_) { lf_batch_->HarvestStatus(&lf); for (uint32_t v : lf) if (v) for (auto& s : *per_unit) s |= v; } }
  (void)any;    // (failed frames: ZeroFailedCoefficients has put zeros back on the device, the planes stay clean for the next user)
}

// Makes *ptr a device allocation of at least `bytes` (kept if it already is; grown with 1/8 of slack otherwise; taken from the arena pool when it has a block of
// that size).  Returns true if the memory is new (contents undefined).
bool Batch::DevReserve(void** ptr, size_t* cap, size_t bytes) {
  if (*ptr && *cap >= bytes) return false;
  // (no decode of this object is in flight when it is prepared again — the contract of Reset / Prepare —, so the outgrown block is idle unless other batches share it)
  if (*ptr) { Pool().Give(*ptr, *cap, device_, /*idle=*/!(ptr == (void**)&dbig_ && big_sharers_ > 0)); *ptr = nullptr; *cap = 0; }
  // below 4 GiB capacities are powers of two (a batch object refilled with jobs of changing size settles after a few allocations), above: 1/8 of slack
  size_t want = std::max<size_t>(bytes + bytes / 8, 256);
  if (bytes < ((size_t)4 << 30)) { want = 1 << 16; while (want < bytes) want *= 2; }
  if (void* p = Pool().Take(std::max<size_t>(bytes, 256), cap, device_)) { *ptr = p; return true; }
  if (hipMalloc(ptr, want) != hipSuccess) {
    (void)hipGetLastError();
    Pool().Trim();
    if (hipMalloc(ptr, want) == hipSuccess) { *cap = want; return true; }
    (void)hipGetLastError();
    HIP_CHECK(hipMalloc(ptr, std::max<size_t>(bytes, 256))); *cap = std::max<size_t>(bytes, 256);
  } else *cap = want;
  return true;
}

// Forgets the images (and everything derived from them) but keeps the device arenas, the pinned staging buffer and the sharing set up with
// ShareBigArena / ShareCoefArena: the batch object can be filled again.  The caller makes sure no decode of the old content is in flight.
void Batch::Reset() {
  images_.clear(); pub_.clear(); cbufs_.clear(); post_ops_.clear(); jpeg_data_.clear();
  frames_host_.clear(); passes_host_.clear(); pass_first_.clear(); local_host_.clear(); local_first_.clear();
  mod_plane_offsets_.clear(); mod_ops_.clear(); vardct_alpha_.clear(); hf_written_.clear();
  // (stage timings recorded so far stay: CollectTimes sums over the object's life)
  any_complex_ = false; any_gab_ = any_vardct_ = any_modular_ = any_modchan_ = any_multipass_ = false;
  prepared_ = false; ran_once_ = false; flags_pending_ = false; decodes_since_finish_ = 0;
  cfg.idct_flags_known = 0; cfg.skip_hf = 0; cfg.max_passes = 0;
  lf_simt_ = LfSimtPlan();
  if (lf_batch_) lf_batch_->Reset();     // (kept for its arenas; Prepare drops it when no unit refers to an LF frame)
}

// Parses `n` images on `threads` host threads (the per-image work of AddImage — container, image and frame headers, TOC, the global
// sections' tables — is independent) and appends them in order.  Returns the index of the first one; throws what the first failing
// image threw.
int Batch::AddImages(const uint8_t* const* datas, const size_t* sizes, int n, int threads) {
  if (n <= 0) return (int)pub_.size();
  vec<int> index;
  std::vector<std::string> errors;
  return AddImagesImpl(datas, sizes, n, threads, /*tolerant=*/false, &index, &errors);
}

// The pipelined form: an image that does not parse (truncated, damaged headers, a feature the device path does not take) is left out instead of failing the call —
// (*index)[i] = its index in the batch or -1, (*errors)[i] = what it threw ("" if it parsed).
void Batch::AddImagesTolerant(const uint8_t* const* datas, const size_t* sizes, int n, int threads, vec<int>* index, std::vector<std::string>* errors) {
  AddImagesImpl(datas, sizes, n, threads, /*tolerant=*/true, index, errors);
}

int Batch::AddImagesImpl(const uint8_t* const* datas, const size_t* sizes, int n, int threads, bool tolerant, vec<int>* index, std::vector<std::string>* errs) {
  index->assign((size_t)std::max(n, 0), -1);
  errs->assign((size_t)std::max(n, 0), std::string());
  if (n <= 0) return (int)pub_.size();
  vec<ParsedImage> parsed((size_t)n);
  vec<std::exception_ptr> errors((size_t)n);
  const int nt = std::max(1, std::min(threads, n));
  std::atomic<int> next{0};
  const MmHooks* hooks = MmCurrent();
  auto work = [&]() {
    MmScope scope(hooks);                  // (worker threads allocate through the caller's memory manager, too)
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n) break;
      try { ParseImage(datas[i], sizes[i], &parsed[i], false); } catch (...) { errors[i] = std::current_exception(); }
    }
  };
  if (nt == 1) work();
  else {
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; t++) pool.emplace_back(work);
    for (auto& t : pool) t.join();
  }
  const int first = (int)pub_.size();
  if (!tolerant) for (int i = 0; i < n; i++) if (errors[i]) std::rethrow_exception(errors[i]);
  for (int i = 0; i < n; i++) {
    if (errors[i]) {
      try { std::rethrow_exception(errors[i]); } catch (const std::exception& e) { (*errs)[i] = e.what(); } catch (...) { (*errs)[i] = "unknown error"; }
      if ((*errs)[i].empty()) (*errs)[i] = "parse error";
      continue;
    }
    (*index)[i] = Append(std::move(parsed[i]));
  }
  return first;
}

int Batch::AddImage(const uint8_t* data, size_t size, bool allow_partial) {
  ParsedImage pi;
  ParseImage(data, size, &pi, allow_partial);
  return Append(std::move(pi));
}
bool Batch::is_partial(int i) const { return images_[pub_[i].first_unit]->plan.partial; }

int Batch::Append(ParsedImage&& im) {
  PubImage pi; pi.first_unit = (int)images_.size(); pi.num_units = (int)im.units.size(); pi.complex = im.complex;
  for (auto& u : im.units) { u->pub_index = (int)pub_.size(); images_.push_back(std::move(u)); }
  pub_.push_back(pi);
  prepared_ = false;
  return (int)pub_.size() - 1;
}

// what the device path cannot take of a frame that parsed (thrown as "unsupported: ...")
static void CheckFrameSupported(const FramePlan& p, const ImageHeader& ih) {
  if (!p.modular) {
    // squeezed extra channels (what cjxl does to a progressive or lossy alpha of an RGBA picture): the sub-channels squeezed by >= 3 ride in the LfGroup sections — the LF
    // kernel decodes them between the LF coefficients and the HF metadata —, the others in the PassGroup sections of ONE pass (ModularGroupFastKernel reads them behind that
    // pass's coefficients); downsampling entries that spread them over several passes of a VarDCT frame are not handled
    for (auto& t : p.gtransforms) if (t.id == 2 && p.num_passes > 1 && !(p.pass_min_shift[p.mod_pass] <= 0 && p.pass_max_shift[p.mod_pass] >= 2))
      throw ParseError("unsupported: squeezed extra channels of a VarDCT frame spread over several passes", true);
    if (!p.has_global_tree) throw ParseError("unsupported: VarDCT frame without a global MA tree (its LF streams would need local trees)", true);
    if (p.subsampled && (p.base_x != 0.f || p.base_b != 0.f)) throw ParseError("unsupported: chroma from luma in a chroma-subsampled frame", true);
    if (!p.local_streams.empty()) throw ParseError("unsupported: local MA tree in the global Modular stream of a VarDCT frame", true);
  }
  for (auto& x : ih.extra) {
    if (x.dim_shift != 0) throw ParseError("unsupported: subsampled extra channel", true);
    if (x.depth.is_float) {
      const uint32_t b = x.depth.bits, eb = x.depth.exp_bits;
      if (b > 32 || eb < 2 || eb > 8 || b < eb + 2 || b - eb - 1 > 23 || (b == 32 && eb != 8)) throw ParseError("unsupported: float sample layout", true);
    }
  }
  if (p.modular && ih.depth.is_float && !ih.xyb_encoded) {
    // dec_modular.cc int_to_float: the sample's bit pattern, exp_bits of exponent; what the format allows and a binary32 can hold
    const uint32_t b = ih.depth.bits, eb = ih.depth.exp_bits;
    if (b > 32 || eb < 2 || eb > 8 || b < eb + 2 || b - eb - 1 > 23 || (b == 32 && eb != 8)) throw ParseError("unsupported: float sample layout", true);
  }
  if (p.feat.has_noise && !ih.xyb_encoded) throw ParseError("noise on a non-XYB frame", false);
}

// The preview of image i (headers.cc PreviewHeader; JXL_DEC_PREVIEW_IMAGE, jpegxl-sys decode.rs:999-1025): the preview frame decoded as a one-frame image of the
// preview's size by a batch of its own — the same kernels, the frame tail for colour and write — and copied to `dst`.
ImageHeader Batch::PreviewHeaderOf(int i) const {
  const ImageEntry& e = *images_[pub_[i].first_unit];
  if (!e.ih.have_preview) throw ParseError("the image has no preview", false);
  ImageHeader ph = e.ih;
  ph.xsize = e.ih.preview_x; ph.ysize = e.ih.preview_y; ph.have_preview = false; ph.intrinsic_x = ph.intrinsic_y = 0;
  return ph;
}
size_t Batch::PreviewOutputSize(int i, const OutputSpec& o) const { return OutputSize(PreviewHeaderOf(i), o); }
void Batch::DecodePreview(int i, const OutputSpec& o, void* dst, size_t cap, void* stream_v) {
  const ImageEntry& e = *images_[pub_[i].first_unit];
  std::shared_ptr<ImageShared> sh(new ImageShared());
  sh->cs = e.cs; sh->ih = PreviewHeaderOf(i);
  std::unique_ptr<ImageEntry> u(new ImageEntry(sh));
  u->frame_bitpos = e.preview_bitpos; u->frame_index = 0; u->complex = true;
  u->visible_frame_index = 1; u->nonvisible_frame_index = 0;
  ParseFrameStart(sh->cs, sh->ih, u->frame_bitpos, &u->plan);
  if (u->plan.frame_type != 0 || u->plan.use_lf_frame) throw ParseError("the preview must be a regular frame", false);
  u->plan.is_last = true;                              // (whatever the frame says: nothing of the image follows it as far as the preview goes)
  CheckFrameSupported(u->plan, sh->ih);
  if (sh->ih.extra.size() > 4) throw ParseError("unsupported: more than 4 extra channels in a preview", true);
  Batch tmp(device_);
  ParsedImage pi; pi.complex = true; pi.units.push_back(std::move(u));
  tmp.Append(std::move(pi));
  tmp.SetOutput(0, o);
  if (tmp.image(0).out_size > cap) throw ParseError("preview output buffer too small", false);
  tmp.Prepare(stream_v);
  tmp.Run(stream_v);
  tmp.Finish(stream_v);
  tmp.CopyOutputToHost(0, dst, tmp.image(0).out_size, stream_v);
}

void Batch::ParseImage(const uint8_t* data, size_t size, ParsedImage* out, bool allow_partial) {
  std::shared_ptr<ImageShared> sh(new ImageShared());
  bool have_container = false, has_jbrd = false;
  if (!ExtractCodestream(data, size, &sh->cs, &have_container, &has_jbrd, &sh->boxes) && !allow_partial) throw ParseError("truncated", false);
  uint64_t bitpos = 0, preview_bitpos = 0;
  ParseImageHeader(sh->cs, &sh->ih, &bitpos);
  sh->ih.have_container = have_container;
  const ImageHeader& ih = sh->ih;
  if (ih.xyb_encoded) {
    // stage_from_linear.cc: sRGB, linear, pure gamma (incl. DCI), Rec.709, PQ and HLG
    const bool ok = ih.color_default || ih.want_icc || ih.have_gamma || ih.tf == 13 || ih.tf == 8 || ih.tf == 17 || ih.tf == 1 || ih.tf == 16 || ih.tf == 18;
    if (!ok) throw ParseError("unsupported: output transfer function", true);
  }
  if (ih.have_preview) {
    // decode.cc: the preview is a frame of its own in front of the image's frames, its default size the PreviewHeader's.  It is decoded for callers that
    // subscribe to JXL_DEC_PREVIEW_IMAGE only — jpegxl-rs never does (decode.rs:334-347) —, so the frame is stepped over: header + TOC say where it ends.
    ImageHeader ph = ih;
    ph.xsize = ih.preview_x; ph.ysize = ih.preview_y;
    FramePlan pp;
    ParseFrameStart(sh->cs, ph, bitpos, &pp, /*header_and_toc_only=*/true);
    if (pp.frame_type != 0) throw ParseError("the preview must be a regular frame", false);
    preview_bitpos = bitpos;
    bitpos = pp.frame_end_bitpos;
  }
  // every frame of the image (frame_header.cc): reference-only / zero-duration layers first, the last one is displayed
  vec<std::unique_ptr<ImageEntry>> units;
  std::shared_ptr<ImageEntry> lf_frames[5];     // dec_cache.h dc_frames: the latest LF frame of every level (1 .. 4)
  uint32_t visible = 0, nonvisible = 0;
  for (int k = 0;; k++) {
    if (k >= 4096) throw ParseError("unsupported: more than 4096 frames", true);     // (a bound on what one image may make the host allocate; every frame is a unit of the batch)
    std::unique_ptr<ImageEntry> e(new ImageEntry(sh));
    e->has_jbrd = has_jbrd;
    e->frame_bitpos = bitpos;
    e->frame_index = (int)units.size();   // (position among the units of the image: LF frames are none)
    e->pub_index = 0;                     // (set when the image is appended)
    // (allow_partial: a single-frame image cut off inside its PassGroup sections parses as far as its LF part — what a progressive flush shows)
    ParseFrameStart(sh->cs, ih, bitpos, &e->plan, /*header_and_toc_only=*/false, /*allow_partial=*/allow_partial && units.empty());
    const FramePlan& p = e->plan;
    if (p.partial && !(p.frame_type == 0 && p.is_last && !p.have_crop)) throw ParseError("truncated", false);
    if (p.frame_type == 0 || p.frame_type == 3) { visible++; nonvisible = 0; } else nonvisible++;
    e->visible_frame_index = visible; e->nonvisible_frame_index = nonvisible;
    if (p.use_lf_frame) {
      // frame_header.cc kUseDcFrame: the LF image is the LF frame of the next level's samples, one per 8x8 block of this frame
      const std::shared_ptr<ImageEntry>& src = lf_frames[p.lf_level + 1];
      if (!src) throw ParseError("the LF frame this frame refers to is not in the stream", false);
      if (p.subsampled || src->plan.width != p.bw || src->plan.height != p.bh) throw ParseError("LF frame of the wrong size", false);
      e->lf_source = src;
    }
    CheckFrameSupported(p, ih);
    const bool last = p.is_last;
    bitpos = p.frame_end_bitpos;
    if (p.frame_type == 1) {              // an LF frame: kept aside for the frames that refer to it (decoded by Batch::lf_batch_), never displayed
      // (every frame carries the image's extra channels, an LF frame too — libjxl's encoder fills them with zeros there; they are decoded with the frame and not looked at)
      if (ih.extra.size() > 4) throw ParseError("unsupported: LF frame of an image with more than 4 extra channels", true);
      lf_frames[p.lf_level] = std::shared_ptr<ImageEntry>(e.release());
      continue;
    }
    units.push_back(std::move(e));
    if (last) break;
  }
  bool complex = units.size() > 1;
  for (auto& u : units) {
    const FramePlan& p = u->plan;
    bool replace_all = p.blend.mode == 0;
    for (auto& b : p.ec_blend) if (b.mode != 0) replace_all = false;
    if ((p.flags & (1 | 2 | 16)) || p.have_crop || !replace_all || p.frame_type != 0) complex = true;
    if (p.modular && ih.xyb_encoded) complex = true;      // XYB Modular frames go through the float planes
    if (p.modular && p.upsampling != 1) complex = true;
    // (chroma-subsampled frames without anything else: OutputKernel upsamples the chroma planes as it reads them; in the frame tail of complex images ChromaUpsampleKernel does)
  }
  for (auto& x : ih.extra) if (x.depth.is_float) complex = true;   // float extra channels are converted in the frame tail (IntToFloatSample)
  if (complex && ih.extra.size() > 4) throw ParseError("unsupported: more than 4 extra channels in a multi-frame / feature image", true);
  for (auto& u : units) { u->complex = complex; u->preview_bitpos = preview_bitpos; }
  out->units = std::move(units);
  out->complex = complex;
}

size_t Batch::OutputStride(const ImageHeader& ih, const OutputSpec& o, uint32_t* channels) {
  uint32_t nc = o.num_channels;
  bool alpha = false;
  for (auto& e : ih.extra) if (e.type == 0) alpha = true;
  if (o.alpha_from_extra >= 0 && (size_t)o.alpha_from_extra < ih.extra.size()) alpha = true;
  if (nc == 0) nc = (ih.color_space == 1 ? 1 : 3) + (alpha ? 1 : 0);
  if (channels) *channels = nc;
  const size_t bps = o.type == 0 ? 1 : o.type == 2 ? 4 : 2;
  size_t stride = (size_t)OrientedWidth(ih, o) * nc * bps;
  if (o.align > 1) stride = (stride + o.align - 1) / o.align * o.align;
  return stride;
}
// Orientations 5..8 transpose the image (codestream_header.rs JxlOrientation); applied unless the caller keeps it.
uint32_t Batch::OrientedWidth(const ImageHeader& ih, const OutputSpec& o) { return (!o.keep_orientation && ih.orientation > 4) ? ih.ysize : ih.xsize; }
uint32_t Batch::OrientedHeight(const ImageHeader& ih, const OutputSpec& o) { return (!o.keep_orientation && ih.orientation > 4) ? ih.xsize : ih.ysize; }
size_t Batch::OutputSize(const ImageHeader& ih, const OutputSpec& o) {
  // jpegxl-sys decode.rs:1100 JxlDecoderImageOutBufferSize: stride * (h - 1) + w * C * bytes
  uint32_t nc;
  const size_t stride = OutputStride(ih, o, &nc);
  const size_t bps = o.type == 0 ? 1 : o.type == 2 ? 4 : 2;
  return stride * (OrientedHeight(ih, o) - 1) + (size_t)OrientedWidth(ih, o) * nc * bps;
}
void Batch::OutputDims(int i, const OutputSpec& o, uint32_t* w, uint32_t* h) const {
  const ImageEntry& first = *images_[pub_[i].first_unit];
  *w = first.ih.xsize; *h = first.ih.ysize;
  if (o.only_frame >= 0 && o.only_frame < pub_[i].num_units) { const FramePlan& p = images_[pub_[i].first_unit + o.only_frame]->plan; *w = p.frame_w; *h = p.frame_h; }
}
size_t Batch::OutputSizeOf(int i, const OutputSpec& o) const {
  ImageHeader dims = images_[pub_[i].first_unit]->ih;       // (OutputStride / OutputSize only look at the size, the orientation and the channel list)
  OutputDims(i, o, &dims.xsize, &dims.ysize);
  return OutputSize(dims, o);
}
static float IntMul(const OutputSpec& o) { const uint32_t full = o.type == 0 ? 8 : 16; const uint32_t b = o.int_bits && o.int_bits < full ? o.int_bits : full; return (float)((1u << b) - 1); }
void Batch::SetOutput(int i, const OutputSpec& o) {
  ImageEntry& e = *images_[pub_[i].first_unit];
  e.out = o;
  if (o.only_frame >= pub_[i].num_units) throw ParseError("non-coalesced output: no such frame", false);
  // (a lone frame that fills the image is the same either way: the plain path keeps it)
  if (o.only_frame == 0 && pub_[i].num_units == 1 && !e.plan.have_crop) e.out.only_frame = -1;
  if (o.upto_frame >= pub_[i].num_units - 1 || o.only_frame >= 0) e.out.upto_frame = -1;     // (the last frame's composite is the image)
  ImageHeader dims = e.ih;
  OutputDims(i, e.out, &dims.xsize, &dims.ysize);
  uint32_t nc;
  e.out_stride = OutputStride(dims, o, &nc);
  e.out.num_channels = nc;
  e.out_size = OutputSize(dims, o);
  e.deliver_frames.clear();
  prepared_ = false;
}
void Batch::SetOutputAllFrames(int i, const OutputSpec& o, const vec<int>& frames) {
  if (o.device_ptr || o.only_frame >= 0 || frames.empty()) throw ParseError("SetOutputAllFrames: internal output buffers, coalesced frames only", false);
  for (size_t k = 0; k < frames.size(); k++)
    if (frames[k] < 0 || frames[k] >= pub_[i].num_units || (k && frames[k] <= frames[k - 1])) throw ParseError("SetOutputAllFrames: frame list must be ascending positions among the image's frames", false);
  OutputSpec oo = o;
  oo.upto_frame = -1;
  SetOutput(i, oo);
  images_[pub_[i].first_unit]->deliver_frames = frames;
}

uint64_t Batch::total_pixels() const { uint64_t n = 0; for (auto& pi : pub_) { const ImageHeader& ih = images_[pi.first_unit]->ih; n += (uint64_t)ih.xsize * ih.ysize; } return n; }
uint64_t Batch::compressed_bytes() const { uint64_t n = 0; for (auto& pi : pub_) n += images_[pi.first_unit]->cs.size; return n; }
void Batch::StageBytes(uint64_t out[6]) const {
  // Compulsory (ALGORITHMIC) HBM traffic of each stage as the kernels are actually structured — every input the stage cannot avoid
  // reading once, every output written once:
  //  lf     LfGroup sections -> quantised LF (3 x i32) + block info + coefficient offset per 8x8 block
  //  lfpost LF dequantisation, smoothing, LLF, EPF sigma: 76 B per block
  //  hf     PassGroup sections -> the non-zero coefficients (i32 each; counted by the kernel, known after the first Finish)
  //  idct   coefficient planes as stored (dense i32: 12 B/px) -> 3 f32 planes (12 B/px)
  //  filter gaborish + EPF (+ colour + write when the stage's last kernel writes the pixels — the fused kernel of the headline, the tiled EPF passes of every other
  //         plain frame): 12 B/px in, C x bytes out — ONCE, however many kernels the stage is split into (their intermediate planes are traffic, not algorithmic
  //         bytes: SURVEY.md 8(d)).  Frames whose filters end in the planes (upsampling, the frame tail of complex images): 12 B/px in, 12 B/px out
  //  out    those frames only: 12 B/px in, C x bytes out
  for (int i = 0; i < 6; i++) out[i] = 0;
  for (size_t u = 0; u < images_.size(); u++) {
    const ImageEntry& e = *images_[u];
    const FramePlan& p = e.plan;
    if (p.modular) continue;
    const ImageEntry& first = *images_[pub_[e.pub_index].first_unit];
    const uint64_t nblk = (uint64_t)p.bw * p.bh, npx = (uint64_t)p.width * p.height;
    uint64_t lf_sec = 0, hf_sec = 0;
    if (p.single_section) { lf_sec = e.cs.size / 4; hf_sec = e.cs.size; }
    else {
      for (uint32_t s = 1; s < 1 + p.num_lf_groups; s++) lf_sec += p.sections[s].size;
      for (size_t s = 2 + p.num_lf_groups; s < p.sections.size(); s++) hf_sec += p.sections[s].size;
    }
    const uint64_t bps = first.out.type == 0 ? 1 : first.out.type == 2 ? 4 : 2;
    const uint64_t out_px = (uint64_t)first.out.num_channels * bps;
    out[0] += lf_sec + nblk * (3 * 4 + 4 + 4);
    out[1] += nblk * (12 + 12 + 12 + 12 + 12 + 16);
    out[2] += hf_sec + (u < hf_written_.size() ? (uint64_t)hf_written_[u] * 4 : 0);
    out[3] += npx * (12 + 12);
    const bool fused = p.lf.gab && p.lf.epf_iters == 1 && e.ih.xyb_encoded && SimpleTransfer(e.ih) && p.upsampling == 1 && !e.complex && !cfg.force_unfused_filters;
    const bool any_filter = p.lf.gab || p.lf.epf_iters > 0;
    const bool epf_writes = !fused && p.lf.epf_iters >= 1 && p.upsampling == 1 && !e.complex && cfg.debug_stop_after == 0;    // kernels.hip EpfWritesOutput
    if (fused || epf_writes) out[4] += npx * (12 + out_px);
    else {
      if (any_filter) out[4] += npx * 24;
      out[5] += npx * (12 + out_px);
    }
  }
}

// Testing: copies one of the device buffers of image i's first frame to the host (after a decode, JxlHipBatchDebugRead): the planes
// between the stages ("debug_stop_after") and the inputs of the pixel stages.  Returns the bytes the buffer holds; copies min(that, cap).
size_t Batch::DebugRead(int i, const std::string& name, int c, void* dst, size_t cap, void* stream_v) {
  if (!prepared_ || i < 0 || (size_t)i >= pub_.size() || c < 0 || (c > 2 && name != "qtable") || c >= 51) throw ParseError("DebugRead: no such image / channel", false);
  if (name == "qtable") {     // dequantisation table (1 / weight) of quant kind c / 3, channel c % 3: 64 x rows x cols floats (kKindRows / kKindCols)
    const FrameDev& f0 = frames_host_[pub_[i].first_unit];
    const size_t bytes = (size_t)64 * kKindRows[c / 3] * kKindCols[c / 3] * 4;
    if (!f0.qtable[c]) throw ParseError("DebugRead: no such table", false);
    HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_v));
    if (dst && cap) HIP_CHECK(hipMemcpy(dst, f0.qtable[c], std::min(bytes, cap), hipMemcpyDeviceToHost));
    return bytes;
  }
  const int u = pub_[i].first_unit;
  const FrameDev& f = frames_host_[u];
  const FramePlan& p = images_[u]->plan;
  if (p.modular) throw ParseError("DebugRead: not a VarDCT frame", false);
  const size_t nb = (size_t)p.bw * p.bh, npx = nb * 64;
  const void* src = nullptr; size_t bytes = 0;
  if (name == "plane_a") { src = f.plane_a[c]; bytes = npx * 4; }
  else if (name == "plane_b") { src = f.plane_b[c]; bytes = npx * 4; }
  else if (name == "lf") { src = f.lf[c]; bytes = nb * 4; }
  else if (name == "lf_smooth") { src = f.lf_tmp[c]; bytes = nb * 4; }     // the LF samples after the adaptive smoothing
  else if (name == "llf") { src = f.llf[c]; bytes = nb * 4; }
  else if (name == "lfq") { src = f.lfq[c]; bytes = nb * 4; }
  else if (name == "inv_sigma") { src = f.inv_sigma; bytes = nb * 4; }
  else if (name == "blk_info") { src = f.blk_info; bytes = nb * 4; }
  else if (name == "coef_off") { src = f.coef_off; bytes = nb * 4; }
  else if (name == "coeff") { src = f.coeff[c]; bytes = (size_t)p.num_groups * 65536 * 4; }
  else if (name == "qtable") { throw ParseError("DebugRead: qtable takes kind * 3 + channel", false); }
  else if (name == "up_weights") { src = f.up_weights; bytes = p.upsampling == 2 ? 60 : p.upsampling == 4 ? 220 : p.upsampling == 8 ? 840 : 0; }   // the 15 / 55 / 210 stored upsampling weights in use
  else if (name == "ytox") { src = f.ytox; bytes = (size_t)((p.bw + 7) / 8) * ((p.bh + 7) / 8); }
  else if (name == "ytob") { src = f.ytob; bytes = (size_t)((p.bw + 7) / 8) * ((p.bh + 7) / 8); }
  else throw ParseError("DebugRead: unknown buffer " + name, false);
  if (!src) throw ParseError("DebugRead: buffer " + name + " is not allocated for this batch", false);
  HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_v));
  if (dst && cap) HIP_CHECK(hipMemcpy(dst, src, std::min(bytes, cap), hipMemcpyDeviceToHost));
  return bytes;
}

int64_t Batch::Info(const std::string& name) const {
  if (name == "lf_simt_frames" || name == "lf_legacy_frames") {
    int64_t simt = 0, legacy = 0;
    for (size_t i = 0; i < frames_host_.size() && i < images_.size(); i++) if (!images_[i]->plan.modular) (frames_host_[i].lf_simt ? simt : legacy)++;
    return name == "lf_simt_frames" ? simt : legacy;
  }
  if (name == "hf_nonzeros") { int64_t t = 0; for (uint32_t v : hf_written_) t += v; return t; }   // non-zero AC coefficients per decode of the batch (known after a Finish)
  if (name == "mod_group_lds_bytes") return prepared_ ? (int64_t)ModularGroupLdsBytes(cfg) : -1;
  if (name == "lf_simt_lanes") return lf_simt_.num_lanes;
  if (name == "lf_simt_wp") return lf_simt_.num_lanes ? lf_simt_.any_wp : 0;     // the SIMT launch is the weighted-predictor instantiation
  if (!images_.empty() && !images_[0]->plan.modular) {   // geometry / quantiser of the first frame (tests that restate a stage from its defining formula)
    const FramePlan& p = images_[0]->plan;
    if (name == "frame0_bw") return p.bw;
    if (name == "frame0_bh") return p.bh;
    if (name == "frame0_global_scale") return p.global_scale;
    if (name == "frame0_quant_lf") return p.quant_lf;
    if (name == "frame0_color_factor") return p.color_factor;
    if (name == "frame0_x_qm_scale") return p.x_qm_scale;
    if (name == "frame0_b_qm_scale") return p.b_qm_scale;
    if (name == "frame0_xgroups") return p.xgroups;
  }
  if (name == "lf_simt_waves") return lf_simt_.num_lanes ? (lf_simt_.num_lanes + lf_simt_.lanes_per_wave - 1) / lf_simt_.lanes_per_wave : 0;
  return -1;
}

void* Batch::device_output(int i) const {
  const ImageEntry& e = *images_[pub_[i].first_unit];
  return e.out.device_ptr ? e.out.device_ptr : (void*)(dwork_ + e.off_out);
}

void Batch::Prepare(void* stream_v, bool wait_upload) {
  hipStream_t stream = (hipStream_t)stream_v;
  HIP_CHECK(hipSetDevice(device_));
  InitDeviceTables(stream_v);
  // Device allocations are kept from one Prepare to the next and only replaced when they are too small (DevReserve): a batch object
  // that is refilled step after step (JxlHipBatchReset + AddImages + Prepare, the streaming loop of bench.py) allocates once —
  // hipFree synchronises the device, hipMalloc of gigabytes takes milliseconds.
  if (clear_stream_) (void)hipStreamSynchronize((hipStream_t)clear_stream_);   // a pending clear of the old coefficient planes
  clear_pending_ = false;
  if (coef_owner_) dcoef_ = nullptr;
  if (big_owner_) dbig_ = nullptr;
  if (big_is_ext_) { dbig_ = nullptr; big_cap_ = 0; big_is_ext_ = false; }          // (pointers into the pipeline's planes: decided anew below)
  if (coef_is_ext_) { dcoef_ = nullptr; coef_cap_ = 0; coef_laid_out_ = 0; coef_is_ext_ = false; }
  const int n = (int)images_.size();
  // JXL_HIP_TIME_PREPARE=1: host milliseconds of the phases of this function on stderr
  const bool time_phases = getenv("JXL_HIP_TIME_PREPARE") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  std::string t_report;
  auto mark = [&](const char* what) {
    if (!time_phases) return;
    const auto now = std::chrono::steady_clock::now();
    char buf[64];
    snprintf(buf, sizeof buf, " %s %.2f", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_report += buf; t_last = now;
  };
  // un-premultiplying alpha (JxlDecoderSetUnpremultiplyAlpha, jpegxl-rs decode.rs:353) happens in the write stage of the frame tail
  for (PubImage& pi : pub_) {
    ImageEntry& first = *images_[pi.first_unit];
    bool premul = false;
    for (auto& x : first.ih.extra) if (x.type == 0) { premul = x.alpha_associated; break; }
    bool spot = false;
    for (auto& x : first.ih.extra) if (x.type == 2) spot = true;
    if (spot && first.out.render_spotcolors) {
      // stage_spot.cc runs in the frame tail; grey images keep one colour plane there (not handled)
      if (first.ih.color_space == 1) throw ParseError("unsupported: spot colours on a grey image", true);
      if (first.ih.extra.size() > 4) throw ParseError("unsupported: more than 4 extra channels with spot colours", true);
      if (!pi.complex) { pi.complex = true; for (int u = pi.first_unit; u < pi.first_unit + pi.num_units; u++) images_[u]->complex = true; }
    }
    if (first.out.only_frame >= 0 && !pi.complex) {     // a single frame as coded: written by the frame tail (its own size, no blending)
      if (first.ih.extra.size() > 4) throw ParseError("unsupported: more than 4 extra channels with non-coalesced output", true);
      pi.complex = true;
      for (int u = pi.first_unit; u < pi.first_unit + pi.num_units; u++) images_[u]->complex = true;
    }
    if (first.out.unpremul_alpha && premul && !pi.complex) {
      if (first.ih.extra.size() > 4) throw ParseError("unsupported: more than 4 extra channels with un-premultiplied output", true);
      pi.complex = true;
      for (int u = pi.first_unit; u < pi.first_unit + pi.num_units; u++) images_[u]->complex = true;
    }
  }
  hconst_.clear();
  {
    size_t guess = (size_t)1 << 20;      // streams + tables: one allocation instead of a dozen doublings
    for (auto& e : images_) { guess += sizeof(FrameDev) + 4 * sizeof(PassDev) + 1024; if (e->frame_index == 0) guess += e->cs.padded_size() + ((size_t)256 << 10); }
    hconst_.Reserve(guess);
  }
  Arena arena(hconst_);
  vec<ConstOffsets> co(n);
  // natural coefficient orders (shared)
  size_t natural_off[13];
  {
    // (process-wide: the orders of the DCT128/256 buckets are 16K..64K entries each)
    static std::mutex mu;
    static std::vector<uint16_t> natural[13];   // process-lifetime cache: never through a caller's allocator
    std::lock_guard<std::mutex> lock(mu);
    for (int b = 0; b < 13; b++) {
      if (natural[b].empty()) { const vec<uint16_t> v = NaturalCoeffOrder(kBucketStrategy[b]); natural[b].assign(v.begin(), v.end()); }
      natural_off[b] = arena.Put(natural[b].data(), natural[b].size() * 2);
    }
  }
  // quant tables are shared between frames with identical specs
  struct QCache { const QuantTableSpec* spec; int kind; size_t off[3]; };
  vec<QCache> qcache;
  auto spec_equal = [](const QuantTableSpec& a, const QuantTableSpec& b) {
    if (a.mode != b.mode || a.num_bands != b.num_bands || a.num_bands4 != b.num_bands4) return false;
    if (memcmp(a.bands, b.bands, sizeof(a.bands)) || memcmp(a.idw, b.idw, sizeof(a.idw)) || memcmp(a.dct2w, b.dct2w, sizeof(a.dct2w))) return false;
    if (memcmp(a.dct4mul, b.dct4mul, sizeof(a.dct4mul)) || memcmp(a.dct4x8mul, b.dct4x8mul, sizeof(a.dct4x8mul))) return false;
    if (a.mode == 5 && (memcmp(a.afvw, b.afvw, sizeof(a.afvw)) || memcmp(a.bands4, b.bands4, sizeof(a.bands4)))) return false;
    if (a.mode == 7 && (a.raw_den != b.raw_den || a.raw[0] != b.raw[0] || a.raw[1] != b.raw[1] || a.raw[2] != b.raw[2])) return false;
    return true;
  };
  max_lf_groups_ = max_groups_ = max_w_ = max_h_ = max_bw_ = max_bh_ = max_epf_ = 0;
  any_gab_ = any_vardct_ = any_modular_ = any_modchan_ = any_multipass_ = false;
  fplan_ = FilterPlan();
  for (int i = 0; i < n; i++) {
    ImageEntry& e = *images_[i];
    FramePlan& p = e.plan;
    ConstOffsets& c = co[i];
    if (e.frame_index == 0 && e.out_size == 0) SetOutput(e.pub_index, e.out);
    if (e.frame_index == 0) c.cs = arena.Put(e.cs.data(), e.cs.padded_size()); else c.cs = co[i - e.frame_index].cs;   // frames share the codestream
    vec<uint64_t> so, ss;
    for (auto& s : p.sections) { so.push_back(s.offset); ss.push_back(s.size); }
    c.sec_off = arena.Put(so.data(), so.size() * 8);
    c.sec_size = arena.Put(ss.data(), ss.size() * 8);
    if (p.has_global_tree) {
      c.tree = arena.Put(p.tree.nodes.data(), p.tree.nodes.size() * sizeof(TreeNode));
      PutCode(arena, p.tree_code, &c.mod_ctx, &c.mod_cfg, &c.mod_alias, &c.mod_pc, &c.mod_po, &c.mod_ps);
    }
    for (auto& ls : p.local_streams) {
      ConstOffsets::Local l;
      l.unit = ls.unit;
      l.tree = arena.Put(ls.tree.nodes.data(), ls.tree.nodes.size() * sizeof(TreeNode));
      PutCode(arena, ls.code, &l.ctx, &l.cfg, &l.alias, &l.pc, &l.po, &l.ps);
      c.local.push_back(l);
    }
    c.bcm = arena.Put(&p.bcm, sizeof(p.bcm));
    if (!p.modular) {
      any_vardct_ = true;
      // (single-section frames: HfGlobal is parsed in the pre-run below; reserve nothing here)
    } else any_modular_ = true;
    max_lf_groups_ = std::max<int>(max_lf_groups_, p.num_lf_groups);
    max_groups_ = std::max<int>(max_groups_, p.num_groups);
    max_w_ = std::max<int>(max_w_, p.width); max_h_ = std::max<int>(max_h_, p.height);
    max_bw_ = std::max<int>(max_bw_, p.bw); max_bh_ = std::max<int>(max_bh_, p.bh);
    if (!p.modular) {
      max_epf_ = std::max<int>(max_epf_, p.lf.epf_iters); any_gab_ |= p.lf.gab != 0;
      const bool fusable = p.lf.gab && p.lf.epf_iters == 1 && e.ih.xyb_encoded && SimpleTransfer(e.ih) && p.upsampling == 1 && !e.complex;
      if (p.upsampling > 1 && !e.complex) { fplan_.any_upsampled = true; fplan_.max_out_w = std::max<int>(fplan_.max_out_w, e.ih.xsize); fplan_.max_out_h = std::max<int>(fplan_.max_out_h, e.ih.ysize); }
      fplan_.any_fused |= fusable; fplan_.any_unfused |= !fusable;
      fplan_.any_gab |= p.lf.gab != 0; fplan_.max_epf = std::max<int>(fplan_.max_epf, p.lf.epf_iters);
    }
  }
  mark("tables1");
  // ---- work arena layout: `take` = the per-batch arena (everything the LF stage writes, scratch, status, outputs);
  // `take_big` = coefficient and pixel planes, only touched between HF decode and the write stage (shareable)
  size_t w = 0, wbig = 0;
  auto take = [&](size_t bytes) { size_t off = Align(w); w = off + bytes; return off; };
  auto take_big = [&](size_t bytes) { size_t off = Align(wbig); wbig = off + bytes; return off; };
  const bool need_plane_b = fplan_.any_unfused || cfg.force_unfused_filters;
  struct WorkOffsets {
    size_t lfq[3], lf[3], lf_tmp[3], llf[3], blk_info, coef_off, vb_list, vb_count, ytox, ytob, coeff[3], plane_a[3], plane_b[3], inv_sigma, lf_scratch, wp_scratch, end_bitpos,
        place_rec = 0, place_cnt = 0, band_start = 0,
        mod_scratch, hf_end = 0, mod_wp = 0, up_plane[4] = {0, 0, 0, 0};
    size_t lf_scratch_stride, wp_scratch_stride, mod_scratch_stride, mod_wp_stride = 0, lz_window = (size_t)-1, lz_ac_window = (size_t)-1;
  };
  vec<WorkOffsets> wo(n);
  mod_plane_offsets_.assign(n, {});
  mod_ops_.assign(n, {});
  cbufs_.assign(n, ComplexBufs());
  for (auto& cb : cbufs_) for (size_t* a : {cb.ecf, cb.up_ec, cb.canvas_ec}) for (int k = 0; k < 4; k++) a[k] = (size_t)-1;
  for (auto& cb : cbufs_) for (size_t* a : {cb.up, cb.noise, cb.rgb, cb.canvas, cb.pa, cb.pb, cb.color_int}) for (int k = 0; k < 3; k++) a[k] = (size_t)-1;
  post_ops_.clear(); any_complex_ = false;
  vardct_alpha_.assign(n, VarDctAlpha());
  status_off_ = take((size_t)n * 4);
  const size_t flags_off = take((size_t)n * 4);
  flags_off_ = flags_off;
  hfw_off_ = take((size_t)n * 4);
  hf_written_.assign(n, 0); decodes_since_finish_ = 0;
  cfg.idct_flags_known = 0; ran_once_ = false; flags_pending_ = false;
  // coefficient buffers of all frames are contiguous so that one memset clears them
  // quantised coefficients: an arena of this batch's own (never shared: it is cleared for the batch's next decode on an
  // internal stream while the next batch's HF stage runs, see ClearCoefficients*)
  size_t wcoef = 0;
  for (int i = 0; i < n; i++) {
    const FramePlan& p = images_[i]->plan;
    if (p.modular) continue;
    for (int c = 0; c < 3; c++) { wo[i].coeff[c] = Align(wcoef); wcoef = wo[i].coeff[c] + (size_t)p.num_groups * 65536 * 4; }
  }
  coeff_bytes_ = Align(wcoef);
  for (int i = 0; i < n; i++) {
    ImageEntry& e = *images_[i];
    const FramePlan& p = e.plan;
    WorkOffsets& o = wo[i];
    o.end_bitpos = take(16);
    if (e.frame_index == 0 && !e.out.device_ptr) e.off_out = take((e.out_size + 64) * std::max<size_t>(1, e.deliver_frames.size()));
  if (p.upsampling > 1) {   // kernel weights: custom (image header) or library default
      const int upk = p.upsampling == 2 ? 0 : p.upsampling == 4 ? 1 : 2;
      const float* const kDefault[3] = {kUp2, kUp4, kUp8};
      static const size_t kCount[3] = {15, 55, 210};
      const vec<float>& cw = e.ih.up_weights[upk];
      co[i].up_weights = cw.empty() ? arena.Put(kDefault[upk], kCount[upk] * 4) : arena.Put(cw.data(), cw.size() * 4);
    }
    if (!p.modular) {
      const size_t nb = (size_t)p.bw * p.bh;
      for (int c = 0; c < 3; c++) { o.lfq[c] = take(nb * 4); o.lf[c] = take(nb * 4); o.lf_tmp[c] = take(nb * 4); o.llf[c] = take(nb * 4); }
      o.blk_info = take(nb * 4); o.coef_off = take(nb * 4); o.inv_sigma = take(nb * 4);
      o.vb_list = take((size_t)p.num_groups * 1024 * 8); o.vb_count = take((size_t)p.num_groups * 4);
      o.place_rec = take(nb * 16); o.place_cnt = take((size_t)p.num_lf_groups * 8 * 4); o.band_start = take((size_t)p.num_lf_groups * 8 * 4);
      const size_t ntile = (size_t)((p.bw + 7) / 8) * ((p.bh + 7) / 8);
      o.ytox = take(ntile); o.ytob = take(ntile);
      const size_t plane = (size_t)p.bw * 8 * p.bh * 8 * 4;
      for (int c = 0; c < 3; c++) { o.plane_a[c] = take_big(plane); o.plane_b[c] = (need_plane_b || e.complex) ? take_big(plane) : (size_t)-1; }
      if (p.upsampling > 1 && !e.complex) for (int c = 0; c < 4; c++) o.up_plane[c] = take_big((size_t)e.ih.xsize * e.ih.ysize * 4);
      o.lf_scratch_stride = 16 + 2 * 1024 + 3 * 65536;
      o.lf_scratch = take(o.lf_scratch_stride * 4 * p.num_lf_groups);
      // weighted-predictor state of an LF-group stream.  LfDecodeSimtKernel: two rows of 258 records of 32 bytes; LfDecodeKernel's global
      // form (channels wider than its LDS slot — only the block-info rows, up to 65 536 wide, can be): five arrays of 2 x (width + 2) ints
      bool wide_wp = false;
      if (p.tree.uses_wp && p.has_global_tree)
        for (uint32_t g = 0; g < p.num_lf_groups && !wide_wp; g++) wide_wp = LfChannelUsesWp(p.tree, 2, 1 + 2 * p.num_lf_groups + g);
      o.wp_scratch_stride = wide_wp ? 10 * (65536 + 2) : kLfSimtWpInts;
      o.wp_scratch = take(o.wp_scratch_stride * 4 * p.num_lf_groups);
      if (!p.gchannels.empty()) {
        // extra channels (alpha, ...) ride in the frame's Modular sub-streams: planes, channel table, undo plan
        any_modchan_ = true;
        vec<size_t>& mp = mod_plane_offsets_[i];
        for (auto& ch : p.gchannels) mp.push_back(take((size_t)ch.w * ch.h * 4 + 64));
        PlanModularUndo(i, take);
        vec<ModChanDev> table;
        for (size_t k = 0; k < p.gchannels.size(); k++) table.push_back(ModChanDev{mp[k], p.gchannels[k].w, p.gchannels[k].h, p.gchannels[k].hshift, p.gchannels[k].vshift});
        co[i].mod_chan = arena.Put(table.data(), table.size() * sizeof(ModChanDev));
        const size_t gd = p.group_dim;
        o.mod_scratch_stride = (8 + 4) * gd * gd + 4 * 65536;
        o.mod_scratch = take(o.mod_scratch_stride * 4 * p.NumModUnits());
        o.hf_end = take((size_t)p.num_groups * 8);
        // the Modular streams keep their WP state apart from the LF streams'
        o.mod_wp_stride = p.tree.uses_wp ? 10 * (65536 + 2) : 16;
        o.mod_wp = take(o.mod_wp_stride * 4 * (1 + p.NumModUnits()));
      }
    } else {
      any_modchan_ = true;
      // planes for every channel of the global image, then the plan that undoes the global transforms
      vec<size_t>& mp = mod_plane_offsets_[i];
      for (auto& ch : p.gchannels) mp.push_back(take((size_t)ch.w * ch.h * 4 + 64));
      PlanModularUndo(i, take);
      vec<ModChanDev> table;
      for (size_t k = 0; k < p.gchannels.size(); k++) table.push_back(ModChanDev{mp[k], p.gchannels[k].w, p.gchannels[k].h, p.gchannels[k].hshift, p.gchannels[k].vshift});
      co[i].mod_chan = arena.Put(table.data(), table.size() * sizeof(ModChanDev));
      const size_t gd = p.group_dim;
      // per-unit scratch: channels of units with transforms of their own are decoded there (worst case 12 planes of a group + palettes: 4 MB per unit, 4.3 GB for the 1040
      // units of an 8192x8192 frame) — the host has read every unit's header: frames without such units get none; weighted-predictor rows: two rows x five arrays of the widest channel
      const bool tight = p.mod_units_scanned && !p.mod_local_transforms;
      o.mod_scratch_stride = tight ? 64 : (8 + 4) * gd * gd + 4 * 65536;
      o.mod_scratch = take(o.mod_scratch_stride * 4 * p.NumModUnits());
      size_t widest = gd;
      for (auto& ch : p.gchannels) widest = std::max<size_t>(widest, ch.w);
      o.wp_scratch_stride = (p.tree.uses_wp || p.mod_local_wp) ? (tight ? 10 * (widest + 2) : 10 * (65536 + 2)) : 16;
      o.wp_scratch = take(o.wp_scratch_stride * 4 * (1 + p.NumModUnits()));
      if (e.complex) for (int c = 0; c < 3; c++) { o.plane_a[c] = take_big((size_t)p.bw * 8 * p.bh * 8 * 4); o.plane_b[c] = (size_t)-1; }
    }
    if (p.has_global_tree && p.tree_code.lz77 && (!p.gchannels.empty() || !p.modular))   // LZ77 windows of the Modular streams (4 MiB each); VarDCT: + one per LF group
      o.lz_window = take((size_t)(1 + p.NumModUnits() + (p.modular ? 0 : p.num_lf_groups)) * (4u << 20));
    if (!p.modular) {   // LZ77-coded AC streams: a window per group stream (1 MiB each, only for the frames that use them)
      bool lz_ac = p.single_section;    // (a one-section frame's AC code is only parsed once its LF streams have been decoded: always reserved, 1 MiB)
      for (auto& code : p.ac_code) lz_ac |= code.lz77;
      if (lz_ac) o.lz_ac_window = take((size_t)p.num_groups * kAcLzWindow * 4);
    }
    if (e.complex) {
      // buffers of the frame tail (PlanPostOps): float extra channels, upsampled planes, noise planes, colour-transformed planes
      // (only when the untransformed ones must survive as a reference frame), canvas (only when the frame is blended)
      any_complex_ = true;
      ComplexBufs& cb = cbufs_[i];
      const size_t ne = e.ih.extra.size();
      const size_t coded = (size_t)p.width * p.height * 4, full = (size_t)p.frame_w * p.frame_h * 4, img = (size_t)e.ih.xsize * e.ih.ysize * 4;
      for (size_t k = 0; k < ne; k++) cb.ecf[k] = take_big(coded);
      if (p.upsampling > 1) { for (int c = 0; c < 3; c++) cb.up[c] = take_big(full); for (size_t k = 0; k < ne; k++) cb.up_ec[k] = take_big(full); }
      if (p.feat.has_noise) for (int c = 0; c < 3; c++) cb.noise[c] = take_big(full);
      const bool can_ref = !p.is_last && p.frame_type != 1 && (p.duration == 0 || p.save_as_reference != 0);
      bool replace_all = p.blend.mode == 0;
      for (auto& b : p.ec_blend) if (b.mode != 0) replace_all = false;
      const bool needs_blending = p.have_crop || !replace_all;
      if (p.frame_type != 2) {
        if (can_ref && p.save_before_ct) for (int c = 0; c < 3; c++) cb.rgb[c] = take_big(full);
        if (needs_blending) { for (int c = 0; c < 3; c++) cb.canvas[c] = take_big(img); for (size_t k = 0; k < ne; k++) cb.canvas_ec[k] = take_big(img); }
      }
      for (int c = 0; c < 3; c++) { cb.pa[c] = o.plane_a[c]; cb.pb[c] = o.plane_b[c]; }
    }
  }
  work_size_ = Align(w);
  {
    // The arena is cleared when it is new.  A refilled batch of plain VarDCT frames (the streaming loop: 14 MB of LF-stage outputs per 4K frame, 3.6 GB per
    // batch of 256) only gets its status / flag / counter words zeroed: every kernel of that path writes what it reads later (valid streams) or treats what it
    // finds as it treats the content of a damaged stream.  JXL_HIP_POISON_WORK=1 (tests) fills the rest with 0xCD even when new, to prove exactly that.
    const bool fresh = DevReserve((void**)&dwork_, &work_cap_, work_size_);
    static const bool poison = getenv("JXL_HIP_POISON_WORK") != nullptr;
    bool plain = n > 0, cautious = false;     // cautious: one-group frames (the LF stage runs ahead, in this function) and output buffers inside the arena (row padding)
    for (int i = 0; i < n; i++) {
      const ImageEntry& e = *images_[i];
      if (e.plan.modular || !e.plan.gchannels.empty() || e.complex) plain = false;
      if (e.plan.single_section || (e.frame_index == 0 && !e.out.device_ptr)) cautious = true;
    }
    const size_t head = std::min(work_size_, Align(hfw_off_ + (size_t)n * 4));
    if (plain && poison) {
      HIP_CHECK(hipMemsetAsync(dwork_ + head, 0xCD, work_size_ - head, stream));
      HIP_CHECK(hipMemsetAsync(dwork_, 0, head, stream));
      for (int i = 0; i < n; i++) { const ImageEntry& e = *images_[i]; if (e.frame_index == 0 && !e.out.device_ptr) HIP_CHECK(hipMemsetAsync(dwork_ + e.off_out, 0, (e.out_size + 64) * std::max<size_t>(1, e.deliver_frames.size()), stream)); }
    } else if (plain && !cautious && !fresh) HIP_CHECK(hipMemsetAsync(dwork_, 0, head, stream));
    else HIP_CHECK(hipMemsetAsync(dwork_, 0, work_size_, stream));
  }
  big_size_ = Align(wbig);
  has_plane_b_ = need_plane_b;
  bool plain_planes = n > 0;      // plain VarDCT frames overwrite every sample of the pixel planes they read: they can take planes that hold another decode's leftovers
  for (int i = 0; i < n; i++) { const ImageEntry& e = *images_[i]; if (e.plan.modular || e.complex || e.plan.upsampling > 1) plain_planes = false; }
  if (big_owner_) {
    if (!big_owner_->dbig_ || big_owner_->big_cap_ < big_size_) throw ParseError("ShareBigArena: the owner's buffers are missing or smaller than this batch needs", false);
    dbig_ = big_owner_->dbig_;
  } else if (ext_big_ && ext_big_->p && ext_big_->cap >= big_size_ && plain_planes) {
    if (dbig_) { Pool().Give(dbig_, big_cap_, device_, /*idle=*/big_sharers_ == 0); dbig_ = nullptr; big_cap_ = 0; }
    dbig_ = ext_big_->p; big_is_ext_ = true;
  } else if (DevReserve((void**)&dbig_, &big_cap_, std::max<size_t>(big_size_, 256)) || big_sharers_ == 0) {
    // (planes other batches share are not cleared again when this object is refilled: their decodes may be using them — every sharer,
    // like a refilled owner, then starts from what the decode before left, which plain frames overwrite completely)
    HIP_CHECK(hipMemsetAsync(dbig_, 0, big_cap_, stream));
  }
  if (coef_owner_) {
    if (!coef_owner_->dcoef_ || coef_owner_->coef_cap_ < coeff_bytes_) throw ParseError("ShareCoefArena: the owner's planes are missing or smaller than this batch needs", false);
    dcoef_ = coef_owner_->dcoef_;                      // (whether they are clean is the owner's knowledge: CoefDirty())
  } else if (ext_coef_ && ext_coef_->p && ext_coef_->cap >= coeff_bytes_ && any_vardct_) {
    if (dcoef_) { Pool().Give(dcoef_, coef_cap_, device_, /*idle=*/true); dcoef_ = nullptr; coef_cap_ = 0; }
    dcoef_ = ext_coef_->p; coef_is_ext_ = true;        // (clean or not: ext_coef_->dirty / clean_extent, kept by the decodes that used the set before)
  } else {
    const bool fresh = DevReserve((void**)&dcoef_, &coef_cap_, std::max<size_t>(coeff_bytes_, 256));
    if (fresh) coef_clean_extent_ = 0;                 // (a new allocation: nothing of it is known to be zero)
    if (fresh || coeff_bytes_ != coef_laid_out_) coef_dirty_ = true;   // new planes, or another layout of them: the first decode clears them in its own stream
  }
  coef_laid_out_ = coeff_bytes_;
  mark("layout+reserve");
  DevReserve((void**)&dframes_, &frames_cap_, sizeof(FrameDev) * std::max(n, 1));

  // ---- single-section VarDCT frames: HfGlobal starts where the device-decoded LfGroup ends.  Pre-run the LF stage
  // for those frames now, read the end position back and parse HfGlobal on the host.
  frames_host_.assign(n, FrameDev());
  auto fill_frame = [&](int i, const uint8_t* cbase) {
    ImageEntry& e = *images_[i];
    const FramePlan& p = e.plan;
    const ConstOffsets& c = co[i];
    const WorkOffsets& o = wo[i];
    FrameDev& f = frames_host_[i];
    memset(&f, 0, sizeof(f));
    f.width = p.width; f.height = p.height; f.bw = p.bw; f.bh = p.bh; f.xgroups = p.xgroups; f.ygroups = p.ygroups; f.num_groups = p.num_groups;
    f.xlfgroups = p.xlfgroups; f.num_lf_groups = p.num_lf_groups; f.cw = (p.bw + 7) / 8; f.ch = (p.bh + 7) / 8;
    f.group_dim = p.group_dim; f.is_modular = p.modular; f.plane_stride = p.bw * 8; f.plane_rows = p.bh * 8;
    f.cs = cbase + c.cs; f.cs_size = e.cs.size;
    f.sec_off = (const uint64_t*)(cbase + c.sec_off); f.sec_size = (const uint64_t*)(cbase + c.sec_size);
    f.single_section = p.single_section;
    f.lf_start_bitpos = p.global_data_bitpos;  // VarDCT without extra channels: LfGroup follows LfGlobal directly
    f.stream_end_bitpos = (uint64_t*)(dwork_ + o.end_bitpos);
    if (p.has_global_tree) {
      f.tree = (const TreeNode*)(cbase + c.tree);
      f.tree_nodes = (uint32_t)p.tree.nodes.size();
      f.mod_code = ViewCode(p.tree_code, cbase, c.mod_ctx, c.mod_cfg, c.mod_alias, c.mod_pc, c.mod_po, c.mod_ps);
      f.mod_cfg_uniform = p.tree_code.cfg.empty() ? 0xFFFFFFFFu : p.tree_code.cfg[0];
      for (uint32_t v : p.tree_code.cfg) if (v != p.tree_code.cfg[0]) f.mod_cfg_uniform = 0xFFFFFFFFu;
    }
    f.uses_wp = p.tree.uses_wp; f.gwp = p.gwp;
    f.mod_unit_passes = p.ModUnitPasses(); f.mod_pass = p.mod_pass;
    for (int k = 0; k < 11; k++) { f.pass_min_shift[k] = p.pass_min_shift[k]; f.pass_max_shift[k] = p.pass_max_shift[k]; }
    f.tree_max_prop = (uint32_t)p.tree.max_prop;
    for (int k = 0; k < 3; k++) { f.hs[k] = p.hs[k]; f.vs[k] = p.vs[k]; }
    f.subsampled = p.subsampled;
    if (!c.local.empty()) {
      // descriptor table of the frame's units (0 = global stream), zero = "uses the frame's tree"
      ModLocalDev* table = &local_host_[local_first_[i]];
      f.mod_local = dlocal_ + local_first_[i];
      for (size_t k = 0; k < c.local.size(); k++) {
        const ConstOffsets::Local& l = c.local[k];
        const FramePlan::LocalStream& ls = p.local_streams[k];
        ModLocalDev& d = table[l.unit];
        d.tree = (const TreeNode*)(cbase + l.tree); d.tree_nodes = (uint32_t)ls.tree.nodes.size(); d.uses_wp = ls.tree.uses_wp; d.max_prop = (uint32_t)ls.tree.max_prop;
        d.data_bitpos = ls.data_bitpos;
        d.code = ViewCode(ls.code, cbase, l.ctx, l.cfg, l.alias, l.pc, l.po, l.ps);
      }
    }
    f.bcm = (const BlockCtxDev*)(cbase + c.bcm);
    f.status = (uint32_t*)(dwork_ + status_off_) + i;
    f.frame_flags = (uint32_t*)(dwork_ + flags_off) + i;
    f.hf_written = (uint32_t*)(dwork_ + hfw_off_) + i;
    f.out = (uint8_t*)(e.out.device_ptr ? e.out.device_ptr : dwork_ + e.off_out);
    f.out_stride = e.out_stride; f.out_channels = e.out.num_channels; f.out_type = e.out.type; f.out_big_endian = e.out.big_endian; f.out_int_mul = IntMul(e.out);
    f.out_orient = e.out.keep_orientation ? 1 : e.ih.orientation;
    f.upsampling = p.upsampling; f.img_w = e.ih.xsize; f.img_h = e.ih.ysize;
    if (p.upsampling > 1) { f.up_weights = (const float*)(cbase + c.up_weights); for (int k = 0; k < 4; k++) f.up_plane[k] = (float*)(dbig_ + o.up_plane[k]); }
    f.is_gray = e.ih.color_space == 1;
    f.post_mode = e.complex ? 1 : 0;
    f.lz_window = o.lz_window == (size_t)-1 ? nullptr : (uint32_t*)(dwork_ + o.lz_window);
    f.lz_ac_window = o.lz_ac_window == (size_t)-1 ? nullptr : (uint32_t*)(dwork_ + o.lz_ac_window);
    f.lz_lf_base = 1 + p.NumModUnits();
    f.wp_scratch = (int32_t*)(dwork_ + o.wp_scratch); f.wp_scratch_stride = o.wp_scratch_stride;
    if (!p.modular) {
      const float inv_gs = 65536.0f / (float)p.global_scale;
      const float inv_quant_lf = inv_gs / (float)p.quant_lf;
      for (int k = 0; k < 3; k++) f.lf_fac[k] = p.m_lf[k] * inv_quant_lf;
      f.cfl_lf_x = p.base_x + (float)p.ytox_lf * (1.0f / (float)p.color_factor);
      f.cfl_lf_b = p.base_b + (float)p.ytob_lf * (1.0f / (float)p.color_factor);
      f.inv_global_scale = inv_gs;
      f.x_dm = std::pow(0.8f, (float)p.x_qm_scale - 2.0f); f.b_dm = std::pow(0.8f, (float)p.b_qm_scale - 2.0f);
      for (int k = 0; k < 4; k++) f.quant_bias[k] = e.ih.quant_bias[k];
      f.color_scale = 1.0f / (float)p.color_factor; f.base_x = p.base_x; f.base_b = p.base_b;
      f.skip_lf_smoothing = (p.flags & 128) != 0 || p.use_lf_frame;     // (dec_frame.cc FinalizeDC: no adaptive smoothing of an LF frame's samples)
      f.use_lf_frame = p.use_lf_frame ? 1 : 0;
      f.gab = p.lf.gab; f.epf_iters = p.lf.epf_iters;
      for (int k = 0; k < 3; k++) {
        const float w1 = p.lf.gab_w[2 * k], w2 = p.lf.gab_w[2 * k + 1];
        const float div = 1.0f + 4.0f * (w1 + w2);
        f.gab_w[3 * k] = 1.0f / div; f.gab_w[3 * k + 1] = w1 / div; f.gab_w[3 * k + 2] = w2 / div;
      }
      for (int k = 0; k < 8; k++) f.epf_sharp_lut[k] = p.lf.sharp_lut[k];
      for (int k = 0; k < 3; k++) f.epf_channel_scale[k] = p.lf.channel_scale[k];
      f.epf_quant_mul = p.lf.quant_mul; f.epf_quant_scale = (float)p.global_scale / 65536.0f;
      const float scales[3] = {p.lf.pass0_sigma_scale, 1.0f, p.lf.pass2_sigma_scale};
      for (int k = 0; k < 3; k++) { f.epf_sm[k] = scales[k] * 1.65f; f.epf_bsm[k] = f.epf_sm[k] * p.lf.border_sad_mul; }
      FillColor(e.ih, p.do_ycbcr, f);
      for (int k = 0; k < 3; k++) {
        f.lfq[k] = (int32_t*)(dwork_ + o.lfq[k]); f.lf[k] = (float*)(dwork_ + o.lf[k]); f.lf_tmp[k] = (float*)(dwork_ + o.lf_tmp[k]);
        f.llf[k] = (float*)(dwork_ + o.llf[k]); f.coeff[k] = (int32_t*)(dcoef_ + o.coeff[k]);
        f.plane_a[k] = (float*)(dbig_ + o.plane_a[k]); f.plane_b[k] = o.plane_b[k] == (size_t)-1 ? nullptr : (float*)(dbig_ + o.plane_b[k]);
      }
      f.blk_info = (uint32_t*)(dwork_ + o.blk_info); f.coef_off = (uint32_t*)(dwork_ + o.coef_off);
      f.vb_list = (uint2*)(dwork_ + o.vb_list); f.vb_count = (uint32_t*)(dwork_ + o.vb_count);
      f.place_rec = (uint4*)(dwork_ + o.place_rec); f.place_cnt = (uint32_t*)(dwork_ + o.place_cnt); f.band_start = (uint32_t*)(dwork_ + o.band_start);
      f.ytox = (int8_t*)(dwork_ + o.ytox); f.ytob = (int8_t*)(dwork_ + o.ytob);
      f.inv_sigma = (float*)(dwork_ + o.inv_sigma);
      f.lf_scratch = (int32_t*)(dwork_ + o.lf_scratch); f.lf_scratch_stride = o.lf_scratch_stride;
      // HfGlobal-derived fields (present once parsed)
      if (!p.ac_code.empty()) {
        f.num_passes = cfg.max_passes > 0 ? std::min<uint32_t>(p.num_passes, (uint32_t)cfg.max_passes) : p.num_passes;     // (a progressive flush shows the first passes only; the later ones' sections stay unread)
        f.passes = dpasses_ + pass_first_[i];
        for (uint32_t ps = 0; ps < p.num_passes; ps++) {
          const ConstOffsets::Pass& cp = c.pass[ps];
          PassDev& pd = passes_host_[pass_first_[i] + ps];
          pd.code = ViewCode(p.ac_code[ps], cbase, cp.ac_ctx, cp.ac_cfg, cp.ac_alias, cp.ac_pc, cp.ac_po, cp.ac_ps);
          for (int k = 0; k < 39; k++) pd.orders[k] = (const uint16_t*)(cbase + cp.orders[k]);
          pd.shift = ps + 1 < p.num_passes ? p.pass_shift[ps] : 0; pd.pad = 0;
        }
        f.ac_code = passes_host_[pass_first_[i]].code;
        for (int k = 0; k < 39; k++) f.orders[k] = passes_host_[pass_first_[i]].orders[k];
        for (int k = 0; k < 17 * 3; k++) f.qtable[k] = c.has_qtable[k / 3] ? (const float*)(cbase + c.qtable[k]) : nullptr;
        f.num_hf_presets = p.num_hf_presets;
        uint32_t bits = 0; while ((1u << bits) < p.num_hf_presets) bits++;
        f.preset_bits = bits;
        f.hf_start_bitpos = p.end_bitpos;
      }
    } else {
      f.hf_start_bitpos = 0;
      f.mod_wp_scratch = f.wp_scratch; f.mod_wp_stride = f.wp_scratch_stride;
    }
    if (!p.gchannels.empty()) {
      f.mod_nchan = (uint32_t)p.gchannels.size(); f.mod_nb_meta = p.nb_meta_channels;
      f.mod_chan = (const ModChanDev*)(cbase + c.mod_chan); f.mod_base = dwork_;
      f.mod_global_decodable = p.global_decodable; f.mod_global_bitpos = p.global_data_bitpos;
      f.mod_group_scratch = (int32_t*)(dwork_ + o.mod_scratch); f.mod_group_scratch_stride = o.mod_scratch_stride;
      f.mod_bits = e.ih.depth.bits;
      if (!p.modular) {
        f.hf_end_bitpos = (uint64_t*)(dwork_ + o.hf_end);
        f.mod_wp_scratch = (int32_t*)(dwork_ + o.mod_wp); f.mod_wp_stride = o.mod_wp_stride;
        f.alpha_plane = vardct_alpha_[i].has ? (const int32_t*)(dwork_ + vardct_alpha_[i].off) : nullptr;
        f.alpha_factor = vardct_alpha_[i].factor;
      }
    }
  };

  vec<int> single;
  for (int i = 0; i < n; i++) if (images_[i]->plan.single_section && !images_[i]->plan.modular) {
    single.push_back(i);
    images_[i]->plan.ac_code.clear();      // (a batch that is prepared again: HfGlobal is parsed afresh below, fill_frame must not look for its tables yet)
  }
  if (!single.empty()) {
    // temporary upload of what exists so far
    uint8_t* tmpc = nullptr;
    HIP_CHECK(hipMalloc((void**)&tmpc, Align(hconst_.size())));
    HIP_CHECK(hipMemcpyAsync(tmpc, hconst_.data(), hconst_.size(), hipMemcpyHostToDevice, stream));
    vec<FrameDev> tmpf;
    for (int i : single) { fill_frame(i, tmpc); tmpf.push_back(frames_host_[i]); }
    HIP_CHECK(hipMemcpyAsync(dframes_, tmpf.data(), sizeof(FrameDev) * tmpf.size(), hipMemcpyHostToDevice, stream));
    if (any_modchan_) LaunchModularGlobal(dframes_, (int)tmpf.size(), cfg, stream_v);   // extra channels of a one-group frame precede the LfGroup
    LaunchLfDecode(dframes_, (int)tmpf.size(), 1, cfg, stream_v);
    HIP_CHECK(hipStreamSynchronize(stream));
    for (size_t k = 0; k < single.size(); k++) {
      const int i = single[k];
      uint32_t status = 0;
      uint64_t endpos[2] = {0, 0};
      HIP_CHECK(hipMemcpy(&status, tmpf[k].status, 4, hipMemcpyDeviceToHost));
      HIP_CHECK(hipMemcpy(endpos, tmpf[k].stream_end_bitpos, 16, hipMemcpyDeviceToHost));
      if (status) { (void)hipFree(tmpc); throw ParseError(status & kErrUnsupported ? "unsupported: stream feature (LF stage)" : "corrupt LF group stream", (status & kErrUnsupported) != 0); }
      ParseHfGlobal(images_[i]->cs, images_[i]->ih, endpos[0], &images_[i]->plan);
    }
    HIP_CHECK(hipMemsetAsync(dwork_ + status_off_, 0, (size_t)n * 4, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    (void)hipFree(tmpc);
  }
  // ---- HfGlobal tables (now available for every VarDCT frame)
  for (int i = 0; i < n; i++) {
    ImageEntry& e = *images_[i];
    FramePlan& p = e.plan;
    ConstOffsets& c = co[i];
    if (p.modular) continue;
    c.pass.assign(p.num_passes, ConstOffsets::Pass());
    for (uint32_t ps = 0; ps < p.num_passes; ps++) {
      ConstOffsets::Pass& cp = c.pass[ps];
      PutCode(arena, p.ac_code[ps], &cp.ac_ctx, &cp.ac_cfg, &cp.ac_alias, &cp.ac_pc, &cp.ac_po, &cp.ac_ps);
      for (int b = 0; b < 13; b++) for (int ch = 0; ch < 3; ch++) {
        const auto& cu = p.custom_order[(size_t)ps * 39 + b * 3 + ch];
        cp.orders[b * 3 + ch] = cu.empty() ? natural_off[b] : arena.Put(cu.data(), cu.size() * 2);
      }
    }
    if (p.num_passes > 1) any_multipass_ = true;
    for (int k = 0; k < 17; k++) {
      int hit = -1;
      for (size_t q = 0; q < qcache.size(); q++) if (qcache[q].kind == k && spec_equal(*qcache[q].spec, p.qspec[k])) { hit = (int)q; break; }
      if (hit < 0) {
        QCache qc; qc.spec = &p.qspec[k]; qc.kind = k;
        for (int ch = 0; ch < 3; ch++) {
          // the DCT128/256 tables are large (up to 64K weights per channel): computed once per process and spec
          struct Big { QuantTableSpec spec; int kind, ch; std::vector<float> t; };   // process-lifetime cache: plain allocator
          static std::mutex mu;
          static std::vector<Big> big;
          vec<float> local;
          const vec<float>* t = &local;
          if (k >= 13) {
            std::lock_guard<std::mutex> lock(mu);
            const Big* found = nullptr;
            for (const Big& b : big) if (b.kind == k && b.ch == ch && spec_equal(b.spec, p.qspec[k])) { found = &b; break; }
            if (!found) {
              if (big.size() >= 48) big.clear();
              big.push_back(Big{p.qspec[k], k, ch, {}});
              { vec<float> tmp; ComputeQuantTable(p.qspec[k], k, ch, &tmp); big.back().t.assign(tmp.begin(), tmp.end()); }
              found = &big.back();
            }
            local.assign(found->t.begin(), found->t.end());
          } else ComputeQuantTable(p.qspec[k], k, ch, &local);
          qc.off[ch] = arena.Put(t->data(), t->size() * 4);
        }
        qcache.push_back(qc); hit = (int)qcache.size() - 1;
      }
      for (int ch = 0; ch < 3; ch++) c.qtable[k * 3 + ch] = qcache[hit].off[ch];
      c.has_qtable[k] = true;
    }
  }
  mark("hf_tables");
  {  // LDS right-sizing for the decode kernels
    auto code_bytes = [](const HostCode& c, bool ctx) { if (c.use_prefix || c.lz77) return 16;   /* prefix codes / LZ77 streams are read through the tables in global memory: nothing to size the LDS for */
      return (int)(((c.num_clusters * 4 + 15) & ~15u) + (ctx ? ((c.num_ctx + 15) & ~15u) : 0) + ((size_t)c.num_clusters << c.log_alpha) * 8); };
    auto code_bytes_compact = [](const HostCode& c) { if (c.use_prefix || c.lz77) return 16;
      return (int)(((c.num_clusters * 4 + 15) & ~15u) + ((c.num_ctx + 15) & ~15u) + (((((size_t)c.num_clusters << c.log_alpha) * 6) + 15) & ~(size_t)15)); };
    cfg.ac_code_bytes_compact = 16;
    cfg.max_tree_nodes = 1; cfg.mod_code_bytes = 16; cfg.ac_code_bytes = 16; cfg.any_wp = 0; cfg.any_local_trees = 0; cfg.any_subsampled = 0; cfg.any_prefix_ac = 0;
    for (int i = 0; i < n; i++) {
      const FramePlan& p = images_[i]->plan;
      if (p.has_global_tree) { cfg.max_tree_nodes = std::max<int>(cfg.max_tree_nodes, (int)p.tree.nodes.size()); cfg.mod_code_bytes = std::max(cfg.mod_code_bytes, code_bytes(p.tree_code, false)); cfg.any_wp |= p.tree.uses_wp ? 1 : 0; }
      for (auto& ls : p.local_streams) {
        cfg.max_tree_nodes = std::max<int>(cfg.max_tree_nodes, (int)ls.tree.nodes.size()); cfg.mod_code_bytes = std::max(cfg.mod_code_bytes, code_bytes(ls.code, false)); cfg.any_wp |= ls.tree.uses_wp ? 1 : 0;
        if (ls.unit != 0) cfg.any_local_trees = 1;
      }
      if (!p.modular) for (auto& code : p.ac_code) { cfg.ac_code_bytes = std::max(cfg.ac_code_bytes, code_bytes(code, true)); cfg.ac_code_bytes_compact = std::max(cfg.ac_code_bytes_compact, code_bytes_compact(code)); }
      if (p.subsampled) cfg.any_subsampled = 1;
      if (!p.modular) for (auto& code : p.ac_code) if (code.use_prefix || code.lz77) cfg.any_prefix_ac = 1;
    }
    if (getenv("JXL_HIP_DEBUG_LDS")) fprintf(stderr, "[jxl-hip] LDS sizing: tree nodes %d, modular code %d B, AC code %d B (compact %d B), BlockCtxDev %zu B\n", cfg.max_tree_nodes, cfg.mod_code_bytes, cfg.ac_code_bytes, cfg.ac_code_bytes_compact, sizeof(BlockCtxDev));
  }
  {  // per-pass table descriptors (device array next to the frame descriptors)
    pass_first_.assign(n, 0);
    size_t total = 0;
    for (int i = 0; i < n; i++) { pass_first_[i] = total; total += images_[i]->plan.modular ? 0 : images_[i]->plan.num_passes; }
    passes_host_.assign(std::max<size_t>(total, 1), PassDev());
    DevReserve((void**)&dpasses_, &passes_cap_, sizeof(PassDev) * passes_host_.size());
  }
  {  // descriptors of the sub-streams with their own tree / code: one table per frame that has any
    local_first_.assign(n, 0);
    size_t total = 0;
    for (int i = 0; i < n; i++) {
      const FramePlan& p = images_[i]->plan;
      local_first_[i] = total;
      if (!p.local_streams.empty()) total += 1 + (size_t)p.NumModUnits();
    }
    local_host_.assign(std::max<size_t>(total, 1), ModLocalDev());
    for (auto& d : local_host_) memset(&d, 0, sizeof(d));
    DevReserve((void**)&dlocal_, &local_cap_, sizeof(ModLocalDev) * local_host_.size());
  }
  if (any_complex_) {
    vec<size_t> upw(n, 0);
    for (int i = 0; i < n; i++) upw[i] = co[i].up_weights;
    PlanPostOps(hconst_, upw);
  }
  mark("misc");
  lf_simt_ = LfSimtPlan();
  // ---- varblock placement units: every 32-row band of every LF group of every VarDCT frame, the tallest / widest first (the lanes of a
  // wavefront then carry bands of similar length)
  size_t place_units_off = 0;
  {
    vec<uint2> units;
    vec<uint32_t> ucost;
    for (int i = 0; i < n; i++) {
      const FramePlan& p = images_[i]->plan;
      if (p.modular) continue;
      for (uint32_t g = 0; g < p.num_lf_groups; g++) {
        const uint32_t gx = g % p.xlfgroups, gy = g / p.xlfgroups;
        const uint32_t gbw = std::min<uint32_t>(256, p.bw - gx * 256), gbh = std::min<uint32_t>(256, p.bh - gy * 256);
        for (uint32_t band = 0; band * 32 < gbh; band++) { units.push_back(make_uint2((uint32_t)i, g | (band << 16))); ucost.push_back(std::min<uint32_t>(32, gbh - band * 32) * gbw); }
      }
    }
    vec<uint32_t> order(units.size());
    for (size_t k = 0; k < order.size(); k++) order[k] = (uint32_t)k;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return ucost[a] > ucost[b]; });
    vec<uint2> sorted;
    for (uint32_t k : order) sorted.push_back(units[k]);
    place_units_off = arena.Put(sorted.data(), sorted.size() * sizeof(uint2));
    lf_simt_.num_units = (uint32_t)sorted.size();
  }
  // ---- SIMT LF decode plan (cfg.lane_stride_lf < 64): streams of eligible frames, spread over lanes of about equal work
  vec<uint8_t> simt_frame(n, 0);
  size_t simt_streams_off = 0, simt_lanes_off = 0, simt_luts_off = 0;
  if (any_vardct_ && cfg.lane_stride_lf < 64) {
    vec<LfSimtStream> streams;
    vec<uint64_t> cost;
    vec<uint8_t> luts;
    std::map<std::string, uint32_t> lut_of;       // table content -> offset in the blob (frames of one encoder share most tables)
    bool general_streams = false;                 // some eligible stream needs the instantiation with two-property tables / the rarer properties and predictors
    bool wp_streams = false;                      // some eligible stream keeps weighted-predictor state: the batch takes that instantiation of the kernel
    for (int i = 0; i < n; i++) {
      const FramePlan& p = images_[i]->plan;
      if (p.modular || !p.has_global_tree || p.tree_code.use_prefix || p.tree_code.lz77 || p.tree_code.log_alpha > 8 || p.tree_code.num_clusters > 256 || p.use_lf_frame) continue;
      {   // extra-channel sub-channels squeezed by >= 3 sit in the middle of the LfGroup sections: only the one-wavefront-per-stream kernel decodes those
        bool lf_modular = false;
        for (size_t c = p.global_decodable; c < p.gchannels.size(); c++) lf_modular |= p.gchannels[c].w && p.gchannels[c].h && std::min(p.gchannels[c].hshift, p.gchannels[c].vshift) >= 3;
        if (lf_modular) continue;
      }
      vec<LfSimtStream> mine;
      bool ok = true;
      for (uint32_t g = 0; g < p.num_lf_groups && ok; g++) {
        LfSimtStream st;
        memset(&st, 0, sizeof(st));
        st.frame = (uint32_t)i; st.group = g;
        for (int c = 0; c < 7 && ok; c++) {
          LfChanClass cl;
          ok = ClassifyLfChannel(p.tree, p.tree_code, c < 3 ? c : c - 3, c < 3 ? 1 + g : 1 + 2 * p.num_lf_groups + g, &cl);
          if (ok && cl.uses_wp && c == 5) ok = false;      // (the block-info rows are up to 65 536 wide: the per-lane weighted-predictor rows hold 256)
          if (!ok) break;
          auto intern = [&](const uint8_t* bytes, size_t size) {      // table content -> offset in the blob (frames of one encoder share most tables)
            const std::string key((const char*)bytes, size);
            auto it = lut_of.find(key);
            if (it == lut_of.end()) { it = lut_of.emplace(key, (uint32_t)luts.size()).first; luts.insert(luts.end(), bytes, bytes + size); }
            return it->second;
          };
          auto word = [&](const LfRowClass& rc) { return rc.kind | (rc.pred << 2) | (rc.sel_a << 5) | (rc.sel_b << 7) | (cl.uses_wp ? kLfSimtWpLive : 0u) | (rc.cluster << 16); };
          if (cl.rows.size() == 1) {
            st.chan[c].lut_off = cl.rows[0].kind ? intern(cl.rows[0].lut.data(), cl.rows[0].lut.size()) : 0;
            st.chan[c].info = word(cl.rows[0]);
          } else {   // the class depends on the row: [512 bytes: row -> class][classes x {table offset, class word}]
            vec<uint8_t> blob(512 + 8 * cl.rows.size());
            memcpy(blob.data(), cl.row_to_class, 512);
            for (size_t k = 0; k < cl.rows.size(); k++) {
              const uint32_t e[2] = {cl.rows[k].kind ? intern(cl.rows[k].lut.data(), cl.rows[k].lut.size()) : 0u, word(cl.rows[k])};
              memcpy(blob.data() + 512 + 8 * k, e, 8);
            }
            st.chan[c].lut_off = intern(blob.data(), blob.size());
            st.chan[c].info = kLfSimtRows | (cl.uses_wp ? kLfSimtWpLive : 0u);
          }
          if (cl.uses_wp) wp_streams = true;
          for (const LfRowClass& rc : cl.rows)      // beyond what the lean instantiation handles: one cluster per row or a table over W + N - NW; zero / W / clamped gradient
            if (rc.kind == 2 || (rc.kind == 1 && rc.sel_a != 0) || !(rc.pred == 0 || rc.pred == 1 || rc.pred == 3)) general_streams = true;
        }
        if (ok) mine.push_back(st);
      }
      if (!ok) continue;
      simt_frame[i] = 1;
      for (auto& st : mine) {
        const uint32_t gx = st.group % p.xlfgroups, gy = st.group / p.xlfgroups;
        const uint64_t gbw = std::min<uint32_t>(256, p.bw - gx * 256), gbh = std::min<uint32_t>(256, p.bh - gy * 256);
        streams.push_back(st); cost.push_back(gbw * gbh * 5 + 2048);
      }
    }
    if (!streams.empty()) {
      // longest-processing-time-first; the number of lanes grows from the lower bound (total work / longest stream) until no lane carries
      // noticeably more than the longest stream: a lane more costs nothing, a lane with two long streams doubles the stage's latency
      uint64_t total = 0, longest = 0;
      for (uint64_t cst : cost) { total += cst; longest = std::max(longest, cst); }
      vec<uint32_t> order(streams.size());
      for (size_t k = 0; k < order.size(); k++) order[k] = (uint32_t)k;
      std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
      vec<vec<uint32_t>> lane_streams;
      for (size_t nlanes = std::max<size_t>(1, std::min<size_t>(streams.size(), (size_t)((total + longest - 1) / longest)));; nlanes = std::min(streams.size(), nlanes + std::max<size_t>(1, nlanes / 64))) {
        lane_streams.assign(nlanes, vec<uint32_t>());
        std::priority_queue<std::pair<uint64_t, uint32_t>, std::vector<std::pair<uint64_t, uint32_t>>, std::greater<std::pair<uint64_t, uint32_t>>> pq;
        for (size_t l = 0; l < nlanes; l++) pq.push({0, (uint32_t)l});
        uint64_t makespan = 0;
        for (uint32_t k : order) { auto top = pq.top(); pq.pop(); lane_streams[top.second].push_back(k); makespan = std::max(makespan, top.first + cost[k]); pq.push({top.first + cost[k], top.second}); }
        if (makespan <= longest + longest / 16 || nlanes >= streams.size()) break;
      }
      // lanes of the same frames next to each other (they read the same tables)
      std::stable_sort(lane_streams.begin(), lane_streams.end(), [&](const vec<uint32_t>& a, const vec<uint32_t>& b) { return streams[a[0]].frame < streams[b[0]].frame; });
      vec<LfSimtStream> flat;
      vec<LfSimtLane> lanes;
      for (auto& ls : lane_streams) { lanes.push_back(LfSimtLane{(uint32_t)flat.size(), (uint32_t)ls.size()}); for (uint32_t k : ls) flat.push_back(streams[k]); }
      simt_streams_off = arena.Put(flat.data(), flat.size() * sizeof(LfSimtStream));
      simt_lanes_off = arena.Put(lanes.data(), lanes.size() * sizeof(LfSimtLane));
      simt_luts_off = arena.Put(luts.data(), luts.size());
      lf_simt_.num_lanes = (uint32_t)lanes.size();
      lf_simt_.lanes_per_wave = (uint32_t)std::max(1, 64 / std::max(1, cfg.lane_stride_lf));
      lf_simt_.any_legacy = 0; lf_simt_.any_wp = wp_streams ? 1 : 0; lf_simt_.any_general = general_streams || wp_streams ? 1 : 0;

      for (int i = 0; i < n; i++) if (!images_[i]->plan.modular && !simt_frame[i]) lf_simt_.any_legacy = 1;
    }
  }
  mark("simt_plan");
  const_size_ = Align(hconst_.size());
  DevReserve((void**)&dconst_, &const_cap_, const_size_);
  HIP_CHECK(hipMemcpyAsync(dconst_, hconst_.data(), hconst_.size(), hipMemcpyHostToDevice, stream));
  for (int i = 0; i < n; i++) { fill_frame(i, dconst_); frames_host_[i].lf_simt = lf_simt_.num_lanes ? simt_frame[i] : 0; }
  lf_simt_.units = (const uint2*)(dconst_ + place_units_off);
  if (lf_simt_.num_lanes) {
    lf_simt_.streams = (const LfSimtStream*)(dconst_ + simt_streams_off); lf_simt_.lanes = (const LfSimtLane*)(dconst_ + simt_lanes_off); lf_simt_.luts = dconst_ + simt_luts_off;
  }
  mark("fill_frames");
  {
    // the descriptor arrays travel through the pinned staging buffer, too (a copy from pageable memory is staged by the runtime and holds the calling thread —
    // and others — up for as long as the device is busy); appended behind the constant arena's content, which has been copied out of it above
    const size_t o_frames = arena.Put(frames_host_.data(), sizeof(FrameDev) * (size_t)n);
    const size_t o_passes = arena.Put(passes_host_.data(), sizeof(PassDev) * passes_host_.size());
    const size_t o_local = arena.Put(local_host_.data(), sizeof(ModLocalDev) * local_host_.size());
    HIP_CHECK(hipMemcpyAsync(dframes_, hconst_.data() + o_frames, sizeof(FrameDev) * (size_t)n, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(dpasses_, hconst_.data() + o_passes, sizeof(PassDev) * passes_host_.size(), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(dlocal_, hconst_.data() + o_local, sizeof(ModLocalDev) * local_host_.size(), hipMemcpyHostToDevice, stream));
  }
  mark("h2d_enqueue");
  // (a pipelined caller enqueues the LF stage on the same stream right behind the upload and never waits for it on the host: the pinned staging buffer is not touched
  // again before the object's next Prepare, which comes after this decode has left the GPU)
  if (wait_upload) HIP_CHECK(hipStreamSynchronize(stream));
  mark("upload_wait");
  if (time_phases) fprintf(stderr, "[jxl-hip] Prepare of %d frames (%.1f MB of tables and streams, %.1f MB work arena), ms:%s\n", n, hconst_.size() / 1e6, work_size_ / 1e6, t_report.c_str());
  // ---- LF frames the units refer to: a batch of their own, every LF frame a one-frame image of its own size that ends in its XYB planes (PlanPostOps)
  {
    vec<int> users;
    for (int i = 0; i < n; i++) if (images_[i]->lf_source) users.push_back(i);
    if (users.empty()) lf_batch_.reset();
    else {
      if (!lf_batch_) lf_batch_.reset(new Batch(device_)); else lf_batch_->Reset();
      lf_batch_->lf_targets_.assign(users.size(), LfTarget());
      for (size_t k = 0; k < users.size(); k++) {
        const ImageEntry& user = *images_[users[k]];
        const ImageEntry& src = *user.lf_source;
        std::shared_ptr<ImageShared> sh(new ImageShared());
        sh->cs = src.cs; sh->ih = src.ih;
        sh->ih.xsize = src.plan.width; sh->ih.ysize = src.plan.height; sh->ih.orientation = 1; sh->ih.intrinsic_x = sh->ih.intrinsic_y = 0; sh->ih.have_preview = false;
        std::unique_ptr<ImageEntry> e(new ImageEntry(sh));
        e->plan = src.plan; e->frame_bitpos = src.frame_bitpos; e->frame_index = 0; e->complex = true;
        e->visible_frame_index = src.visible_frame_index; e->nonvisible_frame_index = src.nonvisible_frame_index;
        e->lf_source = src.lf_source;                      // (an LF frame may itself sit on the LF frame of the next level)
        ParsedImage pi; pi.complex = true; pi.units.push_back(std::move(e));
        lf_batch_->Append(std::move(pi));
        LfTarget& t = lf_batch_->lf_targets_[k];
        for (int c = 0; c < 3; c++) t.dst[c] = frames_host_[users[k]].lf[c];
        t.pitch = user.plan.bw; t.w = user.plan.bw; t.h = user.plan.bh;
      }
      lf_batch_->cfg.lane_stride_lf = 64;
      lf_batch_->Prepare(stream_v, wait_upload);
    }
  }
  // (a frame cut off inside its AC groups, FramePlan::partial: the HF kernels leave out the group streams that are not completely there; nothing to do when none is)
  for (int i = 0; i < n; i++) if (images_[i]->plan.partial && images_[i]->plan.partial_ac_sections == 0) cfg.skip_hf = 1;
  cfg.any_multipass = any_multipass_ ? 1 : 0;
  if (any_multipass_) cfg.lane_stride_hf = 1;   // progressive frames: only the SIMT HF kernel walks the passes
  if (cfg.any_subsampled) cfg.lane_stride_hf = 1;   // so do chroma-subsampled frames (per-channel block grids)
  prepared_ = true;
}

// JXL_HIP_DEBUG_SYNC=1: name every stage on stderr and wait for it (finding the kernel behind a device fault)
static void DebugSync(const char* what, void* stream) {
  static const bool on = getenv("JXL_HIP_DEBUG_SYNC") != nullptr;
  if (!on) return;
  fprintf(stderr, "[jxl-hip] %s ...", what); fflush(stderr);
  const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
  fprintf(stderr, " %s (launch status: %s)\n", hipGetErrorString(e), hipGetErrorString(hipPeekAtLastError())); fflush(stderr);
}

void Batch::Run(void* stream_v) { RunPart(stream_v, 0, false); }

// Everything Modular after the entropy decode of the global stream: the LfGroup / PassGroup sub-streams, then the
// host-planned inverse transforms (and, for Modular frames, the write stage).
void Batch::EnqueueModularTail(void* stream_v) {
  const int n = (int)images_.size();
  int max_units = 1;
  for (auto& im : images_) max_units = std::max<int>(max_units, (int)im->plan.NumModUnits());
  static const bool dbg = getenv("JXL_HIP_DEBUG_SYNC") != nullptr;
  auto check = [&](const char* what, int kind) { if (!dbg) return; const hipError_t e = hipGetLastError(); if (e != hipSuccess) fprintf(stderr, "[jxl-hip] launch of %s (%d) rejected: %s\n", what, kind, hipGetErrorString(e)); };
  check("(before the Modular tail)", -1);
  LaunchModularGroups(dframes_, n, max_units, cfg, stream_v);
  check("ModularGroupFastKernel", max_units);
  // The inverse transforms of different images are independent chains of latency-bound kernels (an inverse Squeeze step is one thread per row / column walking a recurrence:
  // 32 workgroups for the last step of an 8192 x 8192 channel, ~40 launches per channel).  Images whose chains have the same shape — the frames of a job usually do — go through
  // them together: step k of up to kSqueezeBatch images is ONE launch (blockIdx.y = image).  Side streams per image gave the same overlap and made every later latency-bound
  // decode of the process 1.3-1.6x slower (profiles/r06_notes.md section 12); the owner's streams (SetTailStreams) are only used when JXL_HIP_MOD_TAIL_STREAMS asks for them.
  auto P = [&](size_t off) { return (int32_t*)(dwork_ + off); };
  auto run_op = [&](int i, const ModOp& op, void* st) {
    check("Modular tail op before", (int)op.kind);
    switch (op.kind) {
      case ModOp::kRct: LaunchModRct(P(op.in[0]), P(op.in[1]), P(op.in[2]), op.n, op.param, st); break;
      case ModOp::kPalette: {
        int32_t* outs[4] = {nullptr, nullptr, nullptr, nullptr};
        for (uint32_t c = 0; c < op.num_c; c++) outs[c] = P(op.out[c]);
        if (op.nb_deltas == 0 && op.predictor == 0) LaunchModPalette(P(op.in[0]), outs, op.param, op.num_c, op.bits, op.n, st);
        else LaunchModPaletteDelta(P(op.in[0]), outs, op.param, op.num_c, op.bits, op.nb_deltas, op.predictor, op.aw, op.ah, images_[i]->plan.gwp,
                                   op.predictor == 6 ? P(op.wp_scratch) : nullptr, op.wp_stride, st);
        break;
      }
      case ModOp::kSqueeze: LaunchModInvSqueeze(P(op.in[0]), P(op.in[1]), P(op.out[0]), op.param, op.aw, op.ah, op.rw, op.rh, st); break;
      case ModOp::kOutput: {
        ModOutputArgs a;
        memset(&a, 0, sizeof(a));
        a.ncolor = op.num_c;
        for (uint32_t c = 0; c < a.ncolor; c++) a.color[c] = P(op.in[c]);
        a.color_factor = op.color_factor; a.float_bits = op.float_bits; a.float_exp_bits = op.float_exp_bits;
        a.alpha = op.has_alpha ? P(op.in[3]) : nullptr; a.alpha_factor = op.alpha_factor;
        LaunchModOutput(dframes_, i, a, images_[i]->plan.width, images_[i]->plan.height, st);
        break;
      }
    }
  };
  static const int tail_streams = getenv("JXL_HIP_MOD_TAIL_STREAMS") ? atoi(getenv("JXL_HIP_MOD_TAIL_STREAMS")) : 0;
  static const bool no_batch = getenv("JXL_HIP_NO_SQUEEZE_BATCH") != nullptr;      // A/B: every image's chain on its own
  vec<int> with_ops;
  for (int i = 0; i < n; i++) if (!mod_ops_[i].empty()) with_ops.push_back(i);
  if (tail_streams >= 2 && with_ops.size() >= 2 && tail_streams_) {
    // (experiments) one chain per side stream of the owner, forked from and joined to stream_v
    vec<void*> side;
    for (int k = 0; k < std::min<int>((int)with_ops.size(), tail_streams); k++) { void* st = tail_streams_(k); if (!st) break; side.push_back(st); }
    if (side.size() >= 2) {
      while (mod_join_events_.size() < side.size()) { hipEvent_t ev; HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); mod_join_events_.push_back(ev); }
      if (!mod_fork_event_) { hipEvent_t ev; HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); mod_fork_event_ = ev; }
      HIP_CHECK(hipEventRecord((hipEvent_t)mod_fork_event_, (hipStream_t)stream_v));
      for (void* st : side) HIP_CHECK(hipStreamWaitEvent((hipStream_t)st, (hipEvent_t)mod_fork_event_, 0));
      for (size_t k = 0; k < with_ops.size(); k++) for (const ModOp& op : mod_ops_[with_ops[k]]) run_op(with_ops[k], op, side[k % side.size()]);
      for (size_t k = 0; k < side.size(); k++) {
        HIP_CHECK(hipEventRecord((hipEvent_t)mod_join_events_[k], (hipStream_t)side[k]));
        HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream_v, (hipEvent_t)mod_join_events_[k], 0));
      }
      return;
    }
  }
  // groups of images whose chains have the same shape (kind and geometry of every step)
  auto same_shape = [&](int a, int b) {
    const vec<ModOp>& x = mod_ops_[a]; const vec<ModOp>& y = mod_ops_[b];
    if (x.size() != y.size()) return false;
    for (size_t k = 0; k < x.size(); k++)
      if (x[k].kind != y[k].kind || x[k].param != y[k].param || x[k].aw != y[k].aw || x[k].ah != y[k].ah || x[k].rw != y[k].rw || x[k].rh != y[k].rh) return false;
    return true;
  };
  vec<char> done((size_t)n, 0);
  for (int lead : with_ops) {
    if (done[(size_t)lead]) continue;
    vec<int> grp;
    for (int i : with_ops) if (!done[(size_t)i] && (int)grp.size() < kSqueezeBatch && (i == lead || (!no_batch && same_shape(lead, i)))) { grp.push_back(i); done[(size_t)i] = 1; }
    const vec<ModOp>& ops = mod_ops_[lead];
    for (size_t k = 0; k < ops.size(); k++) {
      if (ops[k].kind == ModOp::kSqueeze && grp.size() > 1) {
        check("Modular tail op before", (int)ops[k].kind);
        SqueezeBatch b;
        memset(&b, 0, sizeof(b));
        for (size_t g = 0; g < grp.size(); g++) { const ModOp& o = mod_ops_[grp[g]][k]; b.avg[g] = P(o.in[0]); b.res[g] = P(o.in[1]); b.out[g] = P(o.out[0]); }
        LaunchModInvSqueezeBatch(b, (int)grp.size(), ops[k].param, ops[k].aw, ops[k].ah, ops[k].rw, ops[k].rh, stream_v);
      } else {
        for (int i : grp) run_op(i, mod_ops_[i][k], stream_v);
      }
    }
  }
}

// Builds the list of kernels that undo the global transforms of Modular image i (transform.cc, reverse order) and feed
// the write stage, tracking the channel list on the host exactly as MetaApply built it; `take` reserves work-arena bytes.
void Batch::PlanModularUndo(int i, const std::function<size_t(size_t)>& take) {
  const ImageEntry& e = *images_[i];
  const FramePlan& p = e.plan;
  struct HC { size_t off; uint32_t w, h; };
  vec<HC> list;
  for (size_t k = 0; k < p.gchannels.size(); k++) list.push_back({mod_plane_offsets_[i][k], p.gchannels[k].w, p.gchannels[k].h});
  vec<ModOp>& ops = mod_ops_[i];
  ops.clear();
  for (int t = (int)p.gtransforms.size() - 1; t >= 0; t--) {
    const TransformDesc& td = p.gtransforms[t];
    if (td.id == 0) {
      if (td.begin_c + 3 > list.size()) throw ParseError("rct range", false);
      const HC &a = list[td.begin_c], &b = list[td.begin_c + 1], &c = list[td.begin_c + 2];
      if (a.w != b.w || a.w != c.w || a.h != b.h || a.h != c.h) throw ParseError("rct over channels of different size", false);
      ModOp op; op.kind = ModOp::kRct; op.in[0] = a.off; op.in[1] = b.off; op.in[2] = c.off; op.n = (size_t)a.w * a.h; op.param = td.rct_type;
      ops.push_back(op);
    } else if (td.id == 1) {
      // list[0] = palette, list[begin_c + 1] = index channel -> num_c channels in its place
      if (td.begin_c + 1 >= list.size()) throw ParseError("palette range", false);
      const HC idx = list[td.begin_c + 1];
      ModOp op; op.kind = ModOp::kPalette; op.in[0] = list[0].off; op.num_c = td.num_c; op.param = td.nb_colors;
      op.bits = std::min<uint32_t>(e.ih.depth.bits, 24); op.n = (size_t)idx.w * idx.h;
      if (td.num_c > 4) throw ParseError("unsupported: palette with more than 4 channels", true);
      op.out[0] = idx.off;
      for (uint32_t c = 1; c < td.num_c; c++) op.out[c] = take(op.n * 4 + 64);
      op.nb_deltas = td.nb_deltas; op.predictor = td.predictor; op.aw = idx.w; op.ah = idx.h;
      if (td.predictor == 6) { op.wp_stride = 10 * (idx.w + 2); op.wp_scratch = take((size_t)op.wp_stride * 4 * td.num_c); }   // weighted-predictor state per output channel
      ops.push_back(op);
      vec<HC> nl;
      for (size_t k = 1; k < list.size(); k++) {
        if (k - 1 == td.begin_c) { for (uint32_t c = 0; c < td.num_c; c++) nl.push_back({op.out[c], idx.w, idx.h}); }
        else nl.push_back(list[k]);
      }
      list.swap(nl);
    } else {
      for (int q = (int)td.squeeze.size() - 1; q >= 0; q--) {
        const SqueezeStep& sq = td.squeeze[q];
        const uint32_t endc = sq.begin_c + sq.num_c - 1;
        const uint32_t offset = sq.in_place ? endc + 1 : (uint32_t)(list.size() + sq.begin_c - endc - 1);
        if (offset + sq.num_c > list.size() || endc >= offset) throw ParseError("squeeze range", false);
        for (uint32_t c = sq.begin_c; c <= endc; c++) {
          const HC avg = list[c], res = list[offset + c - sq.begin_c];
          ModOp op; op.kind = ModOp::kSqueeze; op.param = sq.horizontal; op.in[0] = avg.off; op.in[1] = res.off;
          op.aw = avg.w; op.ah = avg.h; op.rw = res.w; op.rh = res.h;
          HC out;
          if (sq.horizontal) { if (avg.h != res.h) throw ParseError("squeeze dims", false); out = {0, avg.w + res.w, avg.h}; }
          else { if (avg.w != res.w) throw ParseError("squeeze dims", false); out = {0, avg.w, avg.h + res.h}; }
          out.off = take((size_t)out.w * out.h * 4 + 64);
          op.out[0] = out.off;
          ops.push_back(op);
          list[c] = out;
        }
        list.erase(list.begin() + offset, list.begin() + offset + sq.num_c);
      }
    }
  }
  ModOp op; op.kind = ModOp::kOutput; op.num_c = p.nb_color_channels;   // (0 for VarDCT frames: only extra channels are Modular)
  if (list.size() < op.num_c + e.ih.extra.size()) throw ParseError("modular channel list", false);
  for (size_t k = 0; k < op.num_c + e.ih.extra.size(); k++)
    if (list[k].w != p.width || list[k].h != p.height) throw ParseError("unsupported: channel of a different size than the image (dim_shift)", true);
  for (uint32_t c = 0; c < op.num_c; c++) op.in[c] = list[c].off;
  op.color_factor = e.ih.depth.is_float ? 1.0f : 1.0f / (float)((1u << e.ih.depth.bits) - 1);
  op.float_bits = e.ih.depth.is_float ? e.ih.depth.bits : 0; op.float_exp_bits = e.ih.depth.exp_bits;
  const int alpha_pick = images_[pub_[e.pub_index].first_unit]->out.alpha_from_extra;
  for (size_t k = 0; k < e.ih.extra.size(); k++) if (alpha_pick >= 0 ? (int)k == alpha_pick : e.ih.extra[k].type == 0) {
    op.has_alpha = true; op.in[3] = list[op.num_c + k].off;
    op.alpha_factor = e.ih.extra[k].depth.is_float ? 1.0f : 1.0f / (float)((1u << e.ih.extra[k].depth.bits) - 1);   // (float alpha: converted in the frame tail)
    break;
  }
  if (e.complex) {
    // frames of complex images end in float planes: the tail converts these integer planes itself (PlanPostOps)
    ComplexBufs& cb = cbufs_[i];
    cb.nb_color_int = op.num_c;
    for (uint32_t c = 0; c < op.num_c; c++) cb.color_int[c] = list[c].off;
    cb.ec_int.clear();
    for (size_t k = 0; k < e.ih.extra.size(); k++) cb.ec_int.push_back(list[op.num_c + k].off);
    return;
  }
  if (p.modular) ops.push_back(op);
  else vardct_alpha_[i] = VarDctAlpha{op.has_alpha, op.in[3], op.alpha_factor};   // the VarDCT write stage reads the plane itself
}

// ---- frame tail of complex images -------------------------------------------------------------------------------------------------
// Walks the frames of every complex image in codestream order and records, as closures over device addresses, the kernels
// that turn each frame's planes into the image: dec_cache.cc PreparePipeline's stage order (patches, splines, upsampling,
// noise | save as reference before the colour transform | XYB / YCbCr -> output colour space | blending onto the canvas |
// save as reference | write).  Reference slots are tracked here (frame_header.cc save_as_reference / CanBeReferenced).
void Batch::PlanPostOps(HostStage& hconst, const vec<size_t>& up_weights_off) {
  Arena arena(hconst);
  struct Slot { bool valid = false, before_ct = false; size_t p[3] = {0, 0, 0}; uint32_t stride = 0; size_t ec[4] = {0, 0, 0, 0}; uint32_t ec_stride = 0; uint32_t w = 0, h = 0; };
  auto B = [this](size_t off) { return (float*)(dbig_ + off); };
  for (const PubImage& pi : pub_) {
    if (!pi.complex) continue;
    Slot slots[4];
    const ImageEntry& first = *images_[pi.first_unit];
    const ImageHeader& ih = first.ih;
    const uint32_t ne = (uint32_t)ih.extra.size();
    uint32_t premul_mask = 0;
    for (uint32_t k = 0; k < ne; k++) if (ih.extra[k].alpha_associated) premul_mask |= 1u << k;
    for (int u = pi.first_unit; u < pi.first_unit + pi.num_units; u++) {
      const ImageEntry& e = *images_[u];
      const FramePlan& p = e.plan;
      const ComplexBufs& cb = cbufs_[u];
      const uint32_t cw = p.width, ch = p.height, fw = p.frame_w, fh = p.frame_h;
      // current planes of the frame: after the restoration filters (VarDCT) / the int -> float conversion (Modular)
      const uint32_t nstages = p.modular ? 0 : (p.lf.gab ? 1 : 0) + (p.lf.epf_iters >= 3 ? 3 : p.lf.epf_iters);
      size_t cur[3]; uint32_t cur_stride = p.bw * 8;
      for (int c = 0; c < 3; c++) cur[c] = (nstages & 1) ? cb.pb[c] : cb.pa[c];
      size_t cur_ec[4] = {0, 0, 0, 0}; uint32_t cur_ec_stride = cw;
      if (p.subsampled) {
        // the subsampled channels sit in the top-left corner of their planes: bring them to full resolution (plane b)
        for (int c = 0; c < 3; c++) {
          if (!p.hs[c] && !p.vs[c]) continue;
          if (cb.pb[c] == (size_t)-1) throw ParseError("chroma upsampling needs the second plane set", false);
          const size_t src = cur[c], dst = cb.pb[c];
          const uint32_t ccw = (cw + (1u << p.hs[c]) - 1) >> p.hs[c], cch = (ch + (1u << p.vs[c]) - 1) >> p.vs[c], chs = p.hs[c], cvs = p.vs[c];
          post_ops_.push_back([=](void* st) { LaunchChromaUpsample(B(src), cur_stride, B(dst), cur_stride, ccw, cch, chs, cvs, cw, ch, st); });
          cur[c] = dst;
        }
      }
      if (p.modular) {
        const uint32_t bits = ih.depth.bits;
        if (ih.xyb_encoded) {
          if (cb.nb_color_int != 3) throw ParseError("XYB Modular frame without three colour channels", false);
          const size_t cy = cb.color_int[0], cx = cb.color_int[1], cbb = cb.color_int[2];
          const float fac[3] = {p.m_lf[0], p.m_lf[1], p.m_lf[2]};
          post_ops_.push_back([=](void* st) {
            float* dst[3] = {B(cur[0]), B(cur[1]), B(cur[2])};
            LaunchXybModToFloat((const int32_t*)(dwork_ + cy), (const int32_t*)(dwork_ + cx), (const int32_t*)(dwork_ + cbb), cw, dst, cur_stride, cw, ch, fac, st);
          });
        } else {
          const bool fl = ih.depth.is_float;
          const float factor = fl ? 1.0f : (float)(1.0 / (double)((1u << bits) - 1));
          const uint32_t fbits = fl ? ih.depth.bits : 0, febits = ih.depth.exp_bits;
          for (int c = 0; c < 3; c++) {
            const size_t src = cb.color_int[cb.nb_color_int == 1 ? 0 : c], dst = cur[c];
            post_ops_.push_back([=](void* st) { LaunchIntToFloat((const int32_t*)(dwork_ + src), cw, B(dst), cur_stride, cw, ch, factor, st, fbits, febits); });
          }
        }
      }
      for (uint32_t k = 0; k < ne; k++) {
        if (k >= cb.ec_int.size()) throw ParseError("missing extra channel", false);
        const size_t src = cb.ec_int[k], dst = cb.ecf[k];
        const bool efl = ih.extra[k].depth.is_float;
        const float factor = efl ? 1.0f : 1.0f / (float)((1u << ih.extra[k].depth.bits) - 1);
        const uint32_t ebits = efl ? ih.extra[k].depth.bits : 0, eexp = ih.extra[k].depth.exp_bits;
        post_ops_.push_back([=](void* st) { LaunchIntToFloat((const int32_t*)(dwork_ + src), cw, B(dst), cw, cw, ch, factor, st, ebits, eexp); });
        cur_ec[k] = dst;
      }
      // ---- patches
      if (p.flags & 2) {
        vec<PatchEntryDev> entries;
        for (const PatchRefH& pr : p.feat.patches) {
          const Slot& sl = slots[pr.ref];
          if (!sl.valid) throw ParseError("patch refers to an empty reference slot", false);
          if (!sl.before_ct) throw ParseError("patch refers to a frame saved after the colour transform", false);
          if (pr.xsize == 0 || pr.ysize == 0) throw ParseError("empty patch", false);
          if ((uint64_t)pr.x0 + pr.xsize > sl.w || (uint64_t)pr.y0 + pr.ysize > sl.h) throw ParseError("patch exceeds its reference frame", false);
          for (const PatchPosH& pp : pr.pos) {
            if (pp.x + pr.xsize > cw || pp.y + pr.ysize > ch) throw ParseError("patch exceeds the frame", false);
            PatchEntryDev en;
            memset(&en, 0, sizeof(en));
            for (int c = 0; c < 3; c++) en.src[c] = B(sl.p[c]) + (size_t)pr.y0 * sl.stride + pr.x0;
            for (uint32_t k = 0; k < ne; k++) en.esrc[k] = B(sl.ec[k]) + (size_t)pr.y0 * sl.ec_stride + pr.x0;
            en.src_stride = sl.stride; en.esrc_stride = sl.ec_stride;
            en.x = (int32_t)pp.x; en.y = (int32_t)pp.y; en.xs = pr.xsize; en.ys = pr.ysize;
            for (uint32_t k = 0; k < 1 + ne; k++) en.mode[k] = pp.blend[k].mode | (pp.blend[k].alpha_channel << 8) | (pp.blend[k].clamp << 16);
            entries.push_back(en);
          }
        }
        if (!entries.empty()) {
          // per 32x32 tile: the placements touching it, in dictionary order
          const uint32_t tx = (cw + 31) / 32, ty = (ch + 31) / 32;
          vec<vec<uint32_t>> lists((size_t)tx * ty);
          for (uint32_t k = 0; k < entries.size(); k++) {
            const PatchEntryDev& en = entries[k];
            // (sizes are >= 1 and the placement lies inside the frame — checked above; clamped all the same: the lists must not be overrun)
            const uint32_t y1 = std::min(ty - 1, ((uint32_t)en.y + std::max(en.ys, 1u) - 1) / 32), x1 = std::min(tx - 1, ((uint32_t)en.x + std::max(en.xs, 1u) - 1) / 32);
            for (uint32_t yy = (uint32_t)en.y / 32; yy <= y1; yy++)
              for (uint32_t xx = (uint32_t)en.x / 32; xx <= x1; xx++) lists[(size_t)yy * tx + xx].push_back(k);
          }
          vec<uint32_t> start(lists.size() + 1, 0), flat;
          for (size_t t = 0; t < lists.size(); t++) { start[t + 1] = start[t] + (uint32_t)lists[t].size(); flat.insert(flat.end(), lists[t].begin(), lists[t].end()); }
          if (flat.empty()) flat.push_back(0);
          const size_t o_e = arena.Put(entries.data(), entries.size() * sizeof(PatchEntryDev)), o_s = arena.Put(start.data(), start.size() * 4), o_l = arena.Put(flat.data(), flat.size() * 4);
          PatchFrameArgs pa;
          memset(&pa, 0, sizeof(pa));
          for (int c = 0; c < 3; c++) pa.p[c] = B(cur[c]);
          for (uint32_t k = 0; k < ne; k++) pa.ec[k] = B(cur_ec[k]);
          pa.stride = cur_stride; pa.ec_stride = cur_ec_stride; pa.w = cw; pa.h = ch; pa.num_extra = ne; pa.premul_mask = premul_mask;
          post_ops_.push_back([=](void* st) { LaunchPatches(pa, (const PatchEntryDev*)(dconst_ + o_e), (const uint32_t*)(dconst_ + o_s), (const uint32_t*)(dconst_ + o_l), st); });
        }
      }
      // ---- splines (segments from the host, host_features.cc)
      if (p.flags & 16) {
        SplineDrawList dl;
        BuildSplineDrawList(p.feat, p.base_x, p.base_b, ch, &dl);
        if (!dl.segments.empty() && !dl.indices.empty()) {
          const size_t o_g = arena.Put(dl.segments.data(), dl.segments.size() * sizeof(SplineSegmentDev)), o_r = arena.Put(dl.row_start.data(), dl.row_start.size() * 4),
                       o_i = arena.Put(dl.indices.data(), dl.indices.size() * 4);
          post_ops_.push_back([=](void* st) {
            float* pl[3] = {B(cur[0]), B(cur[1]), B(cur[2])};
            LaunchSplines(pl, cur_stride, cw, ch, (const SplineSegmentDev*)(dconst_ + o_g), (const uint32_t*)(dconst_ + o_r), (const uint32_t*)(dconst_ + o_i), st);
          });
        }
      }
      // ---- upsampling to the frame size
      if (p.upsampling > 1) {
        const size_t o_w = up_weights_off[u];
        const uint32_t up = p.upsampling;
        for (int c = 0; c < 3; c++) {
          const size_t src = cur[c], dst = cb.up[c]; const uint32_t ss = cur_stride;
          post_ops_.push_back([=](void* st) { LaunchUpsamplePlane(B(src), ss, cw, ch, B(dst), fw, fw, fh, up, (const float*)(dconst_ + o_w), st); });
          cur[c] = dst;
        }
        for (uint32_t k = 0; k < ne; k++) {
          const size_t src = cur_ec[k], dst = cb.up_ec[k];
          post_ops_.push_back([=](void* st) { LaunchUpsamplePlane(B(src), cw, cw, ch, B(dst), fw, fw, fh, up, (const float*)(dconst_ + o_w), st); });
          cur_ec[k] = dst;
        }
        cur_stride = fw; cur_ec_stride = fw;
      }
      // ---- noise
      if (p.feat.has_noise) {
        NoiseArgs na;
        memset(&na, 0, sizeof(na));
        for (int c = 0; c < 3; c++) { na.p[c] = B(cur[c]); na.noise[c] = B(cb.noise[c]); }
        na.stride = cur_stride; na.w = fw; na.h = fh; na.noise_stride = fw; na.group_dim = p.group_dim;
        na.visible_frame_index = e.visible_frame_index; na.nonvisible_frame_index = e.nonvisible_frame_index;
        for (int k = 0; k < 8; k++) na.lut[k] = p.feat.noise_lut[k];
        na.ytox = p.base_x; na.ytob = p.base_b;
        post_ops_.push_back([=](void* st) { LaunchNoise(na, st); });
      }
      if (p.frame_type == 1) {
        // an LF frame (this batch is another batch's lf_batch_): its planes, before any colour transform, are the LF image of the frame that refers to it
        const int pub = e.pub_index;
        const size_t s0 = cur[0], s1 = cur[1], s2 = cur[2];
        post_ops_.push_back([=](void* st) {
          if ((size_t)pub >= lf_targets_.size() || !lf_targets_[pub].dst[0]) return;
          const LfTarget& t = lf_targets_[pub];
          const size_t src[3] = {s0, s1, s2};
          for (int c = 0; c < 3; c++)
            HIP_CHECK(hipMemcpy2DAsync(t.dst[c], (size_t)t.pitch * 4, B(src[c]), (size_t)cur_stride * 4, (size_t)t.w * 4, t.h, hipMemcpyDeviceToDevice, (hipStream_t)st));
        });
        continue;
      }
      const bool can_ref = !p.is_last && p.frame_type != 1 && (p.duration == 0 || p.save_as_reference != 0);
      if (can_ref && p.save_before_ct) {
        Slot& sl = slots[p.save_as_reference];
        sl.valid = true; sl.before_ct = true; sl.stride = cur_stride; sl.ec_stride = cur_ec_stride; sl.w = fw; sl.h = fh;
        for (int c = 0; c < 3; c++) sl.p[c] = cur[c];
        for (uint32_t k = 0; k < ne; k++) sl.ec[k] = cur_ec[k];
      }
      if (p.frame_type == 2) continue;    // reference-only frames are not displayed
      // ---- colour transform into the output space
      bool has_spot = false;
      for (uint32_t k = 0; k < ne; k++) if (ih.extra[k].type == 2) has_spot = true;
      bool blends = p.have_crop || p.blend.mode != 0;
      for (auto& b : p.ec_blend) if (b.mode != 0) blends = true;
      const bool spot_after_linear = has_spot && first.out.render_spotcolors && p.is_last && !blends;
      ColorArgs deferred_tf;
      memset(&deferred_tf, 0, sizeof(deferred_tf));
      bool have_deferred_tf = false;
      {
        ColorArgs ca;
        memset(&ca, 0, sizeof(ca));
        ca.w = fw; ca.h = fh; ca.src_stride = cur_stride;
        ca.mode = ih.xyb_encoded ? 0 : p.do_ycbcr ? 1 : 2;
        const bool separate = cb.rgb[0] != (size_t)-1;
        if (ca.mode != 2 || separate) {
          for (int c = 0; c < 3; c++) { ca.src[c] = B(cur[c]); ca.dst[c] = separate ? B(cb.rgb[c]) : B(cur[c]); }
          ca.dst_stride = separate ? fw : cur_stride;
          FrameDev fd;
          FillColor(ih, p.do_ycbcr, fd);
          for (int k = 0; k < 9; k++) ca.opsin_inv[k] = fd.opsin_inv[k];
          for (int k = 0; k < 3; k++) { ca.neg_bias[k] = fd.neg_bias[k]; ca.neg_bias_cbrt[k] = fd.neg_bias_cbrt[k]; }
          ca.tf_kind = fd.color_mode == 0 ? 0 : fd.color_mode == 1 ? 1 : fd.color_mode == 4 ? 2 : fd.color_mode == 5 ? 3 : fd.color_mode == 6 ? 4 : 5;
          for (int k = 0; k < 5; k++) ca.hdr_par[k] = fd.hdr_par[k];
          ca.inverse_gamma = fd.inverse_gamma;
          // spot colours are mixed in linear light when the frame goes straight to the output (dec_cache.cc PreparePipeline: XYB stage,
          // [from-linear + blending], spot stage, from-linear): the transfer function then follows the spot stage
          if (spot_after_linear && ca.mode == 0) { deferred_tf = ca; deferred_tf.mode = 3; ca.tf_kind = 1; have_deferred_tf = true; }
          post_ops_.push_back([=](void* st) { LaunchColor(ca, st); });
          if (separate) { for (int c = 0; c < 3; c++) cur[c] = cb.rgb[c]; cur_stride = fw; }
        }
      }
      if (first.out.only_frame >= 0 && u == pi.first_unit + first.out.only_frame) {
        // ---- non-coalesced output (JxlDecoderSetCoalescing(false)): this frame's own pixels after the colour transform, frame-sized, not
        // blended; the frames before it have been composed as usual (it may draw patches from them), the ones after it are not needed
        WriteArgs wa;
        memset(&wa, 0, sizeof(wa));
        for (int c = 0; c < 3; c++) wa.p[c] = B(cur[c]);
        wa.stride = cur_stride;
        for (uint32_t k = 0; k < ne; k++) if (first.out.alpha_from_extra >= 0 ? (int)k == first.out.alpha_from_extra : ih.extra[k].type == 0) {
          wa.alpha = B(cur_ec[k]); wa.alpha_stride = cur_ec_stride;
          wa.unpremul = first.out.unpremul_alpha && ih.extra[k].alpha_associated && (first.out.num_channels == 2 || first.out.num_channels == 4);
          break;
        }
        wa.img_w = fw; wa.img_h = fh;
        wa.out = (uint8_t*)(first.out.device_ptr ? first.out.device_ptr : dwork_ + first.off_out);
        wa.out_stride = first.out_stride; wa.out_channels = first.out.num_channels; wa.out_type = first.out.type; wa.out_big_endian = first.out.big_endian; wa.out_int_mul = IntMul(first.out);
        wa.out_orient = first.out.keep_orientation ? 1 : ih.orientation; wa.is_gray = ih.color_space == 1;
        if (have_deferred_tf) {          // (the transfer function had been put off for a spot-colour stage this output does not run)
          ColorArgs ta = deferred_tf;
          for (int c = 0; c < 3; c++) { ta.src[c] = B(cur[c]); ta.dst[c] = B(cur[c]); }
          ta.src_stride = ta.dst_stride = cur_stride; ta.w = fw; ta.h = fh;
          post_ops_.push_back([=](void* st) { LaunchColor(ta, st); });
        }
        post_ops_.push_back([=](void* st) { LaunchWrite(wa, st); });
        break;
      }
      // ---- blending onto the canvas
      bool replace_all = p.blend.mode == 0;
      for (auto& b : p.ec_blend) if (b.mode != 0) replace_all = false;
      const bool needs_blending = p.have_crop || !replace_all;
      size_t canvas[3], canvas_ec[4] = {0, 0, 0, 0}; uint32_t canvas_stride, canvas_ec_stride;
      if (!needs_blending) {
        if (fw != ih.xsize || fh != ih.ysize) throw ParseError("frame size differs from the image size without a crop", false);
        for (int c = 0; c < 3; c++) canvas[c] = cur[c];
        for (uint32_t k = 0; k < ne; k++) canvas_ec[k] = cur_ec[k];
        canvas_stride = cur_stride; canvas_ec_stride = cur_ec_stride;
      } else {
        auto source = [&](uint32_t idx) -> const Slot* {
          const Slot& sl = slots[idx];
          if (!sl.valid) return nullptr;
          if (sl.before_ct) throw ParseError("blending source was saved before the colour transform", false);
          if (sl.w != ih.xsize || sl.h != ih.ysize) throw ParseError("blending source has the wrong size", false);
          return &sl;
        };
        BlendArgs ba;
        memset(&ba, 0, sizeof(ba));
        for (int c = 0; c < 3; c++) { ba.fg[c] = B(cur[c]); ba.canvas[c] = B(cb.canvas[c]); }
        for (uint32_t k = 0; k < ne; k++) { ba.fg_ec[k] = B(cur_ec[k]); ba.canvas_ec[k] = B(cb.canvas_ec[k]); }
        ba.fg_stride = cur_stride; ba.fg_ec_stride = cur_ec_stride; ba.fw = fw; ba.fh = fh; ba.x0 = p.x0; ba.y0 = p.y0;
        ba.canvas_stride = ih.xsize; ba.canvas_ec_stride = ih.xsize; ba.img_w = ih.xsize; ba.img_h = ih.ysize; ba.num_extra = ne; ba.premul_mask = premul_mask;
        const Slot* bg = source(p.blend.source);
        if (bg) { for (int c = 0; c < 3; c++) ba.bg[c] = B(bg->p[c]); ba.bg_stride = bg->stride; }
        ba.mode[0] = p.blend.mode | (p.blend.alpha_channel << 8) | ((uint32_t)p.blend.clamp << 16);
        if (p.blend.mode == 2 || p.blend.mode == 3) {
          if (ne == 0) throw ParseError("alpha blending without extra channels", false);
          if (bg) { ba.bg_alpha = B(bg->ec[p.blend.alpha_channel]); ba.bg_alpha_stride = bg->ec_stride; }
        }
        for (uint32_t k = 0; k < ne; k++) {
          const BlendInfoH& bi = p.ec_blend[k];
          ba.mode[1 + k] = bi.mode | (bi.alpha_channel << 8) | ((uint32_t)bi.clamp << 16);
          const Slot* eb = source(bi.source);
          if (eb) { ba.bg_ec[k] = B(eb->ec[k]); ba.bg_ec_alpha[k] = B(eb->ec[bi.alpha_channel]); ba.bg_ec_stride[k] = eb->ec_stride; }
        }
        post_ops_.push_back([=](void* st) { LaunchBlend(ba, st); });
        for (int c = 0; c < 3; c++) canvas[c] = cb.canvas[c];
        for (uint32_t k = 0; k < ne; k++) canvas_ec[k] = cb.canvas_ec[k];
        canvas_stride = ih.xsize; canvas_ec_stride = ih.xsize;
      }
      if (can_ref && !p.save_before_ct) {
        Slot& sl = slots[p.save_as_reference];
        sl.valid = true; sl.before_ct = false; sl.stride = canvas_stride; sl.ec_stride = canvas_ec_stride; sl.w = ih.xsize; sl.h = ih.ysize;
        for (int c = 0; c < 3; c++) sl.p[c] = canvas[c];
        for (uint32_t k = 0; k < ne; k++) sl.ec[k] = canvas_ec[k];
      }
      // coalescing: the composite of the last frame is what the caller receives — or, for a frame of an animation, the canvas after that frame
      // ... or, decoded once for all of them (SetOutputAllFrames), after every frame of the list, each canvas to its own slot of the output area
      int slot = -1;
      for (size_t k = 0; k < first.deliver_frames.size(); k++) if (first.deliver_frames[k] == u - pi.first_unit) slot = (int)k;
      const bool all_frames = !first.deliver_frames.empty();
      const bool deliver = all_frames ? slot >= 0 : (first.out.upto_frame >= 0 ? u == pi.first_unit + first.out.upto_frame : p.is_last);
      if (!deliver) continue;
      if (all_frames) {   // (delivery must leave the canvas as it is: the frames behind blend onto it)
        bool spot = false;
        for (uint32_t k = 0; k < ne; k++) spot |= ih.extra[k].type == 2 && first.out.render_spotcolors;
        if (spot || have_deferred_tf) throw ParseError("unsupported: all frames of an animation in one decode with spot colours to render", true);
      }
      // ---- spot colours (stage_spot.cc): colour = mix * spot + (1 - mix) * colour with mix = solidity * channel, channel by channel
      if (first.out.render_spotcolors) {
        for (uint32_t k = 0; k < ne; k++) {
          if (ih.extra[k].type != 2) continue;
          SpotArgs sa;
          for (int c = 0; c < 3; c++) { sa.p[c] = B(canvas[c]); sa.color[c] = ih.extra[k].spot[c]; }
          sa.stride = canvas_stride; sa.spot = B(canvas_ec[k]); sa.spot_stride = canvas_ec_stride; sa.scale = ih.extra[k].spot[3]; sa.w = ih.xsize; sa.h = ih.ysize;
          post_ops_.push_back([=](void* st) { LaunchSpot(sa, st); });
        }
      }
      if (have_deferred_tf) {
        ColorArgs ta = deferred_tf;
        for (int c = 0; c < 3; c++) { ta.src[c] = B(canvas[c]); ta.dst[c] = B(canvas[c]); }
        ta.src_stride = ta.dst_stride = canvas_stride; ta.w = ih.xsize; ta.h = ih.ysize;
        post_ops_.push_back([=](void* st) { LaunchColor(ta, st); });
      }
      // ---- write stage
      WriteArgs wa;
      memset(&wa, 0, sizeof(wa));
      for (int c = 0; c < 3; c++) wa.p[c] = B(canvas[c]);
      wa.stride = canvas_stride;
      for (uint32_t k = 0; k < ne; k++) if (first.out.alpha_from_extra >= 0 ? (int)k == first.out.alpha_from_extra : ih.extra[k].type == 0) {
        wa.alpha = B(canvas_ec[k]); wa.alpha_stride = canvas_ec_stride;
        wa.unpremul = first.out.unpremul_alpha && ih.extra[k].alpha_associated && (first.out.num_channels == 2 || first.out.num_channels == 4);
        break;
      }
      wa.img_w = ih.xsize; wa.img_h = ih.ysize;
      wa.out = (uint8_t*)(first.out.device_ptr ? first.out.device_ptr : dwork_ + first.off_out) + (all_frames ? (size_t)slot * (first.out_size + 64) : 0);
      wa.out_stride = first.out_stride; wa.out_channels = first.out.num_channels; wa.out_type = first.out.type; wa.out_big_endian = first.out.big_endian; wa.out_int_mul = IntMul(first.out);
      wa.out_orient = first.out.keep_orientation ? 1 : ih.orientation; wa.is_gray = ih.color_space == 1;
      post_ops_.push_back([=](void* st) { LaunchWrite(wa, st); });
      if (all_frames && slot + 1 < (int)first.deliver_frames.size()) continue;
      break;                      // (frames behind the delivered one: nothing of theirs is needed)
    }
  }
}

void Batch::EnqueuePostOps(void* stream) { for (auto& op : post_ops_) op(stream); }

// The HF stage only writes non-zero coefficients into zeroed planes.  The IDCT kernels zero what they consumed (kernels.hip,
// IdctTileKernel pass 0), so a batch that is decoded again and again never clears its planes as a whole; only a decode whose
// tail did not run over them (first decode, JPEG reconstruction, a failed stream) leaves them dirty.
void Batch::ClearCoefficientsBeforeHf(void* stream_v) {
  // The planes are known to be zero over [0, clean extent) of the owner: hipMalloc'd memory is not cleared, and a decode only puts zeros
  // back inside its own layout — a sharer (or a refill) whose layout reaches further than anything cleared so far needs the dense clear too.
  size_t& extent = coef_is_ext_ ? ext_coef_->clean_extent : coef_owner_ ? coef_owner_->coef_clean_extent_ : coef_clean_extent_;
  if (CoefDirty() || coeff_bytes_ > extent) {
    const size_t cap = coef_is_ext_ ? ext_coef_->cap : coef_owner_ ? coef_owner_->coef_cap_ : coef_cap_;
    const size_t bytes = std::min(cap, std::max(extent, coeff_bytes_));
    HIP_CHECK(hipMemsetAsync(dcoef_, 0, bytes, (hipStream_t)stream_v));
    extent = bytes;
  }
  CoefDirty() = true;
}
void Batch::ClearCoefficientsAfterDecode(void*) { CoefDirty() = false; }

void Batch::CheckFilterBuffers() const {
  if ((fplan_.any_unfused || cfg.force_unfused_filters) && !has_plane_b_)
    throw ParseError("force_unfused_filters must be set before Prepare (the second pixel plane is only allocated when a frame needs it)", false);
}

void Batch::RunTimed(void* stream_v) { RunPart(stream_v, 0, true); }

// part 0 = whole decode, 1 = front (coefficient clear + LF decode + LF post-processing), 2 = rest (HF decode, IDCT,
// filters, output).  Front and rest of one decode may be enqueued on different streams (ordered by the caller with
// events) so that the latency-bound LF stage of the next batch overlaps the bandwidth stages of the current one.
// A launch the runtime refuses (too much LDS, an empty grid ...) does not fail at the call site: it leaves an error code that the next runtime call of the thread
// reports — possibly somebody else's.  Every enqueued part ends with this check, so that such a launch fails the decode it belongs to, by name.
static void CheckLaunches(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) throw ParseError(std::string("kernel launch rejected by the runtime (") + what + "): " + hipGetErrorString(e), false);
}

void Batch::RunPart(void* stream_v, int part, bool timed) {
  // part 0 = whole decode, 1 = front (LF decode + LF post-processing), 2 = rest (HF decode, IDCT, filters, output);
  // the rest can be enqueued in two pieces, 3 = HF decode only, 4 = everything after it, so that a caller can record an
  // event between them (bench.py starts the LF stage of a later batch when an HF stage has ended, not when it starts).
  hipStream_t stream = (hipStream_t)stream_v;
  if (!prepared_) Prepare(stream_v);
  // (whatever an earlier runtime call of this thread left unread is not this decode's — it is named on stderr once, not silently dropped, and then cleared so that
  // the checks behind this part's launches report this part's launches only)
  if (const hipError_t stale = hipPeekAtLastError()) {
    static std::atomic<bool> told{false};
    if (!told.exchange(true)) fprintf(stderr, "[jxl-hip] a HIP error left pending by an earlier call on this thread (not by this decode) is being cleared: %s\n", hipGetErrorString(stale));
    (void)hipGetLastError();
  }
  const int n = (int)images_.size();
  if (any_vardct_) CheckFilterBuffers();
  // The front, too, comes in two pieces: 5 = LF decode (what the HF stage needs: block info, varblock lists, coefficient offsets),
  // 6 = LF post-processing (dequantised LF, LLF, EPF sigma: what the IDCT and the filters need) — a pipelined caller lets the HF stage
  // wait for piece 5 only, so that the post-processing kernels, which wait for wavefront slots beside the pixel kernels of the batch
  // before, are off the critical path.
  const bool do_lf = part == 0 || part == 1 || part == 5, do_lfpost = part == 0 || part == 1 || part == 6, do_front = do_lf || do_lfpost;
  // ... and so may the tail: 7 = IDCT (the last stage that touches the coefficient planes: a caller that rotates coefficient sets lets the next
  // HF stage of that set start behind it), 8 = restoration filters, colour, write
  const bool do_hf = part == 0 || part == 2 || part == 3, do_tail = part == 0 || part == 2 || part == 4 || part == 7 || part == 8;
  const bool do_idct = do_tail && part != 8, do_post = do_tail && part != 7;
  const bool split = part != 0;                       // halves timed separately
  if (do_hf || do_tail) ran_once_ = true;
  if (do_hf) decodes_since_finish_++;
  vec<void*>* evs = nullptr;
  if (timed) {
    if (do_lf) { timed_events_.emplace_back(9, nullptr); }
    if (timed_events_.empty()) timed_events_.emplace_back(9, nullptr);
    evs = !do_front ? &timed_events_[timed_rest_cursor_ < timed_events_.size() ? timed_rest_cursor_ : timed_events_.size() - 1] : &timed_events_.back();
  }
  auto rec = [&](int i) {
    if (!evs) return;
    hipEvent_t ev; HIP_CHECK(hipEventCreate(&ev)); (*evs)[i] = ev;
    HIP_CHECK(hipEventRecord(ev, stream));
  };
  if (do_lf) {
    rec(0);
    DebugSync("start", stream_v);
    if (any_modchan_) LaunchModularGlobal(dframes_, n, cfg, stream_v);   // Modular frames; extra channels of VarDCT frames
    DebugSync("modular global", stream_v);
    cfg.lf_head_start = part == 1 || part == 5;   // front enqueued on its own: a pipelined caller, the HF stage of another batch is about to start
    if (any_vardct_) LaunchLfDecode(dframes_, n, max_lf_groups_, cfg, stream_v, &lf_simt_);
    cfg.lf_wide_once = 0;
    DebugSync("LF decode", stream_v);
    CheckLaunches("LF stage");
    if (any_vardct_ && !cfg.idct_flags_known && part != 0 && !cfg.no_flag_wait) {
      // a caller that enqueues the stages separately: the placement flags travel to pinned host memory behind the LF stage, and the tail —
      // enqueued steps later — waits for that copy (long done by then) instead of launching every IDCT kernel variant
      if (!flags_pinned_ || flags_pinned_n_ < (size_t)n) {
        if (flags_pinned_) (void)hipHostFree(flags_pinned_);      // (waits for the device: the capacity is generous so that a refilled batch object rarely gets here)
        flags_pinned_ = nullptr;
        const size_t cap = std::max<size_t>(4096, 2 * (size_t)n);
        HIP_CHECK(hipHostMalloc((void**)&flags_pinned_, cap * 4));
        flags_pinned_n_ = cap;
      }
      if (!flags_event_) { hipEvent_t ev; HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); flags_event_ = ev; }
      HIP_CHECK(hipMemcpyAsync(flags_pinned_, dwork_ + flags_off_, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipEventRecord((hipEvent_t)flags_event_, stream));
      flags_pending_ = true;
    }
    rec(1);
  }
  if (do_lfpost) {
    if (lf_batch_) lf_batch_->RunPart(stream_v, 0, false);     // LF frames: decoded to the end, their planes copied into the LF planes of the units that refer to them
    if (any_vardct_) LaunchLfPost(dframes_, n, max_bw_, max_bh_, max_groups_, stream_v);
    DebugSync("LF post", stream_v);
    CheckLaunches("LF post-processing");
    if (part == 1 || part == 6) rec(2);
  }
  if (!any_vardct_) {
    if (do_hf) { rec(split ? 7 : 2); rec(3); }
    if (do_tail) {
      if (do_idct) rec(4);
      if (do_post) {
        rec(5);
        if (any_modchan_) EnqueueModularTail(stream_v);
        if (any_complex_) EnqueuePostOps(stream_v);
        CheckLaunches("Modular sub-streams / frame tail");
        rec(6);
        if (timed && split) timed_rest_cursor_++;
      }
    }
    return;
  }
  if (do_hf) {
    // the HF decoder only writes non-zero coefficients into cleared planes (outside the per-stage brackets when the
    // halves are timed separately)
    ClearCoefficientsBeforeHf(stream_v);
    rec(split ? 7 : 2);
    // (progressive flush at the kDC step: no AC group is decoded — the planes stay zero, the IDCT sees the LF part only)
    if (!cfg.skip_hf) LaunchHfDecode(dframes_, n, max_groups_, cfg, stream_v);
    DebugSync("HF decode", stream_v);
    if (any_modchan_) EnqueueModularTail(stream_v);   // (the PassGroup Modular parts start where the HF streams ended)
    LaunchZeroFailedCoefficients(dframes_, n, stream_v);   // (frames that failed up to here are skipped by the IDCT kernels: their planes are zeroed now)
    CheckLaunches("HF stage / Modular sub-streams");
    rec(3);
  }
  if (do_tail) {
    if (!cfg.idct_flags_known && flags_pending_ && part != 0) {
      HIP_CHECK(hipEventSynchronize((hipEvent_t)flags_event_));
      ApplyIdctFlags(flags_pinned_);
      flags_pending_ = false;
    }
    if (do_idct) {
      if (split) rec(8);                        // the tail may sit on another stream than the HF stage: its own start mark
      LaunchIdct(dframes_, n, max_groups_, max_bw_, max_bh_, cfg, stream_v);
      DebugSync("IDCT", stream_v);
      CheckLaunches("IDCT stage");
      rec(4);
      ClearCoefficientsAfterDecode(stream_v);   // (the IDCT kernels zeroed what they read)
    }
    if (do_post) {
      if (cfg.debug_stop_after != 1) LaunchFilters(dframes_, n, max_w_, max_h_, fplan_, cfg, stream_v);
      DebugSync("filters", stream_v);
      rec(5);
      if (!cfg.debug_stop_after) LaunchOutput(dframes_, n, max_w_, max_h_, fplan_, cfg, stream_v);
      if (any_complex_ && !cfg.debug_stop_after) EnqueuePostOps(stream_v);   // frame tail of multi-frame / feature images
      DebugSync("output / frame tail", stream_v);
      CheckLaunches("filters / output / frame tail");
      rec(6);
      if (timed && split) timed_rest_cursor_++;
    }
  }
}

// (tracing) when the nine stage marks of the timed decode recorded last were reached, in ms after `ref_event`: LF start, LF end, LF post end, HF end, IDCT end,
// filters end, output end, HF start, IDCT start; -1 = not recorded
void Batch::DebugTimeline(void* ref_event, float out[9]) {
  for (int i = 0; i < 9; i++) out[i] = -1;
  if (timed_events_.empty()) return;
  auto& evs = timed_events_.back();
  for (int i = 0; i < 9 && i < (int)evs.size(); i++) {
    if (!evs[(size_t)i]) continue;
    if (hipEventSynchronize((hipEvent_t)evs[(size_t)i]) != hipSuccess) { (void)hipGetLastError(); continue; }
    float ms = 0;
    if (hipEventElapsedTime(&ms, (hipEvent_t)ref_event, (hipEvent_t)evs[(size_t)i]) == hipSuccess) out[i] = ms; else (void)hipGetLastError();
  }
}

StageTimes Batch::CollectTimes(int* runs) {
  StageTimes t;
  *runs = (int)timed_events_.size();
  for (auto& evs : timed_events_) {
    bool complete = true;
    for (size_t i = 0; i < 7; i++) if (!evs[i]) complete = false;
    if (!complete) { for (auto e : evs) if (e) (void)hipEventDestroy((hipEvent_t)e); (*runs)--; continue; }
    HIP_CHECK(hipEventSynchronize((hipEvent_t)evs[6]));
    float* dst[6] = {&t.lf_ms, &t.lfpost_ms, &t.hf_ms, &t.idct_ms, &t.filter_ms, &t.out_ms};
    for (int i = 0; i < 6; i++) { float ms = 0; void* st = (i == 2 && evs[7]) ? evs[7] : (i == 3 && evs[8]) ? evs[8] : evs[i]; HIP_CHECK(hipEventElapsedTime(&ms, (hipEvent_t)st, (hipEvent_t)evs[i + 1])); *dst[i] += ms; }
    float tot = 0; HIP_CHECK(hipEventElapsedTime(&tot, (hipEvent_t)evs[0], (hipEvent_t)evs[6])); t.total_ms += tot;
    for (auto e : evs) if (e) (void)hipEventDestroy((hipEvent_t)e);
  }
  timed_events_.clear();
  timed_rest_cursor_ = 0;
  return t;
}

void Batch::Finish(void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (lf_batch_) lf_batch_->Finish(stream_v);            // (a damaged LF frame fails the decode like a damaged frame)
  HIP_CHECK(hipStreamSynchronize(stream));
  const int n = (int)images_.size();
  vec<uint32_t> status(n, 0);
  HIP_CHECK(hipMemcpy(status.data(), dwork_ + status_off_, (size_t)n * 4, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) {
    if (!status[i]) continue;
    HIP_CHECK(hipMemset(dwork_ + status_off_, 0, (size_t)n * 4));
    CoefDirty() = true;                                  // (a failed stream may have written where no IDCT looked)
    if (status[i] & kErrUnsupported) throw ParseError("unsupported: stream feature on the device path (frame " + std::to_string(i) + ")", true);
    throw ParseError("corrupt stream (device status " + std::to_string(status[i]) + ", frame " + std::to_string(i) + ")", false);
  }
  if (any_vardct_ && decodes_since_finish_ > 0) {
    // non-zero coefficients per decode of every frame (deterministic per stream): the HF stage's written bytes for StageBytes
    vec<uint32_t> cnt(n, 0);
    HIP_CHECK(hipMemcpy(cnt.data(), dwork_ + hfw_off_, (size_t)n * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) hf_written_[i] = cnt[i] / decodes_since_finish_;
    HIP_CHECK(hipMemset(dwork_ + hfw_off_, 0, (size_t)n * 4));
    decodes_since_finish_ = 0;
  }
  if (any_vardct_ && !cfg.idct_flags_known && ran_once_) {
    // the LF stage has classified every frame's varblock placement: later decodes of this batch skip the kernels nobody needs
    vec<uint32_t> flags(n, 0);
    HIP_CHECK(hipMemcpy(flags.data(), dwork_ + flags_off_, (size_t)n * 4, hipMemcpyDeviceToHost));
    ApplyIdctFlags(flags.data());
  }
}

// Per-frame flags of the varblock placement (LfPlaceBand) -> which IDCT kernels a decode of this batch needs.
void Batch::ApplyIdctFlags(const uint32_t* flags) {
  const int n = (int)images_.size();
  cfg.any_irregular_blocks = cfg.any_big_blocks = 0;
  cfg.need_tile4_plain = cfg.need_tile4_special = cfg.need_tile8_plain = cfg.need_tile8_special = 0;
  for (int i = 0; i < n; i++) {
    if (images_[i]->plan.modular) continue;
    const uint32_t v = flags[i];
    cfg.any_irregular_blocks |= (v & 1) != 0; cfg.any_big_blocks |= (v & 2) != 0;
    if (v & 1) continue;                                  // generic IdctKernel frame
    if (v & 4) { if (v & 8) cfg.need_tile8_special = 1; else cfg.need_tile8_plain = 1; }
    else { if (v & 8) cfg.need_tile4_special = 1; else cfg.need_tile4_plain = 1; }
  }
  cfg.idct_flags_known = 1;
}

// ---- JPEG reconstruction -----------------------------------------------------------------------------------------------------------------
bool Batch::CanReconstructJpeg(int i, std::string* why) {
  auto no = [&](const char* m) { if (why) *why = m; return false; };
  const PubImage& pi = pub_[i];
  const ImageEntry& e = *images_[pi.first_unit];
  if (e.shared->boxes.jbrd.empty()) return no("no jbrd box");
  const FramePlan& p = e.plan;
  if (pi.num_units != 1 || (pi.complex && !p.subsampled) || p.modular || e.ih.xyb_encoded || !p.do_ycbcr && e.ih.color_space != 1) return no("not a plain JPEG-transcoded frame");
  if (p.upsampling != 1 || p.num_passes != 1 || !e.ih.extra.empty()) return no("not a plain JPEG-transcoded frame");
  if (p.subsampled && ((p.flags & (1 | 2 | 16)) || p.have_crop || p.frame_type != 0)) return no("not a plain JPEG-transcoded frame");
  if (jpeg_data_.size() < pub_.size()) jpeg_data_.resize(pub_.size());
  if (!jpeg_data_[i]) {
    std::unique_ptr<JpegData> jd(new JpegData());
    std::string err;
    const MetadataBoxes& bx = e.shared->boxes;
    if (!ParseJbrd(bx.jbrd.data(), bx.jbrd.size(), jd.get(), &err)) { if (why) *why = err; return false; }
    JpegMetadataSources src;
    src.icc = e.ih.icc.data(); src.icc_size = e.ih.icc.size();
    src.exif = bx.have_exif ? bx.exif.data() : nullptr; src.exif_size = bx.exif.size(); src.exif_brob = bx.exif_brob;
    src.xml = bx.have_xml ? bx.xml.data() : nullptr; src.xml_size = bx.xml.size(); src.xml_brob = bx.xml_brob;
    if (!FillJpegMetadata(jd.get(), src, &err)) { if (why) *why = err; return false; }
    if (jd->components.size() != 3 && jd->components.size() != 1) return no("component count");
    for (uint8_t m : jd->marker_order) if (m == 0xC9 || m == 0xCA) return no("unsupported: arithmetic-coded JPEG");
    jpeg_data_[i] = std::move(jd);
  }
  return true;
}

vec<uint8_t> Batch::ReconstructJpeg(int i, void* stream_v) {
  std::string why;
  if (!CanReconstructJpeg(i, &why)) throw ParseError("JPEG reconstruction: " + why, true);
  hipStream_t stream = (hipStream_t)stream_v;
  if (!prepared_) Prepare(stream_v);
  const int u = pub_[i].first_unit;
  const ImageEntry& e = *images_[u];
  const FramePlan& p = e.plan;
  JpegData& jd = *jpeg_data_[i];
  {
    // quantisation tables from the frame's RAW table (HfGlobal; for one-group frames known only after Prepare): jxl channels X, Y, B
    // = Cb, Y, Cr, and libjxl's coefficient layout is the transpose of JPEG's
    const QuantTableSpec& q = p.qspec[0];
    if (q.mode != 7) throw ParseError("JPEG reconstruction: quantisation table is not a RAW (JPEG) table", true);
    for (size_t c = 0; c < jd.components.size(); c++) {
      const int ch = jd.components.size() == 1 ? 1 : (c == 0 ? 1 : c == 1 ? 0 : 2);
      if (q.raw[ch].size() != 64) throw ParseError("JPEG reconstruction: RAW table size", true);
      JpegQuantTable& t = jd.quant[jd.components[c].quant_idx];
      for (int v = 0; v < 8; v++) for (int u = 0; u < 8; u++) t.values[v * 8 + u] = q.raw[ch][u * 8 + v];
    }
  }
  // entropy decode only: LF stage (LF coefficients = JPEG DC, block metadata) and HF stage (AC coefficients)
  RunPart(stream_v, 1, false);
  RunPart(stream_v, 3, false);
  const size_t ncomp = jd.components.size();
  // component planes: the frame's (padded) block grid shifted by the channel's subsampling; sampling factors for the SOF marker from
  // the frame header (libjxl dec_frame.cc: h_samp_factor = 1 << (max shift - shift), JPEG components Y, Cb, Cr = channels 1, 0, 2)
  size_t comp_blocks[3] = {0, 0, 0}, comp_first[3] = {0, 0, 0}, nblk = 0;
  {
    int max_hs = 0, max_vs = 0;
    for (int c = 0; c < 3; c++) { max_hs = std::max(max_hs, (int)p.hs[c]); max_vs = std::max(max_vs, (int)p.vs[c]); }
    for (size_t c = 0; c < ncomp; c++) {
      const int ch = ncomp == 1 ? 1 : (c == 0 ? 1 : c == 1 ? 0 : 2);
      jd.components[c].h_samp = 1u << (max_hs - p.hs[ch]);
      jd.components[c].v_samp = 1u << (max_vs - p.vs[ch]);
      comp_blocks[c] = (size_t)(p.bw >> p.hs[ch]) * (p.bh >> p.vs[ch]);
      comp_first[c] = nblk;
      nblk += comp_blocks[c];
    }
  }
  int16_t* dcoef = nullptr;
  HIP_CHECK(hipMalloc((void**)&dcoef, nblk * 64 * sizeof(int16_t)));
  struct DevFree { int16_t* p; ~DevFree() { if (p) (void)hipFree(p); } } dcoef_guard{dcoef};   // (released on every exit, incl. exceptions below)
  JpegCoefArgs a;
  memset(&a, 0, sizeof(a));
  a.ncomp = (uint32_t)ncomp;
  for (size_t c = 0; c < ncomp; c++) a.comp_off[c] = (uint32_t)comp_first[c];
  for (size_t c = 0; c < ncomp; c++) for (int k = 0; k < 64; k++) a.qt[c][k] = jd.quant[jd.components[c].quant_idx].values[k];
  a.out = dcoef;
  LaunchJpegCoefficients(dframes_, u, a, p.bw, p.bh, stream_v);
  DebugSync("JPEG coefficients", stream_v);
  vec<int16_t> host(nblk * 64);
  hipError_t err = hipMemcpyAsync(host.data(), dcoef, host.size() * sizeof(int16_t), hipMemcpyDeviceToHost, stream);
  if (err == hipSuccess) err = hipStreamSynchronize(stream);
  if (err != hipSuccess) throw ParseError(std::string("HIP error: ") + hipGetErrorString(err), false);
  Finish(stream_v);
  const int16_t* planes[3] = {host.data(), host.data() + comp_first[1] * 64, host.data() + comp_first[2] * 64};
  vec<uint8_t> out;
  if (!WriteJpeg(jd, e.ih.xsize, e.ih.ysize, planes, &out, &why)) throw ParseError(why, true);
  return out;
}

void Batch::CopyOutputSlotToHost(int i, int slot, void* dst, size_t size, void* stream_v) {
  const ImageEntry& e = *images_[pub_[i].first_unit];
  if (slot < 0 || (size_t)slot >= e.deliver_frames.size() || e.out.device_ptr) throw ParseError("CopyOutputSlotToHost: no such slot", false);
  HIP_CHECK(hipMemcpyAsync(dst, dwork_ + e.off_out + (size_t)slot * (e.out_size + 64), std::min(size, e.out_size), hipMemcpyDeviceToHost, (hipStream_t)stream_v));
  HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_v));
}
void Batch::CopyOutputToHost(int i, void* dst, size_t size, void* stream_v) {
  const ImageEntry& e = *images_[pub_[i].first_unit];
  HIP_CHECK(hipMemcpyAsync(dst, device_output(i), std::min(size, e.out_size), hipMemcpyDeviceToHost, (hipStream_t)stream_v));
  HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_v));
}

}  // namespace jxlhip
