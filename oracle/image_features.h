// ORACLE — TEST INFRASTRUCTURE ONLY (see bits.h).
// Image features and frame compositing: restates libjxl v0.11.2 lib/jxl/{dec_patch_dictionary.cc, splines.cc,
// dec_noise.cc, blending.cc, base/fast_math-inl.h (FastErff, FastCosf), base/random.h (Xorshift128Plus)} and
// lib/jxl/render_pipeline/{stage_patches,stage_splines,stage_noise,stage_blending}.cc.
// patch and spline *syntax* is [V] on samples/sample_grey.jxl and samples/2bit.jxl
// (every stream ends in ANS state 0x130000 and the decoded fields equal App. C's); rendering is recalled ([R]) and
// PARITY UNPINNED against libjxl — the fast-math constants are checked against the exact functions
// (tests/test_oracle_goldens.py: |FastErff - erf| < 7e-4, |FastCosf - cos| < 2e-5).
#pragma once
#include "headers.h"
#include "render.h"
#include "vardct.h"
#include <algorithm>
#include <cmath>

namespace jxlo {

// ---- reference frames (dec_cache.h ReferenceFrame) ------------------------------------------------------------------
struct RefFrame {
  bool valid = false;
  bool is_xyb = false;      // saved before the colour transform
  int w = 0, h = 0;
  Image3 color;
  std::vector<Plane> extra;
};

// ---- patches (dec_patch_dictionary.cc) ---------------------------------------------------------------------------------
enum PatchBlendMode { kPatchNone = 0, kPatchReplace, kPatchAdd, kPatchMul, kPatchBlendAbove, kPatchBlendBelow, kPatchAlphaAddAbove, kPatchAlphaAddBelow, kNumPatchBlendModes };
struct PatchBlend { uint32_t mode = 0, alpha_channel = 0; bool clamp = false; };
struct PatchPos { int64_t x = 0, y = 0; std::vector<PatchBlend> blend; };
struct PatchRef { uint32_t ref = 0, x0 = 0, y0 = 0, xsize = 0, ysize = 0; std::vector<PatchPos> pos; };
struct PatchDictionary { std::vector<PatchRef> refs; };

inline bool PatchUsesAlpha(uint32_t m) { return m == kPatchBlendAbove || m == kPatchBlendBelow || m == kPatchAlphaAddAbove || m == kPatchAlphaAddBelow; }
inline bool PatchUsesClamp(uint32_t m) { return PatchUsesAlpha(m) || m == kPatchMul; }

// PatchDictionary::Decode: 10 contexts {0 #refs, 1 reference frame, 2 size-1, 3 position in the reference, 4 first position,
// 5 blend mode, 6 position delta, 7 count-1, 8 alpha channel, 9 clamp}
inline void ReadPatches(BitReader& br, size_t num_extra, size_t frame_pixels, PatchDictionary& pd) {
  EntropyCode ec;
  ReadEntropyCode(br, 10, ec);
  SymbolReader sr;
  sr.Init(&ec, br);
  const uint32_t num = sr.Read(br, 0);
  if ((uint64_t)num > frame_pixels + 1024) JXLO_FAIL("too many patches");
  pd.refs.resize(num);
  size_t total = 0;
  for (auto& r : pd.refs) {
    r.ref = sr.Read(br, 1);
    if (r.ref >= 4) JXLO_FAIL("patch reference frame out of range");
    r.x0 = sr.Read(br, 3); r.y0 = sr.Read(br, 3);
    r.xsize = sr.Read(br, 2) + 1; r.ysize = sr.Read(br, 2) + 1;
    const uint32_t count = sr.Read(br, 7) + 1;
    total += count;
    if (total > frame_pixels + 1024) JXLO_FAIL("too many patch positions");
    r.pos.resize(count);
    for (uint32_t i = 0; i < count; i++) {
      PatchPos& p = r.pos[i];
      if (i == 0) { p.x = sr.Read(br, 4); p.y = sr.Read(br, 4); }
      else { p.x = r.pos[i - 1].x + UnpackSigned(sr.Read(br, 6)); p.y = r.pos[i - 1].y + UnpackSigned(sr.Read(br, 6)); }
      if (p.x < 0 || p.y < 0) JXLO_FAIL("patch position out of range");
      p.blend.resize(1 + num_extra);
      for (auto& b : p.blend) {
        b.mode = sr.Read(br, 5);
        if (b.mode >= kNumPatchBlendModes) JXLO_FAIL("bad patch blend mode");
        if (PatchUsesAlpha(b.mode) && num_extra > 1) { b.alpha_channel = sr.Read(br, 8); if (b.alpha_channel >= num_extra) JXLO_FAIL("bad patch alpha channel"); }
        if (PatchUsesClamp(b.mode)) b.clamp = sr.Read(br, 9) != 0;
      }
    }
  }
  if (!sr.CheckFinal()) JXLO_FAIL("patch dictionary ANS final state");
}

inline float Clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

// blending.cc PerformBlending, one sample of one channel.  fg/bg naming follows the mode: "above" = patch over frame.
inline float BlendSample(uint32_t mode, bool clamp, bool premultiplied, float frame, float patch, float frame_a, float patch_a) {
  switch (mode) {
    case kPatchNone: return frame;
    case kPatchReplace: return patch;
    case kPatchAdd: return frame + patch;
    case kPatchMul: return frame * (clamp ? Clamp01(patch) : patch);
    case kPatchBlendAbove: case kPatchBlendBelow: {
      const bool above = mode == kPatchBlendAbove;
      const float fg = above ? patch : frame, bg = above ? frame : patch;
      float fa = above ? patch_a : frame_a; const float ba = above ? frame_a : patch_a;
      if (clamp) fa = Clamp01(fa);
      if (premultiplied) return fg + bg * (1.0f - fa);
      const float new_a = 1.0f - (1.0f - fa) * (1.0f - ba);
      const float rnew_a = new_a > 0 ? 1.0f / new_a : 0.0f;
      return (fg * fa + bg * ba * (1.0f - fa)) * rnew_a;
    }
    case kPatchAlphaAddAbove: { const float a = clamp ? Clamp01(patch_a) : patch_a; return frame + patch * a; }
    default: { const float a = clamp ? Clamp01(frame_a) : frame_a; return patch + frame * a; }
  }
}
inline float BlendAlphaSample(uint32_t mode, bool clamp, float frame_a, float patch_a) {  // the alpha channel itself
  switch (mode) {
    case kPatchBlendAbove: case kPatchBlendBelow: {
      float fa = mode == kPatchBlendAbove ? patch_a : frame_a; const float ba = mode == kPatchBlendAbove ? frame_a : patch_a;
      if (clamp) fa = Clamp01(fa);
      return 1.0f - (1.0f - fa) * (1.0f - ba);
    }
    case kPatchAlphaAddAbove: return frame_a;
    case kPatchAlphaAddBelow: return patch_a;
    default: return 0.0f;   // (not reached)
  }
}

// stage_patches.cc / PatchDictionary::AddOneRow: patches are applied in dictionary order onto the frame planes
inline void ApplyPatches(const PatchDictionary& pd, const RefFrame* refs, bool frame_is_xyb, const std::vector<bool>& ec_premultiplied,
                         Image3& img, std::vector<Plane>& extra) {
  const int w = img.w(), h = img.h();
  for (const PatchRef& r : pd.refs) {
    const RefFrame& rf = refs[r.ref];
    if (!rf.valid) JXLO_FAIL("patch refers to an empty reference slot");
    if (rf.is_xyb != frame_is_xyb) JXLO_FAIL("patch reference frame is in a different colour space than the frame");
    if ((uint64_t)r.x0 + r.xsize > (uint64_t)rf.w || (uint64_t)r.y0 + r.ysize > (uint64_t)rf.h) JXLO_FAIL("patch exceeds its reference frame");
    if (rf.extra.size() != extra.size()) JXLO_FAIL("patch reference frame has a different number of extra channels");
    for (const PatchPos& p : r.pos) {
      if (p.x + r.xsize > w || p.y + r.ysize > h) JXLO_FAIL("patch exceeds the frame");
      for (uint32_t iy = 0; iy < r.ysize; iy++) {
        const int fy = (int)p.y + (int)iy;
        for (uint32_t ix = 0; ix < r.xsize; ix++) {
          const int fx = (int)p.x + (int)ix;
          const int rx = r.x0 + ix, ry = r.y0 + iy;
          // alpha values before this patch touches them (blending.cc reads all inputs, then writes)
          const PatchBlend& b0 = p.blend[0];
          float fa = 1.0f, pa = 1.0f;
          bool premul = false;
          if (PatchUsesAlpha(b0.mode)) {
            if (extra.empty()) JXLO_FAIL("alpha patch blending without extra channels");
            fa = extra[b0.alpha_channel].row(fy)[fx]; pa = rf.extra[b0.alpha_channel].row(ry)[rx];
            premul = ec_premultiplied[b0.alpha_channel];
          }
          std::vector<float> ec_out(extra.size());
          for (size_t e = 0; e < extra.size(); e++) {
            const PatchBlend& b = p.blend[1 + e];
            const float fv = extra[e].row(fy)[fx], pv = rf.extra[e].row(ry)[rx];
            float efa = 1.0f, epa = 1.0f;
            if (PatchUsesAlpha(b.mode)) { efa = extra[b.alpha_channel].row(fy)[fx]; epa = rf.extra[b.alpha_channel].row(ry)[rx]; }
            if (PatchUsesAlpha(b.mode) && b.alpha_channel == e) ec_out[e] = BlendAlphaSample(b.mode, b.clamp, efa, epa);
            else ec_out[e] = BlendSample(b.mode, b.clamp, PatchUsesAlpha(b.mode) ? (bool)ec_premultiplied[b.alpha_channel] : false, fv, pv, efa, epa);
          }
          for (int c = 0; c < 3; c++) {
            float& o = img.p[c].row(fy)[fx];
            o = BlendSample(b0.mode, b0.clamp, premul, o, rf.color.p[c].row(ry)[rx], fa, pa);
          }
          for (size_t e = 0; e < extra.size(); e++) extra[e].row(fy)[fx] = ec_out[e];
        }
      }
    }
  }
}

// ---- fast math (base/fast_math-inl.h) ----------------------------------------------------------------------------------
inline float FastErff(float x) {
  const bool xle0 = x <= 0.0f;
  const float absx = std::fabs(x);
  float d = std::fmaf(absx, 7.77394369e-02f, 2.05260015e-04f);
  d = std::fmaf(d, absx, 2.32120216e-01f);
  d = std::fmaf(d, absx, 2.77820801e-01f);
  d = std::fmaf(d, absx, 1.0f);
  const float d2 = d * d;
  const float inv = 1.0f / d2;
  const float r = std::fmaf(-inv, inv, 1.0f);
  return xle0 ? -r : r;
}
inline float FastCosf(float x) {
  const float kPi = 3.14159265358979323846f;
  const float pi2 = kPi * 2.0f, pi2_inv = 0.5f / kPi;
  const float npi2 = std::floor(x * pi2_inv) * pi2;
  const float xmodpi2 = x - npi2;
  const float x_pi = std::min(xmodpi2, pi2 - xmodpi2);
  const bool above = x_pi >= kPi / 2.0f;
  const float x_pihalf = above ? kPi - x_pi : x_pi;
  const float xs = x_pihalf * 0.25f;
  const float x2 = xs * xs, x4 = x2 * x2;
  const float pre = std::fmaf(x4, 0.06960438f, std::fmaf(x2, -0.84087373f, 1.68179268f));
  const float s1 = std::fmaf(pre, pre, -1.414213562f);
  const float s2 = std::fmaf(s1, s1, -1.0f);
  return above ? -s2 : s2;
}

// ---- splines (splines.cc) ------------------------------------------------------------------------------------------------
struct SplinePoint { float x, y; };
struct QuantizedSpline {
  std::vector<std::pair<int64_t, int64_t>> control_points;  // double deltas
  int32_t color_dct[3][32];
  int32_t sigma_dct[32];
};
struct Spline {
  std::vector<SplinePoint> control_points;
  float color_dct[3][32];
  float sigma_dct[32];
};
struct SplineSegment { float center_x, center_y, maximum_distance, inv_sigma, sigma_over_4_times_intensity, color[3]; };
struct Splines {
  int32_t quantization_adjustment = 0;
  std::vector<QuantizedSpline> splines;
  std::vector<SplinePoint> starting_points;
  // draw cache
  std::vector<SplineSegment> segments;
  std::vector<size_t> segment_indices, segment_y_start;
  bool empty() const { return splines.empty(); }
};

// Splines::Decode: 6 contexts {0 quantisation adjustment, 1 starting position, 2 #splines - 1, 3 #control points,
// 4 control point (double) deltas, 5 DCT coefficients}
inline void ReadSplines(BitReader& br, size_t num_pixels, Splines& s) {
  EntropyCode ec;
  ReadEntropyCode(br, 6, ec);
  SymbolReader sr;
  sr.Init(&ec, br);
  const size_t num_splines = 1 + (size_t)sr.Read(br, 2);
  const size_t max_control_points = std::min<size_t>(1u << 20, num_pixels / 2);
  if (num_splines > max_control_points) JXLO_FAIL("too many splines");
  s.starting_points.resize(num_splines);
  int64_t lx = 0, ly = 0;
  for (size_t i = 0; i < num_splines; i++) {
    int64_t x, y;
    if (i == 0) { x = sr.Read(br, 1); y = sr.Read(br, 1); }
    else { x = lx + UnpackSigned(sr.Read(br, 1)); y = ly + UnpackSigned(sr.Read(br, 1)); }
    if (std::llabs(x) >= (1 << 23) || std::llabs(y) >= (1 << 23)) JXLO_FAIL("spline starting point out of range");
    s.starting_points[i] = {(float)x, (float)y};
    lx = x; ly = y;
  }
  s.quantization_adjustment = UnpackSigned(sr.Read(br, 0));
  s.splines.resize(num_splines);
  size_t total = 0;
  for (auto& q : s.splines) {
    const size_t n = sr.Read(br, 3);
    total += n;
    if (total > max_control_points) JXLO_FAIL("too many spline control points");
    q.control_points.resize(n);
    for (auto& cp : q.control_points) {
      cp.first = UnpackSigned(sr.Read(br, 4)); cp.second = UnpackSigned(sr.Read(br, 4));
      if (std::llabs(cp.first) >= (1 << 30) || std::llabs(cp.second) >= (1 << 30)) JXLO_FAIL("spline delta out of range");
    }
    for (int c = 0; c < 3; c++) for (int i = 0; i < 32; i++) q.color_dct[c][i] = UnpackSigned(sr.Read(br, 5));
    for (int i = 0; i < 32; i++) q.sigma_dct[i] = UnpackSigned(sr.Read(br, 5));
  }
  if (!sr.CheckFinal()) JXLO_FAIL("splines ANS final state");
}

static const float kSplineChannelWeight[4] = {0.0042f, 0.075f, 0.07f, 0.3333f};
inline float InvAdjustedQuant(int32_t adjustment) { return adjustment >= 0 ? 1.0f / (1.0f + 0.125f * adjustment) : 1.0f - 0.125f * adjustment; }

// QuantizedSpline::Dequantize (the area-limit bookkeeping only rejects pathological streams and is omitted)
inline void DequantizeSpline(const QuantizedSpline& q, SplinePoint start, int32_t quant_adjust, float y_to_x, float y_to_b, Spline& out) {
  out.control_points.clear();
  int cx = (int)std::roundf(start.x), cy = (int)std::roundf(start.y);
  out.control_points.push_back({(float)cx, (float)cy});
  int dx = 0, dy = 0;
  for (auto& p : q.control_points) {
    dx += (int)p.first; dy += (int)p.second;
    cx += dx; cy += dy;
    if (std::abs(cx) >= (1 << 23) || std::abs(cy) >= (1 << 23)) JXLO_FAIL("spline control point out of range");
    out.control_points.push_back({(float)cx, (float)cy});
  }
  const float inv_quant = InvAdjustedQuant(quant_adjust);
  const float kSqrt0_5 = 0.70710678118654752440f;
  for (int c = 0; c < 3; c++) for (int i = 0; i < 32; i++) {
    const float inv_dct_factor = i == 0 ? kSqrt0_5 : 1.0f;
    out.color_dct[c][i] = q.color_dct[c][i] * inv_dct_factor * kSplineChannelWeight[c] * inv_quant;
  }
  for (int i = 0; i < 32; i++) {
    out.color_dct[0][i] += y_to_x * out.color_dct[1][i];
    out.color_dct[2][i] += y_to_b * out.color_dct[1][i];
  }
  for (int i = 0; i < 32; i++) {
    const float inv_dct_factor = i == 0 ? kSqrt0_5 : 1.0f;
    out.sigma_dct[i] = q.sigma_dct[i] * inv_dct_factor * kSplineChannelWeight[3] * inv_quant;
  }
}

inline SplinePoint operator+(SplinePoint a, SplinePoint b) { return {a.x + b.x, a.y + b.y}; }
inline SplinePoint operator-(SplinePoint a, SplinePoint b) { return {a.x - b.x, a.y - b.y}; }
inline SplinePoint operator*(float s, SplinePoint a) { return {s * a.x, s * a.y}; }

// splines.cc DrawCentripetalCatmullRomSpline: 16 points per control-point interval
inline void CatmullRom(std::vector<SplinePoint> points, std::vector<SplinePoint>& result) {
  if (points.empty()) return;
  if (points.size() == 1) { result.push_back(points[0]); return; }
  const int kNumPoints = 16;
  points.insert(points.begin(), points[0] + (points[0] - points[1]));
  points.push_back(points[points.size() - 1] + (points[points.size() - 1] - points[points.size() - 2]));
  for (size_t start = 0; start + 3 < points.size(); start++) {
    const SplinePoint* p = &points[start];
    result.push_back(p[1]);
    float d[3], t[4];
    t[0] = 0;
    for (int k = 0; k < 3; k++) {
      d[k] = std::sqrt(hypotf(p[k + 1].x - p[k].x, p[k + 1].y - p[k].y));
      t[k + 1] = t[k] + d[k];
    }
    for (int i = 1; i < kNumPoints; i++) {
      const float tt = d[0] + ((float)i / kNumPoints) * d[1];
      SplinePoint a[3];
      for (int k = 0; k < 3; k++) a[k] = p[k] + ((tt - t[k]) / d[k]) * (p[k + 1] - p[k]);
      SplinePoint b[2];
      for (int k = 0; k < 2; k++) b[k] = a[k] + ((tt - t[k]) / (d[k] + d[k + 1])) * (a[k + 1] - a[k]);
      result.push_back(b[0] + ((tt - t[1]) / d[1]) * (b[1] - b[0]));
    }
  }
  result.push_back(points[points.size() - 2]);
}

// splines.cc ForEachEquallySpacedPoint (desired distance 1)
inline void EquallySpaced(const std::vector<SplinePoint>& points, std::vector<std::pair<SplinePoint, float>>& out) {
  const float kDist = 1.0f;
  if (points.empty()) return;
  SplinePoint current = points.front();
  out.push_back({current, kDist});
  size_t next = 0;
  while (next < points.size()) {
    const SplinePoint* previous = &current;
    float arclength_from_previous = 0.0f;
    for (;;) {
      if (next == points.size()) { out.push_back({*previous, arclength_from_previous}); return; }
      const SplinePoint d = points[next] - *previous;
      const float arclength_to_next = std::sqrt(d.x * d.x + d.y * d.y);
      if (arclength_from_previous + arclength_to_next >= kDist) {
        current = *previous + ((kDist - arclength_from_previous) / arclength_to_next) * (points[next] - *previous);
        out.push_back({current, kDist});
        break;
      }
      arclength_from_previous += arclength_to_next;
      previous = &points[next];
      ++next;
    }
  }
}

// splines.cc ContinuousIDCT: sum_i sqrt2 * dct[i] * cos(pi/32 * i * (t + 0.5)); libjxl evaluates it in SIMD lanes with
// FastCosf and a lane-wise partial-sum order that depends on the build target — scalar order here (PARITY UNPINNED).
inline float ContinuousIDCT(const float dct[32], float t) {
  const float kPi = 3.14159265358979323846f;
  float result = 0.0f;
  const float tandhalf = t + 0.5f;
  for (int i = 0; i < 32; i++) {
    const float cos_arg = (kPi / 32 * i) * tandhalf;
    const float local = dct[i] * FastCosf(cos_arg);
    result = std::fmaf(kSqrt2f, local, result);
  }
  return result;
}

inline void BuildSplineSegments(Splines& s, float y_to_x, float y_to_b) {
  s.segments.clear(); s.segment_indices.clear(); s.segment_y_start.clear();
  std::vector<std::pair<size_t, size_t>> by_y;
  for (size_t i = 0; i < s.splines.size(); i++) {
    Spline sp;
    DequantizeSpline(s.splines[i], s.starting_points[i], s.quantization_adjustment, y_to_x, y_to_b, sp);
    for (size_t k = 1; k < sp.control_points.size(); k++)
      if (sp.control_points[k].x == sp.control_points[k - 1].x && sp.control_points[k].y == sp.control_points[k - 1].y) JXLO_FAIL("identical successive spline control points");
    std::vector<SplinePoint> inter;
    CatmullRom(sp.control_points, inter);
    std::vector<std::pair<SplinePoint, float>> pts;
    EquallySpaced(inter, pts);
    const float arc_length = (float)((double)pts.size() - 2) * 1.0f + pts.back().second;
    if (arc_length <= 0.0f) continue;
    const float inv_arc_length = 1.0f / arc_length;
    int k = 0;
    for (auto& pt : pts) {
      const float progress = std::min(1.0f, ((float)k * 1.0f) * inv_arc_length);
      ++k;
      float color[3];
      for (int c = 0; c < 3; c++) color[c] = ContinuousIDCT(sp.color_dct[c], (32 - 1) * progress);
      const float sigma = ContinuousIDCT(sp.sigma_dct, (32 - 1) * progress);
      const float intensity = pt.second;
      // ComputeSegments
      if (!(std::isfinite(sigma) && sigma != 0.0f && std::isfinite(1.0f / sigma) && std::isfinite(intensity))) continue;
      const float kDistanceExp = 5;
      float max_color = 0.01f;
      for (int c = 0; c < 3; c++) max_color = std::max(max_color, std::fabs(color[c] * intensity));
      const float maximum_distance = std::sqrt(-2 * sigma * sigma * (std::log(0.1) * kDistanceExp - std::log(max_color)));
      SplineSegment seg;
      seg.center_x = pt.first.x; seg.center_y = pt.first.y;
      for (int c = 0; c < 3; c++) seg.color[c] = color[c];
      seg.inv_sigma = 1.0f / sigma;
      seg.sigma_over_4_times_intensity = 0.25f * sigma * intensity;
      seg.maximum_distance = maximum_distance;
      const int64_t y0 = std::llround(pt.first.y - maximum_distance), y1 = std::llround(pt.first.y + maximum_distance) + 1;
      for (int64_t y = std::max<int64_t>(y0, 0); y < y1; y++) by_y.push_back({(size_t)y, s.segments.size()});
      s.segments.push_back(seg);
    }
  }
  std::sort(by_y.begin(), by_y.end());
  s.segment_indices.resize(by_y.size());
  s.segment_y_start.clear();
  for (size_t i = 0; i < by_y.size(); i++) {
    s.segment_indices[i] = by_y[i].second;
    const size_t y = by_y[i].first;
    if (y >= s.segment_y_start.size()) s.segment_y_start.resize(y + 1, i);
  }
  s.segment_y_start.push_back(by_y.size());
}

// Splines::AddTo / DrawSegment: adds every segment's Gaussian-blurred contribution to the three planes
inline void DrawSplines(const Splines& s, Image3& img) {
  const int w = img.w(), h = img.h();
  for (int y = 0; y < h; y++) {
    if ((size_t)y + 1 >= s.segment_y_start.size()) break;
    float* rows[3] = {img.p[0].row(y), img.p[1].row(y), img.p[2].row(y)};
    for (size_t i = s.segment_y_start[y]; i < s.segment_y_start[y + 1]; i++) {
      const SplineSegment& seg = s.segments[s.segment_indices[i]];
      int64_t x0 = std::max<int64_t>(0, std::llround(seg.center_x - seg.maximum_distance));
      const int64_t x1 = std::min<int64_t>(w, std::llround(seg.center_x + seg.maximum_distance) + 1);
      for (int64_t x = x0; x < x1; x++) {
        const float dx = (float)x - seg.center_x, dy = (float)y - seg.center_y;
        const float sqd = std::fmaf(dx, dx, dy * dy);
        const float distance = std::sqrt(sqd);
        const float f = FastErff(std::fmaf(distance, 0.5f, 0.353553391f) * seg.inv_sigma) - FastErff(std::fmaf(distance, 0.5f, -0.353553391f) * seg.inv_sigma);
        const float local_intensity = seg.sigma_over_4_times_intensity * (f * f);
        for (int c = 0; c < 3; c++) rows[c][x] = std::fmaf(seg.color[c], local_intensity, rows[c][x]);
      }
    }
  }
}

// ---- noise (dec_noise.cc, stage_noise.cc, base/random.h) ------------------------------------------------------------------
struct NoiseParams { float lut[8] = {0}; bool HasAny() const { for (float v : lut) if (std::fabs(v) > 1e-3f) return true; return false; } };
inline void ReadNoise(BitReader& br, NoiseParams& n) { for (float& v : n.lut) v = (float)br.u(10) * (1.0f / 1024.0f); }

struct Xorshift128Plus {
  static constexpr int N = 8;
  uint64_t s0[N], s1[N];
  static uint64_t SplitMix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  Xorshift128Plus(uint32_t seed1, uint32_t seed2, uint32_t seed3, uint32_t seed4) {
    s0[0] = SplitMix64((((uint64_t)seed1 << 32) + seed2) + 0x9E3779B97F4A7C15ull);
    s1[0] = SplitMix64((((uint64_t)seed3 << 32) + seed4) + 0x9E3779B97F4A7C15ull);
    for (int i = 1; i < N; i++) { s0[i] = SplitMix64(s0[i - 1]); s1[i] = SplitMix64(s1[i - 1]); }
  }
  void Fill(uint64_t* bits) {
    for (int i = 0; i < N; i++) {
      uint64_t a = s0[i];
      const uint64_t b = s1[i];
      bits[i] = a + b;
      s0[i] = b;
      a ^= a << 23;
      a ^= b ^ (a >> 18) ^ (b >> 5);
      s1[i] = a;
    }
  }
};

// dec_noise.cc Random3Planes / RandomImage: per 256x256 group (in upsampled coordinates) three planes of floats in [1, 2),
// 16 per generator batch, a fresh batch at the start of every row
inline void RandomNoisePlanes(uint32_t visible_frame_index, uint32_t nonvisible_frame_index, int group_dim, int w, int h, Image3& noise) {
  for (int c = 0; c < 3; c++) noise.p[c] = Plane(w, h);
  for (int gy0 = 0; gy0 < h; gy0 += group_dim) {
    for (int gx0 = 0; gx0 < w; gx0 += group_dim) {
      Xorshift128Plus rng(visible_frame_index, nonvisible_frame_index, (uint32_t)gx0, (uint32_t)gy0);
      const int xs = std::min(group_dim, w - gx0), ys = std::min(group_dim, h - gy0);
      for (int c = 0; c < 3; c++) {
        for (int y = 0; y < ys; y++) {
          float* row = noise.p[c].row(gy0 + y) + gx0;
          uint64_t batch[8];
          for (int x = 0; x < xs; x += 16) {
            rng.Fill(batch);
            for (int i = 0; i < 16 && x + i < xs; i++) {
              const uint32_t bits = (uint32_t)(batch[i >> 1] >> ((i & 1) * 32));
              const uint32_t f = (bits >> 9) | 0x3F800000u;
              memcpy(&row[x + i], &f, 4);
            }
          }
        }
      }
    }
  }
}

inline float NoiseStrengthLut(const float* lut, float vx) {
  const float kScale = 6.0f;  // kNumNoisePoints - 2
  float scaled = std::max(0.0f, vx * kScale);
  float floor_x = std::floor(scaled), frac = scaled - floor_x;
  if (scaled >= kScale + 1) { floor_x = kScale; frac = 1.0f; }
  const int i = (int)floor_x;
  const float low = lut[i], hi = lut[i + 1];
  const float v = std::fmaf(hi - low, frac, low);
  return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
}

// stage_noise.cc ConvolveNoiseStage + AddNoiseStage (XYB planes, after upsampling)
inline void AddNoise(const NoiseParams& np, uint32_t visible_frame_index, uint32_t nonvisible_frame_index, int group_dim, float ytox, float ytob, Image3& img) {
  const int w = img.w(), h = img.h();
  Image3 rnd, conv;
  RandomNoisePlanes(visible_frame_index, nonvisible_frame_index, group_dim, w, h, rnd);
  for (int c = 0; c < 3; c++) {
    conv.p[c] = Plane(w, h);
    for (int y = 0; y < h; y++) {
      const float* rows[5];
      for (int i = -2; i <= 2; i++) rows[i + 2] = rnd.p[c].row(Mirror(y + i, h));
      float* out = conv.p[c].row(y);
      for (int x = 0; x < w; x++) {
        auto px = [&](int r, int dx) { return rows[r][Mirror(x + dx, w)]; };
        const float p00 = px(2, 0);
        float others = 0.0f;
        for (int i = -2; i <= 2; i++) { others += px(0, i); others += px(1, i); others += px(3, i); others += px(4, i); }
        others += px(2, -2); others += px(2, -1); others += px(2, 1); others += px(2, 2);
        out[x] = std::fmaf(others, 0.16f, p00 * -3.84f);
      }
    }
  }
  const float kNorm = 0.22f, kRGCorr = 0.9921875f, kRGNCorr = 0.0078125f;
  for (int y = 0; y < h; y++) {
    float* rx = img.p[0].row(y); float* ry = img.p[1].row(y); float* rb = img.p[2].row(y);
    const float* nr = conv.p[0].row(y); const float* ng = conv.p[1].row(y); const float* nc = conv.p[2].row(y);
    for (int x = 0; x < w; x++) {
      const float vx = rx[x], vy = ry[x];
      const float in_g = vy - vx, in_r = vy + vx;
      const float sg = NoiseStrengthLut(np.lut, in_g * 0.5f), srr = NoiseStrengthLut(np.lut, in_r * 0.5f);
      const float ar = nr[x] * kNorm, ag = ng[x] * kNorm, ac = nc[x] * kNorm;
      const float red = srr * std::fmaf(kRGNCorr, ar, kRGCorr * ac);
      const float green = sg * std::fmaf(kRGNCorr, ag, kRGCorr * ac);
      const float rg = red + green;
      rx[x] = std::fmaf(ytox, rg, red - green) + vx;
      ry[x] = vy + rg;
      rb[x] = std::fmaf(ytob, rg, rb[x]);
    }
  }
}

// ---- frame blending (blending.cc PerformBlending for whole frames; stage_blending.cc) ------------------------------------------
// BlendMode of a frame: 0 replace, 1 add, 2 blend, 3 alpha-weighted add (kMulAdd), 4 mul
inline float FrameBlendSample(uint32_t mode, bool clamp, bool premultiplied, float bg, float fg, float bga, float fga) {
  switch (mode) {
    case 0: return fg;
    case 1: return bg + fg;
    case 2: {
      const float fa = clamp ? Clamp01(fga) : fga;
      if (premultiplied) return fg + bg * (1.0f - fa);
      const float new_a = 1.0f - (1.0f - fa) * (1.0f - bga);
      const float rnew_a = new_a > 0 ? 1.0f / new_a : 0.0f;
      return (fg * fa + bg * bga * (1.0f - fa)) * rnew_a;
    }
    case 3: { const float fa = clamp ? Clamp01(fga) : fga; return bg + fg * fa; }
    default: return bg * (clamp ? Clamp01(fg) : fg);
  }
}

}  // namespace jxlo
