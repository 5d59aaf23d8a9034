// jxl-hip: host-side preparation of the image features the GPU renders (see host_parse.h).
// Splines: the centripetal Catmull-Rom curve of every spline is sampled at unit arc length and each sample becomes one
// Gaussian "segment" (splines.cc InitializeDrawCache: Dequantize, DrawCentripetalCatmullRomSpline, ForEachEquallySpacedPoint,
// SegmentsFromPoints, ComputeSegments).  This is O(total arc length) scalar work per frame — the per-pixel accumulation of the
// segments is what kernels_features.hip does.  libjxl evaluates the colour/sigma DCT with FastCosf in SIMD lanes whose partial
// sums depend on the build target; the scalar order used here is shared with the CPU oracle.
#include "host_parse.h"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace jxlhip {

namespace {

struct Pt { float x, y; };
inline Pt operator+(Pt a, Pt b) { return {a.x + b.x, a.y + b.y}; }
inline Pt operator-(Pt a, Pt b) { return {a.x - b.x, a.y - b.y}; }
inline Pt operator*(float s, Pt a) { return {s * a.x, s * a.y}; }

// base/fast_math-inl.h FastCosf
float FastCosf(float x) {
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

// splines.cc ContinuousIDCT
float ContinuousIDCT(const float dct[32], float t) {
  const float kPi = 3.14159265358979323846f, kSqrt2 = 1.41421356237309504880f;
  float result = 0.0f;
  const float tandhalf = t + 0.5f;
  for (int i = 0; i < 32; i++) {
    const float cos_arg = (kPi / 32 * i) * tandhalf;
    const float local = dct[i] * FastCosf(cos_arg);
    result = std::fmaf(kSqrt2, local, result);
  }
  return result;
}

void CatmullRom(vec<Pt> points, vec<Pt>& result) {
  if (points.empty()) return;
  if (points.size() == 1) { result.push_back(points[0]); return; }
  const int kNumPoints = 16;
  points.insert(points.begin(), points[0] + (points[0] - points[1]));
  points.push_back(points[points.size() - 1] + (points[points.size() - 1] - points[points.size() - 2]));
  for (size_t start = 0; start + 3 < points.size(); start++) {
    const Pt* p = &points[start];
    result.push_back(p[1]);
    float d[3], t[4];
    t[0] = 0;
    for (int k = 0; k < 3; k++) {
      d[k] = std::sqrt(hypotf(p[k + 1].x - p[k].x, p[k + 1].y - p[k].y));
      t[k + 1] = t[k] + d[k];
    }
    for (int i = 1; i < kNumPoints; i++) {
      const float tt = d[0] + ((float)i / kNumPoints) * d[1];
      Pt a[3];
      for (int k = 0; k < 3; k++) a[k] = p[k] + ((tt - t[k]) / d[k]) * (p[k + 1] - p[k]);
      Pt b[2];
      for (int k = 0; k < 2; k++) b[k] = a[k] + ((tt - t[k]) / (d[k] + d[k + 1])) * (a[k + 1] - a[k]);
      result.push_back(b[0] + ((tt - t[1]) / d[1]) * (b[1] - b[0]));
    }
  }
  result.push_back(points[points.size() - 2]);
}

void EquallySpaced(const vec<Pt>& points, vec<std::pair<Pt, float>>& out) {
  const float kDist = 1.0f;
  if (points.empty()) return;
  Pt current = points.front();
  out.push_back({current, kDist});
  size_t next = 0;
  while (next < points.size()) {
    const Pt* previous = &current;
    float from_previous = 0.0f;
    for (;;) {
      if (next == points.size()) { out.push_back({*previous, from_previous}); return; }
      const Pt d = points[next] - *previous;
      const float to_next = std::sqrt(d.x * d.x + d.y * d.y);
      if (from_previous + to_next >= kDist) {
        current = *previous + ((kDist - from_previous) / to_next) * (points[next] - *previous);
        out.push_back({current, kDist});
        break;
      }
      from_previous += to_next;
      previous = &points[next];
      ++next;
    }
  }
}

}  // namespace

void BuildSplineDrawList(const FrameFeatures& f, float y_to_x, float y_to_b, uint32_t height, SplineDrawList* out) {
  out->segments.clear(); out->row_start.clear(); out->indices.clear();
  static const float kChannelWeight[4] = {0.0042f, 0.075f, 0.07f, 0.3333f};
  const float kSqrt0_5 = 0.70710678118654752440f;
  vec<std::pair<uint32_t, uint32_t>> by_y;
  const int32_t adj = f.spline_quant_adjust;
  const float inv_quant = adj >= 0 ? 1.0f / (1.0f + 0.125f * adj) : 1.0f - 0.125f * adj;
  for (size_t si = 0; si < f.splines.size(); si++) {
    const SplineH& q = f.splines[si];
    // QuantizedSpline::Dequantize
    vec<Pt> cps;
    int cx = (int)std::roundf((float)f.spline_start[si].first), cy = (int)std::roundf((float)f.spline_start[si].second);
    cps.push_back({(float)cx, (float)cy});
    int dx = 0, dy = 0;
    for (auto& p : q.control_points) {
      dx += (int)p.first; dy += (int)p.second;
      cx += dx; cy += dy;
      if (std::abs(cx) >= (1 << 23) || std::abs(cy) >= (1 << 23)) throw ParseError("spline control point out of range", false);
      cps.push_back({(float)cx, (float)cy});
    }
    for (size_t k = 1; k < cps.size(); k++)
      if (cps[k].x == cps[k - 1].x && cps[k].y == cps[k - 1].y) throw ParseError("identical successive spline control points", false);
    float color_dct[3][32], sigma_dct[32];
    for (int c = 0; c < 3; c++) for (int i = 0; i < 32; i++) {
      const float inv_dct_factor = i == 0 ? kSqrt0_5 : 1.0f;
      color_dct[c][i] = q.color_dct[c][i] * inv_dct_factor * kChannelWeight[c] * inv_quant;
    }
    for (int i = 0; i < 32; i++) {
      color_dct[0][i] += y_to_x * color_dct[1][i];
      color_dct[2][i] += y_to_b * color_dct[1][i];
    }
    for (int i = 0; i < 32; i++) {
      const float inv_dct_factor = i == 0 ? kSqrt0_5 : 1.0f;
      sigma_dct[i] = q.sigma_dct[i] * inv_dct_factor * kChannelWeight[3] * inv_quant;
    }
    vec<Pt> inter;
    CatmullRom(cps, inter);
    vec<std::pair<Pt, float>> pts;
    EquallySpaced(inter, pts);
    if (pts.size() > (1u << 24)) throw ParseError("spline too long", false);
    const float arc_length = (float)((double)pts.size() - 2) * 1.0f + pts.back().second;
    if (arc_length <= 0.0f) continue;
    const float inv_arc_length = 1.0f / arc_length;
    int k = 0;
    for (auto& pt : pts) {
      const float progress = std::min(1.0f, ((float)k * 1.0f) * inv_arc_length);
      ++k;
      float color[3];
      for (int c = 0; c < 3; c++) color[c] = ContinuousIDCT(color_dct[c], (32 - 1) * progress);
      const float sigma = ContinuousIDCT(sigma_dct, (32 - 1) * progress);
      const float intensity = pt.second;
      // ComputeSegments
      if (!(std::isfinite(sigma) && sigma != 0.0f && std::isfinite(1.0f / sigma) && std::isfinite(intensity))) continue;
      const float kDistanceExp = 5;
      float max_color = 0.01f;
      for (int c = 0; c < 3; c++) max_color = std::max(max_color, std::fabs(color[c] * intensity));
      const float maximum_distance = std::sqrt(-2 * sigma * sigma * (std::log(0.1) * kDistanceExp - std::log(max_color)));
      SplineSegmentDev seg;
      seg.center_x = pt.first.x; seg.center_y = pt.first.y;
      for (int c = 0; c < 3; c++) seg.color[c] = color[c];
      seg.inv_sigma = 1.0f / sigma;
      seg.sigma_over_4_times_intensity = 0.25f * sigma * intensity;
      seg.maximum_distance = maximum_distance;
      const long long y0 = std::llround(pt.first.y - maximum_distance), y1 = std::llround(pt.first.y + maximum_distance) + 1;
      for (long long y = std::max<long long>(y0, 0); y < y1 && y < (long long)height; y++) by_y.push_back({(uint32_t)y, (uint32_t)out->segments.size()});
      out->segments.push_back(seg);
      // (row, segment) pairs: bounded by the frame, like splines.cc bounds the estimated area the segments may touch
      if (by_y.size() > ((size_t)height << 14) + (1u << 20)) throw ParseError("splines cover too large an area", false);
    }
  }
  std::sort(by_y.begin(), by_y.end());
  out->indices.resize(by_y.size());
  out->row_start.assign((size_t)height + 1, 0);
  for (size_t i = 0; i < by_y.size(); i++) { out->indices[i] = by_y[i].second; out->row_start[by_y[i].first + 1]++; }
  for (uint32_t y = 0; y < height; y++) out->row_start[y + 1] += out->row_start[y];
}

}  // namespace jxlhip
