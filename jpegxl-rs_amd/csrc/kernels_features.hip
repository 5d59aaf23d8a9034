// jxl-hip: kernels of the frame "tail" that images with more than one frame or with image features take (gfx950):
// integer Modular planes -> float, patches, splines, upsampling, noise, colour transform, blending onto the canvas and the
// write stage — libjxl's render_pipeline stages stage_{patches,splines,upsampling,noise,xyb,from_linear,ycbcr,blending,write}.cc,
// reached by the reference through JxlDecoderProcessInput (jpegxl-rs/src/decode.rs:238).  Explicit arguments, one launch per
// stage and frame, planned by the host (decoder.cc PlanPostOps): these stages are plain streaming passes over the planes
// (HBM-bound, 8..24 B/px each), only frames that need them pay for them — single-frame images without features keep the
// fused tile kernels of kernels.hip.  Arithmetic and operation order are those of oracle/image_features.h.
#include "kernels.h"
#include "host_parse.h"
#include <hip/hip_runtime.h>
#include <math.h>

namespace jxlhip {

namespace {

__device__ __forceinline__ int MirrorF(int x, int size) {
  while (x < 0 || x >= size) x = x < 0 ? -x - 1 : 2 * size - 1 - x;
  return x;
}
__device__ __forceinline__ float Clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

// base/fast_math-inl.h
__device__ __forceinline__ float FastLog2fD(float x) {
  const int32_t x_bits = __float_as_int(x);
  const int32_t exp_bits = x_bits - 0x3f2aaaab;
  const int32_t exp_shifted = exp_bits >> 23;
  const float mantissa = __int_as_float(x_bits - (int32_t)((uint32_t)exp_shifted << 23));
  const float t = mantissa - 1.0f;
  float yp = fmaf(7.4245873327820566E-01f, t, 1.4287160470083755E+00f); yp = fmaf(yp, t, -1.8503833400518310E-06f);
  float yq = fmaf(1.7409343003366853E-01f, t, 1.0096718572241148E+00f); yq = fmaf(yq, t, 9.9032814277590719E-01f);
  return yp / yq + (float)exp_shifted;
}
__device__ __forceinline__ float FastPow2fD(float x) {
  const float floorx = floorf(x);
  const float exp = __int_as_float((int32_t)((uint32_t)((int32_t)floorx + 127) << 23));
  const float frac = x - floorx;
  float num = frac + 1.01749063e+01f;
  num = fmaf(num, frac, 4.88687798e+01f);
  num = fmaf(num, frac, 9.85506591e+01f);
  num = num * exp;
  float den = fmaf(frac, 2.10242958e-01f, -2.22328856e-02f);
  den = fmaf(den, frac, -1.94414990e+01f);
  den = fmaf(den, frac, 9.85506633e+01f);
  return num / den;
}
__device__ __forceinline__ float FastPowfD(float b, float e) { return FastPow2fD(FastLog2fD(b) * e); }
__device__ __forceinline__ float FastErffD(float x) {
  const float absx = fabsf(x);
  float d = fmaf(absx, 7.77394369e-02f, 2.05260015e-04f);
  d = fmaf(d, absx, 2.32120216e-01f);
  d = fmaf(d, absx, 2.77820801e-01f);
  d = fmaf(d, absx, 1.0f);
  const float d2 = d * d;
  const float inv = 1.0f / d2;
  const float r = fmaf(-inv, inv, 1.0f);
  return x <= 0.0f ? -r : r;
}
__device__ __forceinline__ float LinearToSrgbF(float v) {   // cms/transfer_functions-inl.h TF_SRGB (same as kernels.hip LinearToSrgb)
  const float x = fabsf(v);
  const float lin = x * 12.92f;
  const float s = sqrtf(x);
  float yp = 7.352629620e-1f, yq = 2.424867759e-2f;
  yp = fmaf(yp, s, 1.474205315f); yq = fmaf(yq, s, 9.258482155e-1f);
  yp = fmaf(yp, s, 3.903842876e-1f); yq = fmaf(yq, s, 1.340816930f);
  yp = fmaf(yp, s, 5.287254571e-3f); yq = fmaf(yq, s, 3.036675394e-1f);
  yp = fmaf(yp, s, -5.135152395e-4f); yq = fmaf(yq, s, 1.004519624e-2f);
  const float poly = yp / yq;
  return copysignf(x > 0.0031308f ? poly : lin, v);
}

// ---- integer Modular planes -> float planes (dec_modular.cc ModularImageToDecodedRect) ---------------------------------
__global__ void IntToFloatKernel(const int32_t* __restrict__ src, uint32_t src_stride, float* __restrict__ dst, uint32_t dst_stride, uint32_t w, uint32_t h, float factor,
                                 uint32_t float_bits, uint32_t float_exp_bits) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  const int32_t v = src[(size_t)y * src_stride + x];
  dst[(size_t)y * dst_stride + x] = float_bits ? IntToFloatSample(v, float_bits, float_exp_bits) : (float)v * factor;
}
// XYB Modular frames code Y, X, B - Y; factors = the LF dequantisation factors (DequantMatrices::DCQuants)
__global__ void XybModToFloatKernel(const int32_t* __restrict__ cy, const int32_t* __restrict__ cx, const int32_t* __restrict__ cb, uint32_t src_stride,
                                    float* dx, float* dy, float* db, uint32_t dst_stride, uint32_t w, uint32_t h, float fx, float fy, float fb) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  const size_t si = (size_t)y * src_stride + x, di = (size_t)y * dst_stride + x;
  const int32_t vy = cy[si];
  dx[di] = (float)cx[si] * fx;
  dy[di] = (float)vy * fy;
  db[di] = (float)(cb[si] + vy) * fb;
}

// ---- patches (stage_patches.cc; blending.cc PerformBlending) ---------------------------------------------------------------
__device__ __forceinline__ float PatchBlendSample(uint32_t mode, bool clamp, bool premultiplied, float frame, float patch, float frame_a, float patch_a) {
  switch (mode) {
    case 0: return frame;
    case 1: return patch;
    case 2: return frame + patch;
    case 3: return frame * (clamp ? Clamp01(patch) : patch);
    case 4: case 5: {
      const bool above = mode == 4;
      const float fg = above ? patch : frame, bg = above ? frame : patch;
      float fa = above ? patch_a : frame_a; const float ba = above ? frame_a : patch_a;
      if (clamp) fa = Clamp01(fa);
      if (premultiplied) return fg + bg * (1.0f - fa);
      const float new_a = 1.0f - (1.0f - fa) * (1.0f - ba);
      const float rnew_a = new_a > 0 ? 1.0f / new_a : 0.0f;
      return (fg * fa + bg * ba * (1.0f - fa)) * rnew_a;
    }
    case 6: { const float a = clamp ? Clamp01(patch_a) : patch_a; return frame + patch * a; }
    default: { const float a = clamp ? Clamp01(frame_a) : frame_a; return patch + frame * a; }
  }
}
__device__ __forceinline__ float PatchBlendAlpha(uint32_t mode, bool clamp, float frame_a, float patch_a) {
  switch (mode) {
    case 4: case 5: {
      float fa = mode == 4 ? patch_a : frame_a; const float ba = mode == 4 ? frame_a : patch_a;
      if (clamp) fa = Clamp01(fa);
      return 1.0f - (1.0f - fa) * (1.0f - ba);
    }
    case 6: return frame_a;
    default: return patch_a;
  }
}

// One workgroup per 32x32 tile of the frame; the tile's list names, in dictionary order, the patch placements that touch it, so
// every pixel sees its patches in the order libjxl applies them (float additions do not commute).
__global__ __launch_bounds__(256) void PatchKernel(PatchFrameArgs a, const PatchEntryDev* __restrict__ entries, const uint32_t* __restrict__ tile_start,
                                                   const uint32_t* __restrict__ tile_list, uint32_t tiles_x) {
  const uint32_t tile = blockIdx.x;
  const uint32_t begin = tile_start[tile], end = tile_start[tile + 1];
  if (begin == end) return;
  const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
  for (uint32_t t = threadIdx.x; t < 1024; t += blockDim.x) {
    const int x = (int)(tx * 32 + (t & 31)), y = (int)(ty * 32 + (t >> 5));
    if (x >= (int)a.w || y >= (int)a.h) continue;
    const size_t fo = (size_t)y * a.stride + x, eo = (size_t)y * a.ec_stride + x;
    for (uint32_t k = begin; k < end; k++) {
      const PatchEntryDev& e = entries[tile_list[k]];
      const int ix = x - e.x, iy = y - e.y;
      if (ix < 0 || iy < 0 || ix >= (int)e.xs || iy >= (int)e.ys) continue;
      const size_t so = (size_t)iy * e.src_stride + ix, seo = (size_t)iy * e.esrc_stride + ix;
      const uint32_t m0 = e.mode[0] & 0xFF, a0 = (e.mode[0] >> 8) & 0xFF; const bool c0 = (e.mode[0] >> 16) & 1;
      float fa = 1.0f, pa = 1.0f; bool premul = false;
      if (m0 >= 4) { fa = a.ec[a0][eo]; pa = e.esrc[a0][seo]; premul = (a.premul_mask >> a0) & 1; }
      float ec_out[4];
      for (uint32_t c = 0; c < a.num_extra; c++) {
        const uint32_t m = e.mode[1 + c] & 0xFF, ac = (e.mode[1 + c] >> 8) & 0xFF; const bool cl = (e.mode[1 + c] >> 16) & 1;
        const float fv = a.ec[c][eo], pv = e.esrc[c][seo];
        float efa = 1.0f, epa = 1.0f;
        if (m >= 4) { efa = a.ec[ac][eo]; epa = e.esrc[ac][seo]; }
        if (m >= 4 && ac == c) ec_out[c] = PatchBlendAlpha(m, cl, efa, epa);
        else ec_out[c] = PatchBlendSample(m, cl, m >= 4 ? ((a.premul_mask >> ac) & 1) != 0 : false, fv, pv, efa, epa);
      }
      for (int c = 0; c < 3; c++) a.p[c][fo] = PatchBlendSample(m0, c0, premul, a.p[c][fo], e.src[c][so], fa, pa);
      for (uint32_t c = 0; c < a.num_extra; c++) a.ec[c][eo] = ec_out[c];
    }
  }
}

// ---- splines (stage_splines.cc; splines.cc DrawSegment) --------------------------------------------------------------------
// One thread per pixel; the row's segment list is walked in order (ascending segment index, as Splines::Apply does), the
// contributions are added one by one — same order of float additions per pixel as the scalar reference.
__global__ __launch_bounds__(256) void SplineKernel(float* p0, float* p1, float* p2, uint32_t stride, uint32_t w, uint32_t h, const SplineSegmentDev* __restrict__ segs,
                                                    const uint32_t* __restrict__ row_start, const uint32_t* __restrict__ indices) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w || y >= h) return;
  const uint32_t begin = row_start[y], end = row_start[y + 1];
  if (begin == end) return;
  const size_t o = (size_t)y * stride + x;
  float v0 = p0[o], v1 = p1[o], v2 = p2[o];
  bool touched = false;
  for (uint32_t i = begin; i < end; i++) {
    const SplineSegmentDev s = segs[indices[i]];
    // column range of the segment: [llround(cx - maxdist), llround(cx + maxdist)]
    const long long x0 = llroundf(s.center_x - s.maximum_distance), x1 = llroundf(s.center_x + s.maximum_distance) + 1;
    if ((long long)x < x0 || (long long)x >= x1) continue;
    const float dx = (float)x - s.center_x, dy = (float)y - s.center_y;
    const float sqd = fmaf(dx, dx, dy * dy);
    const float distance = sqrtf(sqd);
    const float f = FastErffD(fmaf(distance, 0.5f, 0.353553391f) * s.inv_sigma) - FastErffD(fmaf(distance, 0.5f, -0.353553391f) * s.inv_sigma);
    const float local_intensity = s.sigma_over_4_times_intensity * (f * f);
    v0 = fmaf(s.color[0], local_intensity, v0);
    v1 = fmaf(s.color[1], local_intensity, v1);
    v2 = fmaf(s.color[2], local_intensity, v2);
    touched = true;
  }
  if (touched) { p0[o] = v0; p1[o] = v1; p2[o] = v2; }
}

// ---- upsampling of one plane (stage_upsampling.cc; same definition as kernels.hip UpsampleKernel) ---------------------------
__global__ void UpsamplePlaneKernel(const float* __restrict__ src, uint32_t src_stride, int w, int h, float* __restrict__ dst, uint32_t dst_stride, int ow, int oh,
                                    int up, const float* __restrict__ weights) {
  const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y * blockDim.y + threadIdx.y;
  if (ox >= ow || oy >= oh) return;
  const int N = up / 2;
  const int x = ox / up, sx = ox % up, y = oy / up, sy = oy % up;
  const int ky = sy < N ? sy : up - 1 - sy, kx = sx < N ? sx : up - 1 - sx;
  const bool fy = sy >= N, fx = sx >= N;
  float sum = 0.0f, mn = 0.0f, mx = 0.0f;
  for (int iy = 0; iy < 5; iy++) {
    const int yy = MirrorF(y + iy - 2, h);
    const int mi = 5 * ky + (fy ? 4 - iy : iy);
    for (int ix = 0; ix < 5; ix++) {
      const int xx = MirrorF(x + ix - 2, w);
      const float v = src[(size_t)yy * src_stride + xx];
      const int mj = 5 * kx + (fx ? 4 - ix : ix);
      const int lo = mi < mj ? mi : mj, hi = mi < mj ? mj : mi;
      const float k = weights[5 * N * lo - lo * (lo - 1) / 2 + hi - lo];
      sum = fmaf(k, v, sum);
      if (iy == 0 && ix == 0) { mn = v; mx = v; } else { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
    }
  }
  dst[(size_t)oy * dst_stride + ox] = sum < mn ? mn : (sum > mx ? mx : sum);
}

// ---- noise (dec_noise.cc Random3Planes, stage_noise.cc) --------------------------------------------------------------------
__device__ __forceinline__ uint64_t SplitMix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// One workgroup per 256x256 group (upsampled coordinates), one lane per Xorshift128+ generator (8 independent ones, base/random.h):
// lane i produces floats 2i and 2i+1 of every 16-float batch; a row of xs samples takes ceil(xs / 16) batches, three planes in turn.
__global__ __launch_bounds__(64) void NoiseRandomKernel(float* n0, float* n1, float* n2, uint32_t stride, uint32_t w, uint32_t h, uint32_t group_dim,
                                                        uint32_t visible_frame_index, uint32_t nonvisible_frame_index) {
  const uint32_t i = threadIdx.x;
  if (i >= 8) return;
  const uint32_t gx0 = blockIdx.x * group_dim, gy0 = blockIdx.y * group_dim;
  if (gx0 >= w || gy0 >= h) return;
  uint64_t s0 = SplitMix64((((uint64_t)visible_frame_index << 32) + nonvisible_frame_index) + 0x9E3779B97F4A7C15ull);
  uint64_t s1 = SplitMix64((((uint64_t)gx0 << 32) + gy0) + 0x9E3779B97F4A7C15ull);
  for (uint32_t k = 0; k < i; k++) { s0 = SplitMix64(s0); s1 = SplitMix64(s1); }
  const uint32_t xs = min(group_dim, w - gx0), ys = min(group_dim, h - gy0);
  float* planes[3] = {n0, n1, n2};
  for (int c = 0; c < 3; c++) {
    for (uint32_t y = 0; y < ys; y++) {
      float* row = planes[c] + (size_t)(gy0 + y) * stride + gx0;
      for (uint32_t x = 0; x < xs; x += 16) {
        uint64_t a = s0; const uint64_t b = s1;
        const uint64_t bits = a + b;
        s0 = b;
        a ^= a << 23;
        a ^= b ^ (a >> 18) ^ (b >> 5);
        s1 = a;
        const uint32_t lo = (uint32_t)bits, hi = (uint32_t)(bits >> 32);
        if (x + 2 * i < xs) row[x + 2 * i] = __uint_as_float((lo >> 9) | 0x3F800000u);
        if (x + 2 * i + 1 < xs) row[x + 2 * i + 1] = __uint_as_float((hi >> 9) | 0x3F800000u);
      }
    }
  }
}
__device__ __forceinline__ float NoiseStrength(const float* lut, float vx) {
  const float kScale = 6.0f;
  const float scaled = fmaxf(0.0f, vx * kScale);
  float floor_x = floorf(scaled), frac = scaled - floor_x;
  if (scaled >= kScale + 1) { floor_x = kScale; frac = 1.0f; }
  const int i = (int)floor_x;
  const float low = lut[i], hi = lut[i + 1];
  const float v = fmaf(hi - low, frac, low);
  return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
}
__global__ void NoiseAddKernel(NoiseArgs a) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  const int w = (int)a.w, h = (int)a.h;
  if (x >= w || y >= h) return;
  float conv[3];
  int xs[5], ys[5];
  for (int i = 0; i < 5; i++) { xs[i] = MirrorF(x + i - 2, w); ys[i] = MirrorF(y + i - 2, h); }
  for (int c = 0; c < 3; c++) {
    const float* n = a.noise[c];
    auto px = [&](int r, int dx) { return n[(size_t)ys[r] * a.noise_stride + xs[dx + 2]]; };
    const float p00 = px(2, 0);
    float others = 0.0f;
    for (int i = -2; i <= 2; i++) { others += px(0, i); others += px(1, i); others += px(3, i); others += px(4, i); }
    others += px(2, -2); others += px(2, -1); others += px(2, 1); others += px(2, 2);
    conv[c] = fmaf(others, 0.16f, p00 * -3.84f);
  }
  const float kNorm = 0.22f, kRGCorr = 0.9921875f, kRGNCorr = 0.0078125f;
  const size_t o = (size_t)y * a.stride + x;
  const float vx = a.p[0][o], vy = a.p[1][o];
  const float in_g = vy - vx, in_r = vy + vx;
  const float sg = NoiseStrength(a.lut, in_g * 0.5f), sr = NoiseStrength(a.lut, in_r * 0.5f);
  const float ar = conv[0] * kNorm, ag = conv[1] * kNorm, ac = conv[2] * kNorm;
  const float red = sr * fmaf(kRGNCorr, ar, kRGCorr * ac);
  const float green = sg * fmaf(kRGNCorr, ag, kRGCorr * ac);
  const float rg = red + green;
  a.p[0][o] = fmaf(a.ytox, rg, red - green) + vx;
  a.p[1][o] = vy + rg;
  a.p[2][o] = fmaf(a.ytob, rg, a.p[2][o]);
}

// ---- colour transform to the output space (stage_xyb.cc, stage_from_linear.cc, stage_ycbcr.cc) -------------------------------
__device__ __forceinline__ float TransferD(uint32_t kind, float inverse_gamma, float v) {
  switch (kind) {
    case 0: return LinearToSrgbF(v);
    case 1: return v;
    case 2: return v <= 1e-5f ? 0.0f : FastPowfD(v, inverse_gamma);
    case 4: return PqFromLinear(v, inverse_gamma);          // (inverse_gamma carries intensity_target / 10000 here)
    case 5: return HlgFromLinear(v);
    default: return v <= 0.018f ? 4.5f * v : fmaf(1.099f, FastPowfD(v, 0.45f), -0.099f);
  }
}
__global__ void ColorKernel(ColorArgs a) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= a.w || y >= a.h) return;
  const size_t si = (size_t)y * a.src_stride + x, di = (size_t)y * a.dst_stride + x;
  const float X = a.src[0][si], Y = a.src[1][si], B = a.src[2][si];
  float r, g, b;
  if (a.mode == 0) {          // XYB -> linear -> transfer function
    const float gr = (Y + X) - a.neg_bias_cbrt[0];
    const float gg = (Y - X) - a.neg_bias_cbrt[1];
    const float gb = B - a.neg_bias_cbrt[2];
    const float mr = fmaf(gr * gr, gr, a.neg_bias[0]);
    const float mg = fmaf(gg * gg, gg, a.neg_bias[1]);
    const float mb = fmaf(gb * gb, gb, a.neg_bias[2]);
    r = fmaf(a.opsin_inv[2], mb, fmaf(a.opsin_inv[1], mg, a.opsin_inv[0] * mr));
    g = fmaf(a.opsin_inv[5], mb, fmaf(a.opsin_inv[4], mg, a.opsin_inv[3] * mr));
    b = fmaf(a.opsin_inv[8], mb, fmaf(a.opsin_inv[7], mg, a.opsin_inv[6] * mr));
    if (a.tf_kind == 5) HlgInverseOotf(a.hdr_par, r, g, b, [](float x, float e) { return FastPowfD(x, e); });
    const float tf_par = a.tf_kind == 4 ? a.hdr_par[0] : a.inverse_gamma;
    r = TransferD(a.tf_kind, tf_par, r); g = TransferD(a.tf_kind, tf_par, g); b = TransferD(a.tf_kind, tf_par, b);
  } else if (a.mode == 1) {   // YCbCr (planes Cb, Y, Cr) -> RGB
    const float c128 = 128.0f / 255, crcr = 1.402f, cgcb = -0.114f * 1.772f / 0.587f, cgcr = -0.299f * 1.402f / 0.587f, cbcb = 1.772f;
    const float yb = Y + c128;
    r = fmaf(crcr, B, yb);
    g = fmaf(cgcr, B, fmaf(cgcb, X, yb));
    b = fmaf(cbcb, X, yb);
  } else if (a.mode == 3) {   // transfer function only: the planes hold linear light (a spot-colour stage came in between)
    r = X; g = Y; b = B;
    if (a.tf_kind == 5) HlgInverseOotf(a.hdr_par, r, g, b, [](float x, float e) { return FastPowfD(x, e); });
    const float tf_par = a.tf_kind == 4 ? a.hdr_par[0] : a.inverse_gamma;
    r = TransferD(a.tf_kind, tf_par, r); g = TransferD(a.tf_kind, tf_par, g); b = TransferD(a.tf_kind, tf_par, b);
  } else { r = X; g = Y; b = B; }
  a.dst[0][di] = r; a.dst[1][di] = g; a.dst[2][di] = b;
}

__global__ void SpotKernel(SpotArgs a) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= a.w || y >= a.h) return;
  const float mix = a.scale * a.spot[(size_t)y * a.spot_stride + x];
  const size_t o = (size_t)y * a.stride + x;
#pragma unroll
  for (int c = 0; c < 3; c++) a.p[c][o] = mix * a.color[c] + (1.0f - mix) * a.p[c][o];
}

// ---- chroma upsampling of subsampled YCbCr frames (stage_chroma_upsampling.cc): horizontal, then vertical, each with the (1/4, 3/4)
// kernel — out[2x] = 0.25 in[x-1] + 0.75 in[x], out[2x+1] = 0.25 in[x+1] + 0.75 in[x] — neighbours clamped at the channel's own edges.
// One thread per output sample; the vertical step works on horizontally upsampled rows, exactly as two stages would.
struct ChromaUpArgs { const float* src; float* dst; uint32_t src_stride, dst_stride, cw, ch, hs, vs, out_w, out_h; };
__global__ void ChromaUpsampleKernel(ChromaUpArgs a) {
  const uint32_t X = blockIdx.x * blockDim.x + threadIdx.x, Y = blockIdx.y * blockDim.y + threadIdx.y;
  if (X >= a.out_w || Y >= a.out_h) return;
  const uint32_t sx = X >> a.hs, sy = Y >> a.vs;
  if (sx >= a.cw || sy >= a.ch) return;
  auto hval = [&](uint32_t row) -> float {
    const float* in = a.src + (size_t)row * a.src_stride;
    if (!a.hs) return in[X];
    const float mid = in[sx] * 0.75f;
    const uint32_t nb = (X & 1) ? min(sx + 1, a.cw - 1) : (sx ? sx - 1 : 0);
    return fmaf(0.25f, in[nb], mid);
  };
  float v;
  if (!a.vs) v = hval(sy);
  else {
    const float mid = hval(sy) * 0.75f;
    const uint32_t nb = (Y & 1) ? min(sy + 1, a.ch - 1) : (sy ? sy - 1 : 0);
    v = fmaf(0.25f, hval(nb), mid);
  }
  a.dst[(size_t)Y * a.dst_stride + X] = v;
}

// ---- blending of a frame onto the image canvas (stage_blending.cc; blending.cc) ----------------------------------------------
__device__ __forceinline__ float FrameBlendSampleD(uint32_t mode, bool clamp, bool premultiplied, float bg, float fg, float bga, float fga) {
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
__global__ void BlendKernel(BlendArgs a) {
  const int X = blockIdx.x * blockDim.x + threadIdx.x, Y = blockIdx.y * blockDim.y + threadIdx.y;
  if (X >= (int)a.img_w || Y >= (int)a.img_h) return;
  const size_t co = (size_t)Y * a.canvas_stride + X, ceo = (size_t)Y * a.canvas_ec_stride + X;
  const size_t bo = (size_t)Y * a.bg_stride + X;
  const int fx = X - a.x0, fy = Y - a.y0;
  const bool inside = fx >= 0 && fy >= 0 && fx < (int)a.fw && fy < (int)a.fh;
  if (!inside) {
    for (int c = 0; c < 3; c++) a.canvas[c][co] = a.bg[0] ? a.bg[c][bo] : 0.0f;
    for (uint32_t e = 0; e < a.num_extra; e++) a.canvas_ec[e][ceo] = a.bg_ec[e] ? a.bg_ec[e][(size_t)Y * a.bg_ec_stride[e] + X] : 0.0f;
    return;
  }
  const size_t fo = (size_t)fy * a.fg_stride + fx, feo = (size_t)fy * a.fg_ec_stride + fx;
  const uint32_t mode = a.mode[0] & 0xFF, ach = (a.mode[0] >> 8) & 0xFF; const bool clamp = (a.mode[0] >> 16) & 1;
  float fga = 1.0f, bga = 1.0f; bool premul = false;
  if (mode == 2 || mode == 3) {
    fga = a.fg_ec[ach][feo];
    bga = a.bg_alpha ? a.bg_alpha[(size_t)Y * a.bg_alpha_stride + X] : 0.0f;
    premul = (a.premul_mask >> ach) & 1;
  }
  float ec_out[4];
  for (uint32_t e = 0; e < a.num_extra; e++) {
    const uint32_t m = a.mode[1 + e] & 0xFF, ac = (a.mode[1 + e] >> 8) & 0xFF; const bool cl = (a.mode[1 + e] >> 16) & 1;
    const float b = a.bg_ec[e] ? a.bg_ec[e][(size_t)Y * a.bg_ec_stride[e] + X] : 0.0f;
    const float fv = a.fg_ec[e][feo];
    if (m == 2 || m == 3) {
      const float efga = a.fg_ec[ac][feo];
      const float ebga = a.bg_ec_alpha[e] ? a.bg_ec_alpha[e][(size_t)Y * a.bg_ec_stride[e] + X] : 0.0f;
      if (ac == e) { const float fa = cl ? Clamp01(efga) : efga; ec_out[e] = m == 2 ? 1.0f - (1.0f - fa) * (1.0f - ebga) : ebga; }
      else ec_out[e] = FrameBlendSampleD(m, cl, (a.premul_mask >> ac) & 1, b, fv, ebga, efga);
    } else ec_out[e] = FrameBlendSampleD(m, cl, false, b, fv, 1.0f, 1.0f);
  }
  for (int c = 0; c < 3; c++) {
    const float b = a.bg[0] ? a.bg[c][bo] : 0.0f;
    a.canvas[c][co] = FrameBlendSampleD(mode, clamp, premul, b, a.fg[c][fo], bga, fga);
  }
  for (uint32_t e = 0; e < a.num_extra; e++) a.canvas_ec[e][ceo] = ec_out[e];
}

// ---- write stage (stage_write.cc): float planes in the output colour space -> caller layout --------------------------------
__device__ __forceinline__ uint16_t HalfBits(float fv) {
  const uint32_t x = __float_as_uint(fv);
  const uint32_t sign = (x >> 16) & 0x8000;
  const int32_t exp = (int32_t)((x >> 23) & 0xFF) - 127 + 15;
  uint32_t mant = x & 0x7FFFFF;
  if (((x >> 23) & 0xFF) == 0xFF) return (uint16_t)(sign | 0x7C00 | (mant ? 0x200 : 0));
  if (exp >= 31) return (uint16_t)(sign | 0x7C00);
  if (exp <= 0) {
    if (exp < -10) return (uint16_t)sign;
    mant |= 0x800000;
    const int shift = 14 - exp;
    uint32_t m = mant >> shift;
    const uint32_t rem = mant & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (m & 1))) m++;
    return (uint16_t)(sign | m);
  }
  const uint32_t m = mant >> 13, rem = mant & 0x1FFF;
  uint32_t r = (uint32_t)(exp << 10) | m;
  if (rem > 0x1000 || (rem == 0x1000 && (m & 1))) r++;
  return (uint16_t)(sign | r);
}
__device__ __forceinline__ void StoreSampleW(const WriteArgs& a, uint8_t* p, float v) {
  if (a.out_type == 0) p[0] = (uint8_t)__float2int_rn(fminf(1.0f, fmaxf(0.0f, v)) * a.out_int_mul);
  else if (a.out_type == 1) {
    const uint32_t u = (uint32_t)__float2int_rn(fminf(1.0f, fmaxf(0.0f, v)) * a.out_int_mul);
    if (a.out_big_endian) { p[0] = (uint8_t)(u >> 8); p[1] = (uint8_t)u; } else { p[0] = (uint8_t)u; p[1] = (uint8_t)(u >> 8); }
  } else if (a.out_type == 2) {
    const uint32_t u = __float_as_uint(v);
    if (a.out_big_endian) { p[0] = (uint8_t)(u >> 24); p[1] = (uint8_t)(u >> 16); p[2] = (uint8_t)(u >> 8); p[3] = (uint8_t)u; }
    else { p[0] = (uint8_t)u; p[1] = (uint8_t)(u >> 8); p[2] = (uint8_t)(u >> 16); p[3] = (uint8_t)(u >> 24); }
  } else {
    const uint32_t u = HalfBits(v);
    if (a.out_big_endian) { p[0] = (uint8_t)(u >> 8); p[1] = (uint8_t)u; } else { p[0] = (uint8_t)u; p[1] = (uint8_t)(u >> 8); }
  }
}
__global__ void WriteKernel(WriteArgs a) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  const int w = (int)a.img_w, h = (int)a.img_h;
  if (x >= w || y >= h) return;
  const size_t o = (size_t)y * a.stride + x;
  float r = a.p[0][o], g = a.p[1][o], b = a.p[2][o];
  float al = 1.0f;
  if (a.alpha) {
    al = a.alpha[(size_t)y * a.alpha_stride + x];
    if (a.unpremul) {   // alpha.cc UnpremultiplyAlpha: colour / max(alpha, 2^-26)
      const float m = 1.0f / fmaxf(1.0f / (float)(1u << 26), al);
      r *= m; g *= m; b *= m;
    }
  }
  int ox = x, oy = y;
  switch (a.out_orient) {
    case 2: ox = w - 1 - x; break;
    case 3: ox = w - 1 - x; oy = h - 1 - y; break;
    case 4: oy = h - 1 - y; break;
    case 5: ox = y; oy = x; break;
    case 6: ox = h - 1 - y; oy = x; break;
    case 7: ox = h - 1 - y; oy = w - 1 - x; break;
    case 8: ox = y; oy = w - 1 - x; break;
    default: break;
  }
  const uint32_t bps = a.out_type == 0 ? 1 : a.out_type == 2 ? 4 : 2;
  uint8_t* p = a.out + (size_t)oy * a.out_stride + (size_t)ox * a.out_channels * bps;
  const uint32_t nc = a.out_channels;
  if (nc <= 2) { StoreSampleW(a, p, a.is_gray ? r : g); if (nc == 2) StoreSampleW(a, p + bps, al); }
  else { StoreSampleW(a, p, r); StoreSampleW(a, p + bps, g); StoreSampleW(a, p + 2 * bps, b); if (nc == 4) StoreSampleW(a, p + 3 * bps, al); }
}

__global__ void CopyPlaneKernel(const float* __restrict__ src, uint32_t src_stride, float* __restrict__ dst, uint32_t dst_stride, uint32_t w, uint32_t h) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  dst[(size_t)y * dst_stride + x] = src[(size_t)y * src_stride + x];
}

// ---- JPEG reconstruction: coefficients back into JPEG layout --------------------------------------------------------------------------
__global__ void JpegCoefKernel(const FrameDev* __restrict__ frames, int fidx, JpegCoefArgs a) {
  const FrameDev& f = frames[fidx];
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;       // (block, natural coefficient index)
  const uint32_t nblk = f.bw * f.bh;
  if (t >= nblk * 64 || *f.status != 0) return;      // (a frame whose entropy stages failed has no usable coefficient offsets)
  const uint32_t o = t >> 6, i = t & 63, v = i >> 3, u = i & 7;
  const uint32_t bx = o % f.bw, by = o / f.bw;
  const uint32_t g = (by / 32) * f.xgroups + bx / 32;
  const size_t base = (size_t)g * 65536 + f.coef_off[o] + (u * 8 + v);   // libjxl stores the transpose of JPEG's (v, u) layout
  if (f.subsampled) {
    // chroma-subsampled frames carry no chroma-from-luma (the LF stage rejects it); a channel's block (sx, sy) keeps its coefficients at
    // the slot of block (sx << hs, sy << vs) and its LF sample at (sx, sy) of the padded grid; component planes are packed one after another
    for (uint32_t c = 0; c < a.ncomp; c++) {
      const int ch = c == 0 ? 1 : c == 1 ? 0 : 2;
      const uint32_t hs = f.hs[ch], vs = f.vs[ch];
      if ((bx & ((1u << hs) - 1u)) | (by & ((1u << vs) - 1u))) continue;
      const uint32_t sx = bx >> hs, sy = by >> vs, cw = f.bw >> hs;
      const int32_t val = i == 0 ? f.lfq[ch][(size_t)sy * f.bw + sx] : f.coeff[ch][base];
      a.out[((size_t)a.comp_off[c] + (size_t)sy * cw + sx) * 64 + i] = (int16_t)val;
    }
    return;
  }
  const int32_t y = i == 0 ? f.lfq[1][o] : f.coeff[1][base];
  if (a.ncomp == 1) { a.out[(size_t)o * 64 + i] = (int16_t)y; return; }
  a.out[(size_t)o * 64 + i] = (int16_t)y;
  const size_t tile = (size_t)(by / 8) * f.cw + bx / 8;
  for (int c = 1; c < 3; c++) {
    const int ch = c == 1 ? 0 : 2;                                    // Cb rides in the X slot, Cr in the B slot
    int32_t val;
    if (i == 0) val = f.lfq[ch][o];
    else {
      const int32_t fac = c == 1 ? (int32_t)f.ytox[tile] : (int32_t)f.ytob[tile];
      const int32_t ff = fac * 2048 / 84;                             // (C++ division: truncates toward zero)
      const int32_t scale = (2048 * a.qt[0][i] / a.qt[c][i]) * ff;
      const int32_t cfl = (y * ((scale + 1024) >> 11) + 1024) >> 11;
      val = f.coeff[ch][base] + cfl;
    }
    a.out[((size_t)c * nblk + o) * 64 + i] = (int16_t)val;
  }
}

inline dim3 Grid2(uint32_t w, uint32_t h) { return dim3((w + 31) / 32, (h + 7) / 8); }
const dim3 kBlock2(32, 8);

}  // namespace

void LaunchIntToFloat(const int32_t* src, uint32_t src_stride, float* dst, uint32_t dst_stride, uint32_t w, uint32_t h, float factor, void* stream, uint32_t float_bits,
                      uint32_t float_exp_bits) {
  hipLaunchKernelGGL(IntToFloatKernel, Grid2(w, h), kBlock2, 0, (hipStream_t)stream, src, src_stride, dst, dst_stride, w, h, factor, float_bits, float_exp_bits);
}
void LaunchXybModToFloat(const int32_t* cy, const int32_t* cx, const int32_t* cb, uint32_t src_stride, float* const dst[3], uint32_t dst_stride, uint32_t w, uint32_t h,
                         const float fac[3], void* stream) {
  hipLaunchKernelGGL(XybModToFloatKernel, Grid2(w, h), kBlock2, 0, (hipStream_t)stream, cy, cx, cb, src_stride, dst[0], dst[1], dst[2], dst_stride, w, h, fac[0], fac[1], fac[2]);
}
void LaunchPatches(const PatchFrameArgs& a, const PatchEntryDev* entries, const uint32_t* tile_start, const uint32_t* tile_list, void* stream) {
  const uint32_t tiles_x = (a.w + 31) / 32, tiles_y = (a.h + 31) / 32;
  hipLaunchKernelGGL(PatchKernel, dim3(tiles_x * tiles_y), dim3(256), 0, (hipStream_t)stream, a, entries, tile_start, tile_list, tiles_x);
}
void LaunchSplines(float* const p[3], uint32_t stride, uint32_t w, uint32_t h, const SplineSegmentDev* segs, const uint32_t* row_start, const uint32_t* indices, void* stream) {
  hipLaunchKernelGGL(SplineKernel, dim3((w + 255) / 256, h), dim3(256), 0, (hipStream_t)stream, p[0], p[1], p[2], stride, w, h, segs, row_start, indices);
}
void LaunchUpsamplePlane(const float* src, uint32_t src_stride, uint32_t w, uint32_t h, float* dst, uint32_t dst_stride, uint32_t ow, uint32_t oh, uint32_t up,
                         const float* weights, void* stream) {
  hipLaunchKernelGGL(UpsamplePlaneKernel, Grid2(ow, oh), kBlock2, 0, (hipStream_t)stream, src, src_stride, (int)w, (int)h, dst, dst_stride, (int)ow, (int)oh, (int)up, weights);
}
void LaunchNoise(const NoiseArgs& a, void* stream) {
  const uint32_t gx = (a.w + a.group_dim - 1) / a.group_dim, gy = (a.h + a.group_dim - 1) / a.group_dim;
  hipLaunchKernelGGL(NoiseRandomKernel, dim3(gx, gy), dim3(64), 0, (hipStream_t)stream, a.noise[0], a.noise[1], a.noise[2], a.noise_stride, a.w, a.h, a.group_dim,
                     a.visible_frame_index, a.nonvisible_frame_index);
  hipLaunchKernelGGL(NoiseAddKernel, Grid2(a.w, a.h), kBlock2, 0, (hipStream_t)stream, a);
}
void LaunchSpot(const SpotArgs& a, void* stream) { hipLaunchKernelGGL(SpotKernel, Grid2(a.w, a.h), kBlock2, 0, (hipStream_t)stream, a); }
void LaunchColor(const ColorArgs& a, void* stream) { hipLaunchKernelGGL(ColorKernel, Grid2(a.w, a.h), kBlock2, 0, (hipStream_t)stream, a); }
void LaunchBlend(const BlendArgs& a, void* stream) { hipLaunchKernelGGL(BlendKernel, Grid2(a.img_w, a.img_h), kBlock2, 0, (hipStream_t)stream, a); }
void LaunchWrite(const WriteArgs& a, void* stream) { hipLaunchKernelGGL(WriteKernel, Grid2(a.img_w, a.img_h), kBlock2, 0, (hipStream_t)stream, a); }
void LaunchJpegCoefficients(const FrameDev* frames, int fidx, const JpegCoefArgs& a, uint32_t bw, uint32_t bh, void* stream) {
  const uint32_t n = bw * bh * 64;
  hipLaunchKernelGGL(JpegCoefKernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, frames, fidx, a);
}
void LaunchCopyPlane(const float* src, uint32_t src_stride, float* dst, uint32_t dst_stride, uint32_t w, uint32_t h, void* stream) {
  hipLaunchKernelGGL(CopyPlaneKernel, Grid2(w, h), kBlock2, 0, (hipStream_t)stream, src, src_stride, dst, dst_stride, w, h);
}

void LaunchChromaUpsample(const float* src, uint32_t src_stride, float* dst, uint32_t dst_stride, uint32_t cw, uint32_t ch, uint32_t hs, uint32_t vs, uint32_t out_w, uint32_t out_h, void* stream) {
  ChromaUpArgs a{src, dst, src_stride, dst_stride, cw, ch, hs, vs, out_w, out_h};
  hipLaunchKernelGGL(ChromaUpsampleKernel, Grid2(out_w, out_h), kBlock2, 0, (hipStream_t)stream, a);
}

}  // namespace jxlhip
