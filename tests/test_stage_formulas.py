"""GPU tests (-m gpu) that pin single float stages of the VarDCT pipeline against float64 restatements of their DEFINING FORMULAS, written
here in numpy and sharing no code with oracle/ (which every other parity test compares with): gaborish (3x3 normalised kernel, mirrored
borders: libjxl stage_gaborish.cc / loop_filter.h), the edge-preserving filter passes 0 / 1 / 2 (epf.cc: per-block sigma from the quantiser
field and the sharpness map, weights 1 + SAD * inv_sigma * scale clamped at 0, plus-shaped SAD patches, border rule), the inverse opsin
transform (stage_xyb.cc: subtract cbrt(bias), cube, add bias, 3x3 matrix) and the sRGB transfer function (stage_from_linear.cc against the
IEC 61966-2-1 formula).  The planes between the stages come off the device through JxlHipBatchSetOption("debug_stop_after") +
JxlHipBatchDebugRead (include/jxl_hip.h); inputs of a stage are what the stage before left, so each test isolates one stage.
Round 4 adds: the six 8x8 transforms that are not a plain DCT (IDENTITY, DCT2X2, DCT4X4, DCT4X8, DCT8X4, AFV0-3: dec_transforms-inl.h), the
DCT128 / 256 family against scipy's idctn, the lowest frequencies of multi-block transforms from the LF samples (the standard's product-of-
cosines scale), the adaptive LF smoothing (compressed_dc.cc), the 2x / 4x / 8x upsampling from the stored weights (stage_upsampling.cc), frame and
patch blending (blending.cc), noise synthesis (dec_noise.cc, stage_noise.cc) and spline rendering (splines.cc).
Tolerances: a few float32 ULPs of the magnitude of the values that enter a sum (stated per test)."""
import numpy as np
import pytest

import synth_lib as S

pytestmark = pytest.mark.gpu
EPS = float(np.finfo(np.float32).eps)


@pytest.fixture(scope="module")
def jx(built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import jpegxl_rs_amd as jx
    return jx


def planes_after(jx, data, stop, gab, epf, dtype="float32"):
    """XYB planes (3, h, w) after stage `stop` (1 IDCT, 2 gaborish, 3 / 4 / 5 EPF pass 0 / 1 / 2; 0: whole decode -> (pixels, None)) + the batch"""
    b = jx.BatchDecoder(0)
    b.add(data, dtype, 3)
    b.set_option("force_unfused_filters", 1)
    b.prepare()
    b.set_option("debug_stop_after", stop)
    b.decode(); b.finish()
    info = b.info(0)
    w, h, bw, bh = info.xsize, info.ysize, b.info_value("frame0_bw"), b.info_value("frame0_bh")
    if stop == 0:
        return b.output(0).reshape(h, w, 3), b
    executed = (1 if gab and stop >= 2 else 0) + (1 if epf >= 3 and stop >= 3 else 0) + (1 if epf >= 1 and stop >= 4 else 0) + (1 if epf >= 2 and stop >= 5 else 0)
    name = "plane_b" if executed % 2 else "plane_a"
    return np.stack([b.debug_read(0, name, c).reshape(bh * 8, bw * 8)[:h, :w] for c in range(3)]).astype(np.float64), b


def shifted(p, dx, dy, r):
    """p[y + dy, x + dx] with libjxl's mirroring at the image borders (Mirror(): -1 -> 0, size -> size - 1), p padded by r"""
    h, w = p.shape[0] - 2 * r, p.shape[1] - 2 * r
    return p[r + dy:r + dy + h, r + dx:r + dx + w]


@pytest.mark.parametrize("w,h,seed", [(200, 136, 3), (77, 61, 4)])
def test_gaborish_is_the_normalised_3x3_kernel(jx, w, h, seed):
    data = S.encode_vardct(S.synthetic_image(seed, w, h), seed=seed, strategy_mix=2, epf_iters=0, gab=1)
    before, _ = planes_after(jx, data, 1, 1, 0)
    after, _ = planes_after(jx, data, 2, 1, 0)
    w1, w2 = float(np.float32(0.115169525)), float(np.float32(0.061248592))      # loop_filter.h defaults (the synthesiser writes no custom weights)
    for c in range(3):
        p = np.pad(before[c], 1, mode="symmetric")
        edge = shifted(p, -1, 0, 1) + shifted(p, 1, 0, 1) + shifted(p, 0, -1, 1) + shifted(p, 0, 1, 1)
        diag = shifted(p, -1, -1, 1) + shifted(p, 1, -1, 1) + shifted(p, -1, 1, 1) + shifted(p, 1, 1, 1)
        want = (before[c] + w1 * edge + w2 * diag) / (1 + 4 * (w1 + w2))
        tol = 8 * EPS * max(1e-3, float(np.abs(before[c]).max()))            # nine terms of at most that magnitude, float32 accumulation
        assert np.abs(after[c] - want).max() <= tol, (c, float(np.abs(after[c] - want).max()), tol)


def epf_pass_float64(src, inv_sigma_px, pass_index):
    """epf.cc, one pass over planes src (3, h, w): out = (p + sum_t w_t p_t) / (1 + sum_t w_t), w_t = max(0, 1 + SAD_t * inv_sigma * scale)"""
    h, w = src.shape[1:]
    channel_scale = (40.0, 5.0, 3.5)
    sigma_scale = (0.9, 1.0, 6.5)[pass_index] * 1.65
    border_mul = 2.0 / 3.0
    yy, xx = np.mgrid[0:h, 0:w]
    border = (xx % 8 == 0) | (xx % 8 == 7) | (yy % 8 == 0) | (yy % 8 == 7)
    vmul = inv_sigma_px * sigma_scale * np.where(border, border_mul, 1.0)
    r = 3
    pad = [np.pad(src[c], r, mode="symmetric") for c in range(3)]
    taps = [(0, -2), (-1, -1), (0, -1), (1, -1), (-2, 0), (-1, 0), (1, 0), (2, 0), (-1, 1), (0, 1), (1, 1), (0, 2)] if pass_index == 0 else [(0, -1), (-1, 0), (1, 0), (0, 1)]
    plus = [(0, 0), (0, -1), (-1, 0), (1, 0), (0, 1)] if pass_index < 2 else [(0, 0)]
    wsum = np.ones((h, w))
    acc = [src[c].copy() for c in range(3)]
    for dx, dy in taps:
        sad = np.zeros((h, w))
        for c in range(3):
            s = np.zeros((h, w))
            for px, py in plus:
                s += np.abs(shifted(pad[c], dx + px, dy + py, r) - shifted(pad[c], px, py, r))
            sad += s * channel_scale[c]
        wgt = np.maximum(0.0, 1.0 + sad * vmul)
        wsum += wgt
        for c in range(3):
            acc[c] += wgt * shifted(pad[c], dx, dy, r)
    out = np.stack([acc[c] / wsum for c in range(3)])
    skip = inv_sigma_px < -3.90524291751269967465540850526868       # sigma below kMinSigma: the pixel passes through
    return np.where(skip[None], src, out)


@pytest.mark.parametrize("gab,epf,w,h,seed", [(0, 1, 200, 136, 5), (1, 2, 133, 90, 6), (1, 3, 200, 136, 7), (0, 3, 64, 72, 8)])
def test_epf_passes_and_sigma_follow_their_definition(jx, gab, epf, w, h, seed):
    data = S.encode_vardct(S.synthetic_image(seed, w, h), seed=seed, strategy_mix=1, epf_iters=epf, gab=gab)
    passes = {1: [1], 2: [1, 2], 3: [0, 1, 2]}[epf]
    stop_of = {0: 3, 1: 4, 2: 5}
    prev, b = planes_after(jx, data, 2 if gab else 1, gab, epf)
    bw, bh = b.info_value("frame0_bw"), b.info_value("frame0_bh")
    inv_sigma = b.debug_read(0, "inv_sigma").reshape(bh, bw).astype(np.float64)
    info = b.debug_read(0, "blk_info", dtype=np.uint32).reshape(bh, bw)
    # ---- sigma (epf.cc ComputeSigma): quant_mul / (quantiser scale * hf_mul * -1.1715728752538099) * sharpness LUT, capped at -1e-4
    hf_mul = ((info >> 8) & 0xFF).astype(np.float64) + 1
    sharp = ((info >> 26) & 7).astype(np.float64)
    quant_scale = b.info_value("frame0_global_scale") / 65536.0
    sigma = np.minimum(-1e-4, 0.46 / (quant_scale * hf_mul * -1.1715728752538099024) * (sharp / 7.0))
    blocks_y, blocks_x = (h + 7) // 8, (w + 7) // 8
    got, want = inv_sigma[:blocks_y, :blocks_x], (1.0 / sigma)[:blocks_y, :blocks_x]
    assert np.abs(got - want).max() <= 4 * EPS * np.abs(want).max() or np.allclose(got, want, rtol=4 * EPS, atol=0), "per-block 1 / sigma"
    inv_sigma_px = np.kron(inv_sigma, np.ones((8, 8)))[:h, :w]
    for p in passes:
        cur, _ = planes_after(jx, data, stop_of[p], gab, epf)
        want = epf_pass_float64(prev, inv_sigma_px, p)
        # weights are 1 + SAD * (a product of up to ~1e3): a float32 rounding of the SAD moves a weight by ~1e-4 relative at most; the
        # normalised sum then stays within a few 1e-6 of the magnitude of the planes
        scale = max(1e-3, float(np.abs(prev).max()))
        err = np.abs(cur - want).max()
        assert err <= 2e-5 * scale, (p, float(err), scale)
        prev = cur


def srgb_oetf(v):
    a = np.abs(v)
    return np.sign(v) * np.where(a <= 0.0031308, a * 12.92, 1.055 * np.power(a, 1 / 2.4) - 0.055)


@pytest.mark.parametrize("tf", ["linear", "srgb"])
def test_inverse_opsin_and_transfer_function(jx, tf):
    img = S.synthetic_image(9, 160, 120)
    if tf == "linear":
        S.set_color(1, 1, 8)
    try:
        data = S.encode_vardct(img, seed=9, strategy_mix=1, epf_iters=0, gab=0)
    finally:
        S.set_color()
    xyb, _ = planes_after(jx, data, 1, 0, 0)
    px, _ = planes_after(jx, data, 0, 0, 0)
    X, Y, B = xyb
    # stage_xyb.cc / opsin_params.h: mixed = (channel - cbrt(bias))^3 + bias on (Y + X, Y - X, B), then the inverse opsin absorbance matrix
    # (intensity target 255: no rescaling)
    bias = -0.0037930732552754493
    cb = np.cbrt(bias)
    mixed = [np.power(Y + X - cb, 3) + bias, np.power(Y - X - cb, 3) + bias, np.power(B - cb, 3) + bias]
    inv = np.array([[11.031566901960783, -9.866943921568629, -0.16462299647058826],
                    [-3.254147380392157, 4.418770392156863, -0.16462299647058826],
                    [-3.6588512862745097, 2.7129230470588235, 1.9459282392156863]])
    lin = np.stack([sum(inv[r, k] * mixed[k] for k in range(3)) for r in range(3)], axis=-1)
    want = lin if tf == "linear" else srgb_oetf(lin)
    got = px.astype(np.float64)
    # the matrix rows sum terms of up to ~11 x the mixed values with cancellation: float32 leaves ~1e-6 absolute on values in [0, 1];
    # libjxl's sRGB curve is a rational approximation good to ~1e-6 as well
    assert np.abs(got - want).max() <= (4e-6 if tf == "linear" else 1.5e-5), float(np.abs(got - want).max())


# ---- dequantisation + chroma from luma + inverse DCT -------------------------------------------------------------------------------------
# covered blocks (columns, rows) and quantisation-table kind of the plain DCT strategies (ac_strategy.h; IDENTITY / DCT2X2 / DCT4X4 / DCT4X8 /
# AFV are transforms of their own and the 128 / 256 family is left out here)
PLAIN = {0: (1, 1, 0), 4: (2, 2, 4), 5: (4, 4, 5), 6: (1, 2, 6), 7: (2, 1, 6), 8: (1, 4, 7), 9: (4, 1, 7), 10: (2, 4, 8), 11: (4, 2, 8), 18: (8, 8, 11), 19: (4, 8, 12), 20: (8, 4, 12)}


COVERED = {0: (1, 1), 1: (1, 1), 2: (1, 1), 3: (1, 1), 4: (2, 2), 5: (4, 4), 6: (1, 2), 7: (2, 1), 8: (1, 4), 9: (4, 1), 10: (2, 4), 11: (4, 2), 12: (1, 1), 13: (1, 1),
           14: (1, 1), 15: (1, 1), 16: (1, 1), 17: (1, 1), 18: (8, 8), 19: (4, 8), 20: (8, 4), 21: (16, 16), 22: (8, 16), 23: (16, 8), 24: (32, 32), 25: (16, 32), 26: (32, 16)}
QUANT_KIND = {0: 0, 1: 1, 2: 2, 3: 3, 4: 4, 5: 5, 6: 6, 7: 6, 8: 7, 9: 7, 10: 8, 11: 8, 12: 9, 13: 9, 14: 10, 15: 10, 16: 10, 17: 10, 18: 11, 19: 12, 20: 12,
              21: 13, 22: 14, 23: 14, 24: 15, 25: 16, 26: 16}


def dequantised_varblocks(jx, data, h, w):
    """Decodes `data` up to the IDCT stage and yields, per varblock that starts inside the image: (strategy, by, bx, [three flat float64 arrays of
    dequantised coefficients in STORAGE order, chroma from luma applied]), after the IDCT planes `got` and the batch.  Dequantisation as in
    dec_group.cc / quantizer.h: AdjustQuantBias(q) x table[k] x (65536 / global_scale) / hf_mul (x 0.8^(qm_scale - 2) for X, B), X and B plus
    (base + map / colour_factor) x Y.  Inputs off the device: quantised coefficients, block info, tables, CfL maps."""
    got, b = planes_after(jx, data, 1, 0, 0)
    bw, bh, xgroups = b.info_value("frame0_bw"), b.info_value("frame0_bh"), b.info_value("frame0_xgroups")
    info = b.debug_read(0, "blk_info", dtype=np.uint32).reshape(bh, bw)
    coef_off = b.debug_read(0, "coef_off", dtype=np.uint32).reshape(bh, bw)
    # (the IDCT zeroes the coefficient planes it consumes: decode up to the HF stage again to read them)
    b.decode_part(1); b.decode_part(3); b.finish()
    coeff = [b.debug_read(0, "coeff", c, dtype=np.int32).astype(np.float64) for c in range(3)]
    cw = (bw + 7) // 8
    ytox = b.debug_read(0, "ytox", dtype=np.int8).astype(np.float64)
    ytob = b.debug_read(0, "ytob", dtype=np.int8).astype(np.float64)
    inv_global_scale = 65536.0 / b.info_value("frame0_global_scale")
    colour_scale = 1.0 / b.info_value("frame0_color_factor")
    dm = [0.8 ** (b.info_value("frame0_x_qm_scale") - 2.0), 1.0, 0.8 ** (b.info_value("frame0_b_qm_scale") - 2.0)]
    bias = [1.0 - 0.05465007330715401, 1.0 - 0.07005449891748593, 1.0 - 0.049935103337343655, 0.145]
    tables = {}

    def blocks():
        for by in range(bh):
            for bx in range(bw):
                word = int(info[by, bx])
                if not (word >> 5) & 1 or by * 8 >= h or bx * 8 >= w:
                    continue                                     # not the first block of its varblock / outside the image
                s = word & 31
                cx, cy = COVERED[s]
                kind = QUANT_KIND[s]
                n = 64 * cx * cy
                hf_mul = ((word >> 8) & 0xFF) + 1
                g = (by // 32) * xgroups + bx // 32
                base = g * 65536 + int(coef_off[by, bx])
                tile = (by // 8) * cw + bx // 8
                k_cfl = [0.0 + ytox[tile] * colour_scale, 0.0, 1.0 + ytob[tile] * colour_scale]
                deq = []
                for c in range(3):
                    if (kind, c) not in tables:
                        tables[(kind, c)] = b.debug_read(0, "qtable", kind * 3 + c).astype(np.float64)
                    q = coeff[c][base:base + n]
                    safe = np.where(q == 0, 1.0, q)
                    adj = np.where(q == 0, 0.0, np.where(np.abs(q) == 1, np.sign(q) * bias[c], q - bias[3] / safe))
                    deq.append(adj * tables[(kind, c)][:n] * (inv_global_scale / hf_mul * dm[c]))
                yield s, by, bx, [deq[0] + k_cfl[0] * deq[1], deq[1], deq[2] + k_cfl[2] * deq[1]]
    return got, b, blocks()


def semantic(stored, R, C):
    """(R, C) array indexed (vertical, horizontal frequency) from the storage order: transposed for transforms at least as tall as wide"""
    return stored.reshape(C, R).T if R >= C else stored.reshape(R, C)


def idct_libjxl(sem):
    """inverse DCT in libjxl's normalisation (coefficient (0, 0) = the mean of the block)"""
    from scipy.fft import idctn
    return idctn(sem * np.sqrt(sem.shape[0] * sem.shape[1]), norm="ortho")


@pytest.mark.parametrize("w,h,mix,seed", [(256, 192, 0, 11), (512, 384, 1, 12), (520, 300, 2, 13)])
def test_dequantisation_cfl_and_idct_follow_their_definition(jx, w, h, mix, seed):
    """the lowest frequencies replaced by the LLF values, then the separable inverse DCT in libjxl's normalisation (coefficient 0 = block mean:
    scipy's orthonormal idctn of coefficients x sqrt(rows x cols)); output: the planes after the IDCT stage."""
    data = S.encode_vardct(S.synthetic_image(seed, w, h), seed=seed, strategy_mix=mix, epf_iters=0, gab=0)
    got, b, blocks = dequantised_varblocks(jx, data, h, w)
    bw, bh = b.info_value("frame0_bw"), b.info_value("frame0_bh")
    llf = [b.debug_read(0, "llf", c).reshape(bh, bw).astype(np.float64) for c in range(3)]
    checked = worst = 0
    scale = max(1e-3, float(np.abs(got).max()))
    for s, by, bx, deq in blocks:
        if s not in PLAIN:
            continue
        cx, cy, _ = PLAIN[s]
        R, Cc = cy * 8, cx * 8
        for c in range(3):
            sem = semantic(deq[c], R, Cc).copy()
            sem[:cy, :cx] = llf[c][by:by + cy, bx:bx + cx]
            want = idct_libjxl(sem)
            y0, x0 = by * 8, bx * 8
            hh, ww = min(R, h - y0), min(Cc, w - x0)
            worst = max(worst, float(np.abs(got[c][y0:y0 + hh, x0:x0 + ww] - want[:hh, :ww]).max()))
        checked += 1
    assert checked > (bw * bh) // 40, checked
    # up to 64 x 64 terms summed in float32 against float64: a few 1e-6 of the largest values in play
    assert worst <= 2e-5 * scale, (worst, scale)


# ---- the 8x8 transforms that are not a plain DCT (dec_transforms-inl.h TransformToPixels) ----------------------------------------------
def hadamard4(a, b, c, d):
    return a + b + c + d, a + b - c - d, a - b + c - d, a - b - c + d


def inverse_identity(c):
    """IDENTITY: four 4x4 quadrants, each a constant (its 'DC' from a 2x2 Hadamard of the four lowest coefficients, minus the mean of its residuals)
    plus per-pixel residuals; the residual of pixel (1, 1) is implied, its slot carries pixel (0, 0)'s"""
    out = np.zeros((8, 8))
    dcs = hadamard4(c[0, 0], c[0, 1], c[1, 0], c[1, 1])
    for y in range(2):
        for x in range(2):
            sub = c[y::2, x::2].copy()                     # sub[iy, ix] = c[y + 2 iy, x + 2 ix]
            anchor = dcs[2 * y + x] - (sub.sum() - sub[0, 0]) / 16.0
            blk = sub + anchor
            blk[0, 0] = sub[1, 1] + anchor
            blk[1, 1] = anchor
            out[4 * y:4 * y + 4, 4 * x:4 * x + 4] = blk
    return out


def inverse_dct2x2(c):
    """DCT2X2: three rounds of 2x2 Hadamard butterflies, each doubling the resolved top-left square (2, 4, 8)"""
    a = c.copy()
    for S_ in (2, 4, 8):
        n = S_ // 2
        r00, r01, r10, r11 = hadamard4(a[:n, :n].copy(), a[:n, n:S_].copy(), a[n:S_, :n].copy(), a[n:S_, n:S_].copy())
        a[0:S_:2, 0:S_:2], a[0:S_:2, 1:S_:2], a[1:S_:2, 0:S_:2], a[1:S_:2, 1:S_:2] = r00, r01, r10, r11
    return a


def inverse_dct4x4(c):
    out = np.zeros((8, 8))
    dcs = hadamard4(c[0, 0], c[0, 1], c[1, 0], c[1, 1])
    for y in range(2):
        for x in range(2):
            sub = c[y::2, x::2].copy()
            sub[0, 0] = dcs[2 * y + x]
            out[4 * y:4 * y + 4, 4 * x:4 * x + 4] = idct_libjxl(sub.T)    # a square block is stored transposed
    return out


def inverse_dct4x8(c):
    """two 4-row x 8-column halves, one above the other; their means are sum and difference of coefficients (0, 0) and (1, 0)"""
    out = np.zeros((8, 8))
    dcs = (c[0, 0] + c[1, 0], c[0, 0] - c[1, 0])
    for y in range(2):
        blk = c[y::2, :].copy()
        blk[0, 0] = dcs[y]
        out[4 * y:4 * y + 4, :] = idct_libjxl(blk)
    return out


def inverse_dct8x4(c):
    out = np.zeros((8, 8))
    dcs = (c[0, 0] + c[1, 0], c[0, 0] - c[1, 0])
    for x in range(2):
        blk = c[x::2, :].copy()
        blk[0, 0] = dcs[x]
        out[:, 4 * x:4 * x + 4] = idct_libjxl(blk.T)                       # 8 rows x 4 columns: stored transposed
    return out


# the 16 basis functions of the AFV corner transform (ISO/IEC 18181-1; a table of constants like the opsin matrix — what can be checked without
# libjxl is checked below: orthonormal, first row constant, every row symmetric or antisymmetric under transposition of the 4x4 block)
AFV_BASIS = np.array([
    [0.25] * 16,
    [0.876902929799142, 0.2206518106944235, -0.10140050393753763, -0.1014005039375375, 0.2206518106944236, -0.10140050393753777, -0.10140050393753772, -0.10140050393753763,
     -0.10140050393753758, -0.10140050393753769, -0.1014005039375375, -0.10140050393753768, -0.10140050393753768, -0.10140050393753759, -0.10140050393753763, -0.10140050393753741],
    [0.0, 0.0, 0.40670075830260755, 0.44444816619734445, 0.0, 0.0, 0.19574399372042936, 0.2929100136981264, -0.40670075830260716, -0.19574399372042872, 0.0, 0.11379074460448091,
     -0.44444816619734384, -0.29291001369812636, -0.1137907446044814, 0.0],
    [0.0, 0.0, -0.21255748058288748, 0.3085497062849767, 0.0, 0.4706702258572536, -0.1621205195722993, 0.0, -0.21255748058287047, -0.16212051957228327, -0.47067022585725277,
     -0.1464291867126764, 0.3085497062849487, 0.0, -0.14642918671266536, 0.4251149611657548],
    [0.0, -0.7071067811865474, 0.0, 0.0, 0.7071067811865476, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0],
    [-0.4105377591765233, 0.6235485373547691, -0.06435071657946274, -0.06435071657946266, 0.6235485373547694, -0.06435071657946284, -0.0643507165794628, -0.06435071657946274,
     -0.06435071657946272, -0.06435071657946279, -0.06435071657946266, -0.06435071657946277, -0.06435071657946277, -0.06435071657946273, -0.06435071657946274, -0.0643507165794626],
    [0.0, 0.0, -0.4517556589999482, 0.15854503551840063, 0.0, -0.04038515160822202, 0.0074182263792423875, 0.39351034269210167, -0.45175565899994635, 0.007418226379244351,
     0.1107416575309343, 0.08298163094882051, 0.15854503551839705, 0.3935103426921022, 0.0829816309488214, -0.45175565899994796],
    [0.0, 0.0, -0.304684750724869, 0.5112616136591823, 0.0, 0.0, -0.290480129728998, -0.06578701549142804, 0.304684750724884, 0.2904801297290076, 0.0, -0.23889773523344604,
     -0.5112616136592012, 0.06578701549142545, 0.23889773523345467, 0.0],
    [0.0, 0.0, 0.3017929516615495, 0.25792362796341184, 0.0, 0.16272340142866204, 0.09520022653475037, 0.0, 0.3017929516615503, 0.09520022653475055, -0.16272340142866173,
     -0.35312385449816297, 0.25792362796341295, 0.0, -0.3531238544981624, -0.6035859033230976],
    [0.0, 0.0, 0.40824829046386274, 0.0, 0.0, 0.0, 0.0, -0.4082482904638628, -0.4082482904638635, 0.0, 0.0, -0.40824829046386296, 0.0, 0.4082482904638634, 0.408248290463863, 0.0],
    [0.0, 0.0, 0.1747866975480809, 0.0812611176717539, 0.0, 0.0, -0.3675398009862027, -0.307882213957909, -0.17478669754808135, 0.3675398009862011, 0.0, 0.4826689115059883,
     -0.08126111767175039, 0.30788221395790305, -0.48266891150598584, 0.0],
    [0.0, 0.0, -0.21105601049335784, 0.18567180916109802, 0.0, 0.0, 0.49215859013738733, -0.38525013709251915, 0.21105601049335806, -0.49215859013738905, 0.0, 0.17419412659916217,
     -0.18567180916109904, 0.3852501370925211, -0.1741941265991621, 0.0],
    [0.0, 0.0, -0.14266084808807264, -0.3416446842253372, 0.0, 0.7367497537172237, 0.24627107722075148, -0.08574019035519306, -0.14266084808807344, 0.24627107722075137,
     0.14883399227113567, -0.04768680350229251, -0.3416446842253373, -0.08574019035519267, -0.047686803502292804, -0.14266084808807242],
    [0.0, 0.0, -0.13813540350758585, 0.3302282550303788, 0.0, 0.08755115000587084, -0.07946706605909573, -0.4613374887461511, -0.13813540350758294, -0.07946706605910261,
     0.49724647109535086, 0.12538059448563663, 0.3302282550303805, -0.4613374887461554, 0.12538059448564315, -0.13813540350758452],
    [0.0, 0.0, -0.17437602599651067, 0.0702790691196284, 0.0, -0.2921026642334881, 0.3623817333531167, 0.0, -0.1743760259965108, 0.36238173335311646, 0.29210266423348785,
     -0.4326608024727445, 0.07027906911962818, 0.0, -0.4326608024727457, 0.34875205199302267],
    [0.0, 0.0, 0.11354987314994337, -0.07417504595810355, 0.0, 0.19402893032594343, -0.435190496523228, 0.21918684838857466, 0.11354987314994257, -0.4351904965232251,
     0.5550443808910661, -0.25468277124066463, -0.07417504595810233, 0.2191868483885728, -0.25468277124066413, 0.1135498731499429]])


def inverse_afv(c, kind):
    """AFV0-3: a 4x4 corner (which one: kind & 1 = right, kind >> 1 = bottom) in the AFV basis, a 4x4 DCT beside it, a 4x8 DCT in the other half"""
    right, bottom = kind & 1, kind >> 1
    out = np.zeros((8, 8))
    c00, c01, c10 = c[0, 0], c[0, 1], c[1, 0]
    corner = c[0::2, 0::2].copy()
    corner[0, 0] = (c00 + c10 + c01) * 4.0
    px = (corner.reshape(16) @ AFV_BASIS).reshape(4, 4)
    if bottom:
        px = px[::-1, :]
    if right:
        px = px[:, ::-1]
    out[4 * bottom:4 * bottom + 4, 4 * right:4 * right + 4] = px
    beside = c[0::2, 1::2].copy()
    beside[0, 0] = c00 + c10 - c01
    x0 = 0 if right else 4
    out[4 * bottom:4 * bottom + 4, x0:x0 + 4] = idct_libjxl(beside.T)
    half = c[1::2, :].copy()
    half[0, 0] = c00 - c10
    y0 = 0 if bottom else 4
    out[y0:y0 + 4, :] = idct_libjxl(half)
    return out


SPECIAL = {1: inverse_identity, 2: inverse_dct2x2, 3: inverse_dct4x4, 12: inverse_dct4x8, 13: inverse_dct8x4,
           14: lambda c: inverse_afv(c, 0), 15: lambda c: inverse_afv(c, 1), 16: lambda c: inverse_afv(c, 2), 17: lambda c: inverse_afv(c, 3)}


def test_afv_basis_is_an_orthonormal_basis_with_the_expected_symmetries():
    assert np.abs(AFV_BASIS @ AFV_BASIS.T - np.eye(16)).max() < 1e-13
    for row in AFV_BASIS:
        m = row.reshape(4, 4)
        assert np.allclose(m, m.T, atol=1e-13) or np.allclose(m, -m.T, atol=1e-13)


@pytest.mark.parametrize("s", sorted(SPECIAL))
def test_special_8x8_transforms_follow_their_definition(jx, s):
    w, h = 136, 104
    data = S.encode_vardct(S.synthetic_image(20 + s, w, h), seed=20 + s, strategy_mix=100 + s, epf_iters=0, gab=0)
    got, b, blocks = dequantised_varblocks(jx, data, h, w)
    bw, bh = b.info_value("frame0_bw"), b.info_value("frame0_bh")
    llf = [b.debug_read(0, "llf", c).reshape(bh, bw).astype(np.float64) for c in range(3)]
    checked = worst = 0
    scale = max(1e-3, float(np.abs(got).max()))
    for bs, by, bx, deq in blocks:
        if bs != s:
            continue
        for c in range(3):
            blk = deq[c].reshape(8, 8).copy()
            blk[0, 0] = llf[c][by, bx]
            want = SPECIAL[s](blk)
            y0, x0 = by * 8, bx * 8
            hh, ww = min(8, h - y0), min(8, w - x0)
            worst = max(worst, float(np.abs(got[c][y0:y0 + hh, x0:x0 + ww] - want[:hh, :ww]).max()))
        checked += 1
    assert checked >= (bw * bh) // 2, checked
    assert worst <= 2e-5 * scale, (s, worst, scale)


# ---- lowest frequencies of a multi-block transform from its LF samples (dec_transforms-inl.h LowestFrequenciesFromDC) -------------------
def llf_from_lf(block):
    """ISO/IEC 18181-1: the cy x cx LF samples are transformed by a DCT of their own size and coefficient (v, u) is scaled by
    prod_{i < 3} cos(v pi 2^i / (16 cy)) x the same in u — cos(a) cos(2a) cos(4a) = sin(8a) / (8 sin a), dct_scales.h DCTTotalResampleScale<N, 8N>.
    (Were the LF samples plain 8x8 means of a band-limited block, the exact relation would DIVIDE by that factor: the format defines the LF image
    through this mapping instead — enc_transforms-inl.h DCFromLowestFrequencies applies the reciprocals — so the product is the definition.)"""
    from scipy.fft import dctn
    cy, cx = block.shape
    coeff = dctn(block, norm="ortho") / np.sqrt(cy * cx)

    def scale(n):
        k = np.arange(n)
        a = k * np.pi / (16.0 * n)
        return np.cos(a) * np.cos(2 * a) * np.cos(4 * a)
    return coeff * scale(cy)[:, None] * scale(cx)[None, :]


@pytest.mark.parametrize("w,h,mix,seed", [(520, 300, 2, 31), (384, 256, 1, 32)])
def test_lowest_frequencies_from_the_lf_samples(jx, w, h, mix, seed):
    data = S.encode_vardct(S.synthetic_image(seed, w, h), seed=seed, strategy_mix=mix, epf_iters=0, gab=0)
    _, b = planes_after(jx, data, 1, 0, 0)
    bw, bh = b.info_value("frame0_bw"), b.info_value("frame0_bh")
    info = b.debug_read(0, "blk_info", dtype=np.uint32).reshape(bh, bw)
    lf = [b.debug_read(0, "lf_smooth", c).reshape(bh, bw).astype(np.float64) for c in range(3)]
    llf = [b.debug_read(0, "llf", c).reshape(bh, bw).astype(np.float64) for c in range(3)]
    shapes, worst = set(), 0.0
    scale = max(1e-3, max(float(np.abs(p).max()) for p in lf))
    for by in range(bh):
        for bx in range(bw):
            word = int(info[by, bx])
            if not (word >> 5) & 1:
                continue
            cx, cy = COVERED[word & 31]
            if cx > 8 or cy > 8:
                continue
            shapes.add((cy, cx))
            for c in range(3):
                want = llf_from_lf(lf[c][by:by + cy, bx:bx + cx])
                worst = max(worst, float(np.abs(llf[c][by:by + cy, bx:bx + cx] - want).max()))
    assert len(shapes) >= (8 if mix == 2 else 5), shapes
    assert worst <= 16 * EPS * scale, (worst, scale)


# ---- DCT128 / DCT256 family: LLF from the smoothed LF samples, then scipy's idctn ------------------------------------------------------
@pytest.mark.parametrize("s,w,h", [(21, 300, 264), (22, 264, 136), (23, 136, 264), (24, 520, 264), (25, 264, 264), (26, 264, 264)])
def test_dct128_256_family_follows_its_definition(jx, s, w, h):
    data = S.encode_vardct(S.synthetic_image(40 + s, w, h), seed=40 + s, strategy_mix=100 + s, epf_iters=0, gab=0)
    got, b, blocks = dequantised_varblocks(jx, data, h, w)
    bw, bh = b.info_value("frame0_bw"), b.info_value("frame0_bh")
    lf = [b.debug_read(0, "lf_smooth", c).reshape(bh, bw).astype(np.float64) for c in range(3)]
    checked = worst = 0
    scale = max(1e-3, float(np.abs(got).max()))
    for bs, by, bx, deq in blocks:
        if bs != s:
            continue
        cx, cy = COVERED[s]
        R, Cc = cy * 8, cx * 8
        for c in range(3):
            sem = semantic(deq[c], R, Cc).copy()
            sem[:cy, :cx] = llf_from_lf(lf[c][by:by + cy, bx:bx + cx])
            want = idct_libjxl(sem)
            y0, x0 = by * 8, bx * 8
            hh, ww = min(R, h - y0), min(Cc, w - x0)
            worst = max(worst, float(np.abs(got[c][y0:y0 + hh, x0:x0 + ww] - want[:hh, :ww]).max()))
        checked += 1
    assert checked >= 1, checked
    # 256 x 256 terms in float32 butterflies against float64
    assert worst <= 4e-5 * scale, (s, worst, scale)


# ---- adaptive LF smoothing (compressed_dc.cc AdaptiveDCSmoothing) ------------------------------------------------------------------------
@pytest.mark.parametrize("w,h,seed", [(520, 300, 51), (136, 104, 52), (16, 16, 53)])
def test_adaptive_lf_smoothing_follows_its_definition(jx, w, h, seed):
    """interior samples: s = w0 c + w1 (4-neighbours) + w2 (diagonals) with w1 = 0.20345139757231578, w2 = 0.0334829185968739, w0 = 1 - 4 (w1 + w2);
    gap = max(0.5, max_c |c - s| / LF step of the channel); result c + (s - c) max(0, 3 - 4 gap); the border row / column is kept.
    The LF step of channel c = default LF dequantisation weight (1 / 4096, 1 / 512, 1 / 256) x 65536 / (global_scale x quant_lf)."""
    data = S.encode_vardct(S.synthetic_image(seed, w, h), seed=seed, strategy_mix=1, epf_iters=0, gab=0)
    _, b = planes_after(jx, data, 1, 0, 0)
    bw, bh = b.info_value("frame0_bw"), b.info_value("frame0_bh")
    before = np.stack([b.debug_read(0, "lf", c).reshape(bh, bw).astype(np.float64) for c in range(3)])
    after = np.stack([b.debug_read(0, "lf_smooth", c).reshape(bh, bw).astype(np.float64) for c in range(3)])
    lfq = np.stack([b.debug_read(0, "lfq", c, dtype=np.int32).reshape(bh, bw).astype(np.float64) for c in range(3)])
    step = np.array([1 / 4096.0, 1 / 512.0, 1 / 256.0]) * 65536.0 / (b.info_value("frame0_global_scale") * b.info_value("frame0_quant_lf"))
    # LF dequantisation of the luma channel (no chroma-from-luma term there): sample = quantised value x step
    assert np.abs(before[1] - lfq[1] * step[1]).max() <= 2 * EPS * np.abs(before[1]).max()
    w1, w2 = 0.20345139757231578, 0.0334829185968739
    w0 = 1.0 - 4.0 * (w1 + w2)
    want = before.copy()
    if bw > 2 and bh > 2:
        c_ = before[:, 1:-1, 1:-1]
        edge = before[:, :-2, 1:-1] + before[:, 2:, 1:-1] + before[:, 1:-1, :-2] + before[:, 1:-1, 2:]
        diag = before[:, :-2, :-2] + before[:, :-2, 2:] + before[:, 2:, :-2] + before[:, 2:, 2:]
        sm = w0 * c_ + w1 * edge + w2 * diag
        gap = np.maximum(0.5, (np.abs(c_ - sm) / step[:, None, None]).max(axis=0))
        want[:, 1:-1, 1:-1] = c_ + (sm - c_) * np.maximum(0.0, 3.0 - 4.0 * gap)
    scale = max(1e-3, float(np.abs(before).max()))
    # |c - s| / step amplifies a float32 rounding of s by 1 / step (~1e2-1e3) into the blend factor: a few 1e-5 of the sample range
    assert np.abs(after - want).max() <= 1e-5 * scale, float(np.abs(after - want).max())
    if w >= 512:    # (on the small image every sample's gap is above 0.75 — blend factor 0 — which the comparison above covers as well)
        assert np.abs(after - before).max() > 0, "the smoothing moved nothing: the test image is too rough to say anything"


# ---- 2x / 4x / 8x upsampling (stage_upsampling.cc) -------------------------------------------------------------------------------------------
def upsample_float64(p, up, weights):
    """every input sample becomes up x up output samples; output (sy, sx) of a sample is a 5x5 kernel over its neighbourhood (mirrored at the
    borders), clamped to the range of those 25 samples.  Kernels: the stored 15 / 55 / 210 weights are the upper triangle of a symmetric
    (5 up / 2)^2 matrix M, kernel[ky][kx][iy][ix] = M[5 ky + iy][5 kx + ix] for the top-left quadrant of sub-positions, mirrored for the others."""
    n = up // 2
    M = np.zeros((5 * n, 5 * n))
    k = 0
    for i in range(5 * n):
        for j in range(i, 5 * n):
            M[i, j] = M[j, i] = weights[k]
            k += 1
    assert k == len(weights)
    h, w = p.shape
    pad = np.pad(p, 2, mode="symmetric")
    win = np.stack([pad[iy:iy + h, ix:ix + w] for iy in range(5) for ix in range(5)])     # (25, h, w)
    lo, hi = win.min(axis=0), win.max(axis=0)
    out = np.zeros((h * up, w * up))
    for sy in range(up):
        ky, flip_y = (sy, False) if sy < n else (up - 1 - sy, True)
        for sx in range(up):
            kx, flip_x = (sx, False) if sx < n else (up - 1 - sx, True)
            kern = M[5 * ky:5 * ky + 5, 5 * kx:5 * kx + 5]
            if flip_y:
                kern = kern[::-1, :]
            if flip_x:
                kern = kern[:, ::-1]
            v = np.tensordot(kern.reshape(25), win, axes=1)
            out[sy::up, sx::up] = np.clip(v, lo, hi)
    return out


@pytest.mark.parametrize("up,custom", [(2, 0), (2, 1), (4, 1), (8, 1), (4, 0), (8, 0)])
def test_upsampling_follows_its_definition(jx, up, custom):
    """linear f32 output of an upsampled frame = inverse opsin (pinned above) of the float64 upsampling of the planes the filters left"""
    S.set_color(1, 1, 8)
    try:
        data = S.encode_vardct(S.synthetic_image(60 + up, 96, 72), seed=60 + up, strategy_mix=1, epf_iters=0, gab=0, upsampling=up, custom_up_weights=custom)
    finally:
        S.set_color()
    b = jx.BatchDecoder(0)
    b.add(data, "float32", 3)
    b.set_option("force_unfused_filters", 1)
    b.prepare()
    b.set_option("debug_stop_after", 1)
    b.decode(); b.finish()
    info = b.info(0)
    W, H, bw, bh = info.xsize, info.ysize, b.info_value("frame0_bw"), b.info_value("frame0_bh")
    w, h = (W + up - 1) // up, (H + up - 1) // up
    xyb = np.stack([b.debug_read(0, "plane_a", c).reshape(bh * 8, bw * 8)[:h, :w] for c in range(3)]).astype(np.float64)
    weights = b.debug_read(0, "up_weights").astype(np.float64)
    assert len(weights) == {2: 15, 4: 55, 8: 210}[up]
    if not custom:     # the library defaults: every kernel of a partition of unity sums to 1 (a constant image stays constant)
        n = up // 2
        M = np.zeros((5 * n, 5 * n)); k = 0
        for i in range(5 * n):
            for j in range(i, 5 * n):
                M[i, j] = M[j, i] = weights[k]; k += 1
        sums = M.reshape(n, 5, n, 5).sum(axis=(1, 3))
        assert np.abs(sums - 1.0).max() < 2e-3, sums
    X, Y, B = [upsample_float64(xyb[c], up, weights)[:H, :W] for c in range(3)]
    b.set_option("debug_stop_after", 0)
    b.decode(); b.finish()
    got = b.output(0).reshape(H, W, 3).astype(np.float64)
    bias = -0.0037930732552754493
    cb = np.cbrt(bias)
    mixed = [np.power(Y + X - cb, 3) + bias, np.power(Y - X - cb, 3) + bias, np.power(B - cb, 3) + bias]
    inv = np.array([[11.031566901960783, -9.866943921568629, -0.16462299647058826],
                    [-3.254147380392157, 4.418770392156863, -0.16462299647058826],
                    [-3.6588512862745097, 2.7129230470588235, 1.9459282392156863]])
    want = np.stack([sum(inv[r, k] * mixed[k] for k in range(3)) for r in range(3)], axis=-1)
    # 25-term kernels in float32 (1e-7 relative on XYB values below 1) through the cube and a matrix with entries up to 11: ~1e-5 absolute
    assert np.abs(got - want).max() <= 2e-5, float(np.abs(got - want).max())


# ---- frame blending (blending.cc PerformBlending) ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_blend_modes_follow_their_definition(jx, mode):
    """A cropped RGBA layer over a saved RGBA frame, every blend mode, non-premultiplied alpha.  Inputs: the background frame decoded as an image of its own and the
    layer as the non-coalesced decode hands it out (its own pixels after the colour transform); expected canvas in float64 from the definitions:
    0 replace; 1 add; 2 blend: a = fa + ba (1 - fa), colour (fg fa + bg ba (1 - fa)) / a (0 where a = 0); 3 alpha-weighted add: colour bg + fg fa, alpha of the
    background kept; 4 multiply: bg x fg.  No oracle code."""
    w, h, cw, ch, x0, y0 = 200, 136, 64, 48, 10, 20
    img, small = S.synthetic_image(5, w, h), S.synthetic_image(9, cw, ch)
    bg_a = (np.add.outer(np.arange(h), np.arange(w)) * 5 % 256).astype(np.uint8)
    fg_a = (np.add.outer(np.arange(ch), np.arange(cw)) * 3 % 256).astype(np.uint8)
    f0 = np.dstack([img, bg_a]); f1 = np.dstack([small, fg_a])
    alone = S.encode_modular_frame(f0, S.frame(), bits=8)
    both = (S.encode_modular_frame(f0, S.frame(is_last=0, save_as_reference=2), bits=8)
            + S.encode_modular_frame(f1, S.frame(emit=1, have_crop=1, crop_x0=x0, crop_y0=y0, canvas_w=w, canvas_h=h, blend_mode=mode, blend_source=2), bits=8))
    _, bg = jx.decoder_builder().decode_with(alone, np.float32)
    _, fg = jx.decoder_builder(coalescing=False).decode_with(both, np.float32)
    _, got = jx.decoder_builder().decode_with(both, np.float32)
    bg = bg.reshape(h, w, 4).astype(np.float64); fg = fg.reshape(ch, cw, 4).astype(np.float64); got = got.reshape(h, w, 4).astype(np.float64)
    assert np.abs(bg * 255 - f0).max() < 1e-4      # (lossless 8-bit samples as floats)
    want = bg.copy()
    B, F = bg[y0:y0 + ch, x0:x0 + cw], fg
    ba, fa = B[..., 3:], F[..., 3:]
    if mode == 0:
        out = F
    elif mode == 1:
        out = B + F
    elif mode == 2:
        a = fa + ba * (1 - fa)
        col = np.where(a > 0, (F[..., :3] * fa + B[..., :3] * ba * (1 - fa)) / np.where(a > 0, a, 1), 0.0)
        out = np.concatenate([col, a], axis=-1)
    elif mode == 3:
        out = np.concatenate([B[..., :3] + F[..., :3] * fa, ba], axis=-1)
    else:
        out = B * F
    want[y0:y0 + ch, x0:x0 + cw] = out
    assert np.abs(got - want).max() <= 4 * EPS * max(1.0, float(np.abs(want).max())), (mode, float(np.abs(got - want).max()))


# ---- patches (stage_patches.cc; blending.cc) ----------------------------------------------------------------------------------------------
def test_patch_blend_modes_follow_their_definition(jx):
    """Rectangles of a saved reference frame (RGBA, lossless Modular, kept before the colour transform) pasted into an RGBA frame under every patch blend mode,
    colour and alpha each with the placement's own mode, non-premultiplied alpha: 1 replace; 2 add; 3 multiply; 4 blend above (patch over frame):
    a = pa + fa (1 - pa), colour (patch pa + frame fa (1 - pa)) / a; 5 blend below (frame over patch); 6 alpha-weighted add above: frame + patch pa, alpha of the
    frame kept; 7 alpha-weighted add below: patch + frame fa, alpha of the patch.  Inputs: the frame without patches and the reference picture, each decoded as an
    image of its own; expected values in float64; no oracle code."""
    w, h, rw, rh = 200, 136, 64, 48
    main, ref = S.synthetic_image(51, w, h), S.synthetic_image(50, rw, rh)
    al_main = (64 + np.add.outer(np.arange(h), np.arange(w)) % 192).astype(np.uint8)
    al_ref = (np.add.outer(np.arange(rh), np.arange(rw)) * 3 % 256).astype(np.uint8)
    f_main, f_ref = np.dstack([main, al_main]), np.dstack([ref, al_ref])
    px0, py0, pw, ph = 2, 3, 24, 20                                            # the rectangle of the reference frame that is pasted
    spots = {1: (5, 5), 2: (60, 10), 3: (120, 40), 4: (30, 90), 5: (170, 110), 6: (100, 100), 7: (150, 5)}      # mode -> where (non-overlapping; one runs off the right edge? no: all inside)
    patches = [(1, px0, py0, pw, ph, [(x, y, [(m, 0, 0), (m, 0, 0)]) for m, (x, y) in spots.items()])]
    plain = S.encode_modular_frame(f_main, S.frame(), bits=8)
    ref_alone = S.encode_modular_frame(f_ref, S.frame(), bits=8)
    ref_frame = S.encode_modular_frame(f_ref, S.frame(frame_type=2, is_last=0, save_as_reference=1, save_before_ct=1, have_crop=1, canvas_w=w, canvas_h=h), bits=8)
    S.set_features(patches=patches, num_extra=1)
    try:
        with_patches = ref_frame + S.encode_modular_frame(f_main, S.frame(emit=1), bits=8)
    finally:
        S.set_features()
    dec = lambda d, hh, ww: jx.decoder_builder().decode_with(d, np.float32)[1].reshape(hh, ww, 4).astype(np.float64)
    frame, src, got = dec(plain, h, w), dec(ref_alone, rh, rw), dec(with_patches, h, w)
    assert np.abs(frame * 255 - f_main).max() < 1e-4 and np.abs(src * 255 - f_ref).max() < 1e-4
    P = src[py0:py0 + ph, px0:px0 + pw]
    want = frame.copy()
    for m, (x, y) in spots.items():
        F = frame[y:y + ph, x:x + pw]
        fa, pa = F[..., 3:], P[..., 3:]
        if m == 1:
            out = P
        elif m == 2:
            out = F + P
        elif m == 3:
            out = F * P
        elif m in (4, 5):
            fg, bg = (P, F) if m == 4 else (F, P)
            ga, ba = fg[..., 3:], bg[..., 3:]
            a = ga + ba * (1 - ga)
            col = np.where(a > 0, (fg[..., :3] * ga + bg[..., :3] * ba * (1 - ga)) / np.where(a > 0, a, 1), 0.0)
            out = np.concatenate([col, a], axis=-1)
        elif m == 6:
            out = np.concatenate([F[..., :3] + P[..., :3] * pa, fa], axis=-1)
        else:
            out = np.concatenate([P[..., :3] + F[..., :3] * fa, pa], axis=-1)
        want[y:y + ph, x:x + pw] = out
    assert np.abs(want - frame).max() > 0.2
    err = np.abs(got - want)
    assert err.max() <= 4 * EPS * max(1.0, float(np.abs(want).max())), {m: float(err[y:y + ph, x:x + pw].max()) for m, (x, y) in spots.items()}


# ---- noise synthesis -------------------------------------------------------------------------------------------------------------------
M64 = (1 << 64) - 1


def splitmix64(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def noise_planes(w, h, group_dim, visible, nonvisible):
    """The three planes of uniform samples in [1, 2) (dec_noise.cc Random3Planes / base/random.h): per group of the frame, eight Xorshift128+
    generators seeded by SplitMix64 chains from (visible frame index, non-visible frame index) and (x0, y0) of the group; a batch of eight 64-bit
    words gives sixteen floats (low word first), mantissa = the top 23 bits of each 32-bit word; rows take whole batches, planes follow each other."""
    out = np.zeros((3, h, w), np.float32)
    golden = 0x9E3779B97F4A7C15
    for gy0 in range(0, h, group_dim):
        for gx0 in range(0, w, group_dim):
            s0 = [splitmix64((((visible << 32) + nonvisible) + golden) & M64)]
            s1 = [splitmix64((((gx0 << 32) + gy0) + golden) & M64)]
            for _ in range(7):
                s0.append(splitmix64(s0[-1])); s1.append(splitmix64(s1[-1]))
            s0, s1 = np.array(s0, np.uint64), np.array(s1, np.uint64)
            xs, ys = min(group_dim, w - gx0), min(group_dim, h - gy0)
            nb = (xs + 15) // 16
            for c in range(3):
                words = np.empty((ys, nb, 8), np.uint64)
                for y in range(ys):
                    for k in range(nb):
                        a, b = s0, s1
                        words[y, k] = a + b
                        s0 = b
                        a = a ^ (a << np.uint64(23))
                        s1 = a ^ b ^ (a >> np.uint64(18)) ^ (b >> np.uint64(5))
                halves = words.view(np.uint32).reshape(ys, nb * 16)          # little endian: low word first
                f = ((halves >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32)
                out[c, gy0:gy0 + ys, gx0:gx0 + xs] = f[:, :xs]
    return out


def noise_strength(lut, v):
    """stage_noise.cc: piecewise-linear LUT over eight points at intensity k / 6, clamped to [0, 1]; beyond the last point the last value"""
    s = np.maximum(0.0, v * 6.0)
    fl = np.floor(s)
    frac = s - fl
    over = s >= 7.0
    fl = np.where(over, 6.0, fl); frac = np.where(over, 1.0, frac)
    i = fl.astype(int)
    return np.clip(lut[i] + (lut[i + 1] - lut[i]) * frac, 0.0, 1.0)


@pytest.mark.parametrize("w,h,lut", [(300, 280, [30, 60, 90, 120, 150, 180, 210, 240]), (77, 61, [1023, 800, 600, 400, 300, 200, 100, 0])])
def test_noise_synthesis_follows_its_definition(jx, w, h, lut):
    """Noise (ISO/IEC 18181-1 noise synthesis; libjxl dec_noise.cc + stage_noise.cc): the pseudo-random planes, the 5x5 high-pass (centre -3.84, the other 24
    taps 0.16, mirrored borders), the intensity-dependent strength from the frame's 8-point LUT, the correlated injection into X / Y / B with the frame's base
    chroma-from-luma factors (0 and 1).  The decoder's linear f32 pixels of the noisy frame against inverse-opsin(float64) of (the XYB planes of the SAME
    frame before noise, read off the device, + the noise restated here)."""
    img = S.synthetic_image(19, w, h)
    S.set_color(1, 1, 8)
    try:
        data = S.encode_vardct_frame(img, S.frame(noise_lut=lut), seed=3, strategy_mix=1, epf_iters=0, gab=0)
    finally:
        S.set_color()
    xyb, _ = planes_after(jx, data, 1, 0, 0)
    px, _ = planes_after(jx, data, 0, 0, 0)
    rnd = noise_planes(w, h, 256, 1, 0).astype(np.float64)            # the first shown frame counts as visible frame 1 (dec_frame.cc InitFrame increments before the frame is decoded)
    pad = np.pad(rnd, ((0, 0), (2, 2), (2, 2)), mode="symmetric")            # Mirror(): -1 -> 0, -2 -> 1
    total = sum(pad[:, 2 + dy:2 + dy + h, 2 + dx:2 + dx + w] for dy in range(-2, 3) for dx in range(-2, 3))
    conv = (total - rnd) * 0.16 + rnd * -3.84
    X, Y, B = xyb
    lutf = np.array(lut, np.float64) / 1024.0
    sg, sr = noise_strength(lutf, (Y - X) * 0.5), noise_strength(lutf, (Y + X) * 0.5)
    ar, ag, ac = conv * 0.22
    red = sr * (0.0078125 * ar + 0.9921875 * ac)
    green = sg * (0.0078125 * ag + 0.9921875 * ac)
    rg = red + green
    X2, Y2, B2 = X + 0.0 * rg + red - green, Y + rg, B + 1.0 * rg
    assert np.abs(rg).max() > 0.02                                           # the noise is far above the tolerance below
    bias = -0.0037930732552754493
    cb = np.cbrt(bias)
    mixed = [np.power(Y2 + X2 - cb, 3) + bias, np.power(Y2 - X2 - cb, 3) + bias, np.power(B2 - cb, 3) + bias]
    inv = np.array([[11.031566901960783, -9.866943921568629, -0.16462299647058826],
                    [-3.254147380392157, 4.418770392156863, -0.16462299647058826],
                    [-3.6588512862745097, 2.7129230470588235, 1.9459282392156863]])
    want = np.stack([sum(inv[r, k] * mixed[k] for k in range(3)) for r in range(3)], axis=-1)
    got = px.astype(np.float64)
    # 25 taps of values in [1, 2) cancel to a high-pass of a few units: ~25 x 2^-23 x 2 absolute on the convolution, x 0.22 x strength, then the inverse opsin's
    # cancelling matrix rows (x ~11 x the slope of the cube): 2e-5 absolute on linear values in [0, 1]
    assert np.abs(got - want).max() <= 2e-5, float(np.abs(got - want).max())


# ---- splines -----------------------------------------------------------------------------------------------------------------------------
def catmull_rom(points):
    """Centripetal Catmull-Rom through the control points, 16 samples per span, the ends extended by reflection (splines.cc DrawCentripetalCatmullRomSpline)"""
    pts = [np.array(p, np.float64) for p in points]
    pts = [pts[0] + (pts[0] - pts[1])] + pts + [pts[-1] + (pts[-1] - pts[-2])]
    out = []
    for s in range(len(pts) - 3):
        p = pts[s:s + 4]
        out.append(p[1])
        d = [np.sqrt(np.hypot(*(p[k + 1] - p[k]))) for k in range(3)]       # knot spacing |chord|^(1/2)
        t = [0.0, d[0], d[0] + d[1], d[0] + d[1] + d[2]]
        for i in range(1, 16):
            tt = d[0] + i / 16 * d[1]
            a = [p[k] + (tt - t[k]) / d[k] * (p[k + 1] - p[k]) for k in range(3)]
            b = [a[k] + (tt - t[k]) / (d[k] + d[k + 1]) * (a[k + 1] - a[k]) for k in range(2)]
            out.append(b[0] + (tt - t[1]) / d[1] * (b[1] - b[0]))
    out.append(pts[-2])
    return out


def equally_spaced(points):
    """Walk the polyline in steps of arc length 1 (splines.cc ForEachEquallySpacedPoint): (point, multiplier) with multiplier 1 except for the remainder at the end"""
    out = [(points[0], 1.0)]
    current, nxt = points[0], 0
    while nxt < len(points):
        previous, from_previous = current, 0.0
        while True:
            if nxt == len(points):
                out.append((previous, from_previous))
                return out
            to_next = float(np.hypot(*(points[nxt] - previous)))
            if from_previous + to_next >= 1.0:
                current = previous + (1.0 - from_previous) / to_next * (points[nxt] - previous)
                out.append((current, 1.0))
                break
            from_previous += to_next
            previous = points[nxt]
            nxt += 1
    return out


def fast_erf_rational(x):
    """base/fast_math-inl.h FastErff: 1 - 1 / (1 + a1 x + a2 x^2 + a3 x^3 + a4 x^4)^4 (Abramowitz & Stegun 7.1.27 with refitted coefficients)"""
    a = np.abs(x)
    d = 1.0 + a * (2.77820801e-01 + a * (2.32120216e-01 + a * (2.05260015e-04 + a * 7.77394369e-02)))
    r = 1.0 - 1.0 / d ** 4
    return np.where(x <= 0, -r, r)


def render_splines(w, h, quant_adjust, splines, erf_fn):
    """(3, h, w) float64: what the splines add to X, Y, B (ISO/IEC 18181-1 splines; libjxl splines.cc): dequantised control points and 32-point DCTs of colour and
    sigma along the arc, one Gaussian-profile dab per unit of arc length: colour(t) * sigma(t) / 4 * multiplier * (erf((d / 2 + sqrt(2) / 4) / sigma) - erf((d / 2 - sqrt(2) / 4) / sigma))^2"""
    out = np.zeros((3, h, w))
    inv_quant = 1.0 / (1.0 + 0.125 * quant_adjust) if quant_adjust >= 0 else 1.0 - 0.125 * quant_adjust
    weight = [0.0042, 0.075, 0.07, 0.3333]
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    for sx, sy, deltas, colour_q, sigma_q in splines:
        cps, cx, cy, dx, dy = [(float(sx), float(sy))], sx, sy, 0, 0
        for ddx, ddy in deltas:
            dx += ddx; dy += ddy; cx += dx; cy += dy
            cps.append((float(cx), float(cy)))
        dct = np.array([np.array(colour_q[c], np.float64) * weight[c] for c in range(3)] + [np.array(sigma_q, np.float64) * weight[3]]) * inv_quant
        dct[:, 0] *= np.sqrt(0.5)
        dct[0] += 0.0 * dct[1]                                                # the frame's base chroma-from-luma factors: 0 for X ...
        dct[2] += 1.0 * dct[1]                                                # ... and 1 for B
        pts = equally_spaced(catmull_rom(cps))
        arc = (len(pts) - 2) + pts[-1][1]
        for k, (pt, mult) in enumerate(pts):
            t = 31.0 * min(1.0, k / arc)
            vals = np.sqrt(2.0) * (dct * np.cos(np.pi / 32 * np.arange(32) * (t + 0.5))).sum(axis=1)
            colour, sigma = vals[:3], vals[3]
            if sigma == 0 or not np.isfinite(1.0 / sigma):
                continue
            max_colour = max(0.01, float(np.abs(colour * mult).max()))
            max_dist = np.sqrt(-2.0 * sigma * sigma * (np.log(0.1) * 5 - np.log(max_colour)))
            y0, y1 = max(0, int(np.floor(pt[1] - max_dist + 0.5))), min(h, int(np.floor(pt[1] + max_dist + 0.5)) + 1)
            x0, x1 = max(0, int(np.floor(pt[0] - max_dist + 0.5))), min(w, int(np.floor(pt[0] + max_dist + 0.5)) + 1)
            if y0 >= y1 or x0 >= x1:
                continue
            d = np.hypot(xx[y0:y1, x0:x1] - pt[0], yy[y0:y1, x0:x1] - pt[1])
            f = erf_fn((d * 0.5 + np.sqrt(2) / 4) / sigma) - erf_fn((d * 0.5 - np.sqrt(2) / 4) / sigma)
            out[:, y0:y1, x0:x1] += colour[:, None, None] * (0.25 * sigma * mult * f * f)
    return out


def test_splines_follow_their_definition(jx):
    """Splines: the XYB planes before the splines come off the device (stop after the IDCT; no filters), the decoder's linear f32 pixels go back to XYB through the
    float64 forward opsin, and the difference is what the spline stage added.  Against the float64 rendering above, (1) with libjxl's rational erf (the curve, the
    arc-length walk, the DCT evaluation with exact cosines, the dab formula, the order-independent sum: what is left is float32 rounding and FastCosf, 4e-6 of
    its argument range) and (2) with the exact erf (the rational form is within 6e-4 of it: a looser bound that does not lean on the recalled coefficients)."""
    from scipy.special import erf
    w, h = 200, 136
    img = S.synthetic_image(23, w, h)
    colour = [[0] * 32 for _ in range(3)]
    colour[1][0] = 24; colour[0][0] = 30; colour[2][1] = -12; colour[1][3] = 5; colour[0][2] = -9
    sigma = [0] * 32
    sigma[0] = 9; sigma[2] = 2
    splines = [(20, 30, [(15, 5), (2, -3), (-4, 6)], colour, sigma), (150, 20, [(-10, 20)], colour, sigma), (60, 110, [(25, -7), (-3, -2)], colour, sigma)]
    S.set_color(1, 1, 8)
    S.set_features(splines=(1, splines))
    try:
        data = S.encode_vardct_frame(img, S.frame(), seed=4, strategy_mix=1, epf_iters=0, gab=0)
    finally:
        S.set_features()
        S.set_color()
    before, _ = planes_after(jx, data, 1, 0, 0)
    px, _ = planes_after(jx, data, 0, 0, 0)
    inv = np.array([[11.031566901960783, -9.866943921568629, -0.16462299647058826],
                    [-3.254147380392157, 4.418770392156863, -0.16462299647058826],
                    [-3.6588512862745097, 2.7129230470588235, 1.9459282392156863]])
    bias = -0.0037930732552754493
    mixed = np.einsum("rk,yxk->ryx", np.linalg.inv(inv), px.astype(np.float64))
    gamma = np.cbrt(mixed - bias) + np.cbrt(bias)
    after = np.stack([(gamma[0] - gamma[1]) / 2, (gamma[0] + gamma[1]) / 2, gamma[2]])
    got = after - before
    want = render_splines(w, h, 1, splines, fast_erf_rational)
    peak = np.abs(want).max(axis=(1, 2))
    assert peak.min() > 0.02, peak                                            # every channel carries a visible spline
    touched = np.abs(want).sum(axis=0) > 0
    assert 0.03 < touched.mean() < 0.6                                         # and most of the picture is left alone
    err = np.abs(got - want).max(axis=(1, 2))
    assert (err <= 2e-5 * peak + 4e-6).all(), (err, peak)
    exact = render_splines(w, h, 1, splines, erf)
    err = np.abs(got - exact).max(axis=(1, 2))
    assert (err <= 5e-3 * peak).all(), (err, peak)
