"""GPU tests (-m gpu) that pin single float stages of the VarDCT pipeline against float64 restatements of their DEFINING FORMULAS, written
here in numpy and sharing no code with oracle/ (which every other parity test compares with): gaborish (3x3 normalised kernel, mirrored
borders: libjxl stage_gaborish.cc / loop_filter.h), the edge-preserving filter passes 0 / 1 / 2 (epf.cc: per-block sigma from the quantiser
field and the sharpness map, weights 1 + SAD * inv_sigma * scale clamped at 0, plus-shaped SAD patches, border rule), the inverse opsin
transform (stage_xyb.cc: subtract cbrt(bias), cube, add bias, 3x3 matrix) and the sRGB transfer function (stage_from_linear.cc against the
IEC 61966-2-1 formula).  The planes between the stages come off the device through JxlHipBatchSetOption("debug_stop_after") +
JxlHipBatchDebugRead (include/jxl_hip.h); inputs of a stage are what the stage before left, so each test isolates one stage.
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


@pytest.mark.parametrize("w,h,mix,seed", [(256, 192, 0, 11), (512, 384, 1, 12), (520, 300, 2, 13)])
def test_dequantisation_cfl_and_idct_follow_their_definition(jx, w, h, mix, seed):
    """dec_group.cc / quantizer.h / dct-inl.h restated: coefficient = AdjustQuantBias(q) x table[k] x (65536 / global_scale) / hf_mul (x 0.8^(qm_scale - 2)
    for X, B), X and B plus (base + map / colour_factor) x Y, the lowest frequencies replaced by the LLF values, then the separable inverse DCT in
    libjxl's normalisation (coefficient 0 = block mean: scipy's orthonormal idctn of coefficients x sqrt(rows x cols)).  Inputs off the device:
    quantised coefficients, block info, LLF planes, tables, CfL maps; output: the planes after the IDCT stage."""
    from scipy.fft import idctn
    data = S.encode_vardct(S.synthetic_image(seed, w, h), seed=seed, strategy_mix=mix, epf_iters=0, gab=0)
    got, b = planes_after(jx, data, 1, 0, 0)
    bw, bh, xgroups = b.info_value("frame0_bw"), b.info_value("frame0_bh"), b.info_value("frame0_xgroups")
    info = b.debug_read(0, "blk_info", dtype=np.uint32).reshape(bh, bw)
    coef_off = b.debug_read(0, "coef_off", dtype=np.uint32).reshape(bh, bw)
    # (the IDCT zeroes the coefficient planes it consumes: decode up to the HF stage again to read them)
    b.decode_part(1); b.decode_part(3); b.finish()
    coeff = [b.debug_read(0, "coeff", c, dtype=np.int32).astype(np.float64) for c in range(3)]
    llf = [b.debug_read(0, "llf", c).reshape(bh, bw).astype(np.float64) for c in range(3)]
    cw = (bw + 7) // 8
    ytox = b.debug_read(0, "ytox", dtype=np.int8).astype(np.float64)
    ytob = b.debug_read(0, "ytob", dtype=np.int8).astype(np.float64)
    inv_global_scale = 65536.0 / b.info_value("frame0_global_scale")
    colour_scale = 1.0 / b.info_value("frame0_color_factor")
    dm = [0.8 ** (b.info_value("frame0_x_qm_scale") - 2.0), 1.0, 0.8 ** (b.info_value("frame0_b_qm_scale") - 2.0)]
    bias = [1.0 - 0.05465007330715401, 1.0 - 0.07005449891748593, 1.0 - 0.049935103337343655, 0.145]
    tables = {}
    checked = worst = 0
    scale = max(1e-3, float(np.abs(got).max()))
    for by in range(bh):
        for bx in range(bw):
            word = int(info[by, bx])
            if not (word >> 5) & 1:
                continue                                     # not the first block of its varblock
            s = word & 31
            if s not in PLAIN or by * 8 >= h or bx * 8 >= w:
                continue
            cx, cy, kind = PLAIN[s]
            R, Cc = cy * 8, cx * 8
            hf_mul = ((word >> 8) & 0xFF) + 1
            g = (by // 32) * xgroups + bx // 32
            base = g * 65536 + int(coef_off[by, bx])
            tile = (by // 8) * cw + bx // 8
            k_cfl = [0.0 + ytox[tile] * colour_scale, 0.0, 1.0 + ytob[tile] * colour_scale]
            # storage order of coefficient (v = vertical, u = horizontal frequency): transposed for blocks at least as tall as wide
            vv, uu = np.mgrid[0:R, 0:Cc]
            kidx = (uu * R + vv) if R >= Cc else (vv * Cc + uu)
            deq = []
            for c in range(3):
                if (kind, c) not in tables:
                    tables[(kind, c)] = b.debug_read(0, "qtable", kind * 3 + c).astype(np.float64)
                q = coeff[c][base + kidx]
                safe = np.where(q == 0, 1.0, q)
                adj = np.where(q == 0, 0.0, np.where(np.abs(q) == 1, np.sign(q) * bias[c], q - bias[3] / safe))
                deq.append(adj * tables[(kind, c)][kidx] * (inv_global_scale / hf_mul * dm[c]))
            sem = [deq[0] + k_cfl[0] * deq[1], deq[1], deq[2] + k_cfl[2] * deq[1]]
            for c in range(3):
                sem[c][:cy, :cx] = llf[c][by:by + cy, bx:bx + cx]
                want = idctn(sem[c] * np.sqrt(R * Cc), norm="ortho")
                y0, x0 = by * 8, bx * 8
                hh, ww = min(R, h - y0), min(Cc, w - x0)
                err = float(np.abs(got[c][y0:y0 + hh, x0:x0 + ww] - want[:hh, :ww]).max())
                worst = max(worst, err)
            checked += 1
    assert checked > (bw * bh) // 40, checked
    # up to 64 x 64 terms summed in float32 against float64: a few 1e-6 of the largest values in play
    assert worst <= 2e-5 * scale, (worst, scale)
