"""CPU tests: the oracle against every golden the reference's fixtures offer and against the
committed regression vectors.  The oracle is the parity checker for the HIP path, so it is pinned first."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import FIXTURES, GOLDEN, fixture_bytes, read_png16
import oracle_lib as O

MANIFEST = json.load(open(os.path.join(GOLDEN, "manifest.json")))


def test_sample_jxl_equals_sample_png_rgba16():
    """The reference's only pixel-exact assertion: jpegxl-rs/src/image.rs:169 (sample.jxl == sample.png as RGBA16)."""
    dec = O.decode(fixture_bytes("sample.jxl"))
    assert (dec.info.xsize, dec.info.ysize, dec.info.bits_per_sample, dec.info.alpha_bits) == (40, 50, 16, 16)
    got = dec.image("u16", 4)
    png = read_png16(os.path.join(FIXTURES, "sample.png"))
    assert png.shape == (50, 40, 4) and png.dtype == np.uint16
    assert np.array_equal(got, png)
    assert hashlib.sha256(got.astype(">u2").tobytes()).hexdigest() == MANIFEST["reference_fixtures"]["sample.jxl"]["sha256_rgba16_be"]
    assert tuple(got[0, 1]) == (61466, 64019, 63957, 65535)  # SURVEY App. C spot value


def test_sample_jxl_u8_derivation():
    """u8 output = round(v * 255 / 65535) = (v + 128) // 257 (no ties — SURVEY App. C)."""
    dec = O.decode(fixture_bytes("sample.jxl"))
    u16 = dec.image("u16", 4).astype(np.int64)
    assert np.array_equal(dec.image("u8", 4), ((u16 + 128) // 257).astype(np.uint8))
    assert np.array_equal(dec.image("u8", 3), ((u16[..., :3] + 128) // 257).astype(np.uint8))


def test_bench_jxl_rgba8_hash():
    """samples/bench.jxl (54 groups, weighted predictor, per-group palettes + RCT) == bench.png, via its raw hash."""
    dec = O.decode(fixture_bytes("bench.jxl"))
    assert (dec.info.xsize, dec.info.ysize) == (2122, 1433)
    assert hashlib.sha256(dec.image("u8", 4).tobytes()).hexdigest() == MANIFEST["reference_fixtures"]["bench.jxl"]["sha256_rgba8"]


def test_sample_jpg_jxl_coefficients_match_the_jpeg():
    """VarDCT syntax / context model known answer: the quantised coefficients decoded from sample_jpg.jxl are those of
    samples/sample.jpg (Huffman-decoded independently by tests/golden/make_golden.py), chroma after the integer
    chroma-from-luma of SURVEY App. B.6."""
    g = np.load(os.path.join(GOLDEN, "sample_jpg_coefficients.npz"))
    coefs, qt = g["coefficients"].astype(np.int64), g["qtables"].astype(np.int64)
    dec = O.decode(fixture_bytes("sample_jpg.jxl"), dump=True)
    assert dec.info.have_container == 1 and dec.info.has_jbrd == 1 and dec.info.xyb_encoded == 0
    bw, bh = 5, 7
    planes = [dec.ints("coeff%d" % c)[: bw * bh * 64].reshape(bh, bw, 8, 8).transpose(0, 1, 3, 2).reshape(bh, bw, 64) for c in range(3)]  # libjxl layout = JPEG^T
    Y, Cb, Cr = planes[1].astype(np.int64), planes[0].astype(np.int64), planes[2].astype(np.int64)
    ytox, ytob = [int(v) for v in dec.ints("cfl")[:2]]
    assert (ytox, ytob) == (-15, 47)
    assert np.array_equal(Y[..., 1:], coefs[0][..., 1:])

    def undo(res, f, qc):
        out = res.copy()
        ff = (f * 2048) // 84 if f >= 0 else -((-f * 2048) // 84)
        for k in range(1, 64):
            scale = (2048 * qt[0][k] // qc[k]) * ff
            out[..., k] = res[..., k] + ((Y[..., k] * ((scale + 1024) >> 11) + 1024) >> 11)
        return out
    assert np.array_equal(undo(Cb, ytox, qt[1])[..., 1:], coefs[1][..., 1:])
    assert np.array_equal(undo(Cr, ytob, qt[2])[..., 1:], coefs[2][..., 1:])
    # LF = JPEG DC unchanged
    lfq = [dec.ints("lfq%d" % c).reshape(bh, bw) for c in range(3)]
    assert np.array_equal(lfq[1], coefs[0][..., 0]) and np.array_equal(lfq[0], coefs[1][..., 0]) and np.array_equal(lfq[2], coefs[2][..., 0])


def test_sample_jpg_jxl_planes_match_an_independent_float_decode():
    """RAW quant tables (layout and 1/(den*v) scaling), smart-dequant bias, float chroma-from-luma and the 8x8 IDCT, checked
    against scipy's orthonormal IDCT on the golden JPEG quant tables: the oracle's pre-colour-transform planes must equal
    idct(adj(q) * qtable [+ ratio * Y]) to float accuracy."""
    from scipy.fft import idctn
    g = np.load(os.path.join(GOLDEN, "sample_jpg_coefficients.npz"))
    qt = g["qtables"].astype(np.float64)
    dec = O.decode(fixture_bytes("sample_jpg.jxl"), dump=True)
    bw, bh = 5, 7
    q = [dec.ints("coeff%d" % c)[: bw * bh * 64].reshape(bh, bw, 8, 8).transpose(0, 1, 3, 2).astype(np.float64) for c in range(3)]
    lfq = [dec.ints("lfq%d" % c).reshape(bh, bw).astype(np.float64) for c in range(3)]
    bias, comp = [1 - 0.05465007330715401, 1 - 0.07005449891748593, 1 - 0.049935103337343655], [1, 0, 2]  # jxl channel -> JPEG component

    def adj(c, v):
        return np.where(np.abs(v) < 1.125, np.sign(v) * bias[c], v - 0.145 / np.where(v == 0, 1, v))
    deq = [adj(c, q[c]) * qt[comp[c]].reshape(8, 8) for c in range(3)]
    deq[0] = deq[0] + (-15 / 84.0) * deq[1]
    deq[2] = deq[2] + (47 / 84.0) * deq[1]
    for c in range(3):
        k = deq[c].copy()
        k[..., 0, 0] = lfq[c] * qt[comp[c]][0]
        want = idctn(k, axes=(2, 3), norm="ortho").transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8)
        got = dec.plane("idct%d" % c)[: bh * 8, : bw * 8] * 255.0
        assert np.abs(got - want).max() < 0.01


def test_sample_grey_jxl_decodes_to_the_grey_logo():
    """samples/sample_grey.jxl (reference test: jpegxl-rs/src/tests/decode.rs:83-93, image.rs:188-208) is the only XYB VarDCT
    stream in the reference written by the real encoder: ReferenceOnly Modular patch frame + VarDCT frame with patches, AFV0-3,
    DCT4x8/8x4/16x8, library quant tables, gaborish, EPF 1, gamma-0.45455 grey output.  No pixel golden exists, but the picture
    is the grey version of samples/sample.png: the decode must agree with its Rec.709 luma to lossy-codec accuracy (a wrong
    AFV basis, dequant table, patch placement or transfer function costs more than 10 dB)."""
    dec = O.decode(fixture_bytes("sample_grey.jxl"), dump=True)
    i = dec.info
    assert (i.xsize, i.ysize, i.bits_per_sample, i.num_color_channels, i.xyb_encoded) == (40, 50, 16, 1, 1)
    g = dec.image("u16", 1)
    assert g.shape == (50, 40, 1) and g.dtype == np.uint16          # tests/decode.rs:90 len == w * h, Pixels::Uint16
    png = read_png16(os.path.join(FIXTURES, "sample.png")).astype(np.float64) / 65535
    luma = png[..., :3] @ [0.2126, 0.7152, 0.0722]
    err = g[..., 0].astype(np.float64) / 65535 - luma
    psnr = -10 * np.log10((err ** 2).mean())
    assert psnr > 41.0, psnr                                         # measured 42.4 dB
    # SURVEY App. C structural known answers: strategies present, AC token count (1334 coefficient + 34*3 nzeros tokens)
    st = dec.ints("strategy")
    assert set(np.abs(st[st >= 0])) >= {0, 6, 12, 13, 14, 15, 16, 17} and (st >= 0).sum() == 34
    assert i.tokens_hf == 1334 + 34 * 3
    # per-strategy error: AFV blocks are as accurate as the DCT blocks around them
    blk = np.sqrt(np.array([[(err[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8] ** 2).mean() for bx in range(5)] for by in range(7)]))
    s = np.where(st >= 0, st, -1 - st).reshape(7, 5)
    assert blk[(s >= 14) & (s <= 17)].max() < 0.02 and blk.max() < 0.02
    # grey XYB invariants that bite without a golden: X == 0 exactly; the plain (un-folded) opsin inverse gives R = G and
    # |B - G| small, i.e. matrix and bias are mutually consistent; the shipped pixels come from the luminance-folded matrix
    X, Y, B = (dec.plane("patches%d" % c) for c in range(3))
    assert np.abs(X).max() == 0.0
    inv = np.array([[11.031566901960783, -9.866943921568629, -0.16462299647058826], [-3.254147380392157, 4.418770392156863, -0.16462299647058826],
                    [-3.6588512862745097, 2.7129230470588235, 1.9459282392156863]])
    bias = 0.0037930732552754493
    mixed = np.stack([(Y + X - np.cbrt(-bias)) ** 3 - bias, (Y - X - np.cbrt(-bias)) ** 3 - bias, (B - np.cbrt(-bias)) ** 3 - bias], -1).astype(np.float64)
    lin = mixed @ inv.T
    assert np.abs(lin[..., 0] - lin[..., 1]).max() < 1e-6 and np.abs(lin[..., 2] - lin[..., 1]).max() < 0.05
    want = np.clip(np.clip(lin @ [0.2126, 0.7152, 0.0722], 1e-5, None) ** 0.45455, 0, 1)
    assert np.abs(want - g[..., 0] / 65535.0).max() < 2e-4          # gamma via FastPowf: 3e-5 relative
    # the patch (5x6 from the 6x6 reference frame) lands twice: the two squares under the logo
    assert g[44:48, 2:5, 0].mean() < 45000 and g[44:48, 35:38, 0].mean() < 45000 and g[44:48, 8:14, 0].mean() > 60000


def test_2bit_jxl_renders_its_splines():
    """samples/2bit.jxl (reference test: tests/decode.rs:70-80): 2-bit RGB Modular frame whose samples are all 3 (white, a
    zero-bit prefix code) and 28 splines that draw the picture.  The decoder writes the full range of the output type
    (jpegxl-sys common/types.rs:107-129: 3 -> 255); the splines are rendered in float, so strokes take intermediate values."""
    dec = O.decode(fixture_bytes("2bit.jxl"), dump=True)
    i = dec.info
    assert (i.xsize, i.ysize, i.bits_per_sample, i.num_color_channels, i.xyb_encoded) == (800, 600, 2, 3, 0)
    px = dec.image("u8", 3)
    assert px.shape == (600, 800, 3)                                 # tests/decode.rs:77 len == w * h * 3, Pixels::Uint8
    assert np.all(dec.ints("modular0") == 3)
    white = (px == 255).all(-1)
    assert 0.90 < white.mean() < 0.99                                # a line drawing on white
    dark = (px < 64).all(-1)
    assert 0.005 < dark.mean() < 0.05
    assert white[:40].all() and white[-40:].all()                    # margins stay untouched
    # strokes are black: the three channels agree wherever the drawing is dark
    assert np.abs(px[dark].astype(int).max(-1) - px[dark].astype(int).min(-1)).max() <= 64


def test_recalled_tables_are_self_consistent():
    """Tables recalled from libjxl that carry their own checksum: the AFV basis is orthonormal, the default upsampling
    kernels (2x / 4x / 8x) are partitions of unity for every sub-pixel position."""
    import ctypes as C
    L = O.lib()
    L.jxlo_table.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_size_t)]

    def table(name):
        p = C.POINTER(C.c_float)(); n = C.c_size_t()
        assert L.jxlo_table(name, C.byref(p), C.byref(n))
        return np.ctypeslib.as_array(p, shape=(n.value,)).astype(np.float64)
    basis = table(b"afv_basis").reshape(16, 16)
    assert np.abs(basis @ basis.T - np.eye(16)).max() < 1e-6
    assert np.allclose(basis[0], 0.25)
    t = basis.reshape(16, 4, 4)
    for k in range(16):                                               # (anti)symmetric under transposition of the 4x4 block
        assert np.allclose(t[k], t[k].T, atol=1e-6) or np.allclose(t[k], -t[k].T, atol=1e-6)
    for name, N in ((b"up2", 1), (b"up4", 2), (b"up8", 4)):
        w = table(name)
        M = np.zeros((5 * N, 5 * N)); k = 0
        for y in range(5 * N):
            for x in range(y, 5 * N):
                M[y, x] = M[x, y] = w[k]; k += 1
        assert k == len(w)
        for ky in range(N):
            for kx in range(N):
                assert abs(M[5 * ky:5 * ky + 5, 5 * kx:5 * kx + 5].sum() - 1.0) < 2e-6


def test_fast_math_approximations():
    """base/fast_math-inl.h restated from memory: each rational approximation stays within libjxl's documented error of the
    exact function (a mis-remembered constant would be orders of magnitude off)."""
    import ctypes as C
    from math import erf
    L = O.lib()
    L.jxlo_fastmath.restype = C.c_float
    L.jxlo_fastmath.argtypes = [C.c_int, C.c_float, C.c_float]
    xs = np.linspace(0.01, 100, 4001)
    assert max(abs(L.jxlo_fastmath(0, x, 0) - np.log2(np.float32(x))) for x in xs) < 4e-6
    ys = np.linspace(-20, 20, 4001)
    assert max(abs(L.jxlo_fastmath(1, y, 0) / np.exp2(np.float64(np.float32(y))) - 1) for y in ys) < 4e-7
    assert max(abs(L.jxlo_fastmath(2, x, 0.45) / np.float64(np.float32(x)) ** 0.45 - 1) for x in xs) < 4e-5
    zs = np.linspace(-4, 4, 2001)
    assert max(abs(L.jxlo_fastmath(3, z, 0) - erf(np.float32(z))) for z in zs) < 7e-4
    ws = np.linspace(0, 100, 4001)
    assert max(abs(L.jxlo_fastmath(4, w, 0) - np.cos(np.float64(np.float32(w)))) for w in ws) < 2e-5


def test_invalid_and_truncated_inputs():
    for bad in (b"", b"\x00\x00", bytes(64), fixture_bytes("sample.jxl")[:100]):
        with pytest.raises(O.OracleError):
            O.decode(bad)


@pytest.mark.parametrize("name", [k for k in MANIFEST if k.startswith("vardct_")])
def test_vardct_regression_vectors(name):
    """Committed synthesised streams: the oracle's float pipeline output is pinned by hash (regression, not libjxl parity)."""
    m = MANIFEST[name]
    data = open(os.path.join(GOLDEN, name + ".jxl"), "rb").read()
    assert hashlib.sha256(data).hexdigest() == m["sha256_stream"]
    dec = O.decode(data)
    assert (dec.info.xsize, dec.info.ysize) == (m["width"], m["height"])
    assert hashlib.sha256(dec.pixels("u8", 3).tobytes()).hexdigest() == m["sha256_u8_rgb"]
    assert hashlib.sha256(dec.pixels("f32", 3).tobytes()).hexdigest() == m["sha256_f32_rgb"]


def test_modular_regression_vector():
    m = MANIFEST["modular_300x280_rgba16_rct"]
    data = open(os.path.join(GOLDEN, "modular_300x280_rgba16_rct.jxl"), "rb").read()
    dec = O.decode(data)
    assert hashlib.sha256(dec.image("u16", 4).astype("<u2").tobytes()).hexdigest() == m["sha256_u16_rgba_le"]


def test_output_layout_rules():
    """Buffer layout contract of decode.rs:387-434: stride rounded up to align, size = stride*(h-1)+w*C*bytes, big endian."""
    dec = O.decode(fixture_bytes("sample.jxl"))
    base = dec.pixels("f32", 3, big_endian=True, align=10)
    assert len(base) == 480 * 49 + 40 * 3 * 4                    # tests/decode.rs:164 (stride 480 is a multiple of 10)
    al = dec.pixels("u16", 3, align=64)
    stride = (40 * 3 * 2 + 63) // 64 * 64
    assert len(al) == stride * 49 + 40 * 3 * 2
    le = dec.pixels("u16", 4).view("<u2")
    be = dec.pixels("u16", 4, big_endian=True).view(">u2")
    assert np.array_equal(le.astype(np.uint16), be.astype(np.uint16))


def test_default_2x_upsampling_weights_are_a_partition_of_unity():
    """The recalled default 2x weights (oracle/render.h kDefaultUp2Weights) must behave like an interpolation kernel: the
    25 taps of every sub-pixel sum to 1, so a constant image stays constant; checked through a synthetic stream."""
    w = np.array([-0.01716200, -0.03452303, -0.04022174, -0.02921014, -0.00624645, 0.14111091, 0.28896755, 0.00278718,
                  -0.01610267, 0.56661550, 0.03777607, -0.01986694, -0.03144731, -0.01185068, -0.00213539])
    m = np.zeros((5, 5))
    k = 0
    for y in range(5):
        for x in range(y, 5):
            m[y, x] = m[x, y] = w[k]; k += 1
    assert abs(m.sum() - 1.0) < 1e-6
    import synth_lib as S
    flat = np.full((40, 56, 3), 128, np.uint8)
    px = O.decode(S.encode_vardct(flat, upsampling=2, epf_iters=0, gab=0)).pixels("u8", 3)
    assert np.abs(px.astype(int) - 128).max() <= 1


@pytest.mark.parametrize("name", [k for k in MANIFEST if k.startswith("vardct2_")])
def test_feature_regression_vectors(name):
    """Committed streams for alpha + progressive passes + permuted TOC, upsampling (default 2x / custom 4x weights), unaligned
    DCT128/256 varblocks and orientation: the oracle's output is pinned by hash (regression, not libjxl parity)."""
    m = MANIFEST[name]
    data = open(os.path.join(GOLDEN, name + ".jxl"), "rb").read()
    assert hashlib.sha256(data).hexdigest() == m["sha256_stream"]
    dec = O.decode(data)
    assert (dec.info.xsize, dec.info.ysize) == (m["width"], m["height"])
    assert hashlib.sha256(dec.pixels("u8", m["channels"]).tobytes()).hexdigest() == m["sha256_u8"]
    assert hashlib.sha256(dec.pixels("f32", m["channels"]).tobytes()).hexdigest() == m["sha256_f32"]


def test_squeeze_golden_stream_is_lossless():
    m = MANIFEST["modular2_333x300_ga16_squeeze"]
    data = open(os.path.join(GOLDEN, "modular2_333x300_ga16_squeeze.jxl"), "rb").read()
    assert hashlib.sha256(data).hexdigest() == m["sha256_stream"]
    assert hashlib.sha256(O.decode(data).pixels("u16", 2).tobytes()).hexdigest() == m["sha256_u16_ga_le"]


def test_idct_matches_direct_formula():
    """The recursive IDCT of the oracle against the defining sum f(n) = F0 + sqrt2 * sum F_k cos((2n+1) k pi / 2N)."""
    import ctypes as C
    rng = np.random.default_rng(1)
    for strategy, (R, Cc) in {0: (8, 8), 4: (16, 16), 6: (16, 8), 7: (8, 16), 5: (32, 32), 18: (64, 64), 19: (64, 32), 21: (128, 128),
                              22: (128, 64), 23: (64, 128), 24: (256, 256), 25: (256, 128), 26: (128, 256)}.items():
        sem = rng.standard_normal((R, Cc)).astype(np.float32)
        stored = (sem.T if R >= Cc else sem).reshape(-1).copy()
        out = np.zeros((R, Cc), np.float32)
        O.lib().jxlo_idct(strategy, stored.ctypes.data, out.ctypes.data, Cc)

        def basis(N):
            n = np.arange(N)[:, None]; k = np.arange(N)[None, :]
            b = np.cos((2 * n + 1) * k * np.pi / (2 * N)) * np.sqrt(2.0)
            b[:, 0] = 1.0
            return b
        ref = basis(R) @ sem.astype(np.float64) @ basis(Cc).T
        assert np.abs(out - ref).max() < 2e-4 * max(R, Cc)


def test_output_colour_encodings_against_independent_colour_science():
    """XYB images whose header names other primaries / a PQ or HLG transfer function (dec_xyb.cc OutputEncodingInfo::SetColorEncoding,
    stage_from_linear.cc OpPq / OpHlg incl. the inverse OOTF): the oracle's pixels, taken back to linear sRGB with textbook float64
    formulas (RGB -> XYZ matrices from the xy chromaticities, sRGB / ST 2084 / BT.2100 HLG EOTFs, OOTF gamma 1.2 at 1000 nits), must
    equal the oracle's linear-sRGB decode of the same codestream.  Pins the primaries adaptation (its D50 detour cancels), the recalled
    PQ rational-polynomial coefficients (they reproduce ST 2084 to 7e-7) and the HLG + OOTF chain."""
    import synth_lib as S
    img = S.synthetic_image(5, 96, 64)

    def dec(**kw):
        S.set_color(**kw)
        try:
            data = S.encode_vardct(img, seed=1)
        finally:
            S.set_color()
        return np.frombuffer(O.decode(data).pixels("f32", 3), np.float32).reshape(64, 96, 3).astype(np.float64)

    def rgb_to_xyz(prim, white):
        P = np.array([[prim[0], prim[2], prim[4]], [prim[1], prim[3], prim[5]], [1 - prim[0] - prim[1], 1 - prim[2] - prim[3], 1 - prim[4] - prim[5]]])
        W = np.array([white[0] / white[1], 1, (1 - white[0] - white[1]) / white[1]])
        return P * np.linalg.solve(P, W)
    srgb, p3, bt2020, d65 = (0.64, 0.33, 0.30, 0.60, 0.15, 0.06), (0.680, 0.320, 0.265, 0.690, 0.150, 0.060), (0.708, 0.292, 0.170, 0.797, 0.131, 0.046), (0.3127, 0.3290)

    def to_srgb(x, prim):
        return x @ (np.linalg.inv(rgb_to_xyz(srgb, d65)) @ rgb_to_xyz(prim, d65)).T

    def srgb_eotf(v):
        return np.where(v <= 0.04045, v / 12.92, ((np.abs(v) + 0.055) / 1.055) ** 2.4)

    def pq_eotf(e):
        m1, m2, c1, c2, c3 = 2610 / 16384, 2523 / 4096 * 128, 3424 / 4096, 2413 / 4096 * 32, 2392 / 4096 * 32
        p = np.abs(e) ** (1 / m2)
        return (np.maximum(p - c1, 0) / (c2 - c3 * p)) ** (1 / m1)

    def hlg_inverse_oetf(e):
        a = 0.17883277
        return np.where(e <= 0.5, e * e / 3, (np.exp((e - 0.5599107295) / a) + 1 - 4 * a) / 12)
    lin255 = dec(white_point=1, primaries=1, tf=8)
    lin1000 = dec(white_point=1, primaries=1, tf=8, intensity_target=1000.0)
    assert np.abs(to_srgb(srgb_eotf(dec(white_point=1, primaries=11, tf=13)), p3) - lin255).max() < 5e-5
    assert np.abs(to_srgb(srgb_eotf(dec(white_point=1, primaries=9, tf=13)), bt2020) - lin255).max() < 5e-5
    assert np.abs(to_srgb(pq_eotf(dec(white_point=1, primaries=9, tf=16, intensity_target=1000.0)) * 10, bt2020) - lin1000).max() < 2e-5
    assert np.abs(pq_eotf(dec(white_point=1, primaries=1, tf=16)) * (10000 / 255) - lin255).max() < 1e-4
    scene = hlg_inverse_oetf(dec(white_point=1, primaries=9, tf=18, intensity_target=1000.0))
    display = scene * ((scene @ rgb_to_xyz(bt2020, d65)[1]) ** 0.2)[..., None]
    assert np.abs(to_srgb(display, bt2020) - lin1000).max() < 2e-5
    # 300 nits: the system gamma is 0.9995, within the 0.01 band where libjxl skips the OOTF
    scene = hlg_inverse_oetf(dec(white_point=1, primaries=1, tf=18, intensity_target=300.0))
    assert np.abs(scene - dec(white_point=1, primaries=1, tf=8, intensity_target=300.0)).max() < 1e-5
    # PQ coefficients against ST 2084 itself
    import ctypes as C
    x = np.concatenate([np.logspace(-9, 0, 4000), [0.0]])
    m1, m2, c1, c2, c3 = 2610 / 16384, 2523 / 4096 * 128, 3424 / 4096, 2413 / 4096 * 32, 2392 / 4096 * 32
    exact = ((c1 + c2 * x ** m1) / (1 + c3 * x ** m1)) ** m2
    p, q = [1.351392e-02, -1.095778e+00, 5.522776e+01, 1.492516e+02, 4.838434e+01], [1.012416e+00, 2.016708e+01, 9.263710e+01, 1.120607e+02, 2.590418e+01]
    plo, qlo = [9.863406e-06, 3.881234e-01, 1.352821e+02, 6.889862e+04, -2.864824e+05], [3.371868e+01, 1.477719e+03, 1.608477e+04, -4.389884e+04, -2.072546e+05]
    t = x ** 0.25
    approx = np.where(x < 1e-4, np.polyval(plo[::-1], t) / np.polyval(qlo[::-1], t), np.polyval(p[::-1], t) / np.polyval(q[::-1], t))
    assert np.abs(approx - exact).max() < 2e-6


def test_float_samples_against_numpy():
    """Modular images with float samples (dec_modular.cc int_to_float): binary16 — every finite pattern incl. subnormals and -0 —, binary32
    and a 24-bit layout (1 + 7 + 16) come out of the oracle as exactly the floats numpy reads from the same bits."""
    import synth_lib as S
    rng = np.random.default_rng(3)
    h, w = 64, 96

    def dec(ints, bits, exp_bits):
        S.set_float(exp_bits)
        try:
            data = S.encode_modular(ints, bits, False, 0)
        finally:
            S.set_float(0)
        return np.frombuffer(O.decode(data).pixels("f32", 3), np.float32).reshape(h, w, 3)
    vals = np.concatenate([np.arange(0, 0x7C00), np.arange(0x8000, 0xFC00)]).astype(np.uint16)
    pat = rng.choice(vals, (h, w, 3))
    pat[0, :8, 0] = [0, 0x8000, 1, 0x8001, 0x03FF, 0x0400, 0x7BFF, 0xFBFF]
    assert np.array_equal(dec(pat.astype(np.int32), 16, 5).view(np.uint32), pat.view(np.float16).astype(np.float32).view(np.uint32))
    f = rng.random((h, w, 3), dtype=np.float32)
    f[0, :3, 0] = [0.0, 1.0, 1e-40]
    assert np.array_equal(dec(f.view(np.int32), 32, 8).view(np.uint32), f.view(np.uint32))
    pat24 = rng.integers(0, 1 << 23, (h, w, 3)).astype(np.int32)
    e, m = pat24 >> 16, pat24 & 0xFFFF
    assert np.array_equal(dec(pat24, 24, 7).astype(np.float64), np.where(e == 0, m * 2.0 ** (-62 - 16), (1 + m / 65536.0) * 2.0 ** (e - 63.0)))


def test_spot_colour_mixing_against_numpy():
    """stage_spot.cc in the oracle: colour = mix * spot + (1 - mix) * colour with mix = solidity * channel, on a lossless RGB image with a
    spot-colour extra channel; with rendering switched off the plain image comes out."""
    import synth_lib as S
    rng = np.random.default_rng(4)
    h, w = 40, 56
    img = rng.integers(0, 256, (h, w, 4)).astype(np.int32)
    spot = (1.0, 0.25, 0.125, 0.75)
    S.set_spot(spot)
    try:
        data = S.encode_modular(img, 8, False, 0)
    finally:
        S.set_spot()
    px = np.frombuffer(O.decode(data).pixels("f32", 3), np.float32).reshape(h, w, 3)
    rgb, s = img[..., :3].astype(np.float32) / np.float32(255), img[..., 3].astype(np.float32) / np.float32(255)
    mix = np.float32(spot[3]) * s
    assert np.abs(px - (mix[..., None] * np.array(spot[:3], np.float32) + (np.float32(1) - mix)[..., None] * rgb)).max() < 1e-6
    O.set_render_spotcolors(False)
    try:
        assert np.array_equal(np.frombuffer(O.decode(data).pixels("u8", 3), np.uint8).reshape(h, w, 3), img[..., :3].astype(np.uint8))
    finally:
        O.set_render_spotcolors(True)


def test_progression_steps_of_the_oracle_are_consistent():
    """The oracle's renders of progression steps (what the -m gpu progressive tests compare JxlDecoderFlushImage with): no pass = the kDC render, all passes = the full image,
    complete input under allow_truncated = the full image, a cut right behind pass k = the render of k passes, and every step changes pixels."""
    import synth_lib as S
    img = S.synthetic_image(31, 520, 300)
    data = S.encode_vardct(img, seed=31, strategy_mix=2, epf_iters=1, gab=1, num_passes=3, pass_ds=1)
    full = O.decode(data).pixels("u8", 3)
    dc = O.decode(data, dc_only=True).pixels("u8", 3)
    steps = [O.decode(data, max_passes=k).pixels("u8", 3) for k in range(4)]
    assert np.array_equal(steps[0], dc) and np.array_equal(steps[3], full)
    assert np.array_equal(O.decode(data, allow_truncated=True).pixels("u8", 3), full)
    for a, b in zip(steps, steps[1:]):
        assert not np.array_equal(a, b)
    # truncated input: cut points from the coarse end to the fine end show monotonically more; one byte short of the end still lacks the last group's last pass
    errs = []
    for frac in (0.55, 0.7, 0.85, 0.97):
        part = O.decode(data[:int(len(data) * frac)], allow_truncated=True).pixels("u8", 3)
        errs.append(float(np.abs(part.astype(np.int32) - full.astype(np.int32)).mean()))
    assert errs[-1] < errs[0] and errs[-1] > 0, errs
    with pytest.raises(O.OracleError):
        O.decode(data[:len(data) // 10], allow_truncated=True)           # inside the LF part: nothing to show
    assert not np.array_equal(O.decode(data[:-1], allow_truncated=True).pixels("u8", 3), full)
