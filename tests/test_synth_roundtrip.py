"""CPU tests: the bit-stream synthesiser (tools/jxlsynth) against the oracle — lossless Modular round trips exactly,
VarDCT round trips within the expected quantisation error for every supported block strategy."""
import numpy as np
import pytest

import oracle_lib as O
import synth_lib as S


def psnr(a, b):
    m = ((a.astype(np.float64) - b.astype(np.float64)) ** 2).mean()
    return 10 * np.log10(255.0 ** 2 / max(m, 1e-12))


@pytest.mark.parametrize("h,w,c,bits,rct", [(50, 40, 3, 8, False), (1, 1, 3, 8, False), (7, 300, 1, 8, False), (300, 520, 4, 8, True),
                                            (64, 64, 1, 16, False), (257, 255, 3, 16, True), (256, 256, 3, 8, True), (513, 10, 2, 8, False)])
def test_modular_lossless_roundtrip(h, w, c, bits, rct):
    base = S.synthetic_image(3, max(w, 8), max(h, 8)).astype(np.int32)[:h, :w]
    img = np.stack([base[..., i % 3] * ((1 << bits) - 1) // 255 for i in range(c)], -1)
    dec = O.decode(S.encode_modular(img, bits, rct))
    assert (dec.info.xsize, dec.info.ysize, dec.info.bits_per_sample) == (w, h, bits)
    out = dec.image("u16" if bits == 16 else "u8", c)
    assert np.array_equal(out, img)


STRATEGIES = {0: "DCT", 1: "IDENTITY", 2: "DCT2X2", 3: "DCT4X4", 4: "DCT16X16", 5: "DCT32X32", 6: "DCT16X8", 7: "DCT8X16", 8: "DCT32X8", 9: "DCT8X32",
              10: "DCT32X16", 11: "DCT16X32", 12: "DCT4X8", 13: "DCT8X4", 18: "DCT64X64", 19: "DCT64X32", 20: "DCT32X64"}


@pytest.mark.parametrize("s", sorted(STRATEGIES))
def test_vardct_every_strategy_roundtrip(s):
    """Forced single strategy at fine quantisation: forward transform (synth) x inverse transform (oracle) must agree."""
    img = S.synthetic_image(7, 256, 128)
    data = S.encode_vardct(img, seed=5, strategy_mix=100 + s, epf_iters=0, gab=0, distance=0.1, skip_lf_smoothing=1)
    out = O.decode(data).image("u8", 3)
    assert psnr(out, img) > 41.0, STRATEGIES[s]


@pytest.mark.parametrize("w,h,mix,epf,gab", [(8, 8, 0, 0, 0), (64, 64, 0, 1, 1), (100, 37, 1, 1, 1), (256, 256, 1, 2, 1), (300, 200, 2, 3, 1), (520, 300, 2, 0, 0)])
def test_vardct_roundtrip_quality(w, h, mix, epf, gab):
    img = S.synthetic_image(11, max(w, 8), max(h, 8))[:h, :w]
    img = np.ascontiguousarray(img)
    dec = O.decode(S.encode_vardct(img, seed=2, strategy_mix=mix, epf_iters=epf, gab=gab))
    assert (dec.info.xsize, dec.info.ysize, dec.info.xyb_encoded) == (w, h, 1)
    assert psnr(dec.image("u8", 3), img) > 33.0
    assert dec.info.tokens_hf > 0 and dec.info.tokens_lf > 0


def test_synth_is_deterministic():
    img = S.synthetic_image(5, 128, 64)
    assert np.array_equal(img, S.synthetic_image(5, 128, 64))
    a = S.encode_vardct(img, seed=1, strategy_mix=2)
    assert a == S.encode_vardct(img, seed=1, strategy_mix=2)
    assert a != S.encode_vardct(img, seed=2, strategy_mix=2)


def test_hdr_float_stream_roundtrip():
    """Config-5 style stream: float32 samples, linear transfer, intensity_target 1000."""
    img = S.synthetic_image(9, 160, 96).astype(np.float32) / 255.0
    lin = (img ** 2.2) * 2.0
    data = S.encode_vardct(lin, seed=4, strategy_mix=1, epf_iters=3, gab=1, out_bits=32, hdr=1)
    dec = O.decode(data)
    assert dec.info.bits_per_sample == 32 and dec.info.exponent_bits == 8 and abs(dec.info.intensity_target - 1000.0) < 1e-3
    out = dec.image("f32", 3)
    assert np.abs(out - lin).mean() < 0.02
