"""GPU parity tests (-m gpu) of round 6's wave-wide entropy decoders (kernels.hip DecodeChannelWave / DecodeChannelWaveGen / HfDecodeWaveKernel) in the launch shapes
the other tests do not reach: LfDecodeKernel with four LF groups per workgroup (what a cold pipeline of 256 frames launches — a variant of this round that made
DecodeChannelCoop a real call passed every other test and failed there), the wave-wide HF kernel asked for explicitly on a batch, the decoders' fall-back on samples
beyond their 32-bit arithmetic, and trees at the limits of the shapes they take.  Expected values: the CPU oracle, bit-exact."""
import numpy as np
import pytest

import oracle_lib as O
import synth_lib as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def jx(built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import jpegxl_rs_amd as jx
    return jx


def batch_decode(jx, streams, lf_stride, hf_stride, options):
    b = jx.BatchDecoder(0)
    b.add_many(streams, "uint8", 3, threads=4)
    b.set_lane_stride(lf_stride, hf_stride)
    b.prepare()
    for k, v in options.items():
        b.set_option(k, v)
    b.decode(); b.finish()
    return [b.output(i) for i in range(len(streams))]


@pytest.fixture(scope="module")
def six_lf_groups():
    """a frame of 3 x 2 LF groups (4200 x 2100) under the gradient LF tree and under the weighted-predictor (cjxl default-effort) one, with the oracle's pixels"""
    img = S.synthetic_image(61, 4200, 2100)
    out = []
    for shape in (0, 1):
        S.set_lf_tree_shape(shape)
        try:
            d = S.encode_vardct(img, seed=7 + shape, strategy_mix=2, epf_iters=1, gab=1)
        finally:
            S.set_lf_tree_shape(0)
        out.append((d, O.decode(d).pixels("u8", 3)))
    return out


@pytest.mark.parametrize("force_big", [1, 2, -1])
def test_lf_kernel_launch_shapes(jx, six_lf_groups, force_big):
    """LfDecodeKernel: four groups per workgroup (a wavefront decodes two LF groups one after the other) under the register-capped and the uncapped instantiation, and
    one group per wavefront — same pixels"""
    streams = [d for d, _ in six_lf_groups]
    got = batch_decode(jx, streams, 64, 1, {"lf_force_big": force_big})
    for (d, ref), px in zip(six_lf_groups, got):
        assert np.array_equal(px.reshape(ref.shape), ref), f"lf_force_big {force_big}: {int((px.reshape(ref.shape) != ref).sum())} samples differ"


@pytest.mark.parametrize("lanes_per_wave", [1, 4, 0])
def test_hf_kernels_agree(jx, six_lf_groups, lanes_per_wave):
    """the wave-wide HF kernel (one group stream per wavefront), the sparse and the dense SIMT packing: same coefficients"""
    streams = [d for d, _ in six_lf_groups]
    got = batch_decode(jx, streams, 64, 1, {"hf_lanes_per_wave": lanes_per_wave})
    for (d, ref), px in zip(six_lf_groups, got):
        assert np.array_equal(px.reshape(ref.shape), ref), f"hf_lanes_per_wave {lanes_per_wave}"


def test_wave_hf_strategies_and_bit_rates(jx):
    """every transform mix of the synthesiser, EPF 0-3, low and high bit rates through the wave-wide HF kernel (tokens with extra bits, blocks whose order table goes
    beyond its 64-entry head, empty blocks)"""
    cases = []
    for seed, (w, h, mix, epf, dist) in enumerate([(530, 400, 0, 0, 1.0), (640, 300, 1, 1, 0.3), (300, 620, 2, 2, 3.0), (1030, 520, 3, 3, 1.0), (260, 260, 2, 1, 8.0), (1300, 300, 1, 1, 0.1)]):
        img = S.synthetic_image(100 + seed, w, h)
        cases.append(S.encode_vardct(img, seed=seed + 1, strategy_mix=mix, epf_iters=epf, gab=1, distance=dist))
    got = batch_decode(jx, cases, 64, 1, {"hf_lanes_per_wave": 1})
    for d, px in zip(cases, got):
        ref = O.decode(d).pixels("u8", 3)
        assert np.array_equal(px.reshape(ref.shape), ref)


def test_wave_lf_falls_back_on_large_samples(jx):
    """quantised LF values beyond +-4095 under a weighted-predictor tree: the wave-wide decoder's 32-bit arithmetic gives the channel back to the general path (quant_lf at
    its finest step makes the LF samples large)"""
    img = S.synthetic_image(77, 520, 300)
    S.set_quant_lf(5000); S.set_lf_tree_shape(1)
    try:
        d = S.encode_vardct(img, seed=3, strategy_mix=2, epf_iters=1, gab=1)
    finally:
        S.set_quant_lf(); S.set_lf_tree_shape(0)
    ref = O.decode(d).pixels("u8", 3)
    meta, px = jx.decoder_builder().decode_with(d, np.uint8)
    assert np.array_equal(px.reshape(ref.shape), ref)


def test_corrupted_round6_streams_fail_cleanly_or_decode(jx):
    """Robustness of the wave-wide decoders: random byte corruption / truncation of streams that take them — a weighted-predictor (cjxl-shaped) LF tree, the reference's bench.jxl
    (big MA tree: the register-row tree walk), free-running Modular streams with deep trees — ends in a DecodeError or a decode of the right size, never a crash or a hang
    (every walk is bounded by the tree's node count, every sample loop by the channel geometry; reads past a section's end give zeros)."""
    from conftest import fixture_bytes
    rng = np.random.default_rng(606)
    img = S.synthetic_image(78, 520, 300)
    S.set_lf_tree_shape(1)
    try:
        wp = S.encode_vardct(img, seed=5, strategy_mix=2, epf_iters=1, gab=1)
    finally:
        S.set_lf_tree_shape(0)
    streams = [(wp, 24), (fixture_bytes("bench.jxl"), 12),
               (S.encode_modular_free(seed=41, w=300, h=260, nchan=3, bits=8, tree_flags=S.TREE_WP | S.TREE_PREV_CHANNELS, tree_depth=9), 16),
               (S.encode_modular_free(seed=42, w=300, h=260, nchan=3, bits=12, tree_flags=31, tree_depth=7), 16)]
    outcomes = {"error": 0, "decoded": 0}
    total = 0
    for data, trials in streams:
        for trial in range(trials):
            bad = bytearray(data)
            for pos in rng.integers(len(bad) // 8, len(bad), 1 + trial % 4):        # (past the headers: the decode reaches the GPU)
                bad[pos] ^= 1 << int(rng.integers(0, 8))
            if trial % 6 == 5:
                bad = bad[: int(rng.integers(len(bad) // 2, len(bad)))]
            total += 1
            try:
                meta, px = jx.decoder_builder().decode_with(bytes(bad), np.uint8)
                assert len(px) == meta.width * meta.height * (4 if meta.has_alpha_channel else 3)
                outcomes["decoded"] += 1
            except jx.DecodeError:
                outcomes["error"] += 1
    assert outcomes["error"] > 0 and outcomes["error"] + outcomes["decoded"] == total
    ref = O.decode(wp).pixels("u8", 3)                                       # the decoder is still healthy afterwards
    meta, px = jx.decoder_builder().decode_with(wp, np.uint8)
    assert np.array_equal(px.reshape(ref.shape), ref)


def _smooth(seed, h, w, c, bits):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = ((np.sin(xx / 37.0) + np.cos(yy / 23.0)) * 0.25 + 0.5) * ((1 << bits) - 1)
    return np.clip(base[..., None] + rng.normal(0, (1 << bits) / 1024.0, (h, w, c)).astype(np.float32), 0, (1 << bits) - 1).astype(np.int32)


def test_inverse_squeeze_steps_of_several_images_in_one_launch(jx):
    """Batch::EnqueueModularTail: images whose inverse-transform chains have the same shape go through them together (SqueezeBatch: step k of up to 16 images in one launch),
    others on their own — 18 squeezed Modular images of one size (two launches' worth), three of other sizes / channel counts and one without Squeeze in one batch: every
    image as the oracle decodes it"""
    imgs = [(_smooth(200 + i, 300, 520, 1, 16), 16, False, 1) for i in range(18)]
    imgs += [(_smooth(300, 200, 136, 1, 16), 16, False, 1), (_smooth(301, 300, 520, 3, 8), 8, True, 1), (_smooth(302, 300, 520, 1, 16), 16, False, 2), (_smooth(303, 300, 520, 1, 16), 16, False, 0)]
    streams = [S.encode_modular(a, bits, rct, sq) for a, bits, rct, sq in imgs]
    b = jx.BatchDecoder(0)
    b.add_many(streams, "uint16", 0, threads=4)
    b.prepare(); b.decode(); b.finish()
    for i, d in enumerate(streams):
        ref = O.decode(d)
        nch = 3 if imgs[i][0].shape[2] == 3 else 1
        assert np.array_equal(b.output(i).view(np.uint16).reshape(-1), ref.pixels("u16", nch).view(np.uint16).reshape(-1)), f"image {i}"


def test_chroma_subsampled_frames_in_one_job(jx):
    """chroma-subsampled YCbCr frames take the plain path since round 6 (IdctSubsampledTileKernel, OutputKernel's upsampling branch): a job that mixes the subsampling modes,
    odd sizes and a plain XYB frame on shared planes, u8 and f32 — every frame as the oracle decodes it"""
    cases = [("420", 530, 300), ("422", 203, 139), ("440", 200, 136), ("mixed", 300, 260), ("444", 260, 300), ("420", 2100, 270), ("420", 64, 48)]
    streams = [S.encode_ycbcr(S.synthetic_image(80 + i, w, h), sub, seed=w + h, distance=0.7) for i, (sub, w, h) in enumerate(cases)]
    streams.append(S.encode_vardct(S.synthetic_image(99, 520, 300), seed=3, strategy_mix=2, epf_iters=1, gab=1))
    for dtype, kind in ((np.uint8, "u8"), (np.float32, "f32")):
        b = jx.BatchDecoder(0)
        b.add_many(streams, np.dtype(dtype).name, 3, threads=4)
        b.prepare(); b.decode(); b.finish()
        for i, d in enumerate(streams):
            ref = O.decode(d).pixels(kind, 3)
            got = b.output(i).view(dtype).reshape(-1)
            assert np.array_equal(got.view(np.uint8), np.ascontiguousarray(ref).reshape(-1).view(np.uint8)), f"frame {i} ({kind})"
