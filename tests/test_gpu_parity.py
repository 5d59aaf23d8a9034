"""GPU parity tests (-m gpu): the HIP decode path, called through the libjxl-compatible C ABI exactly like
jpegxl-rs/src/decode.rs drives libjxl, must reproduce the CPU oracle bit-exactly for integer outputs and (so far also)
exactly for f32 — tolerance: u8/u16 exact, f32 <= 1 ULP (BASELINE.json north_star).  Mirrors the reference's decode tests
(jpegxl-rs/src/tests/decode.rs, errors.rs, image.rs) plus synthesised VarDCT streams for the path north_star names."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import FIXTURES, GOLDEN, fixture_bytes, read_png16
import oracle_lib as O
import synth_lib as S

pytestmark = pytest.mark.gpu
MANIFEST = json.load(open(os.path.join(GOLDEN, "manifest.json")))


@pytest.fixture(scope="module")
def jx(built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import jpegxl_rs_amd as jx
    return jx


def ulp_diff(a, b):
    ai = a.view(np.int32).astype(np.int64); bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai); bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    return np.abs(ai - bi).max() if a.size else 0


def check_against_oracle(jx, data, dtype, nch=0, **fmt):
    meta, px = jx.decoder_builder(pixel_format=jx.PixelFormat(num_channels=nch, **fmt)).decode_with(data, dtype)
    kind = {"uint8": "u8", "uint16": "u16", "float32": "f32", "float16": "f16"}[np.dtype(dtype).name]
    ref = O.decode(data).pixels(kind, nch, big_endian=fmt.get("endianness", 0) == 2, align=fmt.get("align", 0))
    order = ">" if fmt.get("endianness", 0) == 2 else "<"
    ref = ref.view(np.dtype(order + np.dtype(dtype).str[1:])).astype(dtype)
    assert px.shape == ref.shape
    if np.dtype(dtype) == np.float32:
        assert ulp_diff(px, ref) <= 1
    else:
        assert np.array_equal(px, ref), f"{int((px != ref).sum())} of {px.size} samples differ"
    return meta, px


# ---- the reference's own fixtures and tests -------------------------------------------------------------------------
def test_sample_jxl_equals_sample_png(jx):
    """image.rs:157-172 through the HIP path: RGBA16 == sample.png."""
    meta, px = jx.decoder_builder(parallel_runner=jx.ThreadsRunner()).decode_with(fixture_bytes("sample.jxl"), np.uint16)
    assert (meta.width, meta.height, meta.num_color_channels, meta.has_alpha_channel) == (40, 50, 3, True)
    png = read_png16(os.path.join(FIXTURES, "sample.png"))
    assert np.array_equal(px.reshape(50, 40, 4), png)
    assert hashlib.sha256(px.astype(">u2").tobytes()).hexdigest() == MANIFEST["reference_fixtures"]["sample.jxl"]["sha256_rgba16_be"]


def test_image_integration(jx):
    """jpegxl-rs/src/image.rs:145-225 (tests invalid / simple / pixel_type of the `image` integration): decode_to_image(SAMPLE_JXL).to_rgba16()
    == sample.png as RGBA16; which (sample type, channel count) pairs have a DynamicImage variant."""
    dec = jx.decoder_builder(parallel_runner=jx.ThreadsRunner())
    for bad_call in (lambda: dec.decode_to_image(b""), lambda: dec.decode_to_image_with(b"", np.float32)):
        with pytest.raises(jx.DecodeError):
            bad_call()
    sample, grey = fixture_bytes("sample.jxl"), fixture_bytes("sample_grey.jxl")
    img = dec.decode_to_image(sample)
    assert img is not None
    assert np.array_equal(img.to_rgba16(), read_png16(os.path.join(FIXTURES, "sample.png")))
    assert dec.decode_to_image_with(sample, np.float16) is None
    for nch, data, f32_ok in ((1, grey, False), (2, grey, False), (3, sample, True), (4, sample, True)):
        d = jx.decoder_builder(parallel_runner=jx.ThreadsRunner(), pixel_format=jx.PixelFormat(num_channels=nch))
        for t, ok in ((np.uint8, True), (np.uint16, True), (np.float32, f32_ok)):
            got = d.decode_to_image_with(data, t)
            assert (got is not None) == ok, (nch, t)
            if ok:
                assert got.pixels.shape[2] == nch and got.pixels.dtype == np.dtype(t)


def test_decode_simple_and_pixel_types(jx):
    """tests/decode.rs:45-67,96-120: inferred type Uint16, len == w*h*4; every pixel type and endianness succeeds."""
    data = fixture_bytes("sample.jxl")
    meta, px = jx.decoder_builder().decode(data)
    assert px.dtype == np.uint16 and len(px) == meta.width * meta.height * 4 and meta.icc_profile is None
    # decode.rs:46,64: icc_profile(true) -> a profile lcms2 accepts (PIL.ImageCms is an lcms2 binding)
    import io
    from PIL import ImageCms
    meta, px2 = jx.decoder_builder(icc_profile=True).decode(data)
    assert np.array_equal(px, px2) and meta.icc_profile == jx.icc_profile_from_headers(data)
    assert ImageCms.ImageCmsProfile(io.BytesIO(meta.icc_profile)).profile.xcolor_space.strip() == "RGB"
    for dt in (np.float32, np.uint8, np.uint16, np.float16):
        for en in (jx.Endianness.Big, jx.Endianness.Little, jx.Endianness.Native):
            _, p = jx.decoder_builder(pixel_format=jx.PixelFormat(endianness=en)).decode_with(data, dt)
            assert len(p) == 40 * 50 * 4 and p.dtype == np.dtype(dt)
    check_against_oracle(jx, data, np.uint8, 3)
    check_against_oracle(jx, data, np.uint8, 4)
    check_against_oracle(jx, data, np.uint16, 4, endianness=2)
    check_against_oracle(jx, data, np.float32, 3)
    check_against_oracle(jx, data, np.uint16, 4)


def test_builder_reuse(jx):
    """tests/decode.rs:142-180: one decoder object re-used with mutated options; f32 RGB big-endian align 10 then RGBA."""
    data = fixture_bytes("sample.jxl")
    dec = jx.decoder_builder(pixel_format=jx.PixelFormat(num_channels=3, endianness=jx.Endianness.Big, align=10), skip_reorientation=True,
                             unpremul_alpha=True, render_spotcolors=True, coalescing=True, desired_intensity_target=0.5, decompress=True,
                             init_jpeg_buffer=512, parallel_runner=jx.ResizableRunner())
    meta, px = dec.decode_with(data, np.float32)
    assert len(px) == meta.width * meta.height * 3
    dec.pixel_format = None
    dec.parallel_runner = jx.ThreadsRunner()
    meta, px = dec.decode_with(data, np.float32)
    assert len(px) == meta.width * meta.height * 4
    meta, px = dec.decode(data)
    assert px.dtype == np.uint16


def test_decode_errors(jx):
    """errors.rs:109-137: truncated input after CloseInput -> GenericError; garbage -> InvalidInput."""
    dec = jx.decoder_builder()
    with pytest.raises(jx.InvalidInput):
        dec.decode(b"")
    with pytest.raises(jx.InvalidInput):
        dec.decode(bytes(64))
    with pytest.raises(jx.GenericError):
        jx.decoder_builder().decode(fixture_bytes("sample.jxl")[:100])
    data = bytearray(S.encode_vardct(S.synthetic_image(1, 64, 64), strategy_mix=0))
    data[len(data) // 2] ^= 0xFF                      # corrupt the token stream: ANS final-state / range checks must fire
    with pytest.raises(jx.GenericError):
        jx.decoder_builder().decode_with(bytes(data), np.uint8)


def test_sample_gray(jx):
    """tests/decode.rs:83-93 (sample_gray) and image.rs:183-209: the real-encoder XYB stream (ReferenceOnly patch frame + VarDCT
    frame with patches, AFV, gaborish, EPF, gamma-0.45455 grey) decodes to Pixels::Uint16 of len w*h, bit-identical to the
    oracle, and is the grey version of sample.png."""
    meta, px = jx.decoder_builder().decode(fixture_bytes("sample_grey.jxl"))
    assert (meta.width, meta.height, meta.num_color_channels, meta.has_alpha_channel) == (40, 50, 1, False)
    assert px.dtype == np.uint16 and len(px) == 40 * 50
    ref = O.decode(fixture_bytes("sample_grey.jxl"))
    assert np.array_equal(px, ref.pixels("u16", 1).view(np.uint16))
    for dt, nch in ((np.uint8, 1), (np.uint8, 3), (np.float32, 1), (np.float32, 3), (np.uint16, 4)):
        check_against_oracle(jx, fixture_bytes("sample_grey.jxl"), dt, nch)
    luma = (read_png16(os.path.join(FIXTURES, "sample.png")).astype(np.float64) / 65535)[..., :3] @ [0.2126, 0.7152, 0.0722]
    err = px.reshape(50, 40).astype(np.float64) / 65535 - luma
    assert -10 * np.log10((err ** 2).mean()) > 41.0


def test_sample_2bit(jx):
    """tests/decode.rs:70-80 (sample_2bit): Modular frame + 28 splines -> Pixels::Uint8 of len w*h*3, bit-identical to the oracle."""
    meta, px = jx.decoder_builder().decode(fixture_bytes("2bit.jxl"))
    assert (meta.width, meta.height, meta.num_color_channels) == (800, 600, 3)
    assert px.dtype == np.uint8 and len(px) == 800 * 600 * 3
    ref = O.decode(fixture_bytes("2bit.jxl"))
    assert np.array_equal(px, ref.pixels("u8", 3))
    check_against_oracle(jx, fixture_bytes("2bit.jxl"), np.float32, 3)
    check_against_oracle(jx, fixture_bytes("2bit.jxl"), np.uint16, 4)


def test_raw_ffi_sequence(jx):
    """jpegxl-sys/src/lib.rs:85-171: raw state machine without CloseInput / Reset, u8 x 3 channels, 40 x 50."""
    L = jx.libjxl()
    dec = L.JxlDecoderCreate(None)
    data = np.frombuffer(fixture_bytes("sample.jxl"), np.uint8)
    assert L.JxlDecoderSubscribeEvents(dec, jx.JXL_DEC_BASIC_INFO | jx.JXL_DEC_FULL_IMAGE) == 0
    assert L.JxlDecoderSetInput(dec, data.ctypes.data, len(data)) == 0
    info = jx.JxlBasicInfo()
    fmt = jx.JxlPixelFormat(3, jx.JXL_TYPE_UINT8, jx.JXL_NATIVE_ENDIAN, 0)
    buf = None
    events = []
    while True:
        st = L.JxlDecoderProcessInput(dec)
        events.append(st)
        if st == jx.JXL_DEC_BASIC_INFO:
            assert L.JxlDecoderGetBasicInfo(dec, C.byref(info)) == 0
            assert (info.xsize, info.ysize) == (40, 50)
        elif st == jx.JXL_DEC_NEED_IMAGE_OUT_BUFFER:
            size = C.c_size_t()
            assert L.JxlDecoderImageOutBufferSize(dec, C.byref(fmt), C.byref(size)) == 0
            assert size.value == 40 * 50 * 3
            buf = np.zeros(size.value, np.uint8)
            assert L.JxlDecoderSetImageOutBuffer(dec, C.byref(fmt), buf.ctypes.data, size.value - 1) == 1   # too small
            assert L.JxlDecoderSetImageOutBuffer(dec, C.byref(fmt), buf.ctypes.data, size.value) == 0
        elif st == jx.JXL_DEC_FULL_IMAGE:
            continue
        elif st == jx.JXL_DEC_SUCCESS:
            break
        else:
            raise AssertionError(st)
    assert events == [jx.JXL_DEC_BASIC_INFO, jx.JXL_DEC_NEED_IMAGE_OUT_BUFFER, jx.JXL_DEC_FULL_IMAGE, jx.JXL_DEC_SUCCESS]
    L.JxlDecoderDestroy(dec)
    assert np.array_equal(buf, O.decode(fixture_bytes("sample.jxl")).pixels("u8", 3))


def test_coalesced_animation_frames(jx):
    """An animation decoded with coalescing (the default): every frame that is shown arrives as JXL_DEC_FRAME (JxlDecoderGetFrameHeader: duration,
    is_last, the image's size) -> JXL_DEC_NEED_IMAGE_OUT_BUFFER -> JXL_DEC_FULL_IMAGE with the canvas as it stands after that frame; frames of
    duration 0 are layers of the frame behind them.  Expected canvases: the oracle's decode of the stream cut after the frame (the frame written as
    the last one).  jpegxl-rs's loop (decode.rs:207-325) refills its buffer per frame and so returns the last canvas."""
    L = jx.libjxl()
    big, small, img = S.synthetic_image(6, 600, 400), S.synthetic_image(9, 64, 48), S.synthetic_image(5, 200, 136)
    S.set_animation(100, 1, 0)
    try:
        def f0(last): return S.encode_vardct_frame(big, S.frame(is_last=last, save_as_reference=0 if last else 1, duration=10), seed=3, strategy_mix=2)
        def f1(last, dur): return S.encode_vardct_frame(small, S.frame(emit=1, is_last=last, have_crop=1, crop_x0=300, crop_y0=160, canvas_w=600, canvas_h=400, blend_mode=1, blend_source=1,
                                                                        save_as_reference=0 if last else 1, duration=dur), seed=4)
        f2 = S.encode_vardct_frame(img, S.frame(emit=1, have_crop=1, crop_x0=-30, crop_y0=300, canvas_w=600, canvas_h=400, blend_mode=0, blend_source=1, duration=7), seed=5, epf_iters=2)
        full, upto0, upto1 = f0(0) + f1(0, 5) + f2, f0(1), f0(0) + f1(1, 5)
        layered = f0(0) + f1(0, 0) + f2                     # the middle frame is a layer of the last one: two frames are shown
    finally:
        S.set_animation(0)
    want = [O.decode(s_).pixels("u8", 3) for s_ in (upto0, upto1, full)]
    fmt = jx.JxlPixelFormat(3, jx.JXL_TYPE_UINT8, jx.JXL_NATIVE_ENDIAN, 0)

    def run(stream):
        data = np.frombuffer(stream, np.uint8)
        dec = L.JxlDecoderCreate(None)
        assert L.JxlDecoderSubscribeEvents(dec, jx.JXL_DEC_BASIC_INFO | jx.JXL_DEC_FRAME | jx.JXL_DEC_FULL_IMAGE) == 0
        assert L.JxlDecoderSetInput(dec, data.ctypes.data, len(data)) == 0
        L.JxlDecoderCloseInput(dec)
        got, headers, buf = [], [], None
        while True:
            st = L.JxlDecoderProcessInput(dec)
            if st == jx.JXL_DEC_BASIC_INFO:
                info = jx.JxlBasicInfo()
                assert L.JxlDecoderGetBasicInfo(dec, C.byref(info)) == 0
                assert info.have_animation == 1 and (info.xsize, info.ysize) == (600, 400)
            elif st == jx.JXL_DEC_FRAME:
                fh = jx.JxlFrameHeader()
                assert L.JxlDecoderGetFrameHeader(dec, C.byref(fh)) == 0
                headers.append((fh.duration, fh.is_last, fh.layer_info.xsize, fh.layer_info.ysize))
            elif st == jx.JXL_DEC_NEED_IMAGE_OUT_BUFFER:
                size = C.c_size_t()
                assert L.JxlDecoderImageOutBufferSize(dec, C.byref(fmt), C.byref(size)) == 0 and size.value == 600 * 400 * 3
                buf = np.zeros(size.value, np.uint8)
                assert L.JxlDecoderSetImageOutBuffer(dec, C.byref(fmt), buf.ctypes.data, size.value) == 0
            elif st == jx.JXL_DEC_FULL_IMAGE:
                got.append(buf); buf = None
            elif st == jx.JXL_DEC_SUCCESS:
                break
            else:
                raise AssertionError(st)
        L.JxlDecoderDestroy(dec)
        return got, headers

    got, headers = run(full)
    assert headers == [(10, 0, 600, 400), (5, 0, 600, 400), (7, 1, 600, 400)]
    assert len(got) == 3
    for k in range(3):
        assert np.array_equal(got[k], want[k]), k
    got, headers = run(layered)
    assert headers == [(10, 0, 600, 400), (7, 1, 600, 400)] and len(got) == 2
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], O.decode(layered).pixels("u8", 3))
    _, px = jx.decoder_builder().decode_with(full, np.uint8)
    assert np.array_equal(px.reshape(-1), want[2])


def test_preview_frames_are_stepped_over(jx):
    """An image with a preview (headers.cc PreviewHeader; a frame of that size in front of the image's frames): jpegxl-rs never subscribes to
    JXL_DEC_PREVIEW_IMAGE (decode.rs:334-347), so the preview is stepped over and the image decodes to what it decodes to without one;
    JxlBasicInfo reports have_preview and the preview's size (jpegxl-sys codestream_header.rs:38-241)."""
    from test_synth_roundtrip import preview_streams
    L = jx.libjxl()
    cases = preview_streams()
    for name, with_preview, plain, (pw, ph), _ in cases:
        ref = O.decode(plain).pixels("u8", 3)
        _, px = check_against_oracle(jx, with_preview, np.uint8, 3)
        assert np.array_equal(px.reshape(-1), ref), name
        check_against_oracle(jx, with_preview, np.float32, 3)
        dec = L.JxlDecoderCreate(None)
        data = np.frombuffer(with_preview, np.uint8)
        assert L.JxlDecoderSubscribeEvents(dec, jx.JXL_DEC_BASIC_INFO) == 0
        assert L.JxlDecoderSetInput(dec, data.ctypes.data, len(data)) == 0
        assert L.JxlDecoderProcessInput(dec) == jx.JXL_DEC_BASIC_INFO
        info = jx.JxlBasicInfo()
        assert L.JxlDecoderGetBasicInfo(dec, C.byref(info)) == 0
        assert (info.have_preview, info.preview_xsize, info.preview_ysize, info.xsize, info.ysize) == (1, pw, ph, 300, 200), name
        L.JxlDecoderDestroy(dec)
    b = jx.BatchDecoder(0)
    for name, with_preview, plain, _, _ in cases:
        b.add(with_preview, "uint8", 3); b.add(plain, "uint8", 3)
    b.prepare(); b.decode(); b.finish()
    for i in range(len(cases)):
        assert np.array_equal(b.output(2 * i), b.output(2 * i + 1))


def test_lf_frames(jx):
    """LF frames (frame_header.cc kDCFrame / kUseDcFrame, what `cjxl --progressive_dc` writes): the LF image of a frame travels as a
    frame of its own at 1/8 scale — VarDCT or Modular, possibly on an LF frame of the next level itself —, and the frame that refers to it has no
    LF coefficients, no LF dequantisation and no adaptive smoothing.  The HIP path decodes the LF frames in a batch of their own in front of the
    LF post-processing of the frames that use them; against the oracle, alone and in batches beside ordinary frames (both LF decode kernels)."""
    from test_synth_roundtrip import lf_frame_streams
    cases = lf_frame_streams()
    plain = S.encode_vardct(S.synthetic_image(81, 300, 200), seed=3)
    refs = {}
    for name, stream, img, _ in cases:
        _, px = check_against_oracle(jx, stream, np.uint8, 3)
        refs[name] = px.reshape(-1)
        check_against_oracle(jx, stream, np.float32, 3)
    want_plain = O.decode(plain).pixels("u8", 3)
    for lf_stride in (64, 8):
        b = jx.BatchDecoder(0)
        order = []
        for name, stream, _, _ in cases:
            b.add(stream, "uint8", 3); order.append(name)
            b.add(plain, "uint8", 3); order.append(None)
        b.set_lane_stride(lf_stride, 1)
        b.prepare()
        for _ in range(2):                    # (decoded twice: the LF frames' batch is run again with the batch)
            b.decode(); b.finish()
            for i, name in enumerate(order):
                assert np.array_equal(b.output(i), want_plain if name is None else refs[name]), (lf_stride, i, name)


def test_corrupted_round3_streams_fail_cleanly_or_decode(jx):
    """The robustness bar of test_corrupted_streams_fail_cleanly_or_decode for what round 3 added: images with a preview frame, LF frames (the nested
    batch of LF frames fails like a frame), Modular frames in several passes, prefix-coded VarDCT frames."""
    from test_synth_roundtrip import preview_streams, lf_frame_streams, multipass_modular_streams
    rng = np.random.default_rng(321)
    S.set_prefix(True)
    try:
        pfx = S.encode_vardct(S.synthetic_image(44, 320, 200), seed=5, strategy_mix=2, epf_iters=1, gab=1)
    finally:
        S.set_prefix(False)
    lf = lf_frame_streams()
    streams = [preview_streams()[1][1], lf[0][1], lf[1][1], lf[3][1], multipass_modular_streams()[0][1], multipass_modular_streams()[5][1], pfx]
    outcomes = {"error": 0, "decoded": 0}
    for data in streams:
        for trial in range(20):
            bad = bytearray(data)
            for pos in rng.integers(16, len(bad), 1 + trial % 4):
                bad[pos] ^= 1 << int(rng.integers(0, 8))
            if trial % 7 == 6:
                bad = bad[: int(rng.integers(len(bad) // 2, len(bad)))]
            try:
                meta, px = jx.decoder_builder().decode_with(bytes(bad), np.uint8)
                assert len(px) == meta.width * meta.height * (4 if meta.has_alpha_channel else 3)
                outcomes["decoded"] += 1
            except jx.DecodeError:
                outcomes["error"] += 1
    assert outcomes["error"] > 0 and outcomes["error"] + outcomes["decoded"] == 20 * len(streams)
    for data in (lf[0][1], streams[0]):                  # the decoder is still healthy afterwards
        check_against_oracle(jx, data, np.uint8, 3)


def test_corrupted_round4_streams_fail_cleanly_or_decode(jx):
    """The same bar for what round 4 added: LZ77-coded AC streams (single pass -> HfDecodeKernel, progressive -> the general SIMT instantiation; copy lengths
    and distances are attacker-controlled), cjxl-shaped weighted-predictor LF trees on the SIMT LF kernel, and a container walked through the box API."""
    from test_synth_roundtrip import lz77_ac_streams, cjxl_shape_streams
    rng = np.random.default_rng(4321)
    lz = lz77_ac_streams()
    streams = [lz[0][1], lz[2][1], lz[3][1], lz[5][1], cjxl_shape_streams()[0][2], cjxl_shape_streams()[4][2]]
    outcomes = {"error": 0, "decoded": 0}
    for data in streams:
        for trial in range(24):
            bad = bytearray(data)
            lo = 16 if trial % 3 else len(bad) // 3          # (two thirds of the trials hit anywhere, one third only the group sections)
            for pos in rng.integers(lo, len(bad), 1 + trial % 4):
                bad[pos] ^= 1 << int(rng.integers(0, 8))
            if trial % 8 == 7:
                bad = bad[: int(rng.integers(len(bad) // 2, len(bad)))]
            try:
                meta, px = jx.decoder_builder().decode_with(bytes(bad), np.uint8)
                assert len(px) == meta.width * meta.height * (4 if meta.has_alpha_channel else 3)
                outcomes["decoded"] += 1
            except jx.DecodeError:
                outcomes["error"] += 1
    assert outcomes["error"] > 0 and outcomes["error"] + outcomes["decoded"] == 24 * len(streams)
    check_against_oracle(jx, lz[0][1], np.uint8, 3)          # the decoder is still healthy afterwards
    # a batch in which one LZ77 frame is damaged: the others decode
    bad = bytearray(lz[2][1]); bad[len(bad) * 2 // 3] ^= 0x55
    b = jx.BatchDecoder(0)
    b.add(lz[0][1], "uint8", 3); b.add(bytes(bad), "uint8", 3); b.add(lz[3][1], "uint8", 3)
    b.prepare(); b.decode()
    try:
        b.finish()
    except jx.DecodeError:
        pass
    assert np.array_equal(b.output(0), O.decode(lz[0][1]).pixels("u8", 3)) and np.array_equal(b.output(2), O.decode(lz[3][1]).pixels("u8", 3))


def test_prefix_coded_progressive_and_subsampled_frames(jx):
    """Prefix-coded AC streams of progressive frames (several passes, each with its own code and orders) and of chroma-subsampled YCbCr frames (what a
    fast-effort JPEG recompression looks like): walked by the general instantiation of HfDecodeSimtKernel with the bit-serial canonical-code reader.
    Against the oracle and the ANS twin, alone and in a batch beside ANS-coded frames."""
    img = S.synthetic_image(41, 520, 300)
    pairs = []
    for make in (lambda: S.encode_vardct(img, seed=5, strategy_mix=2, num_passes=3), lambda: S.encode_vardct(img, seed=5, strategy_mix=1, num_passes=2, permute_toc=3, epf_iters=2),
                 lambda: S.encode_ycbcr(img, subsampling="420", seed=3), lambda: S.encode_ycbcr(img, subsampling="422", seed=3), lambda: S.encode_ycbcr(img, subsampling="444", seed=3)):
        ans = make()
        S.set_prefix(True)
        try:
            pfx = make()
        finally:
            S.set_prefix(False)
        assert pfx != ans
        pairs.append((pfx, ans))
    for pfx, ans in pairs:
        _, px = check_against_oracle(jx, pfx, np.uint8, 3)
        assert np.array_equal(px.reshape(-1), O.decode(ans).pixels("u8", 3))
    b = jx.BatchDecoder(0)
    for pfx, ans in pairs:
        b.add(pfx, "uint8", 3); b.add(ans, "uint8", 3)
    b.prepare(); b.decode(); b.finish()
    for i in range(len(pairs)):
        assert np.array_equal(b.output(2 * i), b.output(2 * i + 1)), i


def test_lz77_coded_lf_streams(jx):
    """LZ77 in the LF-group Modular streams of VarDCT frames (dec_ans.h): the LF kernel's general symbol reader with one 4 MB window
    per LF group; against the oracle and the ANS twin of the same frame, alone and in a batch beside plain frames (both LF decode kernels)."""
    from test_synth_roundtrip import lz77_lf_streams
    cases = lz77_lf_streams()
    for name, lz, ans in cases:
        _, px = check_against_oracle(jx, lz, np.uint8, 3)
        assert np.array_equal(px.reshape(-1), O.decode(ans).pixels("u8", 3)), name
        check_against_oracle(jx, lz, np.float32, 3)
    for lf_stride in (64, 8):
        b = jx.BatchDecoder(0)
        for name, lz, ans in cases:
            b.add(lz, "uint8", 3); b.add(ans, "uint8", 3)
        b.set_lane_stride(lf_stride, 1)
        b.prepare(); b.decode(); b.finish()
        for i in range(len(cases)):
            assert np.array_equal(b.output(2 * i), b.output(2 * i + 1)), (lf_stride, cases[i][0])


def test_lz77_coded_ac_streams(jx):
    """LZ77 in the AC coefficient streams (dec_ans.h): the general symbol reader of the HF kernels with a 1 MB window per group stream
    in global memory — single-pass frames through HfDecodeKernel, progressive ones through the general HfDecodeSimtKernel instantiation; against the
    oracle and the ANS twin, alone and in a batch beside plain frames."""
    from test_synth_roundtrip import lz77_ac_streams
    cases = lz77_ac_streams()
    for name, lz, ans, nc in cases:
        _, px = check_against_oracle(jx, lz, np.uint8, nc)
        assert np.array_equal(px.reshape(-1), O.decode(ans).pixels("u8", nc)), name
    check_against_oracle(jx, cases[0][1], np.float32, 3)
    b = jx.BatchDecoder(0)
    for name, lz, ans, nc in cases:
        b.add(lz, "uint8", nc); b.add(ans, "uint8", nc)
    b.prepare(); b.decode(); b.finish()
    for i in range(len(cases)):
        assert np.array_equal(b.output(2 * i), b.output(2 * i + 1)), cases[i][0]
    b.decode(); b.finish()     # (a second decode of the prepared batch: the windows are scratch, nothing carries over)
    for i in range(len(cases)):
        assert np.array_equal(b.output(2 * i), b.output(2 * i + 1)), cases[i][0]


def test_previous_channel_properties_in_lf_streams(jx):
    """MA trees of the LF-group streams of VarDCT frames that split on previous-channel properties (16 + 4 r + k, `cjxl -E`): the LF
    kernel's general tree walk with the earlier channels of the stream as references; against the oracle and the plain-tree twin, alone and batched."""
    from test_synth_roundtrip import prev_channel_streams
    cases = prev_channel_streams()
    for name, pc, plain in cases:
        _, px = check_against_oracle(jx, pc, np.uint8, 3)
        assert np.array_equal(px.reshape(-1), O.decode(plain).pixels("u8", 3)), name
    for lf_stride in (64, 8):
        b = jx.BatchDecoder(0)
        for name, pc, plain in cases:
            b.add(pc, "uint8", 3); b.add(plain, "uint8", 3)
        b.set_lane_stride(lf_stride, 1)
        b.prepare(); b.decode(); b.finish()
        for i in range(len(cases)):
            assert np.array_equal(b.output(2 * i), b.output(2 * i + 1)), (lf_stride, cases[i][0])


def test_multipass_modular_frames(jx):
    """Modular frames in several passes (frame_header.cc Passes, passes.h GetDownsamplingBracket): PassGroup (pass, group) carries the
    channels whose shift falls into the pass's bracket.  The HIP path decodes every (pass, group) sub-stream as a unit of its own; lossless against the
    source samples and the oracle, alone and in one batch."""
    from test_synth_roundtrip import multipass_modular_streams
    cases = multipass_modular_streams()
    for name, data, img, bits in cases:
        dt = np.uint16 if bits > 8 else np.uint8
        _, px = check_against_oracle(jx, data, dt, img.shape[2])
        assert np.array_equal(px.reshape(img.shape), img), name
    b = jx.BatchDecoder(0)
    for name, data, img, bits in cases:
        b.add(data, "uint16" if bits > 8 else "uint8", img.shape[2])
    b.prepare(); b.decode(); b.finish()
    for i, (name, data, img, bits) in enumerate(cases):
        got = b.output(i)
        got = got.view(np.uint16) if bits > 8 else got
        assert np.array_equal(got.reshape(img.shape), img), name


def test_colour_encoding_and_extra_channel_info(jx):
    """JxlDecoderGetColorAsEncodedProfile (jpegxl-sys decode.rs:833, color_encoding.rs:125-159) and JxlDecoderGetExtraChannelInfo / Name (decode.rs:756-782,
    codestream_header.rs:247-279): the enumerated colour description and the extra channels as the image header states them."""
    L = jx.libjxl()
    img = S.synthetic_image(31, 104, 72)
    rgba = np.dstack([img, (np.arange(104 * 72) % 251).astype(np.uint8).reshape(72, 104)]).astype(np.int32)

    def headers(stream):
        data = np.frombuffer(stream, np.uint8)
        dec = L.JxlDecoderCreate(None)
        assert L.JxlDecoderSizeHintBasicInfo(dec) > 0 and L.JxlDecoderGetIntendedDownsamplingRatio(dec) == 1
        assert L.JxlDecoderSubscribeEvents(dec, jx.JXL_DEC_BASIC_INFO) == 0
        assert L.JxlDecoderSetInput(dec, data.ctypes.data, len(data)) == 0
        assert L.JxlDecoderProcessInput(dec) == jx.JXL_DEC_BASIC_INFO
        ce = jx.JxlColorEncoding()
        st = L.JxlDecoderGetColorAsEncodedProfile(dec, 0, C.byref(ce))
        extras = []
        info = jx.JxlBasicInfo()
        assert L.JxlDecoderGetBasicInfo(dec, C.byref(info)) == 0
        for k in range(info.num_extra_channels):
            ei = jx.JxlExtraChannelInfo()
            assert L.JxlDecoderGetExtraChannelInfo(dec, k, C.byref(ei)) == 0
            name = C.create_string_buffer(ei.name_length + 1)
            assert L.JxlDecoderGetExtraChannelName(dec, k, name, ei.name_length + 1) == 0
            extras.append((ei.type, ei.bits_per_sample, ei.alpha_premultiplied, [round(v, 4) for v in ei.spot_color], name.value))
        ei = jx.JxlExtraChannelInfo()
        assert L.JxlDecoderGetExtraChannelInfo(dec, info.num_extra_channels, C.byref(ei)) == 1
        L.JxlDecoderDestroy(dec)
        return st, ce, extras

    st, ce, extras = headers(S.encode_vardct(img, seed=5))                                   # all defaults: sRGB
    assert st == 0 and (ce.color_space, ce.white_point, ce.primaries, ce.transfer_function, ce.rendering_intent) == (0, 1, 1, 13, 1) and extras == []
    assert abs(ce.white_point_xy[0] - 0.3127) < 1e-9 and abs(ce.primaries_red_xy[0] - 0.639998686) < 1e-9 and abs(ce.primaries_blue_xy[1] - 0.059997204) < 1e-9
    S.set_color(white_point=1, primaries=9, tf=16, intensity_target=1000.0)
    try:
        pq = S.encode_vardct(img, seed=5)
    finally:
        S.set_color()
    st, ce, _ = headers(pq)                                                                   # BT.2100 primaries, PQ
    assert st == 0 and (ce.primaries, ce.transfer_function) == (9, 16) and abs(ce.primaries_green_xy[1] - 0.797) < 1e-9
    S.set_color(white_point=11, primaries=11, tf=0, gamma=0.45455)
    try:
        g = S.encode_vardct(img, seed=5)
    finally:
        S.set_color()
    st, ce, _ = headers(g)                                                                    # DCI white, P3, gamma
    assert st == 0 and (ce.white_point, ce.primaries, ce.transfer_function) == (11, 11, 65535) and abs(ce.gamma - 0.45455) < 1e-6 and abs(ce.white_point_xy[1] - 0.351) < 1e-9
    st, _, extras = headers(S.encode_vardct(img, seed=5, alpha=rgba[..., 3].astype(np.uint8)))
    assert st == 0 and extras == [(0, 8, 0, [0.0, 0.0, 0.0, 0.0], b"")]
    S.set_spot((1.0, 0.25, 0.125, 0.75))
    try:
        spot = S.encode_modular(rgba, 8, False, 0)
    finally:
        S.set_spot()
    _, _, extras = headers(spot)
    assert extras == [(2, 8, 0, [1.0, 0.25, 0.125, 0.75], b"")]
    from PIL import ImageCms
    S.set_icc(ImageCms.ImageCmsProfile(ImageCms.createProfile("sRGB")).tobytes())
    try:
        icc = S.encode_vardct(S.synthetic_image(3, 64, 48), seed=1)
    finally:
        S.set_icc(b"")
    assert headers(icc)[0] == 1                                                               # an ICC profile: no enumerated description


def test_preview_image_is_delivered_when_subscribed(jx):
    """JXL_DEC_PREVIEW_IMAGE (jpegxl-sys decode.rs:999-1025, status 0x200): a caller that subscribes gets JXL_DEC_NEED_PREVIEW_OUT_BUFFER, sets a buffer of
    JxlDecoderPreviewOutBufferSize bytes and receives the preview — the preview frame decoded on the GPU like an image of its own — before the frames of
    the image; expected pixels: the oracle's decode of the preview written as an image of its own."""
    from test_synth_roundtrip import preview_streams
    L = jx.libjxl()
    for name, with_preview, plain, (pw, ph), alone in preview_streams():
        for dtype, jt, kind in ((np.uint8, jx.JXL_TYPE_UINT8, "u8"), (np.float32, jx.JXL_TYPE_FLOAT, "f32")):
            fmt = jx.JxlPixelFormat(3, jt, jx.JXL_NATIVE_ENDIAN, 0)
            data = np.frombuffer(with_preview, np.uint8)
            dec = L.JxlDecoderCreate(None)
            assert L.JxlDecoderSubscribeEvents(dec, jx.JXL_DEC_BASIC_INFO | jx.JXL_DEC_PREVIEW_IMAGE | jx.JXL_DEC_FULL_IMAGE) == 0
            assert L.JxlDecoderSetInput(dec, data.ctypes.data, len(data)) == 0
            L.JxlDecoderCloseInput(dec)
            events, prev, buf = [], None, None
            while True:
                st = L.JxlDecoderProcessInput(dec)
                events.append(st)
                if st == jx.JXL_DEC_NEED_PREVIEW_OUT_BUFFER:
                    size = C.c_size_t()
                    assert L.JxlDecoderPreviewOutBufferSize(dec, C.byref(fmt), C.byref(size)) == 0
                    assert size.value == pw * ph * 3 * np.dtype(dtype).itemsize
                    prev = np.zeros(pw * ph * 3, dtype)
                    assert L.JxlDecoderSetPreviewOutBuffer(dec, C.byref(fmt), prev.ctypes.data, size.value - 1) == 1     # too small
                    assert L.JxlDecoderSetPreviewOutBuffer(dec, C.byref(fmt), prev.ctypes.data, size.value) == 0
                elif st == jx.JXL_DEC_NEED_IMAGE_OUT_BUFFER:
                    size = C.c_size_t()
                    assert L.JxlDecoderImageOutBufferSize(dec, C.byref(fmt), C.byref(size)) == 0
                    buf = np.zeros(300 * 200 * 3, dtype)
                    assert L.JxlDecoderSetImageOutBuffer(dec, C.byref(fmt), buf.ctypes.data, size.value) == 0
                elif st == jx.JXL_DEC_SUCCESS:
                    break
                elif st not in (jx.JXL_DEC_BASIC_INFO, jx.JXL_DEC_PREVIEW_IMAGE, jx.JXL_DEC_FULL_IMAGE):
                    raise AssertionError(st)
            L.JxlDecoderDestroy(dec)
            assert events == [jx.JXL_DEC_BASIC_INFO, jx.JXL_DEC_NEED_PREVIEW_OUT_BUFFER, jx.JXL_DEC_PREVIEW_IMAGE, jx.JXL_DEC_NEED_IMAGE_OUT_BUFFER, jx.JXL_DEC_FULL_IMAGE, jx.JXL_DEC_SUCCESS], name
            want_prev, want_img = O.decode(alone).pixels(kind, 3).view(dtype), O.decode(plain).pixels(kind, 3).view(dtype)
            if dtype == np.float32:
                assert ulp_diff(prev, want_prev) <= 1 and ulp_diff(buf, want_img) <= 1, name
            else:
                assert np.array_equal(prev, want_prev) and np.array_equal(buf, want_img), name


def test_non_coalesced_frames_and_frame_headers(jx):
    """JxlDecoderSetCoalescing(false) (jpegxl-sys decode.rs:622, forwarded by jpegxl-rs decode.rs:356-358): every regular frame arrives as coded —
    JXL_DEC_FRAME (JxlDecoderGetFrameHeader: crop, size, blending, is_last), a buffer of the FRAME's size, its pixels un-blended — and equals
    the decode of that frame written as an image of its own; reference-only frames are not delivered; SkipFrames / SkipCurrentFrame / Rewind."""
    L = jx.libjxl()
    big, small, img = S.synthetic_image(6, 600, 400), S.synthetic_image(9, 64, 48), S.synthetic_image(5, 200, 136)
    stream = (S.encode_vardct_frame(big, S.frame(is_last=0, save_as_reference=1), seed=3, strategy_mix=2)
              + S.encode_vardct_frame(img, S.frame(emit=1, is_last=0, frame_type=2, save_as_reference=2, save_before_ct=1, have_crop=1, canvas_w=600, canvas_h=400), seed=5)    # reference only: never delivered
              + S.encode_vardct_frame(small, S.frame(emit=1, is_last=0, have_crop=1, crop_x0=300, crop_y0=160, canvas_w=600, canvas_h=400, blend_mode=1, blend_source=1), seed=4)
              + S.encode_vardct_frame(img, S.frame(emit=1, have_crop=1, crop_x0=-30, crop_y0=300, canvas_w=600, canvas_h=400, blend_mode=0, blend_source=1), seed=5, epf_iters=2))
    alone = [S.encode_vardct_frame(big, S.frame(), seed=3, strategy_mix=2), S.encode_vardct_frame(small, S.frame(), seed=4), S.encode_vardct_frame(img, S.frame(), seed=5, epf_iters=2)]
    want = [O.decode(a).pixels("u8", 3) for a in alone]
    geometry = [(0, 0, 0, 600, 400, 0, 0), (1, 300, 160, 64, 48, 1, 0), (1, -30, 300, 200, 136, 0, 1)]     # have_crop, x0, y0, w, h, blend mode, is_last
    data = np.frombuffer(stream, np.uint8)
    fmt = jx.JxlPixelFormat(3, jx.JXL_TYPE_UINT8, jx.JXL_NATIVE_ENDIAN, 0)

    def run(dec, expect_first):
        got, headers, events, buf = [], [], [], None
        while True:
            st = L.JxlDecoderProcessInput(dec)
            events.append(st)
            if st == jx.JXL_DEC_FRAME:
                fh = jx.JxlFrameHeader()
                assert L.JxlDecoderGetFrameHeader(dec, C.byref(fh)) == 0
                li = fh.layer_info
                headers.append((li.have_crop, li.crop_x0, li.crop_y0, li.xsize, li.ysize, li.blend_info.blendmode, fh.is_last))
                name = C.create_string_buffer(8)
                assert L.JxlDecoderGetFrameName(dec, name, 8) == 0 and name.value == b"" and fh.name_length == 0
            elif st == jx.JXL_DEC_NEED_IMAGE_OUT_BUFFER:
                size = C.c_size_t()
                assert L.JxlDecoderImageOutBufferSize(dec, C.byref(fmt), C.byref(size)) == 0
                buf = np.zeros(size.value, np.uint8)
                assert L.JxlDecoderSetImageOutBuffer(dec, C.byref(fmt), buf.ctypes.data, size.value) == 0
            elif st == jx.JXL_DEC_FULL_IMAGE:
                got.append(buf)
            elif st == jx.JXL_DEC_SUCCESS:
                break
            else:
                assert st == jx.JXL_DEC_BASIC_INFO, (st, jx.last_error())
        assert headers == geometry[expect_first:]
        assert len(got) == len(want) - expect_first
        for g, w_ in zip(got, want[expect_first:]):
            assert np.array_equal(g, w_)
        return events

    dec = L.JxlDecoderCreate(None)
    assert L.JxlDecoderSubscribeEvents(dec, jx.JXL_DEC_BASIC_INFO | jx.JXL_DEC_FRAME | jx.JXL_DEC_FULL_IMAGE) == 0
    assert L.JxlDecoderSetCoalescing(dec, 0) == 0
    assert L.JxlDecoderSetInput(dec, data.ctypes.data, len(data)) == 0
    ev = run(dec, 0)
    assert ev.count(jx.JXL_DEC_FRAME) == 3 and ev.count(jx.JXL_DEC_FULL_IMAGE) == 3
    L.JxlDecoderRewind(dec)                                    # settings (events, coalescing off) stay; the input is set again
    L.JxlDecoderSkipFrames(dec, 1)
    assert L.JxlDecoderSetInput(dec, data.ctypes.data, len(data)) == 0
    run(dec, 1)
    L.JxlDecoderDestroy(dec)
    # the jpegxl-rs loop (decode.rs:207-325) with coalescing(false): one buffer, refilled per frame — the last layer is what is left in it
    meta, px = jx.decoder_builder(coalescing=False, pixel_format=jx.PixelFormat(num_channels=3)).decode_with(stream, np.uint8)
    assert (meta.width, meta.height) == (600, 400) and np.array_equal(px, want[2])
    # coalesced (default): the composite, and a frame header that hides the layers
    dec = L.JxlDecoderCreate(None)
    assert L.JxlDecoderSubscribeEvents(dec, jx.JXL_DEC_FRAME | jx.JXL_DEC_FULL_IMAGE) == 0
    assert L.JxlDecoderSetInput(dec, data.ctypes.data, len(data)) == 0
    assert L.JxlDecoderProcessInput(dec) == jx.JXL_DEC_FRAME
    fh = jx.JxlFrameHeader()
    assert L.JxlDecoderGetFrameHeader(dec, C.byref(fh)) == 0
    assert (fh.layer_info.have_crop, fh.layer_info.xsize, fh.layer_info.ysize, fh.is_last) == (0, 600, 400, 1)
    assert L.JxlDecoderSkipCurrentFrame(dec) == 0 and L.JxlDecoderProcessInput(dec) == jx.JXL_DEC_SUCCESS
    L.JxlDecoderDestroy(dec)
    _, composite = jx.decoder_builder(pixel_format=jx.PixelFormat(num_channels=3)).decode_with(stream, np.uint8)
    assert np.array_equal(composite, O.decode(stream).pixels("u8", 3))


def independent_float_decode_of_sample_jpg():
    """Float decode of samples/sample.jpg that shares NO code with the oracle or the product: the Huffman-decoded JPEG coefficients
    and quant tables (tests/golden/sample_jpg_coefficients.npz, made by make_golden.py's own JPEG parser), the integer
    chroma-from-luma residual of SURVEY App. B.6, libjxl's dequantisation bias (dec_group.cc AdjustQuantBias: |q| = 1 ->
    bias_c, else q - 0.145 / q), scipy's orthonormal IDCT and the JFIF YCbCr matrix — in float64."""
    from scipy.fft import idctn
    g = np.load(os.path.join(GOLDEN, "sample_jpg_coefficients.npz"))
    co, qt = g["coefficients"].astype(np.int64), g["qtables"].astype(np.float64)
    bh, bw = 7, 5
    bias = [1 - 0.07005449891748593, 1 - 0.05465007330715401, 1 - 0.049935103337343655]       # Y, Cb (X slot), Cr (B slot)

    def adj(b, v):
        return np.where(v == 0, 0.0, np.where(np.abs(v) == 1, np.sign(v) * b, v - 0.145 / np.where(v == 0, 1, v)))

    def residual(cq, f, qc):       # what the codestream stores for a chroma coefficient (ytox = -15, ytob = 47 in this file)
        out = cq.copy()
        ff = (f * 2048) // 84 if f >= 0 else -((-f * 2048) // 84)
        for k in range(1, 64):
            scale = (2048 * int(qt[0][k]) // int(qc[k])) * ff
            out[..., k] = cq[..., k] - ((co[0][..., k] * ((scale + 1024) >> 11) + 1024) >> 11)
        return out
    dy = adj(bias[0], co[0]) * qt[0]
    dcb = adj(bias[1], residual(co[1], -15, qt[1])) * qt[1] + (-15 / 84.0) * dy
    dcr = adj(bias[2], residual(co[2], 47, qt[2])) * qt[2] + (47 / 84.0) * dy
    for d, c in ((dy, 0), (dcb, 1), (dcr, 2)):
        d[..., 0] = co[c][..., 0] * qt[c][0]                                                   # DC is not biased (it is the LF image)
    Y, Cb, Cr = [idctn(k.reshape(bh, bw, 8, 8), axes=(2, 3), norm="ortho").transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8) / 255.0 for k in (dy, dcb, dcr)]
    yb = Y + 128 / 255
    rgb = np.stack([yb + 1.402 * Cr, yb + (-0.114 * 1.772 / 0.587) * Cb + (-0.299 * 1.402 / 0.587) * Cr, yb + 1.772 * Cb], -1)
    return rgb[:50, :40]


def test_sample_jpg_jxl_pixels(jx):
    """tests/decode.rs:123-139 input: a JPEG-transcoded VarDCT frame written by the real encoder (YCbCr, RAW quant tables, custom
    block contexts and coefficient order).  The HIP path's float output must equal an independent float64 decode of sample.jpg
    (see above) to 0.01 of an 8-bit step — this pins RAW-table layout and scaling, the dequantisation bias, float chroma-from-luma,
    the 8x8 IDCT, the YCbCr stage and the write stage ON THE GPU without the oracle.  (Against a *plain* JPEG decode the bound
    would be 22 steps: libjxl's bias moves every |q| = 1 coefficient by 5-7 % of its quantisation step.)"""
    from PIL import Image
    data = fixture_bytes("sample_jpg.jxl")
    meta, px = check_against_oracle(jx, data, np.uint8, 3)
    assert (meta.width, meta.height) == (40, 50)
    want = independent_float_decode_of_sample_jpg()
    _, pf = jx.decoder_builder().decode_with(data, np.float32)
    assert np.abs(pf.reshape(50, 40, 3).astype(np.float64) - want).max() * 255 < 0.01
    q = np.clip(want, 0, 1) * 255
    sure = np.abs(q - np.round(q)) < 0.49                                   # away from rounding ties
    assert np.array_equal(px.reshape(50, 40, 3)[sure], np.round(q)[sure].astype(np.uint8))
    jpg = np.array(Image.open(os.path.join(FIXTURES, "sample.jpg")).convert("RGB")).astype(np.int32)
    diff = np.abs(px.reshape(50, 40, 3).astype(np.int32) - jpg)
    assert diff.mean() < 2.5 and diff.max() <= 24                          # libjpeg (integer IDCT, no bias): sanity only


def test_bench_jxl_modular_groups(jx):
    """benches/decode.rs:10 input: 54 groups, global MA tree with weighted predictor, per-group palettes and RCTs."""
    meta, px = jx.decoder_builder().decode_with(fixture_bytes("bench.jxl"), np.uint8)
    assert (meta.width, meta.height) == (2122, 1433)
    assert hashlib.sha256(px.tobytes()).hexdigest() == MANIFEST["reference_fixtures"]["bench.jxl"]["sha256_rgba8"]


# ---- VarDCT (the path north_star names) ---------------------------------------------------------------------------------
@pytest.mark.parametrize("name", [k for k in MANIFEST if k.startswith("vardct_")])
def test_vardct_golden_streams(jx, name):
    data = open(os.path.join(GOLDEN, name + ".jxl"), "rb").read()
    _, px = jx.decoder_builder().decode_with(data, np.uint8)
    assert hashlib.sha256(px.tobytes()).hexdigest() == MANIFEST[name]["sha256_u8_rgb"]
    _, pf = jx.decoder_builder().decode_with(data, np.float32)
    ref = O.decode(data).pixels("f32", 3).view(np.float32)
    assert ulp_diff(pf, ref) <= 1


@pytest.mark.parametrize("name", [k for k in MANIFEST if k.startswith("vardct2_") or k.startswith("modular2_")])
def test_feature_golden_streams(jx, name):
    """The committed feature streams through the C ABI (orientation kept as stored: the hashes are of the stored raster)."""
    m = MANIFEST[name]
    data = open(os.path.join(GOLDEN, name + ".jxl"), "rb").read()
    if name.startswith("modular2_"):
        _, px = jx.decoder_builder().decode_with(data, np.uint16)
        assert hashlib.sha256(px.astype("<u2").tobytes()).hexdigest() == m["sha256_u16_ga_le"]
        return
    dec = jx.decoder_builder(skip_reorientation=True, pixel_format=jx.PixelFormat(num_channels=m["channels"]))
    _, px = dec.decode_with(data, np.uint8)
    assert hashlib.sha256(px.tobytes()).hexdigest() == m["sha256_u8"]
    _, pf = dec.decode_with(data, np.float32)
    assert ulp_diff(pf, O.decode(data).pixels("f32", m["channels"]).view(np.float32)) <= 1


@pytest.mark.parametrize("s", list(range(27)))
def test_vardct_every_strategy(jx, s):
    img = S.synthetic_image(7, 256, 128) if s < 21 else S.synthetic_image(7, 520, 300)   # DCT128/256 need room; 520x300 leaves ragged edges
    data = S.encode_vardct(img, seed=5, strategy_mix=100 + s, epf_iters=1, gab=1)
    check_against_oracle(jx, data, np.float32, 3)
    check_against_oracle(jx, data, np.uint8, 3)


@pytest.mark.parametrize("w,h,mix,epf,gab,skip", [(8, 8, 0, 0, 0, 0), (9, 17, 0, 1, 1, 0), (64, 64, 0, 2, 1, 1), (100, 37, 1, 1, 1, 0), (256, 256, 1, 3, 1, 0),
                                                  (257, 255, 2, 1, 0, 0), (300, 200, 2, 2, 1, 0), (520, 300, 2, 3, 1, 0), (2100, 300, 2, 1, 1, 0), (130, 2060, 1, 0, 1, 0)])
def test_vardct_shapes_and_filters(jx, w, h, mix, epf, gab, skip):
    """Ragged sizes (non-multiples of 8 / 256 / 2048), single-section and multi-LF-group frames, every filter combination."""
    img = np.ascontiguousarray(S.synthetic_image(21, max(w, 8), max(h, 8))[:h, :w])
    data = S.encode_vardct(img, seed=6, strategy_mix=mix, epf_iters=epf, gab=gab, skip_lf_smoothing=skip)
    check_against_oracle(jx, data, np.uint8, 3)
    check_against_oracle(jx, data, np.uint16, 4, align=32)
    check_against_oracle(jx, data, np.float32, 3)
    check_against_oracle(jx, data, np.float16, 3)


@pytest.mark.parametrize("w,h", [(64, 40), (68, 33), (66, 40), (70, 24), (260, 50), (1028, 70)])
def test_fused_filter_dword_stores(jx, w, h):
    """The fused filter kernel writes u8 RGB / RGBA as whole dwords (RGB: a DPP exchange inside every quad of lanes) when rows and buffer are 4-byte
    aligned, the width is a multiple of 4 and the image is written in its own orientation; every other case keeps the byte stores.  Widths on both
    sides of that condition, with and without an alpha plane, aligned and unaligned rows, a rotated image, sample types that never take the path."""
    img = np.ascontiguousarray(S.synthetic_image(33, max(w, 8), max(h, 8))[:h, :w])
    alpha = ((np.arange(h)[:, None] * 7 + np.arange(w)[None, :] * 3) % 256).astype(np.uint8)
    data = S.encode_vardct(img, seed=9, strategy_mix=2, epf_iters=1, gab=1)
    data_a = S.encode_vardct(img, seed=9, strategy_mix=2, epf_iters=1, gab=1, alpha=alpha)
    for d in (data, data_a):
        for nch in (3, 4):
            check_against_oracle(jx, d, np.uint8, nch)
            check_against_oracle(jx, d, np.uint8, nch, align=64)
        check_against_oracle(jx, d, np.uint16, 3)
    rotated = S.encode_vardct(img, seed=9, strategy_mix=2, epf_iters=1, gab=1, orientation=6)
    for nch in (3, 4):      # (the oracle renders the stored raster; orientation 6 = rotated by 90 degrees clockwise on the way out: never the dword path)
        want = _orient(O.decode(rotated).pixels("u8", nch).reshape(h, w, nch), 6)
        _, px = jx.decoder_builder(pixel_format=jx.PixelFormat(num_channels=nch)).decode_with(rotated, np.uint8)
        assert np.array_equal(px.reshape(want.shape), want)
    # a batch writes frame after frame into one buffer: frames whose byte size is not a multiple of 4 leave the next one's rows unaligned
    odd = np.ascontiguousarray(S.synthetic_image(34, 72, 24)[:21, :66])
    streams = [S.encode_vardct(odd, seed=3, epf_iters=1, gab=1), data, S.encode_vardct(odd, seed=4, epf_iters=1, gab=1)]
    b = jx.BatchDecoder(0)
    for st in streams:
        b.add(st, "uint8", 3)
    b.prepare(); b.decode(); b.finish()
    for i, st in enumerate(streams):
        assert np.array_equal(b.output(i).reshape(-1), O.decode(st).pixels("u8", 3).reshape(-1)), i


@pytest.mark.parametrize("kind", ["flat", "one_edge", "dots"])
def test_degenerate_ac_histograms(jx, kind):
    """Pictures whose AC token statistics collapse — a constant colour (every block: zero non-zeros), one vertical edge, a few isolated dots — give AC
    codes with one-symbol histograms (alias slots that all point at the symbol, frequency 4096) and clusters no token ever uses: the HF kernel's compact
    alias tables (a u16 frequency per symbol, written in two passes) must decode them like the 8-byte slots did.  Alone and in one batch."""
    w, h = 264, 200
    img = np.full((h, w, 3), (90, 140, 200), dtype=np.uint8)
    if kind == "one_edge":
        img[:, w // 2:] = (200, 60, 30)
    elif kind == "dots":
        for (y, x) in [(13, 17), (100, 130), (190, 250), (64, 64)]:
            img[y, x] = (255, 255, 255)
    streams = []
    for mix, epf in ((0, 0), (2, 1)):
        data = S.encode_vardct(img, seed=5, strategy_mix=mix, epf_iters=epf, gab=epf)
        check_against_oracle(jx, data, np.uint8, 3)
        check_against_oracle(jx, data, np.float32, 3)
        streams.append(data)
    streams.append(S.encode_vardct(S.synthetic_image(12, w, h), seed=2, strategy_mix=2, epf_iters=1, gab=1))    # an ordinary frame in the same launch
    b = jx.BatchDecoder(0)
    for st in streams:
        b.add(st, "uint8", 3)
    b.prepare(); b.decode(); b.finish()
    for i, st in enumerate(streams):
        assert np.array_equal(b.output(i).reshape(-1), O.decode(st).pixels("u8", 3).reshape(-1)), i


def test_unaligned_varblocks_and_generic_idct(jx):
    """Varblocks that are not contained in a 64x64 tile (legal, never produced by encoders) take the generic IDCT kernel;
    forcing the generic kernel on a regular stream must give the same pixels as the tiled kernel."""
    img = S.synthetic_image(17, 520, 300)
    data = S.encode_vardct(img, seed=8, strategy_mix=3, epf_iters=1, gab=1)
    check_against_oracle(jx, data, np.uint8, 3)
    check_against_oracle(jx, data, np.float32, 3)
    regular = S.encode_vardct(img, seed=8, strategy_mix=2, epf_iters=1, gab=1)
    ref = O.decode(regular).pixels("f32", 3).view(np.float32)
    for force in (0, 1):
        b = jx.BatchDecoder(0)
        b.add(regular, "float32", 3)
        b.set_option("force_generic_idct", force)
        b.prepare(); b.decode(); b.finish()
        assert ulp_diff(b.output(0), ref) <= 1


@pytest.mark.parametrize("mix", [4, 5])
def test_dct128_256_family_in_mixed_streams(jx, mix):
    """north_star's "variable-block IDCT 2x2...256x256": DCT128x128 ... DCT256x256 varblocks mixed with every smaller strategy,
    tile-aligned (mix 4: wave-per-row LDS transforms next to the tiled kernel) and at arbitrary positions (mix 5: generic path),
    across several groups and with ragged right/bottom edges; forcing the generic small-block kernel must not change a bit."""
    img = S.synthetic_image(11, 1100, 800)
    data = S.encode_vardct(img, seed=5, strategy_mix=mix, epf_iters=2, gab=1)
    check_against_oracle(jx, data, np.uint8, 3)
    ref = O.decode(data).pixels("f32", 3).view(np.float32)
    for force in (0, 1):
        b = jx.BatchDecoder(0)
        b.add(data, "float32", 3)
        b.set_option("force_generic_idct", force)
        b.prepare(); b.decode(); b.finish()
        assert ulp_diff(b.output(0), ref) <= 1


def test_fused_and_unfused_filter_paths_agree(jx):
    """gaborish + EPF 1 frames take the fused LDS-tiled kernel; the stage-by-stage kernels must give the same bits."""
    img = S.synthetic_image(19, 330, 150)
    data = S.encode_vardct(img, seed=12, strategy_mix=2, epf_iters=1, gab=1)
    ref = O.decode(data)
    for force in (0, 1):
        for dt, kind in (("uint8", "u8"), ("float32", "f32")):
            b = jx.BatchDecoder(0)
            b.add(data, dt, 3)
            b.set_option("force_unfused_filters", force)
            b.prepare(); b.decode(); b.finish()
            want = ref.pixels(kind, 3).view(np.dtype(dt))
            got = b.output(0)
            if dt == "uint8":
                assert np.array_equal(got, want), force
            else:
                assert ulp_diff(got, want) <= 1, force


def test_hdr_float_stream(jx):
    """Config-5 style: f32 samples, linear transfer, intensity_target 1000, EPF 3."""
    lin = ((S.synthetic_image(9, 320, 200).astype(np.float32) / 255.0) ** 2.2) * 2.0
    data = S.encode_vardct(lin, seed=4, strategy_mix=2, epf_iters=3, gab=1, out_bits=32, hdr=1)
    meta, px = check_against_oracle(jx, data, np.float32, 3)
    assert abs(meta.intensity_target - 1000.0) < 1e-3
    m, p2 = jx.decoder_builder().decode(data)
    assert p2.dtype == np.float32


@pytest.mark.parametrize("h,w,c,bits,rct", [(50, 40, 3, 8, False), (300, 520, 4, 8, True), (64, 64, 1, 16, False), (257, 255, 3, 16, True), (700, 300, 2, 8, False)])
def test_modular_synth(jx, h, w, c, bits, rct):
    base = S.synthetic_image(3, w, h).astype(np.int32)
    img = np.stack([base[..., i % 3] * ((1 << bits) - 1) // 255 for i in range(c)], -1)
    _, px = jx.decoder_builder().decode_with(S.encode_modular(img, bits, rct), np.uint16 if bits == 16 else np.uint8)
    assert np.array_equal(px.reshape(h, w, c), img)


def _smooth_image(seed, h, w, c, bits):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = ((np.sin(xx / 37.0) + np.cos(yy / 23.0)) * 0.25 + 0.5) * ((1 << bits) - 1)
    noise = rng.normal(0, (1 << bits) / 1024.0, (h, w, c)).astype(np.float32)
    return np.clip(base[..., None] + noise, 0, (1 << bits) - 1).astype(np.int32)


@pytest.mark.parametrize("h,w,c,bits,rct,squeeze", [(50, 40, 1, 8, 0, 1), (280, 300, 3, 8, 1, 1), (280, 300, 4, 16, 1, 2), (520, 700, 1, 16, 0, 1),
                                                   (2070, 2100, 1, 16, 0, 1), (600, 2300, 3, 8, 0, 1), (777, 333, 2, 8, 0, 2), (9, 1, 1, 8, 0, 1)])
def test_modular_squeeze(jx, h, w, c, bits, rct, squeeze):
    """Lossless Modular with the Squeeze transform (default chain as `cjxl -d 0 -R 1` signals it, and explicit chains with
    appended residuals): residual channels ride in GlobalModular / LfGroup (shift >= 3) / PassGroup sections; the decode
    must return the source samples exactly and agree with the oracle."""
    img = _smooth_image(5, h, w, c, bits)
    data = S.encode_modular(img, bits, bool(rct), squeeze)
    dt = np.uint16 if bits == 16 else np.uint8
    _, px = jx.decoder_builder().decode_with(data, dt)
    assert np.array_equal(px.reshape(h, w, c), img)
    assert np.array_equal(px, O.decode(data).pixels("u16" if bits == 16 else "u8", c).view(dt))


def test_full_size_8k_modular_squeeze(jx):
    """BASELINE config 4: lossless Modular (Squeeze) 8192x8192 u16 on one GPU, bit-exact (lossless round trip of the source)."""
    img = _smooth_image(6, 8192, 8192, 1, 16)
    data = S.encode_modular(img, 16, False, 1)
    meta, px = jx.decoder_builder().decode_with(data, np.uint16)
    assert (meta.width, meta.height, meta.num_color_channels) == (8192, 8192, 1)
    assert np.array_equal(px.reshape(8192, 8192), img[..., 0])


@pytest.mark.parametrize("w,h", [(200, 136), (256, 256), (600, 400), (1030, 270)])
def test_vardct_with_alpha_extra_channel(jx, w, h):
    """VarDCT colour + a lossless 8-bit alpha extra channel in the frame's Modular sub-streams: GlobalModular for one-group
    images (the LfGroup then starts where that stream ends), else the Modular tail of every PassGroup (starting where the
    group's HF coefficients end).  RGBA / RGB / grey+alpha outputs, u8 and f32."""
    img = S.synthetic_image(50, w, h)
    yy, xx = np.mgrid[0:h, 0:w]
    al = ((np.sin(xx / 17.0) * np.cos(yy / 11.0) * 0.5 + 0.5) * 255).astype(np.uint8)
    data = S.encode_vardct(img, seed=9, strategy_mix=2, epf_iters=1, gab=1, alpha=al)
    meta, px = jx.decoder_builder().decode_with(data, np.uint8)
    assert meta.has_alpha_channel and len(px) == w * h * 4
    assert np.array_equal(px.reshape(h, w, 4)[..., 3], al)
    check_against_oracle(jx, data, np.uint8, 4)
    check_against_oracle(jx, data, np.uint8, 3)
    check_against_oracle(jx, data, np.float32, 4)
    check_against_oracle(jx, data, np.uint16, 4)
    data = S.encode_vardct(img, seed=9, strategy_mix=1, epf_iters=3, gab=0, alpha=al)   # unfused filter path
    check_against_oracle(jx, data, np.uint8, 4)


@pytest.mark.parametrize("gab,epf", [(1, 0), (0, 1), (1, 1), (1, 2), (1, 3), (0, 3)])
def test_custom_restoration_filter_parameters(jx, gab, epf):
    """RestorationFilter with every custom field (loop_filter.cc): gaborish weights per channel, the EPF sharpness LUT, channel scales, quant_mul, pass 0 / 2 sigma scales,
    border SAD multiplier.  The fused and the staged filter kernels take them from the frame, not from constants: HIP == oracle, and the picture differs from the
    default-parameter twin."""
    img = S.synthetic_image(12, 300, 280)
    plain = S.encode_vardct(img, seed=3, strategy_mix=2, epf_iters=epf, gab=gab)
    S.set_custom_filters(True)
    try:
        data = S.encode_vardct(img, seed=3, strategy_mix=2, epf_iters=epf, gab=gab)
        big = S.encode_vardct(S.synthetic_image(13, 1030, 270), seed=5, strategy_mix=1, epf_iters=epf, gab=gab)
    finally:
        S.set_custom_filters(False)
    _, a = check_against_oracle(jx, data, np.uint8, 3)
    check_against_oracle(jx, data, np.float32, 3)
    check_against_oracle(jx, big, np.uint8, 3)
    _, b = jx.decoder_builder().decode_with(plain, np.uint8)
    assert not np.array_equal(a, b)
    bd = jx.BatchDecoder(0)            # the unfused path and a batch mixing both kinds of frames
    bd.add(data, "uint8", 3); bd.add(plain, "uint8", 3); bd.add(big, "uint8", 3)
    bd.set_option("force_unfused_filters", 1)
    bd.prepare(); bd.decode(); bd.finish()
    assert np.array_equal(bd.output(0), a.reshape(-1)) and np.array_equal(bd.output(1), b.reshape(-1))
    bd2 = jx.BatchDecoder(0)
    bd2.add(plain, "uint8", 3); bd2.add(data, "uint8", 3)
    bd2.prepare(); bd2.decode(); bd2.finish()
    assert np.array_equal(bd2.output(1), a.reshape(-1)) and np.array_equal(bd2.output(0), b.reshape(-1))


def test_modular_group_sizes(jx):
    """Modular frames with groups of 128, 512 and 1024 samples a side (frame header group_size_shift 0, 2, 3): the section grid, the LfGroup size (8 groups) and
    which channels fit GlobalModular all follow the group size.  Lossless: the source samples come back; also against the oracle."""
    from test_synth_roundtrip import modular_group_size_streams
    for name, data, img, bits in modular_group_size_streams():
        dtype = np.uint8 if bits == 8 else np.uint16
        meta, px = jx.decoder_builder().decode_with(data, dtype)
        assert np.array_equal(px.reshape(img.shape), img), name
        check_against_oracle(jx, data, np.float32, img.shape[2])


def test_block_context_maps(jx):
    """Frames with a BlockCtxMap of their own (LF and quantiser-field thresholds, 16 block contexts — what libjxl's encoder writes at default effort): both HF kernels,
    alone, batched beside the default-map twins (per-frame context tables in one launch), damaged."""
    from test_synth_roundtrip import block_ctx_map_streams
    cases = block_ctx_map_streams()
    for name, many, one in cases:
        check_against_oracle(jx, many, np.uint8, 3)
        check_against_oracle(jx, many, np.float32, 3)
    b = jx.BatchDecoder(0)
    for name, many, one in cases:
        b.add(many, "uint8", 3); b.add(one, "uint8", 3)
    b.prepare()
    for _ in range(2):
        b.decode(); b.finish()
        for i, (name, many, one) in enumerate(cases):
            assert np.array_equal(b.output(2 * i), b.output(2 * i + 1)), name
    rng = np.random.default_rng(6)
    for name, many, one in cases[:3]:
        for k in range(30):
            bad = bytearray(many)
            for _ in range(1 + k % 3):
                bad[int(rng.integers(20, len(bad)))] ^= 1 << int(rng.integers(0, 8))
            try:
                jx.decoder_builder().decode_with(bytes(bad), np.uint8)
            except jx.DecodeError:
                pass
        check_against_oracle(jx, many, np.uint8, 3)


def test_custom_lf_dequantisation_and_colour_correlation(jx):
    """LfGlobal with its own LF dequantisation steps and chroma-from-luma parameters (colour factor, base correlations, LF factors): LF dequantisation, the LLF and the
    dequantisation in the IDCT kernels take them from the frame.  Alone and in a batch beside default-parameter frames."""
    from test_synth_roundtrip import custom_lf_global_streams
    cases = custom_lf_global_streams()
    for name, data, img in cases:
        check_against_oracle(jx, data, np.uint8, 3)
        check_against_oracle(jx, data, np.float32, 3)
    b = jx.BatchDecoder(0)
    plain = S.encode_vardct(cases[1][2], seed=4, strategy_mix=2, epf_iters=1, gab=1)
    for name, data, img in cases:
        b.add(data, "uint8", 3); b.add(plain, "uint8", 3)
    b.prepare(); b.decode(); b.finish()
    for i, (name, data, img) in enumerate(cases):
        assert np.array_equal(b.output(2 * i), O.decode(data).pixels("u8", 3)), name
        assert np.array_equal(b.output(2 * i + 1), O.decode(plain).pixels("u8", 3)), name


def test_custom_opsin_inverse_matrix_and_quant_biases(jx):
    """Image header with an OpsinInverseMatrix bundle of its own (inverse matrix, opsin biases, quantisation biases; binary16 values close to the defaults): the dequantisation
    bias in the IDCT kernels and the XYB stage of the fused and staged tails take them from the image.  u8 / f32, with custom upsampling weights in the same bundle, batched."""
    img = S.synthetic_image(33, 300, 280)
    plain = S.encode_vardct(img, seed=4, strategy_mix=2, epf_iters=1, gab=1)
    S.set_custom_opsin(True)
    try:
        data = S.encode_vardct(img, seed=4, strategy_mix=2, epf_iters=1, gab=1)
        ups = S.encode_vardct(S.synthetic_image(34, 520, 300), seed=4, strategy_mix=2, epf_iters=2, gab=1, upsampling=2, custom_up_weights=1)
        staged = S.encode_vardct(img, seed=4, strategy_mix=1, epf_iters=3, gab=0)
    finally:
        S.set_custom_opsin(False)
    _, a = check_against_oracle(jx, data, np.uint8, 3)
    check_against_oracle(jx, data, np.float32, 3)
    check_against_oracle(jx, ups, np.uint8, 3)
    check_against_oracle(jx, staged, np.uint8, 3)
    _, b = jx.decoder_builder().decode_with(plain, np.uint8)
    assert not np.array_equal(a, b) and np.abs(a.astype(int) - b.astype(int)).max() <= 3       # close to the default parameters, not equal to them
    bd = jx.BatchDecoder(0)
    for d in (plain, data, ups, staged, plain):
        bd.add(d, "uint8", 3)
    bd.prepare(); bd.decode(); bd.finish()
    assert np.array_equal(bd.output(0), b.reshape(-1)) and np.array_equal(bd.output(1), a.reshape(-1)) and np.array_equal(bd.output(4), b.reshape(-1))
    assert np.array_equal(bd.output(2), O.decode(ups).pixels("u8", 3)) and np.array_equal(bd.output(3), O.decode(staged).pixels("u8", 3))


@pytest.mark.parametrize("xq,bq", [(0, 0), (2, 4), (5, 2), (7, 7)])
def test_qm_scales(jx, xq, bq):
    """x_qm_scale / b_qm_scale other than the default 3 / 2 (libjxl's encoder raises x_qm_scale with the distance): X / B dequantisation steps times 0.8^(scale - 2)"""
    img = S.synthetic_image(33, 300, 280)
    S.set_qm_scales(xq, bq)
    try:
        data = S.encode_vardct(img, seed=4, strategy_mix=2, epf_iters=1, gab=1)
    finally:
        S.set_qm_scales()
    _, px = check_against_oracle(jx, data, np.uint8, 3)
    check_against_oracle(jx, data, np.float32, 3)
    err = px.reshape(280, 300, 3).astype(np.float64) - img
    assert 10 * np.log10(255.0 ** 2 / (err ** 2).mean()) > 37.5


@pytest.mark.parametrize("q", [1, 7, 40, 300, 5000])
def test_quant_lf_values(jx, q):
    """quant_lf over its coded range (1, 1 + 5 bits, 1 + 8 bits, 1 + 16 bits): LF steps from a few per cent of the range down to 1e-6 — large quantised LF values, the
    block-context LF thresholds with them, LF smoothing on fine steps"""
    img = S.synthetic_image(33, 300, 280)
    S.set_quant_lf(q); S.set_custom_block_ctx(True)
    try:
        data = S.encode_vardct(img, seed=4, strategy_mix=2, epf_iters=1, gab=1)
    finally:
        S.set_quant_lf(); S.set_custom_block_ctx(False)
    check_against_oracle(jx, data, np.uint8, 3)
    check_against_oracle(jx, data, np.float32, 3)


@pytest.mark.parametrize("e", [1, 2, 3])
def test_lf_extra_precision(jx, e):
    """LF groups with extra_precision 1..3 (LF coefficients in steps 2 / 4 / 8 times finer; libjxl's encoder uses it at low distances): the cooperative LF kernel and the
    SIMT one (a batch of frames, gradient tree and weighted-predictor tree), several LF groups"""
    img, wide = S.synthetic_image(33, 300, 280), S.synthetic_image(35, 2300, 400)
    S.set_lf_extra_precision(e)
    try:
        small = S.encode_vardct(img, seed=4, strategy_mix=2, epf_iters=1, gab=1)
        big = S.encode_vardct(wide, seed=5, strategy_mix=1, epf_iters=2, gab=1)
        S.set_lf_tree_shape(1)
        wp = S.encode_vardct(img, seed=6, strategy_mix=2, epf_iters=1, gab=1)
    finally:
        S.set_lf_extra_precision(); S.set_lf_tree_shape(0)
    for d in (small, big, wp):
        check_against_oracle(jx, d, np.uint8, 3)
    check_against_oracle(jx, small, np.float32, 3)
    for group in ([small, big] * 4, [wp] * 8):
        b = jx.BatchDecoder(0)
        b.set_lane_stride(4, 1)
        for d in group:
            b.add(d, "uint8", 3)
        b.prepare(); b.decode(); b.finish()
        assert b.info_value("lf_simt_frames") == len(group)
        for i, d in enumerate(group):
            assert np.array_equal(b.output(i), O.decode(d).pixels("u8", 3)), i


def test_several_hf_histogram_sets(jx):
    """HfGlobal num_hf_presets > 1 (what libjxl's encoder writes for larger pictures): every PassGroup picks one of several sets of AC histograms.  Alone (both HF kernels: the
    SIMT one for ANS streams, HfDecodeKernel for prefix codes), in one batch beside their one-set twins, and with the selector damaged."""
    from test_synth_roundtrip import hf_preset_streams
    cases = hf_preset_streams()
    for name, many, one in cases:
        check_against_oracle(jx, many, np.uint8, 3)
        check_against_oracle(jx, many, np.float32, 3)
    b = jx.BatchDecoder(0)
    for name, many, one in cases:
        b.add(many, "uint8", 3); b.add(one, "uint8", 3)
    b.prepare()
    for _ in range(2):
        b.decode(); b.finish()
        for i, (name, many, one) in enumerate(cases):
            assert np.array_equal(b.output(2 * i), b.output(2 * i + 1)), name
    rng = np.random.default_rng(5)
    for name, many, one in cases[:3]:
        for k in range(30):
            bad = bytearray(many)
            for _ in range(1 + k % 3):
                bad[int(rng.integers(len(bad) // 6, len(bad)))] ^= 1 << int(rng.integers(0, 8))
            try:
                jx.decoder_builder().decode_with(bytes(bad), np.uint8)
            except jx.DecodeError:
                pass
        check_against_oracle(jx, many, np.uint8, 3)


def test_vardct_with_squeezed_alpha(jx):
    """The extra channel of a VarDCT frame under the default Squeeze chain (what cjxl does to a progressive or lossy alpha of an RGBA picture): residual channels in GlobalModular,
    in the LfGroup sections (shift >= 3: decoded by the LF kernel between the LF coefficients and the HF metadata) and in the PassGroup tails; inverse
    Squeeze before the write stage.  Alone, batched beside plain frames, and with the stream cut / bit-flipped."""
    from test_synth_roundtrip import squeezed_alpha_streams
    cases = squeezed_alpha_streams()
    for name, sq, plain, al in cases:
        meta, px = jx.decoder_builder().decode_with(sq, np.uint8)
        assert meta.has_alpha_channel, name
        h, w = al.shape
        assert np.array_equal(px.reshape(h, w, 4)[..., 3], al), name
        check_against_oracle(jx, sq, np.uint8, 4)
        check_against_oracle(jx, sq, np.float32, 4)
        check_against_oracle(jx, sq, np.uint8, 3)
    # one batch: squeezed frames beside their plain twins (the SIMT LF path for the twins, the cooperative kernel for the LfGroup sub-channels)
    streams = [c[1] for c in cases] + [c[2] for c in cases]
    b = jx.BatchDecoder(0)
    for s in streams:
        b.add(s, "uint8", 4)
    b.prepare()
    for _ in range(2):
        b.decode()
        b.finish()
        for i, (name, sq, plain, al) in enumerate(cases):
            assert np.array_equal(b.output(i), b.output(i + len(cases))), name
            assert np.array_equal(np.asarray(b.output(i)).reshape(al.shape + (4,))[..., 3], al), name
    rng = np.random.default_rng(12)
    for name, sq, plain, al in cases[:3]:
        for k in range(24):
            bad = bytearray(sq)
            if k % 3 == 0:
                bad = bad[: int(rng.integers(40, len(bad)))]
            else:
                for _ in range(int(rng.integers(1, 4))):
                    bad[int(rng.integers(30, len(bad)))] ^= 1 << int(rng.integers(0, 8))
            try:
                jx.decoder_builder().decode_with(bytes(bad), np.uint8)
            except Exception:
                pass
        meta, px = jx.decoder_builder().decode_with(sq, np.uint8)      # the decoder is still sound
        assert np.array_equal(px.reshape(al.shape + (4,))[..., 3], al), name


@pytest.mark.parametrize("up,custom,with_alpha", [(2, 0, False), (2, 1, False), (4, 1, False), (8, 1, False), (2, 0, True), (4, 1, True)])
def test_upsampled_frames(jx, up, custom, with_alpha):
    """north_star's "upsampling": frames coded at 1/2, 1/4, 1/8 of the image size and brought back by the non-separable 5x5
    kernels (explicit weights from the image header; library default for 2x), alpha upsampled alongside, ragged sizes."""
    w, h = 333, 201
    img = S.synthetic_image(60 + up, w, h)
    yy, xx = np.mgrid[0:h, 0:w]
    al = ((np.sin(xx / 17.0) * np.cos(yy / 11.0) * 0.5 + 0.5) * 255).astype(np.uint8) if with_alpha else None
    data = S.encode_vardct(img, seed=4, strategy_mix=2, epf_iters=2, gab=1, upsampling=up, custom_up_weights=custom, alpha=al)
    nch = 4 if with_alpha else 3
    meta, px = check_against_oracle(jx, data, np.uint8, nch)
    assert (meta.width, meta.height) == (w, h)
    check_against_oracle(jx, data, np.float32, nch)
    err = px.reshape(h, w, nch)[..., :3].astype(np.float64) - img
    assert 10 * np.log10(255.0 ** 2 / (err ** 2).mean()) > (33.0, 28.0, 23.0)[(2, 4, 8).index(up)]   # still the source picture
    _, po = jx.decoder_builder().decode_with(S.encode_vardct(img, seed=4, upsampling=up, custom_up_weights=1, orientation=6), np.uint8)
    assert len(po) == w * h * 3


@pytest.mark.parametrize("up", [4, 8])
def test_default_4x_8x_upsampling_weights(jx, up):
    """Streams relying on the library-default 4x / 8x kernels (image_metadata.cc kWeights4 / kWeights8)."""
    data = S.encode_vardct(S.synthetic_image(3, 200, 136), seed=2, strategy_mix=1, upsampling=up, custom_up_weights=0)
    check_against_oracle(jx, data, np.uint8, 3)
    check_against_oracle(jx, data, np.float32, 3)


@pytest.mark.parametrize("npass,with_alpha,w,h", [(2, False, 200, 136), (3, False, 600, 400), (2, True, 600, 400), (3, True, 1030, 270)])
def test_progressive_passes(jx, npass, with_alpha, w, h):
    """Multi-pass (progressive) VarDCT frames: every PassGroup section (pass, group) adds value << shift under the pass's own
    entropy code; alpha rides in the last pass.  The pixels must equal the single-pass encoding of the same coefficients."""
    img = S.synthetic_image(80, w, h)
    al = (np.add.outer(np.arange(h), np.arange(w)) % 256).astype(np.uint8) if with_alpha else None
    data = S.encode_vardct(img, seed=6, strategy_mix=2, epf_iters=1, gab=1, num_passes=npass, alpha=al)
    nch = 4 if with_alpha else 3
    _, px = check_against_oracle(jx, data, np.uint8, nch)
    check_against_oracle(jx, data, np.float32, nch)
    _, one = jx.decoder_builder().decode_with(S.encode_vardct(img, seed=6, strategy_mix=2, epf_iters=1, gab=1, alpha=al), np.uint8)
    assert np.array_equal(px, one)
    b = jx.BatchDecoder(0)            # a lane-stride setting that would pick the per-wavefront HF kernel is overridden
    b.add(data, "uint8", nch); b.set_lane_stride(64, 64); b.prepare(); b.decode(); b.finish()
    assert np.array_equal(b.output(0), px)


def test_permuted_toc(jx):
    """Sections stored in a shuffled order with the Lehmer-coded permutation in the TOC (what cjxl emits for centre-first /
    progressive orderings): same pixels as the in-order stream."""
    img = S.synthetic_image(81, 600, 400)
    plain = S.encode_vardct(img, seed=7, strategy_mix=2)
    _, want = jx.decoder_builder().decode_with(plain, np.uint8)
    for kw in ({"permute_toc": 7}, {"permute_toc": 9, "num_passes": 2}):
        data = S.encode_vardct(img, seed=7, strategy_mix=2, **kw)
        assert data != plain
        _, px = check_against_oracle(jx, data, np.uint8, 3)
        assert np.array_equal(px, want)


def test_mixed_batch_in_two_halves_on_two_streams(jx):
    """The pipelined use of the batch API (bench.py): part 1 (global Modular streams + LF stage) on a side stream, part 2 on the
    main stream, for a batch mixing Modular frames, VarDCT frames with alpha, progressive and upsampled frames; two batches
    sharing one set of coefficient / pixel planes; repeated to cover the cached varblock-flag paths."""
    import torch
    img = S.synthetic_image(90, 600, 400)
    al = (np.add.outer(np.arange(400), np.arange(600)) % 256).astype(np.uint8)
    streams = [(S.encode_vardct(img, seed=1, strategy_mix=2, alpha=al), 4), (S.encode_modular(_smooth_image(9, 280, 300, 3, 8), 8, True, 1), 3),
               (S.encode_vardct(img, seed=2, strategy_mix=4, num_passes=2), 3), (S.encode_vardct(img, seed=3, strategy_mix=1, upsampling=2), 3),
               (S.encode_vardct(S.synthetic_image(91, 200, 136), seed=4, strategy_mix=3, epf_iters=3), 3)]
    want = [O.decode(d).pixels("u8", n) for d, n in streams]
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    batches = []
    for k in range(2):
        b = jx.BatchDecoder(0)
        for d, n in streams:
            b.add(d, "uint8", n)
        if k:
            b.share_buffers(batches[0])
        b.prepare(main.cuda_stream)
        batches.append(b)
    torch.cuda.synchronize()
    ev_front = [torch.cuda.Event() for _ in range(2)]
    ev_rest = [torch.cuda.Event() for _ in range(2)]
    for step in range(5):
        k = step % 2
        with torch.cuda.stream(side):
            if step >= 2:
                side.wait_event(ev_rest[k])
            batches[k].decode_part(1, side.cuda_stream)
            ev_front[k].record(side)
        main.wait_event(ev_front[k])
        batches[k].decode_part(2, main.cuda_stream)
        ev_rest[k].record(main)
        if step == 1:
            torch.cuda.synchronize()
            for b in batches:
                b.finish(main.cuda_stream)     # reads the varblock flags: later steps skip the kernels nobody needs
    torch.cuda.synchronize()
    for b in batches:
        b.finish(main.cuda_stream)
        for i in range(len(streams)):
            assert np.array_equal(b.output(i), want[i]), i


def test_corrupted_streams_fail_cleanly_or_decode(jx):
    """Robustness: random byte corruption in the section payloads (VarDCT and Modular streams) must end in a DecodeError or
    a decode of the right size — never a crash or a hang (the kernels bound every loop by the frame geometry and read
    zeros past the end of a section)."""
    rng = np.random.default_rng(123)
    img = S.synthetic_image(70, 300, 280)
    al = (np.arange(300 * 280) % 251).astype(np.uint8).reshape(280, 300)
    streams = [S.encode_vardct(img, seed=2, strategy_mix=2, epf_iters=2, gab=1, alpha=al),
               S.encode_modular(_smooth_image(8, 280, 300, 3, 8), 8, True, 1),
               S.encode_vardct(img, seed=2, strategy_mix=4, upsampling=2)]
    outcomes = {"error": 0, "decoded": 0}
    for data in streams:
        for trial in range(24):
            bad = bytearray(data)
            n = 1 + trial % 4
            for pos in rng.integers(24, len(bad), n):        # keep the signature / size header so the decode reaches the GPU
                bad[pos] ^= 1 << int(rng.integers(0, 8))
            if trial % 6 == 5:
                bad = bad[: int(rng.integers(len(bad) // 2, len(bad)))]   # truncation
            try:
                meta, px = jx.decoder_builder().decode_with(bytes(bad), np.uint8)
                assert len(px) == meta.width * meta.height * (4 if meta.has_alpha_channel else 3)
                outcomes["decoded"] += 1
            except jx.DecodeError:
                outcomes["error"] += 1
    assert outcomes["error"] > 0 and outcomes["error"] + outcomes["decoded"] == 72
    # the decoder is still healthy afterwards
    check_against_oracle(jx, streams[0], np.uint8, 4)


def _orient(a, o):
    """EXIF-style orientation o applied to an (h, w, c) array (codestream_header.rs JxlOrientation)."""
    return {1: lambda v: v, 2: lambda v: v[:, ::-1], 3: lambda v: v[::-1, ::-1], 4: lambda v: v[::-1], 5: lambda v: v.transpose(1, 0, 2),
            6: lambda v: v.transpose(1, 0, 2)[:, ::-1], 7: lambda v: v[::-1, ::-1].transpose(1, 0, 2), 8: lambda v: v.transpose(1, 0, 2)[::-1]}[o](a)


@pytest.mark.parametrize("o", [2, 3, 4, 5, 6, 7, 8])
def test_orientation_is_applied_by_the_write_stage(jx, o):
    """JxlBasicInfo.orientation / skip_reorientation (decode.rs:340-346): by default the decoder hands out the oriented image
    (dimensions swapped for 5..8, orientation reported as 1); with skip_reorientation the stored raster and the header value."""
    img = S.synthetic_image(40 + o, 200, 136)
    data = S.encode_vardct(img, seed=3, strategy_mix=2, epf_iters=1, gab=1, orientation=o)
    stored = O.decode(data).pixels("u8", 3).reshape(136, 200, 3)
    meta, px = jx.decoder_builder().decode_with(data, np.uint8)
    want = _orient(stored, o)
    assert (meta.height, meta.width) == want.shape[:2] and meta.orientation == 1
    assert np.array_equal(px.reshape(want.shape), want)
    meta, px = jx.decoder_builder(skip_reorientation=True).decode_with(data, np.uint8)
    assert (meta.width, meta.height, meta.orientation) == (200, 136, o)
    assert np.array_equal(px.reshape(136, 200, 3), stored)
    # float output with row alignment goes through the unfused write kernel
    _, pf = jx.decoder_builder(pixel_format=jx.PixelFormat(num_channels=3, align=64)).decode_with(data, np.float32)
    ref = O.decode(data).pixels("f32", 3).view(np.float32).reshape(136, 200, 3)
    wantf = _orient(ref, o)
    stride = (wantf.shape[1] * 12 + 63) // 64 * 64 // 4
    rows = np.stack([pf[r * stride: r * stride + wantf.shape[1] * 3] for r in range(wantf.shape[0])]).reshape(wantf.shape)
    assert ulp_diff(rows, wantf) <= 1


def test_full_size_8k_hdr_frame(jx):
    """BASELINE config 5: 7680x4320 f32 HDR VarDCT (linear, values up to 4.0, intensity_target 1000), EPF 3, <= 1 ULP vs the CPU decode."""
    lin = ((S.synthetic_image(6, 7680, 4320).astype(np.float32) / 255.0) ** 2.2) * 4.0
    data = S.encode_vardct(lin, seed=6, strategy_mix=1, epf_iters=3, gab=1, out_bits=32, hdr=1)
    meta, px = check_against_oracle(jx, data, np.float32, 3)
    assert (meta.width, meta.height) == (7680, 4320) and abs(meta.intensity_target - 1000.0) < 1e-3
    err = px.reshape(4320, 7680, 3) - lin
    assert float(np.sqrt((err.astype(np.float64) ** 2).mean())) < 0.05   # decodes to the source picture (size-independent property)


def test_full_size_4k_frame(jx):
    """BASELINE config 2: one 3840x2160 VarDCT d1 frame, u8 output, bit-exact vs the CPU decode."""
    img = S.synthetic_image(1000, 3840, 2160)
    data = S.encode_vardct(img, seed=1000, strategy_mix=1, epf_iters=1, gab=1)
    _, px = check_against_oracle(jx, data, np.uint8, 3)
    err = px.reshape(2160, 3840, 3).astype(np.float64) - img
    assert 10 * np.log10(255.0 ** 2 / (err ** 2).mean()) > 36.0     # size-independent property: decodes to the source picture


def test_lf_frames_of_images_with_alpha(jx):
    """An RGBA image whose LF image is an LF frame: the LF frame carries the image's extra channel as well (zeros at 1/8 scale in a `cjxl --progressive_dc` file); it is decoded
    with the frame and ignored, the alpha comes from the frame proper — plain or squeezed."""
    from test_synth_roundtrip import lf_frame_alpha_streams
    for name, stream, al in lf_frame_alpha_streams():
        meta, px = check_against_oracle(jx, stream, np.uint8, 4)
        assert meta.has_alpha_channel and np.array_equal(px.reshape(al.shape + (4,))[..., 3], al), name
        check_against_oracle(jx, stream, np.float32, 4)
        check_against_oracle(jx, stream, np.uint8, 3)


def test_full_size_4k_frame_with_lf_frame_and_prefix_codes(jx):
    """BASELINE config 2's size with what round 3 added: a 3840x2160 frame whose LF image is an LF frame (480x270, itself a VarDCT frame), both under
    prefix codes.  Bit-exact vs the CPU decode; decodes to the source picture (size-independent property)."""
    from test_synth_roundtrip import _block_means
    img = S.synthetic_image(1001, 3840, 2160)
    S.set_prefix(True)
    try:
        data = (S.encode_vardct_frame(img, S.frame(emit=2), seed=3)
                + S.encode_vardct_frame(_block_means(img), S.frame(emit=1, is_last=0, frame_type=1, lf_level=1), seed=5, distance=0.3, epf_iters=0, gab=0)
                + S.encode_vardct_frame(img, S.frame(emit=1, use_lf_frame=1), seed=1001, strategy_mix=1, epf_iters=1, gab=1))
    finally:
        S.set_prefix(False)
    _, px = check_against_oracle(jx, data, np.uint8, 3)
    err = px.reshape(2160, 3840, 3).astype(np.float64) - img
    assert 10 * np.log10(255.0 ** 2 / (err ** 2).mean()) > 36.0


def test_batch_api_mixed_sizes_and_lane_strides(jx):
    """Resident batch decode (include/jxl_hip.h JxlHipBatch*): frames of different sizes, several decode-thread packings,
    repeated decodes of the same prepared batch (idempotence)."""
    streams = []
    for i, (w, h, mix) in enumerate([(320, 200, 2), (64, 64, 0), (600, 520, 1), (320, 200, 1), (1030, 270, 2)]):
        streams.append(S.encode_vardct(S.synthetic_image(30 + i, w, h), seed=30 + i, strategy_mix=mix, epf_iters=i % 4, gab=i % 2))
    refs = [O.decode(s).pixels("u8", 3) for s in streams]
    for lf, hf in [(64, 64), (1, 1), (8, 16)]:
        b = jx.BatchDecoder(0)
        for s in streams:
            b.add(s, "uint8", 3)
        b.set_lane_stride(lf, hf)
        b.prepare()
        for _ in range(2):
            b.decode()
            b.finish()
            for i, r in enumerate(refs):
                assert np.array_equal(b.output(i), r), (lf, hf, i)
        assert b.total_pixels == sum(len(r) // 3 for r in refs)


@pytest.mark.parametrize("w,h,mix,epf", [(320, 200, 2, 1), (64, 64, 0, 0), (600, 520, 1, 2), (2100, 300, 1, 1)])
def test_prefix_coded_vardct_streams(jx, w, h, mix, epf):
    """What `cjxl -e 1..3` writes: LF (Modular) and AC streams under prefix (Huffman) codes instead of ANS.  The synthesiser's
    prefix writer and the oracle agree with the ANS form of the same frame (test_synth_roundtrip.py); here the HIP path — general symbol
    reader in the LF kernel, HfDecodeKernel for the AC streams — against the oracle, alone and in a batch beside ANS-coded frames."""
    img = S.synthetic_image(40 + w % 7, w, h)
    ans = S.encode_vardct(img, seed=5, strategy_mix=mix, epf_iters=epf, gab=1)
    S.set_prefix(True)
    try:
        pfx = S.encode_vardct(img, seed=5, strategy_mix=mix, epf_iters=epf, gab=1)
    finally:
        S.set_prefix(False)
    assert pfx != ans
    _, px = check_against_oracle(jx, pfx, np.uint8, 3)
    check_against_oracle(jx, pfx, np.float32, 3)
    ref = O.decode(ans).pixels("u8", 3)
    assert np.array_equal(px.reshape(-1), ref)                     # the same coefficients either way
    for lf_stride in (64, 4):
        b = jx.BatchDecoder(0)
        for s_ in (ans, pfx, pfx, ans):
            b.add(s_, "uint8", 3)
        b.set_lane_stride(lf_stride, 1)
        b.prepare(); b.decode(); b.finish()
        for i in range(4):
            assert np.array_equal(b.output(i), ref), (lf_stride, i)


def test_batch_objects_are_refilled_without_reallocating(jx):
    """The streaming loop of bench.py: JxlHipBatchReset + JxlHipBatchAddImages (parser threads) + JxlHipBatchPrepare on one batch object,
    again and again with other images (other sizes, other counts), a second object sharing its pixel and coefficient planes; SIMT LF decode
    and the one-wavefront-per-stream kernel ("lf_wide_once") alike.  Every refill decodes bit-exactly."""
    sets = []
    for r, shapes in enumerate([[(320, 200, 2), (600, 520, 1), (64, 64, 0)], [(600, 520, 2), (200, 136, 1)], [(1030, 270, 2), (320, 200, 0), (600, 520, 0), (24, 17, 1)]]):
        streams = [S.encode_vardct(S.synthetic_image(70 + 10 * r + i, w, h), seed=70 + 10 * r + i, strategy_mix=mix, epf_iters=(i + r) % 3, gab=(i + r) % 2) for i, (w, h, mix) in enumerate(shapes)]
        sets.append((streams, [O.decode(s).pixels("u8", 3) for s in streams]))
    big = max(sets, key=lambda t: sum(len(r) for r in t[1]))
    owner = jx.BatchDecoder(0); owner.set_lane_stride(4, 1)
    owner.add_many(big[0], "uint8", 3, threads=3); owner.prepare()
    sharer = jx.BatchDecoder(0); sharer.set_lane_stride(4, 1)
    sharer.share_buffers(owner); sharer.share_coefficients(owner)
    for rnd in range(5):
        for bi, b in enumerate((owner, sharer)):
            streams, refs = sets[(rnd + bi) % len(sets)]
            if b is sharer and sum(len(r) for r in refs) > sum(len(r) for r in big[1]):
                continue
            b.reset()
            assert b.add_many(streams, "uint8", 3, threads=1 + rnd % 3) == 0
            b.prepare()
            assert b.info_value("lf_simt_frames") == len(streams)
            if rnd % 2:
                b.set_option("lf_wide_once", 1)
            b.decode(); b.finish()
            for i, r in enumerate(refs):
                assert np.array_equal(b.output(i), r), (rnd, bi, i)
    with pytest.raises(jx.GenericError):                        # a bad image among many: nothing is appended
        owner.reset(); owner.add_many([sets[0][0][0], b"\xff\x0a" + bytes(40)], "uint8", 3, threads=2)


def test_lf_simt_weighted_predictor_trees(jx):
    """LF-group streams under the MA-tree shape of a default-effort cjxl encode (weighted-predictor leaves under a fixed tree over property 15;
    HF metadata under the row / N / W tree) take the SIMT LF kernel's weighted-predictor instantiation: every frame SIMT, HIP == oracle.
    Streams that spell out other predictor parameters, and — "lf_wp_narrow_test" — streams whose samples leave the range of the lanes'
    32-bit arithmetic, are handed back to the one-wavefront-per-stream kernel on the device and still decode bit-exactly; so do batches that
    mix tree shapes, single images (no SIMT) and the wide first launch of a pipeline."""
    from test_synth_roundtrip import cjxl_shape_streams
    cases = cjxl_shape_streams()
    refs = {}
    for name, shape, cj, plain in cases:
        if name not in refs:
            refs[name] = O.decode(plain).pixels("u8", 3)
        _, px = jx.decoder_builder().decode_with(cj, np.uint8)             # single image: one wavefront per stream
        assert np.array_equal(px.reshape(-1), refs[name].reshape(-1)), (name, shape)
    for shape in (1, 2):
        mine = [(n, cj) for n, sh, cj, _ in cases if sh == shape]
        for lf_stride, narrow, wide in ((4, 0, 0), (8, 0, 0), (1, 0, 0), (4, 1, 0), (4, 0, 1)):
            b = jx.BatchDecoder(0)
            b.set_lane_stride(lf_stride, 1)
            b.add_many([cj for _, cj in mine] * 2, "uint8", 3)
            b.set_option("lf_wp_narrow_test", narrow)
            b.prepare()
            assert b.info_value("lf_simt_frames") == 2 * len(mine) and b.info_value("lf_simt_wp") == 1
            if wide:
                b.set_option("lf_wide_once", 1)
            for _ in range(2):                                              # (decoded again: the hand-back marks are per decode)
                b.decode(); b.finish()
                for i, (n, _) in enumerate(mine * 2):
                    assert np.array_equal(b.output(i).reshape(-1), refs[n].reshape(-1)), (shape, lf_stride, narrow, wide, n)
    # mixed batch: weighted-predictor trees, gradient trees (SIMT without predictor state of their own), a prefix-coded frame (legacy kernel)
    S.set_prefix(True)
    try:
        pfx = S.encode_vardct(S.synthetic_image(81, 320, 200), seed=1, strategy_mix=1, epf_iters=1)
    finally:
        S.set_prefix(False)
    mixed = [(n, cj) for n, sh, cj, _ in cases if sh == 1][:3] + [(n, pl) for n, sh, _, pl in cases if sh == 2][:3] + [("small", pfx)]
    b = jx.BatchDecoder(0)
    b.set_lane_stride(4, 1)
    b.add_many([d for _, d in mixed], "uint8", 3)
    b.prepare()
    assert b.info_value("lf_simt_frames") == 6 and b.info_value("lf_legacy_frames") == 1 and b.info_value("lf_simt_wp") == 1
    b.decode(); b.finish()
    for i, (n, _) in enumerate(mixed):
        assert np.array_equal(b.output(i).reshape(-1), refs[n].reshape(-1)), (i, n)


def test_batch_reset_and_unsharing_leave_a_consistent_object(jx):
    """Round-3 advisor findings on Batch::Reset / ShareCoefArena / ShareBigArena / ClearCoefficientsBeforeHf: (1) Reset() really forgets the
    prepared state — sharing can be set up afterwards and a decode of the empty batch is a no-op; (2) leaving a sharing arrangement
    (owner = NULL) gives the sharer planes of its own and never frees the owner's; (3) a sharer whose coefficient layout reaches
    beyond what the owner's own content ever cleared starts from zeroed planes (hipMalloc'd memory is not)."""
    small = [S.encode_vardct(S.synthetic_image(31 + i, 200, 136), seed=31 + i, strategy_mix=2, epf_iters=1, gab=1) for i in range(2)]
    large = [S.encode_vardct(S.synthetic_image(41 + i, 600, 520), seed=41 + i, strategy_mix=2, epf_iters=1, gab=1) for i in range(3)]
    ref_small = [O.decode(s).pixels("u8", 3) for s in small]
    ref_large = [O.decode(s).pixels("u8", 3) for s in large]

    def run(b, refs):
        b.prepare(); b.decode(); b.finish()
        for i, r in enumerate(refs):
            assert np.array_equal(b.output(i), r), i

    owner = jx.BatchDecoder(0)
    owner.add_many(large, "uint8", 3); run(owner, ref_large)             # the owner's arenas get the size of the large set ...
    owner.reset(); owner.add_many(small, "uint8", 3); run(owner, ref_small)   # ... but its current content only covers a corner of them
    sharer = jx.BatchDecoder(0)
    sharer.share_buffers(owner); sharer.share_coefficients(owner)
    sharer.add_many(large, "uint8", 3); run(sharer, ref_large)           # (3): layout larger than the owner's current one
    run(owner, ref_small)
    # (1) after Reset the object is unprepared: sharing may change, an empty decode does nothing
    sharer.reset()
    sharer.share_coefficients(owner); sharer.share_buffers(owner)
    sharer.decode(); sharer.finish()
    assert sharer.total_pixels == 0
    sharer.reset()
    # (2) leave the arrangement: own planes again; the owner's planes are still there and still the owner's
    sharer.share_coefficients(None); sharer.share_buffers(None)
    sharer.add_many(large, "uint8", 3); run(sharer, ref_large)
    run(owner, ref_small)
    owner.reset(); owner.add_many(large, "uint8", 3); run(owner, ref_large)
    del sharer                                                            # destruction of a former sharer does not touch the owner either
    run(owner, ref_large)
    third = jx.BatchDecoder(0); third.share_buffers(owner); third.share_coefficients(owner)
    third.add_many(small, "uint8", 3); run(third, ref_small)
    del third
    owner.reset(); owner.add_many(small, "uint8", 3); run(owner, ref_small)


# ---- multi-frame images and image features (round 2): frame tail kernels vs the oracle -----------------------------------------------
def _stream_cases():
    img = S.synthetic_image(5, 200, 136)
    small = S.synthetic_image(9, 64, 48)
    big = S.synthetic_image(6, 600, 400)
    cases = {}
    m0 = S.encode_modular_frame(img, S.frame(is_last=0, save_as_reference=1), bits=8)
    for mode in (0, 1, 4):
        cases["modular_layer_mode%d" % mode] = m0 + S.encode_modular_frame(small, S.frame(emit=1, have_crop=1, crop_x0=40, crop_y0=30, canvas_w=200, canvas_h=136, blend_mode=mode, blend_source=1), bits=8)
    cases["modular_layer_offscreen"] = m0 + S.encode_modular_frame(small, S.frame(emit=1, have_crop=1, crop_x0=-20, crop_y0=100, canvas_w=200, canvas_h=136, blend_source=1), bits=8)
    al = np.full((136, 200, 1), 200, np.uint8)
    m0a = S.encode_modular_frame(np.dstack([img, al]), S.frame(is_last=0, save_as_reference=2), bits=8)
    sa = np.dstack([small, (np.add.outer(np.arange(48), np.arange(64)) * 3 % 256).astype(np.uint8)])
    for mode in (2, 3):
        cases["modular_alpha_mode%d" % mode] = m0a + S.encode_modular_frame(sa, S.frame(emit=1, have_crop=1, crop_x0=10, crop_y0=20, canvas_w=200, canvas_h=136, blend_mode=mode, blend_source=2, blend_clamp=mode == 3), bits=8)
    g0 = S.encode_vardct_frame(big, S.frame(is_last=0, save_as_reference=1), seed=3, strategy_mix=2)
    cases["vardct_overlay"] = g0 + S.encode_vardct_frame(small, S.frame(emit=1, have_crop=1, crop_x0=300, crop_y0=160, canvas_w=600, canvas_h=400, blend_source=1), seed=4)
    g1 = S.encode_vardct_frame(small, S.frame(emit=1, is_last=0, have_crop=1, crop_x0=100, crop_y0=60, canvas_w=600, canvas_h=400, blend_source=1), seed=4)
    cases["vardct_three_layers_add"] = g0 + g1 + S.encode_vardct_frame(img, S.frame(emit=1, have_crop=1, crop_x0=-30, crop_y0=300, canvas_w=600, canvas_h=400, blend_mode=1, blend_source=0), seed=5, epf_iters=2)
    cases["vardct_noise"] = S.encode_vardct_frame(big, S.frame(noise_lut=[30, 60, 90, 120, 150, 180, 210, 240]), seed=3, strategy_mix=2)
    cases["vardct_noise_small_epf3"] = S.encode_vardct_frame(small, S.frame(noise_lut=[200] * 8), seed=3, epf_iters=3)
    alpha = (np.add.outer(np.arange(400), np.arange(600)) % 256).astype(np.uint8)
    v0a = S.encode_vardct_frame(big, S.frame(is_last=0, save_as_reference=3), seed=3, alpha=alpha)
    cases["vardct_alpha_blend"] = v0a + S.encode_vardct_frame(img, S.frame(emit=1, have_crop=1, crop_x0=50, crop_y0=70, canvas_w=600, canvas_h=400, blend_mode=2, blend_source=3), seed=7,
                                                            alpha=(np.add.outer(np.arange(136), np.arange(200)) * 2 % 256).astype(np.uint8))
    return cases


@pytest.mark.parametrize("name", ["modular_layer_mode0", "modular_layer_mode1", "modular_layer_mode4", "modular_layer_offscreen", "modular_alpha_mode2", "modular_alpha_mode3",
                                  "vardct_overlay", "vardct_three_layers_add", "vardct_noise", "vardct_noise_small_epf3", "vardct_alpha_blend"])
def test_multi_frame_and_feature_streams(jx, name):
    """Reference / zero-duration layers, cropped frames (also partly outside the canvas), every frame blend mode, alpha, noise:
    the frame tail of the HIP path against the oracle (u8 exact, f32 <= 1 ULP)."""
    data = _stream_cases()[name]
    nch = 4 if "alpha" in name else 3
    check_against_oracle(jx, data, np.uint8, nch)
    check_against_oracle(jx, data, np.float32, nch)
    check_against_oracle(jx, data, np.uint16, 3)


def test_unpremultiply_alpha_setter(jx):
    """decode.rs:353 JxlDecoderSetUnpremultiplyAlpha: an image whose alpha is associated comes out divided by alpha (alpha.cc
    UnpremultiplyAlpha) when RGBA is requested, untouched otherwise."""
    img = S.synthetic_image(5, 200, 136)
    alpha = (64 + np.add.outer(np.arange(136), np.arange(200)) % 192).astype(np.uint8)
    data = S.encode_vardct_frame(img, S.frame(alpha_premultiplied=1), seed=3, alpha=alpha)
    ref = O.decode(data)
    plain = ref.pixels("u8", 4).copy()
    ref.set_unpremultiply_alpha(True)
    want = ref.pixels("u8", 4).copy()
    assert not np.array_equal(plain, want)
    _, px = jx.decoder_builder(unpremul_alpha=True).decode_with(data, np.uint8)
    assert np.array_equal(px, want)
    _, px = jx.decoder_builder(unpremul_alpha=False).decode_with(data, np.uint8)
    assert np.array_equal(px, plain)
    _, pf = jx.decoder_builder(unpremul_alpha=True).decode_with(data, np.float32)
    assert ulp_diff(pf, ref.pixels("f32", 4).view(np.float32)) <= 1
    _, p3 = jx.decoder_builder(unpremul_alpha=True, pixel_format=jx.PixelFormat(num_channels=3)).decode_with(data, np.uint8)
    assert np.array_equal(p3, ref.pixels("u8", 3))          # no alpha requested: nothing to divide by


def test_streaming_input_sequence(jx):
    """libjxl's streaming protocol (jpegxl-sys decode.rs:664-706): ProcessInput -> NEED_MORE_INPUT, ReleaseInput returns the
    unconsumed bytes, SetInput again with more data."""
    L = jx.libjxl()
    L.JxlDecoderReleaseInput.restype = C.c_size_t
    L.JxlDecoderReleaseInput.argtypes = [C.c_void_p]
    data = fixture_bytes("sample.jxl")
    dec = L.JxlDecoderCreate(None)
    try:
        assert L.JxlDecoderSubscribeEvents(dec, jx.JXL_DEC_BASIC_INFO | jx.JXL_DEC_FULL_IMAGE) == 0
        buf = np.frombuffer(data, np.uint8)
        for cut in (1, 100, 2000):
            assert L.JxlDecoderSetInput(dec, buf.ctypes.data, cut) == 0
            assert L.JxlDecoderProcessInput(dec) == jx.JXL_DEC_NEED_MORE_INPUT
            assert L.JxlDecoderReleaseInput(dec) == cut
        assert L.JxlDecoderSetInput(dec, buf.ctypes.data, len(data)) == 0
        L.JxlDecoderCloseInput(dec)
        assert L.JxlDecoderProcessInput(dec) == jx.JXL_DEC_BASIC_INFO
        assert L.JxlDecoderReleaseInput(dec) == 0
    finally:
        L.JxlDecoderDestroy(dec)
    with pytest.raises(jx.GenericError):       # two channels of a colour image: libjxl's "number of channels is too low"
        jx.decoder_builder(pixel_format=jx.PixelFormat(num_channels=2)).decode_with(data, np.uint8)


def test_against_real_libjxl_when_the_box_has_one(jx, capsys):
    """SURVEY §8c run-time oracle adapter: probe (in a subprocess — the soname collides with the look-alike) for a system
    libjxl.so*, djxl / cjxl and Python JXL plugins.  If one exists, every committed stream and every reference fixture is decoded
    with it and compared with the HIP path: u8 / u16 exact, f32 <= 1 ULP (BASELINE north_star).  If none exists the test says
    exactly what was probed and skips — the float pipeline then stays PARITY-UNPINNED against libjxl."""
    import libjxl_probe as P
    found = P.probe()
    with capsys.disabled():
        print("\n[libjxl probe] " + P.describe(found))
    if not found["available"]:
        pytest.skip(P.describe(found))
    names = [os.path.join(FIXTURES, n) for n in ("sample.jxl", "sample_grey.jxl", "2bit.jxl", "sample_jpg.jxl", "bench.jxl")]
    names += sorted(os.path.join(GOLDEN, n) for n in os.listdir(GOLDEN) if n.endswith(".jxl"))
    checked = 0
    for path in names:
        data = open(path, "rb").read()
        info = O.decode(data).info
        nch = 1 if info.num_color_channels == 1 else 3
        for dtype, npdt in (("u8", np.uint8), ("u16", np.uint16), ("f32", np.float32)):
            ref = P.decode(found, data, dtype, nch)
            if ref is None:
                continue
            _, px = jx.decoder_builder(pixel_format=jx.PixelFormat(num_channels=nch)).decode_with(data, npdt)
            assert px.shape == ref.shape, path
            if dtype == "f32":
                assert ulp_diff(px, ref.astype(np.float32)) <= 1, (path, dtype)
            else:
                assert np.array_equal(px, ref), (path, dtype, int((px != ref).sum()))
            checked += 1
    assert checked > 0


def test_jpeg_reconstruction(jx):
    """tests/decode.rs:123-139 (`jpeg`): reconstruct() of the JPEG-transcoded fixture yields Data::Jpeg — here byte-identical to
    samples/sample.jpg — growing the caller's buffer from 512 bytes through JXL_DEC_JPEG_NEED_MORE_OUTPUT; a file without a jbrd box
    falls back to Data::Pixels(Uint16)."""
    want = open(os.path.join(FIXTURES, "sample.jpg"), "rb").read()
    dec = jx.decoder_builder(init_jpeg_buffer=512)
    meta, (kind, val) = dec.reconstruct(fixture_bytes("sample_jpg.jxl"))
    assert kind == "jpeg" and val == want
    assert (meta.width, meta.height) == (40, 50)
    from PIL import Image
    import io
    assert Image.open(io.BytesIO(val)).size == (40, 50)
    meta, (kind, val) = dec.reconstruct(fixture_bytes("sample.jxl"))      # same decoder object again (Reset on success)
    assert kind == "pixels" and val.dtype == np.uint16 and len(val) == 40 * 50 * 4
    meta, (kind, val) = jx.decoder_builder().reconstruct(fixture_bytes("sample_jpg.jxl"))   # default 512 KiB buffer: one shot
    assert kind == "jpeg" and val == want
    # the pixel path of the same file is unaffected
    check_against_oracle(jx, fixture_bytes("sample_jpg.jxl"), np.uint8, 3)


def _bump(tmp_path, size):
    import subprocess
    so = str(tmp_path / "libbump.so")
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", so, os.path.join(os.path.dirname(__file__), "bump_alloc.c")])
    B = C.CDLL(so)
    B.bump_create.restype = C.c_void_p; B.bump_create.argtypes = [C.c_size_t]
    B.bump_destroy.argtypes = [C.c_void_p]
    B.bump_stats.argtypes = [C.c_void_p, C.POINTER(C.c_size_t * 5)]
    return B, B.bump_create(size)


def test_memory_manager_bump_arena(jx, tmp_path):
    """memory.rs:128-138 (`test_mm`): a whole decode of sample.jxl through a 50 MiB bump allocator that never frees.  Every host
    allocation the decoder makes for this image (decoder object, Batch, parser state, tables, staging) comes out of the caller's
    arena, the cumulative total stays far below the arena size, and every block is handed back by Destroy."""
    B, arena = _bump(tmp_path, 50 << 20)
    mm = jx.JxlMemoryManager(arena, C.cast(B.bump_alloc, C.c_void_p), C.cast(B.bump_free, C.c_void_p))
    dec = jx.decoder_builder(memory_manager=mm, icc_profile=True)
    meta, px = dec.decode_with(fixture_bytes("sample.jxl"), np.uint8)
    assert np.array_equal(np.frombuffer(px, np.uint8), O.decode(fixture_bytes("sample.jxl")).pixels("u8", meta.num_color_channels + (1 if meta.has_alpha_channel else 0)))
    st = (C.c_size_t * 5)()
    B.bump_stats(arena, C.byref(st))
    used, allocs, frees, failed, largest = list(st)
    print(f"bump arena: {used} bytes in {allocs} allocations ({frees} freed so far), largest {largest}")
    assert failed == 0 and allocs > 20 and 64 << 10 < used < 50 << 20
    meta, (kind, val) = dec.reconstruct(fixture_bytes("sample_jpg.jxl"))        # JPEG path through the same arena
    assert kind == "jpeg" and val == open(os.path.join(FIXTURES, "sample.jpg"), "rb").read()
    del dec
    import gc
    gc.collect()
    B.bump_stats(arena, C.byref(st))
    assert st[3] == 0 and st[1] == st[2], list(st)      # balanced: nothing allocated through the manager is leaked or freed elsewhere
    B.bump_destroy(arena)


def test_memory_manager_out_of_memory_is_an_error_not_a_crash(jx, tmp_path):
    """An arena too small for the decode: allocation failure surfaces as JXL_DEC_ERROR (DecodeError::GenericError), not a crash."""
    B, arena = _bump(tmp_path, 16 << 10)
    mm = jx.JxlMemoryManager(arena, C.cast(B.bump_alloc, C.c_void_p), C.cast(B.bump_free, C.c_void_p))
    dec = jx.decoder_builder(memory_manager=mm)
    with pytest.raises(jx.DecodeError):
        dec.decode_with(fixture_bytes("sample.jxl"), np.uint8)
    del dec
    B.bump_destroy(arena)


from free_cases import FREE_CASES


@pytest.mark.parametrize("name", sorted(FREE_CASES))
def test_free_running_modular_streams(jx, name):
    """Modular features no encoder run is available for (SURVEY.md B.5 [R] rows): random MA trees over every property incl. the
    weighted predictor's error (15) and previous-channel properties (>= 16), all 14 predictors with offsets / multipliers, custom
    WP headers, per-section local trees, LZ77 with special distances, palettes with delta entries and a predictor.  The streams
    come from tools/synth_free.h (one shared histogram, so any token sequence is a valid stream); the pixels are whatever a
    decoder makes of them and the HIP path must make the same of them as the oracle, at 16 bits so that little is clamped."""
    kw = dict(FREE_CASES[name])
    kw.setdefault("bits", 16)
    data = S.encode_modular_free(**kw)
    nch = (kw.get("nchan", 3)) + (1 if kw.get("has_alpha") else 0)
    check_against_oracle(jx, data, np.uint16, nch)
    # float output is not clamped: every decoded integer takes part (sample / 65535 in both decoders)
    _, pf = check_against_oracle(jx, data, np.float32, nch)
    assert np.unique(pf).size > (20 if "palette" in name else 200)
    if "palette_delta_every" in name:
        for pred in range(14):
            d2 = S.encode_modular_free(**dict(kw, pal_pred=pred, seed=100 + pred))
            check_against_oracle(jx, d2, np.float32, nch)


def test_repeated_decodes_leave_clean_coefficient_planes(jx):
    """The HF stage writes only non-zero coefficients and the IDCT kernels zero what they consumed (no dense clear between the
    decodes of a resident batch): every kernel family that reads coefficients — 32x32 and 64x64 tiles, the special 8x8 transforms,
    IDENTITY / DCT2X2 / AFV (halves of a block per lane inside the tile kernel since round 5), DCT128/256 (BigIdctKernel), unaligned varblocks and the forced generic
    kernel, progressive passes (accumulating) — must leave the planes such that a second and a third decode give the same
    pixels."""
    img = S.synthetic_image(21, 520, 300)
    al = (np.add.outer(np.arange(300), np.arange(520)) % 256).astype(np.uint8)
    families = {
        "tiles_plain": [S.encode_vardct(img, seed=3, strategy_mix=0), S.encode_vardct(S.synthetic_image(22, 300, 200), seed=4, strategy_mix=1, epf_iters=2)],
        "special_and_rare": [S.encode_vardct(img, seed=5, strategy_mix=2, epf_iters=1, gab=1)] + [S.encode_vardct(S.synthetic_image(40 + s, 96, 80), seed=s, strategy_mix=100 + s) for s in (1, 2, 3, 12, 13, 14, 15, 16, 17)],
        "unaligned_generic": [S.encode_vardct(img, seed=8, strategy_mix=3, epf_iters=1, gab=1)],
        "big": [S.encode_vardct(img, seed=9, strategy_mix=4), S.encode_vardct(img, seed=10, strategy_mix=5, epf_iters=3, gab=1)],
        "passes": [S.encode_vardct(img, seed=11, strategy_mix=2, num_passes=3, alpha=al), S.encode_vardct(img, seed=12, strategy_mix=1, num_passes=2)],
    }
    for name, streams in families.items():
        refs = [O.decode(s).pixels("u8", 3) for s in streams]
        for force in ((0, 1) if name == "tiles_plain" else (0,)):
            b = jx.BatchDecoder(0)
            for s in streams:
                b.add(s, "uint8", 3)
            b.set_option("force_generic_idct", force)
            b.prepare()
            for rep in range(3):
                b.decode()
                b.finish()
                for i, r in enumerate(refs):
                    assert np.array_equal(b.output(i), r), (name, force, rep, i)


def test_images_with_an_embedded_icc_profile(jx):
    """decode.rs:368-385 with icc_profile(true) on files whose colour encoding is an embedded ICC profile: non-XYB samples are
    handed out untouched with the embedded profile; XYB images are rendered to sRGB (libjxl without a CMS) and report the sRGB
    profile for the pixel data.  Pixels equal the oracle's either way."""
    from PIL import ImageCms
    icc = ImageCms.ImageCmsProfile(ImageCms.createProfile("sRGB")).tobytes()
    img = S.synthetic_image(8, 300, 200).astype(np.int32)
    S.set_icc(icc)
    try:
        lossless = S.encode_modular(np.dstack([img, 255 - img[..., :1]]), 8, True)
        lossy = S.encode_vardct(S.synthetic_image(8, 300, 200), seed=2, strategy_mix=2)
    finally:
        S.set_icc(b"")
    meta, px = jx.decoder_builder(icc_profile=True).decode_with(lossless, np.uint8)
    assert meta.icc_profile == icc and np.array_equal(px, O.decode(lossless).pixels("u8", 4))
    meta, px = jx.decoder_builder(icc_profile=True).decode_with(lossy, np.uint8)
    assert meta.icc_profile == jx.icc_profile_from_headers(S.encode_vardct(S.synthetic_image(8, 64, 64), seed=2))    # the enumerated sRGB profile
    assert np.array_equal(px, O.decode(lossy).pixels("u8", 3))


def _feature_streams():
    """Synthetic patch dictionaries and splines (tools/jxl_synth.cc WriteFeatures): every patch blend mode (none, replace, add,
    multiply, blend above / below, alpha-weighted add above / below), clamping, several reference frames, patches touching the
    frame edges, splines of one and several control points with colour and sigma DCTs, alone and together with noise."""
    ref_a = S.synthetic_image(50, 64, 48)
    ref_b = S.synthetic_image(52, 40, 72)
    main = S.synthetic_image(51, 200, 136)
    colour = [[0] * 32 for _ in range(3)]
    colour[1][0] = 300; colour[0][0] = 40; colour[2][1] = -60; colour[1][3] = 25
    sigma = [0] * 32
    sigma[0] = 30; sigma[2] = 4
    splines = (1, [(20, 30, [(15, 5), (2, -3), (-4, 6)], colour, sigma), (150, 20, [(-10, 20)], colour, sigma), (100, 100, [], colour, sigma)])
    cases = {}
    hdr = dict(frame_type=2, is_last=0, save_before_ct=1, have_crop=1, canvas_w=200, canvas_h=136)
    ra = S.encode_vardct_frame(ref_a, S.frame(save_as_reference=1, **hdr), seed=3)
    rb = S.encode_vardct_frame(ref_b, S.frame(emit=1, save_as_reference=2, **hdr), seed=5)
    patches = [(1, 4, 6, 20, 16, [(10, 12, [(1, 0, 0)]), (100, 50, [(2, 0, 0)]), (150, 100, [(3, 0, 1)]), (180, 120, [(2, 0, 0)]), (0, 0, [(0, 0, 0)])]),
               (2, 3, 20, 8, 8, [(0, 128, [(1, 0, 0)]), (192, 0, [(3, 0, 0)]), (96, 64, [(2, 0, 0)])]),
               (1, 0, 0, 64, 48, [(70, 80, [(2, 0, 0)])])]

    def with_features(fn, **feat):
        S.set_features(**feat)
        try:
            return fn()
        finally:
            S.set_features()
    cases["patches"] = ra + rb + with_features(lambda: S.encode_vardct_frame(main, S.frame(emit=1), seed=4), patches=patches)
    cases["splines"] = with_features(lambda: S.encode_vardct_frame(main, S.frame(), seed=4), splines=splines)
    cases["patches_splines_noise"] = ra + rb + with_features(lambda: S.encode_vardct_frame(main, S.frame(emit=1, noise_lut=[40, 80, 120, 160, 200, 240, 280, 320]), seed=4, epf_iters=2),
                                                            patches=patches, splines=splines)
    # alpha blend modes: image with alpha; the reference frame carries alpha too
    al_ref = (np.add.outer(np.arange(48), np.arange(64)) * 3 % 256).astype(np.uint8)
    al_main = (64 + np.add.outer(np.arange(136), np.arange(200)) % 192).astype(np.uint8)
    raa = S.encode_vardct_frame(ref_a, S.frame(save_as_reference=1, **hdr), seed=3, alpha=al_ref)
    pa = [(1, 2, 2, 24, 20, [(5, 5, [(4, 0, 0), (1, 0, 0)]), (60, 10, [(5, 0, 1), (2, 0, 0)]), (120, 40, [(6, 0, 0), (0, 0, 0)]), (30, 90, [(7, 0, 1), (3, 0, 1)]), (170, 110, [(4, 0, 1), (4, 0, 1)])])]
    cases["patches_alpha_modes"] = raa + with_features(lambda: S.encode_vardct_frame(main, S.frame(emit=1), seed=4, alpha=al_main), patches=pa, num_extra=1)
    # Modular main frame with patches from a Modular reference frame (integer samples through the same tail)
    mref = S.encode_modular_frame(ref_a, S.frame(frame_type=2, is_last=0, save_as_reference=3, save_before_ct=1, have_crop=1, canvas_w=200, canvas_h=136), bits=8)
    cases["patches_modular"] = mref + with_features(lambda: S.encode_modular_frame(main, S.frame(emit=1), bits=8), patches=[(3, 8, 8, 30, 30, [(20, 20, [(1, 0, 0)]), (100, 60, [(2, 0, 0)])])])
    return cases


@pytest.mark.parametrize("name", ["patches", "splines", "patches_splines_noise", "patches_alpha_modes", "patches_modular"])
def test_synthetic_patches_and_splines(jx, name):
    """The two real-encoder fixtures exercise one patch blend mode and one spline each; these streams cover the rest of
    dec_patch_dictionary.cc / splines.cc as the oracle restates them, HIP frame tail against oracle."""
    data = _feature_streams()[name]
    nch = 4 if "alpha" in name else 3
    assert np.unique(O.decode(data).pixels("u8", nch)).size > 50
    check_against_oracle(jx, data, np.uint8, nch)
    check_against_oracle(jx, data, np.float32, nch)
    check_against_oracle(jx, data, np.uint16, 3)


def test_corrupted_feature_streams_fail_cleanly_or_decode(jx):
    """Same robustness bar as test_corrupted_streams_fail_cleanly_or_decode for what round 2 added: the two real-encoder fixtures
    (patches, splines, AFV, prefix codes), multi-frame / patch / spline synth streams, free-running Modular streams with
    LZ77, local trees, weighted predictor, delta palettes, an embedded ICC profile, and the JPEG reconstruction path."""
    from free_cases import FREE_CASES
    rng = np.random.default_rng(321)
    feats = _feature_streams()
    streams = [fixture_bytes("sample_grey.jxl"), fixture_bytes("2bit.jxl"), feats["patches_splines_noise"], feats["patches_alpha_modes"], feats["patches_modular"]]
    for name in ("gray_alpha_16bit_everything", "lz77_local_trees", "palette_delta_wp_sections", "previous_channel_properties_groups", "local_tree_everywhere"):
        streams.append(S.encode_modular_free(**dict(FREE_CASES[name], bits=16)))
    streams += [S.encode_ycbcr(S.synthetic_image(70, 300, 280), "420", seed=3), S.encode_ycbcr(S.synthetic_image(71, 203, 139), "mixed", seed=4)]
    outcomes = {"error": 0, "decoded": 0}
    for data in streams:
        for trial in range(16):
            bad = bytearray(data)
            for pos in rng.integers(12, len(bad), 1 + trial % 3):
                bad[pos] ^= 1 << int(rng.integers(0, 8))
            if trial % 8 == 7:
                bad = bad[: int(rng.integers(len(bad) // 2, len(bad)))]
            try:
                meta, px = jx.decoder_builder().decode_with(bytes(bad), np.uint8)
                assert len(px) == meta.width * meta.height * (meta.num_color_channels + (1 if meta.has_alpha_channel else 0))
                outcomes["decoded"] += 1
            except jx.DecodeError:
                outcomes["error"] += 1
    assert outcomes["error"] > 0 and outcomes["error"] + outcomes["decoded"] == 16 * len(streams)
    # JPEG reconstruction with a damaged jbrd box / codestream: JPEG bytes, pixels or an error
    import jpeg_cases as JC
    import jpeg_tools as J
    data = fixture_bytes("sample_jpg.jxl")
    for src in (data, J.transcode(JC.jpeg_bytes(JC.CASES[4])), J.transcode(JC.jpeg_bytes(JC.CASES[5]))):   # + 4:2:0 / 4:2:2 with restart markers
        for trial in range(24):
            bad = bytearray(src)
            for pos in rng.integers(40, len(bad), 1 + trial % 2):
                bad[pos] ^= 1 << int(rng.integers(0, 8))
            try:
                meta, (kind, val) = jx.decoder_builder().reconstruct(bytes(bad))
                assert kind in ("jpeg", "pixels") and len(val) > 0
            except jx.DecodeError:
                pass
    check_against_oracle(jx, fixture_bytes("sample_grey.jxl"), np.uint8, 3)
    meta, (kind, val) = jx.decoder_builder().reconstruct(data)
    assert kind == "jpeg" and val == open(os.path.join(FIXTURES, "sample.jpg"), "rb").read()


def test_image_out_callback(jx):
    """jpegxl-sys decode.rs:289-309, :1172 JxlDecoderSetImageOutCallback: rows arrive through the callback (x = 0, one row each, in
    order) and add up to the same pixels as the buffer API; a buffer and a callback exclude each other."""
    L = jx.libjxl()
    CB = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p)
    L.JxlDecoderSetImageOutCallback.restype = C.c_int
    L.JxlDecoderSetImageOutCallback.argtypes = [C.c_void_p, C.c_void_p, CB, C.c_void_p]
    data = S.encode_vardct(S.synthetic_image(12, 300, 200), seed=3, strategy_mix=2)
    buf = np.frombuffer(data, np.uint8)
    rows = {}

    def cb(opaque, x, y, n, pixels):
        assert x == 0 and y not in rows
        rows[y] = np.ctypeslib.as_array(C.cast(pixels, C.POINTER(C.c_uint16)), shape=(n * 3,)).copy()
    cbf = CB(cb)
    dec = L.JxlDecoderCreate(None)
    assert L.JxlDecoderSubscribeEvents(dec, jx.JXL_DEC_BASIC_INFO | jx.JXL_DEC_FULL_IMAGE) == 0
    assert L.JxlDecoderSetInput(dec, buf.ctypes.data, len(buf)) == 0
    L.JxlDecoderCloseInput(dec)
    fmt = jx.JxlPixelFormat(3, jx.JXL_TYPE_UINT16, jx.JXL_NATIVE_ENDIAN, 0)
    events = []
    while True:
        st = L.JxlDecoderProcessInput(dec)
        events.append(st)
        if st == jx.JXL_DEC_NEED_IMAGE_OUT_BUFFER:
            assert L.JxlDecoderSetImageOutCallback(dec, C.byref(fmt), cbf, None) == 0
            dummy = np.zeros(16, np.uint8)
            assert L.JxlDecoderSetImageOutBuffer(dec, C.byref(fmt), dummy.ctypes.data, 16) == 1      # already set
        elif st in (jx.JXL_DEC_SUCCESS, jx.JXL_DEC_ERROR):
            break
    assert events[-1] == jx.JXL_DEC_SUCCESS, jx.last_error()
    L.JxlDecoderDestroy(dec)
    ref = O.decode(data)
    want = ref.pixels("u16", 3).view(np.uint16)
    got = np.concatenate([rows[y] for y in range(len(rows))])
    assert len(rows) == 200 and np.array_equal(got, want)


@pytest.mark.parametrize("sub", ["420", "422", "440", "mixed", "444"])
@pytest.mark.parametrize("w,h", [(200, 136), (203, 139), (64, 48), (530, 300), (2100, 270)])
def test_chroma_subsampled_ycbcr_frames(jx, sub, w, h):
    """SURVEY row b10 (stage_chroma_upsampling.cc, dec_group.cc with !Is444()): YCbCr VarDCT frames whose chroma (or luma, "mixed")
    channels live on coarser block grids — what a 4:2:0 / 4:2:2 / 4:4:0 JPEG becomes.  Streams from tools/synth_ycbcr.h (the encoder
    and the oracle's decoder were written separately and reproduce the source image); the HIP path — per-channel LF grids in the
    LF kernel, the skip rule and per-channel non-zero contexts in the HF kernel, IdctSubsampledKernel, ChromaUpsampleKernel —
    must match the oracle bit for bit, odd sizes (padding to whole cells, clamped upsampling taps) and several groups included."""
    img = S.synthetic_image(60 + w % 7, w, h)
    data = S.encode_ycbcr(img, sub, seed=w + h, distance=0.7)
    ref = O.decode(data).image("u8", 3)
    assert np.abs(ref.astype(int) - img.astype(int)).mean() < 6          # the stream does describe the source image
    check_against_oracle(jx, data, np.uint8, 3)
    check_against_oracle(jx, data, np.float32, 3)
    check_against_oracle(jx, data, np.uint16, 4 if False else 3)


@pytest.mark.gpu
@pytest.mark.parametrize("case", __import__("jpeg_cases").CASES + __import__("jpeg_cases").PROGRESSIVE,
                         ids=lambda c: "%dx%d_ss%d_q%d%s" % (c[:4] + ("_progressive" if c[4].get("progressive") else "",)))
def test_jpeg_transcodes_of_real_jpegs(jx, case):
    """decode.rs:493-514 `reconstruct` on JPEG XL files that stand for lossless transcodes of real JPEGs: Pillow's libjpeg writes the
    JPEG (4:4:4 / 4:2:2 / 4:2:0, optimised tables, restart intervals, COM marker; baseline and progressive scan scripts), tests/jpeg_tools.py
    turns it into jbrd box + VarDCT
    codestream (RAW quantisation tables, subsampled chroma grids).  reconstruct() must give back the JPEG byte for byte — entropy stages
    on the GPU, per-component coefficient planes (JpegCoefKernel), MCU interleave with the sampling factors from the frame header — and
    the pixel path must match the oracle bit for bit and libjpeg's own decode of the JPEG within integer-IDCT distance."""
    import jpeg_cases as JC
    import jpeg_tools as J
    data = JC.jpeg_bytes(case)
    jxl = J.transcode(data)
    meta, (kind, val) = jx.decoder_builder().reconstruct(jxl)
    assert kind == "jpeg" and val == data
    assert (meta.width, meta.height) == case[:2]
    check_against_oracle(jx, jxl, np.uint8, 3)
    check_against_oracle(jx, jxl, np.float32, 3)
    res = jx.decoder_builder(pixel_format=jx.PixelFormat(num_channels=3)).decode_with(jxl, np.uint8)[1]
    d = np.abs(np.asarray(res).reshape(case[1], case[0], 3).astype(int) - JC.pil_pixels(data))
    assert d.max() <= 6 and d.mean() < 0.7
    meta, (kind, val) = jx.decoder_builder(init_jpeg_buffer=256).reconstruct(jxl)      # grown through JXL_DEC_JPEG_NEED_MORE_OUTPUT
    assert kind == "jpeg" and val == data


@pytest.mark.gpu
def test_jpeg_transcode_metadata_from_boxes(jx):
    """The layout cjxl gives a transcoded photo: the ICC profile travels in the codestream's image header, Exif and XMP in their own
    boxes (Brotli-compressed `brob` boxes by default), and jbrd only records the sizes of the APP1 / APP2 markers (jpeg_data.h
    AppMarkerType kICC / kExif / kXMP; decode.cc SetJPEGDataFromICC, decode_to_jpeg.cc SetExif / SetXmp).  reconstruct() has to put
    the markers back — a 150 KB profile spans three APP2 chunks — and give the libjpeg-written file back byte for byte; a file whose
    boxes do not match what jbrd announces falls back to pixels (decode.rs:507-513) instead of emitting a damaged JPEG."""
    import io
    import struct
    from PIL import Image, ImageCms
    import jpeg_cases as JC
    import jpeg_tools as J
    icc = ImageCms.ImageCmsProfile(ImageCms.createProfile("sRGB")).tobytes()
    icc += bytes(np.random.default_rng(1).integers(0, 255, 150000).astype(np.uint8))
    ex = Image.Exif()
    ex[0x010E] = "a test image"
    ex[0x0131] = "jxl-hip tests"
    buf = io.BytesIO()
    Image.fromarray(JC.photo(67, 45)).save(buf, "JPEG", quality=85, subsampling=2, icc_profile=icc, exif=ex.tobytes(),
                                           xmp=b"<x:xmpmeta xmlns:x='adobe:ns:meta/'><!-- packet --></x:xmpmeta>")
    data = buf.getvalue()
    kinds = [J.app_type(a) for a in J.parse_jpeg(data).app_data]
    assert kinds.count(1) == 3 and 2 in kinds and 3 in kinds
    for kw in (dict(), dict(compress_boxes=True), dict(compress_boxes=True, jbrd_last=True)):
        jxl = J.transcode(data, typed_metadata=True, **kw)
        meta, (kind, val) = jx.decoder_builder().reconstruct(jxl)
        assert kind == "jpeg" and val == data, kw
        assert jx.decoder_builder(icc_profile=True).decode_with(jxl, np.uint8)[0].icc_profile == icc
    check_against_oracle(jx, jxl, np.uint8, 3)
    # damaged metadata: drop the Exif box / change the size of the xml box -> pixels
    jxl = J.transcode(data, typed_metadata=True)
    pos, boxes = 12, []
    while pos < len(jxl):
        n, t = struct.unpack(">I4s", jxl[pos:pos + 8])
        boxes.append((t, jxl[pos:pos + n]))
        pos += n
    no_exif = jxl[:12] + b"".join(b for t, b in boxes if t != b"Exif")
    longer = jxl[:12] + b"".join(struct.pack(">I4s", len(b) + 1, t) + b[8:] + b" " if t == b"xml " else b for t, b in boxes)
    for bad in (no_exif, longer):
        meta, (kind, val) = jx.decoder_builder().reconstruct(bad)
        assert kind == "pixels" and val.dtype == np.uint8 and len(val) == 67 * 45 * 3


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(progressive=True), dict(restart_marker_rows=1)], ids=["baseline", "progressive", "restarts"])
def test_grey_jpeg_transcodes(jx, kw):
    """One-component JPEGs written by libjpeg: reconstruct() == the file, grey pixels == oracle (and within 2 of libjpeg's decode)."""
    import io
    from PIL import Image
    import jpeg_cases as JC
    import jpeg_tools as J
    for (w, h) in ((75, 52), (300, 270)):
        data = JC.grey_jpeg_bytes(w, h, 85, **kw)
        jxl = J.transcode(data)
        meta, (kind, val) = jx.decoder_builder().reconstruct(jxl)
        assert kind == "jpeg" and val == data
        check_against_oracle(jx, jxl, np.uint8, 1)
        check_against_oracle(jx, jxl, np.float32, 1)
        res = jx.decoder_builder(pixel_format=jx.PixelFormat(num_channels=1)).decode_with(jxl, np.uint8)[1]
        d = np.abs(np.asarray(res).reshape(h, w).astype(int) - np.asarray(Image.open(io.BytesIO(data))).astype(int))
        assert d.max() <= 2 and d.mean() < 0.5


COLOUR_ENCODINGS = {
    "p3_srgb_tf": dict(white_point=1, primaries=11, tf=13),
    "bt2100_pq_1000": dict(white_point=1, primaries=9, tf=16, intensity_target=1000.0),
    "bt2100_hlg_1000": dict(white_point=1, primaries=9, tf=18, intensity_target=1000.0),
    "srgb_pq_255": dict(white_point=1, primaries=1, tf=16),
    "srgb_hlg_300": dict(white_point=1, primaries=1, tf=18, intensity_target=300.0),
    "dci_white_p3_gamma26": dict(white_point=11, primaries=11, tf=17),
    "bt2100_rec709_tf": dict(white_point=1, primaries=9, tf=1),
    "white_e_gamma22": dict(white_point=10, primaries=1, gamma=1 / 2.2),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(COLOUR_ENCODINGS))
def test_output_colour_encodings(jx, name):
    """XYB images that name their own primaries, white point and transfer function (dec_xyb.cc SetColorEncoding: inverse opsin matrix
    times linear-sRGB -> own primaries; stage_from_linear.cc: sRGB, Rec.709, gamma, PQ, HLG with the inverse OOTF): the fast path (default
    filters, OutputKernel), the feature path (ColorKernel, a frame with patches-style tail: two frames blended) and an upsampled frame
    against the oracle, all output types.  The oracle's side of these encodings is pinned by tests/test_oracle_goldens.py."""
    img = S.synthetic_image(31, 200, 136)
    S.set_color(**COLOUR_ENCODINGS[name])
    try:
        plain = S.encode_vardct(img, seed=5, strategy_mix=2)                                  # gaborish + EPF 1: unfused filters + OutputKernel
        nofilter = S.encode_vardct(img, seed=6, gab=0, epf_iters=0)
        layered = S.encode_vardct_frame(img, S.frame(is_last=0, save_as_reference=1), seed=3) + \
            S.encode_vardct_frame(S.synthetic_image(9, 64, 48), S.frame(emit=1, have_crop=1, crop_x0=40, crop_y0=30, canvas_w=200, canvas_h=136, blend_mode=1, blend_source=1), seed=4)
    finally:
        S.set_color()
    for data in (plain, nofilter, layered):
        check_against_oracle(jx, data, np.uint8, 3)
        check_against_oracle(jx, data, np.uint16, 3)
        check_against_oracle(jx, data, np.float32, 3)
    srgb = O.decode(S.encode_vardct(img, seed=5, strategy_mix=2)).image("u8", 3).astype(int)
    assert np.abs(O.decode(plain).image("u8", 3).astype(int) - srgb).max() > 8                 # (the encoding does change the pixels)


@pytest.mark.gpu
@pytest.mark.parametrize("bits,exp_bits", [(16, 5), (32, 8), (24, 7)])
def test_float_modular_samples(jx, bits, exp_bits):
    """Lossless float images (BitDepth.float_sample; dec_modular.cc int_to_float): binary16 / binary32 / 24-bit samples through the plain
    Modular output kernel and — two frames, the second blended on — through the float planes of the frame tail; f32 output is the exact
    float, integer outputs clamp and scale.  decode() hands out Float16 / Float pixels for such files (result.rs:65-72)."""
    rng = np.random.default_rng(bits)
    h, w = 70, 90
    if bits == 16:
        vals = np.concatenate([np.arange(0, 0x3C01), np.arange(0x8000, 0xBC01)]).astype(np.uint16)       # [-1, 1] incl. subnormals, -0
        ints = rng.choice(vals, (h, w, 3)).astype(np.int32)
    elif bits == 32:
        ints = (rng.random((h, w, 3), dtype=np.float32) * 1.25 - 0.125).view(np.int32).copy()
        ints &= 0x7FFFFFFF                                                                                 # (keeps the synthesiser's residuals in 32 bits)
    else:
        ints = rng.integers(0, 64 << 16, (h, w, 3)).astype(np.int32)                                       # up to 2.0 in 1 + 7 + 16 bits
    S.set_float(exp_bits)
    try:
        plain = S.encode_modular(ints, bits, False, 0)
        layered = S.encode_modular_frame(ints, S.frame(is_last=0, save_as_reference=1), bits) + \
            S.encode_modular_frame(ints[:32, :40], S.frame(emit=1, have_crop=1, crop_x0=10, crop_y0=20, canvas_w=w, canvas_h=h, blend_mode=1, blend_source=1), bits)
    finally:
        S.set_float(0)
    for data in (plain, layered):
        for dtype in (np.float32, np.uint8, np.uint16):
            check_against_oracle(jx, data, dtype, 3)
    if bits == 24:
        with pytest.raises(jx.UnsupportedBitWidth):          # decode.rs:394-401: float samples are 16 or 32 bits wide for the inferred type
            jx.decoder_builder().decode(plain)
        return
    meta, px = jx.decoder_builder().decode(plain)
    assert px.dtype == (np.float16 if bits == 16 else np.float32)
    if bits == 16:
        assert np.array_equal(np.asarray(px).view(np.uint16).reshape(h, w, 3), ints.astype(np.uint16))       # the file's halves, bit for bit


@pytest.mark.gpu
def test_spot_colour_channels(jx):
    """JxlDecoderSetRenderSpotcolors (decode.rs:350-362) / stage_spot.cc: a spot-colour extra channel is mixed into the colour channels —
    colour = mix * spot + (1 - mix) * colour, mix = solidity * channel — in linear light when an XYB frame goes straight to the output, in
    the output space otherwise (after blending); render_spotcolors(false) hands out the plain image.  Lossless RGB + spot (the mixing
    formula checked with numpy), XYB VarDCT + spot with sRGB and PQ output, and a two-frame image with blending."""
    rng = np.random.default_rng(4)
    h, w = 72, 104
    img = rng.integers(0, 256, (h, w, 4)).astype(np.int32)
    spot = (1.0, 0.25, 0.125, 0.75)
    S.set_spot(spot)
    try:
        lossless = S.encode_modular(img, 8, False, 0)
        lossy = S.encode_vardct(S.synthetic_image(3, w, h), seed=5, alpha=img[..., 3].astype(np.uint8))
        S.set_color(white_point=1, primaries=9, tf=16, intensity_target=1000.0)
        lossy_pq = S.encode_vardct(S.synthetic_image(3, w, h), seed=5, alpha=img[..., 3].astype(np.uint8))
        S.set_color()
        layered = S.encode_vardct_frame(S.synthetic_image(3, w, h), S.frame(is_last=0, save_as_reference=1), seed=3, alpha=img[..., 3].astype(np.uint8)) + \
            S.encode_vardct_frame(S.synthetic_image(9, 40, 32), S.frame(emit=1, have_crop=1, crop_x0=10, crop_y0=20, canvas_w=w, canvas_h=h, blend_mode=1, blend_source=1), seed=4,
                                  alpha=img[:32, :40, 3].astype(np.uint8))
    finally:
        S.set_spot()
        S.set_color()
    for data in (lossless, lossy, lossy_pq, layered):
        for dtype in (np.float32, np.uint8, np.uint16):
            check_against_oracle(jx, data, dtype, 3)
    px = jx.decoder_builder(pixel_format=jx.PixelFormat(num_channels=3)).decode_with(lossless, np.float32)[1].reshape(h, w, 3)
    rgb, s = img[..., :3].astype(np.float32) / np.float32(255), img[..., 3].astype(np.float32) / np.float32(255)
    mix = np.float32(spot[3]) * s
    want = mix[..., None] * np.array(spot[:3], np.float32) + (np.float32(1) - mix)[..., None] * rgb
    assert np.abs(px - want).max() < 1e-6
    plain = jx.decoder_builder(pixel_format=jx.PixelFormat(num_channels=3), render_spotcolors=False).decode_with(lossless, np.uint8)[1].reshape(h, w, 3)
    assert np.array_equal(plain, img[..., :3].astype(np.uint8))
    O.set_render_spotcolors(False)
    try:
        for data in (lossy, layered):
            ref = O.decode(data).pixels("u8", 3)
            got = jx.decoder_builder(pixel_format=jx.PixelFormat(num_channels=3), render_spotcolors=False).decode_with(data, np.uint8)[1]
            assert np.array_equal(np.asarray(got).ravel(), np.frombuffer(ref, np.uint8))
    finally:
        O.set_render_spotcolors(True)


@pytest.mark.gpu
def test_float_alpha_channel(jx):
    """Float extra channels (BitDepth.float_sample of an ExtraChannelInfo): a binary16 RGBA image, colour and alpha both bit patterns of
    halves; they take the frame tail (IntToFloatSample per plane).  Exact halves back in f32, the usual clamping for integer outputs."""
    rng = np.random.default_rng(16)
    h, w = 50, 70
    vals = np.concatenate([np.arange(0, 0x3C01), np.arange(0x8000, 0xBC01)]).astype(np.uint16)
    ints = rng.choice(vals, (h, w, 4)).astype(np.int32)
    S.set_float(5)
    try:
        data = S.encode_modular(ints, 16, False, 0)
    finally:
        S.set_float(0)
    for dtype in (np.float32, np.uint8, np.uint16):
        check_against_oracle(jx, data, dtype, 4)
    px = jx.decoder_builder(pixel_format=jx.PixelFormat(num_channels=4)).decode_with(data, np.float32)[1].reshape(h, w, 4)
    assert np.array_equal(px.view(np.uint32), ints.astype(np.uint16).view(np.float16).astype(np.float32).view(np.uint32))


def test_modular_group_streams_with_more_than_64k_of_lds_tables(jx):
    """Round 4: the request for ModularGroupFastKernel's dynamic-LDS ceiling was refused by the runtime (static arrays + request > 160 KB) and the refusal ignored, so launches
    that needed more than the default 64 KB would have been refused in turn.  Multi-group Modular streams with deep trees and wide alphabets (16-bit samples) plan more
    than that; they must decode like the oracle — and a refused launch now fails the decode by name (Batch::RunPart CheckLaunches)."""
    seen_big = 0
    for seed, depth, flags in [(31, 9, S.TREE_ALL_PREDICTORS | S.TREE_MULTIPLIERS), (32, 10, S.TREE_WP | S.TREE_PREV_CHANNELS), (33, 10, 31)]:
        data = S.encode_modular_free(seed=seed, w=700, h=560, nchan=3, bits=16, tree_flags=flags, tree_depth=depth)
        b = jx.BatchDecoder(0)
        b.add(data, "uint16", 3)
        b.prepare(); b.decode(); b.finish()
        seen_big += b.info_value("mod_group_lds_bytes") > 65536
        assert np.array_equal(b.output(0).view(np.uint16).reshape(-1), O.decode(data).pixels("u16", 3).view(np.uint16)), seed
    assert seen_big >= 1


def test_animation_with_more_than_256_frames(jx):
    """300 small frames (a long GIF turned into JPEG XL): the last canvas equals the oracle's; the frame count was capped at 256 until round 4"""
    n, w, h = 300, 48, 40
    base = S.synthetic_image(3, w, h)
    S.set_animation(100, 1, 0)
    try:
        parts = [S.encode_vardct_frame(base, S.frame(is_last=0, save_as_reference=1, duration=1), seed=3, strategy_mix=0)]
        for k in range(1, n):
            tile = S.synthetic_image(100 + k % 7, 16, 16)
            parts.append(S.encode_vardct_frame(tile, S.frame(emit=1, is_last=1 if k == n - 1 else 0, have_crop=1, crop_x0=(k * 5) % (w - 16), crop_y0=(k * 3) % (h - 16), canvas_w=w, canvas_h=h,
                                                             blend_mode=0, blend_source=1, save_as_reference=0 if k == n - 1 else 1, duration=1), seed=k, strategy_mix=0))
        data = b"".join(parts)
    finally:
        S.set_animation(0)
    b = jx.BatchDecoder(0)
    b.add(data, "uint8", 3)
    b.prepare(); b.decode(); b.finish()
    assert np.array_equal(b.output(0), O.decode(data).pixels("u8", 3))


def test_long_animation_is_decoded_once(jx):
    """120 shown frames, coalesced, through the raw event loop: every canvas arrives (checked against the oracle at four places) and the whole animation costs one decode —
    JXL_DEC_FULL_IMAGE after JXL_DEC_FULL_IMAGE comes out of the canvases kept in device memory (round 3: every frame replayed all frames before it).  A caller that changes
    its buffer format half way gets the frame-by-frame path for the rest."""
    import time
    L = jx.libjxl()
    n, w, h = 120, 96, 80
    base = S.synthetic_image(3, w, h)

    def frames(upto):
        parts = [S.encode_vardct_frame(base, S.frame(is_last=1 if upto == 0 else 0, save_as_reference=0 if upto == 0 else 1, duration=2), seed=3, strategy_mix=0)]
        for k in range(1, upto + 1):
            tile = S.synthetic_image(100 + k % 5, 24, 16)
            last = k == upto
            parts.append(S.encode_vardct_frame(tile, S.frame(emit=1, is_last=1 if last else 0, have_crop=1, crop_x0=(k * 7) % (w - 24), crop_y0=(k * 5) % (h - 16), canvas_w=w, canvas_h=h,
                                                             blend_mode=1 if k % 3 == 0 else 0, blend_source=1, save_as_reference=0 if last else 1, duration=1 + k % 3), seed=k, strategy_mix=0))
        return b"".join(parts)
    S.set_animation(100, 1, 0)
    try:
        full = frames(n - 1)
        prefixes = {k: frames(k) for k in (0, 1, 57)}
    finally:
        S.set_animation(0)

    def run(switch_at=None):
        data = np.frombuffer(full, np.uint8)
        dec = L.JxlDecoderCreate(None)
        assert L.JxlDecoderSubscribeEvents(dec, jx.JXL_DEC_FRAME | jx.JXL_DEC_FULL_IMAGE) == 0
        assert L.JxlDecoderSetInput(dec, data.ctypes.data, len(data)) == 0
        L.JxlDecoderCloseInput(dec)
        got, buf = [], None
        while True:
            st = L.JxlDecoderProcessInput(dec)
            if st == jx.JXL_DEC_NEED_IMAGE_OUT_BUFFER:
                nc = 4 if switch_at is not None and len(got) >= switch_at else 3
                fmt = jx.JxlPixelFormat(nc, jx.JXL_TYPE_UINT8, jx.JXL_NATIVE_ENDIAN, 0)
                buf = np.zeros(w * h * nc, np.uint8)
                assert L.JxlDecoderSetImageOutBuffer(dec, C.byref(fmt), buf.ctypes.data, buf.size) == 0
            elif st == jx.JXL_DEC_FULL_IMAGE:
                got.append(buf); buf = None
            elif st == jx.JXL_DEC_SUCCESS:
                break
            elif st != jx.JXL_DEC_FRAME:
                raise AssertionError((st, jx.last_error()))
        L.JxlDecoderDestroy(dec)
        return got
    run()                                   # (warm-up: first use of the kernels of the frame tail)
    t0 = time.perf_counter()
    got = run()
    once = time.perf_counter() - t0
    assert len(got) == n
    for k, stream in prefixes.items():
        assert np.array_equal(got[k], O.decode(stream).pixels("u8", 3)), k
    assert np.array_equal(got[n - 1], O.decode(full).pixels("u8", 3))
    t0 = time.perf_counter()
    mixed = run(switch_at=100)              # RGBA from frame 100 on: those 20 frames replay their predecessors
    per_frame_tail = time.perf_counter() - t0
    assert len(mixed) == n and np.array_equal(mixed[57], got[57])
    assert np.array_equal(mixed[n - 1].reshape(h, w, 4)[..., :3].reshape(-1), got[n - 1])
    assert once < per_frame_tail, (once, per_frame_tail)      # 120 frames out of one decode take less than 100 + 20 replayed ones
