"""Progressive output through the libjxl ABI (-m gpu): JXL_DEC_FRAME_PROGRESSION at the kDC / kLastPasses / kPasses steps, JxlDecoderFlushImage and
JxlDecoderGetIntendedDownsamplingRatio (jpegxl-sys/src/decode.rs:243, :1482, :1495, :1513).
A frame's LF image and HF metadata are decodable before any AC group: the flush at the kDC step shows the frame with every AC coefficient zero through the regular IDCT /
restoration / colour stages; the later steps add the first passes of every group — each compared bit for bit with the oracle's render of that step —, on complete streams
(events, flushes, then the full image) and on streams cut off inside their AC groups (JXL_DEC_NEED_MORE_INPUT, flush — the groups that have arrived in full, the others from
their LF part —, more input, full image)."""
import ctypes as C

import numpy as np
import pytest

from conftest import fixture_bytes
import oracle_lib as O
import synth_lib as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def jx(built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import jpegxl_rs_amd as jx
    return jx


def _run(jx, data, cut=None, detail=None, events=None):
    """Drives the raw ABI; returns (event list, {label: pixel array}).  cut: bytes handed in first (then the whole stream)."""
    L = jx.libjxl()
    raw = np.frombuffer(data, np.uint8)
    dec = L.JxlDecoderCreate(None)
    ev = jx.JXL_DEC_BASIC_INFO | jx.JXL_DEC_FULL_IMAGE | jx.JXL_DEC_FRAME_PROGRESSION if events is None else events
    assert L.JxlDecoderSubscribeEvents(dec, ev) == 0
    if detail is not None:
        assert L.JxlDecoderSetProgressiveDetail(dec, detail) == 0
    first = raw if cut is None else raw[:cut].copy()
    assert L.JxlDecoderSetInput(dec, first.ctypes.data, len(first)) == 0
    if cut is None:
        L.JxlDecoderCloseInput(dec)
    fmt = jx.JxlPixelFormat(3, jx.JXL_TYPE_UINT8, jx.JXL_NATIVE_ENDIAN, 0)
    seen, shots, px = [], {}, None
    for _ in range(32):
        st = L.JxlDecoderProcessInput(dec)
        seen.append(st)
        if st == jx.JXL_DEC_BASIC_INFO:
            info = jx.JxlBasicInfo()
            assert L.JxlDecoderGetBasicInfo(dec, C.byref(info)) == 0
        elif st == jx.JXL_DEC_NEED_IMAGE_OUT_BUFFER:
            size = C.c_size_t()
            assert L.JxlDecoderImageOutBufferSize(dec, C.byref(fmt), C.byref(size)) == 0
            px = np.zeros(size.value, np.uint8)
            assert L.JxlDecoderSetImageOutBuffer(dec, C.byref(fmt), px.ctypes.data, px.size) == 0
        elif st == jx.JXL_DEC_FRAME_PROGRESSION:
            assert L.JxlDecoderFlushImage(dec) == 0, jx.last_error()
            shots.setdefault("progression", px.copy())             # the first step (kDC)
            shots.setdefault("steps", []).append((int(L.JxlDecoderGetIntendedDownsamplingRatio(dec)), px.copy()))
        elif st == jx.JXL_DEC_NEED_MORE_INPUT:
            if px is not None:
                rc = L.JxlDecoderFlushImage(dec)
                shots["flush_rc"] = rc
                if rc == 0:
                    shots["truncated"] = px.copy()
            if "resupplied" in shots:
                break
            assert L.JxlDecoderReleaseInput(dec) == len(first)       # nothing counts as consumed: the stream comes again from its start
            assert L.JxlDecoderSetInput(dec, raw.ctypes.data, len(raw)) == 0
            L.JxlDecoderCloseInput(dec)
            shots["resupplied"] = True
        elif st == jx.JXL_DEC_FULL_IMAGE:
            shots["full"] = px.copy()
        elif st in (jx.JXL_DEC_SUCCESS, jx.JXL_DEC_ERROR):
            break
    L.JxlDecoderDestroy(dec)
    return seen, shots


def test_frame_progression_and_flush_on_a_complete_stream(jx):
    img = S.synthetic_image(21, 1030, 520)
    data = S.encode_vardct(img, seed=21, strategy_mix=1, epf_iters=1, gab=1)
    seen, shots = _run(jx, data)
    assert seen[:3] == [jx.JXL_DEC_BASIC_INFO, jx.JXL_DEC_NEED_IMAGE_OUT_BUFFER, jx.JXL_DEC_FRAME_PROGRESSION] and seen[-2:] == [jx.JXL_DEC_FULL_IMAGE, jx.JXL_DEC_SUCCESS]
    dc = O.decode(data, dc_only=True).pixels("u8", 3)
    full = O.decode(data).pixels("u8", 3)
    assert np.array_equal(shots["progression"], dc) and np.array_equal(shots["full"], full)
    assert not np.array_equal(dc, full)
    err = shots["progression"].astype(np.float64) - img.reshape(-1)
    assert 10 * np.log10(255.0 ** 2 / (err ** 2).mean()) > 20.0          # the LF image is a recognisable preview of the picture
    # kFrames: no progression event; not subscribed: none either
    seen, _ = _run(jx, data, detail=0)
    assert jx.JXL_DEC_FRAME_PROGRESSION not in seen and seen[-1] == jx.JXL_DEC_SUCCESS
    seen, _ = _run(jx, data, events=jx.JXL_DEC_BASIC_INFO | jx.JXL_DEC_FULL_IMAGE)
    assert jx.JXL_DEC_FRAME_PROGRESSION not in seen


@pytest.mark.parametrize("frac", [0.5, 0.9])
def test_flush_on_a_truncated_stream_then_more_input(jx, frac):
    img = S.synthetic_image(22, 2100, 600)
    data = S.encode_vardct(img, seed=22, strategy_mix=2, epf_iters=2, gab=1)
    cut = int(len(data) * frac)
    seen, shots = _run(jx, data, cut=cut)
    assert seen[:3] == [jx.JXL_DEC_BASIC_INFO, jx.JXL_DEC_NEED_IMAGE_OUT_BUFFER, jx.JXL_DEC_NEED_MORE_INPUT]
    assert shots["flush_rc"] == 0
    # every group whose stream is completely there is drawn in full, the others from their LF part
    part = O.decode(data[:cut], allow_truncated=True).pixels("u8", 3)
    assert np.array_equal(shots["truncated"], part)
    dc, full = O.decode(data, dc_only=True).pixels("u8", 3), O.decode(data).pixels("u8", 3)
    assert not np.array_equal(part, dc) and not np.array_equal(part, full)
    assert seen[-2:] == [jx.JXL_DEC_FULL_IMAGE, jx.JXL_DEC_SUCCESS] and np.array_equal(shots["full"], O.decode(data).pixels("u8", 3))


def test_no_flush_without_the_lf_part_or_for_other_frame_kinds(jx):
    L = jx.libjxl()
    img = S.synthetic_image(23, 1030, 520)
    data = S.encode_vardct(img, seed=23, strategy_mix=1)
    # cut inside the LF part: not even the headers' frame index can be used
    seen, shots = _run(jx, data, cut=len(data) // 50)
    assert seen[0] == jx.JXL_DEC_NEED_MORE_INPUT and "truncated" not in shots
    # a Modular image and a one-group frame: no progression event, and a flush right after the buffer was set is refused
    for other in (fixture_bytes("sample.jxl"), S.encode_vardct(S.synthetic_image(3, 200, 136), seed=3)):
        seen, shots = _run(jx, other)
        assert jx.JXL_DEC_FRAME_PROGRESSION not in seen and seen[-1] == jx.JXL_DEC_SUCCESS
    raw = np.frombuffer(fixture_bytes("sample.jxl"), np.uint8)
    dec = L.JxlDecoderCreate(None)
    assert L.JxlDecoderFlushImage(dec) == 1                       # nothing in progress
    assert L.JxlDecoderSubscribeEvents(dec, jx.JXL_DEC_FULL_IMAGE) == 0 and L.JxlDecoderSetInput(dec, raw.ctypes.data, len(raw)) == 0
    assert L.JxlDecoderProcessInput(dec) == jx.JXL_DEC_NEED_IMAGE_OUT_BUFFER
    fmt = jx.JxlPixelFormat(4, jx.JXL_TYPE_UINT8, jx.JXL_NATIVE_ENDIAN, 0)
    px = np.zeros(40 * 50 * 4, np.uint8)
    assert L.JxlDecoderSetImageOutBuffer(dec, C.byref(fmt), px.ctypes.data, px.size) == 0
    assert L.JxlDecoderFlushImage(dec) == 1 and "no flush was done" in jx.last_error()
    assert L.JxlDecoderProcessInput(dec) == jx.JXL_DEC_FULL_IMAGE
    L.JxlDecoderDestroy(dec)


@pytest.mark.parametrize("num_passes,pass_ds", [(3, 1), (3, 0), (2, 1)])
def test_pass_steps_of_a_progressive_frame(jx, num_passes, pass_ds):
    """kPasses: a step after every pass but the last; kLastPasses: after the passes the frame header names as the last ones of a downsampling ratio (none without such
    entries); kDC: the LF step alone.  Every flush equals the oracle's render with that many passes of every group; the ratio is the frame header's."""
    img = S.synthetic_image(31, 1030, 520)
    data = S.encode_vardct(img, seed=31, strategy_mix=2, epf_iters=1, gab=1, num_passes=num_passes, pass_ds=pass_ds)
    full = O.decode(data).pixels("u8", 3)
    renders = [O.decode(data, dc_only=True).pixels("u8", 3)] + [O.decode(data, max_passes=k).pixels("u8", 3) for k in range(1, num_passes)]
    for a, b in zip(renders, renders[1:] + [full]):
        assert not np.array_equal(a, b)                                          # every step adds something
    ratios = [8] + ([4, 2][-(num_passes - 1):] if pass_ds else [8] * (num_passes - 1))
    for detail, want in ((3, list(range(num_passes))), (2, list(range(num_passes)) if pass_ds else [0]), (1, [0])):
        seen, shots = _run(jx, data, detail=detail)
        assert seen.count(jx.JXL_DEC_FRAME_PROGRESSION) == len(want) and seen[-2:] == [jx.JXL_DEC_FULL_IMAGE, jx.JXL_DEC_SUCCESS]
        for (ratio, px), k in zip(shots["steps"], want):
            assert np.array_equal(px, renders[k]), (detail, k)
            assert ratio == ratios[k], (detail, k, ratio)
        assert np.array_equal(shots["full"], full)


@pytest.mark.parametrize("frac", [0.45, 0.7, 0.95])
def test_flush_on_a_truncated_progressive_stream(jx, frac):
    """Three passes, cut inside one of them: groups show as many passes as have arrived in full."""
    img = S.synthetic_image(32, 1500, 700)
    data = S.encode_vardct(img, seed=32, strategy_mix=1, epf_iters=1, gab=1, num_passes=3)
    cut = int(len(data) * frac)
    seen, shots = _run(jx, data, cut=cut)
    assert jx.JXL_DEC_NEED_MORE_INPUT in seen and shots["flush_rc"] == 0
    assert np.array_equal(shots["truncated"], O.decode(data[:cut], allow_truncated=True).pixels("u8", 3))
    assert np.array_equal(shots["full"], O.decode(data).pixels("u8", 3))
