"""GPU tests (-m gpu) of the decode-side C-ABI entry points beyond what jpegxl-rs itself calls (jpegxl-sys/src/decode.rs:1200-1532): the box API, the
multithreaded image-out callback, extra-channel buffers, JxlDecoderSetImageOutBitDepth / SetProgressiveDetail / FlushImage, the frame header after
JXL_DEC_FULL_IMAGE, and the honest failure of a decode that would need libjxl's tone mapping.  All through ctypes on lib/libjxl.so, pixels against
the oracle."""
import ctypes as C
import struct

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


def box(kind, payload):
    return struct.pack(">I", 8 + len(payload)) + kind + payload


def brotli_compress(data):
    L = C.CDLL("libbrotlienc.so.1")
    L.BrotliEncoderCompress.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_char_p, C.POINTER(C.c_size_t), C.c_char_p]
    out = C.create_string_buffer(len(data) + 1024)
    n = C.c_size_t(len(out))
    assert L.BrotliEncoderCompress(5, 22, 0, len(data), data, C.byref(n), out) == 1
    return out.raw[:n.value]


def container(codestream, split=False):
    """ISO BMFF container around a codestream: signature, ftyp, an Exif box, the codestream (one jxlc box or two jxlp boxes), a Brotli-compressed xml box"""
    exif = b"\0\0\0\0II*\0" + bytes(range(40))
    xml = b"<x:xmpmeta>" + b"jpeg xl " * 300 + b"</x:xmpmeta>"
    out = box(b"JXL ", b"\r\n\x87\n") + box(b"ftyp", b"jxl \0\0\0\0jxl ") + box(b"Exif", exif)
    if split:
        h = len(codestream) // 3
        out += box(b"jxlp", struct.pack(">I", 0) + codestream[:h]) + box(b"jxlp", struct.pack(">I", 0x80000001) + codestream[h:])
    else:
        out += box(b"jxlc", codestream)
    out += box(b"brob", b"xml " + brotli_compress(xml))
    return out, exif, xml


@pytest.mark.parametrize("split,decompress,chunk", [(False, True, 1 << 16), (True, True, 100), (False, False, 64)])
def test_box_api_walks_the_container(jx, split, decompress, chunk):
    L = jx.libjxl()
    img = S.synthetic_image(3, 96, 64)
    cs = S.encode_vardct(img, seed=4)
    data, exif, xml = container(cs, split)
    raw = np.frombuffer(data, np.uint8)
    dec = L.JxlDecoderCreate(None)
    assert L.JxlDecoderSubscribeEvents(dec, jx.JXL_DEC_BASIC_INFO | jx.JXL_DEC_FULL_IMAGE | jx.JXL_DEC_BOX) == 0
    assert L.JxlDecoderSetDecompressBoxes(dec, 1 if decompress else 0) == 0
    assert L.JxlDecoderSetInput(dec, raw.ctypes.data, len(raw)) == 0
    L.JxlDecoderCloseInput(dec)
    fmt = jx.JxlPixelFormat(3, jx.JXL_TYPE_UINT8, jx.JXL_NATIVE_ENDIAN, 0)
    events, types, contents, cur, bufs = [], [], {}, None, []
    px = None

    def collect():
        nonlocal cur, bufs
        if cur is not None and bufs:
            unused = L.JxlDecoderReleaseBoxBuffer(dec)
            got = b"".join(bytes(b) for b in bufs)
            contents[cur] = got[:len(got) - unused]
        cur, bufs = None, []
    while True:
        st = L.JxlDecoderProcessInput(dec)
        events.append(st)
        if st == jx.JXL_DEC_BOX:
            collect()
            t = C.create_string_buffer(4)
            assert L.JxlDecoderGetBoxType(dec, t, 1) == 0
            rawt = C.create_string_buffer(4)
            assert L.JxlDecoderGetBoxType(dec, rawt, 0) == 0
            size = C.c_uint64()
            assert L.JxlDecoderGetBoxSizeRaw(dec, C.byref(size)) == 0
            types.append((rawt.raw, t.raw, size.value))
            if t.raw in (b"Exif", b"xml "):
                cur = t.raw
                bufs = [np.zeros(chunk, np.uint8)]
                assert L.JxlDecoderSetBoxBuffer(dec, bufs[-1].ctypes.data, chunk) == 0
                assert L.JxlDecoderSetBoxBuffer(dec, bufs[-1].ctypes.data, chunk) == 1       # already set
        elif st == jx.JXL_DEC_BOX_NEED_MORE_OUTPUT:
            assert L.JxlDecoderReleaseBoxBuffer(dec) == 0
            bufs.append(np.zeros(chunk, np.uint8))
            assert L.JxlDecoderSetBoxBuffer(dec, bufs[-1].ctypes.data, chunk) == 0
        elif st == jx.JXL_DEC_BASIC_INFO:
            info = jx.JxlBasicInfo()
            assert L.JxlDecoderGetBasicInfo(dec, C.byref(info)) == 0 and info.have_container == 1
        elif st == jx.JXL_DEC_NEED_IMAGE_OUT_BUFFER:
            px = np.zeros(96 * 64 * 3, np.uint8)
            assert L.JxlDecoderSetImageOutBuffer(dec, C.byref(fmt), px.ctypes.data, px.size) == 0
        elif st == jx.JXL_DEC_FULL_IMAGE:
            pass
        elif st == jx.JXL_DEC_SUCCESS:
            collect()
            break
        else:
            raise AssertionError((st, jx.last_error()))
    L.JxlDecoderDestroy(dec)
    want_types = [b"JXL ", b"ftyp", b"Exif"] + ([b"jxlp", b"jxlp"] if split else [b"jxlc"]) + [b"brob"]
    assert [t[0] for t in types] == want_types
    assert types[-1][1] == b"xml " and types[2][2] == 8 + len(exif)
    # boxes in front of the codestream come before the basic info, the ones behind it after the image
    assert events.index(jx.JXL_DEC_BASIC_INFO) > events.index(jx.JXL_DEC_BOX) and events[-2] != jx.JXL_DEC_FULL_IMAGE
    assert contents[b"Exif"] == exif
    assert contents[b"xml "] == (xml if decompress else b"xml " + brotli_compress(xml))
    assert np.array_equal(px, O.decode(cs).pixels("u8", 3))


def test_box_api_without_a_container(jx):
    L = jx.libjxl()
    cs = np.frombuffer(S.encode_vardct(S.synthetic_image(3, 64, 48), seed=4), np.uint8)
    dec = L.JxlDecoderCreate(None)
    assert L.JxlDecoderSubscribeEvents(dec, jx.JXL_DEC_BASIC_INFO | jx.JXL_DEC_BOX) == 0
    assert L.JxlDecoderSetInput(dec, cs.ctypes.data, len(cs)) == 0
    L.JxlDecoderCloseInput(dec)
    assert L.JxlDecoderProcessInput(dec) == jx.JXL_DEC_BASIC_INFO
    t = C.create_string_buffer(4)
    assert L.JxlDecoderGetBoxType(dec, t, 0) == 1
    assert L.JxlDecoderProcessInput(dec) == jx.JXL_DEC_SUCCESS
    L.JxlDecoderDestroy(dec)


def _decode_loop(jx, L, dec, on_need_buffer, on_full=None):
    while True:
        st = L.JxlDecoderProcessInput(dec)
        if st == jx.JXL_DEC_NEED_IMAGE_OUT_BUFFER:
            on_need_buffer()
        elif st == jx.JXL_DEC_FULL_IMAGE:
            if on_full:
                on_full()
        elif st == jx.JXL_DEC_SUCCESS:
            return
        elif st in (jx.JXL_DEC_BASIC_INFO, jx.JXL_DEC_FRAME):
            continue
        else:
            raise AssertionError((st, jx.last_error()))


def test_multithreaded_image_out_callback(jx):
    L = jx.libjxl()
    img = S.synthetic_image(8, 200, 136)
    data = np.frombuffer(S.encode_vardct(img, seed=2, strategy_mix=2), np.uint8)
    INIT = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t)
    RUN = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p)
    DESTROY = C.CFUNCTYPE(None, C.c_void_p)
    L.JxlDecoderSetMultithreadedImageOutCallback.argtypes = [C.c_void_p, C.c_void_p, INIT, RUN, DESTROY, C.c_void_p]
    rows, calls = {}, []

    def init(opaque, nthreads, npx):
        calls.append(("init", nthreads, npx)); return 1234

    def run(opaque, thread, x, y, n, px):
        assert opaque == 1234 and thread == 0 and x == 0
        rows[y] = bytes(C.cast(px, C.POINTER(C.c_uint8 * (n * 3))).contents)

    def destroy(opaque):
        calls.append(("destroy", opaque))
    cbs = (INIT(init), RUN(run), DESTROY(destroy))
    fmt = jx.JxlPixelFormat(3, jx.JXL_TYPE_UINT8, jx.JXL_NATIVE_ENDIAN, 0)
    dec = L.JxlDecoderCreate(None)
    assert L.JxlDecoderSubscribeEvents(dec, jx.JXL_DEC_FULL_IMAGE) == 0
    assert L.JxlDecoderSetInput(dec, data.ctypes.data, len(data)) == 0
    L.JxlDecoderCloseInput(dec)
    _decode_loop(jx, L, dec, lambda: L.JxlDecoderSetMultithreadedImageOutCallback(dec, C.byref(fmt), *cbs, None) == 0 or pytest.fail(jx.last_error()))
    L.JxlDecoderDestroy(dec)
    assert calls == [("init", 1, 200), ("destroy", 1234)]
    got = np.frombuffer(b"".join(rows[y] for y in range(136)), np.uint8)
    assert np.array_equal(got, O.decode(bytes(data)).pixels("u8", 3))


@pytest.mark.parametrize("dtype,jt", [(np.uint8, 2), (np.uint16, 3), (np.float32, 0)])
def test_alpha_as_an_extra_channel_buffer(jx, dtype, jt):
    L = jx.libjxl()
    w, h = 150, 90
    img = S.synthetic_image(8, w, h)
    al = (np.arange(w * h, dtype=np.uint32).reshape(h, w) * 7 % 256).astype(np.uint8)
    data = np.frombuffer(S.encode_vardct(img, seed=2, alpha=al), np.uint8)
    fmt = jx.JxlPixelFormat(3, jx.JXL_TYPE_UINT8, jx.JXL_NATIVE_ENDIAN, 0)
    efmt = jx.JxlPixelFormat(1, jt, jx.JXL_NATIVE_ENDIAN, 0)
    dec = L.JxlDecoderCreate(None)
    assert L.JxlDecoderSubscribeEvents(dec, jx.JXL_DEC_BASIC_INFO | jx.JXL_DEC_FULL_IMAGE) == 0
    assert L.JxlDecoderSetInput(dec, data.ctypes.data, len(data)) == 0
    L.JxlDecoderCloseInput(dec)
    px, plane = np.zeros(w * h * 3, np.uint8), np.zeros(w * h, dtype)

    def need():
        size = C.c_size_t()
        assert L.JxlDecoderExtraChannelBufferSize(dec, C.byref(efmt), C.byref(size), 0) == 0 and size.value == plane.nbytes
        assert L.JxlDecoderExtraChannelBufferSize(dec, C.byref(efmt), C.byref(size), 1) == 1          # no such channel
        assert L.JxlDecoderSetImageOutBuffer(dec, C.byref(fmt), px.ctypes.data, px.size) == 0
        assert L.JxlDecoderSetExtraChannelBuffer(dec, C.byref(efmt), plane.ctypes.data, plane.nbytes - 1, 0) == 1
        assert L.JxlDecoderSetExtraChannelBuffer(dec, C.byref(efmt), plane.ctypes.data, plane.nbytes, 0) == 0
    _decode_loop(jx, L, dec, need)
    L.JxlDecoderDestroy(dec)
    ref = O.decode(bytes(data))
    assert np.array_equal(px, ref.pixels("u8", 3))
    kind = {np.uint8: "u8", np.uint16: "u16", np.float32: "f32"}[dtype]
    want = ref.pixels(kind, 4).view(dtype).reshape(h, w, 4)[..., 3]
    assert np.array_equal(plane.reshape(h, w), want)


def test_bit_depth_progressive_detail_and_flush(jx):
    L = jx.libjxl()
    data = np.frombuffer(S.encode_vardct(S.synthetic_image(8, 64, 48), seed=2), np.uint8)
    dec = L.JxlDecoderCreate(None)
    for detail, want in [(0, 0), (1, 0), (2, 0), (3, 0), (4, 1), (6, 1), (-1, 1)]:
        assert L.JxlDecoderSetProgressiveDetail(dec, detail) == want
    assert L.JxlDecoderSubscribeEvents(dec, jx.JXL_DEC_FULL_IMAGE | jx.JXL_DEC_FRAME_PROGRESSION) == 0
    assert L.JxlDecoderSetInput(dec, data.ctypes.data, len(data)) == 0
    L.JxlDecoderCloseInput(dec)
    assert L.JxlDecoderSetImageOutBitDepth(dec, C.byref(jx.JxlBitDepth(0, 0, 0))) == 1          # no buffer yet
    assert L.JxlDecoderFlushImage(dec) == 1
    fmt = jx.JxlPixelFormat(3, jx.JXL_TYPE_UINT16, jx.JXL_NATIVE_ENDIAN, 0)
    px = np.zeros(64 * 48 * 3, np.uint16)

    def need():
        assert L.JxlDecoderSetImageOutBuffer(dec, C.byref(fmt), px.ctypes.data, px.nbytes) == 0
        assert L.JxlDecoderSetImageOutBitDepth(dec, C.byref(jx.JxlBitDepth(0, 0, 0))) == 0      # from the pixel format
        assert L.JxlDecoderSetImageOutBitDepth(dec, C.byref(jx.JxlBitDepth(2, 16, 0))) == 0     # custom = the type's range
        assert L.JxlDecoderSetImageOutBitDepth(dec, C.byref(jx.JxlBitDepth(2, 17, 0))) == 1     # does not fit
        assert L.JxlDecoderSetImageOutBitDepth(dec, C.byref(jx.JxlBitDepth(2, 8, 3))) == 1      # a float depth for an integer buffer
        assert L.JxlDecoderSetImageOutBitDepth(dec, C.byref(jx.JxlBitDepth(0, 0, 0))) == 0
        assert L.JxlDecoderFlushImage(dec) == 1                                                 # nothing partial to flush
    _decode_loop(jx, L, dec, need)
    L.JxlDecoderDestroy(dec)
    assert np.array_equal(px, O.decode(bytes(data)).pixels("u16", 3).view(np.uint16))


def test_frame_header_stays_current_after_full_image(jx):
    """libjxl keeps the frame's header valid until the next JXL_DEC_FRAME (advisor finding, round 3)"""
    L = jx.libjxl()
    a, b = S.synthetic_image(6, 120, 80), S.synthetic_image(7, 64, 48)
    stream = (S.encode_vardct_frame(a, S.frame(is_last=0, save_as_reference=1, duration=0), seed=3)
              + S.encode_vardct_frame(b, S.frame(emit=1, have_crop=1, crop_x0=10, crop_y0=20, canvas_w=120, canvas_h=80, blend_mode=1, blend_source=1), seed=4))
    data = np.frombuffer(stream, np.uint8)
    dec = L.JxlDecoderCreate(None)
    assert L.JxlDecoderSetCoalescing(dec, 0) == 0
    assert L.JxlDecoderSubscribeEvents(dec, jx.JXL_DEC_FRAME | jx.JXL_DEC_FULL_IMAGE) == 0
    assert L.JxlDecoderSetInput(dec, data.ctypes.data, len(data)) == 0
    L.JxlDecoderCloseInput(dec)
    fmt = jx.JxlPixelFormat(3, jx.JXL_TYPE_UINT8, jx.JXL_NATIVE_ENDIAN, 0)
    seen, keep = [], []

    def need():
        size = C.c_size_t()
        assert L.JxlDecoderImageOutBufferSize(dec, C.byref(fmt), C.byref(size)) == 0
        keep.append(np.zeros(size.value, np.uint8))
        assert L.JxlDecoderSetImageOutBuffer(dec, C.byref(fmt), keep[-1].ctypes.data, size.value) == 0

    def full():
        hdr = jx.JxlFrameHeader()
        assert L.JxlDecoderGetFrameHeader(dec, C.byref(hdr)) == 0
        seen.append((hdr.layer_info.xsize, hdr.layer_info.ysize, hdr.is_last))
    _decode_loop(jx, L, dec, need, full)
    L.JxlDecoderDestroy(dec)
    assert seen == [(120, 80, 0), (64, 48, 1)]


def test_tone_mapping_request_fails_loudly_instead_of_being_ignored(jx):
    """decode.rs:360-362: a desired intensity target below a PQ image's own asks for libjxl's Rec. 2408 tone mapper; it is not built here, and the decode says so"""
    lin = np.clip(S.synthetic_image(9, 96, 64).astype(np.float32) / 255.0, 0, 1) ** 2.2
    S.set_color(1, 1, 16, intensity_target=4000.0)
    try:
        data = S.encode_vardct(lin, seed=3, out_bits=32, hdr=1)
    finally:
        S.set_color()
    meta, px = jx.decoder_builder().decode_with(data, np.float32)                       # no target: PQ pixels as coded
    assert px.size == 96 * 64 * 3
    meta, px = jx.decoder_builder(desired_intensity_target=10000.0).decode_with(data, np.float32)   # brighter display: nothing to map
    with pytest.raises(jx.DecodeError, match="tone mapping"):
        jx.decoder_builder(desired_intensity_target=255.0).decode_with(data, np.float32)
    srgb = S.encode_vardct(S.synthetic_image(9, 96, 64), seed=3)
    jx.decoder_builder(desired_intensity_target=100.0).decode_with(srgb, np.uint8)      # an SDR image never passes through the stage


@pytest.mark.parametrize("bits,mode", [(10, 1), (12, 1), (10, 2), (5, 2)])
def test_image_out_bit_depth_from_the_codestream_gives_back_the_coded_integers(jx, bits, mode):
    """decode.rs:1528 JxlDecoderSetImageOutBitDepth: a lossless 10 / 12-bit image decoded into a 16-bit buffer with the depth taken from the codestream (type 1) or given
    (type 2) comes out as the integers that were coded (libjxl: sample = round(v x (2^bits - 1))); with the default the same buffer holds them scaled to 65535.
    A custom depth of 5 bits on an 8-bit buffer: round(v x 31)."""
    L = jx.libjxl()
    rng = np.random.default_rng(bits)
    w, h = 90, 70
    src_bits = 8 if bits == 5 else bits
    img = rng.integers(0, 1 << src_bits, (h, w, 3)).astype(np.int32)
    data = np.frombuffer(S.encode_modular(img, src_bits, False, 0), np.uint8)
    u16 = bits != 5
    fmt = jx.JxlPixelFormat(3, jx.JXL_TYPE_UINT16 if u16 else jx.JXL_TYPE_UINT8, jx.JXL_NATIVE_ENDIAN, 0)
    out = {}
    for name, depth in (("set", jx.JxlBitDepth(mode, bits if mode == 2 else 0, 0)), ("default", None)):
        dec = L.JxlDecoderCreate(None)
        assert L.JxlDecoderSubscribeEvents(dec, jx.JXL_DEC_FULL_IMAGE) == 0
        assert L.JxlDecoderSetInput(dec, data.ctypes.data, len(data)) == 0
        L.JxlDecoderCloseInput(dec)
        px = np.zeros(w * h * 3, np.uint16 if u16 else np.uint8)

        def need():
            assert L.JxlDecoderSetImageOutBuffer(dec, C.byref(fmt), px.ctypes.data, px.nbytes) == 0
            if depth is not None:
                assert L.JxlDecoderSetImageOutBitDepth(dec, C.byref(depth)) == 0, jx.last_error()
        _decode_loop(jx, L, dec, need)
        L.JxlDecoderDestroy(dec)
        out[name] = px.reshape(h, w, 3).astype(np.int64)
    full = 65535 if u16 else 255
    v = img / float((1 << src_bits) - 1)
    assert np.abs(out["default"] - np.rint(v * full).astype(np.int64)).max() <= 1      # (float32 sample x 65535 lands within one code of the float64 value)
    if bits == 5:
        assert np.array_equal(out["set"], np.rint(v.astype(np.float32) * np.float32(31)).astype(np.int64))
    else:
        assert np.array_equal(out["set"], img)


@pytest.mark.parametrize("lossy", [False, True])
def test_a_spot_colour_channel_as_an_extra_channel_buffer(jx, lossy):
    """an extra channel that is not alpha (a spot colour) handed out as a plane of its own: the coded samples, exactly (the extra channels of a VarDCT frame are lossless too);
    the colour buffer of the same decode carries the image with the spot colour rendered, as without the request"""
    L = jx.libjxl()
    rng = np.random.default_rng(5)
    w, h = 104, 72
    img = rng.integers(0, 256, (h, w, 4)).astype(np.int32)
    S.set_spot((1.0, 0.25, 0.125, 0.75))
    try:
        stream = S.encode_vardct(S.synthetic_image(3, w, h), seed=5, alpha=img[..., 3].astype(np.uint8)) if lossy else S.encode_modular(img, 8, False, 0)
    finally:
        S.set_spot()
    data = np.frombuffer(stream, np.uint8)
    fmt = jx.JxlPixelFormat(3, jx.JXL_TYPE_UINT8, jx.JXL_NATIVE_ENDIAN, 0)
    efmt = jx.JxlPixelFormat(1, jx.JXL_TYPE_UINT8, jx.JXL_NATIVE_ENDIAN, 0)
    dec = L.JxlDecoderCreate(None)
    assert L.JxlDecoderSubscribeEvents(dec, jx.JXL_DEC_FULL_IMAGE) == 0
    assert L.JxlDecoderSetInput(dec, data.ctypes.data, len(data)) == 0
    L.JxlDecoderCloseInput(dec)
    px, plane = np.zeros(w * h * 3, np.uint8), np.zeros(w * h, np.uint8)

    def need():
        assert L.JxlDecoderSetImageOutBuffer(dec, C.byref(fmt), px.ctypes.data, px.size) == 0
        assert L.JxlDecoderSetExtraChannelBuffer(dec, C.byref(efmt), plane.ctypes.data, plane.size, 0) == 0, jx.last_error()
    _decode_loop(jx, L, dec, need)
    L.JxlDecoderDestroy(dec)
    assert np.array_equal(plane.reshape(h, w), img[..., 3].astype(np.uint8))
    assert np.array_equal(px, O.decode(stream).pixels("u8", 3))


def test_box_api_survives_damaged_containers(jx):
    """bit flips and truncations anywhere in a container (box sizes, types, the brob payload, the codestream): the box walk ends in JXL_DEC_SUCCESS or a clean error,
    never writes past a box buffer, and the decoder works afterwards"""
    L = jx.libjxl()
    rng = np.random.default_rng(77)
    cs = S.encode_vardct(S.synthetic_image(3, 64, 48), seed=4)
    outcomes = {"ok": 0, "error": 0}
    for split in (False, True):
        data, exif, xml = container(cs, split)
        for trial in range(150):
            bad = bytearray(data)
            hi = len(bad) if trial % 2 else 120                   # half of the trials aim at the box headers in front
            for pos in rng.integers(12, hi, 1 + trial % 3):
                bad[pos] ^= 1 << int(rng.integers(0, 8))
            if trial % 9 == 8:
                bad = bad[: int(rng.integers(40, len(bad)))]
            raw = np.frombuffer(bytes(bad), np.uint8)
            dec = L.JxlDecoderCreate(None)
            assert L.JxlDecoderSubscribeEvents(dec, jx.JXL_DEC_FULL_IMAGE | jx.JXL_DEC_BOX | jx.JXL_DEC_BOX_COMPLETE) == 0
            L.JxlDecoderSetDecompressBoxes(dec, 1)
            assert L.JxlDecoderSetInput(dec, raw.ctypes.data, len(raw)) == 0
            L.JxlDecoderCloseInput(dec)
            guard = np.full(64 + 16, 0xA5, np.uint8)               # 64 bytes handed out, 16 bytes of canary behind them
            px = np.zeros(64 * 48 * 3, np.uint8)
            fmt = jx.JxlPixelFormat(3, jx.JXL_TYPE_UINT8, jx.JXL_NATIVE_ENDIAN, 0)
            for _ in range(400):
                st = L.JxlDecoderProcessInput(dec)
                if st == jx.JXL_DEC_BOX:
                    L.JxlDecoderReleaseBoxBuffer(dec)
                    assert L.JxlDecoderSetBoxBuffer(dec, guard.ctypes.data, 64) == 0
                elif st == jx.JXL_DEC_BOX_NEED_MORE_OUTPUT:
                    L.JxlDecoderReleaseBoxBuffer(dec)
                    assert L.JxlDecoderSetBoxBuffer(dec, guard.ctypes.data, 64) == 0
                elif st == jx.JXL_DEC_NEED_IMAGE_OUT_BUFFER:
                    if L.JxlDecoderSetImageOutBuffer(dec, C.byref(fmt), px.ctypes.data, px.size) != 0:
                        outcomes["error"] += 1
                        break
                elif st == jx.JXL_DEC_SUCCESS:
                    outcomes["ok"] += 1
                    break
                elif st in (jx.JXL_DEC_ERROR, jx.JXL_DEC_NEED_MORE_INPUT):
                    outcomes["error"] += 1
                    break
            else:
                raise AssertionError("the event loop does not end")
            assert (guard[64:] == 0xA5).all()
            L.JxlDecoderDestroy(dec)
    assert outcomes["ok"] > 20 and outcomes["error"] > 20, outcomes
    _, px = jx.decoder_builder().decode_with(cs, np.uint8)
    assert np.array_equal(px.reshape(-1), O.decode(cs).pixels("u8", 3))
