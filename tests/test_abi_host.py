"""CPU tests of the drop-in boundary: the look-alike libjxl.so / libjxl_threads.so load and export every symbol
jpegxl-sys declares, struct layouts match, and the host-side logic of the jpegxl-rs mirror behaves like the reference
(no compute calls: there is no GPU here, and the product has no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, fixture_bytes


@pytest.fixture(scope="module")
def jx(built):
    import jpegxl_rs_amd as jx
    return jx


def test_every_declared_symbol_is_exported(jx):
    L = jx.libjxl()
    T = jx.libjxl_threads()
    header = open(os.path.join(ROOT, "include", "jxl_hip.h")).read()
    declared = set(re.findall(r"\b(Jxl\w+)\s*\((?!\*)", header))
    assert len(declared) >= 45
    for name in declared:
        lib = T if re.match(r"Jxl(Thread|Resizable)ParallelRunner", name) else L
        assert hasattr(lib, name), name
    stubs = re.findall(r"^\w[\w\*]*\s+(Jxl\w+)\(void\)", open(os.path.join(ROOT, "jpegxl-rs_amd", "csrc", "jxl_stubs.cc")).read(), re.M)
    assert len(stubs) == 62                      # 120 declared by jpegxl-sys - 58 live ones (round 4: the box API x 6, SetMultithreadedImageOutCallback, ExtraChannelBufferSize / SetExtraChannelBuffer, SetProgressiveDetail, FlushImage, SetImageOutBitDepth); what is left: encoder, CMS, gain map, compressed ICC, output colour profiles
    for name in stubs:
        assert hasattr(L, name), name
    assert not (set(stubs) & declared)


def test_struct_layouts_and_version(jx):
    assert C.sizeof(jx.JxlBasicInfo) == 204          # jpegxl-sys codestream_header.rs:108-241
    assert jx.JxlBasicInfo.orientation.offset == 48 and jx.JxlBasicInfo.intrinsic_xsize.offset == 96 and jx.JxlBasicInfo.padding.offset == 104
    assert C.sizeof(jx.JxlPixelFormat) == 24 and jx.JxlPixelFormat.align.offset == 16
    assert C.sizeof(jx.JxlMemoryManager) == 24
    assert jx.libjxl().JxlDecoderVersion() == 11002  # jpegxl-sys/src/lib.rs:79


def test_signature_check(jx):
    """utils.rs:42-47 and jpegxl-sys/src/lib.rs:98-99 (2 bytes are enough for a bare codestream)."""
    s = fixture_bytes("sample.jxl")
    assert jx.check_valid_signature(b"") is None
    assert jx.check_valid_signature(bytes(64)) is False
    assert jx.check_valid_signature(s) is True
    assert jx.libjxl().JxlSignatureCheck(s, 2) == 2
    assert jx.libjxl().JxlSignatureCheck(fixture_bytes("sample_jpg.jxl"), 12) == 3
    assert jx.libjxl().JxlSignatureCheck(fixture_bytes("sample_jpg.jxl"), 5) == 0


def test_invalid_input_errors(jx):
    """tests/decode.rs:33-42, errors.rs:109-137: [] and zeros are InvalidInput before any library state is touched."""
    dec = jx.decoder_builder()
    for bad in (b"", b"\x00\x00", bytes(64)):
        with pytest.raises(jx.InvalidInput):
            dec.decode(bad)


def test_raw_state_machine_without_gpu(jx):
    """The event order of jpegxl-sys/src/lib.rs:85-171 up to the point where pixels are needed; on a box without a GPU
    the hot path must fail loudly (JXL_DEC_ERROR), never fall back to a CPU decoder."""
    import torch
    L = jx.libjxl()
    dec = L.JxlDecoderCreate(None)
    assert dec
    data = np.frombuffer(fixture_bytes("sample.jxl"), np.uint8)
    assert L.JxlDecoderSubscribeEvents(dec, jx.JXL_DEC_BASIC_INFO | jx.JXL_DEC_FULL_IMAGE) == 0
    assert L.JxlDecoderSubscribeEvents(dec, 1) == 1                       # non-event bits are rejected
    assert L.JxlDecoderProcessInput(dec) == jx.JXL_DEC_NEED_MORE_INPUT    # no input yet
    L.JxlDecoderReset(dec)
    assert L.JxlDecoderSubscribeEvents(dec, jx.JXL_DEC_BASIC_INFO | jx.JXL_DEC_FULL_IMAGE) == 0
    assert L.JxlDecoderSetInput(dec, data.ctypes.data, len(data)) == 0
    assert L.JxlDecoderSetInput(dec, data.ctypes.data, len(data)) == 1    # input already set (decode.rs:680)
    st = L.JxlDecoderProcessInput(dec)
    if torch.cuda.is_available():
        assert st == jx.JXL_DEC_BASIC_INFO
    else:
        assert st == jx.JXL_DEC_ERROR and "no CPU fallback" in jx.last_error()
    L.JxlDecoderDestroy(dec)


def test_memory_manager_is_copied_and_used(jx):
    """memory.rs:24-39: alloc/free go through the caller's manager; the struct itself is a temporary (copied)."""
    L = jx.libjxl()
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]
    libc.free.argtypes = [C.c_void_p]
    calls = {"alloc": 0, "free": 0}
    ALLOC = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
    FREE = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)

    def alloc(opaque, size):
        calls["alloc"] += 1
        return libc.malloc(size)

    def free(opaque, ptr):
        calls["free"] += 1
        libc.free(ptr)
    a, f = ALLOC(alloc), FREE(free)
    mm = jx.JxlMemoryManager(None, C.cast(a, C.c_void_p), C.cast(f, C.c_void_p))
    dec = L.JxlDecoderCreate(C.byref(mm))
    mm.alloc = None; mm.free = None     # the caller's struct may die right after Create
    assert dec and calls["alloc"] == 1
    L.JxlDecoderDestroy(dec)
    assert calls["free"] == 1
    half = jx.JxlMemoryManager(None, C.cast(a, C.c_void_p), None)
    assert not L.JxlDecoderCreate(C.byref(half))   # alloc without free is rejected
    null_alloc = ALLOC(lambda o, s: None)
    oom = jx.JxlMemoryManager(None, C.cast(null_alloc, C.c_void_p), C.cast(f, C.c_void_p))
    assert not L.JxlDecoderCreate(C.byref(oom))    # -> DecodeError::CannotCreateDecoder (decode.rs:184-186)


def test_threads_runner_contract(jx):
    """parallel_runner.rs:55-122: init once on the calling thread, func for every i in [start, end), return codes."""
    T = jx.libjxl_threads()
    INIT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t)
    FUNC = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_size_t)
    T.JxlThreadParallelRunner.restype = C.c_int
    T.JxlThreadParallelRunner.argtypes = [C.c_void_p, C.c_void_p, INIT, FUNC, C.c_uint32, C.c_uint32]
    T.JxlResizableParallelRunner.restype = C.c_int
    T.JxlResizableParallelRunner.argtypes = [C.c_void_p, C.c_void_p, INIT, FUNC, C.c_uint32, C.c_uint32]
    seen, inits, tids = [], [], set()
    init = INIT(lambda o, n: inits.append(n) or 0)
    func = FUNC(lambda o, i, t: (seen.append(i), tids.add(t)) and None)
    r = T.JxlThreadParallelRunnerCreate(None, 4)
    assert T.JxlThreadParallelRunner(r, None, init, func, 5, 105) == 0
    assert sorted(seen) == list(range(5, 105)) and inits == [4] and tids <= {0, 1, 2, 3}
    assert T.JxlThreadParallelRunner(r, None, init, func, 7, 7) == 0 and inits == [4]      # empty range: no init
    assert T.JxlThreadParallelRunner(r, None, INIT(lambda o, n: 42), func, 0, 3) == 42       # init failure propagates
    assert T.JxlThreadParallelRunner(r, None, init, func, 3, 2) == -1
    T.JxlThreadParallelRunnerDestroy(r)
    seen.clear()
    rr = T.JxlResizableParallelRunnerCreate(None)
    assert T.JxlResizableParallelRunner(rr, None, init, func, 0, 10) == 0 and sorted(seen) == list(range(10))   # 0 workers: inline
    T.JxlResizableParallelRunnerSetThreads(rr, T.JxlResizableParallelRunnerSuggestThreads(3840, 2160))
    seen.clear()
    assert T.JxlResizableParallelRunner(rr, None, init, func, 0, 1000) == 0 and sorted(seen) == list(range(1000))
    T.JxlResizableParallelRunnerDestroy(rr)
    assert T.JxlThreadParallelRunnerDefaultNumWorkerThreads() >= 1


def test_runner_objects_are_small(jx):
    """threads_runner.rs:97-102 / resizable_runner.rs:96-102: the runner object itself costs < 1 KiB from the custom allocator."""
    T = jx.libjxl_threads()
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]; libc.free.argtypes = [C.c_void_p]
    total = []
    ALLOC = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
    FREE = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
    a = ALLOC(lambda o, s: total.append(s) or libc.malloc(s))
    f = FREE(lambda o, p: libc.free(p))
    mm = jx.JxlMemoryManager(None, C.cast(a, C.c_void_p), C.cast(f, C.c_void_p))
    r = T.JxlThreadParallelRunnerCreate(C.byref(mm), 10)
    assert r and sum(total) < 1024
    T.JxlThreadParallelRunnerDestroy(r)


def test_builder_fields_are_public_and_mutable(jx):
    """decode.rs:85-154 + tests/decode.rs:167-173: options can be changed between decodes of one decoder object."""
    dec = jx.decoder_builder(pixel_format=jx.PixelFormat(num_channels=3, endianness=jx.Endianness.Big, align=10), icc_profile=False, init_jpeg_buffer=512)
    assert dec.pixel_format.align == 10 and dec.init_jpeg_buffer == 512
    dec.pixel_format = None
    dec.skip_reorientation = True
    dec.unpremul_alpha = True
    assert dec.pixel_format is None


def test_icc_profile_synthesis(jx):
    """tests/decode.rs:45-67 asks for icc_profile(true) and requires lcms2 to accept the result.  The profile is built from
    the image header alone (host code): parse it with lcms2 (PIL.ImageCms), check the ICC header, the MD5 profile ID, the
    D50-adapted sRGB colorants and that an sRGB -> profile transform is the identity for an sRGB-tagged file."""
    import hashlib, io, struct
    from PIL import Image, ImageCms
    fixtures = os.path.join(ROOT, "tests", "fixtures")
    icc = jx.icc_profile_from_headers(open(os.path.join(fixtures, "bench.jxl"), "rb").read())
    assert struct.unpack(">I", icc[:4])[0] == len(icc) and icc[36:40] == b"acsp" and icc[12:20] == b"mntrRGB " and icc[8:12] == bytes([4, 0x40, 0, 0])
    zeroed = bytearray(icc)
    zeroed[44:48] = bytes(4); zeroed[64:68] = bytes(4); zeroed[84:100] = bytes(16)
    assert hashlib.md5(bytes(zeroed)).digest() == icc[84:100]
    prof = ImageCms.ImageCmsProfile(io.BytesIO(icc))
    assert ImageCms.getProfileDescription(prof).strip() == "RGB_D65_SRG_Rel_SRG"
    want = ((0.4360, 0.2225, 0.0139), (0.3851, 0.7169, 0.0971), (0.1431, 0.0606, 0.7141))  # sRGB colorants adapted to D50
    for got, ref in zip((prof.profile.red_colorant, prof.profile.green_colorant, prof.profile.blue_colorant), want):
        assert np.abs(np.array(got[0]) - np.array(ref)).max() < 6e-4
    ramp = np.arange(256, dtype=np.uint8)
    img = Image.fromarray(np.stack([np.tile(ramp, (4, 1)), np.tile(ramp[::-1], (4, 1)), np.full((4, 256), 77, np.uint8)], -1), "RGB")
    out = ImageCms.profileToProfile(img, ImageCms.createProfile("sRGB"), prof, renderingIntent=1)
    assert np.abs(np.asarray(out).astype(int) - np.asarray(img).astype(int)).max() <= 1
    # gamma-tagged RGBA16 file and a grey file: still valid profiles; the grey one has a kTRC and no colorants
    g = jx.icc_profile_from_headers(open(os.path.join(fixtures, "sample.jxl"), "rb").read())
    assert "g0.45455" in ImageCms.getProfileDescription(ImageCms.ImageCmsProfile(io.BytesIO(g)))
    k = jx.icc_profile_from_headers(open(os.path.join(fixtures, "sample_grey.jxl"), "rb").read())
    assert k[16:20] == b"GRAY" and b"kTRC" in k and b"rXYZ" not in k
    assert ImageCms.ImageCmsProfile(io.BytesIO(k)).profile.xcolor_space.strip() == "GRAY"
    with pytest.raises(jx.GenericError):
        jx.icc_profile_from_headers(b"\xff\x0a")


def test_sharding_plan():
    from jpegxl_rs_amd.sharding import shard_range, shard_sizes
    assert shard_sizes(1024, 8) == [128] * 8            # BASELINE config 3
    assert shard_sizes(10, 4) == [3, 3, 2, 2]
    assert [shard_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_sizes(3, 8) == [1, 1, 1, 0, 0, 0, 0, 0]
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_library_quant_tables_equal_the_oracle(jx):
    """Host-side table parity (no GPU): every library-default dequantisation table (17 kinds x 3 channels, incl. the AFV table
    and the FastPowf band interpolation of quant_weights.cc) the product computes is bit-identical to the oracle's."""
    import ctypes as C
    import numpy as np
    import oracle_lib as O
    L = jx.libjxl()
    L.JxlHipLibraryQuantTable.restype = C.c_size_t
    L.JxlHipLibraryQuantTable.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    OL = O.lib()
    OL.jxlo_library_qtable.restype = C.c_size_t
    OL.jxlo_library_qtable.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    rows = [1, 1, 1, 1, 2, 4, 1, 1, 2, 1, 1, 8, 4, 16, 8, 32, 16]
    cols = [1, 1, 1, 1, 2, 4, 2, 4, 4, 1, 1, 8, 8, 16, 16, 32, 32]
    for kind in range(17):
        n = rows[kind] * cols[kind] * 64
        for c in range(3):
            a = np.zeros(n, np.float32); b = np.zeros(n, np.float32)
            assert L.JxlHipLibraryQuantTable(kind, c, a.ctypes.data, n) == n
            assert OL.jxlo_library_qtable(kind, c, b.ctypes.data, n) == n
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (kind, c)
            assert np.all(a > 0) and np.all(np.isfinite(a))


def test_resizable_runner_survives_resizing_between_jobs(jx):
    """resizable_runner.rs:63 calls SetThreads on every basic-info event, i.e. between jobs: freshly started workers must
    neither replay the previous job nor miss the next one, and every index is visited exactly once."""
    import ctypes as C
    T = jx.libjxl_threads()
    INIT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t)
    FUNC = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_size_t)
    T.JxlResizableParallelRunner.restype = C.c_int
    T.JxlResizableParallelRunner.argtypes = [C.c_void_p, C.c_void_p, INIT, FUNC, C.c_uint32, C.c_uint32]
    pool = T.JxlResizableParallelRunnerCreate(None)
    try:
        for threads, (lo, hi) in ((0, (0, 7)), (3, (5, 300)), (8, (0, 1000)), (2, (10, 11)), (5, (0, 64))):
            T.JxlResizableParallelRunnerSetThreads(pool, threads)
            hits = [0] * hi
            def func(_, i, thread):
                hits[i] += 1
            assert T.JxlResizableParallelRunner(pool, None, INIT(lambda _, n: 0), FUNC(func), lo, hi) == 0
            assert hits[lo:hi] == [1] * (hi - lo) and sum(hits[:lo]) == 0
    finally:
        T.JxlResizableParallelRunnerDestroy(pool)


def test_libjxl_probe_reports_what_it_looked_at(jx, monkeypatch):
    """The run-time libjxl probe (tests/libjxl_probe.py) must never mistake this repository's look-alike for the real
    library, and must say what it probed when nothing is found."""
    import libjxl_probe as P
    monkeypatch.setenv("LD_LIBRARY_PATH", os.path.join(ROOT, "jpegxl-rs_amd", "lib"))   # the look-alike is on the path: still not "real"
    found = P.probe()
    assert found["lib"] is None or os.path.realpath(os.path.dirname(found["lib"])) != os.path.realpath(os.path.join(ROOT, "jpegxl-rs_amd", "lib"))
    text = P.describe(found)
    assert ("no libjxl on this box: probed" in text) != found["available"]
    v, own = P._check_lib(os.path.join(ROOT, "jpegxl-rs_amd", "lib", "libjxl.so"))
    assert v == 11002 and own          # a copy of the look-alike elsewhere on the system would be recognised and rejected


def test_jpeg_writer_reproduces_sample_jpg(jx):
    """Host half of reconstruct() (decode.rs:493-514; reference test tests/decode.rs:123-139), no GPU needed: the jbrd box of
    samples/sample_jpg.jxl plus the JPEG's quantised coefficients (Huffman-decoded independently by tests/golden/make_golden.py)
    must serialise to samples/sample.jpg byte for byte (sha256 d28d532e...)."""
    import ctypes as C
    import hashlib
    import struct
    import numpy as np
    from conftest import FIXTURES, GOLDEN
    data = open(os.path.join(FIXTURES, "sample_jpg.jxl"), "rb").read()
    pos, jbrd = 0, None
    while pos < len(data):
        n, t = struct.unpack(">I4s", data[pos:pos + 8])
        if t == b"jbrd":
            jbrd = data[pos + 8:pos + n]
        pos += n
    assert jbrd is not None and len(jbrd) == 162
    g = np.load(os.path.join(GOLDEN, "sample_jpg_coefficients.npz"))
    coef = np.ascontiguousarray(g["coefficients"].astype(np.int16))          # [3][7][5][64] natural order
    qt = np.ascontiguousarray(g["qtables"].astype(np.int32))
    L = jx.libjxl()
    L.JxlHipDebugWriteJpeg.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
    out = np.zeros(1 << 16, np.uint8)
    n = C.c_size_t(10)
    assert L.JxlHipDebugWriteJpeg(jbrd, len(jbrd), 40, 50, coef.ctypes.data, qt.ctypes.data, out.ctypes.data, C.byref(n)) == 2 and n.value == 1779
    n = C.c_size_t(len(out))
    assert L.JxlHipDebugWriteJpeg(jbrd, len(jbrd), 40, 50, coef.ctypes.data, qt.ctypes.data, out.ctypes.data, C.byref(n)) == 0, jx.last_error()
    want = open(os.path.join(FIXTURES, "sample.jpg"), "rb").read()
    assert out[:n.value].tobytes() == want
    assert hashlib.sha256(want).hexdigest().startswith("d28d532e")


def test_icc_profile_for_pq_and_hlg_images(jx):
    """decode.rs:368-385 (icc_profile(true)) on HDR images: PQ / HLG have no parametric ICC curve — the profile carries a sampled one (checked
    against SMPTE ST 2084 / ARIB STD-B67 in float64) and the H.273 code points in a `cicp` tag; lcms2 accepts it."""
    import io, struct
    import synth_lib as S
    from PIL import ImageCms
    img = S.synthetic_image(3, 64, 48)
    for tf, prim_enum, cicp_prim in ((16, 9, 9), (18, 9, 9), (16, 1, 1), (18, 11, 12)):
        S.set_color(1, prim_enum, tf, intensity_target=1000.0)
        try:
            data = S.encode_vardct(img, seed=3)
        finally:
            S.set_color()
        icc = jx.icc_profile_from_headers(data)
        assert struct.unpack(">I", icc[:4])[0] == len(icc)
        prof = ImageCms.ImageCmsProfile(io.BytesIO(icc))                 # lcms2 parses it
        assert ("PeQ" if tf == 16 else "HLG") in ImageCms.getProfileDescription(prof)
        ntags = struct.unpack(">I", icc[128:132])[0]
        tags = {icc[132 + 12 * i:136 + 12 * i]: struct.unpack(">II", icc[136 + 12 * i:144 + 12 * i]) for i in range(ntags)}
        off, size = tags[b"cicp"]
        assert icc[off:off + 4] == b"cicp" and tuple(icc[off + 8:off + 12]) == (cicp_prim, tf, 0, 1) and size == 12
        off, size = tags[b"rTRC"]
        assert tags[b"gTRC"] == tags[b"rTRC"] == tags[b"bTRC"] and icc[off:off + 4] == b"curv"
        n = struct.unpack(">I", icc[off + 8:off + 12])[0]
        curve = np.frombuffer(icc[off + 12:off + 12 + 2 * n], ">u2").astype(np.float64) / 65535
        e = np.linspace(0, 1, n)
        if tf == 16:
            m1, m2, c1, c2, c3 = 2610 / 16384, 2523 / 4096 * 128, 3424 / 4096, 2413 / 4096 * 32, 2392 / 4096 * 32
            p = e ** (1 / m2)
            want = (np.maximum(p - c1, 0) / (c2 - c3 * p)) ** (1 / m1)
        else:
            a = 0.17883277; b = 1 - 4 * a; c = 0.5 - a * np.log(4 * a)
            want = np.where(e <= 0.5, e * e / 3, (np.exp((e - c) / a) + b) / 12)
        assert np.abs(curve - want).max() < 1e-5 and curve[0] == 0 and abs(curve[-1] - 1) < 1e-4 and np.all(np.diff(curve) >= 0)


def test_embedded_icc_profile_round_trip(jx):
    """SURVEY §8f.3 "decode the ANS-coded embedded ICC": image headers with want_icc carry the profile as an entropy-coded,
    predicted byte stream (icc_codec.cc).  tools/jxl_synth.cc writes one (header differences, tag commands incl. the TRC / XYZ
    triples and implicit offsets, insert / shuffle / predict content commands of every width and order); the product's host
    parser and the oracle — written separately — must both give back the exact profile.  Profiles: lcms2's sRGB / Lab / XYZ
    (PIL.ImageCms) and this library's own synthesised one.  (libjxl itself is not available: the stream syntax is [R].)"""
    import numpy as np
    import oracle_lib as O
    import synth_lib as S
    from PIL import ImageCms
    img = np.random.default_rng(1).integers(0, 255, (40, 50, 3)).astype(np.int32)
    plain = S.encode_modular(img, 8, False, 0)
    profiles = {n: ImageCms.ImageCmsProfile(ImageCms.createProfile(n)).tobytes() for n in ("sRGB", "LAB", "XYZ")}
    profiles["own"] = jx.icc_profile_from_headers(plain)
    want_px = O.decode(plain).pixels("u8", 3)
    for name, icc in profiles.items():
        S.set_icc(icc)
        try:
            data = S.encode_modular(img, 8, False, 0)
            xyb = S.encode_vardct(S.synthetic_image(3, 64, 48), seed=1)
        finally:
            S.set_icc(b"")
        assert len(data) > len(plain) + 100
        assert jx.icc_profile_from_headers(data) == icc, name
        assert jx.icc_profile_from_headers(xyb) == icc, name
        ref = O.decode(data)
        assert ref.icc() == icc, name
        assert np.array_equal(ref.pixels("u8", 3), want_px)          # the samples are untouched by the profile
        assert O.decode(xyb).icc() == icc
    # damaged stream: an error, not a crash / wrong profile
    S.set_icc(profiles["sRGB"])
    try:
        data = bytearray(S.encode_modular(img, 8, False, 0))
    finally:
        S.set_icc(b"")
    bad = 0
    for pos in range(20, 200, 7):
        d2 = bytearray(data); d2[pos] ^= 0x5A
        try:
            got = jx.icc_profile_from_headers(bytes(d2))
            bad += got != profiles["sRGB"]
        except jx.DecodeError:
            bad += 1
    assert bad >= 20


def _describe(jx, data):
    import ctypes as C
    L = jx.libjxl()
    L.JxlHipDebugDescribe.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    buf = C.create_string_buffer(1 << 16)
    if L.JxlHipDebugDescribe(data, len(data), buf, len(buf)):
        raise jx.GenericError(jx.last_error())
    lines = buf.value.decode().strip().split("\n")
    rows = []
    for l in lines:
        if l.startswith("  quantizer"):                      # detail line of the VarDCT frame above it
            rows[-1]["quantizer"] = dict(t.split("=") for t in l.split() if "=" in t)
            continue
        if l.startswith(("  lf_code", "  ac_code")):          # entropy-code sizes of the VarDCT frame above it
            rows[-1].setdefault("codes", []).append(dict([("which", l.split()[0])] + [t.split("=") for t in l.split() if "=" in t]))
            continue
        kv = dict(t.split("=") for t in l.split() if "=" in t)
        kv["kind"] = l.split()[0] + (" " + l.split()[2] if l.startswith("frame") else "")
        kv["raw"] = l
        rows.append(kv)
    return rows


def test_host_parser_on_fixtures_and_synthetic_streams(jx):
    """The host half of the decode (container, image header, frame headers, TOC, LfGlobal incl. patches / splines / noise, local
    Modular streams, embedded ICC) runs without a GPU through JxlHipDebugDescribe: what it finds in the reference's fixtures and
    in the synthesiser's streams is checked here, so that parser regressions show up in the CPU suite."""
    import numpy as np
    import synth_lib as S
    from conftest import fixture_bytes
    from free_cases import FREE_CASES
    d = _describe(jx, fixture_bytes("sample_grey.jxl"))
    assert d[0]["frames"] == "2" and d[0]["gray"] == "1" and d[0]["xyb"] == "1"
    assert d[1]["kind"] == "frame modular" and d[1]["type"] == "2" and d[2]["kind"] == "frame vardct" and d[2]["patches"] == "1"
    d = _describe(jx, fixture_bytes("2bit.jxl"))
    assert d[0]["bits"] == "2" and d[1]["splines"] == "28" and d[1]["prefix"] == "1" and d[1]["kind"] == "frame modular"
    d = _describe(jx, fixture_bytes("bench.jxl"))
    assert d[1]["groups"] == "54" and d[1]["tree_nodes"] == "6643" and d[1]["wp"] == "1" and d[1]["sections"] == "58"
    d = _describe(jx, fixture_bytes("sample_jpg.jxl"))
    assert d[1]["kind"] == "frame vardct" and d[0]["xyb"] == "0"
    for name, kw in FREE_CASES.items():
        kw = dict(kw); kw.setdefault("bits", 16)
        d = _describe(jx, S.encode_modular_free(**kw))
        f = d[1]
        groups = int(f["groups"])
        if kw.get("local_trees") == 2:
            assert int(f["local_streams"]) == (1 + groups if groups > 1 else 1) and f["tree_nodes"] == "0", name
        elif kw.get("local_trees") == 1:
            assert int(f["local_streams"]) == (groups if groups > 1 else 0), name
        else:
            assert f["local_streams"] == "0", name
        assert (f["lz77"] == "1") == bool(kw.get("lz77")) or kw.get("local_trees") == 2, name
        if name.startswith("previous_channel_properties"):
            assert int(f["max_prop"]) >= 16, name
        assert int(f["transforms"]) == (1 if kw.get("palette") else 0), name
    # multi-frame stream with crop + blending, noise
    img = S.synthetic_image(5, 200, 136)
    small = S.synthetic_image(9, 64, 48)
    two = S.encode_vardct_frame(img, S.frame(is_last=0, save_as_reference=1), seed=3) + \
        S.encode_vardct_frame(small, S.frame(emit=1, have_crop=1, crop_x0=40, crop_y0=30, canvas_w=200, canvas_h=136, blend_mode=2, blend_source=1, noise_lut=[100] * 8), seed=4)
    d = _describe(jx, two)
    assert d[0]["frames"] == "2" and "at (40,30)" in d[2]["raw"] and d[2]["blend"] == "2" and d[2]["noise"] == "1" and d[2]["last"] == "1"
    # the JPEG transcode of the reference: JPEG-style quantiser set-up (what tools/synth_ycbcr.h's transcodes copy)
    q = _describe(jx, fixture_bytes("sample_jpg.jxl"))[1]["quantizer"]
    assert (q["global_scale"], q["quant_lf"], q["ycbcr"], q["sampling"], q["cfl_base"]) == ("65536", "1", "1", "0,0,0", "0,0")
    # rejected inputs say why
    with pytest.raises(jx.GenericError, match="truncated|unsupported|corrupt|signature|header"):
        _describe(jx, fixture_bytes("bench.jxl")[:1000])


def test_host_parser_survives_corrupted_input(jx):
    """Memory safety of the host half: a few thousand single- and multi-bit corruptions and truncations of every kind of stream
    (fixtures incl. the jbrd container, multi-frame / feature streams, free-running Modular streams with local trees and LZ77,
    an embedded ICC profile) must end in a description or a clean rejection."""
    import numpy as np
    import synth_lib as S
    from conftest import fixture_bytes
    from free_cases import FREE_CASES
    from PIL import ImageCms
    rng = np.random.default_rng(99)
    streams = [fixture_bytes(n) for n in ("sample.jxl", "sample_grey.jxl", "2bit.jxl", "sample_jpg.jxl")]
    streams += [S.encode_modular_free(**dict(FREE_CASES[n], bits=16)) for n in ("lz77_local_trees", "local_tree_single_group", "palette_delta_gradient", "previous_channel_properties_global")]
    S.set_icc(ImageCms.ImageCmsProfile(ImageCms.createProfile("sRGB")).tobytes())
    try:
        streams.append(S.encode_vardct(S.synthetic_image(3, 64, 48), seed=1))
    finally:
        S.set_icc(b"")
    streams.append(S.encode_vardct(S.synthetic_image(4, 300, 280), seed=2, strategy_mix=2, num_passes=3, permute_toc=3))
    from test_synth_roundtrip import preview_streams, lf_frame_streams, multipass_modular_streams       # round 3: preview frame, LF frames, Modular passes
    streams += [preview_streams()[1][1], lf_frame_streams()[0][1], lf_frame_streams()[1][1], multipass_modular_streams()[0][1]]
    accepted = rejected = 0
    for data in streams:
        for trial in range(300):
            bad = bytearray(data)
            hi = len(bad) if trial % 3 else min(len(bad), 300)
            for pos in rng.integers(2, hi, 1 + trial % 4):
                bad[pos] ^= 1 << int(rng.integers(0, 8))
            if trial % 10 == 9:
                bad = bad[: int(rng.integers(8, len(bad)))]
            try:
                _describe(jx, bytes(bad))
                accepted += 1
            except jx.GenericError:
                rejected += 1
    assert accepted > 100 and rejected > 100


def test_xcd_contiguous_tile_mapping_is_a_bijection():
    """kernels.hip XcdContiguous (workgroup id -> tile of FusedGabEpf1OutKernel): every XCD (id mod 8) gets a contiguous run of tiles and
    every tile is visited exactly once, for any tile count — restated here in Python, the kernel's outputs are checked by the GPU tests."""
    def xcd_contiguous(bid, nwg):
        q, r, c = nwg >> 3, nwg & 7, bid & 7
        return (c * (q + 1) if c < r else r * (q + 1) + (c - r) * q) + (bid >> 3)
    for nwg in list(range(1, 200)) + [10800, 10801, 10807, 43200, 65535]:
        tiles = [xcd_contiguous(b, nwg) for b in range(nwg)]
        assert sorted(tiles) == list(range(nwg)), nwg
        for c in range(min(8, nwg)):
            run = [tiles[b] for b in range(c, nwg, 8)]
            assert run == list(range(run[0], run[0] + len(run))), (nwg, c)
