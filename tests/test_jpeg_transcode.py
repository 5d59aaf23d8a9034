"""JPEG transcodes anchored on real JPEG files (`-m "not gpu"` half).  Pillow's libjpeg writes baseline JPEGs (4:4:4 / 4:2:2 / 4:2:0,
optimised Huffman tables, restart intervals, COM markers); tests/jpeg_tools.py Huffman-decodes them and writes the JPEG XL file of the
lossless transcode (jbrd box + VarDCT codestream with RAW quantisation tables and chroma subsampling, tools/synth_ycbcr.h).  Checked here:
 * the serialisation half of reconstruct() (csrc/jpeg_recon.cc, no GPU needed) reproduces every file byte for byte from its jbrd box and
   coefficients — interleaved MCUs with sampling factors, restart markers, padding bits;
 * a file re-serialised with one scan per component (the other MCU geometry of the writer) decodes in libjpeg to the same pixels;
 * the oracle's pixels for the transcode agree with libjpeg's decode of the JPEG within integer-IDCT / rounding distance — an anchor for
   the oracle's YCbCr, RAW-table, subsampled-grid and chroma-upsampling arithmetic that owes nothing to this repository's encoders.
The GPU half (reconstruct() == the JPEG's bytes, pixels == oracle) is tests/test_gpu_parity.py::test_jpeg_transcodes_of_real_jpegs."""
import ctypes as C
import io

import numpy as np
import pytest

import jpeg_cases as JC
import jpeg_tools as J
import oracle_lib as O


@pytest.fixture(scope="module")
def jx():
    import jpegxl_rs_amd as jx
    return jx


def _write(jx, j, jbrd, may_fail=False):
    L = jx.libjxl()
    L.JxlHipDebugWriteJpegSampled.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
    samp = np.array([[c["h"], c["v"]] for c in j.components], np.uint32)
    coef = np.concatenate([c.reshape(-1) for c in j.coef]).astype(np.int16)
    qt = np.array([j.qt[c["tq"]] for c in j.components], np.int32)
    out = np.zeros(coef.size * 4 + (1 << 16), np.uint8)
    n = C.c_size_t(len(out))
    rc = L.JxlHipDebugWriteJpegSampled(jbrd, len(jbrd), j.width, j.height, samp.ctypes.data, coef.ctypes.data, qt.ctypes.data, out.ctypes.data, C.byref(n))
    if rc != 0 and may_fail:
        return None
    assert rc == 0, jx.last_error()
    return out[:n.value].tobytes()


@pytest.mark.parametrize("case", JC.CASES, ids=lambda c: "%dx%d_ss%d_q%d" % c[:4])
def test_writer_reproduces_libjpeg_files(jx, case):
    data = JC.jpeg_bytes(case)
    j = J.parse_jpeg(data)
    assert _write(jx, j, J.build_jbrd(j)) == data
    if case[4].get("optimize"):
        return                                               # (tables optimised for the interleaved scan lack codes other DC differences need)
    # one scan per component: blocks of the component's own grid, only those that hold image data (T.81 A.2.4)
    first = j.marker_order.index(0xDA)
    j.marker_order[first:first + 1] = [0xDA] * 3
    comps = j.scans[0]["comps"]
    j.scans = [dict(comps=[c]) for c in comps]
    j.padding_bits = []                                      # (ones; the three scans pad on their own)
    split = _write(jx, j, J.build_jbrd(j))
    assert split != data and split.count(b"\xff\xda") >= 3
    assert np.array_equal(JC.pil_pixels(split), JC.pil_pixels(data))


@pytest.mark.parametrize("case", JC.PROGRESSIVE, ids=lambda c: "%dx%d_ss%d_q%d" % c[:4])
def test_writer_reproduces_progressive_libjpeg_files(jx, case):
    """Progressive JPEGs (SOF2; dec_jpeg_data_writer.cc EncodeDCTBlockProgressive / EncodeRefinementBits / DCTCodingState): DC and AC
    first passes at reduced precision, refinement passes, end-of-band runs with buffered correction bits, restart markers inside
    runs.  The parser decodes the ten scans of libjpeg's script into the same coefficients as the baseline file of the same image;
    the writer gives the progressive file back byte for byte — for "gratings" only thanks to the reset points jbrd carries."""
    data = JC.jpeg_bytes(case)
    j = J.parse_jpeg(data)
    assert j.sof == 0xC2 and len(j.scans) == 10
    base = J.parse_jpeg(JC.jpeg_bytes(case[:4] + ({k: v for k, v in case[4].items() if k not in ("progressive", "optimize", "restart_marker_blocks")},)))
    assert all(np.array_equal(a, b) for a, b in zip(j.coef, base.coef))
    assert _write(jx, j, J.build_jbrd(j)) == data
    if case[4].get("image") == "gratings":
        assert sum(len(s["reset_points"]) for s in j.scans) > 10
        j.scans = [dict(s, reset_points=[]) for s in j.scans]
        assert _write(jx, j, J.build_jbrd(j), may_fail=True) != data      # (longer runs: other bytes, or a run length libjpeg's optimised table has no code for)


@pytest.mark.parametrize("case", JC.CASES, ids=lambda c: "%dx%d_ss%d_q%d" % c[:4])
def test_oracle_pixels_of_transcode_match_libjpeg(case):
    data = JC.jpeg_bytes(case)
    w, h = case[:2]
    px = np.frombuffer(O.decode(J.transcode(data)).pixels("u8", 3), np.uint8).reshape(h, w, 3).astype(int)
    ref = JC.pil_pixels(data)
    d = np.abs(px - ref)
    assert d.max() <= 6 and d.mean() < 0.7, (int(d.max()), float(d.mean()))      # float IDCT + one rounding vs libjpeg's integer pipeline


def test_metadata_markers_from_boxes_host_half(jx):
    """cjxl's layout for ICC / Exif / XMP (profile in the image header, `Exif` / `xml ` boxes, plain or `brob`): the host half of
    reconstruct() — jbrd parse + marker rebuild — accepts matching boxes and names the reason when they do not match."""
    import struct
    from PIL import Image, ImageCms
    icc = ImageCms.ImageCmsProfile(ImageCms.createProfile("sRGB")).tobytes() + bytes(range(256)) * 300
    ex = Image.Exif()
    ex[0x010E] = "a test image"
    buf = io.BytesIO()
    Image.fromarray(JC.photo(40, 50)).save(buf, "JPEG", quality=85, subsampling=2, icc_profile=icc, exif=ex.tobytes(), xmp=b"<x:xmpmeta xmlns:x='adobe:ns:meta/'/>")
    data = buf.getvalue()

    def describe(jxl):
        L = jx.libjxl()
        L.JxlHipDebugDescribe.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        out = C.create_string_buffer(1 << 16)
        assert L.JxlHipDebugDescribe(jxl, len(jxl), out, len(out)) == 0, jx.last_error()
        return out.value.decode()
    for kw in (dict(), dict(compress_boxes=True), dict(jbrd_last=True)):
        d = describe(J.transcode(data, typed_metadata=True, **kw))
        assert "jpeg_reconstruction=1" in d and "icc=%d" % len(icc) in d, d
    assert "jpeg_reconstruction=1" in describe(J.transcode(data))          # everything inside jbrd
    jxl = J.transcode(data, typed_metadata=True)
    pos, boxes = 12, []
    while pos < len(jxl):
        n, t = struct.unpack(">I4s", jxl[pos:pos + 8])
        boxes.append((t, jxl[pos:pos + n]))
        pos += n
    assert "Exif marker without an Exif box" in describe(jxl[:12] + b"".join(b for t, b in boxes if t != b"Exif"))
    assert "XMP size mismatch" in describe(jxl[:12] + b"".join(struct.pack(">I4s", len(b) + 1, t) + b[8:] + b" " if t == b"xml " else b for t, b in boxes))


@pytest.mark.parametrize("kw", [dict(), dict(progressive=True), dict(restart_marker_rows=1)], ids=["baseline", "progressive", "restarts"])
def test_grey_jpegs(jx, kw):
    """One-component JPEGs: a grey image header over a YCbCr frame whose chroma channels are empty; the writer takes component 0 from
    the Y channel.  Bytes back from the host writer; the oracle's grey pixels against libjpeg's."""
    from PIL import Image
    data = JC.grey_jpeg_bytes(75, 52, 85, **kw)
    j = J.parse_jpeg(data)
    assert len(j.components) == 1 and _write(jx, j, J.build_jbrd(j)) == data
    px = np.frombuffer(O.decode(J.transcode(data)).pixels("u8", 1), np.uint8).reshape(52, 75).astype(int)
    d = np.abs(px - np.asarray(Image.open(io.BytesIO(data))).astype(int))
    assert d.max() <= 2 and d.mean() < 0.5


def test_writer_survives_damaged_reconstruction_data(jx):
    """Robustness of the host half of reconstruct(): bit flips in the jbrd bundle (marker order, table definitions, scan scripts,
    reset points) and wild coefficients give an error or some byte string, never a crash or an out-of-bounds access."""
    L = jx.libjxl()
    L.JxlHipDebugWriteJpegSampled.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
    rng = np.random.default_rng(5)
    outcomes = [0, 0]
    for case in (JC.PROGRESSIVE[0], JC.PROGRESSIVE[3], JC.CASES[4]):
        data = JC.jpeg_bytes(case)
        j = J.parse_jpeg(data)
        jbrd = J.build_jbrd(j)
        samp = np.array([[c["h"], c["v"]] for c in j.components], np.uint32)
        coef = np.concatenate([c.reshape(-1) for c in j.coef]).astype(np.int16)
        qt = np.array([j.qt[c["tq"]] for c in j.components], np.int32)
        out = np.zeros(len(data) * 8 + 65536, np.uint8)
        for trial in range(500):
            bad = bytearray(jbrd)
            for pos in rng.integers(0, min(len(bad), 120), 1 + trial % 3):
                bad[pos] ^= 1 << int(rng.integers(0, 8))
            cf = coef
            if trial % 5 == 0:
                cf = coef.copy()
                cf[rng.integers(0, len(cf), 20)] = rng.integers(-32768, 32767, 20)
            n = C.c_size_t(len(out))
            rc = L.JxlHipDebugWriteJpegSampled(bytes(bad), len(bad), j.width, j.height, samp.ctypes.data, cf.ctypes.data, qt.ctypes.data, out.ctypes.data, C.byref(n))
            assert rc in (0, 1, 2)
            outcomes[rc == 0] += 1
    assert outcomes[0] > 0 and outcomes[1] > 0
