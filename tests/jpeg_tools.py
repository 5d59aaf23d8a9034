"""Test infrastructure: a baseline-JPEG parser (markers, tables, Huffman-decoded quantised coefficients incl. subsampled /
interleaved scans and restart intervals) and the writer of the `jbrd` box (JPEG bit-stream reconstruction data, libjxl
lib/jxl/jpeg/jpeg_data.cc JPEGData::VisitFields + enc_jpeg_data.cc) — together with tools/jxl_synth.cc's jxlsynth_jpeg_transcode
they turn a real JPEG file (written by Pillow's libjpeg) into the JPEG XL file a lossless JPEG transcode would be, so that
reconstruct() can be checked byte for byte against files this repository did not produce.  Independent of oracle/ and of the product."""
import ctypes as C
import struct

import numpy as np

ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


class Jpeg:
    pass


def parse_jpeg(data: bytes) -> Jpeg:
    """Baseline / extended sequential Huffman JPEG -> structure with everything the file consists of."""
    j = Jpeg()
    j.marker_order, j.app_data, j.com_data, j.quant, j.huff, j.scans = [], [], [], [], [], []
    j.restart_interval, j.tail_data, j.padding_bits = 0, b"", []
    assert data[:2] == b"\xff\xd8"
    pos = 2
    qt, dc_tab, ac_tab = {}, {}, {}
    while True:
        assert data[pos] == 0xFF, "inter-marker data is not handled"
        m = data[pos + 1]
        if m == 0xD9:
            j.marker_order.append(m)
            j.tail_data = data[pos + 2:]
            break
        ln = struct.unpack(">H", data[pos + 2:pos + 4])[0]
        seg = data[pos + 4:pos + 2 + ln]
        j.marker_order.append(m)
        if 0xE0 <= m <= 0xEF:
            j.app_data.append(data[pos + 1:pos + 2 + ln])
        elif m == 0xFE:
            j.com_data.append(data[pos + 1:pos + 2 + ln])
        elif m == 0xDB:
            p, first = 0, len(j.quant)
            while p < len(seg):
                prec, idx = seg[p] >> 4, seg[p] & 15
                n = 128 if prec else 64
                vals = list(struct.unpack(">64H", seg[p + 1:p + 129])) if prec else list(seg[p + 1:p + 65])
                nat = [0] * 64
                for k in range(64):
                    nat[ZIGZAG[k]] = vals[k]
                qt[idx] = nat
                j.quant.append(dict(precision=prec, index=idx, is_last=False, values=nat))
                p += 1 + n
            j.quant[-1]["is_last"] = True
        elif m == 0xC4:
            p = 0
            while p < len(seg):
                cls, idx = seg[p] >> 4, seg[p] & 15
                counts = list(seg[p + 1:p + 17])
                n = sum(counts)
                vals = list(seg[p + 17:p + 17 + n])
                (ac_tab if cls else dc_tab)[idx] = _huff_lut(counts, vals)
                j.huff.append(dict(is_ac=cls, id=idx, is_last=False, counts=counts, values=vals))
                p += 17 + n
            j.huff[-1]["is_last"] = True
        elif m in (0xC0, 0xC1, 0xC2):
            prec, j.height, j.width, nc = struct.unpack(">BHHB", seg[:6])
            assert prec == 8
            j.components = [dict(id=seg[6 + 3 * i], h=seg[7 + 3 * i] >> 4, v=seg[7 + 3 * i] & 15, tq=seg[8 + 3 * i]) for i in range(nc)]
            j.sof = m
        elif m == 0xDD:
            j.restart_interval = struct.unpack(">H", seg[:2])[0]
        elif m == 0xDA:
            ns = seg[0]
            comps = []
            for i in range(ns):
                cid, tabs = seg[1 + 2 * i], seg[2 + 2 * i]
                comps.append(([c["id"] for c in j.components].index(cid), tabs >> 4, tabs & 15))
            ss, se, ahal = seg[1 + 2 * ns:4 + 2 * ns]
            scan = dict(comps=comps, ss=ss, se=se, ah=ahal >> 4, al=ahal & 15, reset_points=[])
            if j.sof != 0xC2:
                assert (ss, se, ahal) == (0, 63, 0)
            end = _decode_scan(j, data, pos + 2 + ln, scan, dc_tab, ac_tab)
            j.scans.append(scan)
            pos = end
            continue
        else:
            raise AssertionError("marker %02x" % m)
        pos += 2 + ln
    j.qt = qt
    return j


def _huff_lut(counts, vals):
    lut, code, k = {}, 0, 0
    for ln in range(1, 17):
        for _ in range(counts[ln - 1]):
            lut[(ln, code)] = vals[k]
            code += 1; k += 1
        code <<= 1
    return lut


def _decode_scan(j, data, pos, scan, dc_tab, ac_tab):
    comps, ss, se, ah, al = scan["comps"], scan["ss"], scan["se"], scan["ah"], scan["al"]
    progressive = j.sof == 0xC2
    maxh = max(c["h"] for c in j.components); maxv = max(c["v"] for c in j.components)
    mcux = -(-j.width // (8 * maxh)); mcuy = -(-j.height // (8 * maxv))
    if not hasattr(j, "coef"):
        j.coef = [np.zeros((mcuy * c["v"], mcux * c["h"], 64), np.int16) for c in j.components]
    inter = len(comps) > 1
    if inter:
        cols, rows = mcux, mcuy
    else:
        c = j.components[comps[0][0]]
        cols = -(-(j.width * c["h"]) // (8 * maxh)); rows = -(-(j.height * c["v"]) // (8 * maxv))
    # entropy-coded segment: unstuff, split at restart markers
    bits = []
    state = dict(buf=0, n=0, pos=pos)

    def getbit():
        if state["n"] == 0:
            b = data[state["pos"]]; state["pos"] += 1
            if b == 0xFF:
                assert data[state["pos"]] == 0; state["pos"] += 1
            state["buf"], state["n"] = b, 8
        state["n"] -= 1
        return (state["buf"] >> state["n"]) & 1

    def decode(lut):
        code, ln = 0, 0
        while True:
            code = (code << 1) | getbit(); ln += 1
            if (ln, code) in lut:
                return lut[(ln, code)]
            assert ln < 17

    def receive(n):
        v = 0
        for _ in range(n):
            v = (v << 1) | getbit()
        return v

    def extend(v, n):
        return v if n == 0 or v >= (1 << (n - 1)) else v - (1 << n) + 1
    last_dc = [0] * len(j.components)
    togo = j.restart_interval
    # progressive scans (T.81 G.1.2; jdphuff.c) + what the canonical writer of the reconstruction side (dec_jpeg_data_writer.cc) would have
    # pending at every block: its end-of-band run and correction bits — where the file starts a new run although that writer would
    # have extended the old one (libjpeg flushes after 937 correction bits, the canonical writer after 65473), jbrd gets a reset point
    eobrun = [0]
    canon = dict(run=0, bits=0)
    block_index = [0]

    def canon_flush():
        canon["run"] = 0; canon["bits"] = 0

    def canon_end_of_band(nbits):
        canon["run"] += 1; canon["bits"] += nbits
        if canon["run"] == 0x7FFF or canon["bits"] > (1 << 16) - 64 + 1:
            canon_flush()

    def prog_block(blk, ci, dct, act):
        p1 = 1 << al
        if ss == 0:
            if ah == 0:
                s = decode(dc_tab[dct])
                last_dc[ci] += extend(receive(s), s)
                blk[0] = last_dc[ci] * p1
            elif getbit():
                blk[0] |= p1
            assert se == 0, "progressive DC scans carry no AC coefficients"
            return
        k = ss
        symbols = 0                                           # coefficient / ZRL symbols of this block so far
        tail_bits = 0                                         # correction bits since the last symbol
        if ah == 0:
            if eobrun[0] > 0:
                eobrun[0] -= 1
                canon_end_of_band(0)
                return
            while k <= se:
                rs = decode(ac_tab[act]); r, s = rs >> 4, rs & 15
                if s == 0 and r < 15:
                    if symbols == 0 and canon["run"] > 0:
                        scan["reset_points"].append(block_index[0]); canon_flush()
                    eobrun[0] = (1 << r) + (receive(r) if r else 0) - 1
                    break
                if symbols == 0:
                    canon_flush()
                symbols += 1
                if s == 0:
                    k += 16; continue
                k += r
                blk[ZIGZAG[k]] = extend(receive(s), s) * p1
                k += 1
            assert k <= se + 1
            if k <= se:
                canon_end_of_band(0)
            return
        m1 = -p1
        if eobrun[0] == 0:
            while k <= se:
                rs = decode(ac_tab[act]); r, s = rs >> 4, rs & 15
                val = 0
                if s:
                    assert s == 1
                    val = p1 if getbit() else m1
                elif r < 15:
                    if symbols == 0 and canon["run"] > 0:
                        scan["reset_points"].append(block_index[0]); canon_flush()
                    eobrun[0] = (1 << r) + (receive(r) if r else 0)
                    break
                if symbols == 0:
                    canon_flush()
                symbols += 1
                tail_bits = 0
                while k <= se:
                    c = int(blk[ZIGZAG[k]])
                    if c != 0:
                        tail_bits += 1
                        if getbit() and (c & p1) == 0:
                            blk[ZIGZAG[k]] = c + (p1 if c >= 0 else m1)
                    else:
                        if r == 0:
                            break
                        r -= 1
                    k += 1
                # (the correction bits read while skipping belong to this symbol: the writer emits them right after it)
                tail_bits = 0
                if s:
                    blk[ZIGZAG[k]] = val
                k += 1
        ended_in_run = False
        if eobrun[0] > 0:
            while k <= se:
                c = int(blk[ZIGZAG[k]])
                if c != 0:
                    tail_bits += 1
                    if getbit() and (c & p1) == 0:
                        blk[ZIGZAG[k]] = c + (p1 if c >= 0 else m1)
                k += 1
            eobrun[0] -= 1
            ended_in_run = True
        if ended_in_run:
            canon_end_of_band(tail_bits)

    for my in range(rows):
        for mx in range(cols):
            if j.restart_interval and togo == 0:
                for _ in range(state["n"]):
                    j.padding_bits.append(getbit())
                assert data[state["pos"]] == 0xFF and 0xD0 <= data[state["pos"] + 1] <= 0xD7
                state["pos"] += 2; state["n"] = 0
                for i in range(len(last_dc)):
                    last_dc[i] = 0
                togo = j.restart_interval
                assert eobrun[0] == 0
                canon_flush()
            for ci, dct, act in comps:
                c = j.components[ci]
                for iy in range(c["v"] if inter else 1):
                    for ix in range(c["h"] if inter else 1):
                        by = my * (c["v"] if inter else 1) + iy; bx = mx * (c["h"] if inter else 1) + ix
                        blk = j.coef[ci][by, bx]
                        if progressive:
                            prog_block(blk, ci, dct, act)
                            block_index[0] += 1
                            continue
                        s = decode(dc_tab[dct])
                        last_dc[ci] += extend(receive(s), s)
                        blk[0] = last_dc[ci]
                        k = 1
                        while k < 64:
                            rs = decode(ac_tab[act])
                            r, s = rs >> 4, rs & 15
                            if s == 0:
                                if r == 15:
                                    k += 16; continue
                                break
                            k += r
                            blk[ZIGZAG[k]] = extend(receive(s), s)
                            k += 1
                        assert k <= 64, "zero run past the block (extra zero runs are not handled)"
            if j.restart_interval:
                togo -= 1
    for _ in range(state["n"]):
        j.padding_bits.append(getbit())
    return state["pos"]


class _Bits:
    def __init__(self):
        self.v, self.n = 0, 0

    def u(self, val, k):
        self.v |= (int(val) & ((1 << k) - 1)) << self.n; self.n += k

    def u32(self, val, dists):
        for sel, (bits, off) in enumerate(dists):
            if val >= off and val - off < (1 << bits):
                self.u(sel, 2); self.u(val - off, bits); return
        raise ValueError(val)

    def bytes(self):
        return self.v.to_bytes((self.n + 7) // 8, "little")


def brotli_compress(data: bytes) -> bytes:
    L = C.CDLL("libbrotlienc.so.1")
    L.BrotliEncoderCompress.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_char_p, C.POINTER(C.c_size_t), C.c_char_p]
    L.BrotliEncoderCompress.restype = C.c_int
    cap = C.c_size_t(len(data) + 1024)
    out = C.create_string_buffer(cap.value)
    assert L.BrotliEncoderCompress(9, 22, 0, len(data), data, C.byref(cap), out) == 1
    return out.raw[:cap.value]


def build_jbrd(j: Jpeg, typed_metadata=False) -> bytes:
    """jbrd box payload for a parsed JPEG (field order of JPEGData::VisitFields; validated against the parser of the product, which in
    turn reads the reference's samples/sample_jpg.jxl)."""
    b = _Bits()
    gray = len(j.components) == 1
    b.u(1 if gray else 0, 1)
    for m in j.marker_order:
        b.u(m - 0xC0, 6)
    for a in j.app_data:
        # type 0: the payload travels in the Brotli stream; 1 / 2 / 3: ICC chunk / Exif / XMP, kept in the codestream / `Exif` / `xml ` box
        b.u32(app_type(a) if typed_metadata else 0, [(0, 0), (0, 1), (1, 2), (2, 4)])
        b.u(len(a) - 1, 16)
    for c in j.com_data:
        b.u(len(c) - 1, 16)
    b.u32(len(j.quant), [(0, 1), (0, 2), (0, 3), (0, 4)])
    for q in j.quant:
        b.u(q["precision"], 1); b.u(q["index"], 2); b.u(1 if q["is_last"] else 0, 1)
    ids = [c["id"] for c in j.components]
    if gray and ids == [1]:
        b.u(0, 2)
    elif ids == [1, 2, 3]:
        b.u(1, 2)
    elif ids == [ord("R"), ord("G"), ord("B")]:
        b.u(2, 2)
    else:
        b.u(3, 2); b.u32(len(ids), [(0, 1), (0, 2), (0, 3), (0, 4)])
        for i in ids:
            b.u(i, 8)
    table_slot = {q["index"]: k for k, q in enumerate(j.quant)}
    for c in j.components:
        b.u(table_slot[c["tq"]], 2)                         # index into the list of tables, in file order
    b.u32(len(j.huff), [(0, 4), (3, 2), (4, 10), (6, 26)])
    for h in j.huff:
        b.u(1 if h["is_ac"] else 0, 1); b.u(h["id"], 2); b.u(1 if h["is_last"] else 0, 1)
        counts = [0] + list(h["counts"])
        maxlen = max(i for i in range(1, 17) if counts[i])
        counts[maxlen] += 1                                 # the sentinel symbol 256 keeps the all-ones code free
        for i in range(17):
            b.u32(counts[i], [(0, 0), (0, 1), (3, 2), (8, 0)])
        for v in list(h["values"]) + [256]:
            b.u32(v, [(2, 0), (2, 4), (4, 8), (8, 1)])
    for s in j.scans:
        b.u32(len(s["comps"]), [(0, 1), (0, 2), (0, 3), (0, 4)])
        b.u(s.get("ss", 0), 6); b.u(s.get("se", 63), 6); b.u(s.get("al", 0), 4); b.u(s.get("ah", 0), 4)
        for ci, dct, act in s["comps"]:
            b.u(ci, 2); b.u(act, 2); b.u(dct, 2)
        b.u32(0, [(0, 0), (0, 1), (0, 2), (3, 3)])          # last_needed_pass
    if 0xDD in j.marker_order:
        b.u(j.restart_interval, 16)
    for s in j.scans:
        rp = s.get("reset_points", [])
        b.u32(len(rp), [(0, 0), (2, 1), (4, 4), (16, 20)])  # reset points: blocks before which the file's writer ended an end-of-band run
        last = -1
        for x in rp:
            b.u32(x - last - 1, [(0, 0), (3, 1), (5, 9), (28, 41)])
            last = x
        b.u32(0, [(0, 0), (2, 1), (4, 4), (16, 20)])        # extra zero runs (libjpeg writes none)
    # (no inter-marker data)
    b.u32(len(j.tail_data), [(0, 0), (8, 1), (16, 257), (22, 65793)])
    zero_pad = any(bit == 0 for bit in j.padding_bits)
    b.u(1 if zero_pad else 0, 1)
    if zero_pad:
        b.u(len(j.padding_bits), 24)
        for bit in j.padding_bits:
            b.u(bit, 1)
    plain = b"".join(a for a in j.app_data if not (typed_metadata and app_type(a))) + b"".join(j.com_data) + j.tail_data
    return b.bytes() + (brotli_compress(plain) if plain else b"")


ICC_TAG, EXIF_TAG, XMP_TAG = b"ICC_PROFILE\0", b"Exif\0\0", b"http://ns.adobe.com/xap/1.0/\0"


def app_type(a: bytes) -> int:
    """jpeg_data.h AppMarkerType of an APPn marker (marker byte, length, payload): 1 ICC chunk, 2 Exif, 3 XMP, 0 anything else."""
    if a[0] == 0xE2 and a[3:3 + len(ICC_TAG)] == ICC_TAG and len(a) >= 17:
        return 1
    if a[0] == 0xE1 and a[3:3 + len(EXIF_TAG)] == EXIF_TAG:
        return 2
    if a[0] == 0xE1 and a[3:3 + len(XMP_TAG)] == XMP_TAG:
        return 3
    return 0


def container(jbrd: bytes, codestream: bytes, exif: bytes = None, xmp: bytes = None, compress_boxes=False, jbrd_last=False) -> bytes:
    """ISO BMFF container as cjxl lays a JPEG transcode out; exif = TIFF data (the box adds the 4-byte offset), xmp = the packet;
    compress_boxes wraps them in `brob` boxes (cjxl's default)."""
    def box(t, payload):
        return struct.pack(">I4s", 8 + len(payload), t) + payload

    def meta(t, payload):
        return box(b"brob", t + brotli_compress(payload)) if compress_boxes else box(t, payload)
    out = b"\x00\x00\x00\x0cJXL \r\n\x87\n" + box(b"ftyp", b"jxl \x00\x00\x00\x00jxl ")
    boxes = []
    if exif is not None:
        boxes.append(meta(b"Exif", b"\0\0\0\0" + exif))
    if xmp is not None:
        boxes.append(meta(b"xml ", xmp))
    if jbrd_last:
        return out + box(b"jxlc", codestream) + b"".join(boxes) + box(b"jbrd", jbrd)
    return out + b"".join(boxes) + box(b"jbrd", jbrd) + box(b"jxlc", codestream)


def transcode(jpeg_bytes: bytes, typed_metadata=False, compress_boxes=False, jbrd_last=False) -> bytes:
    """JPEG file -> the JPEG XL file of its lossless transcode (container with jbrd + VarDCT codestream).  typed_metadata: the way
    cjxl stores ICC / Exif / XMP — the ICC profile in the codestream's image header, Exif / XMP in their own boxes, and only their
    marker sizes in jbrd (otherwise all APPn payloads ride in jbrd's Brotli stream)."""
    import synth_lib as S
    j = parse_jpeg(jpeg_bytes)
    icc = b"".join(a[17:] for a in j.app_data if app_type(a) == 1) if typed_metadata else b""
    exif = next((a[3 + len(EXIF_TAG):] for a in j.app_data if app_type(a) == 2), None) if typed_metadata else None
    xmp = next((a[3 + len(XMP_TAG):] for a in j.app_data if app_type(a) == 3), None) if typed_metadata else None
    if len(j.components) == 1:
        # a grey JPEG: grey image header over a three-channel YCbCr frame whose chroma channels are empty (enc_jpeg_data.cc)
        S.set_icc(icc)
        try:
            qt = np.array([j.qt[j.components[0]["tq"]]] * 3, np.int32)
            cs = S.jpeg_transcode_codestream(j.width, j.height, [0, 0, 0], [None, j.coef[0].reshape(-1, 64), None], qt)
        finally:
            S.set_icc(b"")
        return container(build_jbrd(j, typed_metadata), cs, exif, xmp, compress_boxes, jbrd_last)
    assert len(j.components) == 3
    maxh = max(c["h"] for c in j.components); maxv = max(c["v"] for c in j.components)
    mode_of = {(1, 1): 0, (2, 2): 1, (2, 1): 2, (1, 2): 3}
    # jxl channel order is Cb, Y, Cr = JPEG components 1, 0, 2; a channel's mode is its sampling factor relative to the others
    order = [1, 0, 2]
    modes = [mode_of[(j.components[i]["h"], j.components[i]["v"])] for i in order]
    planes = [np.ascontiguousarray(j.coef[i].reshape(-1, 64)) for i in order]
    qts = np.ascontiguousarray(np.array([j.qt[j.components[i]["tq"]] for i in order], np.int32))
    S.set_icc(icc)
    try:
        cs = S.jpeg_transcode_codestream(j.width, j.height, modes, planes, qts)
    finally:
        S.set_icc(b"")
    return container(build_jbrd(j, typed_metadata), cs, exif, xmp, compress_boxes, jbrd_last)
