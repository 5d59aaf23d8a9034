#!/usr/bin/env python3
"""Regenerates tests/golden/*: (1) quantised DCT coefficients of the reference's samples/sample.jpg obtained with an
independent baseline-JPEG Huffman decoder written here (the known answer SURVEY.md App. C derives for
sample_jpg.jxl), (2) small synthesised streams with the oracle's output hashes (regression pins for the float
pipeline — NOT libjxl parity, which the reference cannot pin).  Run in the authoring container."""
import hashlib, json, os, struct, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))

ZZ = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36,
      29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def jpeg_coefficients(path):
    """Baseline (SOF0, Huffman, non-interleaved or interleaved 4:4:4) JPEG -> quantised coefficients [comp][by][bx][64] (natural order)."""
    d = open(path, "rb").read()
    pos = 2
    qt, ht, comps, scan = {}, {}, [], None
    while pos < len(d):
        assert d[pos] == 0xFF
        m = d[pos + 1]
        if m == 0xD9: break
        n = struct.unpack(">H", d[pos + 2:pos + 4])[0]
        body = d[pos + 4:pos + 2 + n]
        if m == 0xDB:
            p = 0
            while p < len(body):
                pq, tq = body[p] >> 4, body[p] & 15
                assert pq == 0
                t = np.zeros(64, np.int32)
                for i in range(64): t[ZZ[i]] = body[p + 1 + i]
                qt[tq] = t
                p += 65
        elif m == 0xC0:
            prec, h, w, nc = struct.unpack(">BHHB", body[:6])
            for i in range(nc):
                cid, hv, tq = body[6 + 3 * i:9 + 3 * i]
                assert hv == 0x11, "only 4:4:4"
                comps.append({"id": cid, "tq": tq})
        elif m == 0xC4:
            p = 0
            while p < len(body):
                tc, th = body[p] >> 4, body[p] & 15
                counts = list(body[p + 1:p + 17])
                syms = list(body[p + 17:p + 17 + sum(counts)])
                code, k, table = 0, 0, {}
                for ln in range(1, 17):
                    for _ in range(counts[ln - 1]):
                        table[(ln, code)] = syms[k]; k += 1; code += 1
                    code <<= 1
                ht[(tc, th)] = table
                p += 17 + sum(counts)
        elif m == 0xDA:
            ns = body[0]
            sel = {}
            for i in range(ns):
                cid, t = body[1 + 2 * i], body[2 + 2 * i]
                sel[cid] = (t >> 4, t & 15)
            scan = (sel, pos + 2 + n)
            break
        pos += 2 + n
    sel, start = scan
    # de-stuff entropy-coded segment
    buf = bytearray()
    p = start
    while True:
        b = d[p]
        if b == 0xFF:
            if d[p + 1] == 0: buf.append(0xFF); p += 2; continue
            break
        buf.append(b); p += 1
    bits = "".join(f"{b:08b}" for b in buf)
    bp = 0
    def huff(table):
        nonlocal bp
        code, ln = 0, 0
        while True:
            code = (code << 1) | int(bits[bp]); bp += 1; ln += 1
            if (ln, code) in table: return table[(ln, code)]
    def receive(n):
        nonlocal bp
        if n == 0: return 0
        v = int(bits[bp:bp + n], 2); bp += n
        return v if v >= (1 << (n - 1)) else v - (1 << n) + 1
    bw, bh = (w + 7) // 8, (h + 7) // 8
    out = np.zeros((len(comps), bh, bw, 64), np.int32)
    pred = [0] * len(comps)
    for by in range(bh):
        for bx in range(bw):
            for ci, c in enumerate(comps):
                td, ta = sel[c["id"]]
                s = huff(ht[(0, td)])
                pred[ci] += receive(s)
                out[ci, by, bx, 0] = pred[ci]
                k = 1
                while k < 64:
                    rs = huff(ht[(1, ta)])
                    r, s = rs >> 4, rs & 15
                    if s == 0:
                        if r == 15: k += 16; continue
                        break
                    k += r
                    out[ci, by, bx, ZZ[k]] = receive(s)
                    k += 1
    return out, [qt[c["tq"]] for c in comps], (w, h)


def main():
    import oracle_lib as O, synth_lib as S
    coefs, qts, (w, h) = jpeg_coefficients(os.path.join(ROOT, "tests", "fixtures", "sample.jpg"))
    np.savez_compressed(os.path.join(HERE, "sample_jpg_coefficients.npz"), coefficients=coefs.astype(np.int16), qtables=np.stack(qts).astype(np.int16), size=np.array([w, h]))
    # synthesised streams + oracle output hashes
    manifest = {}
    cases = [("vardct_64x64_dct8", dict(w=64, h=64, mix=0, epf=0, gab=0)), ("vardct_96x80_mix1_epf1", dict(w=96, h=80, mix=1, epf=1, gab=1)),
             ("vardct_300x200_mix2_epf2", dict(w=300, h=200, mix=2, epf=2, gab=1)), ("vardct_272x264_mix2_epf3", dict(w=272, h=264, mix=2, epf=3, gab=1))]
    for name, c in cases:
        img = S.synthetic_image(42, c["w"], c["h"])
        data = S.encode_vardct(img, seed=9, strategy_mix=c["mix"], epf_iters=c["epf"], gab=c["gab"])
        open(os.path.join(HERE, name + ".jxl"), "wb").write(data)
        d = O.decode(data)
        manifest[name] = {"sha256_stream": hashlib.sha256(data).hexdigest(), "sha256_u8_rgb": hashlib.sha256(d.pixels("u8", 3).tobytes()).hexdigest(),
                          "sha256_f32_rgb": hashlib.sha256(d.pixels("f32", 3).tobytes()).hexdigest(), "width": c["w"], "height": c["h"]}
    # streams exercising the features added later in the round (alpha extra channel, progressive passes + permuted TOC,
    # upsampling with the default 2x weights, DCT128/256 varblocks at arbitrary positions, orientation); hashes are of the
    # stored raster as the oracle renders it
    img = S.synthetic_image(44, 520, 300)
    al = (np.add.outer(np.arange(300), np.arange(520)) % 256).astype(np.uint8)
    extra = [("vardct2_520x300_alpha_passes3_toc", dict(strategy_mix=2, epf_iters=1, gab=1, alpha=al, num_passes=3, permute_toc=5), 4),
             ("vardct2_520x300_ups2_default", dict(strategy_mix=2, epf_iters=2, gab=1, upsampling=2), 3),
             ("vardct2_520x300_ups4_custom_alpha", dict(strategy_mix=1, epf_iters=1, gab=1, upsampling=4, custom_up_weights=1, alpha=al), 4),
             ("vardct2_520x300_mix5_orient7", dict(strategy_mix=5, epf_iters=3, gab=1, orientation=7), 3)]
    for name, kw, nch in extra:
        data = S.encode_vardct(img, seed=10, **kw)
        open(os.path.join(HERE, name + ".jxl"), "wb").write(data)
        d = O.decode(data)
        manifest[name] = {"sha256_stream": hashlib.sha256(data).hexdigest(), "channels": nch, "sha256_u8": hashlib.sha256(d.pixels("u8", nch).tobytes()).hexdigest(),
                          "sha256_f32": hashlib.sha256(d.pixels("f32", nch).tobytes()).hexdigest(), "width": 520, "height": 300}
    yy, xx = np.mgrid[0:300, 0:333].astype(np.float32)
    sq = np.clip((((np.sin(xx / 37.0) + np.cos(yy / 23.0)) * 0.25 + 0.5) * 65535)[..., None] + np.random.default_rng(3).normal(0, 64, (300, 333, 2)), 0, 65535).astype(np.int32)
    data = S.encode_modular(sq, 16, False, 1)
    open(os.path.join(HERE, "modular2_333x300_ga16_squeeze.jxl"), "wb").write(data)
    manifest["modular2_333x300_ga16_squeeze"] = {"sha256_stream": hashlib.sha256(data).hexdigest(), "sha256_u16_ga_le": hashlib.sha256(sq.astype("<u2").tobytes()).hexdigest(), "width": 333, "height": 300}
    img = S.synthetic_image(43, 300, 280).astype(np.int32)
    rgba = np.concatenate([img * 257, (65535 - img[..., :1] * 257)], -1)
    data = S.encode_modular(rgba, 16, True)
    open(os.path.join(HERE, "modular_300x280_rgba16_rct.jxl"), "wb").write(data)
    manifest["modular_300x280_rgba16_rct"] = {"sha256_stream": hashlib.sha256(data).hexdigest(), "sha256_u16_rgba_le": hashlib.sha256(rgba.astype("<u2").tobytes()).hexdigest(), "width": 300, "height": 280}
    manifest["reference_fixtures"] = {
        "sample.jxl": {"sha256_rgba16_be": "4d2f3d44cce6a65bba6c8ec5be174a6ec3d57b0e9542edd28bfa3cde21126bec", "source": "SURVEY.md App. C; jpegxl-rs/src/image.rs:169 (sample.png)"},
        "bench.jxl": {"sha256_rgba8": "0ffc6538fd97f022cefdd58bcf768bd7aa88206f098723b5d23d6c453d9c7a29", "source": "SURVEY.md App. C (bench.png, verified equal with Pillow while authoring)"}}
    json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)
    print("golden written:", sorted(manifest))


if __name__ == "__main__":
    main()
