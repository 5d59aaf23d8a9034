"""Ad-hoc GPU check used during development (not a pytest): compares the HIP path with the oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import jpegxl_rs_amd as jx
import oracle_lib as O
import synth_lib as S

def cmp(name, data, dtype=np.uint8, nch=0):
    t = time.time()
    try:
        dec = jx.decoder_builder(pixel_format=jx.PixelFormat(num_channels=nch))
        meta, px = dec.decode_with(data, dtype)
    except Exception as e:
        print(name, "GPU ERR", type(e).__name__, e); return
    tg = time.time() - t
    od = O.decode(data)
    kind = {np.uint8: "u8", np.uint16: "u16", np.float32: "f32"}[dtype]
    ref = od.pixels(kind, nch).view(np.dtype(dtype))
    if px.shape != ref.shape:
        print(name, "SHAPE", px.shape, ref.shape); return
    if dtype == np.float32:
        d = np.abs(px - ref)
        print(name, "f32 maxabs", d.max(), "nonequal", int((px != ref).sum()), "of", px.size, "%.2fs" % tg)
    else:
        ne = int((px != ref).sum())
        print(name, "OK" if ne == 0 else "MISMATCH %d/%d maxdiff %d" % (ne, px.size, np.abs(px.astype(int) - ref.astype(int)).max()), "%.2fs" % tg)

fix = os.path.join(ROOT, "tests", "fixtures")
for f, dt, nc in [("sample.jxl", np.uint16, 4), ("sample.jxl", np.uint8, 3), ("bench.jxl", np.uint8, 4)]:
    p = os.path.join(fix, f)
    if os.path.exists(p):
        cmp(f, open(p, "rb").read(), dt, nc)
img = S.synthetic_image(5, 96, 64)
m = np.stack([img[..., 0], img[..., 1], img[..., 2], 255 - img[..., 0]], -1).astype(np.int32)
cmp("modular 96x64 rgba", S.encode_modular(m, 8, True), np.uint8, 4)
img = S.synthetic_image(6, 600, 300)
cmp("modular 600x300 rgb16", S.encode_modular(img.astype(np.int32) * 257, 16, False), np.uint16, 3)
for (w, h, mix, epf, gab) in [(64, 64, 0, 0, 0), (64, 64, 0, 1, 1), (256, 256, 1, 1, 1), (300, 200, 2, 2, 1), (520, 300, 2, 3, 1), (1000, 700, 2, 1, 1)]:
    img = S.synthetic_image(7, w, h)
    data = S.encode_vardct(img, seed=5, strategy_mix=mix, epf_iters=epf, gab=gab)
    cmp("vardct %dx%d mix%d epf%d gab%d" % (w, h, mix, epf, gab), data, np.uint8, 3)
    cmp("vardct %dx%d f32" % (w, h), data, np.float32, 3)
