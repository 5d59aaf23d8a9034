import os, sys, concurrent.futures as cf
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import jpegxl_rs_amd as jx
import oracle_lib as O
from conftest import fixture_bytes
jpg_jxl, jpg = fixture_bytes("sample_jpg.jxl"), fixture_bytes("sample.jpg")
rgba = O.decode(fixture_bytes("sample.jxl")).pixels("u16", 4)
def work(i):
    try:
        if i % 2 == 0:
            meta, (kind, data) = jx.decoder_builder().reconstruct(jpg_jxl)
            return (i, kind, bytes(data) == jpg if kind == "jpeg" else None, jx.last_error())
        meta, px = jx.decoder_builder(pixel_format=jx.PixelFormat(num_channels=4)).decode_with(fixture_bytes("sample.jxl"), np.uint16)
        return (i, "px", bool(np.array_equal(px.reshape(-1), rgba)), "")
    except Exception as e:
        return (i, "exc", repr(e), jx.last_error())
for T in (1, 8):
    with cf.ThreadPoolExecutor(T) as ex:
        res = list(ex.map(work, range(32)))
    print(T, [r for r in res if r[2] is not True])
