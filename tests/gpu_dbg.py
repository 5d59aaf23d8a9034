import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import jpegxl_rs_amd as jx
import oracle_lib as O
import synth_lib as S
img = S.synthetic_image(7, 520, 300)
for s in [int(a) for a in sys.argv[1:]] or [21]:
    data = S.encode_vardct(img, seed=5, strategy_mix=100 + s, epf_iters=1, gab=1)
    try:
        meta, px = jx.decoder_builder(pixel_format=jx.PixelFormat(num_channels=3)).decode_with(data, np.float32)
    except Exception as e:
        print(s, "ERR", type(e).__name__, e, flush=True); continue
    ref = O.decode(data).pixels("f32", 3).view(np.float32)
    d = np.abs(px - ref)
    print(s, "maxabs", d.max(), "nonequal", int((px != ref).sum()), "of", px.size, flush=True)
