"""Scratch: where does a free-running stream differ between the HIP path and the oracle (not a pytest)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import jpegxl_rs_amd as jx
import oracle_lib as O, synth_lib as S
kw = eval(sys.argv[1])
data = S.encode_modular_free(**kw)
nch = kw.get("nchan", 3) + (1 if kw.get("has_alpha") else 0)
ref = O.decode(data).pixels("f32", nch).view(np.float32).reshape(kw["h"], kw["w"], nch)
_, px = jx.decoder_builder(pixel_format=jx.PixelFormat(num_channels=nch)).decode_with(data, np.float32)
px = px.reshape(ref.shape)
bad = np.argwhere(px != ref)
print("differing samples:", len(bad), "of", px.size)
for y, x, c in bad[:12]:
    print((y, x, c), px[y, x, c] * 65535, ref[y, x, c] * 65535)
