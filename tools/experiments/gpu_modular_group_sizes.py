"""Which Modular group sizes / channel layouts the device path decodes (lossless round trip); bisecting helper, not a pytest."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import synth_lib as S
import jpegxl_rs_amd as jx
from test_gpu_parity import _smooth_image
for shift in (3, 2):
    for (w, h, nch, bits, sq) in [(1300, 1100, 4, 16, 1), (1300, 1100, 4, 16, 0), (1300, 1100, 3, 16, 1), (1300, 1100, 4, 8, 1), (1300, 1100, 3, 8, 1), (1300, 1100, 1, 8, 1), (1100, 1300, 3, 8, 1),
                                   (1024, 1024, 3, 8, 1), (1025, 1024, 3, 8, 1), (1300, 600, 3, 8, 1), (1300, 1100, 1, 16, 1), (2048, 300, 1, 8, 1), (2049, 300, 1, 8, 1), (4000, 300, 1, 8, 1)]:
        img = _smooth_image(10, h, w, nch, bits)
        S.set_modular_group_shift(shift)
        try:
            data = S.encode_modular(img, bits, True, sq)
        finally:
            S.set_modular_group_shift(1)
        try:
            meta, px = jx.decoder_builder().decode_with(data, np.uint8 if bits == 8 else np.uint16)
            print(shift, (w, h, nch, bits, sq), "ok" if np.array_equal(px.reshape(img.shape), img) else "MISMATCH %d" % int((px.reshape(img.shape) != img).sum()), flush=True)
        except Exception as e:
            print(shift, (w, h, nch, bits, sq), "ERR", str(e)[:100], flush=True)
