"""Times free-running Modular streams (general MA trees) through the one-shot API (not a pytest).  argv[1] = library path override."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import jpegxl_rs_amd as jx
if len(sys.argv) > 1:
    jx.LIBJXL_PATH = sys.argv[1]
import synth_lib as S
cases = {"depth5_plain": dict(tree_flags=S.TREE_ALL_PREDICTORS, tree_depth=5), "depth5_wp": dict(tree_flags=S.TREE_ALL_PREDICTORS | S.TREE_WP, tree_depth=5),
         "depth3_plain": dict(tree_flags=0, tree_depth=3), "depth7_wp": dict(tree_flags=S.TREE_ALL_PREDICTORS | S.TREE_WP | S.TREE_MULTIPLIERS, tree_depth=7)}
for name, kw in cases.items():
    data = S.encode_modular_free(seed=5, w=2048, h=2048, bits=16, **kw)
    dec = jx.decoder_builder()
    dec.decode_with(data, np.uint16)
    t = time.time()
    for _ in range(3):
        dec.decode_with(data, np.uint16)
    ms = (time.time() - t) / 3 * 1e3
    print("%s: %.1f ms (%.2f us per sample of a 256x256x3 group)" % (name, ms, ms * 1e3 / (65536 * 3)))
