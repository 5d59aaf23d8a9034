import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import jpegxl_rs_amd as jx
import oracle_lib as O
import synth_lib as S
from gpu_lf_wp_dbg_common import enc


def run(tag, datas, refs, lf=4, narrow=0, wide=0):
    b = jx.BatchDecoder(0)
    b.set_lane_stride(lf, 1)
    b.add_many(datas, "uint8", 3)
    b.set_option("lf_wp_narrow_test", narrow)
    b.prepare()
    if wide:
        b.set_option("lf_wide_once", 1)
    b.decode(); b.finish()
    out = []
    for i in range(len(datas)):
        px = b.output(i).reshape(-1); r = refs[i].reshape(-1)
        nbad = int((px != r).sum())
        lfq = [b.debug_read(i, "lfq", c, np.int32) for c in range(3)]
        out.append(lfq)
        print(tag, "frame", i, "bad px", nbad, "of", r.size, "first bad", int(np.argmax(px != r)) if nbad else -1)
    return out


one = enc(82, 64, 48, mix=0, epf=2); r_one = O.decode(one).pixels("u8", 3)
small = enc(81, 320, 200); r_small = O.decode(small).pixels("u8", 3)
base = run("plain   [small, one]", [small, one], [r_small, r_one])
for tag, kw in (("narrow  [small, one]", dict(narrow=1)), ("wide    [small, one]", dict(wide=1)), ("narrow lf1 [small, one]", dict(narrow=1, lf=1)), ("narrow lf64?", dict(narrow=1, lf=32))):
    got = run(tag, [small, one], [r_small, r_one], **kw)
    for i in range(2):
        for c in range(3):
            d = np.nonzero(base[i][c] != got[i][c])[0]
            if d.size:
                print("   lfq differs: frame", i, "chan", c, "count", d.size, "first idx", d[:8], "base", base[i][c][d[:4]], "got", got[i][c][d[:4]])
base2 = run("plain   [one, small]", [one, small], [r_one, r_small])
got = run("narrow  [one, small]", [one, small], [r_one, r_small], narrow=1)
for i in range(2):
    for c in range(3):
        d = np.nonzero(base2[i][c] != got[i][c])[0]
        if d.size:
            print("   lfq differs: frame", i, "chan", c, "count", d.size, "first idx", d[:8])
