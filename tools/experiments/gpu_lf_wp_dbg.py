"""GPU helper (not a pytest file): the weighted-predictor instantiation of the SIMT LF kernel on the cases the round-4 bench tripped over.
usage: python tools/experiments/gpu_lf_wp_dbg.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import jpegxl_rs_amd as jx
import oracle_lib as O
import synth_lib as S


def enc(seed, w, h, shape=1, mix=1, epf=1):
    S.set_lf_tree_shape(shape)
    try:
        return S.encode_vardct(S.synthetic_image(seed, w, h), seed=seed, strategy_mix=mix, epf_iters=epf)
    finally:
        S.set_lf_tree_shape(0)


def run(tag, datas, refs, lf=4, narrow=0, wide=0, reps=2):
    b = jx.BatchDecoder(0)
    b.set_lane_stride(lf, 1)
    b.add_many(datas, "uint8", 3)
    b.set_option("lf_wp_narrow_test", narrow)
    b.prepare()
    info = [b.info_value(k) for k in ("lf_simt_frames", "lf_legacy_frames", "lf_simt_wp", "lf_simt_lanes")]
    for r in range(reps):
        if wide:
            b.set_option("lf_wide_once", 1)
        try:
            b.decode(); b.finish()
        except Exception as e:
            print(tag, "rep", r, "ERROR", e, info); return
        bad = [i for i in range(len(datas)) if not np.array_equal(b.output(i).reshape(-1), refs[i].reshape(-1))]
        print(tag, "rep", r, "lf", lf, "narrow", narrow, "wide", wide, info, "bad frames:", bad[:10], len(bad))


one = enc(82, 64, 48, mix=0, epf=2); r_one = O.decode(one).pixels("u8", 3)
small = enc(81, 320, 200); r_small = O.decode(small).pixels("u8", 3)
for narrow in (0, 1):
    run("one_group", [one], [r_one], narrow=narrow)
    run("one_group x3", [one] * 3, [r_one] * 3, narrow=narrow)
    run("small", [small], [r_small], narrow=narrow)
    run("small+one", [small, one], [r_small, r_one], narrow=narrow)
run("130 small wide (big launch)", [small] * 130, [r_small] * 130, wide=1, reps=1)
run("130 small simt", [small] * 130, [r_small] * 130, reps=1)
run("130 small simt narrow", [small] * 130, [r_small] * 130, narrow=1, reps=1)
k4 = [enc(1000 + i, 3840, 2160) for i in range(2)]
r4 = [O.decode(d).pixels("u8", 3) for d in k4]
run("4K x2", k4, r4, lf=8, reps=1)
run("4K x8", k4 * 4, r4 * 4, lf=8, reps=2)
run("4K x8 wide", k4 * 4, r4 * 4, lf=8, wide=1, reps=1)
run("4K x40 wide (big)", k4 * 20, r4 * 20, lf=8, wide=1, reps=1)
run("4K x8 narrow", k4 * 4, r4 * 4, lf=8, narrow=1, reps=1)
