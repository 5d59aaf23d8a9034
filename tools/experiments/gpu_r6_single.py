"""Round 6: stage times of ONE 4K frame alone on the GPU (the latency a decode_with caller sees), for the gradient LF tree (TREE_SHAPE=0) and the
cjxl-shaped weighted-predictor tree (1); and of bench.jxl (Modular)."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import torch
import numpy as np
import jpegxl_rs_amd as jx
for tree_shape in (0, 1):
    streams = bench.make_streams(2, 3840, 2160, 1, texture=float(os.environ.get("TEXTURE", "0")), tree_shape=tree_shape)
    for n in (1, 4):
        b = jx.BatchDecoder(0)
        b.add_many([streams[i % len(streams)] for i in range(n)], "uint8", 3, threads=8)
        b.set_lane_stride(64, 1)
        if n <= 8 and not os.environ.get('NO_SPARSE'): b.set_option('hf_lanes_per_wave', 1)
        b.prepare()
        b.decode(); b.finish()
        b.collect_times()
        for _ in range(3):
            b.decode_timed()
        b.finish()
        t, runs = b.collect_times()
        print(json.dumps({"tree_shape": tree_shape, "frames": n, **{k: round(v / runs, 2) for k, v in t.items()}}), flush=True)
        del b
data = open(os.path.join(ROOT, "tests", "fixtures", "bench.jxl"), "rb").read()
d = jx.decoder_builder()
for i in range(4):
    t0 = time.perf_counter(); meta, px = d.decode_with(data, np.uint8); t1 = time.perf_counter()
    print(json.dumps({"bench_jxl_ms": round((t1 - t0) * 1e3, 1), "shape": list(px.shape)}), flush=True)
