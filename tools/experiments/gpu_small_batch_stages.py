"""Round 5: stage times of SMALL batches (1 / 8 / 64 4K frames, nothing else on the GPU) under the lane packings of the two entropy stages — what the latency-mode
scheduler should pick (lane_stride_lf / lane_stride_hf)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
tree_shape = int(os.environ.get("TREE_SHAPE", "0"))
texture = float(os.environ.get("TEXTURE", "0"))
streams = bench.make_streams(16, 3840, 2160, 1, texture=texture, tree_shape=tree_shape)
import torch
import jpegxl_rs_amd as jx
for n in (1, 8, 64):
    for lf, hf in ((64, 1), (64, 64), (64, 16), (64, 4), (8, 1)):
        b = jx.BatchDecoder(0)
        b.add_many([streams[i % len(streams)] for i in range(n)], "uint8", 3, threads=8)
        b.set_lane_stride(lf, hf)
        b.prepare()
        b.decode(); b.finish()
        b.collect_times()
        for _ in range(3):
            b.decode_timed()
        b.finish()
        t, runs = b.collect_times()
        print(json.dumps({"frames": n, "lane_stride_lf": lf, "lane_stride_hf": hf, **{k: round(v / runs, 2) for k, v in t.items()}}), flush=True)
        del b
