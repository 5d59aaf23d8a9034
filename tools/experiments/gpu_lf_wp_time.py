"""GPU helper (not a pytest file): LF stage alone (no other batch on the GPU), 4K frames, gradient tree vs the cjxl-shaped (weighted-predictor) tree,
SIMT at several lanes per wavefront vs the one-wavefront-per-stream kernel.   usage: python tools/experiments/gpu_lf_wp_time.py [frames] [distinct]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import jpegxl_rs_amd as jx
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
distinct = int(sys.argv[2]) if len(sys.argv) > 2 else 16
strides = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [64, 4, 8, 16]
stream = torch.cuda.current_stream().cuda_stream
for shape in ((0,) if os.environ.get("JXL_AB_SHAPE0_ONLY") else (0, 1)):
    streams = bench.make_streams(distinct, 3840, 2160, 1, texture=5.0, tree_shape=shape)
    for lf in strides:
        b = jx.BatchDecoder(0)
        out = torch.empty((n, 2160, 3840, 3), dtype=torch.uint8, device="cuda")
        b.add_many([streams[i % distinct] for i in range(n)], "uint8", 3, device_ptrs=[out.data_ptr() + i * 3840 * 2160 * 3 for i in range(n)], threads=16)
        b.set_lane_stride(lf, 1); b.prepare(stream)
        b.decode(stream); b.finish(stream)
        for _ in range(2):
            b.decode_timed(stream)
        b.finish(stream)
        t, runs = b.collect_times()
        info = [b.info_value(k) for k in ("lf_simt_frames", "lf_simt_lanes")]
        print(f"tree shape {shape} lane stride {lf}: simt frames/lanes/wp {info} lf {t['lf_ms'] / runs:.2f} ms, lfpost {t['lfpost_ms'] / runs:.2f}, hf {t['hf_ms'] / runs:.2f}, idct {t['idct_ms'] / runs:.2f}, filter {t['filter_ms'] / runs:.2f}", flush=True)
        del b, out
        torch.cuda.empty_cache()
