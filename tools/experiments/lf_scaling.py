"""round 6: the wave-wide LF kernel (one LF-group stream per wavefront, LfDecodeKernel<false>) as the number of resident wavefronts grows — one frame's four wavefronts take 55 ms (gradient tree) / 108 ms
(weighted-predictor tree); what do 16 ... 256 frames' take?   JXL_HIP_TIME_LF=1 prints the stage's decode time per launch.
usage: JXL_HIP_TIME_LF=1 python tools/experiments/lf_scaling.py [tree_shape]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import jpegxl_rs_amd as jx
shape = int(sys.argv[1]) if len(sys.argv) > 1 else 1
streams = bench.make_streams(8, 3840, 2160, 1, texture=5.0 if shape else 0.0, tree_shape=shape)
for n in (1, 4, 16, 64, 128, 256):
    for fb in (-1, 2):
        b = jx.BatchDecoder(0)
        b.add_many([streams[i % len(streams)] for i in range(n)], "uint8", 3, threads=8)
        b.set_lane_stride(64, 1)
        b.prepare()
        b.set_option("lf_force_big", fb)
        print(f"--- tree_shape {shape}, {n} frames, lf_force_big {fb}", file=sys.stderr, flush=True)
        for _ in range(2):
            b.decode(); b.finish()
        del b
