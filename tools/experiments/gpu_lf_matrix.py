"""GPU helper: LF SIMT kernel time (JXL_HIP_TIME_LF=1 prints decode / placement per launch) over batch sizes and lanes per wavefront.
usage: JXL_HIP_TIME_LF=1 python tools/experiments/gpu_lf_matrix.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import jpegxl_rs_amd as jx
import bench
distinct = int(os.environ.get("DISTINCT", "8"))
streams = bench.make_streams(distinct, 3840, 2160, 1)
main = torch.cuda.current_stream().cuda_stream
for n in (int(v) for v in os.environ.get("FRAMES", "32,64,128,256").split(",")):
    for lf in (int(v) for v in os.environ.get("STRIDES", "1,2,4,8,16").split(",")):
        bt = jx.BatchDecoder(0)
        for i in range(n):
            bt.add(streams[i % distinct], "uint8", 3)
        bt.set_lane_stride(lf, 1)
        bt.prepare(main)
        print(f"--- {n} frames, lane stride {lf}", file=sys.stderr, flush=True)
        for _ in range(2):
            bt.decode_part(5, main, False)
        torch.cuda.synchronize()
        del bt
