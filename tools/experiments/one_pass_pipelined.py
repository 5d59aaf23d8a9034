"""round 6: BASELINE config 3's per-GPU share (128 4K frames, decoded once, prepare included) as J jobs through a library pipeline instead of one JxlHipBatch: the parse of job k + 1 runs while job k decodes.
usage: python tools/experiments/one_pass_pipelined.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
import jpegxl_rs_amd as jx
W, H, n = 3840, 2160, 128
streams = bench.make_streams(64, W, H, 1)
frames = [streams[i % len(streams)] for i in range(n)]
dst = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda:0")
fb = W * H * 3
for per_job, opts in [(128, {}), (64, {}), (128, dict(small_job_frames=128)), (64, dict(small_job_frames=64)), (32, dict(small_job_frames=64)), (16, dict(small_job_frames=64)), (32, dict(small_job_frames=64, hf_streams=3)),
                      (32, dict(small_job_frames=64, prepare_threads=4)), (43, dict(small_job_frames=64)), (8, dict(small_job_frames=64))]:
    o = dict(jobs_in_flight=8, lf_streams=8, hf_streams=2, prepare_threads=3, parse_threads=8, lane_stride_lf=8, lane_stride_hf=1, wide_first=4, reserve_frames=per_job, reserve_width=W, reserve_height=H)
    o.update(opts)
    p = jx.Pipeline(0, **o)
    ts = []
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tk = [p.submit(frames[k:k + per_job], "uint8", 3, device_ptrs=[dst.data_ptr() + i * fb for i in range(k, min(n, k + per_job))]) for k in range(0, n, per_job)]
        for t in tk: p.wait(t)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps({"frames_per_job": per_job, "options": opts, "ms": [round(t, 1) for t in ts], "gpixel_per_s_best_after_first": round(n * W * H / 1e6 / min(ts[1:]), 2)}), flush=True)
    p.close(); del p
