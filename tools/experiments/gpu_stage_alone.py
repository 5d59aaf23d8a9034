"""Stage times (HIP events) of one prepared batch of 4K frames decoded alone — nothing else on the GPU.  usage: gpu_stage_alone.py <frames> <decodes> [texture]
A/B of library builds: JXL_HIP_LIBJXL=<path to libjxl.so>."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
n, reps = int(sys.argv[1]), int(sys.argv[2])
texture = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
streams = bench.make_streams(min(n, 32), 3840, 2160, 1, texture=texture) if texture else bench.make_streams(min(n, 32), 3840, 2160, 1)
import jpegxl_rs_amd as jx
b = jx.BatchDecoder(0)
b.add_many([streams[i % len(streams)] for i in range(n)], "uint8", 3, threads=8)
b.set_lane_stride(8, 1)
b.prepare()
b.decode(); b.finish()
b.decode_timed(); b.finish(); b.collect_times()
for _ in range(reps):
    b.decode_timed(); b.finish()
t, runs = b.collect_times()
print(json.dumps({"lib": os.environ.get("JXL_HIP_LIBJXL", "default"), "frames": n, "runs": runs, **{k: round(v / max(runs, 1), 3) for k, v in t.items()}}))
