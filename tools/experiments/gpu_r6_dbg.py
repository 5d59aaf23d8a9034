import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import jpegxl_rs_amd as jx
streams = bench.make_streams(4, 3840, 2160, 1)
n = int(sys.argv[1])
b = jx.BatchDecoder(0)
b.add_many([streams[i % len(streams)] for i in range(n)], "uint8", 3, threads=8)
b.set_lane_stride(64, 1)
b.prepare()
try:
    b.decode(); b.finish()
    print(n, os.environ.get("JXL_HIP_LF_BIG"), "ok", flush=True)
except Exception as e:
    print(n, os.environ.get("JXL_HIP_LF_BIG"), "FAIL", repr(e)[:200], flush=True)
