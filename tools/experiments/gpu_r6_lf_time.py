"""Round 6: LF stage of ONE 4K frame split into entropy decode and varblock placement (JXL_HIP_TIME_LF=1 prints both on stderr)."""
import os, sys, json
os.environ["JXL_HIP_TIME_LF"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import jpegxl_rs_amd as jx
for tree_shape in (0, 1):
    streams = bench.make_streams(1, 3840, 2160, 1, texture=float(os.environ.get("TEXTURE", "0")), tree_shape=tree_shape)
    print("stream bytes", len(streams[0]), flush=True)
    b = jx.BatchDecoder(0)
    b.add_many(streams, "uint8", 3, threads=1)
    b.set_lane_stride(64, 1)
    b.prepare()
    for _ in range(3):
        b.decode(); b.finish()
    del b
