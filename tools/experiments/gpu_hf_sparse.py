"""Round 5: the SIMT HF kernel with very sparse wavefronts (1 / 2 / 4 group streams per wavefront) on small batches — is a lane that has a wavefront to itself faster than the one-stream-per-wavefront kernel (HfDecodeKernel)?"""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bench
    streams = bench.make_streams(16, 3840, 2160, 1, texture=float(os.environ.get("TEXTURE", "0")))
    import jpegxl_rs_amd as jx
    for n in (1, 8, 64):
        b = jx.BatchDecoder(0)
        b.add_many([streams[i % len(streams)] for i in range(n)], "uint8", 3, threads=8)
        b.set_lane_stride(64, 1)
        b.prepare(); b.decode(); b.finish(); b.collect_times()
        for _ in range(3): b.decode_timed()
        b.finish()
        t, runs = b.collect_times()
        print(json.dumps({"frames": n, "lanes_per_wave": os.environ.get("JXL_HIP_HF_LANES"), "lanes_per_wg": os.environ.get("JXL_HIP_HF_LANES_PER_WG"), "hf_ms": round(t["hf_ms"] / runs, 2), "lf_ms": round(t["lf_ms"] / runs, 2)}), flush=True)
        del b
else:
    for lpw, cap in ((None, None), (1, 16), (2, 32), (4, 64), (8, 128)):
        env = dict(os.environ)
        if lpw: env["JXL_HIP_HF_LANES"] = str(lpw); env["JXL_HIP_HF_LANES_PER_WG"] = str(cap)
        out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout
        print("".join(l + "\n" for l in out.splitlines() if l.startswith("{")), end="", flush=True)
