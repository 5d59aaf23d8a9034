"""round 6: tools/api_concurrent with 64 (and 8) caller threads under scheduler knobs (largest job, jobs in flight, quiet window): JXL_HIP_SCHED_* of csrc/scheduler.cc
usage: gpu_r6_api_sweep.py"""
import os, sys, json, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
streams = bench.make_streams(64, 3840, 2160, 1)
exe = os.path.join(ROOT, "tools", "_build", "api_concurrent"); lib = os.path.join(ROOT, "jpegxl-rs_amd", "lib", "libjxl.so")
with tempfile.TemporaryDirectory() as d:
    for i, s in enumerate(streams):
        open(os.path.join(d, f"f{i:03d}.jxl"), "wb").write(s)
    K = lambda mj, jobs=None: {"JXL_HIP_SCHED_MAX_JOB": str(mj), **({"JXL_HIP_SCHED_JOBS": str(jobs)} if jobs else {})}
    for knobs in [{}, {}, {}, K(64, 3), {}]:
        env = dict(os.environ, GPU_MAX_HW_QUEUES="16", **knobs)
        out = subprocess.run([exe, lib, d, "1,8,64", "8", "3"], env=env, capture_output=True, text=True, timeout=600)
        lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{") and "threads" in l]
        print(json.dumps({"knobs": knobs, **{f"t{l['threads']}": (l["mpixel_per_s"], l["latency_ms_median"]) for l in lines}}), flush=True)
