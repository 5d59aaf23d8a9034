"""Round 5: the library pipeline (JxlHipPipeline*) on the bench workload — ms per step of 256 fresh 4K frames, device outputs; then 64-thread decode_with."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
distinct = int(os.environ.get("N_DISTINCT", "64"))
steps = int(os.environ.get("STEPS", "30"))
B = int(os.environ.get("BATCH", "256"))
streams = bench.make_streams(distinct, 3840, 2160, 1)
import numpy as np, torch
import jpegxl_rs_amd as jx
W, H = 3840, 2160
SKIP = os.environ.get('SKIP_PIPE') == '1'
if not SKIP:
  exec('''
p = jx.Pipeline(0, timed=1, reserve_frames=B, reserve_width=W, reserve_height=H)
outs = [torch.empty((B, H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
def frames_of(k):
    off = (k * 37) % len(streams)
    return [streams[(off + i) % len(streams)] for i in range(B)]
def run(n):
    p.reset_clock()
    t0 = time.perf_counter()
    ts = []
    for k in range(n):
        o = outs[k % 2]
        ts.append(p.submit(frames_of(k), "uint8", 3, device_ptrs=[o.data_ptr() + i * W * H * 3 for i in range(B)]))
    ends = [p.wait(t)[1] for t in ts]
    torch.cuda.synchronize()
    return time.perf_counter() - t0, ends
run(3); p.collect_times()
el, ends = run(steps)
times, runs = p.collect_times()
print(json.dumps({"ms_per_step": el / steps * 1e3, "mpixel_per_s": steps * B * W * H / 1e6 / el, "ends": [round(e, 1) for e in ends],
                  "stage_ms": {k: v / max(runs, 1) for k, v in times.items()}, "device_bytes": p.info("device_bytes"), "private": p.info("private_plane_jobs"),
                  "prepare_ms_per_job": p.info("prepare_us_total") / 1e3 / max(1, p.info("prepared_jobs"))}))
import oracle_lib as O
ok = True
k = steps - 1
for fi in (0, B // 2, B - 1):
    ok = ok and bool(np.array_equal(outs[k % 2][fi].cpu().numpy().reshape(-1), O.decode(frames_of(k)[fi]).pixels("u8", 3)))
print("verified", ok)
p.close(); del outs; torch.cuda.empty_cache()
''')
# concurrent decode_with callers
import concurrent.futures as cf, threading
for T in [int(x) for x in os.environ.get('THREADS', '1,8,64').split(',')]:
    n = max(T * int(os.environ.get("PER_THREAD", "6")), 6)
    barrier = threading.Barrier(T)
    lat = []
    def work(i):
        dec = jx.decoder_builder(pixel_format=jx.PixelFormat(num_channels=3))
        barrier.wait()
        for k in range(i, n, T):
            t0 = time.perf_counter()
            dec.decode_with(streams[k % len(streams)], np.uint8)
            lat.append(time.perf_counter() - t0)
    with cf.ThreadPoolExecutor(T) as ex:
        list(ex.map(work, range(T)))          # warm-up round
        barrier.reset()
        t0 = time.perf_counter()
        list(ex.map(work, range(T)))
        dt = time.perf_counter() - t0
    print(json.dumps({"threads": T, "frames": n, "s": round(dt, 3), "mpixel_per_s": round(n * W * H / 1e6 / dt, 1), "ms_per_frame_per_thread": round(dt / (n / T) * 1e3, 1), "decode_with_ms_mean": round(1e3 * sum(lat[-n:]) / n, 1)}))
