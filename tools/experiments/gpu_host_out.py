"""Round 5: streaming with host destinations — Mpixel/s of the library pipeline writing decoded 4K frames into pinned host memory, against the PCIe ceiling."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
streams = bench.make_streams(64, 3840, 2160, 1)
import torch, jpegxl_rs_amd as jx
W, H = 3840, 2160
fb = W * H * 3
t = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); hbuf = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True)
for chunk in (1 << 30, fb):
    n = (1 << 30) // chunk
    hbuf.copy_(t); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        for i in range(n):
            hbuf[i * chunk:(i + 1) * chunk].copy_(t[i * chunk:(i + 1) * chunk], non_blocking=True)
    torch.cuda.synchronize()
    print(json.dumps({"d2h_chunk_bytes": chunk, "gbs": round(3 * n * chunk / (time.perf_counter() - t0) / 1e9, 1)}))
del t, hbuf
for B, infl in ((128, 6), (256, 4)):
    p = jx.Pipeline(0, jobs_in_flight=infl, reserve_frames=B, reserve_width=W, reserve_height=H)
    nout = p.info("slots") + 1
    pins = [jx.PinnedBuffer(B * fb) for _ in range(nout)]
    def run(n, k0):
        ts = []
        t0 = time.perf_counter()
        for k in range(n):
            base = pins[(k0 + k) % nout].ptr
            ts.append(p.submit([streams[(k * 37 + i) % len(streams)] for i in range(B)], "uint8", 3, host_ptrs=[base + i * fb for i in range(B)]))
            if k >= nout - 2: p.wait(ts[k - (nout - 2)])
        for k in range(max(0, n - (nout - 2)), n): p.wait(ts[k])
        return time.perf_counter() - t0
    run(nout, 0); run(4, 0)
    n = 16
    el = run(n, 0)
    print(json.dumps({"frames_per_job": B, "in_flight": infl, "ms_per_job": round(el / n * 1e3, 2), "mpixel_per_s": round(n * B * W * H / 1e6 / el, 1), "gbs_to_host": round(n * B * fb / el / 1e9, 1),
                      "pitched": os.environ.get("JXL_HIP_NO_PITCHED_D2H") is None}), flush=True)
    p.close(); del pins
