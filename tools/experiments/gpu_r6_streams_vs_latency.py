"""round 6: does the number of HIP streams a process has created (and destroyed) change the latency of a later single-image decode?"""
import os, sys, json, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench, jpegxl_rs_amd as jx
streams = bench.make_streams(2, 3840, 2160, 1)
dec = jx.decoder_builder()
hip = ctypes.CDLL("libamdhip64.so")
def lat():
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); dec.decode_with(streams[0], np.uint8); ts.append((time.perf_counter() - t0) * 1e3)
    return round(sorted(ts[1:])[2], 1)
total = 0
print(json.dumps({"streams_created": total, "single_frame_ms": lat()}), flush=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "destroy"
keep = []
for n in (8, 8, 16, 32, 64, 128):
    hs = []
    for i in range(n):
        s = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
        hs.append(s)
    # use each once (a queue is only attached to a stream that has had work)
    x = torch.zeros(1024, device="cuda:0")
    for s in hs:
        with torch.cuda.stream(torch.cuda.ExternalStream(s.value)):
            x += 1
    torch.cuda.synchronize()
    if mode == "destroy":
        for s in hs: hip.hipStreamDestroy(s)
    else:
        keep += hs
    total += n
    print(json.dumps({"streams_created": total, "mode": mode, "single_frame_ms": lat()}), flush=True)
