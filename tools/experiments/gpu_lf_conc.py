"""GPU helper: N LF stages (SIMT form) of N batches on N streams at once, nothing else running — does the LF launch time depend on its
company?  usage: python tools/experiments/gpu_lf_conc.py [frames] [lane_stride]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import jpegxl_rs_amd as jx
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lf = int(sys.argv[2]) if len(sys.argv) > 2 else 2
streams = bench.make_streams(8, 3840, 2160, 1)
main = torch.cuda.current_stream().cuda_stream
bs = []
for b in range(7):
    bt = jx.BatchDecoder(0)
    for i in range(n):
        bt.add(streams[i % 8], "uint8", 3)
    bt.set_lane_stride(lf, 1)
    if b:
        bt.share_buffers(bs[0]); bt.share_coefficients(bs[0])
    bt.prepare(main)
    bs.append(bt)
for conc in (1, 2, 4, 7):
    ss = [torch.cuda.Stream() for _ in range(conc)]
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        for j in range(conc):
            bs[j].decode_part(5, ss[j].cuda_stream, False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
    print(f"{conc} LF stages at once ({n} frames each, lane stride {lf}): {dt:.1f} ms", flush=True)
