"""GPU helper: cost of LF SIMT stages in flight for the pixel stages (tail) of another batch.  usage: python tools/experiments/gpu_corun_lf.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import jpegxl_rs_amd as jx
import bench
n = 256
streams = bench.make_streams(8, 3840, 2160, 1)
main = torch.cuda.current_stream()
out = torch.empty((n, 2160, 3840, 3), dtype=torch.uint8, device="cuda")
nlf = int(os.environ.get("NLF", "8"))
lane = int(os.environ.get("LANE", "8"))
bs = []
for b in range(1 + nlf):
    bt = jx.BatchDecoder(0)
    for i in range(n):
        bt.add(streams[i % 8], "uint8", 3, device_ptr=out.data_ptr() + i * 3840 * 2160 * 3)
    bt.set_lane_stride(lane, 1)
    if b:
        bt.share_buffers(bs[0]); bt.share_coefficients(bs[0])
    bt.prepare(main.cuda_stream)
    bs.append(bt)
bs[0].decode(main.cuda_stream); bs[0].finish(main.cuda_stream)
sides = [torch.cuda.Stream() for _ in range(nlf)]

def run(k, part=5):
    A = bs[0]
    A.decode_part(1, main.cuda_stream); A.decode_part(3, main.cuda_stream)
    torch.cuda.synchronize()
    for j in range(k):
        bs[1 + j].decode_part(part, sides[j].cuda_stream)
    time.sleep(0.01)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record(main); A.decode_part(4, main.cuda_stream); e1.record(main)
    A.decode_part(3, main.cuda_stream); e2.record(main)     # then an HF stage beside them
    torch.cuda.synchronize()
    A.decode_part(4, main.cuda_stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1), e1.elapsed_time(e2)

for k in (0, 1, 2, 4, 8):
    if k > nlf: break
    r = [run(k) for _ in range(2)]
    print(f"{k} LF stages in flight (lane stride {lane}, env {os.environ.get('JXL_HIP_LF_PRIO')}): tail {min(x[0] for x in r):.1f} ms, HF {min(x[1] for x in r):.1f} ms", flush=True)
