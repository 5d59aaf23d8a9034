"""GPU helper: which shared resource are the pixel stages (IDCT, filters) short of when the HF stage of another batch runs beside them?
The tail of a prepared batch (JxlHipBatchDecodePart 4) is timed alone, beside the real HF stage of a second batch, and beside synthetic
co-runners (tools/microbench/spin.hip) that each exercise one resource.   usage: python tools/experiments/gpu_corun.py [frames]"""
import ctypes as C, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import jpegxl_rs_amd as jx
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
spin = C.CDLL(os.path.join(ROOT, "tools", "microbench", "libspin.so"))
spin.spin_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p]
streams = bench.make_streams(8, 3840, 2160, 1)
main = torch.cuda.current_stream()
out = torch.empty((n, 2160, 3840, 3), dtype=torch.uint8, device="cuda")
bs = []
for b in range(2):
    bt = jx.BatchDecoder(0)
    for i in range(n):
        bt.add(streams[i % 8], "uint8", 3, device_ptr=out.data_ptr() + i * 3840 * 2160 * 3)
    bt.set_lane_stride(2, 1)
    if b:
        bt.share_buffers(bs[0])
    bt.prepare(main.cuda_stream)
    bs.append(bt)
for bt in bs:                       # one complete decode each: planes clean, IDCT variants known
    bt.decode(main.cuda_stream); bt.finish(main.cuda_stream)
s2 = torch.cuda.Stream()
big = torch.empty(8 << 30, dtype=torch.uint8, device="cuda")      # 8 GiB for the scattered-access co-runners
sink = torch.zeros(4, dtype=torch.int32, device="cuda")

def timed_tail(corun=None):
    """front + HF of batch 0 done; then its tail on the main stream, optionally with a co-runner started just before on s2"""
    A, B = bs
    A.decode_part(1, main.cuda_stream); A.decode_part(3, main.cuda_stream)
    B.decode_part(1, main.cuda_stream)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    if corun:
        with torch.cuda.stream(s2):
            e[2].record(s2); corun(s2); e[3].record(s2)
        time.sleep(0.002)
    e[0].record(main)
    A.decode_part(4, main.cuda_stream)
    e[1].record(main)
    torch.cuda.synchronize()
    if corun == hf_real:
        B.decode_part(4, main.cuda_stream); torch.cuda.synchronize()     # consume B's coefficients again
    return e[0].elapsed_time(e[1]), (e[2].elapsed_time(e[3]) if corun else 0.0)

def hf_real(s):
    bs[1].decode_part(3, s.cuda_stream)

def spinner(blocks, threads, lds, mode, ms=80.0, lanes=64, prio=0):
    def f(s):
        spin.spin_launch(s.cuda_stream, blocks, threads, lds, big.data_ptr(), big.numel(), mode, ms, lanes, prio, sink.data_ptr())
    return f

cases = [("alone", None), ("HF stage of another batch", hf_real),
         ("spin: 256 x 256 threads, ALU only", spinner(256, 256, 0, 0)),
         ("spin: 256 x 256 threads, ALU only, s_setprio 3", spinner(256, 256, 0, 0, prio=1)),
         ("spin: 256 x 256 threads, ALU + 80 KB LDS held", spinner(256, 256, 80 * 1024, 1)),
         ("spin: 256 x 256 threads, ALU + scattered loads over 8 GiB (34 lanes)", spinner(256, 256, 0, 2, lanes=34)),
         ("spin: 256 x 256 threads, ALU + scattered stores over 8 GiB (34 lanes)", spinner(256, 256, 0, 3, lanes=34)),
         ("spin: 1024 x 64 threads, ALU only", spinner(1024, 64, 0, 0)),
         ("spin: 2048 x 256 threads (8 waves per SIMD), ALU only", spinner(2048, 256, 0, 0))]
for name, c in cases:
    r = [timed_tail(c) for _ in range(3)]
    print(f"{name}: tail {min(x[0] for x in r):.1f} ms (co-runner {min(x[1] for x in r):.1f} ms)", flush=True)
