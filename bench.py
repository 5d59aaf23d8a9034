#!/usr/bin/env python3
"""bench.py — Mpixel/s of JPEG XL VarDCT (d1) decode of 3840x2160 frames on MI355X (BASELINE.json metric).

A "step" decodes one batch of B synthetic 4K VarDCT frames per GPU (u8 RGB out) through the C-ABI batch API
(include/jxl_hip.h); decoded pixels stay in HBM (a torch tensor).  Two modes, both run by default:

  streaming (the headline `value`): every step decodes a batch of FRESH inputs.  Prepare workers refill a ring of batch objects
      (JxlHipBatchReset / AddImages: parse on host threads into pinned staging; Prepare: tables + upload on a copy stream) and
      enqueue the latency-bound LF stage on side streams, ten or so batches ahead; the main thread issues HF decode, IDCT and the
      filter / write stages of step k.  The host never holds a decoded stream longer than the ring.
  resident (`resident_mpixel_per_s`): the same pipeline over batch objects prepared before the timed region — compressed streams
      and tables are in HBM when the clock starts (the configuration round 2's number was quoted on).

`workload_realistic` repeats the headline mode on frames with photograph-like texture (1.5 - 2.5 bits per pixel).
With N > 1 ranks every rank decodes its own shard of the batch (weak scaling, no data-path collective) and the decoded pixels are
gathered to rank 0 over RCCL (BASELINE.json north_star); `decode_only_mpixel_per_s` / `gather_ms` split the two.
--scaling strong --total-frames T fixes the job instead (BASELINE config 3: 1024 frames over the node): every rank decodes
T / N frames per step, in chunks of at most --batch.

Besides the headline the N = 1 line reports what a caller of the drop-in API sees: single_frame_ms (config 2: one 4K frame through
decode_with, host to host), one_pass (config-3 shape: a fresh batch of 128 frames decoded once, cold, no pipelining, with and
without the host-side parse + upload), pcie_inclusive (the same pass plus the copy of the pixels back to host memory),
step_end_ms / steady_state_ms_per_step (the pipeline fill is inside the K timed steps) and verified_vs_oracle (pixels of the
timed batches against the CPU oracle).

Contract: python bench.py --gpus N --steps K --warmup W  → rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); ~6300 GB/s measured copy rate


def _pmc_bytes(kernels, frames):
    """HBM bytes of `frames` frames through the named kernels from the PMC passes (profiles/pmc_traffic.json), or None."""
    try:
        per = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["per_kernel"]
        return int(sum(per[k] for k in kernels) * frames)
    except Exception:
        return None


def _valu_issue(frames, s_per_step, steady_ms):
    """VALU issue rate of a step: wavefront-instructions per second over the chip's peak (1024 SIMDs, one wave64 FP32 instruction per 4 cycles, 2.4 GHz) — a lower bound of
    how busy the vector ALUs are (transcendental and 64-bit operations take longer than 4 cycles).  Counts from the SQ counter passes of the round (profiles/sq_valu.json)."""
    try:
        per = json.load(open(os.path.join(ROOT, "profiles", "sq_valu.json")))["per_kernel"]
        instr = sum(v["valu_wave_instr_per_frame"] for v in per.values() if v["calls"] == 2) * frames
        peak = 256 * 4 * 2.4e9 / 4
        out = {"wave_instr_per_step": int(instr), "peak_wave_instr_per_s": peak, "frac": round(instr / s_per_step / peak, 4)}
        if steady_ms:
            out["frac_steady_state"] = round(instr / (steady_ms * 1e-3) / peak, 4)
        return out
    except Exception:
        return None


def _textured(img, amp, seed):
    """Photograph-like detail on top of the smooth synthetic picture: fine grain plus 4x4-pixel texture, `amp` sRGB levels strong, correlated
    between the channels.  amp 5 takes a 4K frame from 0.8 to about 2 bits per pixel at distance 1 — what real `cjxl -d 1` photographs run."""
    import numpy as np
    rng = np.random.default_rng(seed)
    h, w, _ = img.shape
    n = rng.standard_normal((h, w)).astype(np.float32)
    n = (n + np.roll(n, 1, 0) + np.roll(n, 1, 1) + np.roll(n, (1, 1), (0, 1))) / 2
    m = np.kron(rng.standard_normal((h // 4 + 1, w // 4 + 1)).astype(np.float32), np.ones((4, 4), np.float32))[:h, :w]
    t = (0.7 * n + 0.7 * m) * amp
    return np.clip(img.astype(np.float32) + t[:, :, None] * np.array([1.0, 0.9, 0.8], np.float32), 0, 255).astype(np.uint8)


def _make_stream(args):
    import synth_lib as S
    seed, width, height, epf, texture, tree_shape = args
    # (JXL_BENCH_STREAM_CACHE=<dir>: keep the synthesised streams between runs of a parameter sweep on one box; unset in a plain run)
    cache = os.environ.get("JXL_BENCH_STREAM_CACHE")
    path = os.path.join(cache, f"s{seed}_{width}x{height}_e{epf}_t{texture}_m{tree_shape}.jxl") if cache else None
    if path and os.path.exists(path):
        return open(path, "rb").read()
    img = S.synthetic_image(seed, width, height)
    if texture:
        img = _textured(img, texture, seed)
    S.set_lf_tree_shape(tree_shape)
    try:
        data = S.encode_vardct(img, seed=seed, distance=1.0, epf_iters=epf, gab=1, strategy_mix=1)
    finally:
        S.set_lf_tree_shape(0)
    if path:
        os.makedirs(cache, exist_ok=True)
        with open(path + f".{os.getpid()}", "wb") as fh:
            fh.write(data)
        os.replace(path + f".{os.getpid()}", path)
    return data


def make_streams(distinct, width, height, epf, seed0=1000, texture=0.0, tree_shape=0):
    """Seeded synthetic frames encoded by tools/jxlsynth, one per seed (different content, different
    varblock maps and token counts).  tree_shape 1: the MA tree of the LF-group streams has the shape a default-effort cjxl encode writes
    (weighted predictor; tests/synth_lib.py set_lf_tree_shape).  Generated on the host cores in parallel (4.6 s per 4K frame).  Returns list of bytes."""
    import multiprocessing as mp
    jobs = [(seed0 + i, width, height, epf, texture, tree_shape) for i in range(distinct)]
    workers = max(1, min(len(jobs), int(os.environ.get("JXL_BENCH_SYNTH_WORKERS", "0")) or (os.cpu_count() or 1), 64))
    if workers == 1:
        return [_make_stream(j) for j in jobs]
    with mp.get_context("fork").Pool(workers) as pool:
        return pool.map(_make_stream, jobs)


def _oracle_decode_worker(data):
    import oracle_lib as O
    t = time.time()
    d = O.decode(data)
    px = d.pixels("u8", 3)
    return time.time() - t, int(px[:: 4097].sum())


def libjxl_baseline(found, streams, width, height, target_seconds=15.0):
    """Real libjxl (if the box has one, tests/libjxl_probe.py): the decode loop of jpegxl-rs/benches/decode.rs:16-37, one process
    per core, in subprocesses (the soname collides with this repository's look-alike)."""
    import concurrent.futures as cf
    import libjxl_probe as P
    cores = max(1, min(os.cpu_count() or 1, 64))
    t0 = time.time()
    if P.decode(found, streams[0], "u8", 3) is None:
        return None
    dt = time.time() - t0
    per_core = max(1, min(4, int(target_seconds / max(dt, 1e-3))))
    jobs = [streams[i % len(streams)] for i in range(cores * per_core)]
    t0 = time.time()
    with cf.ThreadPoolExecutor(cores) as ex:
        list(ex.map(lambda d: P.decode(found, d, "u8", 3) is not None, jobs))
    wall = time.time() - t0
    return {"value": round(len(jobs) * width * height / 1e6 / wall, 3), "unit": "Mpixel/s", "cores": cores, "kind": "libjxl",
            "sample": f"{len(jobs)} decodes of {width}x{height} VarDCT d1 frames by {found.get('lib') or found.get('djxl')} (version {found.get('lib_version')}), "
                      f"one subprocess per decode incl. start-up, {cores} at a time"}


def cpu_baseline(streams, width, height, target_seconds=15.0):
    """The CPU oracle (a scalar port of the libjxl algorithm: kind "port") timed on the host cores: one process per
    core, each decoding whole frames of the same workload.  Bounded sample (~10-30 s of CPU work per core).  A real libjxl
    found by the run-time probe takes precedence (kind "libjxl")."""
    import multiprocessing as mp
    try:
        import libjxl_probe as P
        found = P.probe()
        if found["available"]:
            r = libjxl_baseline(found, streams, width, height, target_seconds)
            if r is not None:
                return r
    except Exception:
        pass
    cores = max(1, min(os.cpu_count() or 1, 64))
    t0 = time.time()
    dt, _ = _oracle_decode_worker(streams[0])           # calibrate
    per_core = max(1, min(4, int(target_seconds / max(dt, 1e-3))))
    jobs = [streams[i % len(streams)] for i in range(cores * per_core)]
    t0 = time.time()
    with mp.get_context("fork").Pool(cores) as pool:
        pool.map(_oracle_decode_worker, jobs)
    wall = time.time() - t0
    mpx = len(jobs) * width * height / 1e6
    return {"value": round(mpx / wall, 3), "unit": "Mpixel/s", "cores": cores, "kind": "port",
            "sample": f"{len(jobs)} decodes of {width}x{height} VarDCT d1 frames by oracle/libjxl_oracle.so ({per_core} per core, {cores} processes), "
                      f"single-core rate {width * height / 1e6 / dt:.2f} Mpixel/s"}


def extras(jx, torch, streams, W, H, device):
    """What a caller of the drop-in API sees, measured after the timed region (N = 1): see the module docstring."""
    import numpy as np
    out = {}
    dec = jx.decoder_builder()
    ts = []
    for i in range(6):
        t0 = time.perf_counter()
        dec.decode_with(streams[i % len(streams)], np.uint8)
        ts.append((time.perf_counter() - t0) * 1e3)
    out["single_frame_ms"] = {"value": round(sorted(ts[1:])[len(ts[1:]) // 2], 3), "what": f"one {W}x{H} frame through decoder_builder().decode_with(u8): host bytes in, host pixels out "
                              "(parse, device allocation, upload, decode, copy back); median of 5 after one warm-up", "mpixel_per_s": round(W * H / 1e6 / (sorted(ts[1:])[2] * 1e-3), 1)}
    # BASELINE config 3 per-GPU shape: 128 frames, one pass, nothing resident beforehand, no pipelining
    n = 128
    torch.cuda.synchronize()
    dst = torch.empty((n, H, W, 3), dtype=torch.uint8, device=torch.device("cuda", device))
    host = torch.empty((n, H, W, 3), dtype=torch.uint8, pin_memory=True)
    stream = torch.cuda.current_stream().cuda_stream
    t0 = time.perf_counter()
    b = jx.BatchDecoder(device)
    b.add_many([streams[i % len(streams)] for i in range(n)], "uint8", 3, device_ptrs=[dst.data_ptr() + i * W * H * 3 for i in range(n)], threads=8)
    b.set_lane_stride(64, 1)
    b.prepare(stream)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    b.decode(stream)
    b.finish(stream)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.copy_(dst, non_blocking=True)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    mpx = n * W * H / 1e6
    out["one_pass_128"] = {"decode_ms": round((t2 - t1) * 1e3, 2), "prepare_ms": round((t1 - t0) * 1e3, 2), "mpixel_per_s_decode_only": round(mpx / (t2 - t1), 1),
                           "mpixel_per_s_with_prepare": round(mpx / (t2 - t0), 1),
                           "what": "fresh batch of 128 frames (BASELINE config 3 per-GPU share), decoded once: prepare = host parse of headers / TOC / entropy tables + device "
                                   "allocation + upload of the compressed streams; decode = every kernel, no overlap between batches"}
    out["pcie_inclusive"] = {"mpixel_per_s": round(mpx / (t3 - t0), 1), "d2h_ms": round((t3 - t2) * 1e3, 2), "d2h_gbs": round(n * W * H * 3 / 1e9 / (t3 - t2), 1),
                             "what": "the same pass plus the copy of the decoded pixels to pinned host memory (compressed input up, 24.9 MB per frame down)"}
    del b
    return out


def _cu_mask(spec, ncu=256):
    """CU mask of an experiment stream: 'first:N' (mask bits 0 .. N-1; the driver deals mask bits round-robin over the XCDs, so this is N / 8
    CUs of every XCD), 'last:N', 'stride:K' (every K-th bit), 'not-first:N', or comma-separated 32-bit hex words.  '' = no mask."""
    if not spec:
        return None
    kind, _, val = spec.partition(":")
    if kind == "first":
        bits = [i < int(val) for i in range(ncu)]
    elif kind == "last":
        bits = [i >= ncu - int(val) for i in range(ncu)]
    elif kind == "not-first":
        bits = [i >= int(val) for i in range(ncu)]
    elif kind == "stride":
        bits = [i % int(val) == 0 for i in range(ncu)]
    else:
        return [int(w, 16) for w in spec.split(",")]
    return [sum(1 << b for b in range(32) if bits[w * 32 + b]) for w in range(ncu // 32)]


def _masked_stream(torch, dev, mask):
    """hipExtStreamCreateWithCUMask through the HIP runtime torch has loaded, wrapped as a torch stream (lives as long as the process)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    words = (ctypes.c_uint32 * len(mask))(*mask)
    st = ctypes.c_void_p()
    with torch.cuda.device(dev):
        err = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(len(mask)), words)
    if err != 0 or not st.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {err}")
    return torch.cuda.ExternalStream(st.value, device=dev)


class Pipeline:
    """The decode pipeline of one GPU.  A decode is LF (entropy decode of the LF groups: a serial chain per stream, ~230 ms per
    launch whatever the batch size, a few dozen wavefronts in the SIMT form) -> varblock placement + LF post-processing -> HF (entropy decode of
    the coefficients, ~40 ms, one sparse workgroup per frame) -> IDCT -> filters + write (the HBM-bound part).  Throughput comes from batches in
    flight: step k runs the tail of batch k on the main stream, the HF stage of batch k + 1 on a stream of its own ("deep"), the LF stages of
    batches k + 1 .. k + ahead on side streams and — in streaming mode — parses, prepares and uploads the batches after that on host threads.
    A batch object (its LF outputs: 14 MB per 4K frame) is busy from its preparation to its tail; the coefficient planes (106 MB per frame)
    exist once per HF stage in flight + 1, the pixel planes once, the outputs as often as --out-buffers says.

    streaming: every step decodes a batch of compressed frames that has not been seen before: the batch object is reset, its frames are
    parsed (JxlHipBatchAddImages, --parse-threads host threads), the tables built and uploaded from pinned memory (JxlHipBatchPrepare, on a copy
    stream) by one of --prepare-threads worker threads, overlapped with the GPU's work on earlier batches.
    resident: the batch objects are prepared once before the timed region and decoded again and again (inputs resident in HBM)."""

    _streams = {}

    def __init__(self, args, jx, torch, dist, streams, dev, local_rank, rank, world, B, inner, streaming, consumer="gather", depth=None):
        self.args, self.jx, self.torch, self.dist, self.streams = args, jx, torch, dist, streams
        # who consumes the decoded pixels: "gather" = rank 0 (RCCL point-to-point gather, north_star's mode); "per_rank" = every rank its own frames, in place
        # in HBM (a reduction over the pixels stands in for the consumer; the ranks only exchange the 8-byte checksums at the end of the run)
        self.consumer = consumer
        self.dev, self.local_rank, self.rank, self.world, self.B, self.inner, self.streaming = dev, local_rank, rank, world, B, inner, streaming
        self.stream_texture, self.stream_tree_shape = args.main_texture, args.main_tree_shape    # how `streams` were made (verify() regenerates other ranks' frames)
        W, H = args.width, args.height
        self.frame_bytes = W * H * 3
        self.pipeline = not args.no_pipeline
        self.deep = self.pipeline and os.environ.get("JXL_BENCH_DEEP", "1") == "1"
        # depth = (batches in flight, LF side streams): as many LF stages in flight as it takes to cover one LF launch with steps (weighted-predictor LF streams
        # take ~2x the time per launch of gradient-tree ones: their pipeline is deeper)
        in_flight, lf_streams = depth if depth else (args.in_flight, args.lf_streams)
        nbuf = int(os.environ.get("JXL_BENCH_NBUF", str(in_flight))) if self.pipeline else 1
        self.nhf = max(1, args.hf_streams) if self.deep else 0        # HF stages in flight beside the tail of the step (each on its own stream)
        self.ncoef = self.nhf + 1                                      # coefficient sets: one per HF stage in flight + the one the tail is consuming
        self.ahead = nbuf - 1 if self.pipeline else 0                  # LF stages issued ahead of the step being finished
        self.prep_ahead = self.ahead + 2 if streaming else 0           # streaming: batches whose preparation has been handed to the host threads
        if streaming:
            nbuf = self.prep_ahead + 1
        if self.deep and nbuf % self.ncoef:
            nbuf += self.ncoef - nbuf % self.ncoef                     # (coefficient sets rotate with k: batch object k % nbuf must always meet set k % ncoef)
        self.nbuf = nbuf
        # tails in flight: 1 = IDCT and filters of consecutive batches one after the other on the main stream; 2 = the filter stage on a stream of
        # its own beside the IDCT of the next batch (two sets of pixel planes)
        self.ntail = max(1, min(2, args.tail_streams)) if self.deep else 1
        if self.ntail > 1 and nbuf % (self.ncoef * self.ntail):
            nbuf += self.ncoef * self.ntail - nbuf % (self.ncoef * self.ntail)
            self.nbuf = nbuf
        self.nout = min(nbuf, max(1, args.out_buffers))
        if os.environ.get("JXL_BENCH_CUMASK_MAIN") and "main" not in Pipeline._streams:          # experiment: the tail's kernels on a CU-masked stream
            Pipeline._streams["main"] = _masked_stream(torch, dev, _cu_mask(os.environ["JXL_BENCH_CUMASK_MAIN"]))
            torch.cuda.set_stream(Pipeline._streams["main"])
        self.main = torch.cuda.current_stream()
        self.stream = self.main.cuda_stream
        self.outs = [torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(self.nout)]
        self.batches = []
        for b in range(nbuf):
            bt = jx.BatchDecoder(local_rank)
            self.fill(bt, b)
            if b >= self.ntail:
                bt.share_buffers(self.batches[b % self.ntail])     # one set of pixel planes per tail in flight (1: the tails run one after the other on the main stream)
            if b >= self.ncoef:
                bt.share_coefficients(self.batches[b % self.ncoef])
            bt.prepare(self.stream)
            self.batches.append(bt)
        self.do_gather = world > 1 and not args.no_gather and consumer == "gather"
        self.consume_local = consumer == "per_rank"
        self.checksum = torch.zeros((), dtype=torch.int64, device=dev)
        self.gathered = torch.empty((world, inner * B, H, W, 3), dtype=torch.uint8, device=dev) if self.do_gather and rank == 0 else None
        E = torch.cuda.Event
        # (HIP streams are made once per process and handed to every pipeline: the runtime spreads streams over GPU_MAX_HW_QUEUES hardware
        # queues, and kernels of two streams that share a queue serialise — an LF stage in front of a tail kernel stalls the step)
        def S(kind, i, priority=0):
            key = (kind, i)
            if key not in Pipeline._streams:
                mask = _cu_mask(os.environ.get("JXL_BENCH_CUMASK_" + kind.upper(), ""))    # experiment: confine the kernels of one kind of stream to a set of CUs
                Pipeline._streams[key] = _masked_stream(torch, dev, mask) if mask else torch.cuda.Stream(device=dev, priority=priority)
            return Pipeline._streams[key]
        lf_prio = (lambda i: -1 if i == 0 else 0) if os.environ.get("JXL_BENCH_LF_PRIO") == "first" else (lambda i: -1)   # experiment: only the stream of the first cold LF stage is a high-priority one
        self.sides = [S("lf", i, lf_prio(i)) for i in range(max(1, min(self.ahead, lf_streams)))] if self.pipeline else []
        self.comm = S("comm", 0) if (self.do_gather or self.consume_local) else None            # RCCL gather / local consumer overlaps the next step's decode
        self.hf_streams = [S("hf", i, -1) for i in range(self.nhf)]
        self.filter_stream = S("filter", 0) if self.ntail > 1 else None
        self.copy_streams = [S("copy", i) for i in range(max(1, args.prepare_threads))] if streaming else []
        self.hf_done, self.front_done, self.lf_done, self.rest_done, self.idct_done = ([E() for _ in range(nbuf)] for _ in range(5))
        self.out_free = [E() for _ in range(self.nout)]               # the gather of the step that used this output buffer last has read it
        self.pool = None
        if streaming:
            import concurrent.futures as cf
            self.pool = cf.ThreadPoolExecutor(max(1, args.prepare_threads))
        self.prepare_s = []

    def frames_of(self, k):
        """compressed frames of step k (streaming: a different rotation of the distinct frames every step; resident: of batch object k)"""
        n, B = len(self.streams), self.B
        off = (k * 37 if self.streaming else k * B) % n
        return [self.streams[(off + i) % n] for i in range(B)]

    def fill(self, bt, k):
        out = self.outs[k % self.nout]
        ptrs = [out.data_ptr() + i * self.frame_bytes for i in range(self.B)]
        bt.add_many(self.frames_of(k), "uint8", 3, device_ptrs=ptrs, threads=max(1, self.args.parse_threads))
        bt.set_lane_stride(self.args.lane_stride_lf, self.args.lane_stride_hf)
        if os.environ.get("JXL_BENCH_LDS_BUDGET"):
            bt.set_option("lds_code_budget", int(os.environ["JXL_BENCH_LDS_BUDGET"]))   # experiment: entropy-code tables of the HF stage through the L2

    def prepare_job(self, k, slot, timed=False):
        """host side of step k (a worker thread): wait until the batch object's previous decode has left the GPU, parse, build, upload"""
        b = k % self.nbuf
        if k >= self.nbuf:
            self.rest_done[b].synchronize()
        t0 = time.perf_counter()
        bt = self.batches[b]
        bt.reset()
        self.fill(bt, k)
        bt.prepare(self.copy_streams[slot % len(self.copy_streams)].cuda_stream)     # (returns when the upload has completed)
        self.prepare_s.append(time.perf_counter() - t0)
        if self.pipeline:
            self.issue_front(k, timed)          # the batch's LF stage goes out right away, from this thread: the earlier it starts the better

    def issue_front(self, k, timed):
        b = k % self.nbuf
        side = self.sides[k % len(self.sides)]
        torch = self.torch
        with torch.cuda.stream(side):
            if k >= self.nbuf and not self.streaming:
                side.wait_event(self.rest_done[b])                  # the batch object's previous decode is complete (streaming: its preparation waited)
            if k < self.args.wide_first and self.args.lane_stride_lf < 64:
                self.batches[b].set_option("lf_wide_once", 1)       # cold pipeline, idle GPU: the wide LF kernel (100 instead of 250 ms until step 0 can go on)
            self.batches[b].decode_part(5, side.cuda_stream, timed)  # LF decode + varblock placement: all the HF stage waits for
            self.lf_done[b].record(side)
            self.batches[b].decode_part(6, side.cuda_stream, timed)  # LF post-processing: needed by the IDCT only
            self.front_done[b].record(side)

    def issue_hf(self, k, timed):
        b = k % self.nbuf
        s_ = self.hf_streams[k % self.nhf] if self.deep else self.main
        with self.torch.cuda.stream(s_):
            s_.wait_event(self.lf_done[b])
            if self.deep and k >= self.ncoef:
                s_.wait_event(self.idct_done[(k - self.ncoef) % self.nbuf])   # the coefficient set's previous user has consumed (and zeroed) it
            self.batches[b].decode_part(3, s_.cuda_stream, timed)
            self.hf_done[b].record(s_)

    def step(self, k, timed, st):
        b = k % self.nbuf
        main, torch = self.main, self.torch
        if not self.pipeline:
            if self.streaming:
                self.prepare_job(k, 0, timed)
            (self.batches[b].decode_timed if timed else self.batches[b].decode)(self.stream)
            self.rest_done[b].record(main)
        else:
            if self.streaming:
                for j in range(0, self.prep_ahead + 1):
                    if k + j < st["limit"] and st["prep_submitted"] <= k + j:
                        st["futures"][k + j] = self.pool.submit(self.prepare_job, k + j, k + j, timed); st["prep_submitted"] = k + j + 1
                # (the worker that prepared a batch has enqueued its LF stage as well) step k's own batch: wait for it — normally long since on the GPU
                while st["front_issued"] < st["limit"] and st["front_issued"] <= k + self.ahead and (st["front_issued"] <= k or st["futures"][st["front_issued"]].done()):
                    st["futures"].pop(st["front_issued"]).result(); st["front_issued"] += 1
            for j in range(0, self.ahead + 1):
                if not self.streaming and k + j < st["limit"] and st["front_issued"] <= k + j:
                    self.issue_front(k + j, timed); st["front_issued"] = k + j + 1
            for j in range(0, self.nhf + 1 if self.deep else 1):
                if st["front_issued"] <= k + j:
                    break                                          # (its LF stage is not enqueued yet: the events it would wait for are a previous decode's)
                if k + j < st["limit"] and st["hf_issued"] <= k + j:
                    self.issue_hf(k + j, timed); st["hf_issued"] = k + j + 1
            if self.deep:
                main.wait_event(self.hf_done[b])
            main.wait_event(self.front_done[b])
            if (self.do_gather or self.consume_local) and st["gathers"] >= self.nout:
                main.wait_event(self.out_free[k % self.nout])   # the previous gather of this output buffer must have read the pixels
            if self.ntail > 1 and k >= self.ntail:
                main.wait_event(self.rest_done[(k - self.ntail) % self.nbuf])   # the plane set's previous user has written its pixels
            self.batches[b].decode_part(7, self.stream, timed)     # IDCT
            self.idct_done[b].record(main)
            if self.ntail > 1:
                fs = self.filter_stream
                with torch.cuda.stream(fs):
                    fs.wait_event(self.idct_done[b])
                    if (self.do_gather or self.consume_local) and st["gathers"] >= self.nout:
                        fs.wait_event(self.out_free[k % self.nout])
                    self.batches[b].decode_part(8, fs.cuda_stream, timed)
                    self.rest_done[b].record(fs)
            else:
                self.batches[b].decode_part(8, self.stream, timed)     # restoration filters, colour, write
                self.rest_done[b].record(main)
        if self.consume_local:
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(self.rest_done[b])
                self.checksum += self.outs[k % self.nout].view(-1).view(torch.int32).sum(dtype=torch.int64)     # every decoded byte is read once, where it was written
                self.out_free[k % self.nout].record(self.comm)
            st["gathers"] += 1
            if not self.pipeline:
                main.wait_event(self.out_free[k % self.nout])
        if self.do_gather:
            from jpegxl_rs_amd.sharding import gather_frames_chunked
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(self.rest_done[b])
                # per-chunk point-to-point transfers straight into their final place (all peers at once, one xGMI link each)
                j = k % self.inner
                gather_frames_chunked(self.outs[k % self.nout], self.gathered[:, j * self.B:(j + 1) * self.B] if self.rank == 0 else None, dst=0, chunk_frames=self.args.gather_chunk)
                self.out_free[k % self.nout].record(self.comm)
            st["gathers"] += 1
            if not self.pipeline:
                main.wait_event(self.out_free[k % self.nout])

    def run(self, nsteps, timed):
        """nsteps steps from an empty pipeline; returns (seconds, per-step end times in ms, seconds of the gather tail)"""
        torch, dist = self.torch, self.dist
        st = {"front_issued": 0, "hf_issued": 0, "prep_submitted": 0, "limit": nsteps, "gathers": 0, "futures": {}}
        self.prepare_s = []
        self.checksum.zero_()
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        cpu0 = time.process_time()
        t_start = torch.cuda.Event(enable_timing=True); t_start.record(self.main)
        marks = []
        t0 = time.perf_counter()
        for k in range(nsteps):
            self.step(k, timed, st)
            ev = torch.cuda.Event(enable_timing=True); ev.record(self.filter_stream if self.filter_stream is not None else self.main); marks.append(ev)
        self.main.synchronize()
        if self.filter_stream is not None:
            self.filter_stream.synchronize()
        t_decode = time.perf_counter() - t0          # every rank's own decode work is done (the gather may still be running)
        if self.consume_local and self.world > 1:
            with torch.cuda.stream(self.comm):
                dist.all_reduce(self.checksum)         # checksum of checksums: the only bytes that cross xGMI in this mode
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        self.cpu_s = time.process_time() - cpu0       # host CPU seconds of this rank's process over the run (all threads: parse, prepare, enqueue)
        return elapsed, [round(t_start.elapsed_time(e), 1) for e in marks], t_decode

    def verify(self, nsteps, O, np):
        """decoded frames of the output buffers the last steps wrote, against the CPU oracle"""
        ok, checked = True, []
        for oj in range(len(self.outs)):
            ks = [k for k in range(nsteps) if k % self.nout == oj]
            if not ks:
                continue
            frames = self.frames_of(ks[-1])
            for fi in sorted({0, self.B // 2, self.B - 1}):
                got = self.outs[oj][fi].cpu().numpy().reshape(-1)
                ok = ok and bool(np.array_equal(got, O.decode(frames[fi]).pixels("u8", 3)))
                checked.append(f"{ks[-1]}:{fi}")
        if self.gathered is not None and self.rank == 0:
            # what the consumer rank holds of the other ranks' shards after the last step's gather (their frames are regenerated here: seeds are per rank)
            self.torch.cuda.synchronize()
            k = nsteps - 1
            n, B = len(self.streams), self.B
            off = (k * 37 if self.streaming else k * B) % n
            for r in range(1, self.world):
                for fi in sorted({0, B - 1}):
                    data = _make_stream((1000 + 1000 * r + (off + fi) % n, self.args.width, self.args.height, self.args.epf, self.stream_texture, self.stream_tree_shape))
                    got = self.gathered[r][(k % self.inner) * B + fi].cpu().numpy().reshape(-1)
                    ok = ok and bool(np.array_equal(got, O.decode(data).pixels("u8", 3)))
                    checked.append(f"rank{r}:{k}:{fi}")
        return ok, checked

    def close(self):
        if self.pool:
            self.pool.shutdown(wait=True)
        self.batches.clear(); self.outs.clear(); self.gathered = None
        self.torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("JXL_BENCH_BATCH", "256")), help="frames per GPU per step")
    ap.add_argument("--distinct", type=int, default=int(os.environ.get("JXL_BENCH_DISTINCT", "0")), help="distinct synthetic frames per GPU (cycled to fill the batches); default 256, 64 per GPU when N > 1 "
                    "(the ranks of a node share its host cores for the synthesis: 4.6 s per frame)")
    ap.add_argument("--mode", choices=["streaming", "resident", "both"], default=os.environ.get("JXL_BENCH_MODE", "both"),
                    help="streaming (the headline): every step parses, prepares, uploads and decodes a fresh batch of compressed frames; resident: prepared "
                         "batches decoded again and again; both: streaming timed first, the resident figure reported beside it")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="weak: --batch frames per GPU per step; strong: --total-frames per step over all GPUs")
    ap.add_argument("--total-frames", type=int, default=1024, help="frames per step of the whole job with --scaling strong (BASELINE config 3)")
    ap.add_argument("--no-extras", action="store_true", help="skip single_frame_ms / one_pass / pcie_inclusive (N = 1 only anyway)")
    ap.add_argument("--no-verify", action="store_true", help="skip the comparison of decoded frames with the CPU oracle after the run")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--epf", type=int, default=1)
    ap.add_argument("--lane-stride-lf", type=int, default=int(os.environ.get("JXL_LANE_STRIDE_LF", "8")),
                    help="< 64: SIMT LF decode, 64 / value LF-group streams per wavefront; 64 = one stream per wavefront")
    ap.add_argument("--lane-stride-hf", type=int, default=int(os.environ.get("JXL_LANE_STRIDE_HF", "1")),
                    help="1 = SIMT HF decode (one group stream per lane), 64 = one stream per wavefront")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL gather of decoded pixels (N > 1)")
    ap.add_argument("--per-rank-consumers", action="store_true", help="N = 1: also run the per-rank-consumer leg that N > 1 runs beside the gather (a reduction over the decoded pixels per step)")
    ap.add_argument("--gather-chunk", type=int, default=32, help="frames per point-to-point transfer of the pixel gather (N > 1)")
    ap.add_argument("--no-pipeline", action="store_true", help="do not overlap the stages of different batches")
    ap.add_argument("--in-flight", type=int, default=11, help="batches in flight on the GPU (pipelined): LF stages run this many steps ahead, minus one")
    ap.add_argument("--lf-streams", type=int, default=7, help="side streams the LF stages of the batches ahead are spread over")
    ap.add_argument("--wp-in-flight", type=int, default=11, help="batches in flight for workloads whose LF streams use the weighted predictor (LF stage ~700 ms per launch instead of ~370)")
    ap.add_argument("--wp-lf-streams", type=int, default=7, help="LF side streams for those workloads")
    ap.add_argument("--hf-streams", type=int, default=int(os.environ.get("JXL_BENCH_HF_STREAMS", "1")), help="HF stages in flight beside the tail of the current step (deep pipeline), one stream and one coefficient set each")
    ap.add_argument("--tail-streams", type=int, default=int(os.environ.get("JXL_BENCH_TAIL_STREAMS", "1")), help="2: the filter stage of a batch on its own stream beside the IDCT of the next (two sets of pixel planes)")
    ap.add_argument("--wide-first", type=int, default=int(os.environ.get("JXL_BENCH_WIDE_FIRST", "4")), help="LF stages at the start of the (cold) pipeline that take the one-wavefront-per-stream kernel")
    ap.add_argument("--out-buffers", type=int, default=2, help="output buffer sets the batches in flight cycle through")
    ap.add_argument("--prepare-threads", type=int, default=int(os.environ.get("JXL_BENCH_PREPARE_THREADS", "3")), help="streaming: host threads that each parse + prepare + upload one batch at a time")
    ap.add_argument("--parse-threads", type=int, default=int(os.environ.get("JXL_BENCH_PARSE_THREADS", "8")), help="host threads JxlHipBatchAddImages parses the frames of one batch on")
    ap.add_argument("--texture", type=float, default=5.0, help="strength (sRGB levels) of the texture of the realistic-bit-rate workload (0: skip it)")
    ap.add_argument("--realistic-distinct", type=int, default=64, help="distinct frames of the realistic-bit-rate workload")
    ap.add_argument("--no-realistic", action="store_true", help="skip the second workload (textured frames, ~2 bpp)")
    ap.add_argument("--cjxl-distinct", type=int, default=32, help="distinct frames of the cjxl-shaped workload (textured frames whose LF-group streams use the MA-tree shape of a default-effort "
                    "cjxl encode: weighted predictor); 0: skip it")
    ap.add_argument("--main-tree-shape", type=int, default=0, help="experiments: LF tree shape of the headline workload's frames (1: the cjxl default-effort shape)")
    ap.add_argument("--main-texture", type=float, default=0.0, help="experiments: texture strength of the headline workload's frames")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    W, H = args.width, args.height
    if args.distinct <= 0:
        args.distinct = 256 if world == 1 else 64
    if world > 1:
        args.realistic_distinct = min(args.realistic_distinct, 16)
        os.environ.setdefault("JXL_BENCH_SYNTH_WORKERS", str(max(1, (os.cpu_count() or 1) // int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))))

    streams = make_streams(args.distinct, W, H, args.epf, seed0=1000 + 1000 * rank, texture=args.main_texture, tree_shape=args.main_tree_shape)
    # the same frames with photograph-like texture: ~2 bpp at distance 1 instead of 0.8 (second workload of the line, fewer distinct frames)
    realistic_streams = make_streams(min(args.distinct, args.realistic_distinct), W, H, args.epf, seed0=1000 + 1000 * rank, texture=args.texture) if args.texture > 0 and not args.no_realistic else None
    cjxl_streams = make_streams(min(args.distinct, args.cjxl_distinct), W, H, args.epf, seed0=1000 + 1000 * rank, texture=args.texture, tree_shape=1) if args.cjxl_distinct > 0 and not args.no_realistic else None
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(streams, W, H)   # before any GPU runtime is initialised in this process (fork safety)

    # the pipeline keeps ~15 HIP streams busy at once (main, HF, LF side streams, copy streams, gather); the runtime maps streams onto 4
    # hardware queues by default and kernels of streams that share a queue serialise
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import numpy as np
    import torch
    import torch.distributed as dist
    import jpegxl_rs_amd as jx
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the decode path is HIP-only (no CPU fallback)")
    # JXL_BENCH_SHARE_GPU=1 + JXL_BENCH_BACKEND=gloo: several ranks on one GPU — a functional check of the N > 1 control flow on a
    # one-GPU box (RCCL refuses two ranks per device); never the configuration a result is quoted on
    if os.environ.get("JXL_BENCH_SHARE_GPU") == "1":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        backend = os.environ.get("JXL_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    B = args.batch
    inner = 1                       # pipeline iterations per step
    if args.scaling == "strong":
        per_rank = max(1, args.total_frames // world)
        B = min(args.batch, per_rank)
        inner = max(1, per_rank // B)

    def measure(streaming, streams=streams, texture=args.main_texture, tree_shape=args.main_tree_shape, consumer="gather"):
        """one mode: W untimed warm-up steps, then exactly K timed steps from an empty pipeline, bracketed by barrier + synchronize"""
        depth = (args.wp_in_flight, args.wp_lf_streams) if tree_shape == 1 and not args.no_pipeline else None
        p = Pipeline(args, jx, torch, dist, streams, dev, local_rank, rank, world, B, inner, streaming, consumer, depth)
        p.stream_texture, p.stream_tree_shape = texture, tree_shape
        p.run(args.warmup * inner, False)
        for bt in p.batches:
            bt.finish(p.stream)
            bt.collect_times()
        elapsed, step_end, t_decode = p.run(args.steps * inner, True)
        for bt in p.batches:
            bt.finish(p.stream)
        cpu_s = p.cpu_s
        if world > 1:
            t = torch.tensor([elapsed, t_decode], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, t_decode = float(t[0].item()), float(t[1].item())
            c = torch.tensor([cpu_s], dtype=torch.float64, device=dev)
            dist.all_reduce(c, op=dist.ReduceOp.SUM)       # host CPU seconds of all ranks (they share one host)
            cpu_s = float(c[0].item())
        times, runs = {}, 0
        for bt in p.batches:
            t_, r_ = bt.collect_times()
            runs += r_
            for kk, vv in t_.items():
                times[kk] = times.get(kk, 0.0) + vv
        r = {"elapsed": elapsed, "t_decode": t_decode, "step_end": step_end, "stage_ms": {k[:-3]: v / max(runs, 1) for k, v in times.items() if k != "total_ms"},
             "stage_bytes": p.batches[0].stage_bytes, "device_bytes": sum(bt.device_bytes for bt in p.batches), "compressed": int(p.batches[0].compressed_bytes // B),
             "nbuf": p.nbuf, "prepare_s": list(p.prepare_s), "gather": bool(p.do_gather), "pipelined": bool(p.pipeline), "cpu_s": cpu_s, "checksum": int(p.checksum.item()),
             "nonzeros": p.batches[0].info_value("hf_nonzeros") // B,
             "lf_simt": [p.batches[0].info_value(k) for k in ("lf_simt_frames", "lf_legacy_frames", "lf_simt_wp")]}
        if not args.no_verify and rank == 0:
            import oracle_lib as O
            r["verified"], r["verified_frames"] = p.verify(args.steps * inner, O, np)
        p.close()
        del p
        torch.cuda.empty_cache()
        return r

    modes = ["streaming", "resident"] if args.mode == "both" else [args.mode]
    res = {m: measure(m == "streaming") for m in modes}
    head = res[modes[0]]
    realistic = None
    if args.texture > 0 and not args.no_realistic and realistic_streams:
        realistic = measure(modes[0] == "streaming", realistic_streams, args.texture, 0)
    cjxl = measure(modes[0] == "streaming", cjxl_streams, args.texture, 1) if cjxl_streams else None
    # N > 1: the same job with per-rank consumers beside the gather to rank 0 (7 peers x ~80 GB/s of pixels into one GPU's xGMI links bound the gather)
    local_leg = measure(modes[0] == "streaming", consumer="per_rank") if (world > 1 or args.per_rank_consumers) and not args.no_gather else None
    if rank == 0:
        total_px = world * B * inner * W * H * args.steps
        rate = lambda r: total_px / r["elapsed"] / 1e6
        value = rate(head)
        stage_ms, stage_bytes = head["stage_ms"], head["stage_bytes"]
        # dominant kernel = stage with the largest device time; roofline from its ALGORITHMIC bytes per launch
        dom = max(stage_ms, key=stage_ms.get)
        kernel_of = {"lf": "LfDecodeSimtKernel" if args.lane_stride_lf < 64 else "LfDecodeKernel", "lfpost": "LlfSigmaKernel",
                     "hf": "HfDecodeSimtKernel" if args.lane_stride_hf == 1 else "HfDecodeKernel",
                     "idct": "IdctTileKernel", "filter": "FusedGabEpf1OutKernel" if args.epf == 1 else "EpfKernel", "out": "OutputKernel"}
        achieved = stage_bytes[dom] / (stage_ms[dom] * 1e-3) / 1e9 if stage_ms[dom] > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:   # HBM bytes per frame of that kernel from the PMC passes (profiles/*_pmc_traffic.md) x frames per launch
                per_frame = json.load(open(pmc))["per_kernel"].get(kernel_of[dom])
                traffic = int(per_frame * B) if per_frame is not None else None
            except Exception:
                traffic = None
        e = head["step_end"]
        n_e = len(e)
        steady = round((e[n_e * 2 // 3] - e[n_e // 5]) / max(1, n_e * 2 // 3 - n_e // 5), 2) if n_e >= 10 else None
        result = {
            "metric": "Mpixel/s decode (4K VarDCT d1)", "value": round(value, 2), "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(head["elapsed"] / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": f"synthetic ({args.distinct} distinct seeded frames per GPU, tools/jxlsynth; own synthesiser, 0.8 bpp — real d1 photographs run 1.5-2.5 bpp: config.workload_realistic; the LF-group MA tree is the gradient tree the SIMT LF kernel is eligible for by "
                                                        f"construction — a default-effort cjxl encode writes weighted-predictor trees: config.workload_cjxl_shape; cpu_baseline kind 'port' is the scalar oracle, not libjxl)",
            "config": {"workload": f"batch of {B} x {W}x{H} VarDCT d1 frames per GPU per step (XYB, ANS, var-block DCT8..32 mix, gaborish, EPF {args.epf}), u8 RGB out; "
                                   + ("streaming: every step's compressed frames come from host memory and are parsed, prepared and uploaded inside the timed region, outputs stay in HBM"
                                      if modes[0] == "streaming" else "inputs and outputs resident in HBM"),
                       "mode": modes[0], "frames_per_gpu": B * inner, "frames_per_launch": B, "width": W, "height": H, "compressed_bytes_per_frame": head["compressed"],
                       "lane_stride_lf": args.lane_stride_lf, "lane_stride_hf": args.lane_stride_hf, "batches_in_flight": head["nbuf"],
                       "gather": head["gather"], "pipelined_steps": head["pipelined"], "parallelism": f"frame-sharded x{world}"},
            "roofline": {"bound": "hbm", "kernel": kernel_of[dom], "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "algorithmic_bytes_per_launch": stage_bytes[dom], "avg_launch_ms": round(stage_ms[dom], 4)},
            # the entropy stages above are serial chains (HBM fraction ~ 0 by construction); the stage that IS bound by HBM is the IDCT:
            # the same figures for it (stage = IdctTileKernel + IdctRareSpecialKernel, measured while the other stages of the pipeline run beside it)
            "roofline_hbm_stage": (lambda ms, by, pf: {"bound": "hbm", "kernel": "IdctTileKernel (+ IdctRareSpecialKernel)", "achieved": round(by / (ms * 1e-3) / 1e9, 3) if ms > 0 else None,
                                                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else None,
                                                     "traffic": pf, "algorithmic_bytes_per_launch": by, "avg_launch_ms": round(ms, 4)})(
                stage_ms.get("idct", 0.0), stage_bytes.get("idct", 0), _pmc_bytes(("IdctTileKernel<4, true>", "IdctRareSpecialKernel"), B)),
            # what the step as a whole runs into is neither HBM nor MFMA but vector-ALU issue (profiles/r03_notes.md): VALU wavefront-instructions of all kernels of a
            # step (SQ_INSTS_VALU, profiles/sq_valu.json) against 256 CUs x 4 SIMDs issuing one wave64 FP32 instruction per 4 cycles at 2.4 GHz
            "valu_issue": _valu_issue(B, head["elapsed"] / args.steps, steady),
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            # when the tail of every timed step had completed (ms after the start of the timed region): pipeline fill, then the steady state
            "step_end_ms": e, "steady_state_ms_per_step": steady,
            "stage_gbs": {k: round(stage_bytes[k] / (stage_ms[k] * 1e-3) / 1e9, 2) if stage_ms[k] > 0 else None for k in stage_ms},
            "device_bytes": head["device_bytes"],
        }
        if modes[0] == "streaming":
            ps = head["prepare_s"]
            result["streaming"] = {"prepare_threads": args.prepare_threads, "parse_threads_per_batch": args.parse_threads, "distinct_frames": args.distinct,
                                   "prepare_ms_per_batch": round(1e3 * sum(ps) / max(1, len(ps)), 2), "prepare_ms_per_frame": round(1e3 * sum(ps) / max(1, len(ps)) / B, 4),
                                   "what": "prepare = JxlHipBatchReset + JxlHipBatchAddImages (parse on the parse threads) + JxlHipBatchPrepare (tables, pinned staging, upload on a copy stream), wall time of one worker thread per batch"}
        if "resident" in res and modes[0] != "resident":
            rr = res["resident"]
            result["resident_mpixel_per_s"] = round(rate(rr), 2)
            result["resident"] = {"ms_per_step": round(rr["elapsed"] / args.steps * 1e3, 3), "stage_ms": {k: round(v, 4) for k, v in rr["stage_ms"].items()}, "step_end_ms": rr["step_end"],
                                  "verified_vs_oracle": rr.get("verified"), "streaming_over_resident": round(rate(head) / rate(rr), 4)}
        result["config"]["nonzero_coefficients_per_frame"] = head["nonzeros"]
        result["config"]["bits_per_pixel"] = round(head["compressed"] * 8 / (W * H), 3)
        if realistic is not None:
            re_ = realistic["step_end"]; n_r = len(re_)
            result["config"]["workload_realistic"] = {
                "what": f"the same pipeline and mode on frames with photograph-like texture (bench.py _textured, {args.texture:g} sRGB levels): the bit rate of real cjxl -d 1 photographs",
                "value": round(rate(realistic), 2), "unit": "Mpixel/s", "ms_per_step": round(realistic["elapsed"] / args.steps * 1e3, 3),
                "steady_state_ms_per_step": round((re_[n_r * 2 // 3] - re_[n_r // 5]) / max(1, n_r * 2 // 3 - n_r // 5), 2) if n_r >= 10 else None,
                "stage_ms": {k: round(v, 4) for k, v in realistic["stage_ms"].items()}, "compressed_bytes_per_frame": realistic["compressed"],
                "bits_per_pixel": round(realistic["compressed"] * 8 / (W * H), 3), "nonzero_coefficients_per_frame": realistic["nonzeros"],
                "distinct_frames": len(realistic_streams), "verified_vs_oracle": realistic.get("verified")}
        if cjxl is not None:
            ce_ = cjxl["step_end"]; n_c = len(ce_)
            result["config"]["workload_cjxl_shape"] = {
                "what": "the realistic-bit-rate frames with LF-group streams under the MA-tree shape of a default-effort cjxl encode (enc_modular.cc tree kinds 'WP fixed DC' + 'AC meta': "
                        "weighted-predictor leaves under a fixed tree over property 15 for the LF coefficients; row / N / W splits for the HF metadata) — the headline's and the realistic "
                        "workload's LF trees are the gradient tree `cjxl --faster_decoding` picks, which the plain SIMT LF kernel was built around",
                "value": round(rate(cjxl), 2), "unit": "Mpixel/s", "ms_per_step": round(cjxl["elapsed"] / args.steps * 1e3, 3),
                "steady_state_ms_per_step": round((ce_[n_c * 2 // 3] - ce_[n_c // 5]) / max(1, n_c * 2 // 3 - n_c // 5), 2) if n_c >= 10 else None,
                "stage_ms": {k: round(v, 4) for k, v in cjxl["stage_ms"].items()}, "compressed_bytes_per_frame": cjxl["compressed"],
                "bits_per_pixel": round(cjxl["compressed"] * 8 / (W * H), 3), "lf_simt_frames": cjxl["lf_simt"][0], "lf_legacy_frames": cjxl["lf_simt"][1],
                "lf_simt_weighted_predictor_kernel": bool(cjxl["lf_simt"][2]), "distinct_frames": len(cjxl_streams), "verified_vs_oracle": cjxl.get("verified"),
                "batches_in_flight": cjxl["nbuf"], "lf_streams": args.wp_lf_streams}
        frames_total = world * B * inner * args.steps
        result["host_cpu"] = {"cpu_s_per_frame": round(head["cpu_s"] / frames_total, 6), "cores_busy": round(head["cpu_s"] / head["elapsed"], 2),
                              "what": "process CPU time of all ranks over the timed steps (parse + prepare + upload + enqueue threads) per decoded frame; cores_busy = CPU seconds per wall second: "
                                      "what one host has to supply for this rate (the ranks of a node share its cores)"}
        if local_leg is not None:
            result["per_rank_consumers"] = {
                "what": "the same job with every rank consuming its own frames where they were decoded (a reduction over the pixels on a side stream stands in for the consumer; the ranks exchange "
                        "only the 8-byte checksums at the end): the rate a sharded consumer sees, beside `value` = everything gathered into rank 0 over xGMI",
                "value": round(rate(local_leg), 2), "unit": "Mpixel/s", "ms_per_step": round(local_leg["elapsed"] / args.steps * 1e3, 3), "checksum_of_checksums": local_leg["checksum"],
                "verified_vs_oracle": local_leg.get("verified"), "host_cpu_s_per_frame": round(local_leg["cpu_s"] / frames_total, 6)}
        if world > 1:
            result["decode_only_mpixel_per_s"] = round(total_px / head["t_decode"] / 1e6, 2)     # until every rank's own decode work was done
            result["gather_ms"] = round((head["elapsed"] - head["t_decode"]) * 1e3, 2)             # what the pixel gather added after that
        if world == 1:
            # practical HBM ceiling next to the spec peak: a device-to-device copy of 4 GiB, read + write counted
            a = torch.empty(1 << 30, dtype=torch.int32, device=dev); bcopy = torch.empty_like(a)
            bcopy.copy_(a); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                bcopy.copy_(a)
            e1.record(); torch.cuda.synchronize()
            result["roofline"]["measured_copy_gbs"] = round(5 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
            del a, bcopy
        if cpu is not None:
            result["cpu_baseline"] = cpu
        if "verified" in head:
            result["verified_vs_oracle"] = head["verified"]
            result["verified_frames"] = head["verified_frames"]
        if world == 1 and not args.no_extras:
            torch.cuda.empty_cache()
            result.update(extras(jx, torch, streams, W, H, local_rank))
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
