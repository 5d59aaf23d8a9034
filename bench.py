#!/usr/bin/env python3
"""bench.py — Mpixel/s of JPEG XL VarDCT (d1) decode of 3840x2160 frames on MI355X (BASELINE.json metric).

A "step" decodes one job of B fresh synthetic 4K VarDCT frames per GPU (u8 RGB out) through the library's streaming pipeline (include/jxl_hip.h
JxlHipPipeline*: the ring of batch objects, the LF / HF / tail streams, the prepare threads and the coefficient-set rotation live in csrc/pipeline.cc —
this file only submits jobs and waits for them).  Every step's compressed frames come from host memory and are parsed, prepared and uploaded inside
the timed region; decoded pixels stay in HBM (a torch tensor).  `workload_realistic` repeats that on frames with photograph-like texture
(1.5 - 2.5 bits per pixel), `workload_cjxl_shape` on frames whose LF-group MA trees have the shape a default-effort cjxl writes (weighted predictor).
With N > 1 ranks every rank decodes its own shard (weak scaling, no data-path collective) and the decoded pixels are gathered to rank 0 over RCCL
(BASELINE.json north_star); `decode_only_mpixel_per_s` / `gather_ms` split the two; `per_rank_consumers` is the same job consumed where it was decoded.
--scaling strong --total-frames T fixes the job instead (BASELINE config 3: 1024 frames over the node).

Besides the headline the N = 1 line reports what callers of the drop-in API see — api_concurrent: the reference crate's call sequence against the libjxl C ABI
from 1 / 8 / 64 host threads (tools/api_concurrent.cc, host bytes in, host pixels out; the shared per-device scheduler coalesces the callers); single_frame_ms:
one 4K frame through decode_with (gradient LF tree, cjxl-shaped tree) and the reference's own criterion input bench.jxl; streaming_host_out: the pipeline with
pinned host destinations beside the measured PCIe ceiling; one_pass_128 / pcie_inclusive: a cold batch of 128 frames; the 8K workloads of BASELINE configs 4 and 5
(f32 HDR EPF 3; lossless Modular Squeeze u16) with the rooflines of their own kernels; step_end_ms / steady_state_ms_per_step; verified_vs_oracle.

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
    """VALU issue rate of a step: wavefront-instructions per second over the chip's measured peak (tools/microbench/valu_issue.hip: 8.96e11 per second, one wave64 FP32 instruction per 2.7 cycles of the nominal 2.4 GHz per SIMD) — a lower bound of
    how busy the vector ALUs are (transcendental and 64-bit operations take longer than 4 cycles).  Counts from the SQ counter passes of the round (profiles/sq_valu.json)."""
    try:
        per = json.load(open(os.path.join(ROOT, "profiles", "sq_valu.json")))["per_kernel"]
        instr = sum(v["valu_wave_instr_per_frame"] for v in per.values() if v["calls"] == 2) * frames
        peak = 8.96e11          # measured: tools/microbench/valu_issue.hip, 8 wavefronts per SIMD of independent v_fma_f32 (profiles/r05_notes.md; rounds 3-4 assumed 6.14e11 = 4 cycles per instruction)
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




def _make_8k_hdr(seed):
    """BASELINE config 5: 7680x4320 f32 HDR VarDCT (linear light, values up to 4.0, intensity target 1000), gaborish + EPF 3."""
    import numpy as np
    import synth_lib as S
    lin = ((S.synthetic_image(seed, 7680, 4320).astype(np.float32) / 255.0) ** 2.2) * 4.0
    return S.encode_vardct(lin, seed=seed, strategy_mix=1, epf_iters=3, gab=1, out_bits=32, hdr=1)


def _make_8k_modular(seed):
    """BASELINE config 4: lossless Modular 8192x8192 u16 (one channel) under the default Squeeze chain."""
    import numpy as np
    import synth_lib as S
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:8192, 0:8192].astype(np.float32)
    base = ((np.sin(xx / (37.0 + seed % 5)) + np.cos(yy / 23.0)) * 0.25 + 0.5) * 65535
    img = np.clip(base[..., None] + rng.normal(0, 64.0, (8192, 8192, 1)).astype(np.float32), 0, 65535).astype(np.int32)
    return S.encode_modular(img, 16, False, 1)


def _make_ycbcr420(seed):
    """a 3840x2160 frame shaped like a lossless JPEG transcode: YCbCr, 4:2:0 chroma subsampling, 8x8 DCT only, no gaborish / EPF (tools/synth_ycbcr.h) — the frames `cjxl photo.jpg` writes,
    minus the jbrd box"""
    import synth_lib as S
    return S.encode_ycbcr(S.synthetic_image(seed, 3840, 2160), "420", seed=seed)


def _pool_map(fn, jobs):
    import multiprocessing as mp
    workers = max(1, min(len(jobs), int(os.environ.get("JXL_BENCH_SYNTH_WORKERS", "0")) or (os.cpu_count() or 1), 64))
    if workers == 1 or len(jobs) <= 1:
        return [fn(j) for j in jobs]
    with mp.get_context("fork").Pool(workers) as pool:
        return pool.map(fn, jobs)


class Run:
    """One measured workload: a library pipeline (jx.Pipeline = JxlHipPipeline*), a ring of output tensors, the job loop of one rank.
    consumer: "none" (pixels stay where they were decoded), "gather" (RCCL point-to-point gather to rank 0, north_star's mode), "per_rank" (a reduction over the decoded
    pixels on a side stream stands in for a consumer on every rank; the ranks exchange the 8-byte checksums at the end)."""

    def __init__(self, args, jx, torch, dist, streams, dev, local_rank, rank, world, B, inner, W, H, dtype="uint8", nch=3, consumer="none", in_flight=None, lf_streams=None, host_out=False, plane_sets=1):
        import numpy as np
        self.args, self.jx, self.torch, self.dist, self.streams = args, jx, torch, dist, streams
        self.dev, self.rank, self.world, self.B, self.inner, self.W, self.H, self.dtype, self.nch = dev, rank, world, B, inner, W, H, dtype, nch
        self.consumer = consumer if (world > 1 or consumer == "per_rank") else "none"
        if consumer == "gather" and (world == 1 or args.no_gather):
            self.consumer = "none"
        self.frame_bytes = W * H * nch * np.dtype(dtype).itemsize
        self.host_out = host_out
        if self.consumer != "none" and not in_flight:
            in_flight = min(args.in_flight, 7)    # (a consumer needs one output buffer per batch object + 1 — 6.4 GB each at 256 frames —, beside rank 0's job buffer)
        self.p = jx.Pipeline(local_rank, timed=1, jobs_in_flight=in_flight or args.in_flight, lf_streams=lf_streams or args.lf_streams, hf_streams=args.hf_streams,
                             prepare_threads=args.prepare_threads, parse_threads=args.parse_threads, lane_stride_lf=args.lane_stride_lf, lane_stride_hf=args.lane_stride_hf,
                             wide_first=args.wide_first, reserve_frames=B, reserve_width=W, reserve_height=H, reserve_plane_sets=plane_sets)
        self.slots = self.p.info("slots")
        # output buffers: two when nobody reads them; with a consumer (gather / reduction on a side stream) one more than the jobs the pipeline can hold, so that a job never
        # writes where the consumer of an earlier one may still be reading
        self.nout = 2 if self.consumer == "none" and not host_out else self.slots + 1
        tdt = {"uint8": torch.uint8, "uint16": torch.uint16 if hasattr(torch, "uint16") else torch.int16, "float32": torch.float32}[np.dtype(dtype).name]
        if host_out:
            self.pinned = [jx.PinnedBuffer(B * self.frame_bytes) for _ in range(self.nout)]
            self.outs = None
        else:
            self.outs = [torch.empty((B, H, W, nch), dtype=tdt, device=dev) for _ in range(self.nout)]
        self.comm = torch.cuda.Stream(device=dev) if self.consumer != "none" else None
        self.out_free = [torch.cuda.Event() for _ in range(self.nout)]
        self.checksum = torch.zeros((), dtype=torch.int64, device=dev)
        self.gathered = torch.empty((world, inner * B, H, W, nch), dtype=tdt, device=dev) if self.consumer == "gather" and rank == 0 else None
        self.ccomm = None
        if self.consumer == "gather" and args.gather_impl == "c" and inner == 1:
            # the gather behind the C ABI (csrc/gather.cc, RCCL loaded by the library): rank 0's communicator id travels through the process group that exists anyway
            import ctypes as C
            L = jx.libjxl()
            uid = (C.c_uint8 * 128)()
            box = [bytes(uid)]
            if rank == 0:
                if L.JxlHipCommGetUniqueId(uid) != 0:
                    raise RuntimeError(jx.last_error())
                box = [bytes(uid)]
            dist.broadcast_object_list(box, src=0)
            uid = (C.c_uint8 * 128).from_buffer_copy(box[0])
            self.ccomm = L.JxlHipCommCreate(local_rank, rank, world, uid)
            if not self.ccomm:
                raise RuntimeError(jx.last_error())
        self.lag = max(0, self.slots - 2)                    # jobs between a submit and the wait for an earlier one (the pipeline's own back-pressure is slots - 1)
        self.step_offset = 0

    def frames_of(self, k):
        n = len(self.streams)
        off = (k * 37) % n                                   # a different rotation of the distinct frames every step
        return [self.streams[(off + i) % n] for i in range(self.B)]

    def _dest(self, k):
        if self.host_out:
            base = self.pinned[k % self.nout].ptr
            return dict(host_ptrs=[base + i * self.frame_bytes for i in range(self.B)])
        base = self.outs[k % self.nout].data_ptr()
        return dict(device_ptrs=[base + i * self.frame_bytes for i in range(self.B)])

    def _consume(self, k):
        """job k's pixels are in outs[k % nout]: hand them to the consumer on the side stream"""
        torch = self.torch
        if self.consumer == "per_rank":
            with torch.cuda.stream(self.comm):
                self.checksum += self.outs[k % self.nout].view(-1).view(torch.int32).sum(dtype=torch.int64)     # every decoded byte is read once, where it was written
                self.out_free[k % self.nout].record(self.comm)
        elif self.consumer == "gather":
            from jpegxl_rs_amd.sharding import gather_frames_chunked
            if self.ccomm:
                with torch.cuda.stream(self.comm):
                    rc = self.jx.libjxl().JxlHipGatherFrames(self.ccomm, self.outs[k % self.nout].data_ptr(), self.frame_bytes, self.B, self.gathered.data_ptr() if self.rank == 0 else None, 0,
                                                             self.args.gather_chunk, self.comm.cuda_stream)
                    if rc != 0:
                        raise RuntimeError(self.jx.last_error())
                    self.out_free[k % self.nout].record(self.comm)
                return
            with torch.cuda.stream(self.comm):
                j = k % self.inner
                gather_frames_chunked(self.outs[k % self.nout], self.gathered[:, j * self.B:(j + 1) * self.B] if self.rank == 0 else None, dst=0, chunk_frames=self.args.gather_chunk)
                self.out_free[k % self.nout].record(self.comm)

    def run(self, njobs):
        """njobs jobs from an idle pipeline; returns (seconds, per-job end times in ms of the GPU clock, seconds until this rank's own decode work was done)"""
        torch, dist, p = self.torch, self.dist, self.p
        self.checksum.zero_()
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        p.reset_clock()
        cpu0 = time.process_time()
        tickets, ends = [], []
        t0 = time.perf_counter()
        for k in range(njobs):
            if self.consumer != "none" and k >= self.nout:
                self.out_free[k % self.nout].synchronize()       # the consumer of the job that used this buffer last has read it (long since)
            tickets.append(p.submit(self.frames_of(self.step_offset + k), self.dtype, self.nch, **self._dest(k)))
            if k >= self.lag:
                ends.append(p.wait(tickets[k - self.lag])[1])
                self._consume(k - self.lag)
        for k in range(max(0, njobs - self.lag), njobs):
            ends.append(p.wait(tickets[k])[1])
            self._consume(k)
        t_decode = time.perf_counter() - t0
        if self.consumer == "per_rank" and self.world > 1:
            with torch.cuda.stream(self.comm):
                dist.all_reduce(self.checksum)                   # checksum of checksums: the only bytes that cross xGMI in this mode
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        self.cpu_s = time.process_time() - cpu0
        self.last_first = self.step_offset
        self.step_offset += njobs
        return elapsed, [round(e, 1) for e in ends], t_decode

    def verify(self, njobs, O, np, kind="u8"):
        """decoded frames of the output buffers the last jobs wrote, against the CPU oracle"""
        ok, checked = True, []
        for oj in range(min(self.nout, njobs, 2)):
            k = njobs - 1 - oj
            frames = self.frames_of(self.last_first + k)
            for fi in sorted({0, self.B // 2, self.B - 1}):
                if self.host_out:
                    got = np.array(self.pinned[k % self.nout].array[fi * self.frame_bytes:(fi + 1) * self.frame_bytes])
                else:
                    got = self.outs[k % self.nout][fi].cpu().numpy().reshape(-1).view(np.uint8)
                ref = O.decode(frames[fi]).pixels(kind, self.nch)
                ok = ok and bool(np.array_equal(got.view(np.uint8).reshape(-1), np.asarray(ref).view(np.uint8).reshape(-1)))
                checked.append(f"{k}:{fi}")
        if self.gathered is not None and self.rank == 0:
            # what the consumer rank holds of the other ranks' shards after the last job's gather (their frames are regenerated here: seeds are per rank)
            self.torch.cuda.synchronize()
            k = njobs - 1
            n, B = len(self.streams), self.B
            off = ((self.last_first + k) * 37) % n
            for r in range(1, self.world):
                for fi in sorted({0, B - 1}):
                    data = _make_stream((1000 + 1000 * r + (off + fi) % n, self.W, self.H, self.args.epf, self.stream_texture, self.stream_tree_shape))
                    got = self.gathered[r][(k % self.inner) * B + fi].cpu().numpy().reshape(-1)
                    ok = ok and bool(np.array_equal(got, O.decode(data).pixels("u8", 3)))
                    checked.append(f"rank{r}:{k}:{fi}")
        return ok, checked

    def close(self):
        if self.ccomm:
            self.torch.cuda.synchronize()
            self.jx.libjxl().JxlHipCommDestroy(self.ccomm)
            self.ccomm = None
        self.p.close()
        self.outs = None; self.pinned = None; self.gathered = None
        self.torch.cuda.empty_cache()


def _steady(e):
    n = len(e)
    return round((e[n * 2 // 3] - e[n // 5]) / max(1, n * 2 // 3 - n // 5), 2) if n >= 10 else None


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def api_concurrent(streams, O, np, threads=(1, 8, 64), per_thread=8):
    """tools/api_concurrent: the reference crate's call sequence (decode.rs:207-325) against the libjxl C ABI from T pthreads, host bytes in, host pixels out —
    what the unchanged Rust crate gets when its callers decode on many threads (decoders are Send, decode.rs:523-532).  In a subprocess (its own HIP runtime)."""
    import subprocess
    import tempfile
    import zlib
    exe = os.path.join(ROOT, "tools", "_build", "api_concurrent")
    lib = os.path.join(ROOT, "jpegxl-rs_amd", "lib", "libjxl.so")
    if not os.path.exists(exe):
        return {"error": "tools/_build/api_concurrent not built"}
    with tempfile.TemporaryDirectory() as d:
        for i, s in enumerate(streams[:64]):
            with open(os.path.join(d, f"f{i:03d}.jxl"), "wb") as fh:
                fh.write(s)
        env = dict(os.environ, GPU_MAX_HW_QUEUES="16")
        try:
            out = subprocess.run([exe, lib, d, ",".join(str(t) for t in threads), str(per_thread), "3", "verify"], env=env, capture_output=True, text=True, timeout=600)
        except subprocess.TimeoutExpired:
            return {"error": "timeout"}
    if out.returncode != 0:
        return {"error": out.stderr[-400:]}
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    crcs = next((l["crc32"] for l in lines if "crc32" in l), {})
    ok = True
    for i in (0, min(len(streams), 64) // 2, min(len(streams), 64) - 1):
        ok = ok and crcs.get(f"f{i:03d}.jxl") == (zlib.crc32(np.asarray(O.decode(streams[i]).pixels("u8", 3)).tobytes()) & 0xFFFFFFFF)
    legs = {f"threads_{l['threads']}": {k: l[k] for k in ("mpixel_per_s", "latency_ms_median", "latency_ms_p90", "decodes", "failures")} for l in lines if "threads" in l}
    return {"what": "tools/api_concurrent.cc: every thread owns a JxlDecoder and runs jpegxl-rs' decode_internal loop (SubscribeEvents, SetInput, CloseInput, ProcessInput ..., zero-filled Vec per image) over "
                    "distinct 3840x2160 frames: host bytes in, host pixels out through the libjxl C ABI; callers that decode at the same time are coalesced into jobs of one shared pipeline per device "
                    "(csrc/scheduler.cc); one warm-up round, then the timed one", "verified_vs_oracle_crc32": ok, **legs}


def extras(jx, torch, streams, cjxl_streams, W, H, device, O, np):
    """What a caller of the drop-in API sees, measured after the timed region (N = 1): see the module docstring."""
    out = {}
    dec = jx.decoder_builder()

    def latency(data, dtype=np.uint8, reps=5):
        ts = []
        for _ in range(reps + 1):
            t0 = time.perf_counter()
            dec.decode_with(data, dtype)
            ts.append((time.perf_counter() - t0) * 1e3)
        return round(_median(ts[1:]), 3)
    one = latency(streams[0])
    out["single_frame_ms"] = {"value": one, "what": f"one {W}x{H} frame through decoder_builder().decode_with(u8): host bytes in, host pixels out (parse, upload, decode, copy back); median of 5 after one "
                              "warm-up; through the shared per-device pipeline, on streams of its own", "mpixel_per_s": round(W * H / 1e6 / (one * 1e-3), 1)}
    if cjxl_streams:
        c = latency(cjxl_streams[0])
        out["single_frame_ms"]["cjxl_shape_ms"] = c
    try:
        bj = open(os.path.join(ROOT, "tests", "fixtures", "bench.jxl"), "rb").read()
        t_or = time.perf_counter(); O.decode(bj); t_or = (time.perf_counter() - t_or) * 1e3
        out["single_frame_ms"]["bench_jxl_ms"] = latency(bj, np.uint8, 3)
        out["single_frame_ms"]["bench_jxl_oracle_1_thread_ms"] = round(t_or, 1)
        out["single_frame_ms"]["bench_jxl_what"] = "samples/bench.jxl, the input of the reference's criterion bench (benches/decode.rs:10): lossless Modular 2122x1433 RGBA"
    except Exception as e:
        out["single_frame_ms"]["bench_jxl_error"] = repr(e)
    t_or = time.perf_counter(); O.decode(streams[0]); out["single_frame_ms"]["oracle_1_thread_ms"] = round((time.perf_counter() - t_or) * 1e3, 1)
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
                           "what": "fresh batch of 128 frames (BASELINE config 3 per-GPU share) through JxlHipBatch*, decoded once: prepare = host parse of headers / TOC / entropy tables + device "
                                   "allocation + upload of the compressed streams; decode = every kernel, no overlap between batches"}
    out["pcie_inclusive"] = {"mpixel_per_s": round(mpx / (t3 - t0), 1), "d2h_ms": round((t3 - t2) * 1e3, 2), "d2h_gbs": round(n * W * H * 3 / 1e9 / (t3 - t2), 1),
                             "what": "the same pass plus the copy of the decoded pixels to pinned host memory (compressed input up, 24.9 MB per frame down), nothing overlapped"}
    del b, host
    # the same 128 frames as two jobs of 64 through a library pipeline in latency mode (small_job_frames: wave-wide LF kernel, sparse HF wavefronts): the host parse of the second job runs
    # beside the first one's LF stage.  The pipeline object (streams, threads, shared planes) exists beforehand — a service would keep it —, the compressed bytes are host memory until submit
    try:
        p = jx.Pipeline(device, jobs_in_flight=4, lf_streams=4, hf_streams=2, prepare_threads=3, parse_threads=8, small_job_frames=64, reserve_frames=64, reserve_width=W, reserve_height=H)
        frames = [streams[i % len(streams)] for i in range(n)]
        fb = W * H * 3
        ts = []
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tk = [p.submit(frames[k:k + 64], "uint8", 3, device_ptrs=[dst.data_ptr() + i * fb for i in range(k, k + 64)]) for k in (0, 64)]
            for t in tk:
                p.wait(t)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        p.close()
        out["one_pass_128"]["pipelined_2x64"] = {"ms": round(_median(ts[1:]), 2), "mpixel_per_s_with_prepare": round(mpx / (_median(ts[1:]) * 1e-3), 1),
                                                 "what": "two jobs of 64 through JxlHipPipeline* with small_job_frames = 64 (parse, upload and every kernel inside the timed span; median of 3 after one warm-up pass)"}
    except Exception as e:
        out["one_pass_128"]["pipelined_2x64"] = {"error": repr(e)}
    del dst
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("JXL_BENCH_BATCH", "256")), help="frames per GPU per step")
    ap.add_argument("--distinct", type=int, default=int(os.environ.get("JXL_BENCH_DISTINCT", "0")), help="distinct synthetic frames per GPU (cycled to fill the jobs); default 256, 64 per GPU when N > 1 "
                    "(the ranks of a node share its host cores for the synthesis: 4.6 s per frame)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="weak: --batch frames per GPU per step; strong: --total-frames per step over all GPUs")
    ap.add_argument("--total-frames", type=int, default=1024, help="frames per step of the whole job with --scaling strong (BASELINE config 3)")
    ap.add_argument("--no-extras", action="store_true", help="skip api_concurrent / single_frame_ms / one_pass / streaming_host_out / the 8K workloads (N = 1 only anyway)")
    ap.add_argument("--no-8k", action="store_true", help="skip the 8K workloads (BASELINE configs 4 and 5)")
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
    ap.add_argument("--gather-impl", choices=["torch", "c"], default=os.environ.get("JXL_BENCH_GATHER_IMPL", "torch"), help="N > 1: the pixel gather through torch.distributed point-to-point "
                    "operations (default) or through the library's own C entry points over RCCL (JxlHipGatherFrames, csrc/gather.cc; weak scaling only)")
    ap.add_argument("--gather-chunk", type=int, default=32, help="frames per point-to-point transfer of the pixel gather (N > 1)")
    ap.add_argument("--in-flight", type=int, default=11, help="jobs in flight in the library pipeline (JxlHipPipelineOptions.jobs_in_flight): LF stages run this many jobs ahead of the tail")
    ap.add_argument("--lf-streams", type=int, default=11, help="side streams the LF stages of the jobs ahead are spread over")
    ap.add_argument("--hf-streams", type=int, default=int(os.environ.get("JXL_BENCH_HF_STREAMS", "2")), help="HF stages in flight beside the tail of the current job, one stream and one coefficient set each")
    ap.add_argument("--wide-first", type=int, default=int(os.environ.get("JXL_BENCH_WIDE_FIRST", "4")), help="LF stages at the start of a cold pipeline that take the one-wavefront-per-stream kernel")
    ap.add_argument("--prepare-threads", type=int, default=int(os.environ.get("JXL_BENCH_PREPARE_THREADS", "3")), help="host threads of the pipeline that each parse + prepare + upload one job at a time")
    ap.add_argument("--parse-threads", type=int, default=int(os.environ.get("JXL_BENCH_PARSE_THREADS", "8")), help="host threads one job's frames are parsed on")
    ap.add_argument("--texture", type=float, default=5.0, help="strength (sRGB levels) of the texture of the realistic-bit-rate workload (0: skip it)")
    ap.add_argument("--realistic-distinct", type=int, default=64, help="distinct frames of the realistic-bit-rate workload")
    ap.add_argument("--no-realistic", action="store_true", help="skip the second and third workload (textured frames, ~2 bpp; cjxl-shaped LF trees)")
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
    do_extras = world == 1 and not args.no_extras

    streams = make_streams(args.distinct, W, H, args.epf, seed0=1000 + 1000 * rank, texture=args.main_texture, tree_shape=args.main_tree_shape)
    # the same frames with photograph-like texture: ~2 bpp at distance 1 instead of 0.8 (second workload of the line, fewer distinct frames)
    realistic_streams = make_streams(min(args.distinct, args.realistic_distinct), W, H, args.epf, seed0=1000 + 1000 * rank, texture=args.texture) if args.texture > 0 and not args.no_realistic else None
    cjxl_streams = make_streams(min(args.distinct, args.cjxl_distinct), W, H, args.epf, seed0=1000 + 1000 * rank, texture=args.texture, tree_shape=1) if args.cjxl_distinct > 0 and not args.no_realistic else None
    hdr_streams = _pool_map(_make_8k_hdr, [6 + i for i in range(8)]) if do_extras and not args.no_8k else None
    mod_streams = _pool_map(_make_8k_modular, [5 + i for i in range(4)]) if do_extras and not args.no_8k else None
    ycbcr_streams = _pool_map(_make_ycbcr420, [700 + i for i in range(16)]) if do_extras and not args.no_8k else None
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(streams, W, H)   # before any GPU runtime is initialised in this process (fork safety)

    # the pipeline keeps ~12 HIP streams busy at once (main, HF, LF side streams, copies); the runtime maps streams onto 4 hardware queues by default and kernels of
    # streams that share a queue serialise (the library sets the same default when it is loaded first)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import numpy as np
    import torch
    import torch.distributed as dist
    import jpegxl_rs_amd as jx
    import oracle_lib as O
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
    inner = 1                       # jobs per step
    if args.scaling == "strong":
        per_rank = max(1, args.total_frames // world)
        B = min(args.batch, per_rank)
        inner = max(1, per_rank // B)

    def measure(streams=streams, texture=args.main_texture, tree_shape=args.main_tree_shape, consumer="gather", **kw):
        """one workload: the pipeline's batch objects are filled once (what constructing it costs: device arenas, pinned staging), W untimed warm-up steps, then exactly K timed
        steps from an idle pipeline, bracketed by barrier + synchronize"""
        steps, warmup = kw.pop("steps", args.steps), args.warmup
        r = Run(args, jx, torch, dist, streams, dev, local_rank, rank, world, kw.pop("B", B), inner, kw.pop("W", W), kw.pop("H", H), consumer=consumer, **kw)
        try:
            r.stream_texture, r.stream_tree_shape = texture, tree_shape
            r.run(1)                              # (a first job alone: should the shared planes be too small for this shape they grow before the ring fills)
            r.run(r.slots)                        # every batch object of the ring allocates its arenas (untimed set-up)
            r.run(warmup * inner)
            r.p.collect_times()
            elapsed, ends, t_decode = r.run(steps * inner)
            cpu_s = r.cpu_s
            if world > 1:
                t = torch.tensor([elapsed, t_decode], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                elapsed, t_decode = float(t[0].item()), float(t[1].item())
                c = torch.tensor([cpu_s], dtype=torch.float64, device=dev)
                dist.all_reduce(c, op=dist.ReduceOp.SUM)       # host CPU seconds of all ranks (they share one host)
                cpu_s = float(c[0].item())
            times, runs = r.p.collect_times()
            p = r.p
            res = {"elapsed": elapsed, "t_decode": t_decode, "step_end": ends[inner - 1::inner], "stage_ms": {k[:-3]: v / max(runs, 1) for k, v in times.items() if k != "total_ms"},
                   "stage_bytes": p.stage_bytes, "device_bytes": p.info("device_bytes"), "compressed": int(p.info("compressed_bytes") // max(1, p.info("frames"))), "slots": r.slots,
                   "prepare_ms_per_job": p.info("prepare_us_total") / 1e3 / max(1, p.info("prepared_jobs")), "gather": r.consumer == "gather", "cpu_s": cpu_s, "checksum": int(r.checksum.item()),
                   "nonzeros": p.info("hf_nonzeros") // max(1, p.info("frames")), "lf_simt": [p.info(k) for k in ("lf_simt_frames", "lf_legacy_frames", "lf_simt_wp")],
                   "private_plane_jobs": p.info("private_plane_jobs"), "B": r.B, "steps": steps}
            if not args.no_verify and rank == 0:
                res["verified"], res["verified_frames"] = r.verify(steps * inner, O, np, {"uint8": "u8", "uint16": "u16", "float32": "f32"}[np.dtype(r.dtype).name])
        finally:
            r.close()
        del r
        torch.cuda.empty_cache()
        return res

    head = measure()
    realistic = measure(realistic_streams, args.texture, 0) if args.texture > 0 and not args.no_realistic and realistic_streams else None
    cjxl = measure(cjxl_streams, args.texture, 1) if cjxl_streams else None
    # N > 1: the same job with per-rank consumers beside the gather to rank 0 (7 peers x ~80 GB/s of pixels into one GPU's xGMI links bound the gather)
    local_leg = measure(consumer="per_rank") if (world > 1 or args.per_rank_consumers) and not args.no_gather else None
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
        steady = _steady(e)
        result = {
            "metric": "Mpixel/s decode (4K VarDCT d1)", "value": round(value, 2), "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(head["elapsed"] / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": f"synthetic ({args.distinct} distinct seeded frames per GPU, tools/jxlsynth; own synthesiser, 0.8 bpp — real d1 photographs run 1.5-2.5 bpp: config.workload_realistic; the LF-group MA tree is the gradient tree the SIMT LF kernel is eligible for by "
                                                        f"construction — a default-effort cjxl encode writes weighted-predictor trees: config.workload_cjxl_shape; cpu_baseline kind 'port' is the scalar oracle, not libjxl)",
            "config": {"workload": f"job of {B} x {W}x{H} VarDCT d1 frames per GPU per step (XYB, ANS, var-block DCT8..32 mix, gaborish, EPF {args.epf}), u8 RGB out, through the library's streaming pipeline "
                                   "(JxlHipPipelineSubmit / Wait): every step's compressed frames come from host memory and are parsed, prepared and uploaded inside the timed region, outputs stay in HBM",
                       "mode": "streaming", "frames_per_gpu": B * inner, "frames_per_launch": B, "width": W, "height": H, "compressed_bytes_per_frame": head["compressed"],
                       "lane_stride_lf": args.lane_stride_lf, "lane_stride_hf": args.lane_stride_hf, "jobs_in_flight": args.in_flight, "batch_objects": head["slots"],
                       "gather": head["gather"], "pipeline": "library (csrc/pipeline.cc)", "parallelism": f"frame-sharded x{world}"},
            "roofline": {"bound": "hbm", "kernel": kernel_of[dom], "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "algorithmic_bytes_per_launch": stage_bytes[dom], "avg_launch_ms": round(stage_ms[dom], 4)},
            # the entropy stages above are serial chains (HBM fraction ~ 0 by construction); the stage that IS bound by HBM is the IDCT:
            # the same figures for it (stage = IdctTileKernel, measured while the other stages of the pipeline run beside it)
            "roofline_hbm_stage": (lambda ms, by, pf: {"bound": "hbm", "kernel": "IdctTileKernel<4, true>", "achieved": round(by / (ms * 1e-3) / 1e9, 3) if ms > 0 else None,
                                                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else None,
                                                     "traffic": pf, "algorithmic_bytes_per_launch": by, "avg_launch_ms": round(ms, 4)})(
                stage_ms.get("idct", 0.0), stage_bytes.get("idct", 0), _pmc_bytes(("IdctTileKernel<4, true>",), B)),
            # what the step as a whole runs into is neither HBM nor MFMA but vector-ALU issue (profiles/r03_notes.md): VALU wavefront-instructions of all kernels of a
            # step (SQ_INSTS_VALU, profiles/sq_valu.json) against the measured issue peak of the chip (tools/microbench/valu_issue.hip)
            "valu_issue": _valu_issue(B, head["elapsed"] / args.steps, steady),
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            # when every timed step's last byte had been written (ms after the start of the timed region, GPU clock): pipeline fill, then the steady state
            "step_end_ms": e, "steady_state_ms_per_step": steady,
            "stage_gbs": {k: round(stage_bytes[k] / (stage_ms[k] * 1e-3) / 1e9, 2) if stage_ms[k] > 0 else None for k in stage_ms},
            "device_bytes": head["device_bytes"],
            "streaming": {"prepare_threads": args.prepare_threads, "parse_threads_per_job": args.parse_threads, "distinct_frames": args.distinct,
                          "prepare_ms_per_job": round(head["prepare_ms_per_job"], 2), "prepare_ms_per_frame": round(head["prepare_ms_per_job"] / B, 4),
                          "what": "prepare = parse on the parse threads + tables into pinned staging + upload and LF stage enqueued on a side stream, wall time of one worker thread of the pipeline per job"},
        }
        result["config"]["nonzero_coefficients_per_frame"] = head["nonzeros"]
        result["config"]["bits_per_pixel"] = round(head["compressed"] * 8 / (W * H), 3)

        def leg(r, what, extra=None):
            d = {"what": what, "value": round(rate(r), 2), "unit": "Mpixel/s", "ms_per_step": round(r["elapsed"] / args.steps * 1e3, 3), "steady_state_ms_per_step": _steady(r["step_end"]),
                 "stage_ms": {k: round(v, 4) for k, v in r["stage_ms"].items()}, "compressed_bytes_per_frame": r["compressed"], "bits_per_pixel": round(r["compressed"] * 8 / (W * H), 3),
                 "nonzero_coefficients_per_frame": r["nonzeros"], "verified_vs_oracle": r.get("verified")}
            d.update(extra or {})
            return d
        if realistic is not None:
            result["config"]["workload_realistic"] = leg(realistic, f"the same pipeline on frames with photograph-like texture (bench.py _textured, {args.texture:g} sRGB levels): the bit rate of real cjxl -d 1 photographs",
                                                         {"distinct_frames": len(realistic_streams)})
        if cjxl is not None:
            result["config"]["workload_cjxl_shape"] = leg(cjxl, "the realistic-bit-rate frames with LF-group streams under the MA-tree shape of a default-effort cjxl encode (enc_modular.cc tree kinds 'WP fixed DC' + 'AC meta': "
                                                          "weighted-predictor leaves under a fixed tree over property 15 for the LF coefficients; row / N / W splits for the HF metadata) — the headline's and the realistic "
                                                          "workload's LF trees are the gradient tree `cjxl --faster_decoding` picks, which the plain SIMT LF kernel was built around",
                                                          {"lf_simt_frames": cjxl["lf_simt"][0], "lf_legacy_frames": cjxl["lf_simt"][1], "lf_simt_weighted_predictor_kernel": bool(cjxl["lf_simt"][2]),
                                                           "distinct_frames": len(cjxl_streams)})
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
        if do_extras:
            torch.cuda.empty_cache()
            jx.arena_pool_trim()
            # ---- host pixels out: the API's contract is a host buffer (decode.rs:417-430) — the pipeline with pinned host destinations, copies overlapped with later jobs
            try:
                hb = max(8, min(256, B))
                ho = measure(B=hb, host_out=True, in_flight=4, steps=max(10, min(args.steps, 20)))      # (a ring of 9 pinned buffers of hb frames: 57 GB of host memory at 256)
                px = hb * W * H * ho["steps"]
                t = torch.empty(1 << 30, dtype=torch.uint8, device=dev); hbuf = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True)
                hbuf.copy_(t); torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(4):
                    hbuf.copy_(t, non_blocking=True)
                torch.cuda.synchronize(); d2h = 4 * (1 << 30) / (time.perf_counter() - t0) / 1e9
                del t, hbuf
                result["streaming_host_out"] = {"value": round(px / ho["elapsed"] / 1e6, 2), "unit": "Mpixel/s", "frames_per_job": hb, "ms_per_job": round(ho["elapsed"] / ho["steps"] * 1e3, 3),
                                                "d2h_gbs_measured": round(d2h, 1), "pcie_ceiling_mpixel_per_s": round(d2h * 1e9 / 3 / 1e6, 1),
                                                "fraction_of_pcie_ceiling": round(px / ho["elapsed"] * 3 / (d2h * 1e9), 3), "verified_vs_oracle": ho.get("verified"),
                                                "what": "the library pipeline with pinned host destinations (JxlHipPipelineSubmit host_out): host bytes in, host pixels out, the copy of job k under the decode of the jobs behind it"}
            except Exception as ex:
                result["streaming_host_out"] = {"error": repr(ex)}
            torch.cuda.empty_cache(); jx.arena_pool_trim()
            # ---- BASELINE configs 5 and 4 at full size: batch throughput through the pipeline + single image through decode_with, with their own stage times
            if hdr_streams:
                try:
                    r5 = measure(hdr_streams, B=32, W=7680, H=4320, dtype="float32", plane_sets=2, steps=max(6, min(args.steps, 12)))      # (the pipeline's defaults: 11 jobs in flight — 92 GB of device memory; four in flight, as until r05i, left the LF stage short of streams: 9.5 against 11.7 Gpixel/s)
                    px = 32 * 7680 * 4320 * r5["steps"]
                    sm, sb = r5["stage_ms"], r5["stage_bytes"]
                    d5 = jx.decoder_builder()
                    ts = []
                    for _ in range(3):
                        t0 = time.perf_counter(); d5.decode_with(hdr_streams[0], np.float32); ts.append((time.perf_counter() - t0) * 1e3)
                    result["config"]["workload_8k_hdr_f32_epf3"] = {
                        "what": "BASELINE config 5: 7680x4320 VarDCT frames, linear-light HDR (values up to 4.0, intensity target 1000), gaborish + EPF 3 iterations, f32 RGB out (398 MB per frame); jobs of 32 through the pipeline",
                        "value": round(px / r5["elapsed"] / 1e6, 2), "unit": "Mpixel/s", "ms_per_job": round(r5["elapsed"] / r5["steps"] * 1e3, 3), "single_image_ms": round(_median(ts), 2),
                        "stage_ms": {k: round(v, 4) for k, v in sm.items()}, "stage_gbs": {k: round(sb[k] / (sm[k] * 1e-3) / 1e9, 2) if sm[k] > 0 else None for k in sm},
                        "roofline_filter_stage": {"bound": "hbm", "kernel": "EpfTileKernel<0> (gaborish applied to its own tile) + EpfTile12Kernel (passes 1 and 2, colour transform, writes the pixels); algorithmic bytes 12 B/px in + 12 B/px out, once (Batch::StageBytes)", "achieved": round(sb["filter"] / (sm["filter"] * 1e-3) / 1e9, 2) if sm["filter"] > 0 else None,
                                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(sb["filter"] / (sm["filter"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if sm["filter"] > 0 else None,
                                                  "algorithmic_bytes_per_launch": sb["filter"], "avg_launch_ms": round(sm["filter"], 3)},
                        "compressed_bytes_per_frame": r5["compressed"], "verified_vs_oracle_bit_exact": r5.get("verified"), "private_plane_jobs": r5["private_plane_jobs"]}
                except Exception as ex:
                    result["config"]["workload_8k_hdr_f32_epf3"] = {"error": repr(ex)}
                torch.cuda.empty_cache(); jx.arena_pool_trim()
            if mod_streams:
                try:
                    r4, bm = None, 2
                    for bm, infl in ((8, 2), (4, 2), (2, 2)):        # (1.6 GB of device memory per frame in flight since the per-unit scratch is sized by the unit headers — 8 GB before —: jobs of 8 = 65 GB over the five batch objects, else smaller ones)
                        try:
                            r4 = measure(mod_streams, B=bm, W=8192, H=8192, dtype="uint16", nch=1, in_flight=infl, lf_streams=2, steps=max(6, min(args.steps, 10)))
                            break
                        except Exception:
                            torch.cuda.empty_cache(); jx.arena_pool_trim()
                            if bm == 2:
                                raise
                    px = bm * 8192 * 8192 * r4["steps"]
                    sm, sb = r4["stage_ms"], r4["stage_bytes"]
                    d4 = jx.decoder_builder()
                    ts = []
                    for _ in range(3):
                        t0 = time.perf_counter(); d4.decode_with(mod_streams[0], np.uint16); ts.append((time.perf_counter() - t0) * 1e3)
                    result["config"]["workload_8k_modular_squeeze_u16"] = {
                        "what": "BASELINE config 4: lossless Modular 8192x8192 u16 (one channel), default Squeeze chain, 1024 groups + 16 LF groups of residual channels; jobs of 2 or 4 (frames_per_job) through the pipeline; stage 'lf' = global "
                                "Modular stream (ModularGlobalFastKernel), 'out' = group sub-streams (ModularGroupFastKernel), inverse Squeeze and the write stage",
                        "value": round(px / r4["elapsed"] / 1e6, 2), "unit": "Mpixel/s", "frames_per_job": bm, "ms_per_job": round(r4["elapsed"] / r4["steps"] * 1e3, 3), "single_image_ms": round(_median(ts), 2),
                        "stage_ms": {k: round(v, 4) for k, v in sm.items() if k in ("lf", "out")}, "compressed_bytes_per_frame": r4["compressed"], "verified_vs_oracle": r4.get("verified"),
                        # the two stages of a Modular job against their own bounds: the entropy chains by samples per second (a serial chain per sub-stream: its HBM fraction is ~0 by
                        # construction), the group stage (sub-streams, inverse Squeeze, write) also by its compulsory bytes — compressed sections in, u16 pixels out, once
                        "roofline_global_stream": {"bound": "latency", "kernel": "ModularGlobalFastKernel (one sub-stream per frame)", "samples_per_s": None, "avg_launch_ms": round(sm["lf"], 3)},
                        "roofline_group_stage": (lambda algo: {"bound": "hbm", "kernel": "ModularGroupFastKernel + ModInvSqueezeH/VKernel + ModularOutputKernel", "achieved": round(algo / (sm["out"] * 1e-3) / 1e9, 2) if sm["out"] > 0 else None,
                                                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(algo / (sm["out"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if sm["out"] > 0 else None,
                                                               "algorithmic_bytes_per_launch": algo, "avg_launch_ms": round(sm["out"], 3),
                                                               "samples_per_s": round(bm * 8192 * 8192 / (sm["out"] * 1e-3)) if sm["out"] > 0 else None})(bm * (r4["compressed"] + 8192 * 8192 * 2))}
                except Exception as ex:
                    result["config"]["workload_8k_modular_squeeze_u16"] = {"error": repr(ex)}
                torch.cuda.empty_cache(); jx.arena_pool_trim()
            if ycbcr_streams:
                # ---- JPEG-transcode-shaped frames: chroma-subsampled YCbCr takes IdctSubsampledKernel, the general SIMT HF instantiation and OutputKernel's
                # subsampled branch (chroma upsampling + YCbCr -> RGB + write in one pass); jobs of the headline's shape
                try:
                    rj = None
                    for jb, jf in ((args.batch, 6), (64, 4)):      # (eleven jobs of 256 in flight need 183 GB: they fit a fresh process, not always this one after the 8K legs; six measured the same or better)
                        try:
                            rj = measure(ycbcr_streams, B=jb, in_flight=jf, lf_streams=jf, consumer="none", steps=max(6, min(args.steps, 12)))
                            break
                        except Exception as ex:
                            if "memory" not in repr(ex) or (jb, jf) == (64, 4):
                                raise
                            torch.cuda.empty_cache(); jx.arena_pool_trim()
                    result["config"]["workload_jpeg_transcode_420"] = {
                        "what": f"3840x2160 frames shaped like lossless JPEG transcodes (YCbCr 4:2:0, 8x8 DCT, no restoration filters; tools/synth_ycbcr.h), jobs of {rj['B']} ({jf} in flight) through the pipeline, u8 RGB out",
                        "value": round(rj["B"] * W * H * rj["steps"] / rj["elapsed"] / 1e6, 2), "unit": "Mpixel/s", "ms_per_step": round(rj["elapsed"] / rj["steps"] * 1e3, 3), "jobs_in_flight": jf,
                        "stage_ms": {k: round(v, 4) for k, v in rj["stage_ms"].items()}, "compressed_bytes_per_frame": rj["compressed"], "verified_vs_oracle": rj.get("verified"), "distinct_frames": len(ycbcr_streams)}
                except Exception as ex:
                    result["config"]["workload_jpeg_transcode_420"] = {"error": repr(ex)}
                torch.cuda.empty_cache(); jx.arena_pool_trim()
            try:
                result.update(extras(jx, torch, streams, cjxl_streams, W, H, local_rank, O, np))
            except Exception as ex:
                result["extras_error"] = repr(ex)
            torch.cuda.empty_cache(); jx.arena_pool_trim()
            result["api_concurrent"] = api_concurrent(streams, O, np)
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
