#!/usr/bin/env python3
"""bench.py — Mpixel/s of JPEG XL VarDCT (d1) decode of 3840x2160 frames on MI355X (BASELINE.json metric).

A "step" decodes one batch of B synthetic 4K VarDCT frames per GPU (u8 RGB out) through the C-ABI batch API
(include/jxl_hip.h): compressed streams and tables are HBM-resident before the timed region, decoded pixels stay in
HBM (a torch tensor).  With N > 1 ranks every rank decodes its own shard of the batch (weak scaling, no data-path
collective) and the decoded pixels are gathered to rank 0 over RCCL (BASELINE.json north_star).
--scaling strong --total-frames T fixes the job instead (BASELINE config 3: 1024 frames over the node): every rank decodes
T / N frames per step, in chunks of at most --batch.

Besides the headline (steady state, three batches in flight) the N = 1 line reports what a caller of the drop-in API sees:
single_frame_ms (config 2: one 4K frame through decode_with, host to host), one_pass (config-3 shape: a fresh batch of 128
frames decoded once, cold, no pipelining, with and without the host-side parse + upload), pcie_inclusive (the same pass plus
the copy of the pixels back to host memory) and verified_vs_oracle (pixels of the timed batches against the CPU oracle).

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


def _make_stream(args):
    import synth_lib as S
    seed, width, height, epf = args
    img = S.synthetic_image(seed, width, height)
    return S.encode_vardct(img, seed=seed, distance=1.0, epf_iters=epf, gab=1, strategy_mix=1)


def make_streams(distinct, width, height, epf, seed0=1000):
    """Seeded synthetic frames encoded by tools/jxlsynth, one per seed (different content, different
    varblock maps and token counts).  Generated on the host cores in parallel (4.6 s per 4K frame).  Returns list of bytes."""
    import multiprocessing as mp
    jobs = [(seed0 + i, width, height, epf) for i in range(distinct)]
    workers = max(1, min(len(jobs), os.cpu_count() or 1, 64))
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
    for i in range(n):
        b.add(streams[i % len(streams)], "uint8", 3, device_ptr=dst.data_ptr() + i * W * H * 3)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("JXL_BENCH_BATCH", "256")), help="frames per GPU per step")
    ap.add_argument("--distinct", type=int, default=int(os.environ.get("JXL_BENCH_DISTINCT", "32")), help="distinct synthetic frames (cycled to fill the batch)")
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
    ap.add_argument("--gather-chunk", type=int, default=32, help="frames per point-to-point transfer of the pixel gather (N > 1)")
    ap.add_argument("--no-pipeline", action="store_true", help="do not overlap the stages of different batches")
    ap.add_argument("--in-flight", type=int, default=10, help="batch objects in flight (pipelined): LF stages run this many steps ahead, minus one")
    ap.add_argument("--lf-streams", type=int, default=9, help="side streams the LF stages of the batches ahead are spread over")
    ap.add_argument("--hf-streams", type=int, default=int(os.environ.get("JXL_BENCH_HF_STREAMS", "1")), help="HF stages in flight beside the tail of the current step (deep pipeline), one stream and one coefficient set each")
    ap.add_argument("--wide-first", type=int, default=int(os.environ.get("JXL_BENCH_WIDE_FIRST", "2")), help="LF stages at the start of the (cold) pipeline that take the one-wavefront-per-stream kernel")
    ap.add_argument("--out-buffers", type=int, default=2, help="output buffer sets the batches in flight cycle through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    W, H = args.width, args.height

    streams = make_streams(args.distinct, W, H, args.epf, seed0=1000 + 100 * rank)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(streams, W, H)   # before any GPU runtime is initialised in this process (fork safety)

    # the pipeline keeps ~10 HIP streams busy at once (main, HF, LF side streams, gather); the runtime maps streams onto 4 hardware
    # queues by default and kernels of streams that share a queue serialise
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
    frame_bytes = W * H * 3
    pipeline = not args.no_pipeline
    # Pipeline (one MI355X): a decode is LF (entropy decode of the LF groups: a serial chain per stream, ~250 ms per launch
    # whatever the batch size, a few dozen wavefronts in the SIMT form) -> LF post-processing -> HF (entropy decode of the coefficients,
    # ~40 ms, one sparse workgroup per frame) -> IDCT -> filters + write (the HBM-bound part).  Throughput comes from batches in flight:
    # step k runs the tail of batch k on the main stream, the HF stage of batch k + 1 on a stream of its own ("deep"), and the LF stages
    # of batches k + 1 .. k + ahead on side streams.  A batch object (its LF outputs: 12 MB per 4K frame) is busy from its LF stage to
    # its tail; the coefficient planes (106 MB per frame) exist twice (HF of k + 1 beside the IDCT of k), the pixel planes and the
    # outputs as often as --out-buffers says.
    deep = pipeline and os.environ.get("JXL_BENCH_DEEP", "1") == "1"
    nbuf = int(os.environ.get("JXL_BENCH_NBUF", str(args.in_flight))) if pipeline else 1   # batches in flight: step k uses batch object k % nbuf
    nhf = max(1, args.hf_streams) if deep else 0     # HF stages in flight beside the tail of the step (each on its own stream)
    ncoef = nhf + 1                      # coefficient sets: one per HF stage in flight + the one the tail is consuming
    if deep and nbuf % ncoef:
        nbuf += ncoef - nbuf % ncoef     # (the coefficient sets rotate with k; batch object k % nbuf must always meet set k % ncoef)
    ahead = nbuf - 1                     # LF stages issued ahead of the step being finished
    nout = min(nbuf, max(1, args.out_buffers))
    outs, batches = [], []
    main = torch.cuda.current_stream()
    stream = main.cuda_stream
    for j in range(nout):
        outs.append(torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev))
    for b in range(nbuf):
        out = outs[b % nout]
        batch = jx.BatchDecoder(local_rank)
        for i in range(B):
            batch.add(streams[(i + b * B) % len(streams)], "uint8", 3, device_ptr=out.data_ptr() + i * frame_bytes)
        batch.set_lane_stride(args.lane_stride_lf, args.lane_stride_hf)
        if os.environ.get("JXL_BENCH_LDS_BUDGET"):
            batch.set_option("lds_code_budget", int(os.environ["JXL_BENCH_LDS_BUDGET"]))   # experiment: entropy-code tables of the HF stage through the L2
        if b > 0:
            batch.share_buffers(batches[0])     # the tails run one after the other on the main stream: one set of pixel planes
        if b >= ncoef:
            batch.share_coefficients(batches[b % ncoef])
        batch.prepare(stream)
        batches.append(batch)
    batch, out = batches[0], outs[0]
    gathered = None
    do_gather = world > 1 and not args.no_gather
    if do_gather and rank == 0:
        gathered = torch.empty((world, inner * B, H, W, 3), dtype=torch.uint8, device=dev)     # where the consumer rank sees the whole job's pixels (one step)
    from jpegxl_rs_amd.sharding import gather_frames_chunked
    sides = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(max(1, min(ahead, args.lf_streams)))] if pipeline else []
    comm = torch.cuda.Stream(device=dev) if do_gather else None               # RCCL gather overlaps the next step's decode
    hf_streams = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(nhf)]
    hf_done = [torch.cuda.Event() for _ in range(nbuf)]
    front_done = [torch.cuda.Event() for _ in range(nbuf)]
    lf_done = [torch.cuda.Event() for _ in range(nbuf)]
    rest_done = [torch.cuda.Event() for _ in range(nbuf)]
    out_free = [torch.cuda.Event() for _ in range(nout)]      # the gather of the step that used this output buffer last has read it
    state = {"k": 0, "front_issued": 0, "hf_issued": 0, "limit": args.warmup * inner, "gathers": 0}

    def issue_front(k, timed):
        b = k % nbuf
        side = sides[k % len(sides)]
        with torch.cuda.stream(side):
            if k >= nbuf:
                side.wait_event(rest_done[b])                       # the batch object's previous decode is complete
            if k < args.wide_first and args.lane_stride_lf < 64:
                batches[b].set_option("lf_wide_once", 1)            # cold pipeline, idle GPU: the wide LF kernel (100 instead of 250 ms until step 0 can go on)
            batches[b].decode_part(5, side.cuda_stream, timed)      # LF decode: all the HF stage waits for
            lf_done[b].record(side)
            batches[b].decode_part(6, side.cuda_stream, timed)      # LF post-processing: needed by the IDCT only
            front_done[b].record(side)

    def issue_hf(k, timed):
        b = k % nbuf
        s_ = hf_streams[k % nhf] if deep else main
        with torch.cuda.stream(s_):
            s_.wait_event(lf_done[b])
            if deep and k >= ncoef:
                s_.wait_event(rest_done[(k - ncoef) % nbuf])        # the coefficient set's previous user has consumed (and zeroed) it
            batches[b].decode_part(3, s_.cuda_stream, timed)
            hf_done[b].record(s_)

    def step(timed, last=False):
        k = state["k"]
        b = k % nbuf
        if not pipeline:
            if timed:
                batches[0].decode_timed(stream)
            else:
                batches[0].decode(stream)
            rest_done[b].record(main)
        else:
            for j in range(0, ahead + 1):
                if k + j < state["limit"] and state["front_issued"] <= k + j:
                    issue_front(k + j, timed); state["front_issued"] = k + j + 1
            for j in range(0, nhf + 1 if deep else 1):
                if k + j < state["limit"] and state["hf_issued"] <= k + j:
                    issue_hf(k + j, timed); state["hf_issued"] = k + j + 1
            if deep:
                main.wait_event(hf_done[b])
            main.wait_event(front_done[b])
            if do_gather and state["gathers"] >= nout:
                main.wait_event(out_free[k % nout])   # the previous gather of this output buffer must have read the pixels
            batches[b].decode_part(4, stream, timed)
            rest_done[b].record(main)
        if do_gather:
            with torch.cuda.stream(comm):
                comm.wait_event(rest_done[b])
                # per-chunk point-to-point transfers straight into their final place (all peers at once, one xGMI link each)
                j = k % inner
                gather_frames_chunked(outs[k % nout] if pipeline else outs[0], gathered[:, j * B:(j + 1) * B] if rank == 0 else None, dst=0, chunk_frames=args.gather_chunk)
                out_free[k % nout].record(comm)
            state["gathers"] += 1
            if not pipeline:
                main.wait_event(out_free[k % nout])
        state["k"] = k + 1

    for i in range(args.warmup * inner):
        step(False, last=(i == args.warmup * inner - 1))
    torch.cuda.synchronize()
    for bt in batches:
        bt.finish(stream)
    state["k"] = 0; state["front_issued"] = 0; state["hf_issued"] = 0; state["limit"] = args.steps * inner
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t_start = torch.cuda.Event(enable_timing=True); t_start.record(main)
    step_marks = []
    for i in range(args.steps * inner):
        step(True, last=(i == args.steps * inner - 1))
        ev = torch.cuda.Event(enable_timing=True); ev.record(main); step_marks.append(ev)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    for bt in batches:
        bt.finish(stream)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    times, runs = {}, 0
    for bt in batches:
        t_, r_ = bt.collect_times()
        runs += r_
        for kk, vv in t_.items():
            times[kk] = times.get(kk, 0.0) + vv
    stage_bytes = batch.stage_bytes
    if rank == 0:
        total_px = world * B * inner * W * H * args.steps
        value = total_px / elapsed / 1e6
        # dominant kernel = stage with the largest device time; roofline from its ALGORITHMIC bytes per launch
        stage_ms = {k[:-3]: v / max(runs, 1) for k, v in times.items() if k != "total_ms"}
        dom = max(stage_ms, key=stage_ms.get)
        kernel_of = {"lf": "LfDecodeSimtKernel" if args.lane_stride_lf < 64 else "LfDecodeKernel", "lfpost": "LlfSigmaKernel", "hf": "HfDecodeSimtKernel" if args.lane_stride_hf == 1 else "HfDecodeKernel",
                     "idct": "IdctTileKernel", "filter": "FusedGabEpf1OutKernel" if args.epf == 1 else "EpfKernel", "out": "OutputKernel"}
        achieved = stage_bytes[dom] / (stage_ms[dom] * 1e-3) / 1e9 if stage_ms[dom] > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:   # HBM bytes per frame of that kernel from the PMC passes (profiles/r01b_pmc_traffic.md) x frames per launch
                per_frame = json.load(open(pmc))["per_kernel"].get(kernel_of[dom])
                traffic = int(per_frame * B) if per_frame is not None else None
            except Exception:
                traffic = None
        result = {
            "metric": "Mpixel/s decode (4K VarDCT d1)", "value": round(value, 2), "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": f"synthetic ({args.distinct} distinct seeded frames per GPU cycled over the batch, tools/jxlsynth; own synthesiser, 0.8 bpp — real d1 photographs run 1.5-2.5 bpp)",
            "config": {"workload": f"batch of {B} x {W}x{H} VarDCT d1 frames per GPU (XYB, ANS, var-block DCT8..32 mix, gaborish, EPF {args.epf}), u8 RGB out, "
                                   "inputs and outputs resident in HBM",
                       "frames_per_gpu": B * inner, "frames_per_launch": B, "width": W, "height": H, "compressed_bytes_per_frame": int(batch.compressed_bytes // B),
                       "lane_stride_lf": args.lane_stride_lf, "lane_stride_hf": args.lane_stride_hf,
                       "gather": bool(do_gather), "pipelined_steps": bool(pipeline), "parallelism": f"frame-sharded x{world}"},
            "roofline": {"bound": "hbm", "kernel": kernel_of[dom], "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "algorithmic_bytes_per_launch": stage_bytes[dom], "avg_launch_ms": round(stage_ms[dom], 4)},
            # the entropy stages above are serial chains (HBM fraction ~ 0 by construction); the stage that IS bound by HBM is the IDCT:
            # the same figures for it (stage = IdctTileKernel + IdctRareSpecialKernel, measured while the LF stages of two other batches run)
            "roofline_hbm_stage": (lambda ms, by, pf: {"bound": "hbm", "kernel": "IdctTileKernel (+ IdctRareSpecialKernel)", "achieved": round(by / (ms * 1e-3) / 1e9, 3) if ms > 0 else None,
                                                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else None,
                                                     "traffic": pf, "algorithmic_bytes_per_launch": by, "avg_launch_ms": round(ms, 4)})(
                stage_ms.get("idct", 0.0), stage_bytes.get("idct", 0), _pmc_bytes(("IdctTileKernel<4, true>", "IdctRareSpecialKernel"), B)),
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            # when the tail of every timed step had completed (ms after the start of the timed region): pipeline fill, then the steady state
            "step_end_ms": [round(t_start.elapsed_time(e), 1) for e in step_marks],
            "stage_gbs": {k: round(stage_bytes[k] / (stage_ms[k] * 1e-3) / 1e9, 2) if stage_ms[k] > 0 else None for k in stage_ms},
            "device_bytes": sum(bt.device_bytes for bt in batches),
        }
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
        if not args.no_verify:
            # outside the timed region: decoded frames of the batches the timed steps wrote, against the CPU oracle
            import oracle_lib as O
            ok, checked = True, []
            total_steps = args.steps * inner
            for oj in range(len(outs)):
                ks = [k for k in range(total_steps) if (k % len(outs) if pipeline else 0) == oj]
                if not ks:
                    continue
                bi = ks[-1] % nbuf                    # the batch object whose tail wrote this output buffer last
                for fi in sorted({0, B // 2, B - 1}):
                    got = outs[oj][fi].cpu().numpy().reshape(-1)
                    ref = O.decode(streams[(fi + bi * B) % len(streams)]).pixels("u8", 3)
                    ok = ok and bool(np.array_equal(got, ref))
                    checked.append(f"{bi}:{fi}")
            result["verified_vs_oracle"] = ok
            result["verified_frames"] = checked
        if world == 1 and not args.no_extras:
            # the caller-side figures are measured on a GPU that holds nothing else: release the resident batches first (a
            # 40 GB hipMalloc next to 117 GB of live allocations took 0.5 s)
            del bt, batch, out
            batches.clear(); outs.clear()
            torch.cuda.empty_cache()
            result.update(extras(jx, torch, streams, W, H, local_rank))
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
