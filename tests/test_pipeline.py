"""The streaming pipeline as a library object (include/jxl_hip.h JxlHipPipeline*) and the shared per-device scheduler behind the libjxl ABI.

-m gpu: jobs of mixed images against the CPU oracle (device and pinned-host destinations), images that fail alone, a damaged frame in front of jobs that
reuse its coefficient set (round-4 advisor finding), many JxlDecoder instances on many host threads (jpegxl-rs/src/decode.rs:523-532: decoders are Send) —
32 threads x distinct 4K frames, bit-exact —, and the arena-pool trim entry point.
not gpu: the symbols exist, the host-only size query works, a pipeline cannot be created without a GPU (no CPU fallback)."""
import concurrent.futures as cf
import ctypes as C
import os
import threading

import numpy as np
import pytest

from conftest import fixture_bytes
import oracle_lib as O
import synth_lib as S


@pytest.fixture(scope="module")
def jxh(built):
    import jpegxl_rs_amd as jx
    return jx


@pytest.fixture(scope="module")
def jx(built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import jpegxl_rs_amd as jx
    return jx


# ---- host side (no GPU) -------------------------------------------------------------------------------------------------------------------
def test_pipeline_symbols_exported(jxh):
    L = jxh.libjxl()
    for name in ("JxlHipPipelineCreate", "JxlHipPipelineDestroy", "JxlHipPipelineSubmit", "JxlHipPipelineWait", "JxlHipPipelineWaitAll", "JxlHipPipelineResetClock",
                 "JxlHipPipelineCollectTimes", "JxlHipPipelineStageBytes", "JxlHipPipelineGetInfo", "JxlHipHostAlloc", "JxlHipHostFree", "JxlHipImageOutSize",
                 "JxlHipArenaPoolTrim", "JxlHipArenaPoolHeld", "JxlHipSchedulerStats", "JxlHipSchedulerShutdown"):
        assert hasattr(L, name), name
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "jxl_hip.h")).read()
    assert "JxlHipPipelineSubmit" in header and "JxlHipPipelineOptions" in header
    assert C.sizeof(jxh.JxlHipPipelineOptions) == 14 * 4


def test_image_out_size_is_host_only(jxh):
    info, size = jxh.image_out_size(fixture_bytes("sample.jxl"), "uint16", 4)
    assert (info.xsize, info.ysize, size) == (40, 50, 40 * 50 * 4 * 2)
    info, size = jxh.image_out_size(fixture_bytes("sample.jxl"), "uint8", 3, align=16)
    assert size == 128 * 49 + 120
    with pytest.raises(jxh.DecodeError):
        jxh.image_out_size(bytes(64), "uint8", 3)


def test_no_pipeline_without_gpu(jxh):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(jxh.CannotCreateDecoder):
        jxh.Pipeline(0)
    assert "no CPU fallback" in jxh.last_error() or "HIP" in jxh.last_error()


# ---- GPU ---------------------------------------------------------------------------------------------------------------------------------
def _small_streams():
    out = []
    for seed, (w, h), kw in ((3, (320, 200), dict(strategy_mix=2, epf_iters=1, gab=1)), (4, (264, 520), dict(strategy_mix=1, epf_iters=1, gab=1)),
                              (5, (512, 512), dict(strategy_mix=0, epf_iters=2, gab=1)), (6, (777, 300), dict(strategy_mix=2, epf_iters=0, gab=0)),
                              (7, (1024, 640), dict(strategy_mix=1, epf_iters=1, gab=1))):
        out.append(S.encode_vardct(S.synthetic_image(seed, w, h), seed=seed, **kw))
    return out


@pytest.mark.gpu
def test_pipeline_jobs_device_and_host_out(jx):
    """Jobs of different shapes through one pipeline: plain VarDCT frames (shared planes), a Modular fixture and a patch / spline fixture (arenas of their own),
    device destinations (torch tensors) and pinned host destinations, every image against the oracle."""
    import torch
    streams = _small_streams()
    extra = [fixture_bytes("sample.jxl"), fixture_bytes("sample_grey.jxl")]
    refs = {id(d): O.decode(d).pixels("u8", 3) for d in streams + extra}
    p = jx.Pipeline(0, jobs_in_flight=3, lf_streams=2, hf_streams=1, prepare_threads=2, parse_threads=2, timed=1, reserve_frames=4, reserve_width=1024, reserve_height=640)
    assert p.info("coefficient_sets") == 2 and p.info("slots") >= 5           # (hf_streams + 1 sets; the default is two HF stages in flight = three sets)
    jobs = []
    for rnd in range(6):
        datas = [streams[(rnd + k) % len(streams)] for k in range(3)] if rnd % 3 != 2 else [extra[0], streams[rnd % len(streams)], extra[1]]
        sizes = [refs[id(d)].size for d in datas]
        if rnd % 2 == 0:
            outs = [torch.empty(s, dtype=torch.uint8, device="cuda") for s in sizes]
            t = p.submit(datas, "uint8", 3, device_ptrs=[o.data_ptr() for o in outs], capacities=sizes)
        else:
            outs = [jx.PinnedBuffer(s) for s in sizes]
            t = p.submit(datas, "uint8", 3, host_ptrs=[o.ptr for o in outs], capacities=sizes)
        jobs.append((t, datas, outs))
    ends = []
    for t, datas, outs in jobs:
        status, end_ms = p.wait(t)
        assert status == [0] * len(datas)
        ends.append(end_ms)
        for d, o in zip(datas, outs):
            got = o.cpu().numpy() if hasattr(o, "cpu") else np.array(o.array)
            assert np.array_equal(got, refs[id(d)])
    assert all(b >= a for a, b in zip(ends, ends[1:])), ends           # jobs complete in submission order
    times, runs = p.collect_times()
    assert runs >= 4 and times["hf_ms"] > 0 and times["idct_ms"] > 0
    assert p.info("shared_big_bytes") > 0 and p.info("private_plane_jobs") >= 2      # (the fixture jobs do not fit / may not use the shared planes)
    assert sum(p.stage_bytes.values()) > 0 and p.info("jobs") == 6
    p.close()


@pytest.mark.gpu
def test_pipeline_images_fail_alone_and_do_not_poison_later_jobs(jx):
    """A damaged AC stream, a truncated file, garbage and a too-small buffer each fail their own image only; the jobs behind them — which rotate through the same
    two coefficient sets without any host-side finish in between (garbage coefficients of a failed frame) — decode bit-exactly."""
    streams = _small_streams()
    refs = [O.decode(d).pixels("u8", 3) for d in streams]
    bad_ac = bytearray(streams[4]); bad_ac[len(bad_ac) * 3 // 4] ^= 0x5A; bad_ac[len(bad_ac) * 3 // 4 + 7] ^= 0xFF
    bad_ac2 = bytearray(streams[2]); bad_ac2[len(bad_ac2) * 2 // 3] ^= 0x77
    p = jx.Pipeline(0, jobs_in_flight=2, lf_streams=2, prepare_threads=1, parse_threads=2, reserve_frames=4, reserve_width=1024, reserve_height=640)
    jobs = []
    def submit(datas, caps=None):
        sizes = [refs[streams.index(d)].size if d in streams else 1024 * 640 * 3 for d in datas]
        outs = [jx.PinnedBuffer(s) for s in sizes]
        jobs.append((p.submit(datas, "uint8", 3, host_ptrs=[o.ptr for o in outs], capacities=caps or sizes), datas, outs))
    submit([streams[0], bytes(bad_ac), streams[1], bytes(bad_ac2)])
    submit([streams[4], streams[2]])                                   # same coefficient set as job 2 below, the other one than job 0
    submit([streams[2], streams[4], streams[3]])                       # reuses job 0's set: layout differs, frames sit where the damaged ones were
    submit([streams[1], streams[0][:len(streams[0]) // 2], bytes(100), streams[3]])
    submit([streams[4], streams[0]], caps=[refs[4].size, 16])
    submit([streams[4], streams[2], streams[0]])
    expect = [[0, None, 0, None], [0, 0], [0, 0, 0], [0, 1, 1, 0], [0, 1], [0, 0, 0]]        # None: a flipped byte may also decode to other pixels
    for (t, datas, outs), exp in zip(jobs, expect):
        status, _ = p.wait(t, check=False)
        for k, (d, o, e) in enumerate(zip(datas, outs, exp)):
            if e is not None:
                assert status[k] == e, (status, exp)
            if status[k] == 0 and d in streams:
                assert np.array_equal(np.array(o.array), refs[streams.index(d)]), f"job {t} image {k}"
    assert jobs and "image" in jx.last_error()
    p.close()


def _decode_threaded(jx, datas, threads, dtype=np.uint8, nch=3):
    barrier = threading.Barrier(threads)
    def work(i):
        dec = jx.decoder_builder(pixel_format=jx.PixelFormat(num_channels=nch))
        barrier.wait()
        out = []
        for k in range(i, len(datas), threads):
            meta, px = dec.decode_with(datas[k], dtype)
            out.append((k, meta, px))
        return out
    with cf.ThreadPoolExecutor(threads) as ex:
        res = [r for part in ex.map(work, range(threads)) for r in part]
    return {k: (meta, px) for k, meta, px in res}


@pytest.mark.gpu
def test_concurrent_decoders_small_images(jx):
    """Many JxlDecoder instances on many host threads (decode.rs:523-532): every image bit-exact, and the scheduler put several requests into one job."""
    streams = _small_streams() + [fixture_bytes("sample.jxl"), fixture_bytes("sample_grey.jxl"), fixture_bytes("sample_jpg.jxl")]
    refs = [O.decode(d).pixels("u8", 3) for d in streams]
    datas = [streams[i % len(streams)] for i in range(96)]
    j0, i0 = C.c_int64(), C.c_int64()
    jx.libjxl().JxlHipSchedulerStats(0, C.byref(j0), C.byref(i0))
    got = _decode_threaded(jx, datas, 16)
    for k, (meta, px) in got.items():
        assert np.array_equal(px.reshape(-1), refs[k % len(streams)]), k
    j1, i1 = C.c_int64(), C.c_int64()
    jx.libjxl().JxlHipSchedulerStats(0, C.byref(j1), C.byref(i1))
    assert i1.value - i0.value == 96
    assert j1.value - j0.value < 96, "no two requests ever shared a job"
    # failures stay with their caller: a damaged stream on one thread, clean ones on the others
    bad = bytearray(streams[4]); bad[len(bad) // 2] ^= 0xFF
    def one(i):
        try:
            return jx.decoder_builder(pixel_format=jx.PixelFormat(num_channels=3)).decode_with(bytes(bad) if i == 3 else streams[i % 5], np.uint8)[1]
        except jx.DecodeError as e:
            return e
    with cf.ThreadPoolExecutor(8) as ex:
        res = list(ex.map(one, range(8)))
    for i, r in enumerate(res):
        if i == 3:
            assert isinstance(r, jx.DecodeError) or len(r) == refs[4].size
        else:
            assert np.array_equal(r.reshape(-1), refs[i % 5])


@pytest.mark.gpu
def test_concurrent_decoders_on_their_own_streams(jx):
    """What does not go through the shared pipeline runs as a batch of one on the decoder's own non-blocking stream (SURVEY 8b "Threading"): eight threads reconstruct
    the JPEG of sample_jpg.jxl (entropy stages on the GPU, Huffman writer on the host) while others decode the 16-bit RGBA fixture through the shared pipeline."""
    jpg_jxl, jpg = fixture_bytes("sample_jpg.jxl"), fixture_bytes("sample.jpg")
    rgba = O.decode(fixture_bytes("sample.jxl")).pixels("u16", 4).view(np.uint16)
    def work(i):
        if i % 2 == 0:
            meta, (kind, data) = jx.decoder_builder().reconstruct(jpg_jxl)
            return kind == "jpeg" and bytes(data) == jpg
        meta, px = jx.decoder_builder(pixel_format=jx.PixelFormat(num_channels=4)).decode_with(fixture_bytes("sample.jxl"), np.uint16)
        return np.array_equal(px.reshape(-1), rgba)
    with cf.ThreadPoolExecutor(8) as ex:
        assert all(ex.map(work, range(32)))


@pytest.mark.gpu
def test_32_threads_distinct_4k_frames(jx):
    """>= 32 host threads x distinct 4K frames through the libjxl ABI, bit-exact vs the oracle."""
    import multiprocessing as mp
    n = 32
    with mp.get_context("fork").Pool(min(n, os.cpu_count() or 1)) as pool:
        datas = pool.starmap(_make_4k, [(1000 + i,) for i in range(n)])
        refs = pool.map(_oracle_u8, datas)
    got = _decode_threaded(jx, datas, n)
    assert len(got) == n
    for k, (meta, px) in got.items():
        assert (meta.width, meta.height) == (3840, 2160)
        assert np.array_equal(px.reshape(-1), refs[k]), f"frame {k}: {int((px.reshape(-1) != refs[k]).sum())} samples differ"


def _make_4k(seed):
    return S.encode_vardct(S.synthetic_image(seed, 3840, 2160), seed=seed, distance=1.0, epf_iters=1, gab=1, strategy_mix=1)


def _oracle_u8(data):
    return O.decode(data).pixels("u8", 3)


@pytest.mark.gpu
def test_arena_pool_trim(jx):
    """the arena pool can be handed back to the runtime (a co-resident allocator needs the memory)."""
    b = jx.BatchDecoder(0)
    b.add(S.encode_vardct(S.synthetic_image(9, 2048, 2048), seed=9, strategy_mix=1), "uint8", 3)
    b.prepare(); b.decode(); b.finish()
    del b
    held = jx.libjxl().JxlHipArenaPoolHeld()
    freed = jx.arena_pool_trim()
    assert freed == held and jx.libjxl().JxlHipArenaPoolHeld() == 0
