"""GPU tests (-m gpu) of bench.py's own control flow: the N > 1 path (frame sharding, per-rank pipelines, chunked gather to rank 0,
max-over-ranks timing) exercised with two ranks on ONE GPU over gloo — RCCL refuses two ranks per device, so this is a functional check, not
a measurement — and the per-GPU share of BASELINE config 3 (1024 frames over 8 GPUs = 128 4K frames in one pass)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
import oracle_lib as O
import synth_lib as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def jx(built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import jpegxl_rs_amd as jx
    return jx


def test_two_ranks_share_one_gpu_and_gather(jx):
    env = dict(os.environ, JXL_BENCH_SHARE_GPU="1", JXL_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--width", "512", "--height", "384", "--batch", "16", "--distinct", "6",
           "--no-extras", "--no-cpu-baseline", "--no-realistic", "--in-flight", "4", "--lf-streams", "3", "--prepare-threads", "2", "--parse-threads", "2"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["gather"] and d["config"]["mode"] == "streaming" and d["config"]["frames_per_gpu"] == 16
    assert d["config"]["pipeline"].startswith("library")            # the schedule is the library's (JxlHipPipeline*), not the benchmark's
    assert d["verified_vs_oracle"] is True
    assert any(v.startswith("rank1:") for v in d["verified_frames"])           # pixels of the other rank's shard, as gathered at rank 0
    assert d["gather_ms"] >= 0 and d["decode_only_mpixel_per_s"] >= d["value"] > 0
    assert abs(d["value"] - 2 * 16 * 512 * 384 * 3 / (d["ms_per_step"] * 3e-3) / 1e6) < 0.01 * d["value"]   # whole-job pixels over the max-over-ranks time
    # the per-rank-consumer leg beside the gather: its own rate, the checksum of checksums of both ranks' pixels, and the host CPU budget of the job
    pr = d["per_rank_consumers"]
    assert pr["value"] > 0 and pr["verified_vs_oracle"] is True and pr["checksum_of_checksums"] != 0
    assert d["host_cpu"]["cpu_s_per_frame"] > 0 and d["host_cpu"]["cores_busy"] > 0


def test_config3_per_gpu_share(jx):
    """128 frames of 3840x2160 in one pass (one GPU's eighth of BASELINE config 3), SIMT LF decode: frames against the oracle, and the
    size-independent property that equal streams give equal pixels wherever they sit in the batch."""
    import torch
    n, distinct, W, H = 128, 4, 3840, 2160
    streams = [S.encode_vardct(S.synthetic_image(2000 + i, W, H), seed=2000 + i, strategy_mix=1, epf_iters=1, gab=1) for i in range(distinct)]
    out = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda")
    b = jx.BatchDecoder(0)
    b.add_many([streams[i % distinct] for i in range(n)], "uint8", 3, device_ptrs=[out.data_ptr() + i * W * H * 3 for i in range(n)], threads=8)
    b.set_lane_stride(8, 1)
    b.prepare(); b.decode(); b.finish()
    assert b.info_value("lf_simt_frames") == n
    for i in range(distinct, n):
        assert torch.equal(out[i], out[i % distinct]), i
    for i in (0, 3):
        assert np.array_equal(out[i].cpu().numpy().reshape(-1), O.decode(streams[i]).pixels("u8", 3)), i
