"""CPU test of the N > 1 path: world_size-2 `gloo` processes shard a batch of frames, "decode" their shard (with the
oracle standing in for the GPU decode — test infrastructure) and gather the pixels to rank 0 exactly like bench.py does
with RCCL on GPUs."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, nframes, tmp):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    import synth_lib as S
    from jpegxl_rs_amd.sharding import shard_range, gather_frames, gather_frames_chunked
    b, e = shard_range(nframes, world, rank)
    frames = []
    for i in range(b, e):
        img = S.synthetic_image(100 + i, 64, 48)
        frames.append(O.decode(S.encode_vardct(img, seed=100 + i, strategy_mix=1)).image("u8", 3))
    local = torch.from_numpy(np.stack(frames))
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)            # the max-over-ranks timing reduction of bench.py
    assert t.item() == world
    dist.barrier()
    out = gather_frames(local, dst=0)
    # the chunked point-to-point form bench.py uses (direct placement, ragged last chunk, a view of a larger job buffer as target)
    job = torch.zeros((world, 3 * local.shape[0]) + tuple(local.shape[1:]), dtype=local.dtype) if rank == 0 else None
    n = local.shape[0]
    for j in range(3):
        res = gather_frames_chunked(local + j, job[:, j * n:(j + 1) * n] if rank == 0 else None, dst=0, chunk_frames=1 + j)
        assert (res is None) == (rank != 0)
    alloc = gather_frames_chunked(local, None, dst=0, chunk_frames=5)
    if rank == 0:
        np.save(os.path.join(tmp, "gathered.npy"), torch.cat(out).numpy())
        for j in range(3):
            assert torch.equal(job[:, j * n:(j + 1) * n].reshape((-1,) + tuple(local.shape[1:])), torch.cat(out) + j)
        assert torch.equal(alloc.reshape((-1,) + tuple(local.shape[1:])), torch.cat(out))
        with pytest.raises(ValueError):
            gather_frames_chunked(local, torch.zeros((world, n + 1) + tuple(local.shape[1:]), dtype=local.dtype), dst=0)
    else:
        assert out is None and alloc is None
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(built, tmp_path):
    import oracle_lib as O
    import synth_lib as S
    world, nframes, port = 2, 4, 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, nframes, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "gathered.npy"))
    assert got.shape == (nframes, 48, 64, 3)
    for i in range(nframes):
        img = S.synthetic_image(100 + i, 64, 48)
        ref = O.decode(S.encode_vardct(img, seed=100 + i, strategy_mix=1)).image("u8", 3)
        assert np.array_equal(got[i], ref)
