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


def _worker4(rank, world, port, nframes, tmp):
    """four ranks, ragged shards (10 frames -> 3, 3, 2, 2): the frame order of the unsharded job comes back at rank 0; the checksum of checksums of the
    per-rank-consumer mode (bench.py Pipeline.consume_local) equals the checksum of the gathered job"""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    import synth_lib as S
    from jpegxl_rs_amd.sharding import shard_range, shard_sizes, gather_frames_ragged
    sizes = shard_sizes(nframes, world)
    b, e = shard_range(nframes, world, rank)
    assert e - b == sizes[rank] and sum(sizes) == nframes
    frames = [O.decode(S.encode_vardct(S.synthetic_image(300 + i, 48, 40), seed=300 + i, strategy_mix=0)).image("u8", 3) for i in range(b, e)]
    local = torch.from_numpy(np.stack(frames))
    for chunk in (1, 2, 8):
        out = gather_frames_ragged(local, sizes, None, dst=0, chunk_frames=chunk)
        assert (out is None) == (rank != 0)
        if rank == 0:
            np.save(os.path.join(tmp, "ragged%d.npy" % chunk), out.numpy())
    with pytest.raises(ValueError):
        gather_frames_ragged(local[:-1] if local.shape[0] > 1 else torch.cat([local, local]), sizes, None, dst=0)
    # per-rank consumers: every rank reduces its own pixels, one all-reduce of the 8-byte checksums
    padded = np.zeros((local.numel() + 3) // 4 * 4, np.uint8); padded[:local.numel()] = local.numpy().reshape(-1)
    chk = torch.tensor(int(padded.view(np.int32).astype(np.int64).sum()), dtype=torch.int64)
    dist.all_reduce(chk)
    if rank == 0:
        np.save(os.path.join(tmp, "checksum.npy"), np.array([chk.item()]))
    dist.destroy_process_group()


def test_four_ranks_ragged_shards_and_per_rank_checksums(built, tmp_path):
    import oracle_lib as O
    import synth_lib as S
    from jpegxl_rs_amd.sharding import shard_range
    world, nframes, port = 4, 10, 31500 + os.getpid() % 2000
    mp.spawn(_worker4, args=(world, port, nframes, str(tmp_path)), nprocs=world, join=True)
    ref = np.stack([O.decode(S.encode_vardct(S.synthetic_image(300 + i, 48, 40), seed=300 + i, strategy_mix=0)).image("u8", 3) for i in range(nframes)])
    for chunk in (1, 2, 8):
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "ragged%d.npy" % chunk)), ref), chunk
    want = 0
    for r in range(world):
        b, e = shard_range(nframes, world, r)
        flat = ref[b:e].reshape(-1)
        padded = np.zeros((flat.size + 3) // 4 * 4, np.uint8); padded[:flat.size] = flat
        want += int(padded.view(np.int32).astype(np.int64).sum())
    assert int(np.load(os.path.join(str(tmp_path), "checksum.npy"))[0]) == want
