"""Frame-level sharding of an image batch across the GPUs of one node: independent frames → contiguous
blocks per rank, no data-path collective during decode; one gather of decoded pixels to the consumer rank."""
from typing import List, Tuple


def shard_range(num_frames: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [begin, end) of frames owned by `rank`; sizes differ by at most one, earlier ranks get the extra."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, extra = divmod(num_frames, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_sizes(num_frames: int, world_size: int) -> List[int]:
    return [shard_range(num_frames, world_size, r)[1] - shard_range(num_frames, world_size, r)[0] for r in range(world_size)]


def gather_frames(local, dst: int = 0, group=None):
    """Gathers per-rank tensors of decoded frames (equal shapes) to rank `dst` (RCCL when the tensors live on GPUs,
    gloo in the CPU tests).  Returns the list of tensors on dst, None elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    out = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local, out, dst=dst, group=group)
    return out


def gather_frames_chunked(local, out=None, dst: int = 0, chunk_frames: int = 32, group=None):
    """Gathers [frames, ...] tensors of equal shape to rank `dst` as point-to-point transfers of `chunk_frames` frames each,
    placed directly at their final position in `out` ([world, frames, ...] on dst; allocated when None).  One group of
    sends/receives per chunk: on GPUs (RCCL) every peer's chunk travels over its own xGMI link at the same time, a chunk is at
    most world x chunk_frames x frame bytes in flight, and the first chunks can leave while later work is still queued
    behind them on the compute stream.  Returns `out` on dst, None elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = local.shape[0]
    chunk_frames = max(1, int(chunk_frames))
    if rank == dst:
        if out is None:
            out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        if tuple(out.shape) != (world,) + tuple(local.shape) or out.dtype != local.dtype:
            raise ValueError("gather destination must be [world, *local.shape] of the same dtype")
        if out[dst].data_ptr() != local.data_ptr():
            out[dst].copy_(local, non_blocking=True)
    # gloo moves host memory only: device tensors (the one-GPU functional check of bench.py's N > 1 flow, JXL_BENCH_BACKEND=gloo) are staged
    staged = local.is_cuda and dist.get_backend(group) == "gloo"
    for c0 in range(0, n, chunk_frames):
        c1 = min(n, c0 + chunk_frames)
        if rank == dst:
            peers = [r for r in range(world) if r != dst]
            bufs = [torch.empty(out[r][c0:c1].shape, dtype=local.dtype) if staged else out[r][c0:c1] for r in peers]
            ops = [dist.P2POp(dist.irecv, b, r, group) for b, r in zip(bufs, peers)]
        else:
            if staged:
                torch.cuda.current_stream().synchronize()
            ops = [dist.P2POp(dist.isend, local[c0:c1].cpu() if staged else local[c0:c1], dst, group)]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()       # (RCCL: orders the current stream after the transfer, does not block the host)
        if staged and rank == dst:
            for b, r in zip(bufs, peers):
                out[r][c0:c1].copy_(b)
    return out if rank == dst else None


def gather_frames_ragged(local, sizes, out=None, dst: int = 0, chunk_frames: int = 32, group=None):
    """The gather for shards of unequal length (shard_sizes: 1024 frames over 3, 5, 6 or 7 GPUs): rank r holds `sizes[r]` frames, rank `dst` receives
    them in rank order into `out` ([sum(sizes), ...]; allocated when None) — the frame order of the unsharded job.  Point-to-point chunks like
    gather_frames_chunked; a peer whose shard is exhausted simply has no transfer in the later rounds.  Returns `out` on dst, None elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [int(v) for v in sizes]
    if len(sizes) != world or local.shape[0] != sizes[rank]:
        raise ValueError("sizes must list every rank's shard length, and this rank's shard must have its length")
    offs = [sum(sizes[:r]) for r in range(world)]
    total = sum(sizes)
    chunk_frames = max(1, int(chunk_frames))
    if rank == dst:
        if out is None:
            out = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        if tuple(out.shape) != (total,) + tuple(local.shape[1:]) or out.dtype != local.dtype:
            raise ValueError("gather destination must be [sum(sizes), *frame shape] of the same dtype")
        if sizes[dst]:
            out[offs[dst]:offs[dst] + sizes[dst]].copy_(local, non_blocking=True)
    staged = local.is_cuda and dist.get_backend(group) == "gloo"
    for c0 in range(0, max(sizes) if sizes else 0, chunk_frames):
        ops, bufs = [], []
        if rank == dst:
            for r in range(world):
                c1 = min(sizes[r], c0 + chunk_frames)
                if r == dst or c1 <= c0:
                    continue
                view = out[offs[r] + c0:offs[r] + c1]
                buf = torch.empty(view.shape, dtype=local.dtype) if staged else view
                bufs.append((buf, view))
                ops.append(dist.P2POp(dist.irecv, buf, r, group))
        else:
            c1 = min(sizes[rank], c0 + chunk_frames)
            if c1 > c0:
                if staged:
                    torch.cuda.current_stream().synchronize()
                ops.append(dist.P2POp(dist.isend, local[c0:c1].cpu() if staged else local[c0:c1], dst, group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if staged:
            for buf, view in bufs:
                view.copy_(buf)
    return out if rank == dst else None
