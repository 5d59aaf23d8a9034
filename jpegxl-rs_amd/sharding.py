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
