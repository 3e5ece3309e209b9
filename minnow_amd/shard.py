"""Block sharding across GPUs: every block's (min, bits, bytes) depends on that block
only (go/group.go:242-255, 312-327), so (column, block) pairs are dealt to ranks in contiguous ranges
and there is NO data-path collective.  The only cross-block dependency is the file byte offset of a
block, a prefix sum of block sizes (go/block_index.go:16-23) that the writing host does when it
gathers the streams in block order."""
from typing import List, Sequence, Tuple


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of n_items owned by `rank`; sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_blocks(block_sizes: Sequence[int], rank: int, world: int) -> List[int]:
    """Indices of the blocks `rank` encodes/decodes (contiguous range of the block list)."""
    lo, hi = shard_range(len(block_sizes), rank, world)
    return list(range(lo, hi))


def block_offsets(nbytes_per_block: Sequence[int]) -> List[int]:
    """blockIndex semantics (go/block_index.go:16-35): offset of block b inside its group = sum of
    the sizes of the blocks before it."""
    out, acc = [], 0
    for nb in nbytes_per_block:
        out.append(acc)
        acc += int(nb)
    return out


def gather_block_sizes(local_sizes: Sequence[int], group=None) -> List[int]:
    """All ranks' per-block stream sizes in block order (an all_gather of a few integers per block:
    metadata only, the streams themselves never cross GPUs)."""
    import torch.distributed as dist

    if group is None and not dist.is_initialized():
        return list(local_sizes)
    world = dist.get_world_size(group)
    bucket = [None] * world
    dist.all_gather_object(bucket, list(int(s) for s in local_sizes), group=group)
    out: List[int] = []
    for b in bucket:
        out.extend(b)
    return out


def max_over_ranks(seconds: float, device=None, group=None) -> float:
    """Whole-job time of a step = the slowest rank's time (bench.py contract)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
