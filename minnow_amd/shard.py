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


def rank_plan(n_global: int, block: int, rank: int, world: int) -> Tuple[int, int, int, int]:
    """A column of n_global elements in blocks of `block`, dealt to `world` ranks in contiguous block ranges:
    -> (blocks in all, first block of this rank, one past its last block, elements of this rank).  Only the column's
    last block can be ragged, and it belongs to the last rank that has any block."""
    if n_global < 0 or block < 1:
        raise ValueError("bad column")
    nblk = (n_global + block - 1) // block
    lo, hi = shard_range(nblk, rank, world)
    n_local = max(0, min(hi * block, n_global) - lo * block)
    return nblk, lo, hi, n_local


def gather_block_meta(local_rows: Sequence[Sequence[int]], nblk_global: int, rank: int, world: int, device=None, group=None):
    """(min, bits, nbytes) of every block of the column in block order, from every rank's rows for its own range (what
    intGroup.writeData appends to g.mins / g.bits and hands to blockIndex.addBlock, go/group.go:251-254): one all_gather of
    three integers per block -- metadata only, the streams never cross ranks.  Without an initialised process group (one
    rank) the rows are returned as they are."""
    import torch
    import torch.distributed as dist

    rows = [[int(v) for v in r] for r in local_rows]
    if world == 1 or not dist.is_initialized():
        if len(rows) != nblk_global:
            raise ValueError("one rank must hold every block")
        return rows
    ranges = [shard_range(nblk_global, q, world) for q in range(world)]
    if len(rows) != ranges[rank][1] - ranges[rank][0]:
        raise ValueError("rank %d holds %d blocks, its range has %d" % (rank, len(rows), ranges[rank][1] - ranges[rank][0]))
    width = max(1, max(hi - lo for lo, hi in ranges))
    mine = torch.zeros((width, 3), dtype=torch.int64, device=device)
    if rows:
        mine[: len(rows)] = torch.tensor(rows, dtype=torch.int64, device=device)
    bucket = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(bucket, mine, group=group)
    out = []
    for q, (lo, hi) in enumerate(ranges):
        out.extend(bucket[q][: hi - lo].cpu().tolist())
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
