"""Multi-GPU path on CPU: world_size-2 gloo processes exercise the block sharding, the ordered
gather of block sizes (blockIndex offsets) and the max-over-ranks timing used by bench.py.
The encode itself is replaced by the oracle here (no GPU in this test): what is checked is that the
sharded result, gathered in block order, equals the unsharded one."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    from minnow_amd.shard import shard_range, block_offsets
    for n in (0, 1, 5, 6, 7, 375, 3000):
        for world in (1, 2, 3, 4, 8):
            got = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                assert 0 <= lo <= hi <= n
                got.extend(range(lo, hi))
            assert got == list(range(n))
            sizes = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert block_offsets([10, 0, 7, 3]) == [0, 10, 10, 17]      # go/block_index.go:16-35


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from minnow_amd.shard import shard_blocks, gather_block_sizes, max_over_ranks, block_offsets
    from oracle import oracle as orc

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    rng = np.random.default_rng(123)
    blocks = [np.mod(125 + 20 * rng.standard_normal(n), 250).astype(np.float32) for n in (1000, 64, 0, 4097, 333, 2048, 5)]
    mine = shard_blocks([b.size for b in blocks], rank, world)
    sizes, streams = [], []
    for k in mine:
        mn, bits, data = orc.float_encode(blocks[k], 0.0, 250.0, 250000, True, literal=False)
        sizes.append(data.size)
        streams.append((k, mn, bits, data.tobytes()))
    all_sizes = gather_block_sizes(sizes)
    offs = block_offsets(all_sizes)
    t = max_over_ranks(0.5 + rank)
    bucket = [None] * world
    dist.all_gather_object(bucket, streams)
    if rank == 0:
        image = bytearray(sum(all_sizes))
        metas = {}
        for part in bucket:
            for k, mn, bits, data in part:
                image[offs[k]: offs[k] + len(data)] = data
                metas[k] = (mn, bits)
        want = bytearray()
        for k, b in enumerate(blocks):
            mn, bits, data = orc.float_encode(b, 0.0, 250.0, 250000, True, literal=False)
            assert metas[k] == (mn, bits)
            want += data.tobytes()
        ok = bytes(image) == bytes(want) and abs(t - (0.5 + world - 1)) < 1e-9
        with open(out_path, "w") as f:
            f.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharded_encode_matches_unsharded(tmp_path, oracle):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def _bench_path_worker(rank, world, port, out_path):
    """bench.py's multi-rank path with the device call stubbed by the oracle: rank_plan -> encode the rank's own block range
    -> gather_block_meta -> block_offsets, against the unsharded column."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from minnow_amd.shard import rank_plan, gather_block_meta, block_offsets, max_over_ranks
    from oracle import oracle as orc

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    ok = True
    for n_global, B in ((50000, 4096), (4096 * 7, 4096), (100, 4096), (4096 * 2 + 1, 4096)):
        rng = np.random.default_rng(7)
        col = np.mod(125 + 20 * rng.standard_normal(n_global), 250).astype(np.float32)
        nblk, lo, hi, n_local = rank_plan(n_global, B, rank, world)
        mine = col[lo * B: lo * B + n_local]
        rows = []
        for k in range(hi - lo):
            mn, bits, data = orc.float_encode(mine[k * B: (k + 1) * B], 0.0, 250.0, 250000, True, literal=False)
            rows.append((mn, bits, data.size))
        allrows = gather_block_meta(rows, nblk, rank, world)
        offs = block_offsets([r[2] for r in allrows])
        want = []
        for k in range(nblk):
            mn, bits, data = orc.float_encode(col[k * B: (k + 1) * B], 0.0, 250.0, 250000, True, literal=False)
            want.append([mn, bits, data.size])
        ok = ok and allrows == want and offs == block_offsets([w[2] for w in want]) and len(offs) == nblk
        # the ranks' element counts partition the column
        ok = ok and sum(rank_plan(n_global, B, q, world)[3] for q in range(world)) == n_global
    t = max_over_ranks(1.0 + rank)
    ok = ok and abs(t - float(world)) < 1e-9
    if rank == 0:
        with open(out_path, "w") as f:
            f.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_bench_path(tmp_path, oracle):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "result2.txt")
    mp.spawn(_bench_path_worker, args=(2, port, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_rank_plan_single_rank_equals_plain():
    from minnow_amd.shard import rank_plan, gather_block_meta
    assert rank_plan(10**9, 2**20, 0, 1) == (954, 0, 954, 10**9)
    assert rank_plan(8 * 10**9, 2**20, 3, 8)[3] in (954 * 2**20, 953 * 2**20)
    parts = [rank_plan(10**9, 2**20, q, 8) for q in range(8)]
    assert sum(p[3] for p in parts) == 10**9 and parts[-1][2] == 954
    assert gather_block_meta([(1, 2, 3)], 1, 0, 1) == [[1, 2, 3]]
