#!/usr/bin/env python3
"""Does the second pass of a two-pass encode find its block in the Infinity Cache?  The same blocks through k_reduce_f32 + k_pack_f32
(single_read = 0) as ONE call (all blocks reduced, then all packed: the first block is long gone when the pack pass comes to it) and as
one call per block (a block's pack pass follows its own reduce pass).  usage: mall_two_pass.py [block_log2 ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import minnow_amd as mnw
N = 1 << 29
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.remainder(125 + 20 * torch.randn(N, device=dev, generator=g), 250.0); x[x >= 250.0] = 0.0
pix = mnw.float_group_pixels(0.0, 250.0, 0.001)
ctx = mnw.Context(0)
ctx.set_option("single_read", 0)
print("device", torch.cuda.get_device_name(0), flush=True)
for lb in [int(a) for a in sys.argv[1:]] or [24, 25, 26, 27]:
    B = 1 << lb
    nblk = N // B
    slot = (4 * B + 15) // 16 * 16
    packed = torch.zeros(nblk * slot, dtype=torch.uint8, device=dev)
    jobs = [mnw.EncJob(x.data_ptr() + 4 * k * B, packed.data_ptr() + k * slot, B, slot, mnw.FLOAT_GROUP, 0.0, 250.0, pix, 1) for k in range(nblk)]
    whole = ctx.enc_batch(jobs)
    single = [ctx.enc_batch([j]) for j in jobs]
    for name, batches in (("one call", [whole]), ("a call per block", single), ("one call", [whole]), ("a call per block", single)):
        for b in batches: ctx.encode_blocks(b, raw=True)
        torch.cuda.synchronize()
        ctx.timing_enable(True); ctx.timing_reset()
        reps = 5
        for _ in range(reps):
            for b in batches: ctx.encode_blocks(b, raw=True)
        torch.cuda.synchronize()
        kt = ctx.timing_get(); ctx.timing_enable(False)
        tot = sum(v["total_ms"] for v in kt.values()) / reps
        print("blocks of 2^%d (%d MB), %-16s: " % (lb, 4 * B >> 20, name) + "  ".join("%s %.1f us" % (k, 1e3 * v["total_ms"] / reps) for k, v in kt.items() if v["launches"])
              + "   sum %.1f us = %.2f of 8 TB/s (6.25 B/elt)" % (1e3 * tot, 6.25 * N / (tot * 1e-3) / 8e12), flush=True)
    del packed
