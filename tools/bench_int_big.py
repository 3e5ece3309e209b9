#!/usr/bin/env python3
"""int64 blocks larger than the flat sweep's 2^21 elements: k_encode_reg_i64 with sweeper workgroups (big_blocks = 1) against
k_reduce_i64 + k_pack_i64 (big_blocks = 0), kernel time by HIP events, alternating in one process.  usage: bench_int_big.py [elements]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import minnow_amd as mnw
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2 * 10**8
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randint(10**6, 10**6 + 2**20, (N,), device=dev, dtype=torch.int64, generator=g)      # 20-bit IDs
ctx = mnw.Context(0)
if os.environ.get("INT_FLAT"): ctx.set_option("int_flat_items", int(os.environ["INT_FLAT"]))
res = {}
for B in (1 << 20, 1 << 21, 3 << 20, 1 << 22, 6 << 20, 2 * 1016 * 4096, 1 << 24):
    nblk = N // B
    slot = (8 * B + 15) // 16 * 16
    packed = torch.zeros(nblk * slot, dtype=torch.uint8, device=dev)
    ej = ctx.enc_batch([mnw.EncJob(x.data_ptr() + 8 * k * B, packed.data_ptr() + k * slot, B, slot, mnw.INT_GROUP) for k in range(nblk)])
    row = {}
    for rnd in range(2):
        for big in (1, 0):
            ctx.set_option("big_blocks", big)
            for _ in range(2): ctx.encode_blocks(ej, raw=True)
            torch.cuda.synchronize()
            ctx.timing_enable(True); ctx.timing_reset()
            for _ in range(5): ctx.encode_blocks(ej, raw=True)
            torch.cuda.synchronize()
            kt = ctx.timing_get(); ctx.timing_enable(False)
            us = sum(v["total_ms"] for v in kt.values()) / 5 * 1e3
            names = "+".join(k for k, v in kt.items() if v["launches"])
            bits = ctx.metas_of(ctx.encode_blocks(ej, raw=True))[0].bits if hasattr(ctx, "metas_of") else 20
            alg = nblk * B * (8 + 20 / 8)
            row["big_blocks=%d" % big] = {"kernels": names, "us": round(us, 1), "frac_of_8TBps": round(alg / (us * 1e-6) / 8e12, 3)}
    res["block %d" % B] = row
    print("block of %9d int64 (%d blocks): " % (B, nblk) + "   ".join("%s: %s %.1f us = %.2f" % (k, v["kernels"], v["us"], v["frac_of_8TBps"]) for k, v in row.items()), flush=True)
    del packed
print(json.dumps(res))
