#!/usr/bin/env python3
"""int64 column encode/decode timing (device resident), widths 8..45 (SURVEY 8d C1 shape, scaled up)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import minnow_amd as mnw
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5 * 10**7
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2**22
ctx = mnw.Context(0)
for bits in (8, 16, 21, 27, 32, 45):
    g = torch.Generator(device=dev); g.manual_seed(bits)
    x = 1000 + torch.randint(0, 2**bits, (N,), device=dev, dtype=torch.int64, generator=g)
    y = torch.empty_like(x)
    nblk = (N + B - 1) // B
    slot = 8 * B
    packed = torch.zeros(nblk * slot, dtype=torch.uint8, device=dev)
    ej = [mnw.EncJob(x.data_ptr() + 8*k*B, packed.data_ptr() + k*slot, min(B, N-k*B), slot, mnw.INT_GROUP) for k in range(nblk)]
    torch.cuda.synchronize()   # the context runs on its own stream: the inputs must exist first
    metas = ctx.encode_blocks(ej)
    dj = [mnw.DecJob(packed.data_ptr() + k*slot, y.data_ptr() + 8*k*B, ej[k].n, metas[k].nbytes, metas[k].min, metas[k].bits, mnw.INT_GROUP) for k in range(nblk)]
    ctx.decode_blocks(dj); torch.cuda.synchronize()
    assert torch.equal(x, y)
    ctx.timing_enable(True); ctx.timing_reset()
    reps = 5
    for _ in range(reps):
        ctx.encode_blocks(ej); ctx.decode_blocks(dj)
    torch.cuda.synchronize()
    kt = ctx.timing_get(); ctx.timing_enable(False)
    pb = sum(m.nbytes for m in metas)
    line = "bits=%2d (got %s) N=%d: " % (bits, sorted(set(m.bits for m in metas)), N)
    for k, v in kt.items():
        if v["launches"]:
            us = 1e3 * v["total_ms"] / reps
            alg = (8.0 * N + pb) if ("pack" in k or "unpack" in k) else 8.0 * N
            line += "%s %.1f us (%.2f TB/s)  " % (k.replace("k_", ""), us, alg / us / 1e6)
    print(line, flush=True)
