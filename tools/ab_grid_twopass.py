#!/usr/bin/env python3
"""Experiment: launch time of the two-pass float encode (k_reduce_f32 + k_pack_f32), the decode and the int64 kernels against
the grid size (option grid_cap; 0 = the persistent default), 1e8 or 1e9 elements."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import minnow_amd as mnw
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**9
lb = int(sys.argv[2]) if len(sys.argv) > 2 else 24
B = 1 << lb
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.empty(N, device=dev, dtype=torch.float32)
for c0 in range(0, N, 1 << 28):
    v = x[c0: c0 + (1 << 28)]
    v.normal_(0.0, 1.0, generator=g); v.mul_(20.0).add_(125.0)
    torch.remainder(v, 250.0, out=v); v[v >= 250.0] = 0.0
xi = (x[: N // 2] * 1000).to(torch.int64) + 10**9
nblk = (N + B - 1) // B
slot = (4 * B + 15) // 16 * 16
packed = torch.zeros(nblk * slot, dtype=torch.uint8, device=dev)
y = torch.empty_like(x)
yi = torch.empty_like(xi)
ctx = mnw.Context(0)
ctx.set_option("single_read", 0)
ej = ctx.enc_batch([mnw.EncJob(x.data_ptr() + 4 * k * B, packed.data_ptr() + k * slot, min(B, N - k * B), slot, mnw.FLOAT_GROUP, 0.0, 250.0, 250000, 1) for k in range(nblk)])
Ni = N // 2
nbi = (Ni + B - 1) // B
eji = ctx.enc_batch([mnw.EncJob(xi.data_ptr() + 8 * k * B, packed.data_ptr() + k * slot, min(B, Ni - k * B), slot, mnw.INT_GROUP) for k in range(nbi)])
torch.cuda.synchronize()
for rnd in range(2):
    for val in [int(v) for v in os.environ.get("AB_VALUES", "0,8192,32768,131072").split(",")]:
        ctx.set_option("grid_cap", val)
        out = []
        for jobs, n, nb, dst, kind in ((ej, N, nblk, y, mnw.FLOAT_GROUP), (eji, Ni, nbi, yi, mnw.INT_GROUP)):
            metas = ctx.encode_blocks(jobs)
            esz = 4 if kind == mnw.FLOAT_GROUP else 8
            dj = ctx.dec_batch([mnw.DecJob(packed.data_ptr() + k * slot, dst.data_ptr() + esz * k * B, min(B, n - k * B), metas[k].nbytes, metas[k].min, metas[k].bits, kind, 0.0, 250.0, 250000, 1, seed=1, block=k) for k in range(nb)])
            ctx.decode_blocks(dj); torch.cuda.synchronize()
            ctx.timing_enable(True); ctx.timing_reset()
            for _ in range(4):
                ctx.encode_blocks(jobs, raw=True); ctx.decode_blocks(dj)
            torch.cuda.synchronize()
            kt = ctx.timing_get(); ctx.timing_enable(False)
            out.append("  ".join("%s %.1f" % (k, 1e3 * v["total_ms"] / v["launches"]) for k, v in kt.items() if v["launches"]))
        print("grid_cap=%d: %s | %s" % (val, out[0], out[1]), flush=True)
