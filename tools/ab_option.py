#!/usr/bin/env python3
"""A/B of one context option on the float-encode kernel time (HIP events), alternating in one process.
usage: ab_option.py <option> [block_log2] [particles]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import minnow_amd as mnw
opt = sys.argv[1]
VALS = [int(v) for v in os.environ.get('AB_VALUES', '0,1').split(',')]
lb = int(sys.argv[2]) if len(sys.argv) > 2 else 20
N = int(float(sys.argv[3])) if len(sys.argv) > 3 else 10**8
B = 1 << lb
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.remainder(125 + 20 * torch.randn(N, device=dev, generator=g), 250.0); x[x >= 250.0] = 0.0
y = torch.empty_like(x)
nblk = (N + B - 1) // B
slot = (4 * B + 15) // 16 * 16
packed = torch.zeros(nblk * slot, dtype=torch.uint8, device=dev)
DX = float(os.environ.get("AB_DX", "0.001"))          # 0.004: 16-bit blocks
pix = mnw.float_group_pixels(0.0, 250.0, DX)
ctx = mnw.Context(0)
ej = ctx.enc_batch([mnw.EncJob(x.data_ptr() + 4 * k * B, packed.data_ptr() + k * slot, min(B, N - k * B), slot, mnw.FLOAT_GROUP, 0.0, 250.0, pix, 1) for k in range(nblk)])
torch.cuda.synchronize()
metas = ctx.encode_blocks(ej)
dj = ctx.dec_batch([mnw.DecJob(packed.data_ptr() + k * slot, y.data_ptr() + 4 * k * B, min(B, N - k * B), metas[k].nbytes, metas[k].min, metas[k].bits, mnw.FLOAT_GROUP, 0.0, 250.0, pix, 1, seed=1, block=k) for k in range(nblk)])
for rnd in range(4):
    for val in VALS:
        ctx.set_option(opt, val)
        for _ in range(3):
            ctx.encode_blocks(ej, raw=True); ctx.decode_blocks(dj)
        torch.cuda.synchronize()
        ctx.timing_enable(True); ctx.timing_reset()
        for _ in range(10):
            ctx.encode_blocks(ej, raw=True); ctx.decode_blocks(dj)
        torch.cuda.synchronize()
        kt = ctx.timing_get(); ctx.timing_enable(False)
        print("%s=%d: " % (opt, val) + "  ".join("%s %.1f us" % (k, 1e3 * v["total_ms"] / v["launches"]) for k, v in kt.items() if v["launches"]), flush=True)
