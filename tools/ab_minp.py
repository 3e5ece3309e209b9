#!/usr/bin/env python3
"""A/B of one context option on the minp vector kernels (HIP events), alternating in one process.
usage: AB_VALUES=0,2,4 ab_minp.py <option> [n_file] [sub_cells]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import minnow_amd as mnw
from minnow_amd import minp
opt = sys.argv[1]
VALS = [int(v) for v in os.environ.get("AB_VALUES", "0,1").split(",")]
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 500
S = int(sys.argv[3]) if len(sys.argv) > 3 else 5
L, dx = 250.0, 0.001
dev = torch.device("cuda:0")
n = nf ** 3
g = torch.Generator(device=dev); g.manual_seed(0xACEDAD)
idx = torch.arange(n, device=dev)
pos = torch.stack([idx % nf, (idx // nf) % nf, idx // (nf * nf)], dim=1).to(torch.float32) * (L / nf)
del idx
pos.add_(torch.randn(pos.shape, device=dev, generator=g).mul_(2.0))
pos = torch.remainder(pos, L).contiguous(); pos[pos >= L] = 0.0
n_sub = nf // S
slot = (4 * n_sub ** 3 + 15) // 16 * 16
packed = torch.zeros(3 * S ** 3 * slot, dtype=torch.uint8, device=dev)
out = torch.empty_like(pos)
ctx = mnw.Context(0)
torch.cuda.synchronize()
st = {}
def enc(): st["m"] = minp.vectors_encode(ctx, pos.data_ptr(), nf, S, dx, True, L, packed.data_ptr(), slot)
def dec():
    metas, low, high, pixels = st["m"]
    minp.vectors_decode(ctx, packed.data_ptr(), slot, metas, nf, S, low, high, pixels, True, L, out.data_ptr(), seed=1)
for rnd in range(3):
    for val in VALS:
        ctx.set_option(opt, val)
        for _ in range(3):
            enc(); dec()
        torch.cuda.synchronize()
        ctx.timing_enable(True); ctx.timing_reset()
        for _ in range(5):
            enc(); dec()
        torch.cuda.synchronize()
        kt = ctx.timing_get(); ctx.timing_enable(False)
        print("%s=%d: " % (opt, val) + "  ".join("%s %.1f us" % (k, 1e3 * v["total_ms"] / v["launches"]) for k, v in kt.items() if v["launches"]), flush=True)
d = (out.double() - pos.double()).abs(); d = torch.minimum(d, L - d)
assert float(d.max()) <= dx * 1.01 + 1e-4
