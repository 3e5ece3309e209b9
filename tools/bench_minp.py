#!/usr/bin/env python3
"""minp.Writer.Vectors / Reader.Vectors on one file cell resident on the GPU (BASELINE configs[3], per-GPU shape:
n_file = 500 -> 1.25e8 particles, sub_cells = 5 -> 375 blocks of 1e6 per vector): time of the layout passes alone and
of the whole write / read paths, with the HIP-event time of every kernel.  Prints one JSON object."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import minnow_amd as mnw  # noqa: E402
from minnow_amd import minp  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--n-file", type=int, default=500)
p.add_argument("--sub-cells", type=int, default=5)
p.add_argument("--reps", type=int, default=10)
p.add_argument("--opt", action="append", default=[], help="library option name=value (mnw_set_option), repeatable")
a = p.parse_args()

dev = torch.device("cuda:0")
nf, S, L, dx = a.n_file, a.sub_cells, 250.0, 0.001
n = nf ** 3
g = torch.Generator(device=dev); g.manual_seed(0xACEDAD)
idx = torch.arange(n, device=dev)
grid = torch.stack([idx % nf, (idx // nf) % nf, idx // (nf * nf)], dim=1).to(torch.float32) * (L / nf)
pos = torch.remainder(grid + 2.0 * torch.randn(grid.shape, device=dev, generator=g), L).contiguous()
pos[pos >= L] = 0.0
del grid, idx
n_sub = nf // S
nb = 3 * S ** 3
slot = (4 * n_sub ** 3 + 15) // 16 * 16
packed = torch.zeros(nb * slot, dtype=torch.uint8, device=dev)
soa = torch.empty(3 * n, dtype=torch.float32, device=dev)
out = torch.empty_like(pos)
ctx = mnw.Context(0)
for kv in a.opt:
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
torch.cuda.synchronize()


def timed(fn, reps):
    for _ in range(4):          # every ring slot of the context allocates its tables on first use
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


res = {"n_file": nf, "sub_cells": S, "particles": n, "blocks": nb}
t_g = timed(lambda: ctx.subcell_gather(pos.data_ptr(), nf, S, soa.data_ptr()), a.reps)
t_s = timed(lambda: ctx.subcell_scatter(soa.data_ptr(), nf, S, True, L, out.data_ptr()), a.reps)
assert torch.equal(out, pos)
res["gather_us"], res["scatter_us"] = t_g * 1e6, t_s * 1e6
res["gather_GBps_of_24B_per_particle"] = 24.0 * n / t_g / 1e9
res["scatter_GBps_of_24B_per_particle"] = 24.0 * n / t_s / 1e9
state = {}


def enc():
    state["m"] = minp.vectors_encode(ctx, pos.data_ptr(), nf, S, dx, True, L, packed.data_ptr(), slot)


def dec():
    metas, low, high, pixels = state["m"]
    minp.vectors_decode(ctx, packed.data_ptr(), slot, metas, nf, S, low, high, pixels, True, L, out.data_ptr(), seed=1)


t_e = timed(enc, a.reps)
t_d = timed(dec, a.reps)
metas = state["m"][0]
packed_bytes = sum(m.nbytes for m in metas)
res["vectors_encode_us"], res["vectors_decode_us"] = t_e * 1e6, t_d * 1e6
res["vectors_encode_GBps_input"] = 12.0 * n / t_e / 1e9
res["vectors_decode_GBps_output"] = 12.0 * n / t_d / 1e9
res["bits"] = sorted(set(m.bits for m in metas))
res["algorithmic_bytes_encode"] = 12.0 * n + packed_bytes          # one read of the grid, one write of the streams
res["encode_frac_of_8TBps"] = res["algorithmic_bytes_encode"] / t_e / 8e12
res["decode_frac_of_8TBps"] = res["algorithmic_bytes_encode"] / t_d / 8e12
ctx.timing_enable(True); ctx.timing_reset()
enc(); dec(); torch.cuda.synchronize()
res["kernels_us"] = {k: 1e3 * v["total_ms"] / v["launches"] for k, v in ctx.timing_get().items() if v["launches"]}
d = (out.double() - pos.double()).abs()
d = torch.minimum(d, L - d)
assert float(d.max()) <= dx * 1.01 + 1e-4 or os.environ.get("MNW_DIAG")      # (diagnostic builds of the library give wrong streams)
print(json.dumps(res))
