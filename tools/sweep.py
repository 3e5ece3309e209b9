#!/usr/bin/env python3
"""Sweep block size / grid cap / group size for the float encode + decode path (device-resident).
Prints per-configuration encode and decode times (torch CUDA events around the whole batch)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import minnow_amd as mnw

p = argparse.ArgumentParser()
p.add_argument("--n", type=int, default=10**8)
p.add_argument("--blocks", type=str, default="22,24,26,27")
p.add_argument("--grids", type=str, default="0")
p.add_argument("--groups", type=str, default="0")
p.add_argument("--fused", type=str, default="0")
p.add_argument("--single", type=str, default="1")
p.add_argument("--reps", type=int, default=10)
a = p.parse_args()

dev = torch.device("cuda:0")
N = a.n
g = torch.Generator(device=dev); g.manual_seed(0xACEDAD)
x = torch.randn(N, device=dev, generator=g) * 20 + 125
x = torch.remainder(x, 250.0); x[x >= 250.0] = 0.0
y = torch.empty_like(x)
ctx = mnw.Context(0)
stream = None
pixels = 250000

def run(B, grid, group_mb, fused=0, single=1):
    ctx.set_option("single_read", single)
    ctx.set_option("grid_cap", grid)
    ctx.set_option("group_bytes", group_mb << 20)
    ctx.set_option("fused", fused)
    nblk = (N + B - 1) // B
    slot = (4 * B + 15) // 16 * 16
    packed = torch.zeros(nblk * slot, dtype=torch.uint8, device=dev)
    ej = [mnw.EncJob(x.data_ptr() + 4*k*B, packed.data_ptr() + k*slot, min(B, N-k*B), slot, mnw.FLOAT_GROUP, 0.0, 250.0, pixels, 1) for k in range(nblk)]
    torch.cuda.synchronize()
    metas = ctx.encode_blocks(ej, stream=stream)
    dj = [mnw.DecJob(packed.data_ptr() + k*slot, y.data_ptr() + 4*k*B, ej[k].n, metas[k].nbytes, metas[k].min, metas[k].bits, mnw.FLOAT_GROUP, 0.0, 250.0, pixels, 1, 1, k) for k in range(nblk)]
    def timeit(fn):
        fn(); torch.cuda.synchronize()
        best = 1e9; tot = 0
        for _ in range(a.reps):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
            t = (time.perf_counter() - t0) * 1e3; best = min(best, t); tot += t
        return best, tot / a.reps
    ctx.timing_enable(True); ctx.timing_reset()
    for _ in range(a.reps):
        ctx.encode_blocks(ej, stream=None); ctx.decode_blocks(dj, stream=None)
    torch.cuda.synchronize()
    kt = ctx.timing_get(); ctx.timing_enable(False)
    ks = "  ".join("%s %.1fus x%d" % (k.replace("k_", ""), 1e3 * v["total_ms"] / max(v["launches"], 1), v["launches"] // a.reps) for k, v in kt.items() if v["launches"])
    eb, db = ctx.enc_batch(ej), ctx.dec_batch(dj)
    te = timeit(lambda: (ctx.encode_blocks_async(eb, stream=stream), ctx.encode_blocks_finish()))
    td = timeit(lambda: ctx.decode_blocks(db, stream=stream))
    pb = sum(m.nbytes for m in metas)
    alg = 4.0 * N + pb
    print("B=2^%-4.1f grid=%-5d group=%-4dMB fused=%d single=%d  enc best %.1f us avg %.1f us (%.2f TB/s alg)   dec best %.1f us avg %.1f (%.2f TB/s alg)" % (
        __import__('math').log2(B), grid, group_mb, fused, single, te[0]*1e3, te[1]*1e3, alg/te[0]/1e9, td[0]*1e3, td[1]*1e3, alg/td[0]/1e9), flush=True)
    print("      events:", ks, flush=True)

for bl in a.blocks.split(","):
    B = N if bl == "all" else int(bl[1:]) * 8192 if bl.startswith("x") else 1 << int(bl)      # "x513": blocks of 513 items of 8,192
    for gr in a.grids.split(","):
        for gp in a.groups.split(","):
            for fu in a.fused.split(","):
                for si in a.single.split(","):
                    run(B, int(gr), int(gp), int(fu), int(si))
