#!/usr/bin/env python3
"""Experiment (not product): does the time of the 1e9-element decode / encode launch depend on WHERE its buffers were
allocated?  Runs of the same build on the same box have given 1.15 ms and 1.27-1.30 ms for k_unpack_f32.  Times the
launches (HIP events) for output / stream / input buffers obtained in different ways."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import minnow_amd as mnw  # noqa: E402

dev = torch.device("cuda:0")
N, B = 10**9, 2**20
nblk = (N + B - 1) // B
slot = 4 * B
pixels = 250000
res = {}
early_y = torch.empty(N, dtype=torch.float32, device=dev)            # before anything else exists
early_p = torch.zeros(nblk * slot, dtype=torch.uint8, device=dev)
ctx = mnw.Context(0)
hip_y = C.c_void_p()
assert ctx._L.mnw_dev_alloc(ctx._h, 4 * N, C.byref(hip_y)) == 0      # raw hipMalloc, outside torch's allocator


def column(chunk=1 << 28):
    g = torch.Generator(device=dev); g.manual_seed(0xACEDAD)
    x = torch.empty(N, device=dev, dtype=torch.float32)
    for c0 in range(0, N, chunk):
        v = x[c0: c0 + chunk]
        v.normal_(0.0, 1.0, generator=g); v.mul_(20.0).add_(125.0)
        torch.remainder(v, 250.0, out=v); v[v >= 250.0] = 0.0
    return x


x = column()
# churn: many allocations of odd sizes, freed in another order (stay in torch's cache)
junk = [torch.empty(int(s), dtype=torch.uint8, device=dev) for s in (3e8, 7e7, 1.3e9, 5e6, 9e8, 2.1e8)]
del junk[1], junk[2]
late_y = torch.empty(N, dtype=torch.float32, device=dev)
late_p = torch.zeros(nblk * slot, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()


def run(tag, y_ptr, packed, reps=6):
    enc = ctx.enc_batch([mnw.EncJob(x.data_ptr() + 4 * k * B, packed.data_ptr() + k * slot, min(B, N - k * B), slot, mnw.FLOAT_GROUP,
                                    0.0, 250.0, pixels, 1) for k in range(nblk)])
    meta = ctx.encode_blocks(enc, raw=True)
    dec = ctx.dec_batch([mnw.DecJob(packed.data_ptr() + k * slot, y_ptr + 4 * k * B, min(B, N - k * B), 0, 0, 0, mnw.FLOAT_GROUP,
                                    0.0, 250.0, pixels, 1, seed=1, block=k) for k in range(nblk)])
    ctx.dec_batch_update(dec, meta)
    for _ in range(2):
        ctx.encode_blocks(enc, raw=True); ctx.decode_blocks(dec)
    torch.cuda.synchronize()
    ctx.timing_enable(True); ctx.timing_reset()
    for _ in range(reps):
        ctx.encode_blocks(enc, raw=True); ctx.decode_blocks(dec)
    torch.cuda.synchronize()
    kt = ctx.timing_get(); ctx.timing_enable(False)
    res[tag] = {k: round(1e3 * v["total_ms"] / v["launches"], 1) for k, v in kt.items() if v["launches"]}
    res[tag]["y_ptr"] = hex(y_ptr); res[tag]["packed_ptr"] = hex(packed.data_ptr())


run("y early, packed early", early_y.data_ptr(), early_p)
run("y late,  packed early", late_y.data_ptr(), early_p)
run("y early, packed late", early_y.data_ptr(), late_p)
run("y late,  packed late", late_y.data_ptr(), late_p)
run("y raw hipMalloc, packed early", hip_y.value, early_p)
run("y early, packed early (again)", early_y.data_ptr(), early_p)
res["x_ptr"] = hex(x.data_ptr())
print(json.dumps(res, indent=1))
