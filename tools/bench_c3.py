#!/usr/bin/env python3
"""BASELINE configs[2] shape on one GPU: a halo catalogue of 28 codec columns x ROWS rows (3 position, 3 velocity,
18 log/linear FloatGroup, 4 IntGroup; the 4 fixed-size raw-copy columns of SURVEY 8d C3 are not codec work), one
group per (block of rows, column) as minh.Writer.Block lays them out (go/minh/minh.go:115-137).  Encodes and
decodes the whole catalogue, one batch per block of rows, and prints one JSON object."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import minnow_amd as mnw  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--rows", type=int, default=10**8)
p.add_argument("--block", type=int, default=2**20)
p.add_argument("--reps", type=int, default=3)
p.add_argument("--blocks-per-call", type=int, default=1, help="row blocks handed to the library per call")
p.add_argument("--opt", action="append", default=[], help="library option name=value (mnw_set_option), repeatable")
p.add_argument("--in-jobs", action="store_true", help="with --minh: pass the transforms as job flags (mnw_enc_job.xform) instead of separate calls")
p.add_argument("--minh", action="store_true", help="include minh's processFloatGroup before the encode (the 18 log columns start as raw\n"
               "values, log10 + clamp on the GPU) and 10^x after the decode")
a = p.parse_args()
dev = torch.device("cuda:0")
R, B = a.rows, a.block
g = torch.Generator(device=dev); g.manual_seed(0xACEDAD)
cols = []   # (kind, tensor, (low, high, dx))
for c in range(3):
    x = torch.remainder(125 + 20 * torch.randn(R, device=dev, generator=g), 250.0); x[x >= 250.0] = 0.0
    cols.append(("f", x.contiguous(), (0.0, 250.0, 0.001)))
for c in range(3):
    cols.append(("f", torch.clamp(300 * torch.randn(R, device=dev, generator=g), -4999, 4999).contiguous(), (-5000.0, 5000.0, 1.0)))
for c in range(18):
    lg = torch.clamp(10 + 0.2 * c + 0.8 * torch.randn(R, device=dev, generator=g), 8.001, 15.99).contiguous()
    cols.append(("L" if a.minh else "f", (10.0 ** lg.double()).float().contiguous() if a.minh else lg, (8.0, 16.0, 0.01)))
cols.append(("i", (5 * 10**9 + torch.arange(R, device=dev, dtype=torch.int64)), None))
cols.append(("i", torch.poisson(torch.full((R,), 40.0, device=dev), generator=g).to(torch.int64), None))
cols.append(("i", torch.randint(-1, 2**20, (R,), device=dev, dtype=torch.int64, generator=g), None))
cols.append(("i", torch.full((R,), 7, device=dev, dtype=torch.int64), None))
nblk = (R + B - 1) // B
ctx = mnw.Context(0)
for kv in a.opt:
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
in_bytes = sum(t.numel() * t.element_size() for _, t, _ in cols)
outs, decs, enc_batches, dec_protos = [], [], [], []
# minh.Writer.Block's buffer (minh.buf): the processed copy of every FloatGroup column, encoded from there
bufs = [torch.empty_like(t) if (a.minh and not a.in_jobs and kind != "i") else None for kind, t, _ in cols]
proc_cols = [(t.data_ptr(), bufs[ci].data_ptr(), R, lim[0], lim[1], 1 if kind == "L" else 0) for ci, (kind, t, lim) in enumerate(cols) if bufs[ci] is not None]
unlog_cols = [(ci, kind) for ci, (kind, t, lim) in enumerate(cols) if kind == "L"]
for kind, t, lim in cols:
    esz = t.element_size()
    slot = (esz * B + 15) // 16 * 16
    outs.append((torch.zeros(nblk * slot, dtype=torch.uint8, device=dev), slot))
    decs.append(torch.empty_like(t))
ej, dj = [], []
for b in range(nblk):
    n = min(B, R - b * B)
    for ci, (kind, t, lim) in enumerate(cols):
        o, slot = outs[ci]
        esz = t.element_size()
        if kind != "i":
            low, high, dx = lim
            pix = mnw.float_group_pixels(low, high, dx)
            src = bufs[ci] if bufs[ci] is not None else t
            xf = ((mnw.XF_LOG10 if kind == "L" else 0) | mnw.XF_CLAMP) if (a.minh and a.in_jobs) else 0
            ej.append(mnw.EncJob(src.data_ptr() + esz * b * B, o.data_ptr() + b * slot, n, slot, mnw.FLOAT_GROUP, low, high, pix, 1, xform=xf))
            dj.append(mnw.DecJob(o.data_ptr() + b * slot, decs[ci].data_ptr() + esz * b * B, n, 0, 0, 0, mnw.FLOAT_GROUP, low, high, pix, 1, seed=1, block=b * len(cols) + ci, xform=xf & mnw.XF_LOG10))
        else:
            ej.append(mnw.EncJob(t.data_ptr() + esz * b * B, o.data_ptr() + b * slot, n, slot, mnw.INT_GROUP))
            dj.append(mnw.DecJob(o.data_ptr() + b * slot, decs[ci].data_ptr() + esz * b * B, n, 0, 0, 0, mnw.INT_GROUP))
    if (b + 1) % a.blocks_per_call == 0 or b == nblk - 1:
        enc_batches.append(ctx.enc_batch(ej))
        dec_protos.append(ctx.dec_batch(dj))
        ej, dj = [], []
torch.cuda.synchronize()


def encode_all():
    metas = []
    if a.minh and not a.in_jobs:
        ctx.minh_process(proc_cols)
    for eb in enc_batches:
        metas.append(ctx.encode_blocks(eb, raw=True))
    return metas


def decode_all(metas):
    for db, m in zip(dec_protos, metas):
        ctx.dec_batch_update(db, m)
        ctx.decode_blocks(db)
    if a.minh and not a.in_jobs:
        ctx.minh_unlog([(decs[ci].data_ptr(), R) for ci, _ in unlog_cols])


for _ in range(4):
    metas = encode_all(); decode_all(metas)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    metas = encode_all()
torch.cuda.synchronize()
t_enc = (time.perf_counter() - t0) / a.reps
t0 = time.perf_counter()
for _ in range(a.reps):
    decode_all(metas)
torch.cuda.synchronize()
t_dec = (time.perf_counter() - t0) / a.reps
packed = sum(m.nbytes for mb in metas for m in ctx.metas_of(mb))
for ci, (kind, t, lim) in enumerate(cols):
    if kind == "i":
        assert torch.equal(decs[ci], t), ci
    elif kind == "L":
        d = (torch.log10(decs[ci].double()) - torch.log10(t.double())).abs()
        assert float(d.max()) <= lim[2] * 1.01 + 1e-3, (ci, float(d.max()))
    else:
        d = (decs[ci].double() - t.double()).abs()
        if lim[0] == 0.0:
            d = torch.minimum(d, 250.0 - d)
        assert float(d.max()) <= lim[2] * 1.01 + 1e-3, (ci, float(d.max()))
ctx.timing_enable(True); ctx.timing_reset()
metas = encode_all(); decode_all(metas); torch.cuda.synchronize()
kt = {k: v for k, v in ctx.timing_get().items() if v["launches"]}
print(json.dumps({
    "rows": R, "block": B, "minh_transforms": ("in jobs" if a.in_jobs else "separate calls") if a.minh else False, "blocks_per_call": a.blocks_per_call, "columns": len(cols), "input_GB": in_bytes / 1e9, "packed_GB": packed / 1e9,
    "encode_ms": t_enc * 1e3, "decode_ms": t_dec * 1e3,
    "encode_GBps_input": in_bytes / t_enc / 1e9, "decode_GBps_output": in_bytes / t_dec / 1e9,
    "encode_frac_of_8TBps": (in_bytes + packed) / t_enc / 8e12, "decode_frac_of_8TBps": (in_bytes + packed) / t_dec / 8e12,
    "kernel_ms_total": {k: v["total_ms"] for k, v in kt.items()},
}))
