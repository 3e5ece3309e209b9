#!/usr/bin/env python3
"""Experiment (not product): does the time of the 1e9-element decode / encode launch depend on WHERE its buffers were
allocated?  Runs of the same build have given 1.11-1.15 ms and 1.27-1.30 ms for k_unpack_f32.  Times the launches (HIP
events) with the output / stream buffers obtained in different ways, twice each, interleaved."""
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
pixels = 250000
res = {"device": torch.cuda.get_device_name(0), "pci": getattr(torch.cuda.get_device_properties(0), "pci_bus_id", None)}
ctx = mnw.Context(0)


def raw(nbytes):
    p = C.c_void_p()
    assert ctx._L.mnw_dev_alloc(ctx._h, int(nbytes), C.byref(p)) == 0
    return p.value


MiB = 1 << 20
y_torch = torch.empty(N, dtype=torch.float32, device=dev)
ys = {
    "torch": y_torch.data_ptr(),
    "raw 4e9": raw(4 * N),
    "raw rounded to 2MiB": raw((4 * N + 2 * MiB - 1) // (2 * MiB) * (2 * MiB)),
    "raw 5GiB +0": None, "raw 5GiB +256B": None, "raw 5GiB +1MiB": None, "raw 5GiB +333MiB": None,
}
big = raw(5 << 30)
ys["raw 5GiB +0"], ys["raw 5GiB +256B"], ys["raw 5GiB +1MiB"], ys["raw 5GiB +333MiB"] = big, big + 256, big + MiB, big + 333 * MiB
slots = {"4MiB": 4 * B, "4MiB+16KiB": 4 * B + 16384, "3MiB": 3 * B}
packs = {k: torch.zeros(nblk * v, dtype=torch.uint8, device=dev) for k, v in slots.items()}
g = torch.Generator(device=dev); g.manual_seed(0xACEDAD)
x = torch.empty(N, device=dev, dtype=torch.float32)
for c0 in range(0, N, 1 << 28):
    v = x[c0: c0 + (1 << 28)]
    v.normal_(0.0, 1.0, generator=g); v.mul_(20.0).add_(125.0)
    torch.remainder(v, 250.0, out=v); v[v >= 250.0] = 0.0
torch.cuda.synchronize()


def run(tag, y_ptr, pk, reps=4):
    slot, packed = slots[pk], packs[pk]
    enc = ctx.enc_batch([mnw.EncJob(x.data_ptr() + 4 * k * B, packed.data_ptr() + k * slot, min(B, N - k * B), slot, mnw.FLOAT_GROUP,
                                    0.0, 250.0, pixels, 1) for k in range(nblk)])
    meta = ctx.encode_blocks(enc, raw=True)
    dec = ctx.dec_batch([mnw.DecJob(packed.data_ptr() + k * slot, y_ptr + 4 * k * B, min(B, N - k * B), 0, 0, 0, mnw.FLOAT_GROUP,
                                    0.0, 250.0, pixels, 1, seed=1, block=k) for k in range(nblk)])
    ctx.dec_batch_update(dec, meta)
    ctx.encode_blocks(enc, raw=True); ctx.decode_blocks(dec)
    torch.cuda.synchronize()
    ctx.timing_enable(True); ctx.timing_reset()
    for _ in range(reps):
        ctx.encode_blocks(enc, raw=True); ctx.decode_blocks(dec)
    torch.cuda.synchronize()
    kt = ctx.timing_get(); ctx.timing_enable(False)
    r = res.setdefault(tag, {"y_ptr": hex(y_ptr), "packed_ptr": hex(packed.data_ptr()), "unpack_us": [], "encode_us": []})
    r["unpack_us"].append(round(1e3 * kt["k_unpack_f32"]["total_ms"] / kt["k_unpack_f32"]["launches"], 1))
    r["encode_us"].append(round(1e3 * kt["k_encode_reg_f32"]["total_ms"] / kt["k_encode_reg_f32"]["launches"], 1))


for rnd in range(2):
    for yk, yp in ys.items():
        run("y %s | slot 4MiB" % yk, yp, "4MiB")
    for pk in slots:
        run("y torch | slot %s" % pk, ys["torch"], pk)
print(json.dumps(res, indent=1))
