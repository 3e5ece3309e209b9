#!/usr/bin/env python3
"""Experiment: decode launch time at 1e9 elements against the size of the chunks of the tile hand-out (option dec_chunk;
0 = one contiguous range per workgroup), interleaved rounds in one process; prints the GPU's PCI bus id with it."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import minnow_amd as mnw
dev = torch.device("cuda:0")
N, B = 10**9, 2**20
nblk = (N + B - 1) // B
slot = 4 * B
g = torch.Generator(device=dev); g.manual_seed(0xACEDAD)
x = torch.empty(N, device=dev, dtype=torch.float32)
for c0 in range(0, N, 1 << 28):
    v = x[c0: c0 + (1 << 28)]
    v.normal_(0.0, 1.0, generator=g); v.mul_(20.0).add_(125.0)
    torch.remainder(v, 250.0, out=v); v[v >= 250.0] = 0.0
y = torch.empty_like(x)
packed = torch.zeros(nblk * slot, dtype=torch.uint8, device=dev)
ctx = mnw.Context(0)
enc = ctx.enc_batch([mnw.EncJob(x.data_ptr() + 4 * k * B, packed.data_ptr() + k * slot, min(B, N - k * B), slot, mnw.FLOAT_GROUP, 0.0, 250.0, 250000, 1) for k in range(nblk)])
meta = ctx.encode_blocks(enc, raw=True)
dec = ctx.dec_batch([mnw.DecJob(packed.data_ptr() + k * slot, y.data_ptr() + 4 * k * B, min(B, N - k * B), 0, 0, 0, mnw.FLOAT_GROUP, 0.0, 250.0, 250000, 1, seed=1, block=k) for k in range(nblk)])
ctx.dec_batch_update(dec, meta)
res = {"device": torch.cuda.get_device_name(0), "pci": getattr(torch.cuda.get_device_properties(0), "pci_bus_id", None), "unpack_us": {}}
chunks = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,8,16,32,64,128".split(","))]
for rnd in range(3):
    for c in chunks:
        try:
            ctx.set_option("dec_chunk", c)
        except Exception:
            pass                                   # an older build without the option
        ctx.decode_blocks(dec); torch.cuda.synchronize()
        ctx.timing_enable(True); ctx.timing_reset()
        for _ in range(4):
            ctx.encode_blocks(enc, raw=True); ctx.decode_blocks(dec)
        torch.cuda.synchronize()
        kt = ctx.timing_get(); ctx.timing_enable(False)
        res["unpack_us"].setdefault(str(c), []).append(round(1e3 * kt["k_unpack_f32"]["total_ms"] / kt["k_unpack_f32"]["launches"], 1))
        res.setdefault("encode_us", []).append(round(1e3 * kt["k_encode_reg_f32"]["total_ms"] / kt["k_encode_reg_f32"]["launches"], 1))
def ev(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / reps * 1e3, 1)
res["torch_fill_4GB_us"] = ev(lambda: y.fill_(1.5))
res["torch_sum_4GB_us"] = ev(lambda: x.sum())
res["torch_copy_4GB_us"] = ev(lambda: y.copy_(x))
print(json.dumps(res))
