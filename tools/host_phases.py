#!/usr/bin/env python3
"""Where the host-inclusive step time of bench.py goes beyond the two kernels: wall time of each call of a step
(encode_blocks = enqueue + wait for the metadata; the decode-table update; decode_blocks = enqueue only) and of the
whole step, against the HIP-event time of the kernels.  Prints one JSON object."""
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
p.add_argument("--particles", type=int, default=10**9)
p.add_argument("--block", type=int, default=2**20)
p.add_argument("--reps", type=int, default=10)
a = p.parse_args()
dev = torch.device("cuda:0")
N, B = a.particles, a.block
g = torch.Generator(device=dev); g.manual_seed(0xACEDAD)
x = torch.randn(N, device=dev, dtype=torch.float32, generator=g).mul_(20.0).add_(125.0)
x = torch.remainder(x, 250.0); x[x >= 250.0] = 0.0
y = torch.empty_like(x)
nblk = (N + B - 1) // B
slot = (4 * B + 15) // 16 * 16
packed = torch.zeros(nblk * slot, dtype=torch.uint8, device=dev)
pixels = mnw.float_group_pixels(0.0, 250.0, 0.001)
ctx = mnw.Context(0)
enc = ctx.enc_batch([mnw.EncJob(x.data_ptr() + 4 * k * B, packed.data_ptr() + k * slot, min(B, N - k * B), slot, mnw.FLOAT_GROUP,
                                0.0, 250.0, pixels, 1) for k in range(nblk)])
dec = ctx.dec_batch([mnw.DecJob(packed.data_ptr() + k * slot, y.data_ptr() + 4 * k * B, min(B, N - k * B), 0, 0, 0, mnw.FLOAT_GROUP,
                                0.0, 250.0, pixels, 1, seed=1, block=k) for k in range(nblk)])
torch.cuda.synchronize()
meta = None
for _ in range(4):
    meta = ctx.encode_blocks(enc, raw=True, meta_out=meta); ctx.dec_batch_update(dec, meta); ctx.decode_blocks(dec)
torch.cuda.synchronize()
acc = {"encode_blocks": 0.0, "dec_batch_update": 0.0, "decode_blocks_enqueue": 0.0, "sync_after_decode": 0.0}
t_all = time.perf_counter()
for _ in range(a.reps):
    t0 = time.perf_counter(); meta = ctx.encode_blocks(enc, raw=True, meta_out=meta)
    t1 = time.perf_counter(); ctx.dec_batch_update(dec, meta)
    t2 = time.perf_counter(); ctx.decode_blocks(dec)
    t3 = time.perf_counter(); torch.cuda.synchronize()
    t4 = time.perf_counter()
    acc["encode_blocks"] += t1 - t0; acc["dec_batch_update"] += t2 - t1; acc["decode_blocks_enqueue"] += t3 - t2; acc["sync_after_decode"] += t4 - t3
t_sync = (time.perf_counter() - t_all) / a.reps
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    meta = ctx.encode_blocks(enc, raw=True, meta_out=meta); ctx.dec_batch_update(dec, meta); ctx.decode_blocks(dec)
torch.cuda.synchronize()
t_pipe = (time.perf_counter() - t0) / a.reps
ctx.timing_enable(True); ctx.timing_reset()
for _ in range(a.reps):
    meta = ctx.encode_blocks(enc, raw=True, meta_out=meta); ctx.dec_batch_update(dec, meta); ctx.decode_blocks(dec)
torch.cuda.synchronize()
kt = ctx.timing_get()
ks = {k: 1e3 * v["total_ms"] / v["launches"] for k, v in kt.items() if v["launches"]}
print(json.dumps({"particles": N, "block": B, "blocks": nblk,
                  "per_call_us_with_sync_each_step": {k: 1e6 * v / a.reps for k, v in acc.items()},
                  "step_us_synced_each_step": 1e6 * t_sync, "step_us_pipelined": 1e6 * t_pipe, "kernel_us": ks,
                  "host_overhead_us_pipelined": 1e6 * t_pipe - sum(ks.values())}))
