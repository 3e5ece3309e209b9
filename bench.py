#!/usr/bin/env python3
"""bench.py -- minnow column quantize + bit-pack hot path on MI355X.

A step = one pass of the hot path over one synthetic float32 position column (BASELINE.json
configs[1]: 1e8 particles, FloatGroup low=0 high=250 dx=0.001 -> pixels=250000, Gaussian positions
wrapped into [0,250)): encode every block of the column and decode it back, inputs and
outputs resident in HBM.  Blocks are 2^20 elements by default -- the block size of the minp files of
configs[3] (about 1e6 particles per sub-cell, go/minp/minp.go:98-118) -- which the encode reads once
(k_encode_reg_f32); the same column in 2^24-element blocks (two-pass encode) is reported beside it.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver through torch.distributed.run (one rank per GPU); blocks shard
embarrassingly -- every rank owns its own column, no data-path collective -- so the
scaling is weak and `value` is the whole-job aggregate.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
LOW, HIGH, DX = 0.0, 250.0, 0.001


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--particles", type=int, default=10**8)
    p.add_argument("--block", type=int, default=2**20, help="elements per block (unit of GPU dispatch)")
    p.add_argument("--alt-block", type=int, default=2**24, help="second block size reported beside the headline (0: none)")
    p.add_argument("--two-pass", action="store_true", help="disable the single-read encode (diagnostic)")
    p.add_argument("--big-particles", type=int, default=10**9,
                   help="column length of the third leg (the metric's 1e9 particles); 0: none")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-kernel-timing", action="store_true")
    p.add_argument("--group-bytes", type=int, default=0)
    p.add_argument("--fused", action="store_true", help="float encode as one persistent launch (opt-in)")
    return p.parse_args()


def make_column(n, device, seed):
    """x = wrap(125 + 20*N(0,1)) into [0,250), Philox (torch CUDA generator), SURVEY 8d C2."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    x = torch.randn(n, device=device, dtype=torch.float32, generator=g)
    x.mul_(20.0).add_(125.0)
    x = torch.remainder(x, 250.0)
    x[x >= 250.0] = 0.0
    return x.contiguous()


def cpu_baseline(mnw_pixels, n_threads, target_s=12.0):
    """The oracle's LITERAL byte-wise restatement of go/bit + group.go (kind "port": the Go
    reference cannot be built in this image) timed on this host's cores, on a bounded sample of the
    same workload: each thread encodes and decodes its own blocks of 2^22 elements.  The number of
    blocks per thread is calibrated so that the sample takes about `target_s` seconds."""
    from oracle import oracle as orc

    orc.build()
    orc.lib()
    blk = 1 << 22
    rng = np.random.default_rng(0xACEDAD)
    x = np.mod(125 + 20 * rng.standard_normal(blk), 250).astype(np.float32)
    x[x >= np.float32(250.0)] = 0.0
    u = orc.dither(1, 0, blk)

    def one():
        mn, bits, data = orc.float_encode(x, LOW, HIGH, mnw_pixels, True, literal=True)
        orc.float_decode(data, mn, bits, LOW, HIGH, mnw_pixels, True, blk, u=u, literal=True)

    t0 = time.perf_counter()
    one()
    t_one = time.perf_counter() - t0
    per_thread = max(2, min(200, int(target_s / max(t_one, 1e-3))))

    def work():
        for _ in range(per_thread):
            one()

    threads = [threading.Thread(target=work) for _ in range(n_threads)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    total_bytes = 4.0 * blk * per_thread * n_threads
    return {
        "value": total_bytes / dt / 1e9, "unit": "GB/s", "cores": n_threads, "kind": "port",
        "sample": "%d threads x %d blocks of 2^22 float32 (same Gaussian column shape), encode+decode with the "
                  "literal byte-wise restatement of go/bit + group.go, %.1f s" % (n_threads, per_thread, dt),
    }


def run_config(mnw, shard, ctx, x, y, device, dist, N, B, pixels, steps, warmup, timing):
    """Encode + decode the column in blocks of B elements: `warmup` untimed steps, exactly `steps` timed ones
    (un-instrumented), then the same steps again with a HIP event pair around every kernel launch."""
    nblk = (N + B - 1) // B
    # packed output: one 16-byte aligned slot per block, worst case 32 bits per element
    slot = (4 * B + 15) // 16 * 16
    packed = torch.zeros(nblk * slot, dtype=torch.uint8, device=device)
    stream = None   # the context's own stream; torch.cuda.synchronize() below is device wide
    enc_jobs = [mnw.EncJob(x.data_ptr() + 4 * k * B, packed.data_ptr() + k * slot, min(B, N - k * B), slot,
                           mnw.FLOAT_GROUP, LOW, HIGH, pixels, 1) for k in range(nblk)]
    # the job tables are built once (a writer/reader keeps them per group); each step the decode table
    # takes (min, bits, nbytes) of every block from what THIS step's encode returned
    enc_arr = ctx.enc_batch(enc_jobs)
    dec_arr = ctx.dec_batch([mnw.DecJob(packed.data_ptr() + k * slot, y.data_ptr() + 4 * k * B, enc_jobs[k].n, 0,
                                        0, 0, mnw.FLOAT_GROUP, LOW, HIGH, pixels, 1, seed=1, block=k)
                             for k in range(nblk)])

    meta_buf = [None]

    def step():
        meta = meta_buf[0] = ctx.encode_blocks(enc_arr, stream=stream, raw=True, meta_out=meta_buf[0])   # (min, bits, nbytes) per block
        ctx.dec_batch_update(dec_arr, meta)
        ctx.decode_blocks(dec_arr, stream=stream)
        return meta

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    sync_all()      # the context has its own stream: the synthetic column must be complete first
    for _ in range(warmup):
        meta = step()
    sync_all()

    # ---- timed region: exactly K steps, no instrumentation ------------------------------------------
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        meta = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = shard.max_over_ranks(dt, device=device)

    # ---- the same K steps again with a HIP event pair around every kernel launch, recorded on the
    # stream the kernel is launched on (the context's stream).  Kept out of the region above because
    # the event records add a few microseconds per launch to the host-inclusive time.
    ktimes = {}
    if timing:
        ctx.timing_enable(True)
        ctx.timing_reset()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ktimes = ctx.timing_get()
        ctx.timing_enable(False)

    # separate, un-instrumented encode-only / decode-only timings (reported, not `value`)
    def time_only(fn, reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps

    reps = max(3, min(steps, 10))
    t_enc = time_only(lambda: ctx.encode_blocks(enc_arr, stream=stream, raw=True), reps)
    t_dec = time_only(lambda: ctx.decode_blocks(dec_arr, stream=stream), reps)

    # ---- sanity on the result of the last step (cheap, outside the timed region) -----------------
    metas = ctx.metas_of(meta)
    assert all(m.status == 0 for m in metas)
    d = (y[: 1 << 20].double() - x[: 1 << 20].double()).abs()
    d = torch.minimum(d, 250.0 - d)
    assert float(d.max()) <= DX * 1.01 + 1e-4, "round trip outside the reference tolerance"
    del packed
    return {"dt": dt, "ktimes": ktimes, "metas": metas, "t_enc": t_enc, "t_dec": t_dec, "nblk": nblk, "steps": steps}


def rooflines(r, N, timing):
    """algorithmic bytes / HIP-event kernel time, per launch.  float encode = (4 + b/8) B per element: one
    read of the block, one write of the stream; float decode = (b/8 + 4) B per element."""
    ktimes, steps = r["ktimes"], r["steps"]
    if not (timing and ktimes):
        return None, None, None
    in_bytes = 4.0 * N
    packed_bytes = float(sum(m.nbytes for m in r["metas"]))

    def avg_us(name):
        kt = ktimes.get(name)
        return 1e3 * kt["total_ms"] / kt["launches"] if kt and kt["launches"] else 0.0

    enc_alg = in_bytes + packed_bytes
    dec_alg = packed_bytes + in_bytes
    # whichever float-encode kernels the library launched for this batch: the single-read kernel, or the
    # reduce + pack pair (blocks too large to be held on chip), or the opt-in fused launch
    enc_names = [k for k in ("k_encode_reg_f32", "k_encode_f32", "k_reduce_f32", "k_pack_f32")
                 if ktimes.get(k, {}).get("launches")]
    enc_ms = sum(ktimes[k]["total_ms"] for k in enc_names) / steps
    dec_ms = ktimes["k_unpack_f32"]["total_ms"] / steps
    enc_ach = enc_alg / (enc_ms * 1e-3) / 1e9
    dec_ach = dec_alg / (dec_ms * 1e-3) / 1e9
    launches_per_step = ktimes[enc_names[-1]]["launches"] / steps
    enc = {
        "bound": "hbm", "kernel": "+".join(enc_names) + " (float encode)",
        "achieved": enc_ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": enc_ach / HBM_PEAK_GBPS,
        "traffic": None,
        "algorithmic_bytes_per_elt": enc_alg / N,
        "elements_per_launch": N / launches_per_step,
        "avg_launch_us": {k: avg_us(k) for k in enc_names},
        "note": "achieved = (4 + b/8) B/elt x elements per batch / summed HIP-event time of the encode launches",
    }
    dec = {
        "bound": "hbm", "kernel": "k_unpack_f32", "achieved": dec_ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": dec_ach / HBM_PEAK_GBPS, "traffic": None, "algorithmic_bytes_per_elt": dec_alg / N,
        "avg_launch_us": avg_us("k_unpack_f32"),
    }
    kernels = [{"kernel": k, "launches": v["launches"], "avg_us": 1e3 * v["total_ms"] / v["launches"]}
               for k, v in ktimes.items() if v["launches"]]
    return enc, dec, kernels


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    import minnow_amd as mnw
    from minnow_amd import build as mnw_build
    from minnow_amd import shard

    if rank == 0:
        mnw_build.build()
    if dist is not None:
        dist.barrier()
    if mnw.device_count() < 1:
        raise RuntimeError("bench.py needs a GPU: libminnow_hip has no CPU fallback")

    pixels = mnw.float_group_pixels(LOW, HIGH, DX)
    N, B = a.particles, a.block
    x = make_column(N, device, 0xACEDAD + rank)
    y = torch.empty_like(x)
    ctx = mnw.Context(local_rank)
    if a.group_bytes:
        ctx.set_option("group_bytes", a.group_bytes)
    if a.fused:
        ctx.set_option("fused", 1)
    if a.two_pass:
        ctx.set_option("single_read", 0)
    timing = not a.no_kernel_timing

    r = run_config(mnw, shard, ctx, x, y, device, dist, N, B, pixels, a.steps, a.warmup, timing)
    # the same column in blocks too large to be held on chip (two-pass encode): reported beside the headline
    r_big = None
    if a.alt_block and a.alt_block != B and world == 1:
        r_big = run_config(mnw, shard, ctx, x, y, device, dist, N, a.alt_block, pixels, min(a.steps, 10), 2, timing)

    # the same column shape at the length BASELINE.json's metric and target are quoted on (10^9 elements)
    r_1e9 = None
    if a.big_particles and a.big_particles > N and world == 1:
        del x, y
        torch.cuda.empty_cache()
        xb = make_column(a.big_particles, device, 0xACEDAD + rank)
        yb = torch.empty_like(xb)
        r_1e9 = run_config(mnw, shard, ctx, xb, yb, device, dist, a.big_particles, B, pixels, 5, 2, timing)
        del xb, yb

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    metas, dt = r["metas"], r["dt"]
    bits = sorted(set(m.bits for m in metas))
    in_bytes = 4.0 * N
    packed_bytes = float(sum(m.nbytes for m in metas))
    ms_per_step = dt / a.steps * 1e3
    value = world * in_bytes / (dt / a.steps) / 1e9
    out = {
        "metric": "encode+decode GB/s (input)", "value": value, "unit": "GB/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32->u32 fixed point (f32 sub/div, integer pack)", "data": "synthetic",
        "config": {"workload": "configs[1]: single float32 position column, %d particles, FloatGroup "
                               "low=0 high=250 dx=0.001 (pixels=%d), blocks of %d (the block size of the minp "
                               "files of configs[3]: ~1e6 particles per sub-cell), encode+decode round trip"
                               % (N, pixels, B),
                   "blocks_per_gpu": r["nblk"], "bits": bits, "parallelism": "blocks sharded per GPU, no collective"},
        "pct_hbm_peak": 100.0 * (in_bytes * 2 + 2 * packed_bytes) / (dt / a.steps) / 1e9 / HBM_PEAK_GBPS,
        "encode_GBps_input": in_bytes / r["t_enc"] / 1e9, "decode_GBps_output": in_bytes / r["t_dec"] / 1e9,
        "compressed_bits_per_elt": 8.0 * packed_bytes / N,
    }

    enc, dec, kernels = rooflines(r, N, timing)
    if enc:
        out["roofline"], out["roofline_decode"], out["kernels"] = enc, dec, kernels
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            # HBM bytes per launch from rocprofv3 --pmc passes (collected separately; see profiles/README.md)
            try:
                with open(pmc) as f:
                    t = json.load(f)
                out["roofline"]["traffic"] = t.get("encode_bytes_per_batch")
                out["roofline_decode"]["traffic"] = t.get("decode_bytes_per_batch")
                out["roofline"]["traffic_source"] = t.get("source")
            except Exception:
                pass
    if r_big:
        enc_b, dec_b, _ = rooflines(r_big, N, timing)
        out["large_blocks"] = {
            "block": a.alt_block, "steps": r_big["steps"], "ms_per_step": r_big["dt"] / r_big["steps"] * 1e3,
            "value": in_bytes / (r_big["dt"] / r_big["steps"]) / 1e9, "unit": "GB/s",
            "roofline": enc_b, "roofline_decode": dec_b,
            "note": "same column, blocks of %d elements: more than the chip can hold, so the encode reads each "
                    "block twice (reduce + pack launches)" % a.alt_block,
        }

    if r_1e9:
        nb = a.big_particles
        enc_b, dec_b, _ = rooflines(r_1e9, nb, timing)
        out["column_1e9"] = {
            "particles": nb, "block": B, "steps": r_1e9["steps"], "ms_per_step": r_1e9["dt"] / r_1e9["steps"] * 1e3,
            "value": 4.0 * nb / (r_1e9["dt"] / r_1e9["steps"]) / 1e9, "unit": "GB/s",
            "roofline": enc_b, "roofline_decode": dec_b,
            "note": "same column shape and block size at %d particles (the length the metric and the 60%% target are "
                    "quoted on): launch ramp-up and tail amortise over ten times the work" % nb,
        }

    if world == 1 and not a.no_cpu_baseline:
        ncpu = os.cpu_count() or 1
        out["cpu_baseline"] = cpu_baseline(pixels, max(1, min(16, ncpu)))

    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
