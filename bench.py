#!/usr/bin/env python3
"""bench.py -- minnow column quantize + bit-pack hot path on MI355X.

A step = one pass of the hot path over one synthetic float32 position column of 1e9 particles -- the
configuration BASELINE.json's metric is quoted on ("encode+decode GB/s (input) ... 1e9 float32 particles"); 4 GB
of input fits one GPU many times over -- FloatGroup low=0 high=250 dx=0.001 -> pixels=250000, Gaussian positions
wrapped into [0,250) (SURVEY.md 8d C2 at the metric's length): encode every block of the column and decode it
back, inputs and outputs resident in HBM.  Blocks are 2^20 elements (the block size of the minp files of
configs[3]: about 1e6 particles per sub-cell, go/minp/minp.go:98-118).  Reported beside the headline, as extra
keys of the same JSON line: the same column in 2^24-element blocks, configs[1] (the first 1e8 particles, both
block sizes) and one minp file cell of configs[3] (500^3 particles, 375 blocks per component).

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver through torch.distributed.run (one rank per GPU).  The column is cut into blocks
and the blocks are dealt to the ranks in contiguous ranges (minnow_amd.shard.shard_range; a block's (min, bits,
bytes) depends on that block only, go/group.go:242-255, 312-327): no data-path collective.  Weak scaling (the
default): the global column has N x 1e9 particles, cut into blocks and the BLOCKS dealt evenly -- a rank owns 954
or 953 blocks of 2^20 (1,000,341,504 or 999,292,928 particles), the last rank the ragged last block; every rank's
own element count is reported (`per_rank`) and the per-GPU fractions are computed from the slowest rank's.
--strong: the global column has 1e9 particles whatever N is.  After the timed steps the ranks' (min, bits, nbytes) are gathered in block order and
prefix-summed into the file offsets of go/block_index.go:16-35 -- a few integers per block, the only thing that
ever crosses ranks.  `value` is the whole-job aggregate; rank 0 prints ONE JSON line.
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
PROFILE_TAG = "r03"             # profiles/<tag>/pmc_traffic.json holds the PMC traffic of this workload


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--particles", type=int, default=10**9, help="elements per rank (weak scaling) or in all (--strong)")
    p.add_argument("--block", type=int, default=2**20, help="elements per block (unit of GPU dispatch)")
    p.add_argument("--alt-block", type=int, default=2**24, help="second block size reported beside the headline (0: none)")
    p.add_argument("--small-particles", type=int, default=10**8, help="configs[1] leg: column length (0: none)")
    p.add_argument("--strong", action="store_true", help="N ranks share ONE column of --particles elements")
    p.add_argument("--legs", choices=("all", "headline"), default="all", help="headline: only the metric's configuration")
    p.add_argument("--no-minp", action="store_true", help="skip the configs[3] file-cell leg")
    p.add_argument("--two-pass", action="store_true", help="disable the single-read encode (diagnostic)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-kernel-timing", action="store_true")
    p.add_argument("--no-stream-ceiling", action="store_true", help="do not run tools/microbench/rw_mix first (profiler passes)")
    p.add_argument("--group-bytes", type=int, default=0)
    p.add_argument("--fused", action="store_true", help="float encode as one persistent launch (opt-in)")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo: rehearsal of the "
                                                     "multi-rank path on a box with fewer GPUs than ranks -- ranks then share GPUs)")
    return p.parse_args()


def make_column(n, device, seed, chunk=1 << 28):
    """x = wrap(125 + 20*N(0,1)) into [0,250), Philox (torch CUDA generator); made in chunks so that
    the temporaries stay small beside a 4 GB column."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    x = torch.empty(n, device=device, dtype=torch.float32)
    for c0 in range(0, n, chunk):
        v = x[c0: c0 + chunk]
        v.normal_(0.0, 1.0, generator=g)
        v.mul_(20.0).add_(125.0)
        torch.remainder(v, 250.0, out=v)
        v[v >= 250.0] = 0.0
    return x


def host_cores():
    """(logical CPUs, physical cores) of this host, from /proc/cpuinfo"""
    logical = os.cpu_count() or 1
    phys = set()
    try:
        pid = cid = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    pid = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    cid = line.split(":")[1].strip()
                elif not line.strip():
                    if pid is not None and cid is not None:
                        phys.add((pid, cid))
                    pid = cid = None
    except OSError:
        pass
    return logical, (len(phys) or None)


def cpu_quota():
    """CPUs this process may use at once: the cgroup's quota if there is one (the GPU boxes give a job a share of the
    host), else the affinity mask"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
        if q != "max":
            return max(1, int(float(q) / float(per) + 0.5))
    except (OSError, ValueError):
        pass
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(mnw_pixels, target_s=5.0):
    """The oracle's LITERAL byte-wise restatement of go/bit + group.go (kind "port": the Go reference cannot be built
    in this image) timed on this host, on a bounded sample of the same workload: every thread encodes and decodes its
    own blocks of 2^22 elements.  Two figures, as BASELINE.md section 3 asks: ONE thread, and ALL the cores this job
    may use (the host's physical cores, capped by the job's CPU quota: a GPU box gives a one-GPU job a share of the
    host, not the host).  About `target_s` seconds each."""
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

    def run(n_threads):
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
        return 4.0 * blk * per_thread * n_threads / dt / 1e9, dt

    logical, phys = host_cores()
    quota = cpu_quota()
    n_all = max(1, min(phys or logical, quota, 256))
    v1, dt1 = run(1)
    va, dta = run(n_all)
    return {
        "value": va, "unit": "GB/s", "cores": n_all, "threads": n_all, "kind": "port",
        "one_thread": {"value": v1, "unit": "GB/s", "cores": 1, "threads": 1},
        "host_logical_cpus": logical, "host_physical_cores": phys, "cpu_quota": quota,
        "sample": "the literal byte-wise restatement of go/bit + group.go, encode+decode of blocks of 2^22 float32 (the same "
                  "Gaussian column shape): 1 thread x %d blocks in %.1f s, then %d threads (host: %d logical CPUs, %s physical "
                  "cores; this job may use %d) x %d blocks each in %.1f s"
                  % (per_thread, dt1, n_all, logical, phys if phys else "?", quota, per_thread, dta),
    }


def stream_ceiling():
    """What a bare stream of the kernels' traffic mix reaches on THIS GPU (tools/microbench/rw_mix: one 16-byte access per
    thread, no arithmetic), run as a child process before the timed steps.  None if the binary is not there."""
    import subprocess

    exe = os.path.join(ROOT, "tools", "microbench", "rw_mix")
    if not os.path.exists(exe):
        return None
    try:
        o = subprocess.run([exe, "4", "json"], capture_output=True, text=True, timeout=120)
        for line in o.stdout.splitlines():
            if line.startswith("{"):
                return json.loads(line)
    except Exception:
        pass
    return None


def run_config(mnw, shard, ctx, x, y, device, dist, N, B, pixels, steps, warmup, timing, block0=0, across_ranks=True, cdev=None):
    """Encode + decode the elements [0, N) of x in blocks of B elements: `warmup` untimed steps, exactly `steps` timed
    ones (un-instrumented), then the same steps again with a HIP event pair around every kernel launch.  block0: index
    of the first block in the global column (the dither key of a block is its global index)."""
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
                                        0, 0, mnw.FLOAT_GROUP, LOW, HIGH, pixels, 1, seed=1, block=block0 + k)
                             for k in range(nblk)])

    meta_buf = [None]

    def step():
        meta = meta_buf[0] = ctx.encode_blocks(enc_arr, stream=stream, raw=True, meta_out=meta_buf[0])   # (min, bits, nbytes) per block
        ctx.dec_batch_update(dec_arr, meta)
        ctx.decode_blocks(dec_arr, stream=stream)
        return meta

    def sync_all():
        if dist is not None and across_ranks:
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
    if dist is not None and across_ranks:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if across_ranks:
        dt = shard.max_over_ranks(dt, device=cdev)

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

    # ---- sanity on the result of the last step (cheap, outside the timed region): the reference's own criterion
    # (|x - x'| <= dx on the ring, go/minnow_test.go:328-334) on the first and the last block.  The bit-exact
    # comparison of this very launch with the oracle lives in tests/test_parity_gpu.py (headline tests).
    metas = ctx.metas_of(meta)
    assert all(m.status == 0 for m in metas)
    for lo in (0, (nblk - 1) * B):
        hi = min(lo + B, N)
        d = (y[lo:hi].double() - x[lo:hi].double()).abs()
        d = torch.minimum(d, 250.0 - d)
        assert float(d.max()) <= DX * 1.01 + 1e-4, "round trip outside the reference tolerance"
    del packed
    return {"dt": dt, "ktimes": ktimes, "metas": metas, "t_enc": t_enc, "t_dec": t_dec, "nblk": nblk, "steps": steps,
            "N": N, "B": B}


def rooflines(r, timing):
    """algorithmic bytes / HIP-event kernel time, per launch.  float encode = (4 + b/8) B per element: one
    read of the block, one write of the stream; float decode = (b/8 + 4) B per element."""
    ktimes, steps, N = r["ktimes"], r["steps"], r["N"]
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
        "algorithmic_bytes_per_launch": enc_alg / launches_per_step,
        "elements_per_launch": N / launches_per_step,
        "avg_launch_us": {k: avg_us(k) for k in enc_names},
        "note": "achieved = (4 + b/8) B/elt x elements per batch / summed HIP-event time of the encode launches",
    }
    dec = {
        "bound": "hbm", "kernel": "k_unpack_f32 (float decode)", "achieved": dec_ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": dec_ach / HBM_PEAK_GBPS, "traffic": None, "algorithmic_bytes_per_elt": dec_alg / N,
        "algorithmic_bytes_per_launch": dec_alg, "elements_per_launch": float(N),
        "avg_launch_us": avg_us("k_unpack_f32"),
        "note": "achieved = (b/8 + 4) B/elt x elements per batch / HIP-event time of the decode launch",
    }
    kernels = [{"kernel": k, "launches": v["launches"], "avg_us": 1e3 * v["total_ms"] / v["launches"]}
               for k, v in ktimes.items() if v["launches"]]
    return enc, dec, kernels


def leg_record(r, timing, note):
    enc, dec, _ = rooflines(r, timing)
    N = r["N"]
    return {
        "particles": N, "block": r["B"], "blocks": r["nblk"], "steps": r["steps"],
        "ms_per_step": r["dt"] / r["steps"] * 1e3, "value": 4.0 * N / (r["dt"] / r["steps"]) / 1e9, "unit": "GB/s",
        "encode_GBps_input": 4.0 * N / r["t_enc"] / 1e9, "decode_GBps_output": 4.0 * N / r["t_dec"] / 1e9,
        "roofline_encode": enc, "roofline_decode": dec, "note": note,
    }


def minp_leg(mnw, ctx, device, reps, timing, velocities=False):
    """configs[3], one rank's share: one minp file cell of 500^3 particles resident on the GPU, sub_cells = 5 -> 375
    blocks of 1e6 per component (go/minp/minp.go:98-118): mnw_vectors_encode + mnw_vectors_decode.  Positions: the Lagrangian
    grid + N(0, 2) wrapped into [0, 250), dx = 0.001, periodic; velocities: N(0, 300), dv = 1, limits from the data
    (SURVEY.md 8d C4)."""
    from minnow_amd import minp

    nf, S, L, dx = 500, 5, 250.0, (1.0 if velocities else 0.001)
    periodic = not velocities
    n = nf ** 3
    g = torch.Generator(device=device)
    g.manual_seed(0xACEDAD + (78 if velocities else 77))
    if velocities:
        pos = torch.randn((n, 3), device=device, generator=g, dtype=torch.float32).mul_(300.0).contiguous()
    else:
        idx = torch.arange(n, device=device)
        pos = torch.stack([idx % nf, (idx // nf) % nf, idx // (nf * nf)], dim=1).to(torch.float32) * (L / nf)
        del idx
        pos.add_(torch.randn(pos.shape, device=device, generator=g).mul_(2.0))
        pos = torch.remainder(pos, L).contiguous()
        pos[pos >= L] = 0.0
    n_sub = nf // S
    nb = 3 * S ** 3
    slot = (4 * n_sub ** 3 + 15) // 16 * 16
    packed = torch.zeros(nb * slot, dtype=torch.uint8, device=device)
    out = torch.empty_like(pos)
    torch.cuda.synchronize()
    state = {}

    def enc():
        state["m"] = minp.vectors_encode(ctx, pos.data_ptr(), nf, S, dx, periodic, L, packed.data_ptr(), slot)

    def dec():
        metas, low, high, pixels = state["m"]
        minp.vectors_decode(ctx, packed.data_ptr(), slot, metas, nf, S, low, high, pixels, periodic, L, out.data_ptr(), seed=1)

    def both():
        enc()
        dec()

    for _ in range(4):
        both()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        both()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    kt = {}
    if timing:
        ctx.timing_enable(True)
        ctx.timing_reset()
        for _ in range(reps):
            both()
        torch.cuda.synchronize()
        kt = {k: 1e3 * v["total_ms"] / v["launches"] for k, v in ctx.timing_get().items() if v["launches"]}
        ctx.timing_enable(False)
    metas = state["m"][0]
    pb = float(sum(m.nbytes for m in metas))
    d = (out.double() - pos.double()).abs()
    if periodic:
        d = torch.minimum(d, L - d)
        assert float(d.max()) <= dx * 1.01 + 1e-4
    else:
        # (the reference's quirk: a component's maximum can quantize to `pixels` and is read back at the bottom of the range)
        assert int((d > dx * 1.01 + 1e-3).sum()) <= 3
    alg = 12.0 * n + pb
    enc_us = sum(v for k, v in kt.items() if k in ("k_encode_reg_vec3", "k_reduce_vec3", "k_pack_vec3", "k_subcell_gather", "k_encode_reg_f32", "k_reduce_f32", "k_pack_f32"))
    dec_us = sum(v for k, v in kt.items() if k in ("k_unpack_vec3", "k_unpack_f32", "k_subcell_scatter"))
    rec = {"what": "velocities N(0,300), dv = 1, data-derived limits" if velocities else "positions, dx = 0.001, periodic",
           "n_file": nf, "sub_cells": S, "particles": n, "blocks": nb, "bits": sorted(set(m.bits for m in metas)),
           "ms_per_step": dt * 1e3, "value": 12.0 * n / dt / 1e9, "unit": "GB/s of input (12 B per particle), encode + decode",
           "kernel_us": kt, "algorithmic_bytes_per_pass": alg}
    if enc_us and dec_us:
        rec["encode_frac_of_8TBps"] = alg / (enc_us * 1e-6) / 1e9 / HBM_PEAK_GBPS
        rec["decode_frac_of_8TBps"] = alg / (dec_us * 1e-6) / 1e9 / HBM_PEAK_GBPS
    del pos, out, packed
    return rec


def catalogue_leg(mnw, ctx, device, rows, reps, timing):
    """configs[2] shape on one rank: a halo catalogue of `rows` rows and 28 codec columns (6 position / velocity FloatGroups, 18
    linear-in-log FloatGroups, 4 IntGroups: ids, counts, flags, a constant), one group per (block of 2^20 rows, column) as
    minh.Writer.Block makes them (go/minh/minh.go:112-137), 16 row blocks per library call.  Encode of everything, decode of
    everything; round trip checked on every column."""
    B = 2 ** 20
    g = torch.Generator(device=device)
    g.manual_seed(0xACEDAD + 3)
    cols = []
    for _ in range(3):
        x = torch.remainder(125 + 20 * torch.randn(rows, device=device, generator=g), 250.0)
        x[x >= 250.0] = 0.0
        cols.append(("f", x.contiguous(), (0.0, 250.0, 0.001)))
    for _ in range(3):
        cols.append(("f", torch.clamp(300 * torch.randn(rows, device=device, generator=g), -4999, 4999).contiguous(), (-5000.0, 5000.0, 1.0)))
    for c in range(18):
        cols.append(("f", torch.clamp(10 + 0.2 * c + 0.8 * torch.randn(rows, device=device, generator=g), 8.001, 15.99).contiguous(), (8.0, 16.0, 0.01)))
    cols.append(("i", 5 * 10 ** 9 + torch.arange(rows, device=device, dtype=torch.int64), None))
    cols.append(("i", torch.poisson(torch.full((rows,), 40.0, device=device), generator=g).to(torch.int64), None))
    cols.append(("i", torch.randint(-1, 2 ** 20, (rows,), device=device, dtype=torch.int64, generator=g), None))
    cols.append(("i", torch.full((rows,), 7, device=device, dtype=torch.int64), None))
    nblk = (rows + B - 1) // B
    in_bytes = sum(t.numel() * t.element_size() for _, t, _ in cols)
    outs, decs, enc_batches, dec_protos = [], [], [], []
    for kind, t, lim in cols:
        slot = (t.element_size() * B + 15) // 16 * 16
        outs.append((torch.zeros(nblk * slot, dtype=torch.uint8, device=device), slot))
        decs.append(torch.empty_like(t))
    ej, dj = [], []
    for b in range(nblk):
        n = min(B, rows - b * B)
        for ci, (kind, t, lim) in enumerate(cols):
            o, slot = outs[ci]
            esz = t.element_size()
            if kind == "f":
                low, high, dx = lim
                pix = mnw.float_group_pixels(low, high, dx)
                ej.append(mnw.EncJob(t.data_ptr() + esz * b * B, o.data_ptr() + b * slot, n, slot, mnw.FLOAT_GROUP, low, high, pix, 1))
                dj.append(mnw.DecJob(o.data_ptr() + b * slot, decs[ci].data_ptr() + esz * b * B, n, 0, 0, 0, mnw.FLOAT_GROUP, low, high, pix, 1,
                                     seed=1, block=b * len(cols) + ci))
            else:
                ej.append(mnw.EncJob(t.data_ptr() + esz * b * B, o.data_ptr() + b * slot, n, slot, mnw.INT_GROUP))
                dj.append(mnw.DecJob(o.data_ptr() + b * slot, decs[ci].data_ptr() + esz * b * B, n, 0, 0, 0, mnw.INT_GROUP))
        if (b + 1) % 16 == 0 or b == nblk - 1:
            enc_batches.append(ctx.enc_batch(ej))
            dec_protos.append(ctx.dec_batch(dj))
            ej, dj = [], []
    torch.cuda.synchronize()

    def encode_all():
        return [ctx.encode_blocks(eb, raw=True) for eb in enc_batches]

    def decode_all(metas):
        for db, m in zip(dec_protos, metas):
            ctx.dec_batch_update(db, m)
            ctx.decode_blocks(db)

    for _ in range(3):
        metas = encode_all()
        decode_all(metas)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        metas = encode_all()
    torch.cuda.synchronize()
    t_enc = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        decode_all(metas)
    torch.cuda.synchronize()
    t_dec = (time.perf_counter() - t0) / reps
    packed = sum(m.nbytes for mb in metas for m in ctx.metas_of(mb))
    for ci, (kind, t, lim) in enumerate(cols):
        if kind == "i":
            assert torch.equal(decs[ci], t), ci
        else:
            d = (decs[ci].double() - t.double()).abs()
            if lim[0] == 0.0:
                d = torch.minimum(d, 250.0 - d)
            assert float(d.max()) <= lim[2] * 1.01 + 1e-3, (ci, float(d.max()))
    kt = {}
    if timing:
        ctx.timing_enable(True)
        ctx.timing_reset()
        decode_all(encode_all())
        torch.cuda.synchronize()
        kt = {k: v["total_ms"] for k, v in ctx.timing_get().items() if v["launches"]}
        ctx.timing_enable(False)
    alg = in_bytes + packed
    rec = {"rows": rows, "block_rows": B, "columns": len(cols), "int_columns": 4, "row_blocks_per_call": 16, "input_GB": in_bytes / 1e9,
           "packed_GB": packed / 1e9, "encode_ms": t_enc * 1e3, "decode_ms": t_dec * 1e3,
           "encode_GBps_input": in_bytes / t_enc / 1e9, "decode_GBps_output": in_bytes / t_dec / 1e9,
           "encode_frac_of_8TBps": alg / t_enc / 1e9 / HBM_PEAK_GBPS, "decode_frac_of_8TBps": alg / t_dec / 1e9 / HBM_PEAK_GBPS,
           "kernel_ms_one_pass": kt,
           "note": "algorithmic bytes = input + streams; wall-clock of all library calls of the pass, not kernel time"}
    del cols, outs, decs
    return rec


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # a bare stream of the codec's traffic mix on this GPU, as a child process BEFORE this process touches the GPU
    ceiling = stream_ceiling() if (world == 1 and rank == 0 and not a.no_stream_ceiling) else None
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=a.backend, rank=rank, world_size=world)
    if a.backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())      # rehearsal: more ranks than GPUs
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    cdev = device if a.backend == "nccl" else None           # where the few integers that cross ranks live (gloo: host tensors)

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
    B = a.block
    # ---- the global column and this rank's contiguous range of its blocks ---------------------
    n_global = a.particles if a.strong else a.particles * world
    nblk_global, b_lo, b_hi, N = shard.rank_plan(n_global, B, rank, world)     # N: this rank's elements
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
    r = run_config(mnw, shard, ctx, x, y, device, dist, N, B, pixels, a.steps, a.warmup, timing, block0=b_lo, cdev=cdev)

    # ---- what a writer does with the ranks' results: (min, bits, nbytes) of every block gathered in block order,
    # the block offsets are their prefix sum (go/block_index.go:16-35).  Metadata only; once, after the steps.
    t0 = time.perf_counter()
    rows = shard.gather_block_meta([(m.min, m.bits, m.nbytes) for m in r["metas"]], nblk_global, rank, world, device=cdev)
    all_sizes = [row[2] for row in rows]
    offsets = shard.block_offsets(all_sizes)
    gather_ms = (time.perf_counter() - t0) * 1e3
    assert len(offsets) == nblk_global and offsets[-1] + all_sizes[-1] == sum(all_sizes)

    # ---- extra legs (every rank runs them on its own device, rank 0 reports its own numbers) -------------------
    extra = {}
    if a.legs == "all":
        if a.alt_block and a.alt_block != B:
            rb = run_config(mnw, shard, ctx, x, y, device, dist, N, a.alt_block, pixels, min(a.steps, 5), 2, timing,
                            across_ranks=False)
            extra["blocks_2p%d" % (a.alt_block.bit_length() - 1)] = leg_record(
                rb, timing, "the same column in blocks of %d elements" % a.alt_block)
        if a.small_particles and a.small_particles < N:
            ns = a.small_particles
            rs = run_config(mnw, shard, ctx, x, y, device, dist, ns, B, pixels, min(a.steps, 20), 3, timing, across_ranks=False)
            extra["column_1e8"] = leg_record(rs, timing, "BASELINE configs[1]: the first %d particles of the column, blocks of %d "
                                             "(at this length the encode's 225 MB of streams are still in the 256 MB Infinity Cache "
                                             "when the decode reads them)" % (ns, B))
            if a.alt_block and a.alt_block != B:
                rs2 = run_config(mnw, shard, ctx, x, y, device, dist, ns, a.alt_block, pixels, min(a.steps, 10), 2, timing,
                                 across_ranks=False)
                extra["column_1e8_blocks_2p%d" % (a.alt_block.bit_length() - 1)] = leg_record(
                    rs2, timing, "configs[1] in blocks of %d elements (SURVEY.md 8d C2's default block)" % a.alt_block)
        if world == 1:
            # one GPU's share of --strong at 8 GPUs (1.25e8 particles of the column), timed here: the 8-GPU speed-up it projects,
            # while no 8-GPU node has measured the real one
            nsh = a.particles // 8
            if 0 < nsh < N:
                rsh = run_config(mnw, shard, ctx, x, y, device, dist, nsh, B, pixels, min(a.steps, 20), 3, timing, across_ranks=False)
                rec = leg_record(rsh, timing, "one GPU's share of `--strong --gpus 8`: %d particles of the column, blocks of %d" % (nsh, B))
                rec["projected_speedup_8_gpus"] = (r["dt"] / r["steps"]) / (rsh["dt"] / rsh["steps"])
                rec["projection"] = "ms_per_step of the whole column on this GPU / ms_per_step of an eighth of it on this GPU (no exchange between ranks: blocks are independent)"
                extra["strong_projection"] = rec
            # SURVEY 8d C2's third block size: the whole column as ONE block (too large to be held on chip: reduce + pack, two reads)
            rw = run_config(mnw, shard, ctx, x, y, device, dist, N, N, pixels, 3, 1, timing, across_ranks=False)
            extra["column_one_block"] = leg_record(rw, timing, "the whole column as ONE block of %d elements (SURVEY.md 8d C2 'whole column'): "
                                                   "k_reduce_f32 + k_pack_f32, the block is read twice" % N)
        if not a.no_minp:
            del y
            torch.cuda.empty_cache()
            if dist is not None:
                dist.barrier()                           # the ranks run their cells side by side
            extra["minp_cell"] = minp_leg(mnw, ctx, device, 5, timing)
            if dist is not None:
                # configs[3] as the reference runs it: one file cell per GPU (go/minp/minp.go:98-118), no exchange between them.
                # All ranks ran their cell at the same time: the snapshot's rate = all particles / the slowest rank's step.
                slowest = shard.max_over_ranks(extra["minp_cell"]["ms_per_step"] * 1e-3, device=cdev)
                extra["minp_cell"]["all_ranks"] = {"cells": world, "particles": world * extra["minp_cell"]["particles"],
                                                   "slowest_rank_ms_per_step": slowest * 1e3,
                                                   "value": 12.0 * world * extra["minp_cell"]["particles"] / slowest / 1e9,
                                                   "unit": "GB/s of input (12 B per particle), encode + decode, all ranks"}
            torch.cuda.empty_cache()
            extra["minp_cell_velocities"] = minp_leg(mnw, ctx, device, 5, timing, velocities=True)
            torch.cuda.empty_cache()
            extra["catalogue_c3"] = catalogue_leg(mnw, ctx, device, 10 ** 8, 3, timing)
    counters = ctx.counters()
    # the single-read kernel must never have given up (a co-tenant kernel would make it: 50 ms + a 2x slower path)
    # (a rehearsal in which ranks share one GPU is exactly that co-tenant: there the give-ups are expected and only reported)
    # Reported, not fatal: the figure printed below then includes those 50 ms waits and says so in `counters`.
    if counters["single_read_aborts"] != 0:
        print("bench.py: rank %d: the single-read launch gave up %d time(s) -- something else was running on this GPU; "
              "the batch was redone by the two-launch path (correct, slower)" % (rank, counters["single_read_aborts"]), file=sys.stderr)

    enc_r, dec_r, _ = rooflines(r, timing)
    mine = {"rank": rank, "particles": N, "blocks": r["nblk"], "ms_per_step_own_clock": r["t_enc"] * 1e3 + r["t_dec"] * 1e3,
            "encode_kernel_us": enc_r["avg_launch_us"] if enc_r else None, "decode_kernel_us": dec_r["avg_launch_us"] if dec_r else None,
            "encode_frac": enc_r["frac"] if enc_r else None, "decode_frac": dec_r["frac"] if dec_r else None,
            "pci_bus_id": getattr(torch.cuda.get_device_properties(local_rank), "pci_bus_id", None),
            "single_read_aborts": counters["single_read_aborts"], "backoff_batches": counters.get("backoff_batches", 0)}
    per_rank = [mine]
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    metas, dt = r["metas"], r["dt"]
    bits = sorted(set(m.bits for m in metas))
    ms_per_step = dt / a.steps * 1e3
    value = 4.0 * n_global / (dt / a.steps) / 1e9
    packed_bytes = float(sum(m.nbytes for m in metas))
    out = {
        "metric": "encode+decode GB/s (input)", "value": value, "unit": "GB/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong" if a.strong else "weak", "vs_baseline": None,
        "dtype": "f32->u32 fixed point (f32 sub/div, integer pack)", "data": "synthetic",
        "config": {"workload": "the metric's configuration: single float32 position column, %d particles%s, FloatGroup "
                               "low=0 high=250 dx=0.001 (pixels=%d), blocks of %d (the block size of the minp files of "
                               "configs[3]: ~1e6 particles per sub-cell), encode+decode round trip, device resident"
                               % (n_global, "" if world == 1 else " in all (%d per GPU)" % N, pixels, B),
                   "particles_per_gpu": N, "blocks_per_gpu": r["nblk"], "blocks": nblk_global, "bits": bits,
                   "parallelism": "contiguous block ranges per GPU (shard_range), no data-path collective"},
        "pct_hbm_peak": 100.0 * (4.0 * N * 2 + 2 * packed_bytes) / (dt / a.steps) / 1e9 / HBM_PEAK_GBPS,
        "encode_GBps_input": 4.0 * N / r["t_enc"] / 1e9, "decode_GBps_output": 4.0 * N / r["t_dec"] / 1e9,
        "compressed_bits_per_elt": 8.0 * packed_bytes / N,
        "block_index": {"blocks": nblk_global, "stream_bytes": int(sum(all_sizes)), "gather_and_prefix_ms": gather_ms},
        "counters": counters,
        "device": {"name": torch.cuda.get_device_name(local_rank),
                   "pci_bus_id": getattr(torch.cuda.get_device_properties(local_rank), "pci_bus_id", None)},
    }

    degraded = [q for q in per_rank if q and (q["single_read_aborts"] or q["backoff_batches"])]
    if degraded:
        out["degraded"] = "the single-read launch gave up on rank(s) %s (%d time(s) in all; %d batches skipped it afterwards): something " \
                          "else was running on the GPU, `value` includes the 50 ms waits and the two-launch reruns" % (
                              [q["rank"] for q in degraded], sum(q["single_read_aborts"] for q in degraded), sum(q["backoff_batches"] for q in degraded))
    if world > 1:
        out["per_rank"] = per_rank
    enc, dec, kernels = rooflines(r, timing)
    if enc:
        pmc = os.path.join(ROOT, "profiles", PROFILE_TAG, "pmc_traffic.json")
        if os.path.exists(pmc) and world == 1 and N == 10**9 and B == 2**20:
            # HBM bytes per launch from rocprofv3 --pmc passes of this very workload (collected separately; profiles/README.md)
            try:
                with open(pmc) as f:
                    t = json.load(f)
                enc["traffic"] = t.get("encode_bytes_per_launch")
                dec["traffic"] = t.get("decode_bytes_per_launch")
                enc["traffic_source"] = dec["traffic_source"] = t.get("source")
            except Exception:
                pass
        # `roofline` = the dominant kernel of the step (the longer of the two launches); both are reported
        enc_us = sum(enc["avg_launch_us"].values())
        out["roofline"] = dict(dec if dec["avg_launch_us"] > enc_us else enc)
        out["roofline_encode"], out["roofline_decode"], out["kernels"] = enc, dec, kernels
        if world > 1:
            # the slowest rank's dominant kernel (every rank's launch averages are in `per_rank`)
            cand = [q for q in per_rank if q and q["decode_kernel_us"]]
            if cand:
                slow = max(cand, key=lambda q: max(q["decode_kernel_us"], sum(q["encode_kernel_us"].values())))
                dec_dom = slow["decode_kernel_us"] > sum(slow["encode_kernel_us"].values())
                us = slow["decode_kernel_us"] if dec_dom else sum(slow["encode_kernel_us"].values())
                alg = (dec if dec_dom else enc)["algorithmic_bytes_per_elt"] * slow["particles"]
                out["roofline"] = {"bound": "hbm", "kernel": (dec if dec_dom else enc)["kernel"] + " on rank %d (the slowest)" % slow["rank"],
                                   "achieved": alg / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                   "frac": alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, "traffic": None, "avg_launch_us": us,
                                   "elements_per_launch": float(slow["particles"]), "pci_bus_id": slow["pci_bus_id"]}
        if ceiling:
            # the same traffic mix as a bare stream on this very GPU, measured seconds before the steps: what `frac` is to be read against
            out["roofline"]["stream_ceiling_GBps"] = ceiling.get("decode_mix_GBps" if out["roofline"]["kernel"].startswith("k_unpack") else "encode_mix_GBps")
            enc["stream_ceiling_GBps"], dec["stream_ceiling_GBps"] = ceiling.get("encode_mix_GBps"), ceiling.get("decode_mix_GBps")
            out["stream_ceiling"] = ceiling
    out.update(extra)

    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pixels)

    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
