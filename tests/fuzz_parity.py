#!/usr/bin/env python3
"""Randomised differential test: GPU (through the C ABI) against the oracle on random blocks, parameters and batch
shapes.  Usage: fuzz_parity.py [seconds] [seed].  Prints a summary line; raises at the first mismatch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import minnow_amd as mnw
from oracle import oracle as orc

def run(budget=60.0, seed=12345):
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    orc.build(); orc.lib()
    ctxs = {"default": mnw.Context(0), "two_pass": mnw.Context(0), "fused": mnw.Context(0)}
    ctxs["two_pass"].set_option("single_read", 0)
    ctxs["fused"].set_option("fused", 1)


    def rand_float_params():
        kind = rng.integers(0, 6)
        if kind == 0:
            return 0.0, 250.0, 0.001
        if kind == 1:
            return -5000.0, 5000.0, 1.0
        if kind == 2:
            return 8.0, 16.0, 0.01
        low = float(np.float32(rng.uniform(-1e4, 1e4)))
        span = float(np.float32(10 ** rng.uniform(-2, 4)))
        dx = float(np.float32(span / 10 ** rng.uniform(0.5, 6)))
        return low, float(np.float32(low + span)), dx


    def rand_float_block(n, low, high):
        span = high - low
        mode = rng.integers(0, 8)
        if mode == 0:
            x = rng.uniform(low, high, n)
        elif mode == 1:
            x = low + span * (0.5 + 0.05 * rng.standard_normal(n))
        elif mode == 2:      # hugging the seam: wrapped ring
            x = low + np.mod(span * 0.01 * rng.standard_normal(n), span)
        elif mode == 3:      # constant
            x = np.full(n, low + span * rng.uniform(0, 1))
        elif mode == 4:      # two clusters more than half a ring apart
            x = np.where(rng.random(n) < 0.5, low + 0.1 * span, low + 0.7 * span) + span * 0.01 * rng.standard_normal(n)
        elif mode == 5:      # values on pixel edges
            pix = max(1, int((high - low) / max(1e-30, (high - low) / 1000)))
            x = low + span * rng.integers(0, 1000, n) / 1000.0
        elif mode == 6:      # out of range on purpose (exact path): below low / at or above high
            x = rng.uniform(low - 0.2 * span, high + 0.2 * span, n)
        else:
            x = rng.uniform(low, high, n)
        x = x.astype(np.float32)
        if mode != 6:
            x = np.clip(x, np.float32(low), np.nextafter(np.float32(high), np.float32(low)))
        r = rng.random()
        if r < 0.03:
            x[rng.integers(0, n)] = np.nan
        elif r < 0.05:
            x[rng.integers(0, n)] = np.inf
        elif r < 0.08:
            x[rng.integers(0, n)] = np.float32(low)          # exact zero quotient
        return x


    def rand_int_block(n):
        mode = rng.integers(0, 5)
        if mode == 0:
            b = int(rng.integers(0, 63))
            return (rng.integers(-2**40, 2**40) + rng.integers(0, 2**b, n, dtype=np.int64, endpoint=True)).astype(np.int64)
        if mode == 1:
            return np.full(n, rng.integers(-10**9, 10**9), dtype=np.int64)
        if mode == 2:
            return rng.integers(-2**62, 2**62, n, dtype=np.int64)
        if mode == 3:
            return (5 * 10**9 + np.arange(n)).astype(np.int64)
        return rng.poisson(40, n).astype(np.int64)


    t_end = time.time() + budget
    t_print = time.time() + 60
    cases = blocks = 0
    while time.time() < t_end:
        if time.time() > t_print:                        # a sign of life every minute (a silent GPU job looks hung)
            print("%d batches, %d blocks so far" % (cases, blocks), flush=True)
            t_print = time.time() + 60
        mode = ["default", "two_pass", "fused"][rng.integers(0, 3)] if rng.random() < 0.5 else "default"
        ctx = ctxs[mode]
        nj = int(rng.integers(1, 12))
        jobs, keep, specs = [], [], []
        for _ in range(nj):
            n = int(rng.choice([1, 2, 31, 32, 33, 1000, 8191, 8192, 8193, 16384, 40000, 100001, 262144 + 5]))
            if rng.random() < 0.3:
                n = int(rng.integers(1, 300000))
            if rng.random() < 0.7:
                low, high, dx = rand_float_params()
                pixels = mnw.float_group_pixels(low, high, dx)
                if pixels <= 0:
                    continue
                x = rand_float_block(n, low, high)
                xf = 0
                r = rng.random()
                if r < 0.15:                          # minh column: clamp only, values partly outside [low, high)
                    xf = mnw.XF_CLAMP
                    x = (low + (high - low) * rng.uniform(-0.1, 1.1, n)).astype(np.float32)
                elif r < 0.3:                         # minh logarithmic column
                    xf = mnw.XF_CLAMP | mnw.XF_LOG10
                    low, high, dx = 8.0, 16.0, 0.01
                    pixels = mnw.float_group_pixels(low, high, dx)
                    x = (10 ** rng.uniform(7.8, 16.2, n)).astype(np.float32)
                    if rng.random() < 0.2:
                        x[rng.integers(0, n)] = [0.0, -1.0, np.inf, 1.0][rng.integers(0, 4)]
                t = torch.from_numpy(x).to(dev)
                o = torch.zeros(8 * n + 16, dtype=torch.uint8, device=dev)
                jobs.append(mnw.EncJob(t.data_ptr(), o.data_ptr(), n, o.numel(), mnw.FLOAT_GROUP, low, high, pixels, 1, xform=xf))
                specs.append(("f", x, (low, high, pixels, xf)))
            else:
                x = rand_int_block(n)
                t = torch.from_numpy(x).to(dev)
                o = torch.zeros(8 * n + 16, dtype=torch.uint8, device=dev)
                jobs.append(mnw.EncJob(t.data_ptr(), o.data_ptr(), n, o.numel(), mnw.INT_GROUP))
                specs.append(("i", x, None))
            keep.append((t, o))
        if not jobs:
            continue
        torch.cuda.synchronize()
        metas = ctx.encode_blocks(jobs, check=False)
        torch.cuda.synchronize()
        djobs, outs = [], []
        for (kind, x, par), m, (t, o) in zip(specs, metas, keep):
            n = x.size
            if kind == "f":
                low, high, pixels, xf = par
                try:
                    xp = orc.process_float_group(x, xf & mnw.XF_LOG10, low, high) if xf else x
                    omin, obits, odata = orc.float_encode(xp, low, high, pixels, True, literal=False)
                except Exception as e:      # the oracle reports what the reference would panic on
                    assert m.status != 0, ("oracle refused but GPU accepted", e)
                    continue
            else:
                omin, obits, odata = orc.int_encode(x, literal=False)
                if obits < 0:
                    assert m.status != 0
                    continue
            if (m.min, m.bits, m.nbytes, m.status) != (omin, obits, odata.size, 0) or \
                    o[: m.nbytes].cpu().numpy().tobytes() != odata.tobytes():
                print("MISMATCH encode", mode, kind, n, par, (m.min, m.bits, m.nbytes, m.status), (omin, obits, odata.size))
                np.save("/tmp/fuzz_fail_x.npy", x)
                raise AssertionError("GPU / oracle mismatch (encode)")
            # decode from an unaligned copy of the stream
            off = int(rng.integers(0, 8))
            s2 = torch.zeros(m.nbytes + 16, dtype=torch.uint8, device=dev)
            s2[off: off + m.nbytes] = o[: m.nbytes]
            if kind == "f":
                y = torch.empty(n, dtype=torch.float32, device=dev)
                djobs.append(mnw.DecJob(s2.data_ptr() + off, y.data_ptr(), n, m.nbytes, m.min, m.bits, mnw.FLOAT_GROUP, par[0], par[1], par[2], 1,
                                        seed=7, block=blocks, xform=par[3] & mnw.XF_LOG10))
            else:
                y = torch.empty(n, dtype=torch.int64, device=dev)
                djobs.append(mnw.DecJob(s2.data_ptr() + off, y.data_ptr(), n, m.nbytes, m.min, m.bits, mnw.INT_GROUP))
            outs.append((kind, x, par, m, odata, y, s2, blocks))
            blocks += 1
        torch.cuda.synchronize()
        if djobs:
            ctx.decode_blocks(djobs)
            torch.cuda.synchronize()
        for kind, x, par, m, odata, y, s2, blk in outs:
            n = x.size
            if kind == "f":
                ref = orc.float_decode(odata, m.min, m.bits, par[0], par[1], par[2], True, n, u=orc.dither(7, blk, n), literal=False)
                if par[3] & mnw.XF_LOG10:
                    ref = orc.unlog(ref)
                got = y.cpu().numpy()
                ok = np.array_equal(got.view(np.uint32), np.ascontiguousarray(ref, dtype=np.float32).view(np.uint32)) or np.array_equal(got, ref, equal_nan=True)
            else:
                ref = orc.int_decode(odata, m.min, m.bits, n, literal=False)
                ok = np.array_equal(y.cpu().numpy(), ref)
            if not ok:
                print("MISMATCH decode", mode, kind, n, par, (m.min, m.bits, m.nbytes))
                raise AssertionError("GPU / oracle mismatch (encode)")
        cases += 1
    return cases, blocks



if __name__ == "__main__":
    b = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
    c, k = run(b, sd)
    print("fuzz ok: %d batches, %d blocks, seed %d, %.0f s" % (c, k, sd, b))
