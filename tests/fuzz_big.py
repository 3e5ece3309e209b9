#!/usr/bin/env python3
"""Randomised differential test of the LARGE-block encode paths: batches of one to three float32 / int64 blocks of 2^20 .. 2^24
elements (ragged lengths) next to small ones, random library options (flat-sweep limits of 128 / 256 / 512 items, big_blocks,
single_read / single_read_int), through mnw_encode_blocks and against the oracle: (min, bits, bytes) of every block.  Exercises the
sweeper workgroups of k_encode_reg_f32<., BIG> and k_encode_reg_i64, the two-level sweep and the reduce + pack fallbacks.
Usage: fuzz_big.py [seconds] [seed]."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import minnow_amd as mnw
from oracle import oracle as orc


def run(budget=60.0, seed=1):
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    orc.build(); orc.lib()
    t_end = time.time() + budget
    t_print = time.time() + 60
    cases = blocks = swept = 0
    with mnw.Context(0) as c:
        c.timing_enable(True)
        while time.time() < t_end:
            if time.time() > t_print:
                print("%d batches, %d blocks so far" % (cases, blocks), flush=True)
                t_print = time.time() + 60
            opts = {"float_flat_items": int(rng.choice([128, 256, 512])), "int_flat_items": int(rng.choice([128, 256, 512])),
                    "big_blocks": int(rng.random() < 0.85), "single_read": int(rng.random() < 0.9), "single_read_int": int(rng.random() < 0.9)}
            for k, v in opts.items():
                c.set_option(k, v)
            specs = []
            for _ in range(int(rng.integers(1, 4))):                        # the large ones
                is_f = rng.random() < 0.5
                lg = rng.uniform(20, 24 if is_f else 23.2)
                n = int(2 ** lg) + int(rng.integers(-5000, 5000))
                specs.append((is_f, n))
            for _ in range(int(rng.integers(0, 4))):                        # small company
                specs.append((rng.random() < 0.5, int(rng.choice([1, 77, 8192, 8193, 100001, 2**20]))))
            order = rng.permutation(len(specs))
            jobs, keep, want = [], [], []
            for i in order:
                is_f, n = specs[i]
                if is_f:
                    mode = int(rng.integers(0, 4))
                    if mode == 0:
                        x = np.mod(125 + 20 * rng.standard_normal(n), 250.0)
                    elif mode == 1:
                        x = np.mod(0.3 * rng.standard_normal(n), 250.0)                       # wrapped ring
                    elif mode == 2:
                        x = np.full(n, 25.0)                                                   # constant
                    else:
                        x = rng.uniform(0, 250.0, n)
                    x = x.astype(np.float32); x[x >= np.float32(250.0)] = 0.0
                    if rng.random() < 0.1:
                        x[rng.integers(0, n)] = np.nan                                         # exact path behind a large block
                    t = torch.from_numpy(x).to(dev)
                    o = torch.zeros(8 * n + 16, dtype=torch.uint8, device=dev)      # (a NaN makes a 64-bit block, as in the reference)
                    jobs.append(mnw.EncJob(t.data_ptr(), o.data_ptr(), n, o.numel(), mnw.FLOAT_GROUP, 0.0, 250.0, 250000, 1))
                    want.append(orc.float_encode(x, 0.0, 250.0, 250000, True, literal=False))
                else:
                    mode = int(rng.integers(0, 4))
                    if mode == 0:
                        x = rng.integers(10**6, 10**6 + 2**int(rng.integers(1, 31)), n, dtype=np.int64)
                    elif mode == 1:
                        x = rng.integers(-2**40, 2**41, n, dtype=np.int64)                     # more than 32 bits
                    elif mode == 2:
                        x = np.full(n, -123456789, dtype=np.int64)
                    else:
                        x = (5 * 10**9 + np.arange(n)).astype(np.int64)
                    t = torch.from_numpy(x).to(dev)
                    o = torch.zeros(8 * n + 16, dtype=torch.uint8, device=dev)
                    jobs.append(mnw.EncJob(t.data_ptr(), o.data_ptr(), n, o.numel(), mnw.INT_GROUP))
                    want.append(orc.int_encode(x, literal=False))
                keep.append((t, o))
            torch.cuda.synchronize()
            c.timing_reset(); c.counters_reset()
            metas = c.encode_blocks(jobs)
            torch.cuda.synchronize()
            what = (seed, cases, opts, [(specs[i][0], specs[i][1]) for i in order])
            for m, (t, o), (omin, obits, odata) in zip(metas, keep, want):
                assert (m.min, m.bits, m.nbytes, m.status) == (omin, obits, odata.size, 0), (what, (m.min, m.bits, m.nbytes, m.status), (omin, obits, odata.size))
                assert o[: m.nbytes].cpu().numpy().tobytes() == odata.tobytes(), what
                assert int(o[m.nbytes:].max()) == 0, what
            assert c.counters()["single_read_aborts"] == 0, what
            kt = c.timing_get()
            swept += any(kt.get(k, {"launches": 0})["launches"] for k in ("k_encode_reg_f32", "k_encode_reg_i64"))
            cases += 1; blocks += len(jobs)
            del keep, jobs
    return cases, blocks, swept


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.time()
    cases, blocks, swept = run(budget, seed)
    print("fuzz_big: %d batches, %d blocks, %d batches through a single-read kernel, no difference (%.0f s, seed %d)" % (cases, blocks, swept, time.time() - t0, seed))
