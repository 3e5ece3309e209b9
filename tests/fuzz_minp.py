#!/usr/bin/env python3
"""Randomised differential test of the minp vector path: mnw_vectors_encode / _decode (through the C ABI) against the oracle's
literal restatement of minp.Writer.Vectors / Reader.Vectors (go/minp/minp.go:85-118, 170-198) on random file-cell geometries,
value distributions, limits (the box, or taken from the data with the reference's Nextafter32 quirk) and library options
(single read of the grid at every depth and hand-out table, the two-launch form, the staging path; every decode ordering).
Usage: fuzz_minp.py [seconds] [seed].  Prints a summary line; raises at the first difference."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import minnow_amd as mnw
from minnow_amd import minp
from oracle import oracle as orc


def run(budget=60.0, seed=1):
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    orc.build(); orc.lib()
    t_end = time.time() + budget
    t_print = time.time() + 60
    cases = blocks = single = staged = 0
    with mnw.Context(0) as c:
        c.timing_enable(True)
        while time.time() < t_end:
            if time.time() > t_print:
                print("%d cells, %d blocks so far" % (cases, blocks), flush=True)
                t_print = time.time() + 60
            sub_cells = int(rng.integers(1, 7))
            if rng.random() < 0.8:
                n_sub = 4 * int(rng.integers(2, 17))                 # the kernels that read the grid themselves: n_sub = 8 .. 64, a multiple of four
            else:
                n_sub = int(rng.integers(1, 30))                     # anything: the staging path
            while (n_sub * sub_cells) ** 3 > 6_000_000:
                sub_cells -= 1
            nf = n_sub * sub_cells
            n = nf ** 3
            L = float(np.float32(rng.choice([250.0, 100.0, 1.0, 62.5])))
            periodic = bool(rng.random() < 0.6)
            mode = int(rng.integers(0, 7))
            if periodic:
                dx = float(np.float32(L / 10 ** rng.uniform(2.5, 5.5)))
                base = np.stack(np.meshgrid(*[np.arange(nf)] * 3, indexing="ij"), -1).reshape(-1, 3)[:, ::-1] * (L / nf)
                if mode == 0:
                    v = rng.uniform(0, L, (n, 3))
                elif mode == 1:                                        # a displaced lattice: every sub-cell a narrow arc, some across the seam
                    v = np.mod(base + L * 0.02 * rng.standard_normal((n, 3)), L)
                elif mode == 2:                                        # hugging the seam everywhere
                    v = np.mod(L * 0.003 * rng.standard_normal((n, 3)), L)
                elif mode == 3:                                        # one constant component, one two-cluster component
                    v = np.mod(base + 0.5, L)
                    v[:, 1] = L * 0.37
                    v[:, 2] = np.where(rng.random(n) < 0.5, 0.1 * L, 0.7 * L) + L * 0.01 * rng.standard_normal(n)
                    v = np.mod(v, L)
                elif mode == 4:                                        # on pixel edges and exact zeros
                    v = np.float32(dx) * rng.integers(0, int(L / dx), (n, 3)).astype(np.float64)
                    v[rng.integers(0, n, 5)] = 0.0
                else:
                    v = np.mod(base + L * 0.2 * rng.standard_normal((n, 3)), L)
                vec = v.astype(np.float32)
                vec[vec >= np.float32(L)] = 0.0
                if mode == 5 and rng.random() < 0.5:
                    vec[rng.integers(0, n), rng.integers(0, 3)] = np.nan          # the exact path: the call falls back to staging as a whole
            else:
                dx = float(np.float32(10 ** rng.uniform(-1, 1)))
                scale = 10 ** rng.uniform(0.5, 3.0)
                vec = (scale * rng.standard_normal((n, 3))).astype(np.float32)
                if mode == 1:
                    vec[:, 0] = np.float32(17.5)                                   # constant: data-derived limits of zero width
                if mode == 2:
                    vec[:, 2] = np.abs(vec[:, 2])                                   # a negative maximum elsewhere: Nextafter32 steps down
                    vec[:, 1] = -np.abs(vec[:, 1]) - 1
            vec = np.ascontiguousarray(vec)
            opts = {"vec3_fused": int(rng.random() < 0.9), "vec3_single_read": int(rng.random() < 0.8),
                    "vec3_deep": int(rng.integers(3, 6)), "vec3_round_table": int(rng.integers(0, 5)),
                    "vec3_x_order": int(rng.integers(0, 3)), "vec3_unpack_tiles": int(rng.choice([1, 2, 3, 8, 16, 100])),
                    "vec3_unpack_db": int(rng.integers(0, 2))}
            for k, val in opts.items():
                c.set_option(k, val)
            c.timing_reset()
            nb = 3 * sub_cells ** 3
            slot = (8 * n_sub ** 3 + 16 + 15) // 16 * 16
            t = torch.from_numpy(vec).to(dev)
            packed = torch.zeros(nb * slot, dtype=torch.uint8, device=dev)
            out = torch.zeros_like(t)
            torch.cuda.synchronize()
            metas, low, high, pixels = minp.vectors_encode(c, t.data_ptr(), nf, sub_cells, dx, periodic, L, packed.data_ptr(), slot)
            torch.cuda.synchronize()
            what = (seed, cases, nf, sub_cells, periodic, mode, dx, opts)
            oblocks, omn, omx, opix = orc.minp_vectors_encode(vec, nf, sub_cells, dx, periodic, L)
            assert np.array_equal(np.float32(low), omn) and np.array_equal(np.float32(high), omx) and list(pixels) == list(opix), what
            host = packed.cpu().numpy()
            for b, (m, (omin, obits, odata)) in enumerate(zip(metas, oblocks)):
                assert (m.min, m.bits, m.nbytes, m.status) == (omin, obits, odata.size, 0), (what, b, (m.min, m.bits, m.nbytes, m.status), (omin, obits, odata.size))
                assert host[b * slot: b * slot + m.nbytes].tobytes() == odata.tobytes(), (what, b)
            sd, fb = int(rng.integers(0, 2**31)), int(rng.integers(0, 1000))
            minp.vectors_decode(c, packed.data_ptr(), slot, metas, nf, sub_cells, low, high, pixels, periodic, L, out.data_ptr(), seed=sd, first_block=fb)
            torch.cuda.synchronize()
            ref = orc.minp_vectors_decode(oblocks, nf, sub_cells, omn, omx, opix, periodic, L, seed=sd, first_block=fb)
            assert np.array_equal(out.cpu().numpy(), ref, equal_nan=True), what
            kt = c.timing_get()
            single += kt.get("k_encode_reg_vec3", {"launches": 0})["launches"] > 0
            staged += kt.get("k_subcell_gather", {"launches": 0})["launches"] > 0
            assert c.counters()["single_read_aborts"] == 0, what
            cases += 1; blocks += nb
    return cases, blocks, single, staged


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.time()
    cases, blocks, single, staged = run(budget, seed)
    print("fuzz_minp: %d file cells, %d blocks, %d through the single-read kernel, %d through the staging path, no difference (%.0f s, seed %d)"
          % (cases, blocks, single, staged, time.time() - t0, seed))
