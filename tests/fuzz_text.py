#!/usr/bin/env python3
"""Randomised differential test of the text ingest (mnw_text_parse) against the oracle's restatement of text.Reader.Block:
random catalogues -- line lengths from 0 to beyond the staged 16 KB, separator runs, comments anywhere, blank and
separator-only lines, every number format, literals only the host path converts -- and random malformed ones, where the
library must fail exactly when the oracle (i.e. the reference's panics) does.

    python tests/fuzz_text.py [--seconds S] [--seed N]        (on a GPU box; tests/test_text_gpu.py runs a short seeded leg)
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

INT_FORMS = [lambda r: str(int(r.integers(-10**18, 10**18))), lambda r: str(int(r.integers(0, 100))), lambda r: "%+d" % int(r.integers(-999, 999)),
             lambda r: "%012d" % int(r.integers(0, 10**9)), lambda r: str(int(r.integers(-2**63, 2**63 - 1, endpoint=True)))]
FLOAT_FORMS = [lambda r: "%.4e" % (10 ** r.uniform(8, 15)), lambda r: "%.5f" % r.uniform(0, 250), lambda r: "%.2f" % r.normal(0, 300),
               lambda r: "%g" % r.normal(0, 1e-3), lambda r: "%.17g" % (r.normal() * 10.0 ** int(r.integers(-40, 40))),
               lambda r: "%d" % int(r.integers(-10**6, 10**6)), lambda r: "%.12f" % r.uniform(-1, 1), lambda r: "%d.%020d" % (int(r.integers(0, 99)), int(r.integers(0, 10**18))),
               lambda r: str(r.choice(["inf", "-Inf", "+infinity", "nan", "NaN", "1e22", "1e23", "1E5", ".5", "5.", "-0", "0", "1e-45", "4.9e-324", "16777217",
                                       "1_000", "0x1p-2", "1_2.5e0_1"])),
               # numpy.savetxt's default format (19 significant digits, trailing zeros) and repr(): the device's 128-bit table path
               lambda r: "%.18e" % (r.normal() * 10.0 ** int(r.integers(-30, 30))), lambda r: "%.18e" % round(r.normal(0, 250), 2),
               lambda r: repr(float(r.normal() * 10.0 ** int(r.integers(-300, 300)))),
               lambda r: "%de%d" % (int(r.integers(10**16, 10**19, dtype=np.uint64)), int(r.integers(-320, 289)))]
BAD_TOKENS = ["x", "1x", "--1", "1e", "1e+", "1..2", "1.2.3", "e5", "1__0", "0x10", "", "+", "1e999", "99999999999999999999", "1.5", "abc", "1e5", "+nan", "_1", "0x1.8"]


def make_case(rng):
    ncol = int(rng.integers(1, 60))
    kinds = rng.integers(0, 2, ncol)                                   # 0 int, 1 float
    sep = " " if rng.random() < 0.8 else "\t"
    comm = "#"
    nlines = int(rng.choice([1, 3, 40, 400, 3000]))
    long_lines = rng.random() < 0.15
    out = []
    for _ in range(int(rng.integers(0, 3))):
        out.append(comm + " header " + "x" * int(rng.integers(0, 200)))
    for i in range(nlines):
        r = rng.random()
        if r < 0.03:
            out.append("")
            continue
        if r < 0.06:
            out.append(sep * int(rng.integers(1, 50)))
            continue
        if r < 0.08:
            out.append(sep * int(rng.integers(0, 3)) + comm + " only a comment")
            continue
        toks = [(INT_FORMS if kinds[k] == 0 else FLOAT_FORMS)[int(rng.integers(0, len(INT_FORMS) if kinds[k] == 0 else len(FLOAT_FORMS)))](rng) for k in range(ncol)]
        runs = 1 if sep == "\t" and rng.random() < 0.5 else int(rng.integers(1, 4))
        ln = sep * int(rng.integers(0, 3)) + (sep * runs).join(toks) + sep * int(rng.integers(0, 3))
        if long_lines and rng.random() < 0.3:
            ln += sep * int(rng.integers(100, 30000))
        if rng.random() < 0.03:
            ln += (sep if rng.random() < 0.5 else "") + comm + " trailing"
        out.append(ln)
    # malformed on purpose, sometimes
    bad = rng.random() < 0.25 and nlines > 0
    if bad:
        data_idx = [j for j, ln in enumerate(out) if ln.strip(sep) and not ln.lstrip(sep).startswith(comm)]
        if data_idx:
            j = int(rng.choice(data_idx))
            body = out[j].split(comm)[0]
            words = body.split(sep)
            nz = [q for q, w in enumerate(words) if w]
            how = rng.integers(0, 3)
            if how == 0 and len(nz) > 1:
                del words[nz[int(rng.integers(0, len(nz)))]]                  # one field fewer
            elif how == 1:
                words.append("7")                                            # one field more
                if not words[-2]:
                    pass
                else:
                    words[-1] = ""
                    words.append("7")
            else:
                words[nz[int(rng.integers(0, len(nz)))]] = str(rng.choice(BAD_TOKENS))     # maybe a token the column's type rejects
            out[j] = sep.join(words)
    text = "\n".join(out) + ("\n" if rng.random() < 0.5 else "")
    nf = int(rng.integers(0, min(ncol, 48) + 1))
    cols = rng.permutation(ncol)
    fcols = [int(c) for c in cols if kinds[c] == 1][:nf]
    icols = [int(c) for c in cols if kinds[c] == 0][:int(rng.integers(0, 49))]
    if rng.random() < 0.02:
        fcols.append(ncol + int(rng.integers(0, 3)))                         # a column the data does not have
    return text.encode(), ord(sep), ord(comm), fcols, icols


def run(seconds=None, cases=None, seed=0, verbose=False):
    import torch
    import minnow_amd as mnw
    from oracle import oracle
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    t0 = time.time()
    n = errors = general = 0
    with mnw.Context(0) as ctx:
        while (seconds is not None and time.time() - t0 < seconds) or (cases is not None and n < cases):
            data, sep, comm, fcols, icols = make_case(rng)
            try:
                rf, ri = oracle.text_block_parse(data, sep, comm, fcols, icols)
                want_err = None
            except (ValueError, IndexError) as e:
                want_err = e
            cap = data.count(b"\n") + 2
            fo = [torch.full((cap,), float("nan"), dtype=torch.float32, device=dev) for _ in fcols]
            io = [torch.full((cap,), -7, dtype=torch.int64, device=dev) for _ in icols]
            torch.cuda.synchronize()
            try:
                rows = ctx.text_parse(data, sep, comm, fcols, [t.data_ptr() for t in fo], icols, [t.data_ptr() for t in io], cap)
                got_err = None
            except mnw.MnwError as e:
                got_err = e
            torch.cuda.synchronize()
            tag = "case %d (seed %d): %d bytes, %d float + %d int columns" % (n, seed, len(data), len(fcols), len(icols))
            if (want_err is None) != (got_err is None):
                open("/tmp/fuzz_text_fail.txt", "wb").write(data)
                raise AssertionError("%s: oracle %r, library %r (text in /tmp/fuzz_text_fail.txt, fcols %s icols %s)" % (tag, want_err, got_err, fcols, icols))
            if want_err is None:
                nrows = rf[0].size if rf else (ri[0].size if ri else rows)
                assert rows == nrows or not (rf or ri), tag
                for a, b in zip(fo, rf):
                    g = a[:rows].cpu().numpy()
                    assert np.array_equal(g.view(np.uint32), b.view(np.uint32)) or np.array_equal(g, b, equal_nan=True), tag
                for a, b in zip(io, ri):
                    assert np.array_equal(a[:rows].cpu().numpy(), b), tag
                general += ctx.text_stats()[2] > 0
            else:
                errors += 1
            n += 1
            if verbose and n % 200 == 0:
                print("%d cases, %d with an error on both sides, %d with general tiles, %.0f s" % (n, errors, general, time.time() - t0), flush=True)
    return n, errors, general


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    n, e, g = run(seconds=a.seconds, seed=a.seed, verbose=True)
    print("ok: %d cases (%d rejected by both, %d with tiles on the general path)" % (n, e, g))
