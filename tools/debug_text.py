#!/usr/bin/env python3
"""Step-by-step run of the text ingest on growing inputs (debugging aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import minnow_amd as mnw
from oracle import oracle

dev = torch.device("cuda:0")
ctx = mnw.Context(0)


def run(name, data, fcols, icols):
    cap = data.count(b"\n") + 2
    fo = [torch.full((cap,), float("nan"), dtype=torch.float32, device=dev) for _ in fcols]
    io = [torch.full((cap,), -7, dtype=torch.int64, device=dev) for _ in icols]
    torch.cuda.synchronize()
    print("run", name, len(data), flush=True)
    n = ctx.text_parse(data, 32, 35, fcols, [t.data_ptr() for t in fo], icols, [t.data_ptr() for t in io], cap)
    torch.cuda.synchronize()
    print("   rows", n, "stats", ctx.text_stats(), flush=True)
    rf, ri = oracle.text_block_parse(data, 32, 35, fcols, icols)
    ok = all(np.array_equal(a[:n].cpu().numpy().view(np.uint32), b.view(np.uint32)) for a, b in zip(fo, rf)) and \
        all(np.array_equal(a[:n].cpu().numpy(), b) for a, b in zip(io, ri))
    print("   match", ok, flush=True)
    return ok


run("tiny-fast", b"1 2.5\n3 4.5\n", [1], [0])
run("tiny-comment", b"# h\n1 2.5\n3 4.5\n", [1], [0])
for rows in (100, 1000, 3000, 20000):
    data = "\n".join("%d %.3f %d" % (i, i * 0.5, i % 7) for i in range(rows)).encode() + b"\n"
    run("rows%d" % rows, data, [1], [0, 2])
rng = np.random.default_rng(1)
data = "\n".join(" ".join("%.4e" % v for v in rng.uniform(1, 1e6, 44)) for i in range(3000)).encode()
run("wide", data, list(range(4, 44)), [])

def gen(rng, rows, ncol, sep, trial):
    lines = ["#id mass x y z flag" + " pad" * 3, "# second header line"]
    for r in range(rows):
        w = []
        for k in range(ncol):
            kind = k % 4
            if kind == 0:
                w.append(str(int(rng.integers(-10**12, 10**12))))
            elif kind == 1:
                w.append("%.6e" % (10 ** rng.uniform(8, 15)))
            elif kind == 2:
                w.append("%.5f" % rng.uniform(0, 250))
            else:
                w.append("%g" % rng.normal(0, 300))
        pad = sep * int(rng.integers(1, 4)) if sep == " " else sep
        ln = (sep * int(rng.integers(0, 3)) if sep == " " else "") + pad.join(w)
        if rng.random() < 0.02:
            ln += " # trailing comment"
        lines.append(ln)
        if rng.random() < 0.01:
            lines.append(sep * int(rng.integers(0, 5)))
    return ("\n".join(lines) + ("\n" if trial % 2 == 0 else "")).encode()


rng = np.random.default_rng(71)
for trial, (rows, ncol) in enumerate([(1, 3), (257, 5), (5000, 44), (20000, 12)]):
    data = gen(rng, rows, ncol, " ", trial)
    fcols = [k for k in range(ncol) if k % 4 != 0][:40]
    icols = [k for k in range(ncol) if k % 4 == 0][:40]
    open(os.path.join(ROOT, "gpurun_out", "debug_text_trial%d.txt" % trial), "wb").write(data[:200000])
    run("trial%d" % trial, data, fcols, icols)
print("done", flush=True)
