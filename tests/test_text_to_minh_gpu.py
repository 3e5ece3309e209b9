"""End to end: text catalogue -> .minh file on the GPU (tools/text_to_minh.cpp = the conversion loop of the reference's
scripts/text_to_minh.go:166-214 over include/minnow_text.hpp, mnw_generic_cut and include/minh.hpp), read back through
minh::Reader and compared with the oracle's restatement of every stage: text.Reader.Block (parse.go), GenericCut, the minh
column transforms within the reference's own tolerances (minh_test.go:430-446), block boundaries as text.findBlocks cuts
them (text.go:82-131)."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "text_to_minh.cpp")
BIN = os.path.join(ROOT, "tools", "text_to_minh")


def _build():
    from minnow_amd import build as b
    b.build()
    lib_dir = os.path.join(ROOT, "minnow_amd")
    deps = [SRC, b.LIB] + [os.path.join(ROOT, "include", h) for h in ("minnow.hpp", "minh.hpp", "minnow_text.hpp", "minnow_hip.h")]
    if (not os.path.exists(BIN)) or os.path.getmtime(BIN) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-o", BIN,
                               "-L", lib_dir, "-lminnow_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    return BIN


def _find_blocks(data, max_block, max_item):
    """text.findBlocks / nextBlock (text.go:82-131)"""
    starts, curr = [0], 0
    while curr + max_block < len(data):
        s = max_block - max_item + curr
        curr = s + data[s:s + max_item].index(b"\n") + 1
        starts.append(curr)
    return starts + [len(data)]


def test_text_to_minh_round_trip(tmp_path, oracle):
    exe = _build()
    rng = np.random.default_rng(123)
    rows = 30000
    ids = rng.integers(0, 10**12, rows)
    mvir = 10 ** rng.uniform(9, 15.5, rows)
    pos = rng.uniform(0, 250, (rows, 3))
    vel = rng.normal(0, 300, (rows, 3))
    npart = rng.integers(1, 10**6, rows)
    spin = rng.uniform(0, 0.2, rows)
    lines = ["#ID Mvir X Y Z VX VY VZ Np Spin", "#a = 1.0000", "#Units: Msun/h, Mpc/h"]
    for i in range(rows):
        lines.append("%d %.4e %.5f %.5f %.5f %.2f %.2f %.2f %d %.5f" % (ids[i], mvir[i], pos[i, 0], pos[i, 1], pos[i, 2], vel[i, 0], vel[i, 1], vel[i, 2],
                                                                      npart[i], spin[i]))
    data = ("\n".join(lines) + "\n").encode()
    src = tmp_path / "hlist.txt"
    src.write_bytes(data)
    out = tmp_path / "hlist.minh"
    max_block = 600000
    cols = ["ID:int", "Mvir:log:1e9:1e16:0.01", "X:pos:0.001", "Y:pos:0.001", "Z:pos:0.001", "VX:lin:-5000:5000:1", "Np:int64", "Spin:float32"]
    cutoff = 1e11
    r = subprocess.run([exe, str(src), str(out), "250", "Mvir", str(cutoff), str(max_block)] + cols, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    info = json.loads(r.stdout)
    d = subprocess.run([exe, "--dump", str(out)], capture_output=True, text=True, timeout=300)
    assert d.returncode == 0, d.stdout + d.stderr
    got = json.loads(d.stdout)

    # the oracle, block by block as the reference's reader cuts them
    bounds = _find_blocks(data, max_block, min(100 * 1024, max_block // 2))      # MaxItemSize: text.go:38
    assert info["blocks"] == len(bounds) - 1 and info["blocks"] >= 3
    fields = {"ID": 0, "Mvir": 1, "X": 2, "Y": 3, "Z": 4, "VX": 5, "Np": 8, "Spin": 9}
    fnames, inames = ["Mvir", "X", "Y", "Z", "VX", "Spin"], ["ID", "Np"]
    want = {k: [] for k in fields}
    lens = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        f, i = oracle.text_block_parse(data[a:b], ord(" "), ord("#"), [fields[k] for k in fnames], [fields[k] for k in inames])
        n, kept = oracle.generic_cut(cutoff, f[0], f + i)
        if n:
            lens.append(n)
            for k, x in zip(fnames + inames, kept):
                want[k].append(x)
    want = {k: np.concatenate(v) for k, v in want.items()}
    assert got["block_lengths"] == lens and got["length"] == sum(lens) == info["rows_kept"] and info["rows"] == rows
    assert got["L"] == 250.0 and got["text_bytes"] > 0
    c = got["columns"]
    assert np.array_equal(np.array(c["ID"], dtype=np.int64), want["ID"])                     # minh.Int: exact
    assert np.array_equal(np.array(c["Np"], dtype=np.int64), want["Np"])                     # minh.Int64: raw copy
    assert np.array_equal(np.array(c["Spin"], dtype=np.float32), want["Spin"])               # minh.Float32: raw copy
    for k, dx in (("X", 0.001), ("Y", 0.001), ("Z", 0.001), ("VX", 1.0)):                    # float32sEq, minh_test.go:430-436
        g, w = np.array(c[k], dtype=np.float32), want[k]
        assert np.all(w + np.float32(dx) >= g) and np.all(w - np.float32(dx) <= g), k
    g, w = np.log10(np.array(c["Mvir"], dtype=np.float64)).astype(np.float32), np.log10(want["Mvir"].astype(np.float64)).astype(np.float32)
    assert np.all(w + np.float32(0.01) >= g) and np.all(w - np.float32(0.01) <= g)           # log32sEq, minh_test.go:438-446
    assert np.all(want["Mvir"] > np.float32(cutoff))
