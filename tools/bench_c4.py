#!/usr/bin/env python3
"""BASELINE configs[3] shape on one GPU at the FILE level: tools/minp_cell (C++ over include/minp.hpp) writes one minp file cell
(500^3 particles, 125 sub-cells: 375 blocks per vector) from a host grid and from a device grid, positions and velocities, reads
the files back and checks the reference's round-trip criterion.  Prints the tool's JSON (stage times and GB/s of 12 B/particle)."""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from minnow_amd import build as b  # noqa: E402

b.build()
src, exe = os.path.join(ROOT, "tools", "minp_cell.cpp"), os.path.join(ROOT, "tools", "minp_cell")
lib_dir = os.path.join(ROOT, "minnow_amd")
deps = [src, b.LIB] + [os.path.join(ROOT, "include", h) for h in ("minnow.hpp", "minp.hpp", "minnow_hip.h")]
if (not os.path.exists(exe)) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                           "-L", lib_dir, "-lminnow_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
n_file = sys.argv[1] if len(sys.argv) > 1 else "500"
sub_cells = sys.argv[2] if len(sys.argv) > 2 else "5"
with tempfile.TemporaryDirectory(prefix="mnw_c4_") as tmp:
    r = subprocess.run([exe, tmp, n_file, sub_cells], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout + r.stderr
    print(json.dumps(json.loads(r.stdout.strip().splitlines()[-1])))
