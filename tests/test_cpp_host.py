"""The C++ host side (include/minnow.hpp: bit.Array, intGroup/floatGroup, Writer/Reader) over the C ABI.
CPU part: the header compiles.  GPU part: tests/cpp/test_minnow_host.cpp, written after the reference's
Go tests, including a byte-for-byte comparison of whole .minw files with the ones the reference's Python
Writer produced (tests/golden/golden_small.json)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_minnow_host.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "test_minnow_host")


def _build():
    from minnow_amd import build as b
    b.build()
    lib_dir = os.path.join(ROOT, "minnow_amd")
    if (not os.path.exists(BIN)) or os.path.getmtime(BIN) < max(os.path.getmtime(SRC), os.path.getmtime(os.path.join(ROOT, "include", "minnow.hpp"))):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-o", BIN,
                               "-L", lib_dir, "-lminnow_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    return BIN


def test_host_header_compiles_and_links():
    """No GPU needed: the header is valid C++17 and every symbol it uses is exported by the library."""
    _build()
    assert os.path.exists(BIN)


@pytest.mark.gpu
def test_cpp_host_tests_on_gpu(tmp_path):
    exe = _build()
    small = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_small.json")))
    gold = tmp_path / "golden"
    gold.mkdir()
    for name, hx in small["minw_files"].items():
        (gold / name).write_bytes(bytes.fromhex(hx))
    scratch = tmp_path / "test_files"
    scratch.mkdir()
    r = subprocess.run([exe, str(scratch), str(gold)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok: 7 host tests" in r.stdout
