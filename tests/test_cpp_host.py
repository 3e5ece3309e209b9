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


MINH_SRC = os.path.join(ROOT, "tests", "cpp", "test_minh_host.cpp")
MINH_BIN = os.path.join(ROOT, "tests", "cpp", "test_minh_host")
TEXT_SRC = os.path.join(ROOT, "tests", "cpp", "test_text_host.cpp")
TEXT_BIN = os.path.join(ROOT, "tests", "cpp", "test_text_host")
MINP_SRC = os.path.join(ROOT, "tests", "cpp", "test_minp_host.cpp")
MINP_BIN = os.path.join(ROOT, "tests", "cpp", "test_minp_host")


def _build(src=SRC, exe=BIN):
    from minnow_amd import build as b
    b.build()
    lib_dir = os.path.join(ROOT, "minnow_amd")
    deps = [src] + [os.path.join(ROOT, "include", h) for h in ("minnow.hpp", "minh.hpp", "minp.hpp", "minnow_text.hpp", "minnow_hip.h")] + [b.LIB]
    if (not os.path.exists(exe)) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                               "-L", lib_dir, "-lminnow_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_host_header_compiles_and_links():
    """No GPU needed: the header is valid C++17 and every symbol it uses is exported by the library."""
    _build()
    assert os.path.exists(BIN)
    _build(MINH_SRC, MINH_BIN)
    assert os.path.exists(MINH_BIN)
    _build(MINP_SRC, MINP_BIN)
    assert os.path.exists(MINP_BIN)


def test_cpp_text_reader_host_only(tmp_path):
    """include/minnow_text.hpp without a GPU: TestLineHeader, TestCommentHeader, the block chain of TestNextBlock and
    TestRockstarHeader of go/text/text_test.go (file handling and header parsing stay on the host)"""
    exe = _build(TEXT_SRC, TEXT_BIN)
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok: 4 text host tests" in r.stdout


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


@pytest.mark.gpu
def test_cpp_minh_host_tests_on_gpu(tmp_path):
    """include/minh.hpp: the reference's TestReaderWriter (go/minh/minh_test.go:10-117), a whole minh file byte for byte
    against the reference Python minh.Writer's, device-array forms against host-array forms, the panics"""
    exe = _build(MINH_SRC, MINH_BIN)
    small = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_small.json")))
    gold = tmp_path / "golden"
    gold.mkdir()
    for name, hx in small["minh_files"].items():
        (gold / name).write_bytes(bytes.fromhex(hx))
    scratch = tmp_path / "test_files"
    scratch.mkdir()
    r = subprocess.run([exe, str(scratch), str(gold)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok: 6 minh host tests" in r.stdout


@pytest.mark.gpu
def test_cpp_minp_host_tests_on_gpu(tmp_path):
    """include/minp.hpp: the reference's TestVecReaderWriter and TestIDs (go/minp/minp_test.go:7-100); the file written from
    the whole grid on the device equals, byte for byte, the one written block by block through getSubCell; device grids"""
    exe = _build(MINP_SRC, MINP_BIN)
    scratch = tmp_path / "test_files"
    scratch.mkdir()
    r = subprocess.run([exe, str(scratch)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok: 3 minp host tests" in r.stdout
