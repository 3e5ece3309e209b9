"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads, exports every symbol
include/minnow_hip.h declares, its host scalars agree with the oracle, and compute entry points
fail loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mnw():
    import minnow_amd
    from minnow_amd import build as b

    b.build()
    return minnow_amd


def test_header_symbols_all_exported(mnw):
    hdr = open(os.path.join(ROOT, "include", "minnow_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mnw_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    L = C.CDLL(mnw.lib_path())
    for name in sorted(declared):
        assert hasattr(L, name), "libminnow_hip.so does not export " + name
    # and the Python binding declares exactly the header's functions
    from minnow_amd import _lib
    assert set(_lib.SIGNATURES) == declared


def test_abi_version_and_strerror(mnw):
    L = mnw.lib()
    assert L.mnw_abi_version() == 5
    assert b"ok" == L.mnw_strerror(0)
    assert b"fallback" in L.mnw_strerror(-3)


def test_struct_layouts_match_header(mnw):
    from minnow_amd import _lib
    assert C.sizeof(_lib.EncJobC) == 64
    assert C.sizeof(_lib.DecJobC) == 96
    assert C.sizeof(_lib.BlockMetaC) == 32
    assert C.sizeof(_lib.KernelTimeC) == 16
    # and against the header itself, as a C compiler lays the structs out
    import subprocess
    import tempfile
    prog = r"""
#include <stdio.h>
#include <stddef.h>
#include "minnow_hip.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(mnw_enc_job), offsetof(mnw_enc_job, xform), sizeof(mnw_dec_job),
           offsetof(mnw_dec_job, xform), sizeof(mnw_block_meta), sizeof(mnw_kernel_time), sizeof(mnw_minh_col),
           offsetof(mnw_minh_col, log), sizeof(mnw_cut_col), offsetof(mnw_cut_col, elem_bytes));
    return 0;
}
"""
    with tempfile.TemporaryDirectory() as td:
        src, exe = os.path.join(td, "l.c"), os.path.join(td, "l")
        open(src, "w").write(prog)
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        got = [int(v) for v in subprocess.check_output([exe]).split()]
    want = [C.sizeof(_lib.EncJobC), _lib.EncJobC.xform.offset, C.sizeof(_lib.DecJobC), _lib.DecJobC.xform.offset,
            C.sizeof(_lib.BlockMetaC), C.sizeof(_lib.KernelTimeC), C.sizeof(_lib.MinhColC), _lib.MinhColC.log.offset,
            C.sizeof(_lib.CutColC), _lib.CutColC.elem_bytes.offset]
    assert got == want


def test_host_scalars_match_oracle(mnw, oracle):
    for k in range(0, 64):
        for m in (2**k - 1, 2**k, 2**k + 1, 2**k + 2, 2**k + 3, 3 * 2**k // 2):
            if 0 <= m < 2**64 - 1:
                assert mnw.precision_needed(m) == oracle.precision_needed(m), m
    assert mnw.precision_needed(2**64 - 1) == -(2**63) == oracle.precision_needed(2**64 - 1)
    rng = np.random.default_rng(3)
    for m in rng.integers(2**48, 2**63, size=2000, dtype=np.uint64):
        assert mnw.precision_needed(int(m)) == oracle.precision_needed(int(m))
    for bits in (0, 1, 7, 18, 33, 64):
        for n in (0, 1, 9, 1000, 10**9):
            assert mnw.array_bytes(bits, n) == oracle.array_bytes(bits, n)
    for lo, hi, dx in ((0, 250, 0.001), (10, 14, 0.01), (-50, 100, 1), (-50, 100, 10), (0, 1000, 5),
                       (-5000, 5000, 1), (8, 16, 0.01), (0.5, 62.5, 0.0003)):
        assert mnw.float_group_pixels(lo, hi, dx) == oracle.float_group_pixels(lo, hi, dx)


def test_no_gpu_fails_loudly(mnw):
    if mnw.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(mnw.MnwError) as e:
        mnw.Context(0)
    assert e.value.status == -3   # MNW_E_NOGPU: there is no CPU fallback


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under minnow_amd/, include/ or tools/ may reference it (only tests/,
    smoke() and bench.py's cpu_baseline leg do)."""
    bad = []
    for base in ("minnow_amd", "include", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", ".sh")):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"mnw_oracle|mnwo_|from oracle|import oracle|oracle/", txt.replace("never includes anything from oracle/", "")):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad
