"""Host code under AddressSanitizer + UndefinedBehaviorSanitizer (the reference has no race detection or
sanitizer runs; the CPU restatement and the host builds of the product's number conversions and text reader get them here).
GPU code cannot be sanitized on this pool; these are host builds only (gcc -fsanitize=address,undefined), run in
subprocesses with the sanitizer runtime preloaded, and any report fails the test (-fno-sanitize-recover, abort_on_error)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]


def _runtime(name):
    p = subprocess.check_output(["gcc", "-print-file-name=" + name]).decode().strip()
    if not os.path.isabs(p):
        pytest.skip(name + " not installed")
    return os.path.realpath(p)


def _env(**extra):
    env = dict(os.environ)
    env["LD_PRELOAD"] = _runtime("libasan.so") + ":" + _runtime("libubsan.so")
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1:verify_asan_link_order=0"      # (CPython itself is not leak clean)
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    env.update(extra)
    return env


def _pytest_under_sanitizers(test_file, env):
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", test_file)],
                       capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert " passed" in out
    return out


def test_oracle_under_asan_ubsan():
    """oracle/mnw_oracle.c (the restatement every parity claim rests on) against all golden fixtures, instrumented"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "libmnw_oracle_san.so"])
    lib = os.path.join(ROOT, "oracle", "libmnw_oracle_san.so")
    _pytest_under_sanitizers("test_oracle_golden.py", _env(MNW_ORACLE_LIB=lib))


def test_text_number_conversions_under_asan_ubsan(tmp_path):
    """the host build of minnow_amd/csrc/mnw_text_num.h (the same source the device compiles), instrumented"""
    lib = str(tmp_path / "libtext_num_harness_san.so")
    subprocess.check_call(["g++", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC"] + SAN +
                          [os.path.join(ROOT, "tests", "cpp", "text_num_harness.cpp"), "-o", lib])
    _pytest_under_sanitizers("test_text_num_cpu.py", _env(MNW_TEXT_NUM_LIB=lib))


def test_text_reader_host_under_asan_ubsan(tmp_path):
    """tests/cpp/test_text_host.cpp (include/minnow_text.hpp: block cutting, headers, Rockstar names), instrumented"""
    from minnow_amd import build as b
    b.build()
    lib_dir = os.path.join(ROOT, "minnow_amd")
    exe = str(tmp_path / "test_text_host_san")
    subprocess.check_call(["g++", "-std=c++17", "-Wall"] + SAN + ["-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_text_host.cpp"), "-o", exe,
                           "-L", lib_dir, "-lminnow_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    env = _env()
    env.pop("LD_PRELOAD")                                  # the executable links the runtimes itself
    work = tmp_path / "w"
    work.mkdir()
    r = subprocess.run([exe, str(work)], capture_output=True, text=True, timeout=300, env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "ok: 4 text host tests" in out and "runtime error:" not in out and "AddressSanitizer" not in out


def test_minh_writer_teardown_under_asan_ubsan(tmp_path):
    """tests/cpp/test_minh_writer_teardown.cpp: a minh::Writer (and its file-writer thread) destroyed with a task running and one
    queued, and a Writer dropped without Close(); instrumented (use-after-free of f_ / cols_)"""
    from minnow_amd import build as b
    b.build()
    lib_dir = os.path.join(ROOT, "minnow_amd")
    exe = str(tmp_path / "test_minh_writer_teardown_san")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-pthread"] + SAN + ["-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_minh_writer_teardown.cpp"), "-o", exe,
                           "-L", lib_dir, "-lminnow_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    env = _env()
    env.pop("LD_PRELOAD")
    work = tmp_path / "w"
    work.mkdir()
    r = subprocess.run([exe, str(work)], capture_output=True, text=True, timeout=300, env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "ok: minh writer teardown" in out and "runtime error:" not in out and "AddressSanitizer" not in out
