"""Builds libminnow_hip.so (the C-ABI library) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so then travels to
the GPU box with the repo snapshot.
"""
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libminnow_hip.so")
SOURCES = ["mnw_kernels.hip", "mnw_api.cpp"]
HEADERS = ["mnw_common.h", "mnw_subcell.hip", "mnw_minh.hip", "mnw_text.hip", "mnw_text_num.h", "mnw_pow10_table.h", "mnw_cut.hip", "mnw_fast_tables.h", "mnw_gomath_fast.h", "mnw_diag.h", os.path.join("..", "..", "include", "minnow_hip.h")]

# -ffp-contract=off: the reference's float32 expressions are never fused (Go on amd64);
# correctly rounded f32 divide and denormal support are hipcc defaults, stated explicitly here.
FLAGS = [
    "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17",
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero",
    "-fno-fast-math", "-x", "hip", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
]


STAMP = LIB + ".srchash"


def source_hash():
    """Digest of everything the library is compiled from (sources, headers, flags): a library is reused only
    if it was built from exactly this text, whatever the files' time stamps say."""
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read() + b"\0")
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != source_hash()


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, f) for f in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(STAMP, "w") as fh:
        fh.write(source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
