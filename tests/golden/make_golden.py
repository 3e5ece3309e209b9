#!/usr/bin/env python3
"""Generate tests/golden/*.npz + golden_small.json from the REFERENCE's own Python/Cython codec.

Runs only in the build container (needs /root/reference).  The reference sources are copied
to a scratch dir OUTSIDE the repo (/tmp/mnw_ref_oracle), python/cy_bit.pyx is compiled there
with Cython (language_level=2: the .pyx relies on C-style integer '/'), and the reference
modules bit.py / minnow.py are imported from that scratch dir.  Only their OUTPUTS (inputs +
expected bytes/values) are written into the repo; no reference source is copied here.

What the Python mirror is authoritative for, and where it differs from the Go text, is listed
in SURVEY.md section 8c.  Float-group vectors are generated only on parameter sets where the
Python float64 derivation and the Go float32 derivation of (pixels, dx) provably agree
(asserted below), with in-range inputs, and pre-masked inputs for bit.array (cy_bit.pyx never
applies its mask).

Usage:  python tests/golden/make_golden.py
"""
import io
import json
import os
import shutil
import subprocess
import sys

import numpy as np

REF = "/root/reference/python"
SCRATCH = "/tmp/mnw_ref_oracle"
HERE = os.path.dirname(os.path.abspath(__file__))


def build_reference():
    if not os.path.isdir(REF):
        sys.exit("reference not present: goldens can only be generated in the build container")
    os.makedirs(SCRATCH, exist_ok=True)
    for f in os.listdir(REF):
        if f.endswith(".py") or f.endswith(".pyx"):
            shutil.copy(os.path.join(REF, f), SCRATCH)
    with open(os.path.join(SCRATCH, "setup_l2.py"), "w") as f:
        f.write(
            "from setuptools import setup, Extension\n"
            "from Cython.Build import cythonize\n"
            "import numpy as np\n"
            "setup(ext_modules=cythonize([Extension('cy_bit', ['cy_bit.pyx'],"
            " include_dirs=[np.get_include()])], language_level=2))\n"
        )
    subprocess.check_call([sys.executable, "setup_l2.py", "-q", "build_ext", "--inplace"],
                          cwd=SCRATCH, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.path.insert(0, SCRATCH)


def go_pixels(low, high, dx):
    """writer.go:73 in float32 arithmetic."""
    low, high, dx = np.float32(low), np.float32(high), np.float32(dx)
    return int(np.ceil(np.float64(np.float32(np.float32(high - low) / dx))))


def main():
    build_reference()
    import bit  # reference python/bit.py
    import minnow  # reference python/minnow.py

    rng = np.random.default_rng(0xACEDAD)
    small = {}
    arrays = {}

    # ---- 1. bit.array / bit.from_array KATs: every width 1..64, ragged lengths ---------
    lengths = [0, 1, 7, 8, 9, 63, 64, 65, 123, 1000]
    X = rng.integers(0, 2**63, size=1000, dtype=np.uint64)  # like rand.Int63() in bit_test.go
    arrays["pack_X"] = X
    arrays["pack_lengths"] = np.array(lengths, dtype=np.int64)
    offs, blob = [], []
    pos = 0
    for b in range(1, 65):
        mask = np.uint64((1 << b) - 1)
        for L in lengths:
            x = X[:L] & mask
            data = bit.array(b, x) if L > 0 else np.zeros(0, np.uint8)
            assert data.size == bit.array_bytes(b, L)
            if L > 0:
                back = bit.from_array(np.array(data), b, L)
                assert np.array_equal(back, x)
            offs.append((b, L, pos, data.size))
            blob.append(np.asarray(data, dtype=np.uint8))
            pos += data.size
    arrays["pack_index"] = np.array(offs, dtype=np.int64)
    arrays["pack_blob"] = np.concatenate(blob)
    small["pack_3bit_12345"] = bit.array(3, [1, 2, 3, 4, 5]).tobytes().hex()  # d158

    # ---- 2. precision_needed / array_bytes (numpy log2; the Go text uses math.Log2) ----
    pn = []
    for k in range(0, 49):
        for m in (2**k - 1, 2**k, 2**k + 1):
            if 0 <= m < 2**48:
                pn.append((m, int(bit.precision_needed(m))))
    small["precision_needed_lt_2p48"] = pn
    small["array_bytes"] = [(b, L, int(bit.array_bytes(b, L)))
                            for b in (0, 1, 3, 7, 8, 13, 18, 32, 45, 63, 64)
                            for L in (0, 1, 5, 8, 9, 123, 1000, 10**6 + 1)]

    # ---- 3. periodic_min: the reference table + random in-range trials ------------------
    small["periodic_min_table"] = {
        "pixels": 20,
        "data": [[0, 1, 2, 3], [10, 11, 12, 13], [18, 19, 0, 1], [1, 0, 19, 18], [1, 19, 18, 0]],
        "mins": [0, 10, 18, 18, 18],  # python/minnow_test.py:216-229
    }
    for d, m in zip(small["periodic_min_table"]["data"], small["periodic_min_table"]["mins"]):
        assert bit.periodic_min(d, 20) == m
    pm_cases = []
    for _ in range(4000):
        pixels = int(rng.integers(1, 60))
        n = int(rng.integers(1, 12))
        x = rng.integers(0, pixels, size=n).astype(np.int64)
        pm_cases.append((pixels, x.tolist(), int(bit.periodic_min(x, pixels))))
    for _ in range(500):  # clustered points on big rings (the realistic case)
        pixels = int(rng.integers(1000, 300000))
        n = int(rng.integers(1, 200))
        c = int(rng.integers(0, pixels))
        w = int(rng.integers(1, pixels))
        x = ((c + rng.integers(0, w, size=n)) % pixels).astype(np.int64)
        pm_cases.append((pixels, x.tolist(), int(bit.periodic_min(x, pixels))))
    small["periodic_min_random"] = pm_cases

    # ---- 4. int groups: (min, bits, stream bytes) from the reference _IntGroup ----------
    def ref_int_group(x):
        x = np.asarray(x, dtype=np.int64)
        g = minnow._IntGroup(0, len(x))
        f = io.BytesIO()
        g.write_data(f, x)
        return int(g.mins[0]), int(g.bits[0]), f.getvalue()

    int_vecs = [
        [100, 101, 102, 104], [1024, 1024, 1024], [0, 1023, 500], [-1000000, -500000],  # minnow_test.go:244-246
        [0], [7], [-1], [-5, 5], list(range(100)), [0, 2**31], [-(2**40), 2**40],
        [3, 3, 3, 4], [2**47 - 1, 0],
    ]
    for b in (1, 5, 8, 11, 16, 21, 27, 32, 33, 45, 47):
        n = int(rng.integers(1, 300))
        off = int(rng.integers(-10**6, 10**6))
        int_vecs.append((off + rng.integers(0, 2**b, size=n, dtype=np.int64)).tolist())
    small["int_group"] = []
    for v in int_vecs:
        mn, bits, data = ref_int_group(v)
        small["int_group"].append({"x": [int(t) for t in v], "min": mn, "bits": bits,
                                   "bytes": data.hex()})

    # ---- 5. float groups on parameter sets where Python float64 == Go float32 ----------
    small["float_group"] = []
    params = [(0, 250, 1), (0, 250, 0.001), (0, 1000, 5), (-50, 100, 1), (-50, 100, 10),
              (100, 200, 1)]
    go_test = {  # minnow_test.go:272-283
        (-50, 100, 1): [[-50, 0, 50, 49], [25, 25, 25, 25]],
        (-50, 100, 10): [[-50, 0, 50, 49, 0], [1, 2, 3, 4, 5], [0, 20, 0, 20, 0]],
    }
    for (low, high, dx) in params:
        pixels = go_pixels(low, high, dx)
        py_pixels = int(np.ceil((high - low) / dx))
        dxq_go = np.float32(np.float32(np.float32(high) - np.float32(low)) / np.float32(pixels))
        dxq_py = np.float32((high - low) / pixels)
        assert dxq_go == dxq_py, (low, high, dx)
        vecs = list(go_test.get((low, high, dx), []))
        L = high - low
        for n, width in ((1, 0.0), (5, 0.01), (64, 0.05), (200, 0.2), (333, 0.7), (1000, 1.0)):
            c = low + L * rng.random()
            x = c + L * width * (rng.random(n) - 0.5)
            x = low + np.mod(x - low, L)  # wrap into [low, high)
            x = np.asarray(x, dtype=np.float32)
            x = x[(x >= np.float32(low)) & (x < np.float32(high))]
            # keep only inputs that quantize strictly inside [0, pixels) under the Go text
            q = np.floor(((x - np.float32(low)) / dxq_go).astype(np.float64))
            x = x[(q >= 0) & (q < pixels)]
            if x.size:
                vecs.append(x.tolist())
        for v in vecs:
            x = np.asarray(v, dtype=np.float32)
            g = minnow._FloatGroup(0, len(x), low, high, pixels, True)
            f = io.BytesIO()
            g.write_data(f, x)
            small["float_group"].append({
                "low": low, "high": high, "dx": dx, "pixels": pixels, "py_pixels": py_pixels,
                "x_f32_hex": x.astype("<f4").tobytes().hex(),
                "min": int(g.ig.mins[0]), "bits": int(g.ig.bits[0]), "bytes": f.getvalue().hex(),
            })

    # ---- 6. group-tail arrays: write(x) of intGroup.writeTail via the reference --------
    small["tail_arrays"] = []
    for v in ([0, 10, 18, 18], [18, 18, 18], [5], [-3, 100000, 77], [12, 0, 3, 9, 9, 1]):
        g = minnow._IntGroup(0, 4)
        g.mins = list(v)
        g.bits = [0] * len(v)
        for _ in v:
            g.add_block(0)
        f = io.BytesIO()
        g.write_tail(f)
        raw = f.getvalue()
        # layout: N, startBlock, blocks (24 B) | write(mins) | write(bits)
        small["tail_arrays"].append({"x": [int(t) for t in v], "tail_hex": raw.hex()})

    # ---- 7. whole .minw files written by the reference python Writer --------------------
    files = {}
    os.makedirs("/tmp/test_files", exist_ok=True)

    def write_file(name, fn):
        path = os.path.join("/tmp/test_files", name)
        w = minnow.create(path)
        fn(w)
        w.close()
        files[name] = open(path, "rb").read().hex()

    def bit_int_record(w):  # minnow_test.go:105-122 / minnow_test.py write_bit_int_record
        import struct
        x1 = np.array([100, 101, 102, 104], dtype=np.int64)
        x2 = [np.array([1024, 1024, 1024], dtype=np.int64), np.array([0, 1023, 500], dtype=np.int64)]
        x3 = np.array([-1000000, -500000], dtype=np.int64)
        w.int_group(len(x1)); w.data(x1)
        w.header(struct.pack("<q", len(x2)))
        w.int_group(len(x2[0]))
        for x in x2:
            w.data(x)
        w.int_group(len(x3)); w.data(x3)

    def q_float_record(w):  # minnow_test.go:270-283 (limits -50..100, dx 1 and 10)
        x1 = [[-50, 0, 50, 49], [25, 25, 25, 25]]
        x2 = [[-50, 0, 50, 49, 0], [1, 2, 3, 4, 5], [0, 20, 0, 20, 0]]
        w.float_group(4, (-50, 100), 1.0)
        for x in x1:
            w.data(np.array(x, dtype=np.float32))
        w.float_group(5, (-50, 100), 10.0)
        for x in x2:
            w.data(np.array(x, dtype=np.float32))

    write_file("bit_int_record.minw", bit_int_record)
    write_file("q_float_record.minw", q_float_record)
    small["minw_files"] = files

    # ---- 8. a whole minh file written by the reference python minh.Writer (python/minh.py:36-98) -----------
    # The columns of TestReaderWriter (go/minh/minh_test.go:21-27) with two changes that keep the Python and the Go
    # derivations identical bit for bit: the logarithmic column's Dx is 0.125 (Python's float64 pixel count for
    # (10, 14, 0.01) is 401, Go's float32 one 400 -- SURVEY.md 8c) and its values keep clear of pixel edges (Python takes
    # the logarithm in float32, Go in float64).  tests/cpp/test_minh_host.cpp writes the same file and compares bytes.
    import minh  # reference python/minh.py
    mcols = [minh.Column(minnow.int64_group), minh.Column(minnow.float32_group), minh.Column(minnow.int_group),
             minh.Column(minnow.float_group, 0, 100, 200, 1), minh.Column(minnow.float_group, 1, 10, 14, 0.125)]
    for c in mcols[3:]:
        assert go_pixels(c.low, c.high, c.dx) == int(np.ceil((c.high - c.low) / c.dx))
    mblocks = [
        [np.array([100, 200, 300, 400, 500], dtype=np.int64), np.array([150, 250, 350, 450, 550], dtype=np.float32),
         np.array([-30, -35, -25, -10, -20], dtype=np.int64), np.array([100, 200, 125, 150, 100], dtype=np.float32),
         np.array([1e10, 2e11, 2e11, 1e14, 3e13], dtype=np.float32)],
        [np.array([125, 225, 325], dtype=np.int64), np.array([1750, 2750, 3750], dtype=np.float32),
         np.array([1000, 1000, 1000], dtype=np.int64), np.array([100, 100, 100], dtype=np.float32),
         np.array([1e14, 1e14, 1e14], dtype=np.float32)],
    ]
    for blk in mblocks:                                       # the two logarithms agree on the pixel of every value
        x = blk[4]
        py = np.log10(x.copy())
        go = np.log10(x.astype(np.float64)).astype(np.float32)
        q = lambda v: np.floor((np.clip(v, 10, np.nextafter(np.float32(14), np.float32(-np.inf))) - 10) / 0.125)
        assert np.array_equal(q(py), q(go))
    mtext = "Cats are the best. Don't we love them?!@#$%^&*(),....[]{};':\"|\\/-=_+`~meow meow meow"
    mpath = "/tmp/test_files/reader_writer_golden.minh"
    w = minh.create(mpath)
    w.header(["int64", "float32", "int", "float", "log"], mtext, mcols)
    w.geometry(100.0, 10.0, 4)
    for blk in mblocks:
        w.block(blk)
    w.close()
    small["minh_files"] = {"reader_writer_golden.minh": open(mpath, "rb").read().hex()}

    np.savez_compressed(os.path.join(HERE, "pack_kat.npz"), **arrays)
    with open(os.path.join(HERE, "golden_small.json"), "w") as f:
        json.dump(small, f, separators=(",", ":"))
    print("wrote", os.path.join(HERE, "pack_kat.npz"), os.path.join(HERE, "golden_small.json"))


if __name__ == "__main__":
    main()
