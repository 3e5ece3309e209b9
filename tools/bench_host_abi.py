#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry points (what the cgo shim binds: Go slices in, bytes out), called straight
through the C ABI on preallocated pageable numpy buffers that have been written before -- like the reference's grow-only
buffers (bit.ArrayBuffer, go/bit/bit.go:183-205; a fresh buffer would add a page fault per 4 KiB to the first call):
mnw_float_encode / mnw_float_decode on 1e8 float32, the int64 pair on 5e7.  Prints one JSON object."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import minnow_amd as mnw  # noqa: E402

ctx = mnw.Context(0)
L, h = ctx._L, ctx.handle
rng = np.random.default_rng(1)


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def best(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return min(ts), sum(ts) / len(ts)


res = {}
n = 10**8
x = np.mod(125 + 20 * rng.standard_normal(n), 250).astype(np.float32)
x[x >= np.float32(250)] = 0
pixels = mnw.float_group_pixels(0.0, 250.0, 0.001)
out = np.zeros(4 * n + 16, dtype=np.uint8)
out[:] = 1                                                    # touched
y = np.ones(n, dtype=np.float32)
mn, bits, olen = C.c_int64(), C.c_int64(), C.c_int64()


def enc():
    rc = L.mnw_float_encode(h, ptr(x), n, 0.0, 250.0, pixels, 1, C.byref(mn), C.byref(bits), ptr(out), out.size, C.byref(olen))
    assert rc == 0, rc


def dec():
    rc = L.mnw_float_decode(h, ptr(out), olen.value, mn.value, bits.value, 0.0, 250.0, pixels, 1, 1, 0, n, ptr(y))
    assert rc == 0, rc


te, te_avg = best(enc)
td, td_avg = best(dec)
d = np.abs(y.astype(np.float64) - x); d = np.minimum(d, 250 - d); assert d.max() <= 0.00102
res["float32_1e8"] = {"encode_s": te, "decode_s": td, "encode_GBps_input": 4.0 * n / te / 1e9, "decode_GBps_output": 4.0 * n / td / 1e9,
                      "encode_GBps_input_mean": 4.0 * n / te_avg / 1e9, "decode_GBps_output_mean": 4.0 * n / td_avg / 1e9,
                      "bits": bits.value, "stream_bytes": olen.value,
                      "link_bytes_per_call": 4 * n + olen.value}
m = 5 * 10**7
xi = (10**9 + rng.integers(0, 2**27, m)).astype(np.int64)
outi = np.ones(8 * m + 16, dtype=np.uint8)
yi = np.ones(m, dtype=np.int64)


def enci():
    assert L.mnw_int_encode(h, ptr(xi), m, C.byref(mn), C.byref(bits), ptr(outi), outi.size, C.byref(olen)) == 0


def deci():
    assert L.mnw_int_decode(h, ptr(outi), olen.value, mn.value, bits.value, m, ptr(yi)) == 0


te, _ = best(enci)
td, _ = best(deci)
assert np.array_equal(xi, yi)
res["int64_5e7"] = {"encode_s": te, "decode_s": td, "encode_GBps_input": 8.0 * m / te / 1e9, "decode_GBps_output": 8.0 * m / td / 1e9, "bits": bits.value}
# BASELINE configs[0]: one int64 block of 1e6 elements, x = 1000 + U{0..2^b-1} -- what one Writer.Data call costs
m1 = 10 ** 6
for b1 in (11, 21):
    x1 = (1000 + rng.integers(0, 2 ** b1, size=m1, dtype=np.int64))
    y1 = np.empty_like(x1)
    o1 = np.empty(8 * m1 + 64, dtype=np.uint8)

    def enc1():
        assert L.mnw_int_encode(h, ptr(x1), m1, C.byref(mn), C.byref(bits), ptr(o1), o1.size, C.byref(olen)) == 0

    def dec1():
        assert L.mnw_int_decode(h, ptr(o1), olen.value, mn.value, bits.value, m1, ptr(y1)) == 0

    te, _ = best(enc1)
    td, _ = best(dec1)
    assert np.array_equal(x1, y1)
    res["int64_1e6_b%d" % b1] = {"encode_us": te * 1e6, "decode_us": td * 1e6, "encode_GBps_input": 8.0 * m1 / te / 1e9,
                                 "decode_GBps_output": 8.0 * m1 / td / 1e9, "bits": bits.value}
print(json.dumps(res))
