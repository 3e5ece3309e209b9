#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry points (what the cgo shim binds: Go slices in, bytes out):
mnw_float_encode / mnw_float_decode on a pageable numpy array of 1e8 float32, and the int64 pair on 5e7."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import minnow_amd as mnw
from minnow_amd import groups

ctx = mnw.Context(0)
rng = np.random.default_rng(1)
n = 10**8
x = np.mod(125 + 20 * rng.standard_normal(n), 250).astype(np.float32)
x[x >= np.float32(250)] = 0
fg = groups.FloatGroupCodec.from_limits(ctx, n, (0.0, 250.0), 0.001)
res = {}
for rep in range(3):
    t = time.perf_counter(); data = fg.write_data(x); te = time.perf_counter() - t
    t = time.perf_counter(); y = fg.read_data(data, 0, seed=1); td = time.perf_counter() - t
res["float32_1e8"] = {"encode_s": te, "decode_s": td, "encode_GBps_input": 4.0 * n / te / 1e9, "decode_GBps_output": 4.0 * n / td / 1e9}
d = np.abs(y.astype(np.float64) - x); d = np.minimum(d, 250 - d); assert d.max() <= 0.00102
m = 5 * 10**7
xi = (10**9 + rng.integers(0, 2**27, m)).astype(np.int64)
ig = groups.IntGroupCodec(ctx, m)
for rep in range(3):
    t = time.perf_counter(); data = ig.write_data(xi); te = time.perf_counter() - t
    t = time.perf_counter(); yi = ig.read_data(data, 0); td = time.perf_counter() - t
assert np.array_equal(xi, yi)
res["int64_5e7"] = {"encode_s": te, "decode_s": td, "encode_GBps_input": 8.0 * m / te / 1e9, "decode_GBps_output": 8.0 * m / td / 1e9}
print(json.dumps(res))
