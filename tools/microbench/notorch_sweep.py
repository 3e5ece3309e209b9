#!/usr/bin/env python3
"""Times the library through the C ABI WITHOUT torch: device memory from hipMalloc via ctypes on the
HIP runtime the library itself is bound to.  Compare with tools/sweep.py (torch memory/runtime)."""
import ctypes as C, os, sys, time
import numpy as np
os.environ["MNW_NO_TORCH_PRELOAD"] = os.environ.get("MNW_NO_TORCH_PRELOAD", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if os.environ["MNW_NO_TORCH_PRELOAD"] != "1":
    import torch
    torch.zeros(1, device="cuda")
import minnow_amd as mnw
mnw.lib()
hip = C.CDLL("libamdhip64.so.7")  # already mapped by the library: same runtime
def malloc(nbytes):
    p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0; return p.value
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10**8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2**24
rng = np.random.default_rng(1)
x = np.mod(125 + 20 * rng.standard_normal(N, dtype=np.float32), 250).astype(np.float32); x[x >= 250] = 0
dx = malloc(4 * N); dy = malloc(4 * N)
assert hip.hipMemcpy(C.c_void_p(dx), x.ctypes.data_as(C.c_void_p), C.c_size_t(4 * N), 1) == 0
nblk = (N + B - 1) // B; slot = (4 * B + 15) // 16 * 16
dp = malloc(nblk * slot)
ctx = mnw.Context(0)
ej = [mnw.EncJob(dx + 4*k*B, dp + k*slot, min(B, N-k*B), slot, mnw.FLOAT_GROUP, 0.0, 250.0, 250000, 1) for k in range(nblk)]
metas = ctx.encode_blocks(ej)
dj = [mnw.DecJob(dp + k*slot, dy + 4*k*B, ej[k].n, metas[k].nbytes, metas[k].min, metas[k].bits, mnw.FLOAT_GROUP, 0.0, 250.0, 250000, 1, 1, k) for k in range(nblk)]
reps = 10
ctx.timing_enable(True); ctx.timing_reset()
for _ in range(reps):
    ctx.encode_blocks(ej); ctx.decode_blocks(dj)
hip.hipDeviceSynchronize()
kt = ctx.timing_get()
print("torch_preload=%s N=%d B=%d  " % (os.environ["MNW_NO_TORCH_PRELOAD"] != "1", N, B) +
      "  ".join("%s %.1fus x%d" % (k, 1e3 * v["total_ms"] / max(v["launches"], 1), v["launches"] // reps) for k, v in kt.items() if v["launches"]))
