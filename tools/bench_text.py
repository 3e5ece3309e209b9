#!/usr/bin/env python3
"""Text ingest of a Rockstar-like catalogue block (BASELINE configs[4] shape, one block): ROWS lines of 44 columns.
Prints one JSON object: GB/s of text for the whole call (host text in, device columns out: includes the pageable
host-to-device copy) and for the kernels alone (HIP events)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import minnow_amd as mnw
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2 * 10**6
ncol = 44
rng = np.random.default_rng(3)
ids = rng.integers(0, 10**11, (rows, 4))
logm = 10 ** rng.uniform(8, 15, (rows, 20))
pos = rng.uniform(0, 250, (rows, 10))
vel = rng.normal(0, 300, (rows, 10))
t0 = time.time()
cols = [ids[:, k].astype(str) for k in range(4)] + [np.char.mod("%.4e", logm[:, k]) for k in range(20)] + \
       [np.char.mod("%.5f", pos[:, k]) for k in range(10)] + [np.char.mod("%.2f", vel[:, k]) for k in range(10)]
lines = cols[0]
for c in cols[1:]:
    lines = np.char.add(np.char.add(lines, " "), c)
data = ("#header " + " ".join("c%d" % k for k in range(ncol)) + "\n" + "\n".join(lines.tolist()) + "\n").encode()
dev = torch.device("cuda:0")
ctx = mnw.Context(0)
fcols, icols = list(range(4, 44)), list(range(4))
fo = [torch.empty(rows + 8, dtype=torch.float32, device=dev) for _ in fcols]
io = [torch.empty(rows + 8, dtype=torch.int64, device=dev) for _ in icols]
torch.cuda.synchronize()


def run():
    return ctx.text_parse(data, ord(" "), ord("#"), fcols, [t.data_ptr() for t in fo], icols, [t.data_ptr() for t in io], rows + 8)


n = run(); assert n == rows
ts = []
for _ in range(3):
    t = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
ctx.timing_enable(True); ctx.timing_reset(); run(); torch.cuda.synchronize()
kt = {k: v["total_ms"] for k, v in ctx.timing_get().items() if v["launches"]}
if not os.environ.get("MNW_NO_CHECK"):      # (diagnostic builds that skip the conversions)
    assert np.allclose(fo[0][:1000].cpu().numpy(), np.float32(np.char.mod("%.4e", logm[:1000, 0]).astype(np.float64)), rtol=0, atol=0)
    assert np.array_equal(io[0][:1000].cpu().numpy(), ids[:1000, 0])
kms = sum(kt.values())
print(json.dumps({"rows": rows, "columns": ncol, "text_GB": len(data) / 1e9, "bytes_per_row": len(data) / rows,
                  "call_s": min(ts), "call_GBps_text": len(data) / min(ts) / 1e9,
                  "kernel_ms": kt, "kernels_GBps_text": len(data) / (kms * 1e-3) / 1e9,
                  "rows_per_s_call": rows / min(ts), "output_GB": (40 * 4 + 4 * 8) * rows / 1e9}))
