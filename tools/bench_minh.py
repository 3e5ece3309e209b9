#!/usr/bin/env python3
"""minh column transforms on device columns: processFloatGroup (clamp only / log10 + clamp) and the reader's 10^x.
One 1e8-element column each, and 64 columns of 2^20 in one call.  Prints one JSON object."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import minnow_amd as mnw
dev = torch.device("cuda:0")
ctx = mnw.Context(0)
g = torch.Generator(device=dev); g.manual_seed(5)
res = {}


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


for name, n, ncol in (("one_column_1e8", 10**8, 1), ("64_columns_2p20", 2**20, 64)):
    x = 10 ** (8 + 8 * torch.rand(n * ncol, device=dev, generator=g))
    y = torch.empty_like(x)
    cols_log = [(x.data_ptr() + 4 * n * k, y.data_ptr() + 4 * n * k, n, 8.0, 16.0, 1) for k in range(ncol)]
    cols_lin = [(x.data_ptr() + 4 * n * k, y.data_ptr() + 4 * n * k, n, 1e9, 1e15, 0) for k in range(ncol)]
    torch.cuda.synchronize()
    t_lin = timed(lambda: ctx.minh_process(cols_lin))
    t_log = timed(lambda: ctx.minh_process(cols_log))
    ctx.minh_process(cols_log); torch.cuda.synchronize()
    z = y.clone()
    un = [(z.data_ptr() + 4 * n * k, n) for k in range(ncol)]

    def unlog():
        z.copy_(y)
        ctx.minh_unlog(un)
    t_copy = timed(lambda: z.copy_(y))
    t_un = timed(unlog) - t_copy
    tot = n * ncol
    res[name] = {"elements": tot, "clamp_us": t_lin * 1e6, "clamp_TBps_of_8B_per_elt": 8.0 * tot / t_lin / 1e12,
                 "log10_clamp_us": t_log * 1e6, "log10_clamp_Gelt_per_s": tot / t_log / 1e9,
                 "pow10_us": t_un * 1e6, "pow10_Gelt_per_s": tot / t_un / 1e9}
    del x, y, z
print(json.dumps(res))
