#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_round.sh into the small files kept under profiles/<tag>/:
kernel_stats.csv (our kernels only), pmc_fetch_size.csv / pmc_write_size.csv (per-dispatch counter values of
our kernels) and pmc_traffic.json (HBM bytes per launch, corrected as MI355X_MICROARCH.md prescribes)."""
import collections
import re
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(out, "summary")
os.makedirs(dst, exist_ok=True)


def find(sub, pat):
    fs = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return fs[0] if fs else None


# ---- kernel-trace stats -----------------------------------------------------------------------------
f = find("stats", "*kernel_stats.csv")
rows = list(csv.DictReader(open(f)))
with open(os.path.join(dst, "kernel_stats.csv"), "w", newline="") as g:
    w = csv.DictWriter(g, fieldnames=rows[0].keys())
    w.writeheader()
    for r in rows:
        if "mnw::" in r["Name"]:
            r["Name"] = r["Name"].split("(")[0].replace("void ", "")
            w.writerow(r)

# ---- PMC passes ---------------------------------------------------------------------------------------
per = collections.defaultdict(lambda: collections.defaultdict(list))
for sub, ctr, name in (("fetch", "FETCH_SIZE", "pmc_fetch_size.csv"), ("write", "WRITE_SIZE", "pmc_write_size.csv")):
    f = find(sub, "*counter_collection.csv")
    with open(os.path.join(dst, name), "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value"])
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "mnw::" not in k or r["Counter_Name"] != ctr:
                continue
            w.writerow([r["Dispatch_Id"], k, r["Grid_Size"], r["Workgroup_Size"], ctr, r["Counter_Value"]])
            per[re.sub(r"<.*", "", k.replace("void ", "").replace("mnw::", ""))][ctr].append(float(r["Counter_Value"]))

kern = {}
for k, d in per.items():
    fs, ws = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
    if not fs or not ws:
        continue
    # drop the largest-block config's launches from the per-launch mean of a kernel only if sizes differ a lot:
    # bench.py runs the headline block size first and the large-block variant second, with different kernels
    fk, wk = sum(fs) / len(fs), sum(ws) / len(ws)
    kern[k] = {"FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk,
               "read_bytes_corrected": 2.0 * fk * 1024.0,      # gfx950: FETCH_SIZE reports half the bytes
               "write_bytes": wk * 1024.0, "launches_sampled": len(fs)}

res = {
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 bench.py --steps 2 "
              "--warmup 1 --legs headline --no-cpu-baseline --no-kernel-timing` (one 1e9-element float32 column, blocks of 2^20), "
              "%s; FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md HBM), WRITE_SIZE "
              "as is; both counters are in KiB" % tag,
    "kernels": kern,
}
enc = kern.get("k_encode_reg_f32")
if enc:
    res["encode_bytes_per_launch"] = enc["read_bytes_corrected"] + enc["write_bytes"]
dec = kern.get("k_unpack_f32")
if dec:
    res["decode_bytes_per_launch"] = dec["read_bytes_corrected"] + dec["write_bytes"]
res["algorithmic_encode_bytes_per_launch"] = 6250000000.0      # (4 + 18/8) B x 1e9 elements
res["algorithmic_decode_bytes_per_launch"] = 6250000000.0
json.dump(res, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: (v["read_bytes_corrected"] / 1e6, v["write_bytes"] / 1e6) for k, v in kern.items()}, indent=1))
for r in csv.DictReader(open(os.path.join(dst, "kernel_stats.csv"))):
    print(r["Name"], r["Calls"], r.get("AverageNs"))
