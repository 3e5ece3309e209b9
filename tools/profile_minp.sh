#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_minp.sh <tag>
# rocprofv3 kernel-trace summary and HBM traffic counters of tools/bench_minp.py (each PMC counter in its own pass).
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out/prof_minp_$tag
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 tools/bench_minp.py --reps 5 > $out/bench_stats.json 2> $out/bench_stats.err || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 tools/bench_minp.py --reps 2 > $out/bench_fetch.json 2> $out/bench_fetch.err || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 tools/bench_minp.py --reps 2 > $out/bench_write.json 2> $out/bench_write.err || exit 1
python3 - $out <<'PY'
import csv, glob, os, sys, collections, json
out = sys.argv[1]
def find(sub, pat):
    fs = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return fs[0] if fs else None
res = collections.defaultdict(dict)
for r in csv.DictReader(open(find("stats", "*kernel_stats.csv"))):
    if "mnw::" in r["Name"]:
        res[r["Name"].split("(")[0].replace("void ", "").replace("mnw::", "")]["avg_us"] = float(r["AverageNs"]) / 1e3
        res[r["Name"].split("(")[0].replace("void ", "").replace("mnw::", "")]["calls"] = int(r["Calls"])
for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(find(sub, "*counter_collection.csv"))):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mnw::", "")
        if r["Counter_Name"] == ctr and "mnw::" in r["Kernel_Name"]:
            acc[k].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        m = sum(v) / len(v)
        res[k]["read_MB_corrected" if ctr == "FETCH_SIZE" else "write_MB"] = (2.0 if ctr == "FETCH_SIZE" else 1.0) * m * 1024 / 1e6
json.dump(res, open(os.path.join(out, "minp_kernels.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
