#!/bin/bash
# usage (GPU box, repo root): tools/pmc_any.sh <outdir> "<counters of pass 1>;<counters of pass 2>;..." <python script> [args...]
# One rocprofv3 pass (--kernel-trace + --pmc only, as gpurun requires) per ';'-separated counter group; prints per-kernel means
# of every counter as one JSON object {kernel: {counter: mean, ..., "launches": n}} and writes it to <outdir>/pmc_summary.json.
out=$1; shift
groups=$1; shift
root=$(pwd)
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
i=0
IFS=';' read -ra G <<< "$groups"
for g in "${G[@]}"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $out/p$i -- python3 "$@" > $out/p$i.out 2> $out/p$i.err || { echo "pass $i ($g) failed"; tail -3 $out/p$i.err; }
done
python3 - $out <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "mnw::" in k:
            agg[k.replace("mnw::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: dict({c: sum(v) / len(v) for c, v in d.items()}, launches=max(len(v) for v in d.values())) for k, d in agg.items()}
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
