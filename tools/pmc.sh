#!/bin/bash
# usage: tools/pmc.sh <outdir> "<counters>" -- collects one PMC pass of tools/sweep.py (2^24 blocks)
# (run on the GPU box; counters in their own pass with --kernel-trace only, as gpurun requires)
out=$1; shift
ctrs=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -- python3 tools/sweep.py --blocks 24 --grids 0 --groups 96 --reps 3 > $out.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    if "mnw::" not in k: continue
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s n=%-4d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
