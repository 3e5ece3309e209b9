#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_sq.sh <outdir> <program> [args...]
# One rocprofv3 pass with the eight SQ counters that split a kernel's wave-cycles into issuing / parked / stalled
# (MI355X_MICROARCH.md "rocprofv3 PMC slots"), --kernel-trace only beside it.  Prints per-kernel means and the fractions
#   valu   = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES      (share of its life a wave spends issuing vector ALU work)
#   any    = SQ_ACTIVE_INST_ANY  / SQ_WAVE_CYCLES
#   parked = SQ_WAIT_ANY         / SQ_WAVE_CYCLES      (s_waitcnt / barrier: waiting for memory or for other waves)
#   stall  = SQ_WAIT_INST_ANY    / SQ_WAVE_CYCLES      (issue stalls)
out=$1; shift
root=$(pwd)
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES \
    --output-format csv -d $out -- "$@" > $out/stdout.log 2> $out/stderr.log || { tail -5 $out/stderr.log; exit 1; }
python3 - $out <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "mnw::" in k:
        agg[k.replace("mnw::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    res[k] = {"launches": len(next(iter(d.values()))), **{c: m[c] for c in sorted(m)},
              "valu": m.get("SQ_ACTIVE_INST_VALU", 0) / wc, "any": m.get("SQ_ACTIVE_INST_ANY", 0) / wc,
              "parked": m.get("SQ_WAIT_ANY", 0) / wc, "stall": m.get("SQ_WAIT_INST_ANY", 0) / wc,
              "valu_insts_per_wave": m.get("SQ_INSTS_VALU", 0) / (m.get("SQ_WAVES", 0) or 1)}
json.dump(res, open(out + "/sq_summary.json", "w"), indent=1)
for k, v in sorted(res.items()):
    print("%-28s n=%-3d valu %.2f any %.2f parked %.2f stall %.2f  VALU insts/wave %.0f  waves %.0f" %
          (k, v["launches"], v["valu"], v["any"], v["parked"], v["stall"], v["valu_insts_per_wave"], v.get("SQ_WAVES", 0)))
PY
