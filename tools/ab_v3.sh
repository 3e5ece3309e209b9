#!/bin/bash
# usage (GPU box): tools/ab_v3.sh "opt=val[,opt=val...]" lib1.so lib2.so ...  -- k_*_vec3 kernel times of several experimental builds
# (tools/build_variant.sh) under the given library options, in one call on one GPU; several option sets separated by spaces in quotes
optsets=$1; shift
for lib in "$@"; do
  for os in $optsets; do
    args=""; for kv in ${os//,/ }; do args="$args --opt $kv"; done
    echo -n "$(basename $lib) $os: "
    MNW_DIAG=1 MNW_LIB_PATH=$lib timeout -k 10 200 python tools/bench_minp.py --reps 5 $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k: round(v,1) for k,v in d['kernels_us'].items()})"
  done
done
