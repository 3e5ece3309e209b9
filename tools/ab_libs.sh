#!/bin/bash
# usage (GPU box): tools/ab_libs.sh <block log2> <particles> lib1.so lib2.so ...   -- the float encode / decode kernel times of several
# builds of the library (MNW_LIB_PATH), interleaved twice, in one call (same GPU, same minute)
lb=$1; n=$2; shift 2
for rnd in 1 2; do
  for lib in "$@"; do
    echo -n "$(basename $lib): "
    MNW_LIB_PATH=$lib AB_VALUES=1 timeout -k 10 200 python tools/ab_option.py single_read $lb $n 2>/dev/null | tail -1
  done
done
