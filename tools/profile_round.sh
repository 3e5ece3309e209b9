#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
# Collects what profiles/<tag>/ holds for the HEADLINE workload of bench.py (one 1e9-element float32 column, blocks of
# 2^20, `--legs headline`): the rocprofv3 kernel-trace summary and the HBM traffic counters, each PMC counter in a pass of
# its own with --kernel-trace only (the pool refuses PMC combined with any other trace domain).  The program after `--`
# is python3 itself (no wrapper).
tag=${1:-r03}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 10 --warmup 2 --legs headline --no-cpu-baseline --no-stream-ceiling > $out/bench_stats.json 2> $out/bench_stats.err || exit 1
echo "stats pass done"
timeout -k 10 280 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 2 --warmup 1 --legs headline --no-cpu-baseline --no-stream-ceiling --no-kernel-timing > $out/bench_fetch.json 2> $out/bench_fetch.err || exit 1
echo "FETCH_SIZE pass done"
timeout -k 10 280 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 2 --warmup 1 --legs headline --no-cpu-baseline --no-stream-ceiling --no-kernel-timing > $out/bench_write.json 2> $out/bench_write.err || exit 1
echo "WRITE_SIZE pass done"
python3 tools/pmc_summarize.py $out $tag
