#!/bin/bash
# rocprofv3 kernel stats of a short bench run; summary copied to gpurun_out/<tag>_kernel_stats.csv
# usage: tools/prof_bench.sh <tag> [bench args...]     (e.g. tools/prof_bench.sh r02_join --join)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pb_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_$TAG -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" > /tmp/pb_$TAG.log 2>&1
grep '^{' /tmp/pb_$TAG.log | cut -c1-400
mkdir -p $R/gpurun_out
cp $(ls /tmp/pb_$TAG/*/*_kernel_stats.csv | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
grep '^{' /tmp/pb_$TAG.log > $R/gpurun_out/${TAG}_bench_line.json
