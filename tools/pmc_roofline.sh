#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes as MI355X_MICROARCH.md prescribes) of the
# dominant kernel (decoder conv3 forward) + kernel-trace stats of the full bench.
# usage: tools/pmc_roofline.sh <outdir-under-repo>
OUT=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/fetch -- python3 $R/tools/run_micro_case.py dec3_up > $R/$OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$OUT/write -- python3 $R/tools/run_micro_case.py dec3_up > $R/$OUT/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/$OUT/bench.log 2>&1
