#!/bin/bash
OUT=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/$OUT/$C -- python3 $R/tools/run_micro_case.py dec3_up dec3_down dec3_wgrad enc2_up > $R/$OUT/$C.log 2>&1
done
