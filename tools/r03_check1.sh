#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests -m gpu -x -q > $O/r03_tests1.log 2>&1; echo "tests rc $?" >> $O/r03_tests1.log
tail -5 $O/r03_tests1.log
REPO_FORCE_DP=1 timeout 300 python3 bench.py --no-cpu-baseline --steps 30 > $O/r03_dp1.log 2>&1; grep '^{' $O/r03_dp1.log | cut -c1-300; tail -3 $O/r03_dp1.log | cut -c1-600
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/envp -- python3 -c "import os; print({k:v for k,v in os.environ.items() if 'ROC' in k.upper() or 'PRELOAD' in k or 'HSA' in k})" > $O/r03_profenv.log 2>&1
grep -v "^[WE]2026" $O/r03_profenv.log | tail -3
