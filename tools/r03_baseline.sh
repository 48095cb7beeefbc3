#!/bin/bash
# round-3 start: tests + bench + PMC of the weak conv layers (VERDICT r2 next #1a)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python3 -m pytest tests -m gpu -x -q > $O/r03_start_tests.log 2>&1; echo "tests rc $?" >> $O/r03_start_tests.log
timeout 300 python3 bench.py > $O/r03_start_bench.log 2>&1
grep '^{' $O/r03_start_bench.log > $O/r03_bench_round_start.json
timeout 400 python3 tools/layers_isolated.py > $O/r03_layers_isolated_round_start.txt 2>&1
timeout 900 bash tools/pmc.sh convs_weak "uconv_scatter|dconv_down|dconv_wgrad" tools/run_micro_case.py "conv enc2" "conv enc3" "conv enc4" "conv dec2" > /dev/null 2>&1
tail -3 $O/r03_start_tests.log; tail -c 400 $O/r03_bench_round_start.json
