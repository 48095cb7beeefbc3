#!/bin/bash
# Everything under profiles/r02_* from the CURRENT kernels, in one call on the GPU box:
#   gpurun --timeout 2400 -- 'bash tools/refresh_profiles.sh'      then      python tools/collect_profiles.py
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
python3 tools/layers_isolated.py > $O/layers_isolated.txt 2>&1
bash tools/prof_bench.sh r02_bench_pipelined > /dev/null 2>&1
bash tools/prof_bench.sh r02_bench_join --join > /dev/null 2>&1
bash tools/pmc.sh dec3 "uconv_scatter|dconv_down|dconv_wgrad" tools/run_micro_case.py "conv dec3" > /dev/null 2>&1
bash tools/pmc.sh scan_rollout "observe_|imagine_" tools/run_scan_rollout.py > /dev/null 2>&1
bash tools/pmc.sh c3 "dconv_dec4|Geo<3, 32|Geo<3,32|wgrad" tools/run_micro_case.py "conv enc1" "conv dec4" "dec4 forward" > /dev/null 2>&1
bash tools/pmc.sh mlp "mlp_(fwd|bwd)_kernel" tools/run_micro_case.py "mlp_fwd value" "mlp_bwd value head input" > /dev/null 2>&1
python3 tools/lane_time.py > $O/lane_time.txt 2>&1
(python3 bench.py --config c4 --no-cpu-baseline | grep '^{'; python3 bench.py --config c5 --no-cpu-baseline | grep '^{') > $O/r02_bench_c4_c5.json 2>/dev/null
python3 bench.py > $O/bench_full.log 2>&1
grep '^{' $O/bench_full.log > $O/r02_bench_final.json
tail -c 600 $O/r02_bench_final.json
