#!/bin/bash
# The bench lines and the scan table only (no PMC / isolated tables): after a change that moves the update but no kernel
# table.  gpurun -- 'bash tools/refresh_bench_only.sh'; then python tools/collect_profiles.py --write-json
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
(python3 bench.py --config c4 | grep '^{'; python3 bench.py --config c5 | grep '^{') > $O/r05_bench_c4_c5.json 2>/dev/null
(for b in 7 6 13 25; do python3 bench.py --no-cpu-baseline --batch $b --steps 50 | grep '^{'; done) > $O/r05_bench_shards.json 2>/dev/null
python3 tools/scan_cs_time.py > $O/r05_scan_cs.txt 2>&1
python3 tools/lane_time.py > $O/r05_lane_time.txt 2>&1
python3 tools/phase_time.py > $O/r05_phase_time.txt 2>&1
bash tools/prof_bench.sh r05_bench_pipelined > /dev/null 2>&1
python3 tools/layers_in_update.py $O/r05_bench_pipelined_kernel_stats.csv --json $O/dominant_kernel_rocprof.json > $O/r05_layers_in_update.txt 2>&1
python3 tools/launch_count.py $O/r05_bench_pipelined_kernel_stats.csv > $O/r05_launch_count.txt 2>&1
python3 bench.py --config c4x128 2>/dev/null | grep '^{' > $O/r05_bench_c4x128.json
python3 bench.py --config tia 2>/dev/null | grep '^{' > $O/r05_bench_tia.json
python3 bench.py > $O/bench_full.log 2>&1
grep '^{' $O/bench_full.log > $O/r05_bench_final.json
tail -c 300 $O/r05_bench_final.json
