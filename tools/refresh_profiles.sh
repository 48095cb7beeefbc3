#!/bin/bash
# Everything under profiles/${RND}_* from the CURRENT kernels, in one call on the GPU box:
#   gpurun --timeout 3000 -- 'bash tools/refresh_profiles.sh'      then      python tools/collect_profiles.py --write-json
RND=${RND:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
python3 tools/layers_isolated.py > $O/${RND}_layers_isolated.txt 2>&1
bash tools/prof_bench.sh ${RND}_bench_pipelined > /dev/null 2>&1
bash tools/prof_bench.sh ${RND}_bench_join --join > /dev/null 2>&1
python3 tools/layers_in_update.py $O/${RND}_bench_pipelined_kernel_stats.csv --json $O/dominant_kernel_rocprof.json --csv-name profiles/${RND}_bench_kernel_stats_pipelined.csv > $O/${RND}_layers_in_update.txt 2>&1
python3 tools/launch_count.py $O/${RND}_bench_pipelined_kernel_stats.csv > $O/${RND}_launch_count.txt 2>&1
bash tools/pmc.sh dec3 "buconv_scatter|uconv_scatter|tconv_down|bconv_down|dconv_down|tconv_wgrad|bconv_wgrad|dconv_wgrad|conv_slab_reduce" tools/run_micro_case.py "conv dec3" > /dev/null 2>&1
bash tools/pmc.sh convs "buconv_scatter|uconv_scatter|tconv_up|tconv_down|bconv_down|dconv_down|tconv_wgrad|bconv_wgrad|dconv_wgrad" tools/run_micro_case.py "conv enc2" "conv enc3" "conv enc4" "conv dec2" > /dev/null 2>&1
bash tools/pmc.sh scan_rollout "observe_|imagine" tools/run_scan_rollout.py > /dev/null 2>&1
bash tools/pmc.sh c3 "dconv_dec4|bdec4|Geo<3, 32|Geo<3,32" tools/run_micro_case.py "conv enc1" "conv dec4" "dec4 forward" > /dev/null 2>&1
bash tools/pmc.sh mlp "mlp_(fwd|bwd)_kernel|wgrad_direct" tools/run_micro_case.py "mlp_fwd value" "mlp_bwd value head" "mlp_bwd actor" > /dev/null 2>&1
python3 tools/lane_time.py > $O/${RND}_lane_time.txt 2>&1
python3 tools/phase_time.py > $O/${RND}_phase_time.txt 2>&1
(python3 bench.py --config c4 | grep '^{'; python3 bench.py --config c5 | grep '^{') > $O/${RND}_bench_c4_c5.json 2>/dev/null
(for b in 7 6 13 25; do python3 bench.py --no-cpu-baseline --batch $b --steps 50 | grep '^{'; done) > $O/${RND}_bench_shards.json 2>/dev/null
python3 tools/scan_cs_time.py > $O/${RND}_scan_cs.txt 2>&1
ISO_IMAGE=128 python3 tools/layers_isolated.py @128 dec3 > $O/${RND}_layers_isolated_128.txt 2>&1
bash tools/prof_bench.sh ${RND}_c4x128 --config c4x128 > /dev/null 2>&1
python3 tools/layers_in_update.py $O/${RND}_c4x128_kernel_stats.csv --nimg 1568 > $O/${RND}_c4x128_layers_in_update.txt 2>&1
python3 bench.py --config c4x128 2>/dev/null | grep '^{' > $O/${RND}_bench_c4x128.json
python3 bench.py --config tia 2>/dev/null | grep '^{' > $O/${RND}_bench_tia.json
python3 bench.py --config mt 2>/dev/null | grep '^{' > $O/${RND}_bench_mt.json
tools/probe/bin/bgemm_probe > $O/${RND}_bgemm_probe.txt 2>&1
python3 tools/rowtile32_ab.py > $O/${RND}_rollout_engines.txt 2>&1
python3 tools/layers_isolated.py gemm > $O/${RND}_gemm_isolated.txt 2>&1
# the round's gain on ONE box: the round-5 tree (git archive 6f1b639 -> .r5tree, built in place) against this tree, alternating
if [ -f .r5tree/bench.py ]; then
  (for i in 1 2 3; do
     echo -n "r5 tree : "; (cd .r5tree && python3 bench.py --no-cpu-baseline --steps 60 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'updates/s', d['ms_per_step'], 'ms', 'resident', d['resident_batch_ms'])")
     echo -n "this tree: "; python3 bench.py --no-cpu-baseline --steps 60 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'updates/s', d['ms_per_step'], 'ms', 'resident', d['resident_batch_ms'])"
   done) > $O/${RND}_ab_vs_round5.txt 2>&1
fi
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke" > $O/${RND}_smoke.txt
python3 -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" > $O/${RND}_gpu_tests.txt
python3 tools/layers_isolated.py "mlp_bwd" > $O/${RND}_heads_isolated.txt 2>&1
bash tools/pmc.sh wtr "wgrad_tr|wgrad_direct" tools/run_micro_case.py "mlp_bwd value head weight" "mlp_bwd actor trunk weight" > /dev/null 2>&1
# the bench line names the trace's top kernel from profiles/dominant_kernel_*.json: bring them up to date first
python3 tools/collect_profiles.py --write-json > /dev/null 2>&1
python3 bench.py > $O/bench_full.log 2>&1
grep '^{' $O/bench_full.log > $O/${RND}_bench_final.json
tail -c 700 $O/${RND}_bench_final.json
