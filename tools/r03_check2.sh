#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python3 -m pytest tests/test_host_gpu.py -m gpu -x -q -k "bucket or shard" > $O/r03_tests2.log 2>&1; tail -3 $O/r03_tests2.log
for b in 2 1 2 1; do
REPO_DP_BUCKETS=$b REPO_FORCE_DP=1 timeout 300 python3 bench.py --no-cpu-baseline --steps 50 2>/dev/null | grep '^{' | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('buckets $b', d['ms_per_step'], d.get('allreduce_ms',{}).get('per_update_by_bucket_bytes'))"
done
timeout 300 python3 bench.py --no-cpu-baseline --steps 50 2>/dev/null | grep '^{' | cut -c1-200
