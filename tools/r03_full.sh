#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do timeout 300 python3 bench.py --no-cpu-baseline --steps 50 2>/dev/null | grep '^{' | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['value'], 'dom', d['roofline']['ms_per_launch'], d['roofline'].get('isolated_ms_per_launch'))"; done
