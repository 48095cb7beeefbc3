#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python3 -m pytest tests/test_rssm_gpu.py -m gpu -x -q -k "observe or philox or update_with" 2>&1 | tail -3
bash tools/ab.sh "$1" "observe scan"
