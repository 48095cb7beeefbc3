#!/bin/bash
# quick loop for conv kernel work: parity of the conv ops, then the isolated table rows named on the command line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python3 -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv or decoder_out" 2>&1 | tail -3
timeout 600 python3 tools/layers_isolated.py "$@" 2>&1 | grep -v "^#\|amdgpu.ids"
