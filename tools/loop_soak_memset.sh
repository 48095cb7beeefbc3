#!/bin/bash
# link check (round 6): the round 3-5 form of the scans' arming (hipMemsetAsync nodes; repo_amd/variants/lib_memset.so =
# SRC=scan_cs tools/build_variant.sh memset -DCS_MEMSET_NODES) under the loop tests WITHOUT the poison, N fresh processes
N=${1:-30}
ok=0
for i in $(seq $N); do
  if REPO_HIP_LIB=repo_amd/variants/lib_memset.so REPO_TEST_POISON=0 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k 'loops_on_fake_env or mt_train_eval_loops' > /tmp/l.txt 2>&1; then ok=$((ok+1)); else echo "--- run $i FAILED"; grep -E "^E  |FAILED" /tmp/l.txt | head -8; fi
done
echo "== memset nodes, no poison: $ok / $N runs green"
echo "== memset nodes, poisoned torch.empty, AMD_SERIALIZE_KERNEL=3 (tools/act_graph_debug.py):"
AMD_SERIALIZE_KERNEL=3 POISON=1 REPO_HIP_LIB=repo_amd/variants/lib_memset.so python tools/act_graph_debug.py 2>&1 | grep -v amdgpu.ids
echo "== memset nodes, poisoned torch.empty, default launch mode:"
POISON=1 REPO_HIP_LIB=repo_amd/variants/lib_memset.so python tools/act_graph_debug.py 2>&1 | grep -v amdgpu.ids
