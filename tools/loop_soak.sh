#!/bin/bash
# The two environment-loop tests (HIP-graph acting + ring pushes + pipelined updates + eval + checkpoint), N times each in
# FRESH processes per environment variant; prints passes / runs per variant.  tools/loop_soak.sh [N_main] [N_other]
N1=${1:-50}; N2=${2:-15}
K='loops_on_fake_env or mt_train_eval_loops'
run() {  # name count env...
  local name=$1 n=$2; shift 2
  local ok=0
  for i in $(seq $n); do
    if env "$@" python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "$K" > /tmp/loop_soak_last.txt 2>&1; then ok=$((ok+1));
    else echo "--- $name run $i FAILED"; tail -25 /tmp/loop_soak_last.txt; fi
  done
  echo "== $name: $ok / $n runs green (2 tests per run)"
}
run "poison (default)" $N1 REPO_TEST_POISON=1
run "no poison" $N1 REPO_TEST_POISON=0
run "poison, REPO_ACT_GRAPH=0" $N2 REPO_ACT_GRAPH=0
run "poison, AMD_SERIALIZE_KERNEL=3" $N2 AMD_SERIALIZE_KERNEL=3
# (torch cannot capture a graph without its caching allocator: the no-caching variant runs the acting path eagerly)
run "poison, PYTORCH_NO_HIP_MEMORY_CACHING=1 REPO_ACT_GRAPH=0" $N2 PYTORCH_NO_HIP_MEMORY_CACHING=1 REPO_ACT_GRAPH=0
