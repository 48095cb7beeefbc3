#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for q in 4 8; do for o in ws wsC; do
  a=$(GPU_MAX_HW_QUEUES=$q REPO_OVL=$o python bench.py --steps 30 --warmup 5 --no-cpu-baseline | sed 's/.*"value": \([0-9.]*\).*/\1/')
  b=$(GPU_MAX_HW_QUEUES=$q REPO_OVL=$o REPO_FORCE_DP=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | sed 's/.*"value": \([0-9.]*\).*/\1/')
  echo "Q=$q OVL=$o  plain $a  rccl $b"
done; done
