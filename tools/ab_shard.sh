#!/bin/bash
# same-box A/B of library variants at a strong-scaling shard's batch sizes: tools/ab_shard.sh  (edit the variant names: repo_amd/variants/lib_<v>.so)
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for b in 7 13 25; do for v in head min2000; do
  echo -n "B=$b $v: "; REPO_HIP_LIB=$R/repo_amd/variants/lib_$v.so python3 bench.py --no-cpu-baseline --batch $b --steps 50 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done; done
