#!/bin/bash
# same-box A/B of the whole update: tools/ab_bench.sh "<variants>"  (repo_amd/variants/lib_<v>.so; `base` = a copy of the shipped library)
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2 3; do for v in ${1:-base notw}; do
  echo -n "$v: "; REPO_HIP_LIB=$R/repo_amd/variants/lib_$v.so python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
