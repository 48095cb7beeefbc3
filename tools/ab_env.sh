#!/bin/bash
# same-box A/B of the whole update over an environment knob: tools/ab_env.sh VAR "v1 v2"
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2 3; do for v in $2; do
  echo -n "$1=$v: "; env $1=$v python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
