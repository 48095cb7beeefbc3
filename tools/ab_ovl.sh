#!/bin/bash
# REPO_OVL variants of the critic's fork (C: borrows the world-model lane's weight-gradient stream, c: its own stream, none: in line)
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2 3 4 5; do for v in wsC wsc ws; do
  echo -n "REPO_OVL=$v B=50: "; REPO_OVL=$v python3 bench.py --no-cpu-baseline --steps 60 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
for rep in 1 2; do for v in wsC wsc ws; do
  echo -n "REPO_OVL=$v B=7: "; REPO_OVL=$v python3 bench.py --no-cpu-baseline --batch 7 --steps 60 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  echo -n "REPO_OVL=$v c5 (dreamer): "; REPO_OVL=$v python3 bench.py --no-cpu-baseline --config c5 --steps 40 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
