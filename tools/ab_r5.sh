#!/bin/bash
# same-box A/B of the whole update: the round-5 tree (.r5tree = git archive 6f1b639, built in place) against this tree, alternating
N=${1:-3}
for i in $(seq $N); do
  echo -n "r5 tree  : "; (cd .r5tree && python3 bench.py --no-cpu-baseline --steps 60 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'updates/s', d['ms_per_step'], 'ms')")
  echo -n "this tree: "; python3 bench.py --no-cpu-baseline --steps 60 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'updates/s', d['ms_per_step'], 'ms')"
done
