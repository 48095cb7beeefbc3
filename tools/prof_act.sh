#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pa
REPO_ACT_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -- python3 $R/tools/act_time.py > /tmp/pa.log 2>&1
tail -2 /tmp/pa.log
python3 - <<PY
import csv, glob, re
f = glob.glob('/tmp/pa/*/*_kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6, "per step us", tot/1e3/440)
for r in rows[:22]:
    n = re.sub(r'repo::|void ', '', r['Name'])[:80]
    print(f"{float(r['TotalDurationNs'])/1e3/440:8.1f} us/step {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:8.1f}us  {n}")
PY
