#!/bin/bash
# kernel-trace the microbench cases and print median GPU-side kernel durations
# usage: tools/prof_micro.sh <tag> <case> [case ...]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pm_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/pm_$TAG -- python3 $R/tools/microbench.py "$@" > /dev/null 2>&1
python3 - <<PY
import csv, glob, re, collections
f = glob.glob('/tmp/pm_$TAG/*/*_kernel_trace.csv')[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = re.sub(r'repo::','',r['Kernel_Name'])
    n = re.sub(r'void igemm_kernel<','',n)[:70]
    if 'at::native' in n: continue
    d[(n, r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in d.items():
    v = sorted(v)
    print(f"[$TAG] {k[0]:72s} grid {k[1]:>7s}x{k[2]:>4s}x{k[3]:>3s} n={len(v):3d} med {v[len(v)//2]/1e3:8.1f} us  min {v[0]/1e3:8.1f}")
PY
