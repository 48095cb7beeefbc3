#!/bin/bash
# Kernel timeline of the pipelined update: rocprofv3 --kernel-trace of a short bench run, then for three steady
# updates the start / end (us) of the kernels that mark the phases of the two lanes.
# usage: tools/timeline.sh <tag> [bench args]   -> gpurun_out/<tag>_timeline.txt
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$TAG -- python3 $R/bench.py --no-cpu-baseline --steps 12 --warmup 4 "$@" > /tmp/tl_$TAG.log 2>&1
python3 - <<PY > $R/gpurun_out/${TAG}_timeline.txt
import csv, glob, os, re
csv.field_size_limit(1 << 30)
f = glob.glob('/tmp/tl_$TAG/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = re.sub(r'repo::|void ', '', n)
    n = re.sub(r'\(.*', '', n)
    return n[:70]
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Queue_Id', '?'), r.get('Stream_Id', '?')) for r in rows]
# steady state: updates are delimited by observe_fwd launches
obs = [i for i, e in enumerate(ev) if e[2].startswith('observe_fwd')]
i0, i1 = obs[8], obs[11]
t0 = ev[i0][0]
print(f"# three consecutive updates (observe_fwd #8..#11): period {(ev[i1][0]-t0)/3e3:.1f} us per update")
print(f"# {'start':>9s} {'end':>9s} {'dur':>8s}  queue stream  kernel")
busy = {}
for s, e, n, q, st in ev[i0:i1 + 40]:
    key = (q, st)
    busy[key] = busy.get(key, 0) + (e - s)
    if os.environ.get('TIMELINE_ALL') or any(k in n for k in ('observe', 'imagine', 'clip_adam', 'dconv_dec4_nll', 'uconv_scatter', 'dconv_wgrad', 'dconv_down', 'tanh_normal', 'lambda', 'kl_kernel', 'index', 'Index')) :
        print(f"{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} {(e-s)/1e3:8.1f}  {q:>5s} {st:>6s}  {n}")
print("# busy time per (queue, stream) over the window, us:", {k: round(v / 1e3) for k, v in busy.items()})
PY
head -3 $R/gpurun_out/${TAG}_timeline.txt
