#!/bin/bash
# kernel trace of the world-model lane alone (tools/lane_time.py wm loop) -> gpurun_out/<tag>_trace.csv
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pw_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/pw_$TAG -- python3 $R/tools/lane_time.py wmonly > /tmp/pw_$TAG.log 2>&1
tail -2 /tmp/pw_$TAG.log
python3 - <<PY
import csv, glob, re
f = glob.glob('/tmp/pw_$TAG/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
with open('$R/gpurun_out/${TAG}_trace.csv', 'w') as o:
    for r in rows:
        n = re.sub(r'repo::|void ', '', r['Kernel_Name'])[:90].replace(',', ';')
        wgs = (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
        o.write(f"{n},{int(r['Start_Timestamp'])-t0},{int(r['End_Timestamp'])-t0},{wgs},{r['Stream_Id'] if 'Stream_Id' in r else r.get('Queue_Id','')}\n")
PY
