#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/trace_step.sh > gpurun_out/trace_step.txt
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_step -o t -- python3 $REPO/tools/trace_step.py 4 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/trace_step/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# last step: from the last prep_weights launch on
idx = max(i for i, r in enumerate(rows) if 'prep_weights' in r['Kernel_Name'])
idx0 = max(i for i, r in enumerate(rows[:idx]) if 'prep_weights' in r['Kernel_Name'])
t0 = int(rows[idx0]['Start_Timestamp'])
for r in rows[idx0:idx]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0)/1e3:9.1f} {(e - s)/1e3:7.1f} q{r.get('Queue_Id', '?'):>3} {r['Kernel_Name'][:80]}")
PY
