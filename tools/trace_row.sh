#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/trace_row.sh dccrn amp > gpurun_out/trace_dccrn.txt
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_row
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_row -o t -- python3 $REPO/tools/trace_row.py $1 ${2:-amp} ${3:-3} > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/trace_row/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the last step = the launches behind the second-to-last optimizer launch
adam = [i for i, r in enumerate(rows) if 'clip_adam' in r['Kernel_Name']]
cut = adam[-2] + 1
t0 = int(rows[cut]['Start_Timestamp'])
for r in rows[cut:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    g = f"{r.get('Grid_Size_X', r.get('Grid_Size', '?'))}x{r.get('Grid_Size_Y', '')}x{r.get('Grid_Size_Z', '')}"
    print(f"{(s - t0)/1e3:9.1f} {(e - s)/1e3:7.1f} q{r.get('Queue_Id', '?'):>2} {g:>16} {r['Kernel_Name'][:110]}")
PY
