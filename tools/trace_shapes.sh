#!/bin/bash
# Per (kernel, grid) totals of a program's kernel trace: which SHAPES of a kernel the time goes to.
#   bash tools/trace_shapes.sh tools/prof_sgmse.py 8 > gpurun_out/shapes_sgmse_b8.txt     (on the GPU box, from the repo root)
REPO=$(pwd)
PROG=$REPO/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_shapes
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_shapes -o t -- python3 $PROG "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/trace_shapes/**/*kernel_trace.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for r in csv.DictReader(open(f)):
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e3
    g = f"{r.get('Grid_Size_X', '?')}x{r.get('Grid_Size_Y', '')}x{r.get('Grid_Size_Z', '')}"
    k = (r['Kernel_Name'][:70], g)
    acc[k][0] += 1; acc[k][1] += d; tot += d
print(f'total kernel time {tot/1e3:.2f} ms')
for (name, g), (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f'{t/tot*100:5.1f}% {n:5d} x {t/n:8.1f} us  {g:>18}  {name}')
PY
