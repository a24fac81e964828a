cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktr -o k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-through-trainer > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/ktr/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'][:50] for r in rows]
# find a window in the middle of the run around a copyBuffer cluster
idx = [i for i, n in enumerate(names) if 'copyBuffer' in n]
print(len(rows), 'kernels;', len(idx), 'copyBuffer')
mid = idx[len(idx)//2]
lo = mid
while lo > 0 and mid - lo < 60: lo -= 1
t0 = int(rows[lo]['Start_Timestamp'])
for r in rows[lo:mid + 40]:
    print('%9.1f us  %7.1f us  q%s  %s' % ((int(r['Start_Timestamp']) - t0)/1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:70]))
PY
