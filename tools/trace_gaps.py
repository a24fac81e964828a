"""Idle time between consecutive dispatches of a rocprofv3 --kernel-trace CSV (last third of the run):
python3 tools/trace_gaps.py DIR  -> busy / span, gap histogram, the kernels that follow the longest gaps."""
import csv
import glob
import sys
import collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows)*2//3:]
span = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp']))/1e3
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)/1e3
gaps = []
end = int(rows[0]['End_Timestamp'])
for r in rows[1:]:
    s = int(r['Start_Timestamp'])
    gaps.append(((s - end)/1e3, r['Kernel_Name'][:60]))
    end = max(end, int(r['End_Timestamp']))
print(f'launches {len(rows)}  span {span/1e3:.3f} ms  busy {busy/1e3:.3f} ms  idle {sum(max(g, 0) for g, _ in gaps)/1e3:.3f} ms')
h = collections.Counter()
for g, _ in gaps:
    h[min(int(max(g, 0)), 20)] += 1
print('gap histogram (us: count):', sorted(h.items()))
by = collections.defaultdict(list)
for g, k in gaps:
    by[k].append(g)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f'{sum(v)/1e3:8.3f} ms idle before {len(v):5d} x {k}  (avg {sum(v)/len(v):.1f} us)')
