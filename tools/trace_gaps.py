import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# take the last 60% of rows (steady state)
n = len(rows)
rows = rows[int(n*0.5):]
gaps = collections.defaultdict(list); durs = collections.defaultdict(list)
prev_end = None
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'][:60]
    durs[name].append(e - s)
    if prev_end is not None:
        gaps[name].append(s - prev_end)
    prev_end = e
tot_d = sum(sum(v) for v in durs.values()); tot_g = sum(sum(v) for v in gaps.values())
print('kernels', len(rows), 'sum dur %.2f ms, sum gaps %.2f ms' % (tot_d/1e6, tot_g/1e6))
for name, v in sorted(durs.items(), key=lambda kv: -sum(kv[1]))[:25]:
    g = gaps.get(name, [0])
    print('%-60s n=%5d dur avg %7.1f us  gap-before avg %6.1f us' % (name, len(v), sum(v)/len(v)/1e3, sum(g)/len(g)/1e3))
