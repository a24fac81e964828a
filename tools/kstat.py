"""Rows of a rocprofv3 kernel_stats.csv whose kernel name contains any of the given substrings:
    python3 tools/kstat.py <stats.csv | directory> [substring ...]   (no substring: the top 15)"""
import csv
import glob
import os
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = glob.glob(os.path.join(path, '**', '*kernel_stats.csv'), recursive=True)[0]
rows = list(csv.DictReader(open(path)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
keys = sys.argv[2:]
sel = [r for r in rows if any(k in r['Name'] for k in keys)] if keys else \
    sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:15]
print(f'total kernel time {tot/1e6:.2f} ms')
for r in sel:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}% calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:90]}")
