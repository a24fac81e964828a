"""Longest dispatches of a kernel in a rocprofv3 --kernel-trace CSV: python3 tools/trace_top.py DIR SUBSTR"""
import csv
import glob
import sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if sys.argv[2] in r['Kernel_Name']]
rows = rows[len(rows)*3//4:]
out = []
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e3
    i = r['Kernel_Name'].find('<')
    out.append((d, r['Kernel_Name'][i:i + 22], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z']))
for o in sorted(out, reverse=True)[:28]:
    print('%9.1f us %s grid %s %s %s' % o)
print('sum ms', sum(o[0] for o in out)/1e3, 'launches', len(out))
