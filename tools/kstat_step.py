import csv,glob,os,sys
path=sys.argv[1]
path=glob.glob(os.path.join(path,'**','*kernel_stats.csv'),recursive=True)[0]
rows=list(csv.DictReader(open(path)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:45]:
    print(f"{float(r['TotalDurationNs'])/4e3:8.1f} us/step calls/step {int(r['Calls'])/4:6.1f} avg {float(r['AverageNs'])/1e3:8.1f}  {r['Name'][:110]}")
