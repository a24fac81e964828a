"""Idle gaps of the device in the last step of a kernel trace listing (tools/trace_row.sh output: start us, duration us, queue,
grid, kernel): wall time, time with at least one kernel running, and the largest gaps with the kernel that ended them.
    python tools/trace_gaps.py gpurun_out/trace_dccrn.txt"""
import sys
rows = []
for line in open(sys.argv[1]):
    p = line.split()
    try:
        s, d = float(p[0]), float(p[1])
    except (ValueError, IndexError):
        continue
    rows.append((s, s + d, ' '.join(p[4:])[:90]))
rows.sort()
end = busy = 0.0
gaps = []
for s, e, name in rows:
    if s > end:
        if end > 0:
            gaps.append((s - end, end, name))
        busy += e - s
    elif e > end:
        busy += e - end
    end = max(end, e)
wall = rows[-1][1] - rows[0][0]
print(f'{len(rows)} launches, wall {wall:.0f} us, busy {busy:.0f} us, idle {wall - busy:.0f} us in {len(gaps)} gaps')
for g in sorted(gaps, reverse=True)[:int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
    print(f'  {g[0]:8.1f} us idle at {g[1]:9.1f}, then {g[2]}')
