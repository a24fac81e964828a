"""`roofline` objects of a widened row from its rocprofv3 kernel statistics and the PMC traffic file of
tools/pmc_traffic.py (tools/profile_rows.sh): for the row's dominant kernel and every kernel >= 5 % of the
kernel time -- average launch duration, measured HBM traffic per launch (FETCH_SIZE x 2 + WRITE_SIZE, the
guide's gfx950 correction), achieved = traffic / duration against the 8 TB/s peak. The traffic is MEASURED
(these kernels serve many shapes in one run; an algorithmic byte count per launch would have to be an
average over them), which is said in the object.

    python tools/rows_roofline.py <row name> <kernel_stats.csv> <pmc_hbm_traffic.json>
"""
import csv
import json
import sys


def short(kernel):
    """`void (anonymous namespace)::conv_kernel<4, true>((anonymous namespace)::P)` -> `conv_kernel<4, true>`"""
    k = kernel.replace('(anonymous namespace)::', '').replace('brv::', '')
    if k.startswith('void '):
        k = k[5:]
    depth = 0
    for i, ch in enumerate(k):          # cut at the argument list: the first '(' outside the template brackets
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0:
            return k[:i]
    return k


def main():
    name, stats_path, pmc_path = sys.argv[1:4]
    rows = list(csv.DictReader(open(stats_path)))
    total = sum(float(r['TotalDurationNs']) for r in rows)
    pmc = json.load(open(pmc_path))
    kernels = pmc.get('kernels', pmc)

    def traffic(kernel):
        k = kernels.get(kernel)
        if k is None:      # the two files may print the argument list differently: match by the name
            for cand, v in kernels.items():
                if short(cand) == short(kernel):
                    k = v
                    break
        return None if k is None else k['hbm_traffic_MB']

    out = []
    for r in rows:
        share = float(r['TotalDurationNs'])/total
        if share < 0.05 and out:
            continue
        avg_us = float(r['AverageNs'])/1e3
        mb = traffic(r['Name'])
        gbs = None if mb is None else mb/avg_us*1e3     # MB / us = TB/s -> GB/s
        out.append({'kernel': short(r['Name'])[:120], 'calls': int(r['Calls']), 'avg_launch_us': avg_us,
                    'share_of_kernel_time': share, 'traffic': None if mb is None else mb*1e6,
                    'achieved': gbs, 'frac': None if gbs is None else gbs/8000.0})
    top = out[0]
    line = {'row': name, 'roofline': {'kernel': top['kernel'], 'bound': 'hbm', 'achieved': top['achieved'],
                                      'peak': 8000.0, 'unit': 'GB/s', 'frac': top['frac'],
                                      'traffic': top['traffic'], 'avg_launch_us': top['avg_launch_us'],
                                      'share_of_kernel_time': top['share_of_kernel_time'],
                                      'basis': 'measured: PMC bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, per-launch '
                                               'average over all shapes the kernel served) / average launch duration of '
                                               'rocprofv3 --kernel-trace --stats'},
            'kernels_over_5pct': out, 'kernel_time_ms_total': total/1e6}
    print(json.dumps(line))


if __name__ == '__main__':
    main()
