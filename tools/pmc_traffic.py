"""HBM traffic per kernel launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

On the GPU box (separate passes: the two counters do not fit one TCC pass; --pmc only with
--kernel-trace, MI355X_MICROARCH.md "rocprofv3 PMC slots"):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python3 tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w gpurun_out/r02_pmc_hbm_traffic.json

Corrections (guide, HBM section): FETCH_SIZE is reported in KB and counts 128-byte read requests
as 64 bytes on gfx950 -> x2; WRITE_SIZE (KB) is exact for 16-byte-per-lane streaming stores.
Values are per-launch averages over all launches of a kernel in the run."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def read_counter(directory, counter):
    sums, counts, peaks = defaultdict(float), defaultdict(int), defaultdict(float)
    files = glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        raise SystemExit(f'no counter_collection.csv under {directory}')
    for path in files:
        with open(path, newline='') as f:
            for row in csv.DictReader(f):
                if row.get('Counter_Name') != counter:
                    continue
                name = row['Kernel_Name']
                value = float(row['Counter_Value'])
                sums[name] += value
                counts[name] += 1
                peaks[name] = max(peaks[name], value)
    return {k: (sums[k]/counts[k], counts[k], peaks[k]) for k in sums}


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fetch = read_counter(fetch_dir, 'FETCH_SIZE')
    write = read_counter(write_dir, 'WRITE_SIZE')
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        f_kb, n, f_max = fetch.get(name, (0.0, 0, 0.0))
        w_kb, n2, w_max = write.get(name, (0.0, 0, 0.0))
        rd = 2.0*f_kb*1024/1e6
        wr = w_kb*1024/1e6
        # one kernel template can serve launches of very different sizes (the grouped
        # weight gradient of 24 blocks and a single layer's): the largest launch is kept too
        kernels[name] = {'launches_sampled': max(n, n2), 'FETCH_SIZE_KB_raw': f_kb,
                         'WRITE_SIZE_KB': w_kb, 'hbm_read_MB_corrected': rd,
                         'hbm_write_MB': wr, 'hbm_traffic_MB': rd + wr,
                         'hbm_traffic_MB_largest_launch': (2.0*f_max + w_max)*1024/1e6}
    note = ('rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py '
            '--steps 3 --warmup 1; per-launch averages; FETCH_SIZE doubled (gfx950: 128-B requests '
            'tallied at 64 B, MI355X_MICROARCH.md HBM section); KB = 1024 B')
    step_mb = sum(v['hbm_traffic_MB']*v['launches_sampled'] for v in kernels.values())
    with open(out, 'w') as f:
        json.dump({'note': note, 'total_MB_all_launches': step_mb, 'kernels': kernels}, f, indent=1)
    print(f'{len(kernels)} kernels -> {out}')


if __name__ == '__main__':
    main()
