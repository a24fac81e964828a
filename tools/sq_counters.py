"""Per-kernel SQ counter summary from rocprofv3 --pmc passes (tools/profile_sq.sh).

    python3 tools/sq_counters.py <out.json> <pass dir> [<pass dir> ...]

Every pass directory holds one `*counter_collection.csv` (one row per dispatch and counter) taken
with `--pmc <up to 8 SQ counters> [GRBM_GUI_ACTIVE] --kernel-trace`. Counters are averaged per
launch over all launches of a kernel name; derived fractions follow MI355X_MICROARCH.md ("rocprofv3
PMC slots", cycle-constants row "s_memtime tick vs SQ PMC units"):

  * SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over all waves;
    WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, so the shares below are fractions of a
    wave's lifetime (wait = parked at s_waitcnt / barrier, issue-stall, issuing);
  * SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16_bf16) summed over the SIMDs:
    mfma_busy_frac = that / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the
    counter is summed over the 8 XCDs);
  * SQ_BUSY_CYCLES-free forms only: nothing here depends on the gfx94x derived-metric formulas.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

N_SIMD = 1024          # 256 CUs x 4
N_CU = 256


def read_pass(directory):
    files = glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        raise SystemExit(f'no counter_collection.csv under {directory}')
    sums = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(lambda: defaultdict(set))
    for path in files:
        with open(path, newline='') as f:
            for row in csv.DictReader(f):
                name, c = row['Kernel_Name'], row['Counter_Name']
                sums[name][c] += float(row['Counter_Value'])
                disp[name][c].add(row.get('Dispatch_Id', len(disp[name][c])))
    return {k: {c: sums[k][c]/max(1, len(disp[k][c])) for c in sums[k]} for k in sums}, \
           {k: max(len(s) for s in disp[k].values()) for k in disp}


def main():
    out = sys.argv[1]
    per = defaultdict(dict)
    launches = {}
    for d in sys.argv[2:]:
        vals, n = read_pass(d)
        for k, cs in vals.items():
            for c, v in cs.items():
                per[k].setdefault(c, v)       # (a counter repeated in a later pass keeps the first value)
            launches[k] = max(launches.get(k, 0), n[k])
    res = {}
    for k, c in per.items():
        r = {'launches_sampled': launches[k], 'counters_per_launch': c}
        wc = c.get('SQ_WAVE_CYCLES', 0.0)
        gui = c.get('GRBM_GUI_ACTIVE', 0.0)
        cyc = gui/8.0
        if wc > 0:
            for src, dst in (('SQ_WAIT_ANY', 'wait_any_share'), ('SQ_WAIT_INST_ANY', 'issue_stall_share'),
                             ('SQ_ACTIVE_INST_ANY', 'issuing_share'), ('SQ_ACTIVE_INST_VALU', 'valu_share'),
                             ('SQ_ACTIVE_INST_LDS', 'lds_share'), ('SQ_ACTIVE_INST_VMEM', 'vmem_share'),
                             ('SQ_ACTIVE_INST_SCA', 'scalar_share'), ('SQ_WAIT_INST_LDS', 'lds_issue_stall_share')):
                if src in c:
                    r[dst] = c[src]/wc
        if cyc > 0:
            r['kernel_cycles'] = cyc
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
                r['mfma_busy_frac'] = c['SQ_VALU_MFMA_BUSY_CYCLES']/(N_SIMD*cyc)
            if 'SQ_ACTIVE_INST_VALU' in c:      # quad-cycles of VALU issue per SIMD and kernel cycle
                r['valu_busy_frac'] = 4.0*c['SQ_ACTIVE_INST_VALU']/(N_SIMD*cyc)
            if 'SQ_LDS_IDX_ACTIVE' in c:
                r['lds_array_busy_frac'] = c['SQ_LDS_IDX_ACTIVE']/(N_CU*cyc)
            if wc > 0:
                r['waves_resident_per_simd'] = 4.0*wc/(N_SIMD*cyc)
        if c.get('SQ_LDS_IDX_ACTIVE', 0) > 0 and 'SQ_LDS_BANK_CONFLICT' in c:
            r['lds_bank_conflict_frac'] = c['SQ_LDS_BANK_CONFLICT']/c['SQ_LDS_IDX_ACTIVE']
        if c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) > 0 and 'SQ_VALU_MFMA_COEXEC_CYCLES' in c:
            r['valu_mfma_coexec_over_mfma_busy'] = c['SQ_VALU_MFMA_COEXEC_CYCLES']/c['SQ_VALU_MFMA_BUSY_CYCLES']
        res[k] = r
    note = ('rocprofv3 --pmc <SQ counters> --kernel-trace, separate passes over bench.py --steps 3 --warmup 1 '
            '(one-chain step, BRV_CTN_STREAMS=1); per-launch averages. Shares are fractions of SQ_WAVE_CYCLES '
            '(wave lifetime); *_busy_frac are fractions of the kernel cycles (GRBM_GUI_ACTIVE / 8) times the '
            'number of SIMDs (1024) or CUs (256). Profiled passes run at a lower clock than plain runs.')
    with open(out, 'w') as f:
        json.dump({'note': note, 'kernels': res}, f, indent=1, sort_keys=True)
    top = sorted(res.items(), key=lambda kv: -kv[1].get('kernel_cycles', 0)*kv[1]['launches_sampled'])[:12]
    for k, r in top:
        print(f"{k[:60]:60s} n={r['launches_sampled']:4d} cyc={r.get('kernel_cycles', 0):9.0f} "
              f"mfma={r.get('mfma_busy_frac', float('nan')):.3f} valu={r.get('valu_busy_frac', float('nan')):.3f} "
              f"lds={r.get('lds_array_busy_frac', float('nan')):.3f} wait={r.get('wait_any_share', float('nan')):.2f} "
              f"stall={r.get('issue_stall_share', float('nan')):.2f} waves/simd={r.get('waves_resident_per_simd', float('nan')):.2f}")


if __name__ == '__main__':
    main()
