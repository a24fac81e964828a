"""Compact view of tools/bench_rows.py JSON lines (stdin)."""
import json
import sys

for line in sys.stdin:
    try:
        d = json.loads(line)
    except Exception:
        continue
    out = [d.get('row', '?')]
    for k in ('value', 'utt_per_s', 'items_per_s', 'ms_per_step', 'equiv_4s_utterances_per_s', 'peak_memory_GB',
              's_per_utt_batch1'):
        if k in d:
            out.append(f'{k}={d[k]:.4g}')
    if 'roofline' in d and d['roofline']:
        out.append(f"frac={d['roofline']['frac']:.3f}")
    if d.get('resident_in_hbm'):
        out.append('resident: ' + ' '.join(f'{k}={v:.4g}' for k, v in d['resident_in_hbm'].items()))
    print(' | '.join(out))
