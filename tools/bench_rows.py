"""Throughput of the widened SURVEY.md section-8 rows on one GPU (BASELINE.json configs[0],
[3], [4] and the section-8f models TF-GridNet / SGMSE+ training); bench.py stays the Conv-TasNet
headline. One JSON line per row.

    python tools/bench_rows.py [--rows ffnn,dccrn,tfgridnet,sgmse,sgmse_train]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, warmup, steps):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0)/steps


def train_row(arch, batch, seconds, steps, use_amp):
    from brever_amd.models import ModelRegistry
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = ModelRegistry.get(arch)().to(dev)
    model.train()
    L = int(seconds*16000)
    wav = 0.1*torch.randn(batch, 2, 2, L, device=dev)         # (B, sources, channels, L)
    items = [model.transform(w) for w in wav]
    x = torch.stack(items) if not isinstance(items[0], (tuple, list)) else None
    if x is None:
        x = tuple(torch.stack([it[i] for it in items]) for i in range(len(items[0])))
    lengths = torch.full((batch,), (x[0] if isinstance(x, tuple) else x).shape[-1], device=dev)
    scaler = torch.amp.GradScaler('cuda', enabled=False)
    dt = timed(lambda: model.train_step(x, lengths, use_amp, scaler), 3, steps)
    row = {'row': f'{arch} train', 'utt_per_s': batch/dt, 'ms_per_step': dt*1e3, 'batch': batch,
           'seconds': seconds}
    if arch == 'dccrn' and seconds == 4.0:
        # 51.2 GFLOP per 4 s utterance forward, x3 for a training step
        row['tflops'] = batch/dt*3*51.2e9/1e12
    return row


def sgmse_row(seconds, steps):
    from brever_amd.models import ModelRegistry
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = ModelRegistry.get('sgmsep')(solver_num_steps=steps).to(dev).eval()
    out = {}
    for amp, batch in ((True, 1), (True, 8), (False, 1)):
        wav = 0.1*torch.randn(batch, 2, int(seconds*16000), device=dev)
        model.enhance(wav, use_amp=amp)                  # captures the HIP graph of this shape
        dt = timed(lambda: model.enhance(wav, use_amp=amp), 0, 1)
        out[f"{'fp16_mfma' if amp else 'fp32'}_b{batch}"] = {
            's_per_utt': dt/batch, 'utt_per_s': batch/dt, 'ms_per_nfe': dt/(2*steps)*1e3,
            'rtf': dt/batch/seconds,
            # 1.04 TFLOP per network evaluation and 4 s utterance
            'tflops': batch*2*steps*1.04*(seconds/4.0)/dt}
    return {'row': f'sgmsep enhance, {steps}-step PC sampler ({2*steps} network evaluations)',
            'seconds': seconds, **out}


def sgmse_train_row(batch, frames, steps, use_amp=False):
    from brever_amd.models import ModelRegistry
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = ModelRegistry.get('sgmsep')().to(dev).train()
    x = 0.3*torch.randn(batch, 2, 256, frames, dtype=torch.complex64, device=dev)
    lengths = torch.full((batch,), frames, device=dev)
    scaler = torch.amp.GradScaler('cuda', enabled=False)
    dt = timed(lambda: model.train_step(x, lengths, use_amp, scaler), 1, steps)
    return {'row': f"sgmsep train ({'bf16 convolutions' if use_amp else 'fp32'}, default 65.6 M-param network)", 'batch': batch,
            'frames': frames, 'seconds_per_item': (frames - 1)*128/16000, 'ms_per_step': dt*1e3,
            'items_per_s': batch/dt}


PEAK = {'bf16': 2500.0, 'fp16': 2500.0, 'fp32': 157.3}     # dense MFMA TFLOP/s (MI355X guide)


def driver_line(row, metric, value, unit, dtype, workload, tflops=None, higher=True):
    """The fields of bench.py's JSON line for a widened row, with a whole-workload MFMA roofline
    when the algorithmic FLOPs are known."""
    row.update({'metric': metric, 'value': value, 'unit': unit, 'n_gpus': 1, 'dtype': dtype,
                'higher_is_better': higher, 'data': 'synthetic', 'vs_baseline': None,
                'config': {'workload': workload}})
    if tflops is not None:
        row['roofline'] = {'bound': 'mfma', 'achieved': tflops, 'peak': PEAK[dtype],
                           'unit': 'TFLOP/s', 'frac': tflops/PEAK[dtype], 'traffic': None,
                           'scope': 'whole workload: algorithmic FLOPs of SURVEY.md 8(d) over the '
                                    'measured wall time (per-kernel tables: profiles/r02_rows_*.csv)'}
    return row


def convtasnet_fp32_row(steps=5):
    """The Conv-TasNet benchmark step on the fp32 path (use_amp=False, brv_ctn_f32_*)."""
    from brever_amd.models import ConvTasNet
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = ConvTasNet().to(dev)
    batch = 0.1*torch.randn(16, 2, 64000, device=dev)
    lengths = torch.full((16,), 64000, device=dev)
    dt = timed(lambda: model.train_step(batch, lengths, False, None), 2, steps)
    row = {'row': 'convtasnet train, fp32 path (use_amp=False)', 'ms_per_step': dt*1e3}
    return driver_line(row, 'utterances/sec (4 s @16 kHz) Conv-TasNet train, fp32 activations',
                       16/dt, 'utterances/s', 'fp32',
                       'Conv-TasNet defaults, 16 x 4 s, fwd + SNR loss + bwd + clip + Adam, '
                       'exact-fp32 MFMA products (brv_ctn_f32_*)', tflops=16/dt*116.46e9/1e12)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', default='convtasnet_fp32,ffnn,dccrn,tfgridnet,sgmse,sgmse_train')
    args = ap.parse_args()
    rows = args.rows.split(',')
    if len(rows) > 1:
        # one process per row: a row measured after others in the same process inherits their
        # caching-allocator state (the SGMSE+ training row read 114 instead of 74 ms that way)
        import subprocess
        import sys
        for r in rows:
            subprocess.run([sys.executable, __file__, '--rows', r], check=False)
        return
    if 'convtasnet_fp32' in rows:
        print(json.dumps(convtasnet_fp32_row()), flush=True)
    if 'ffnn' in rows:
        print(json.dumps(train_row('ffnn', 32, 2.0, 20, False)), flush=True)
    if 'dccrn' in rows:
        for amp in (False, True):
            row = train_row('dccrn', 16, 4.0, 5, amp)
            if amp:
                row['row'] += ' (use_amp: bf16 matrix products)'
            driver_line(row, 'utterances/sec (4 s @16 kHz) DCCRN train (BASELINE config 3)',
                        row['utt_per_s'], 'utterances/s', 'bf16' if amp else 'fp32',
                        'DCCRN defaults (3 671 053 params), 16 x 4 s, STFT 512/128 -> complex '
                        'Conv2d + LSTM -> iSTFT, fwd + SNR loss + bwd + clip 5.0 + Adam',
                        tflops=row['tflops'])
            print(json.dumps(row), flush=True)
    if 'tfgridnet' in rows:
        print(json.dumps(train_row('tfgridnet', 4, 4.0, 3, False)), flush=True)
        row = train_row('tfgridnet', 4, 4.0, 3, True)
        row['row'] += ' (use_amp: bf16 matrix products)'
        print(json.dumps(row), flush=True)
    if 'sgmse_train' in rows:
        print(json.dumps(sgmse_train_row(4, 128, 3)), flush=True)
        print(json.dumps(sgmse_train_row(4, 128, 3, True)), flush=True)
    if 'sgmse' in rows:
        row = sgmse_row(4.0, 30)
        b8 = row['fp16_mfma_b8']
        driver_line(row, 'utterances/sec SGMSE+ enhance (BASELINE config 4)', b8['utt_per_s'],
                    'utterances/s', 'fp16',
                    'SGMSE+ defaults (65.6 M params), 30-step reverse SDE (PC sampler, 60 network '
                    'evaluations), 8 x 4 s utterances, fp16-MFMA convolutions, iSTFT overlap-add',
                    tflops=b8['tflops'])
        row['s_per_utt_batch1'] = row['fp16_mfma_b1']['s_per_utt']
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
