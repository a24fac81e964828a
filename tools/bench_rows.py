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
    if arch == 'tfgridnet' and seconds == 4.0:
        # 138.98 GFLOP per 4 s utterance forward (torch.utils.flop_counter over oracle/tfgridnet.py: the mm / bmm /
        # addmm / conv products, recurrences included), x3 for a training step
        row['tflops'] = batch/dt*3*138.98e9/1e12
    return row


def sgmse_row(seconds, steps):
    from brever_amd.models import ModelRegistry
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = ModelRegistry.get('sgmsep')(solver_num_steps=steps).to(dev).eval()
    out = {}
    for amp, batch in ((True, 1), (True, 8), (True, 32), (False, 1)):
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
    row = {'row': f"sgmsep train ({'bf16 convolutions' if use_amp else 'fp32'}, default 65.6 M-param network)", 'batch': batch,
           'frames': frames, 'seconds_per_item': (frames - 1)*128/16000, 'ms_per_step': dt*1e3,
           'items_per_s': batch/dt}
    # 1.04 TFLOP per network evaluation of a 501-frame spectrogram; a training step = 3 x the forward
    tflops = batch/dt*3*1.04*frames/501
    return driver_line(row, 'items/sec SGMSE+ score-network training (SURVEY 8f rank 4)', batch/dt, 'items/s',
                       'bf16' if use_amp else 'fp32',
                       f'SGMSE+ defaults (65.6 M params), {batch} x {frames} frames, denoising-score-matching loss + backward '
                       '+ Adam', tflops=tflops)


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
                                    'measured wall time (per-kernel tables of the same round: profiles/rNN_rows_<row>_kernel_stats.csv)'}
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


def convtasnet_bigbatch_row(budget_s, n_batches=5, seed=0):
    """The Conv-TasNet bf16 training step on HBM-sized DYNAMIC batches (north_star: "bucket batcher sized for
    288 GB HBM3E"; reference brever/batching.py:191-204,248-251): ragged synthetic mixtures of 0.5 - 4 s through
    BucketBatchSampler(dynamic=True, batch_size=budget_s seconds of padded audio) -> collate -> pinned async H2D
    -> train_step. budget_s <= 0: the automatic size of `trainer.batch_size: 0` (batching.hbm_batch_seconds)."""
    from brever_amd.batching import BucketBatchSampler
    from brever_amd.data import BreverDataLoader, DevicePrefetcher, SyntheticMixtureDataset
    from brever_amd.models import ConvTasNet
    from brever_amd.training import BreverTrainer
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = ConvTasNet().to(dev)
    auto = budget_s <= 0
    if auto:
        budget_s = BreverTrainer.auto_batch_seconds(model, 0, 16000)
    mean_len = 0.5*(0.5 + 4.0)
    n_items = int(budget_s/mean_len*(n_batches + 0.5))
    dset = SyntheticMixtureDataset(n_items, 64000, min_length=8000, transform=lambda s: s.mean(axis=-2), seed=seed)
    dset.preload('cpu')
    sampler = BucketBatchSampler(dset, batch_size=budget_s, dynamic=True, fs=16000, seed=seed)
    sampler.set_epoch(0)
    loader = BreverDataLoader(dataset=dset, batch_sampler=sampler, num_workers=0)
    torch.set_num_threads(4)
    scaler = torch.amp.GradScaler('cuda', enabled=False)
    torch.cuda.reset_peak_memory_stats()
    items = seconds = padded = 0.0
    sizes = []
    t0 = None
    losses = []
    for i, (batch, lengths) in enumerate(DevicePrefetcher(loader, dev)):
        if i == 1:                       # the first batch pays the allocations
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        loss = model.train_step(batch, lengths, True, scaler)
        if i >= 1:
            items += batch.shape[0]
            seconds += float(lengths.sum())/16000
            padded += batch.shape[0]*batch.shape[-1]/16000
        sizes.append(int(batch.shape[0]))
        losses.append(loss)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    losses = [float(x) for x in losses]
    assert all(x == x and abs(x) < 1e4 for x in losses), losses
    # the same batches resident in HBM (bench.py's `value` convention): what the device does per step
    sampler._previous_epoch = None       # the same epoch again (same batches)
    resident = [(b.to(dev), n.to(dev)) for b, n in loader][1:-1] or None
    res = None
    if resident:
        for b, n in resident[:1]:
            model.train_step(b, n, True, scaler)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for b, n in resident:
            model.train_step(b, n, True, scaler)
        torch.cuda.synchronize()
        dtr = time.perf_counter() - t1
        rs = sum(float(n.sum())/16000 for _, n in resident)
        res = {'ms_per_step': dtr/len(resident)*1e3, 'utterances_per_s': sum(b.shape[0] for b, _ in resident)/dtr,
               'audio_seconds_per_s': rs/dtr, 'equiv_4s_utterances_per_s': rs/dtr/4.0,
               'padded_equiv_4s_utterances_per_s': sum(b.shape[0]*b.shape[-1] for b, _ in resident)/16000/dtr/4.0}
    row = {'row': f"convtasnet train, dynamic bucket batches of {budget_s:.0f} s{' (automatic: HBM-sized)' if auto else ''}",
           'budget_seconds': budget_s, 'batches_timed': len(sizes) - 1, 'utterances_per_batch': sizes,
           'ms_per_step': dt/max(len(sizes) - 1, 1)*1e3, 'audio_seconds_per_s': seconds/dt,
           'padded_seconds_per_s': padded/dt, 'equiv_4s_utterances_per_s': seconds/dt/4.0,
           'peak_memory_GB': torch.cuda.max_memory_allocated()/1e9,
           'total_memory_GB': torch.cuda.get_device_properties(0).total_memory/1e9, 'losses': losses,
           'resident_in_hbm': res}
    return driver_line(row, 'utterances/sec Conv-TasNet train on ragged 0.5-4 s mixtures, dynamic bucket batches',
                       items/dt, 'utterances/s', 'bf16',
                       f'Conv-TasNet defaults, bucket batches of {budget_s:.0f} s of padded audio (ragged 0.5 - 4 s items), '
                       'fwd + SNR loss + bwd + clip + Adam through the trainer data path',
                       tflops=seconds/dt/4.0*116.46e9/1e12)


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
    for r in rows:
        if r.startswith('convtasnet_bigbatch'):          # convtasnet_bigbatch:<seconds> (0 = automatic)
            budget = float(r.split(':')[1]) if ':' in r else 0.0
            print(json.dumps(convtasnet_bigbatch_row(budget)), flush=True)
    if 'ffnn' in rows:
        print(json.dumps(train_row('ffnn', 32, 2.0, 20, False)), flush=True)
    if 'dccrn' in rows:
        for amp in (False, True):
            row = train_row('dccrn', 16, 4.0, 5, amp)
            if amp:
                row['row'] += ' (use_amp: bf16 matrix products)'
            driver_line(row, 'utterances/sec (4 s @16 kHz) DCCRN train (BASELINE.json configs[3], the 4th: DCCRN bf16 1 x MI355X)',
                        row['utt_per_s'], 'utterances/s', 'bf16' if amp else 'fp32',
                        'DCCRN defaults (3 671 053 params), 16 x 4 s, STFT 512/128 -> complex '
                        'Conv2d + LSTM -> iSTFT, fwd + SNR loss + bwd + clip 5.0 + Adam',
                        tflops=row['tflops'])
            print(json.dumps(row), flush=True)
    if 'tfgridnet' in rows:
        for amp in (False, True):
            row = train_row('tfgridnet', 4, 4.0, 3, amp)
            if amp:
                row['row'] += ' (use_amp: bf16 matrix products)'
            driver_line(row, 'utterances/sec (4 s @16 kHz) TF-GridNet train (SURVEY 8f rank 4)', row['utt_per_s'],
                        'utterances/s', 'bf16' if amp else 'fp32',
                        'TF-GridNet defaults (3 735 344 params), 4 x 4 s, STFT 256/128 -> 6 grid blocks (BLSTMs, attention) '
                        '-> iSTFT, multiresyu loss + bwd + clip + Adam', tflops=row['tflops'])
            print(json.dumps(row), flush=True)
    if 'sgmse_train' in rows:
        print(json.dumps(sgmse_train_row(4, 128, 3)), flush=True)
        print(json.dumps(sgmse_train_row(4, 128, 3, True)), flush=True)
    if 'sgmse' in rows:
        row = sgmse_row(4.0, 30)
        b8 = row['fp16_mfma_b8']
        driver_line(row, 'utterances/sec SGMSE+ enhance (BASELINE.json configs[4], the 5th: SGMSE+ fp16 inference)', b8['utt_per_s'],
                    'utterances/s', 'fp16',
                    'SGMSE+ defaults (65.6 M params), 30-step reverse SDE (PC sampler, 60 network '
                    'evaluations), 8 x 4 s utterances, fp16-MFMA convolutions, iSTFT overlap-add',
                    tflops=b8['tflops'])
        row['s_per_utt_batch1'] = row['fp16_mfma_b1']['s_per_utt']
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
