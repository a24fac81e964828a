"""Headline benchmark: Conv-TasNet training throughput on synthetic 4 s @ 16 kHz
mixtures (BASELINE.json configs[1]; weak scaling over N GPUs of one node).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one optimizer step of the reference's hot path on one batch of 16
utterances per GPU, inputs already resident in HBM: ConvTasNet.train_step =
forward -> SNR loss -> backward in parts -> [bucketed gradient all-reduce over RCCL,
overlapped with backward] -> clip(5.0) + Adam (brever/models/base.py:178-210,
convtasnet.py:78-89). Rank 0 prints ONE JSON line. Beside `value` (resident inputs) the
line carries `through_trainer`: the same step fed by the trainer's own data path (bucket
sampler -> BreverDataLoader collate -> pinned double-buffered async H2D), `roofline` for the
kernel with the largest share plus `kernels` (every launch label >= 5 % of the step with its
algorithmic and measured HBM bytes), `allreduce_exposed_ms` for N > 1, and `cpu_baseline`
(N = 1 only): the CPU oracle (oracle/) on the same configuration, B = 16, fp32 and CPU-bf16.

Measurement hygiene: whatever --warmup says, the warm-up runs for at least MIN_WARMUP_S
seconds of steps (`warmup_effective`: the clocks and the power state have settled when the timed region starts);
`value` is the --steps block as before; `value_repeats` are five further blocks of --steps (min / median /
max); `clock` holds sclk / mclk / power / temperature read from sysfs before and after the timed region.
`other_configs` (N = 1 only, --no-other-configs to skip): BASELINE.json configs[3] (DCCRN bf16 train step,
16 x 4 s) and configs[4] (SGMSE+ fp16 30-step enhance, batch 1 and 8) timed in the same run.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC for RCCL (before HIP initialises)
# the step runs two kernel chains on two streams; with RCCL's streams the default 4 hardware queues
# are oversubscribed and the chains share one (9.6 instead of 7.7 ms per step)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import torch                                                  # noqa: E402
import torch.distributed as dist                              # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from brever_amd import hip                                    # noqa: E402
from brever_amd.batching import BucketBatchSampler                # noqa: E402
from brever_amd.data import (BreverDataLoader, DevicePrefetcher,   # noqa: E402
                             SyntheticMixtureDataset)
from brever_amd.models import ConvTasNet                      # noqa: E402
from brever_amd.parallel import GradSynchronizer, broadcast_parameters  # noqa: E402

FS = 16000
SECONDS = 4.0
BATCH = 16
FLOP_PER_UTT_TRAIN = 116.46e9       # 3 x 38.82 GFLOP forward
PEAK_HBM_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PEAK_MFMA_TFLOPS = 2500.0           # dense bf16 MFMA
MIN_WARMUP_S = 1.5                  # seconds of steps before the timed region, whatever --warmup says
N_REPEATS = 5                       # further blocks of --steps behind the one `value` is taken from


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _sysfs_device(index):
    """sysfs directory of the amdgpu device behind cuda:<index> (matched by PCI address), or None."""
    import glob
    want = None
    try:
        pr = torch.cuda.get_device_properties(index)
        want = f'{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}'
    except Exception:           # noqa: BLE001  (older torch: no PCI fields)
        pass
    cards = [d for d in sorted(glob.glob('/sys/class/drm/card*/device')) if os.path.exists(d + '/pp_dpm_sclk')]
    if want is not None:
        for d in cards:
            if os.path.basename(os.path.realpath(d)).startswith(want):
                return d
    return cards[index] if len(cards) > index else (cards[0] if cards else None)


def read_clock(index):
    """Clock / power / temperature of the device right now, from sysfs (no subprocess: microseconds, so it can sit
    directly in front of and behind the timed region): the active level of pp_dpm_sclk / pp_dpm_mclk, hwmon's
    instantaneous sclk (freq1_input), board power and its cap, edge / junction / memory temperatures. Fields the
    box does not expose are left out; {} when there is no amdgpu sysfs node at all."""
    import glob
    d = _sysfs_device(index)
    if d is None:
        return {}
    out = {'sysfs': os.path.basename(os.path.realpath(d))}
    for key, name in (('sclk_mhz', 'pp_dpm_sclk'), ('mclk_mhz', 'pp_dpm_mclk')):
        txt = _read(f'{d}/{name}')
        if txt:
            levels = [ln for ln in txt.splitlines() if ln.strip()]
            act = [ln for ln in levels if ln.rstrip().endswith('*')]
            try:
                out[key] = float((act or levels[-1:])[0].split(':')[1].lower().replace('mhz', '').replace('*', ''))
                out[key + '_max'] = max(float(ln.split(':')[1].lower().replace('mhz', '').replace('*', ''))
                                        for ln in levels)
            except (IndexError, ValueError):
                pass
    for hw in glob.glob(f'{d}/hwmon/hwmon*'):
        for key, name, scale in (('sclk_now_mhz', 'freq1_input', 1e-6), ('power_w', 'power1_average', 1e-6),
                                 ('power_w', 'power1_input', 1e-6), ('power_cap_w', 'power1_cap', 1e-6),
                                 ('temp_edge_c', 'temp1_input', 1e-3), ('temp_junction_c', 'temp2_input', 1e-3),
                                 ('temp_mem_c', 'temp3_input', 1e-3)):
            txt = _read(f'{hw}/{name}')
            if txt and key not in out:
                try:
                    out[key] = float(txt)*scale
                except ValueError:
                    pass
    return out


def make_batches(n_batches, rank, device):
    """Synthetic (mixture, clean) items of SURVEY.md section 8(d), collated exactly as
    the trainer would (transform = mono mean, zero-pad collate) and moved to HBM."""
    model_transform = lambda s: s.mean(axis=-2)   # noqa: E731  ConvTasNet.transform
    dset = SyntheticMixtureDataset(n_batches*BATCH, int(SECONDS*FS),
                                   transform=model_transform, seed=rank)
    batches = []
    for i in range(n_batches):
        items = [dset[i*BATCH + j] for j in range(BATCH)]
        batch, lengths = BreverDataLoader._collate_fn(items)
        batches.append((batch.to(device), lengths.to(device)))
    return batches


def physical_cores():
    """Physical cores of the host (SMT siblings counted once); falls back to the logical count."""
    try:
        pairs = set()
        phys = core = None
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('physical id'):
                    phys = line.split(':')[1].strip()
                elif line.startswith('core id'):
                    core = line.split(':')[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        pairs.add((phys, core))
                    phys = core = None
        if pairs:
            return len(pairs)
    except OSError:
        pass
    return os.cpu_count() or 1


def cpu_baseline():
    """Oracle (pure-PyTorch CPU restatement of the reference path) timed on the host cores,
    bounded to about a minute and a half: the thread count is scanned over {8, 32, physical cores} on train
    steps of 4 x 4 s utterances (one warm-up step on 2 utterances each; all physical cores oversubscribe the
    host: VERDICT r02 item 9), then the benchmark batch itself (16 x 4 s, fp32) runs at the best count:
    1 warm-up step + 3 timed steps -> `value` (SURVEY 8d asks 3 + 5: that is 2 minutes of host time on
    this box, the warm step is what matters -- the first pass over ~30 GB of fresh activation memory is
    30 % slower). `value_cpu_bf16`: a 4-utterance step under CPU bf16 autocast (the reference's CPU autocast
    dtype, convtasnet.py:81)."""
    from oracle.convtasnet import OracleConvTasNet
    cores = physical_cores()
    model_transform = lambda s: s.mean(axis=-2)   # noqa: E731
    dset = SyntheticMixtureDataset(BATCH, int(SECONDS*FS), transform=model_transform)
    batch, lengths = BreverDataLoader._collate_fn([dset[i] for i in range(BATCH)])
    scaler = torch.amp.GradScaler('cuda', enabled=False)

    def run(threads, amp, items, warm_items=2, steps=1):
        torch.set_num_threads(threads)
        torch.manual_seed(0)
        model = OracleConvTasNet()
        model.train_step(batch[:warm_items], lengths[:warm_items], amp, scaler)   # warm-up (threads, allocator)
        times = []
        for _ in range(steps):
            t0 = time.perf_counter()
            model.train_step(batch[:items], lengths[:items], amp, scaler)
            times.append(time.perf_counter() - t0)
        return items*steps/sum(times), times

    counts = sorted({min(8, cores), min(32, cores), cores})
    by_threads = {n: run(n, False, 4)[0] for n in counts}
    best = max(by_threads, key=by_threads.get)
    full, step_times = run(best, False, BATCH, warm_items=BATCH, steps=3)
    bf16 = run(best, True, 4)[0]
    return {
        'value': full, 'unit': 'utterances/s', 'cores': best, 'kind': 'port',
        'physical_cores': cores, 'by_threads_4_utterances': {str(k): v for k, v in by_threads.items()},
        'step_seconds': step_times, 'value_cpu_bf16_4_utterances': bf16,
        'sample': f'value: 3 timed train steps of the benchmark batch ({BATCH} x 4 s, fp32) after 1 warm-up step of '
                  f'the same batch, Conv-TasNet defaults, torch CPU oracle, {best} threads = the best of {counts} '
                  'on a 4-utterance step (by_threads_4_utterances; a 4-utterance step runs ~2.8x the per-'
                  'utterance rate of the 16-utterance one: its activations stay in the last-level cache); '
                  f'value_cpu_bf16_4_utterances: 1 step of 4 utterances under CPU bf16 autocast at {best} threads',
    }


# bench label -> substring of the HIP kernel name in the rocprofv3 output
KERNEL_OF_LABEL = {
    'gln_prelu_bwd': 'dz_kernel', 'dwconv_bwd': 'dwconv_bwd_halo_kernel', 'dwpw2_bwd': ['dwconv_bwd_fused_kernel<3, 256>', 'dwconv_bwd_fused_kernel'],
    'dwconv_fwd': 'dwconv_fwd_kernel', 'pw2_wgrad': 'wgrad_full_kernel',
    # (fused forward: 23 of the 24 first-conv launches finish the block input while staging it)
    'pw1_fwd': ['gemm_ws_kernel<128, 64, 1, 0, 2,', 'gemm_ws_kernel<128, 64, 1, 0,'],
    'pw2_fwd': 'gemm_ws_kernel<512,', 'dwpw2_fwd': 'dwpw2_fused_kernel',
    'skip_combine': 'skip_combine_kernel',
    'pw2_dgrad': 'gemm_ws_kernel<256,', 'pw1_dgrad': ['pw1_dgrad_ws_kernel', 'gemm_rows_kernel<128, 3, 5>'],
    'pw1_wgrad': ('gemm_wgrad_kernel<128, 0>', 'largest'), 'clip_adam': 'clip_adam_kernel',
}
PMC_FILES = ('r06_pmc_hbm_traffic.json', 'r05_pmc_hbm_traffic.json', 'r04_pmc_hbm_traffic.json', 'r03_pmc_hbm_traffic.json', 'r02_pmc_hbm_traffic.json', 'r01_pmc_hbm_traffic.json')


def pmc_traffic(label):
    """(HBM bytes per launch, source file) of the kernel behind `label`, from the committed
    rocprofv3 PMC passes (profiles/rNN_pmc_hbm_traffic.json, tools/pmc_traffic.py: FETCH_SIZE
    x2 on gfx950 + WRITE_SIZE, separate passes, same bench command). bench.py cannot run the
    profiler on itself; (None, None) when the file or the kernel is missing."""
    key = KERNEL_OF_LABEL.get(label)
    field = 'hbm_traffic_MB'
    if isinstance(key, tuple):        # the template also serves smaller launches: take the largest
        key, field = key[0], 'hbm_traffic_MB_largest_launch'
    keys = key if isinstance(key, list) else [key]      # candidates, most specific first
    for fname in PMC_FILES:
        path = os.path.join(ROOT, 'profiles', fname)
        if key is None or not os.path.exists(path):
            continue
        with open(path) as f:
            kernels = json.load(f)['kernels']
        for k in keys:
            for name, v in kernels.items():
                if k in name and field in v:
                    return v[field]*1e6, f'profiles/{fname}'
    return None, None


SQ_FILES = ('r06_sq_counters.json', 'r05_sq_counters.json', 'r04_sq_counters.json')


def sq_counters(label):
    """MFMA-busy and the other SQ shares of the kernel behind `label` from the committed rocprofv3 --pmc passes
    (profiles/rNN_sq_counters.json, tools/profile_sq.sh + tools/sq_counters.py: SQ_VALU_MFMA_BUSY_CYCLES /
    (1024 SIMDs x GRBM_GUI_ACTIVE / 8), VALU issue cycles likewise, wait / issue-stall shares of SQ_WAVE_CYCLES).
    None when the file or the kernel is missing."""
    key = KERNEL_OF_LABEL.get(label)
    if isinstance(key, tuple):
        key = key[0]
    keys = key if isinstance(key, list) else [key]
    for fname in SQ_FILES:
        path = os.path.join(ROOT, 'profiles', fname)
        if key is None or not os.path.exists(path):
            continue
        with open(path) as f:
            kernels = json.load(f)['kernels']
        for k in keys:
            for name, v in kernels.items():
                if k in name and 'mfma_busy_frac' in v:
                    out = {n: v[n] for n in ('mfma_busy_frac', 'valu_busy_frac', 'lds_array_busy_frac',
                                             'wait_any_share', 'issue_stall_share', 'waves_resident_per_simd',
                                             'lds_bank_conflict_frac') if n in v}
                    out['source'] = f'profiles/{fname} (rocprofv3 --pmc SQ_*, one-chain step)'
                    return out
    return None


def kernel_roofline(model, batches, scaler, steps=3):
    """Event-timed steps (HIP events around every launch, on the launch stream):
    roofline of the kernel with the largest total time."""
    lib = hip.lib()
    # one chain for this pass: the timed region runs the two half-batch chains of
    # ConvTasNet._train_step_two_chains concurrently, where a launch's wall time includes the other
    # chain's workgroups; a kernel's own duration and bytes per launch are those of the whole batch
    chains = os.environ.get('BRV_CTN_STREAMS')
    os.environ['BRV_CTN_STREAMS'] = '1'
    hip.prof_enable(1)
    for i in range(steps):
        batch, lengths = batches[i % len(batches)]
        model.train_step(batch, lengths, True, scaler)
    torch.cuda.synchronize()
    prof = hip.profile_collect()
    hip.prof_enable(0)
    if chains is None:
        del os.environ['BRV_CTN_STREAMS']
    else:
        os.environ['BRV_CTN_STREAMS'] = chains
    # the same kernel as launched inside the timed region (default mode), for the record
    hip.prof_enable(1)
    for i in range(steps):
        batch, lengths = batches[i % len(batches)]
        model.train_step(batch, lengths, True, scaler)
    torch.cuda.synchronize()
    prof_timed = hip.profile_collect()
    hip.prof_enable(0)
    if not prof:
        return None, {}
    label, top = max(prof.items(), key=lambda kv: kv[1]['ms'])
    avg_s = top['ms']/top['calls']*1e-3
    flops = top['flops']/top['calls']
    nbytes = top['bytes']/top['calls']
    gbs = nbytes/avg_s/1e9
    tfs = flops/avg_s/1e12
    intensity = flops/max(nbytes, 1.0)
    ridge = PEAK_MFMA_TFLOPS*1e12/(PEAK_HBM_GBS*1e9)
    hbm_bound = intensity < ridge
    traffic, traffic_src = pmc_traffic(label)
    roof = {
        'kernel': label,
        'bound': 'hbm' if hbm_bound else 'mfma',
        'achieved': gbs if hbm_bound else tfs,
        'peak': PEAK_HBM_GBS if hbm_bound else PEAK_MFMA_TFLOPS,
        'unit': 'GB/s' if hbm_bound else 'TFLOP/s',
        'frac': (gbs/PEAK_HBM_GBS) if hbm_bound else (tfs/PEAK_MFMA_TFLOPS),
        'traffic': traffic,
        'traffic_source': f'{traffic_src} (rocprofv3 --pmc, bytes per launch)' if traffic_src else None,
        'sq_counters': sq_counters(label),
        'mfma_busy_frac': (sq_counters(label) or {}).get('mfma_busy_frac'),
        'mode': 'one kernel chain (BRV_CTN_STREAMS=1): whole-batch launches, the kernel alone on the chip; '
                'the timed region overlaps two half-batch chains',
        'timed_region': None if label not in prof_timed else {
            'launches_per_step': prof_timed[label]['calls']/steps,
            'avg_launch_us': prof_timed[label]['ms']/prof_timed[label]['calls']*1e3,
            'algorithmic_bytes_per_launch': prof_timed[label]['bytes']/prof_timed[label]['calls'],
            'note': 'event-timed wall time of a launch inside the default step: half-batch launches '
                    'of two chains that share the chip'},
        'avg_launch_us': avg_s*1e6,
        'launches_per_step': top['calls']/steps,
        'algorithmic_bytes_per_launch': nbytes,
        'algorithmic_flops_per_launch': flops,
        'tflops': tfs, 'gbs': gbs,
        'share_of_kernel_time': top['ms']/sum(v['ms'] for v in prof.values()),
    }
    if label == 'dwpw2_bwd':
        roof['note'] = ('fused backward stage (csrc/bwd_fused.cuh, round 3): one launch does the work of '
                        'pw2_dgrad + dwconv_bwd (round 2: 163 + 262 MB algorithmic in 48 + 66 us = 3.7 TB/s) '
                        'with 229 MB algorithmic: fewer bytes AND less time per block, hence a lower '
                        'bytes-per-second fraction than the kernel pair it replaces')
    total_ms = sum(v['ms'] for v in prof.values())
    kernels = []
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms']):
        if v['ms'] < 0.05*total_ms:
            continue
        per = v['ms']/v['calls']*1e-3
        kb = v['bytes']/v['calls']
        kernels.append({
            'kernel': k, 'launches_per_step': v['calls']/steps, 'avg_launch_us': per*1e6,
            'share_of_kernel_time': v['ms']/total_ms,
            'algorithmic_bytes_per_launch': kb, 'achieved_GBs': kb/per/1e9,
            'frac_of_hbm_peak': kb/per/1e9/PEAK_HBM_GBS, 'traffic': pmc_traffic(k)[0],
            'mfma_busy_frac': (sq_counters(k) or {}).get('mfma_busy_frac'),
            'valu_busy_frac': (sq_counters(k) or {}).get('valu_busy_frac')})
    roof['kernels_over_5pct'] = kernels
    table = {k: {'calls_per_step': v['calls']/steps,
                 'ms_per_step': v['ms']/steps,
                 'gbs': v['bytes']/max(v['ms'], 1e-9)/1e6,
                 'tflops': v['flops']/max(v['ms'], 1e-9)/1e9}
             for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])}
    return roof, table


def fp32_path(device, steps=8, warmup=2):
    """The same step with `use_amp=False` (the trainer's and `enhance()`'s default precision; fp32
    activations, products of fp32 accuracy: DESIGN.md 5a) -- reported beside the headline, never as
    `value`. A fresh model on the same synthetic batch shape."""
    torch.manual_seed(1)
    model = ConvTasNet().to(device)
    scaler = torch.amp.GradScaler('cuda', enabled=False)
    batch = 0.1*torch.randn(BATCH, 2, int(SECONDS*FS), device=device)
    lengths = torch.full((BATCH,), int(SECONDS*FS), device=device)
    for _ in range(warmup):
        model.train_step(batch, lengths, False, scaler)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model.train_step(batch, lengths, False, scaler)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0)/steps
    del model
    return {'value': BATCH/dt, 'unit': 'utterances/s', 'ms_per_step': dt*1e3, 'dtype': 'fp32', 'steps': steps,
            'tflops': BATCH/dt*FLOP_PER_UTT_TRAIN/1e12,
            'note': 'use_amp=False path (brv_ctn_f32_*): fp32 activations, fp32-MFMA and split-bf16 products, '
                    'bitwise repeatable; round 2: 278 utterances/s'}


ROWS_ROOFLINE_FILES = ('r06_rows_roofline.json', 'r05_rows_roofline.json')


def rows_dominant_kernel(row):
    """`roofline` object of the dominant kernel of a widened row from the committed per-round profile
    (profiles/rNN_rows_roofline.json, one JSON line per row: tools/profile_rows.sh -> tools/rows_roofline.py: PMC
    bytes per launch over the rocprofv3 --stats average duration). bench.py cannot run the profiler on itself."""
    for fname in ROWS_ROOFLINE_FILES:
        path = os.path.join(ROOT, 'profiles', fname)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            for ln in f:
                ln = ln.strip()
                if not ln:
                    continue
                rec = json.loads(ln)
                if rec.get('row') == row and rec.get('roofline'):
                    return dict(rec['roofline'], source=f'profiles/{fname} (rocprofv3 --kernel-trace --stats + --pmc '
                                                        'FETCH_SIZE / WRITE_SIZE passes of the same workload)')
    return None


def _timed_steps(fn, steps, min_warm_s=1.0, warm_steps=3):
    """(seconds per call, warm-up calls): `warm_steps` calls, then more until `min_warm_s` seconds have run, then
    `steps` timed calls between two synchronisations."""
    t0 = time.perf_counter()
    n = 0
    while n < warm_steps or time.perf_counter() - t0 < min_warm_s:
        fn()
        torch.cuda.synchronize()
        n += 1
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0)/steps, n


def other_configs(device, dccrn_steps=30, sgmse_repeats=3):
    """BASELINE.json configs[3] and configs[4] in the driver-run line (the Conv-TasNet step
    stays the headline `value`). Each entry has the fields of the top-level line: value / unit / ms_per_step /
    dtype / config.workload, and `roofline` = whole-workload MFMA fraction (algorithmic FLOPs of SURVEY.md 8d over
    the measured wall time) with the row's dominant kernel from the committed profile of the round.

    * configs[3] DCCRN (brever/models/dccrn/dccrn.py:28-142), defaults, 3 671 053 parameters, bf16 matrix products
      (`use_amp`), train step = STFT -> complex Conv2d encoder -> complex LSTM -> decoder -> mask -> iSTFT -> SNR
      loss -> backward -> clip 5.0 + Adam on 16 x 4 s utterances resident in HBM;
    * configs[4] SGMSE+ (brever/models/sgmse/sgmse.py:178-193), defaults, 65.6 M parameters, fp16-MFMA convolutions
      (`use_amp`): `enhance` = STFT -> 30-step reverse SDE with the predictor-corrector sampler (60 network
      evaluations) -> iSTFT overlap-add, at batch 1 (latency) and batch 8 (throughput)."""
    from brever_amd.models import ModelRegistry
    out = []
    L = int(SECONDS*FS)
    scaler = torch.amp.GradScaler('cuda', enabled=False)
    # ---- DCCRN ------------------------------------------------------------------------------------
    torch.manual_seed(0)
    model = ModelRegistry.get('dccrn')().to(device)
    model.train()
    wav = 0.1*torch.randn(BATCH, 2, 2, L, device=device)                    # (B, sources, channels, L)
    x = torch.stack([model.transform(w) for w in wav])
    lengths = torch.full((BATCH,), x.shape[-1], device=device)
    dt, nwarm = _timed_steps(lambda: model.train_step(x, lengths, True, scaler), dccrn_steps)
    tflops = BATCH/dt*3*51.2e9/1e12                 # 51.2 GFLOP per 4 s utterance forward, x 3
    out.append({
        'metric': 'utterances/sec (4 s @16 kHz) DCCRN train', 'value': BATCH/dt, 'unit': 'utterances/s',
        'n_gpus': 1, 'steps': dccrn_steps, 'warmup': nwarm, 'ms_per_step': dt*1e3, 'higher_is_better': True,
        'dtype': 'bf16', 'data': 'synthetic', 'vs_baseline': None,
        'config': {'baseline_config': 'configs[3]: DCCRN (complex STFT Conv2d+LSTM) bf16, 1xMI355X, STFT/iSTFT HIP kernels',
                   'workload': 'DCCRN defaults (3 671 053 params) train step: STFT 512/128 -> complex Conv2d + LSTM -> '
                               'iSTFT, fwd + SNR loss + bwd + clip 5.0 + Adam, 16 x 4 s @ 16 kHz utterances resident in HBM',
                   'global_batch': BATCH, 'seq_len': L},
        'roofline': {'bound': 'mfma', 'achieved': tflops, 'peak': PEAK_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': tflops/PEAK_MFMA_TFLOPS, 'traffic': None,
                     'scope': 'whole workload: 3 x 51.2 GFLOP per utterance (SURVEY.md 8d) over the measured wall time',
                     'dominant_kernel': rows_dominant_kernel('dccrn_bf16')}})
    del model, x, wav
    torch.cuda.empty_cache()
    # ---- SGMSE+ -----------------------------------------------------------------------------------
    torch.manual_seed(0)
    nsteps = 30
    model = ModelRegistry.get('sgmsep')(solver_num_steps=nsteps).to(device).eval()
    for batch in (1, 8):
        wav = 0.1*torch.randn(batch, 2, L, device=device)
        dt, nwarm = _timed_steps(lambda: model.enhance(wav, use_amp=True), sgmse_repeats, min_warm_s=0.0, warm_steps=2)
        tflops = batch*2*nsteps*1.04/dt             # 1.04 TFLOP per network evaluation and 4 s utterance
        out.append({
            'metric': 'utterances/sec SGMSE+ enhance (30-step reverse SDE)', 'value': batch/dt, 'unit': 'utterances/s',
            'n_gpus': 1, 'steps': sgmse_repeats, 'warmup': nwarm, 'ms_per_step': dt*1e3, 'higher_is_better': True,
            'dtype': 'fp16', 'data': 'synthetic', 'vs_baseline': None,
            's_per_utterance': dt/batch, 'ms_per_network_evaluation': dt/(2*nsteps)*1e3, 'rtf': dt/batch/SECONDS,
            'config': {'baseline_config': 'configs[4]: SGMSE+ score-model inference (30-step reverse SDE) fp16, 1xMI355X, '
                                          'iSTFT overlap-add kernel',
                       'workload': f'SGMSE+ defaults (65.6 M params) enhance: STFT -> 30-step reverse SDE, PC sampler = 60 '
                                   f'network evaluations (HIP-graph replay) -> iSTFT overlap-add, {batch} x 4 s @ 16 kHz, '
                                   'fp16-MFMA convolutions; one step = one enhance call',
                       'global_batch': batch, 'seq_len': L},
            'roofline': {'bound': 'mfma', 'achieved': tflops, 'peak': PEAK_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': tflops/PEAK_MFMA_TFLOPS, 'traffic': None,
                         'scope': 'whole workload: 60 evaluations x 1.04 TFLOP per utterance (SURVEY.md 8d) over the measured '
                                  'wall time',
                         'dominant_kernel': rows_dominant_kernel(f'sgmse_b{batch}')}})
    del model
    torch.cuda.empty_cache()
    return out


def trainer_sampler(rank, world, n_steps):
    """Dataset and batch sampler of `through_trainer`: ONE synthetic dataset of world x 16 x n_steps items (the
    same on every rank), bucket batches of 64 s, and for world > 1 the reference's DistributedBatchSamplerWrapper
    (brever/training.py:119-125, batching.py:279-290): every rank takes a disjoint share of the SAME batch list and
    runs the same number of steps."""
    from brever_amd.batching import DistributedBatchSamplerWrapper
    dset = SyntheticMixtureDataset(world*BATCH*n_steps, int(SECONDS*FS),
                                   transform=lambda s: s.mean(axis=-2), seed=100)
    sampler = BucketBatchSampler(dset, batch_size=BATCH*SECONDS, dynamic=True, fs=FS, seed=0)
    if world > 1:
        sampler = DistributedBatchSamplerWrapper(sampler, num_replicas=world, rank=rank, shuffle=False)
    sampler.set_epoch(0)
    return dset, sampler


def through_trainer(model, scaler, rank, world, device, steps, warmup):
    """The same train step fed by the trainer's data path instead of resident batches: items of
    a host-side SyntheticMixtureDataset (what BreverDataset.__getitem__ + model.transform
    return) -> BucketBatchSampler (dynamic, 64 s = 16 utterances) -> [DistributedBatchSamplerWrapper,
    world > 1: ONE dataset shared by all ranks, every rank takes its disjoint share of the batch list, as
    brever/training.py:119-125 + batching.py:279-290 do] -> BreverDataLoader collate -> DevicePrefetcher
    (pinned double-buffered async H2D) -> train_step. Returns (ms per step, batches this rank consumed)."""
    # as BreverTrainer does on a GPU: few host threads (collate is small copies; 128 spinning
    # OpenMP workers cost 40+ ms per step on this host, tools/trainer_path_debug.py)
    torch.set_num_threads(min(torch.get_num_threads(), int(os.environ.get('BREVER_HOST_THREADS', '4'))))
    dset, sampler = trainer_sampler(rank, world, steps + warmup)
    mine = sorted({i for batch in sampler for i in batch})
    sampler.set_epoch(0)              # (a shuffling sampler composes its batches once per set_epoch call)
    inner = getattr(sampler, 'sampler', sampler)
    inner._previous_epoch = None      # the same epoch again: the loader must see the batches listed above
    dset.preload_indices(mine)        # synthesis is not part of the path being timed (this rank's items only)
    loader = BreverDataLoader(dataset=dset, batch_sampler=sampler, num_workers=0)
    t0 = None
    done = 0
    for batch, lengths in DevicePrefetcher(loader, device):
        if done == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        model.train_step(batch, lengths, True, scaler)
        done += 1
    torch.cuda.synchronize()
    return (time.perf_counter() - t0)/max(done - warmup, 1)*1e3, done


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fp32-path', action='store_true', help='skip the use_amp=False step reported beside the headline')
    ap.add_argument('--no-through-trainer', action='store_true')
    ap.add_argument('--min-warmup-s', type=float, default=MIN_WARMUP_S,
                    help='seconds of steps before the timed region whatever --warmup says (the profiler passes of '
                         'tools/profile_*.sh pass 0: under --pmc every launch is serialised)')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip BASELINE configs[3] / configs[4] (DCCRN train, SGMSE+ enhance) behind the headline; '
                         'they are N = 1 rows and never run with a process group')
    ap.add_argument('--buckets', type=int, default=3,
                    help='gradient buckets of the overlapped all-reduce (N > 1)')
    ap.add_argument('--kernel-table', action='store_true',
                    help='also print the per-kernel table (stderr)')
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'],
                    help='nccl = RCCL (the measured configuration). gloo exists for the test that runs the N > 1 '
                         'code of this file with two ranks on ONE GPU (ranks share cuda:0, gradients move through '
                         'the host): its numbers mean nothing')
    ap.add_argument('--dist', action='store_true',
                    help='initialise the process group even at --gpus 1 (world-1 RCCL: every N > 1 branch runs)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with `python -m '
                         f'torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} ...` '
                         '(one rank per GPU) or pass --gpus 1')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm device (no CPU fallback)')
    if args.dist_backend == 'gloo':
        local_rank %= torch.cuda.device_count()      # (test mode: the ranks share the GPUs there are)
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    single = world == 1 and not args.dist            # no process group: the driver's N = 1 run
    if not single:
        from brever_amd.parallel import init_process_group
        kw = {'device_id': device} if args.dist_backend == 'nccl' else {}
        if world == 1:
            os.environ.setdefault('MASTER_PORT', '29533')
            kw.update(rank=0, world_size=1)
        init_process_group(args.dist_backend, timeout_s=float(os.environ.get('BRV_DIST_TIMEOUT_S', '120')),
                           **kw)                     # short timeout: a bad rendezvous exits non-zero

    torch.manual_seed(0)
    model = ConvTasNet().to(device)          # defaults = BASELINE config
    sync = None
    if not single:
        broadcast_parameters(model)
        sync = GradSynchronizer(model, nparts=args.buckets)
    scaler = torch.amp.GradScaler('cuda', enabled=False)   # bf16: no loss scaling
    batches = make_batches(4, rank, device)

    def step(i):
        batch, lengths = batches[i % len(batches)]
        return model.train_step(batch, lengths, True, scaler)

    t_first = time.perf_counter()
    for i in range(args.warmup):
        step(i)
    fallback = None
    if sync is not None and args.buckets > 1:
        # the bucketed all-reduce must hide behind backward; if the warm-up shows more than 0.5 ms of
        # it exposed on any rank (xGMI ring latency x buckets), use ONE all-reduce after backward
        t = torch.tensor([sync.exposed_ms()], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        forced = os.environ.get('BRV_FORCE_AR_FALLBACK', '0') == '1'      # (tests: take the branch once)
        if float(t) > 0.5 or forced:
            fallback = (f'one all-reduce after backward: {args.buckets} buckets left {float(t):.2f} ms '
                        'exposed per step in the warm-up' + (' (forced by BRV_FORCE_AR_FALLBACK)' if forced else ''))
            sync = GradSynchronizer(model, nparts=1)
            for i in range(max(2, args.warmup//2)):
                step(i)
    # warm-up by TIME: the --warmup steps above are 33 ms of work -- the chip is still climbing out of its idle
    # power state when they end (round 5: the driver's box read 4 % under the builder's with nothing to attribute
    # it to). Steps continue until MIN_WARMUP_S seconds of steady steps have run; the count is agreed over the
    # ranks first (every step holds collectives: all ranks must run the same number).
    torch.cuda.synchronize()
    probe = 5
    t_p = time.perf_counter()
    for i in range(probe):
        step(i)
    torch.cuda.synchronize()
    extra = 0
    while args.min_warmup_s > 0:
        # (rounds of steps until the time is reached: the first estimate of a step comes from a chip that is still
        # warming up and is too long; every round's count is the maximum over the ranks)
        done_s = time.perf_counter() - t_p
        per_step = done_s/(probe + extra)
        more = int((args.min_warmup_s - done_s)/max(per_step, 1e-4)) + 1 if done_s < args.min_warmup_s else 0
        if not single:
            t = torch.tensor([more], dtype=torch.int64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            more = int(t)
        if more <= 0:
            break
        for i in range(more):
            step(i)
        torch.cuda.synchronize()
        extra += more
    warm = {'steps': args.warmup + probe + extra, 'seconds': time.perf_counter() - t_first,
            'seconds_of_steady_steps': time.perf_counter() - t_p,
            'min_seconds_of_steady_steps': args.min_warmup_s,
            'note': f'--warmup {args.warmup} (incl. first-step allocations) + {probe} probe steps + {extra} more until '
                    f'{args.min_warmup_s} s of steps had run'}

    def timed_block():
        if not single:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            last = step(i)
        torch.cuda.synchronize()
        if not single:
            dist.barrier()
        dt_ = time.perf_counter() - t0
        if not single:
            t = torch.tensor([dt_], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t)
        return dt_, last

    clock_before = read_clock(local_rank)
    dt, loss = timed_block()                       # THE timed region: exactly --steps steps -> `value`
    clock_after = read_clock(local_rank)
    repeats = [timed_block()[0] for _ in range(N_REPEATS)]
    clock_end = read_clock(local_rank)
    final_loss = float(loss)
    exposed = sync.exposed_ms() if sync is not None else None
    trainer_ms = None
    trainer_batches = None
    if not args.no_through_trainer:
        trainer_ms, trainer_batches = through_trainer(model, scaler, rank, world, device, min(args.steps, 40), 6)
        if not single:
            t = torch.tensor([trainer_ms], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            trainer_ms = float(t)
            n = torch.tensor([trainer_batches], dtype=torch.int64, device=device)
            lo, hi = n.clone(), n.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            if int(lo) != int(hi):      # the wrapper pads by repetition: every rank must run the same steps
                raise SystemExit(f'ranks consumed different batch counts: {int(lo)} .. {int(hi)}')

    roof, table = kernel_roofline(model, batches, scaler)
    if not single:
        dist.barrier()
    if rank == 0:
        value = world*BATCH*args.steps/dt
        line = {
            'metric': 'utterances/sec (4 s @16 kHz) Conv-TasNet train',
            'value': value, 'unit': 'utterances/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt/args.steps*1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16', 'data': 'synthetic',
            'config': {
                'workload': 'Conv-TasNet (defaults, 4 935 217 params) train step: fwd + '
                            'SNR loss + bwd + clip 5.0 + Adam, 16 x 4 s @ 16 kHz '
                            'utterances per GPU resident in HBM',
                'global_batch': world*BATCH, 'seq_len': int(SECONDS*FS),
                'parallelism': f'dp{world}',
            },
            'value_includes_h2d': False,        # (timed on batches resident in HBM; `through_trainer` includes H2D)
            'warmup_effective': warm,
            'value_repeats': {
                'blocks': N_REPEATS, 'steps_per_block': args.steps,
                'min': world*BATCH*args.steps/max(repeats), 'median': world*BATCH*args.steps/sorted(repeats)[len(repeats)//2],
                'max': world*BATCH*args.steps/min(repeats), 'unit': 'utterances/s',
                'ms_per_step': [r/args.steps*1e3 for r in repeats],
                'note': 'further blocks of --steps timed like `value`, right behind it; `value` is the first block'},
            'clock': {'before_timed_region': clock_before, 'after_timed_region': clock_after,
                      'after_repeats': clock_end,
                      'source': 'sysfs (pp_dpm_sclk / pp_dpm_mclk active level, hwmon freq1_input, power1_*, temp*_input) '
                                'of the device behind cuda:<local rank>, rank 0'},
            'final_loss': final_loss,
            'whole_step_mfma_frac': value/world*FLOP_PER_UTT_TRAIN/(PEAK_MFMA_TFLOPS*1e12),
            'roofline': roof,
        }
        if trainer_ms is not None:
            line['through_trainer'] = {
                'value': world*BATCH/(trainer_ms*1e-3), 'unit': 'utterances/s',
                'ms_per_step': trainer_ms,
                'batches_per_rank': trainer_batches,
                'includes_h2d': True,            # SURVEY 8(d)'s "incl. H2D of the batch" is THIS number
                'path': 'host items of ONE dataset -> BucketBatchSampler'
                        + (' -> DistributedBatchSamplerWrapper (disjoint batches per rank)' if world > 1 else '')
                        + ' -> BreverDataLoader collate -> pinned double-buffered async H2D (DevicePrefetcher) '
                          '-> train_step'}
        line['two_chain'] = bool(ConvTasNet.uses_two_chains(BATCH, True))
        if exposed is not None:
            line['allreduce_exposed_ms'] = exposed
            line['allreduce_buckets'] = sync.nparts
            line['rccl_world_size'] = dist.get_world_size()
            line['hw_queues_ok'] = bool(__import__('brever_amd').HW_QUEUES_OK)
            if fallback:
                line['allreduce_fallback'] = fallback
        if single and not args.no_fp32_path:
            line['fp32_path'] = fp32_path(device)
        if single and not args.no_other_configs:
            t_oc = time.perf_counter()
            line['other_configs'] = other_configs(device)
            line['other_configs_wall_s'] = time.perf_counter() - t_oc
        if single and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline()
        if args.kernel_table:
            for k, v in table.items():
                print(f'{k:18s} {v["calls_per_step"]:6.1f} calls/step '
                      f'{v["ms_per_step"]:8.3f} ms/step {v["gbs"]:8.1f} GB/s '
                      f'{v["tflops"]:8.1f} TFLOP/s', file=sys.stderr)
        print(json.dumps(line))
    if not single:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
