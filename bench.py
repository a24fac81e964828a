"""Headline benchmark: Conv-TasNet training throughput on synthetic 4 s @ 16 kHz
mixtures (BASELINE.json configs[1]; weak scaling over N GPUs of one node).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one optimizer step of the reference's hot path on one batch of 16
utterances per GPU, inputs already resident in HBM: ConvTasNet.train_step =
forward -> SNR loss -> backward -> [gradient all-reduce over RCCL] -> clip(5.0) +
Adam (brever/models/base.py:178-210, convtasnet.py:78-89). Rank 0 prints ONE JSON
line. The CPU oracle (oracle/) is used only for the `cpu_baseline` field.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from brever_amd import hip                                    # noqa: E402
from brever_amd.data import BreverDataLoader, SyntheticMixtureDataset  # noqa: E402
from brever_amd.models import ConvTasNet                      # noqa: E402
from brever_amd.parallel import GradSynchronizer, broadcast_parameters  # noqa: E402

FS = 16000
SECONDS = 4.0
BATCH = 16
FLOP_PER_UTT_TRAIN = 116.46e9       # 3 x 38.82 GFLOP forward
PEAK_HBM_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PEAK_MFMA_TFLOPS = 2500.0           # dense bf16 MFMA


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


def cpu_baseline():
    """Oracle (pure-PyTorch CPU restatement of the reference path) timed on the host
    cores on a bounded sample of the same workload."""
    from oracle.convtasnet import OracleConvTasNet
    torch.manual_seed(0)
    model = OracleConvTasNet()
    bsz = 2
    model_transform = lambda s: s.mean(axis=-2)   # noqa: E731
    dset = SyntheticMixtureDataset(bsz, int(SECONDS*FS), transform=model_transform)
    batch, lengths = BreverDataLoader._collate_fn([dset[i] for i in range(bsz)])
    scaler = torch.amp.GradScaler('cuda', enabled=False)
    model.train_step(batch, lengths, False, scaler)           # warm-up
    steps = 2
    t0 = time.perf_counter()
    for _ in range(steps):
        model.train_step(batch, lengths, False, scaler)
    dt = time.perf_counter() - t0
    return {
        'value': steps*bsz/dt, 'unit': 'utterances/s',
        'cores': torch.get_num_threads(), 'kind': 'port',
        'sample': f'{steps} fp32 train steps of {bsz} x 4 s utterances (1 warm-up), '
                  f'Conv-TasNet defaults, torch CPU oracle',
    }


# bench label -> substring of the HIP kernel name in the rocprofv3 output
KERNEL_OF_LABEL = {
    'gln_prelu_bwd': 'dz_kernel', 'dwconv_bwd': 'dwconv_bwd_halo_kernel',
    'dwconv_fwd': 'dwconv_fwd_kernel', 'pw2_wgrad': 'wgrad_full_kernel',
    'pw1_fwd': 'gemm_ws_kernel<128, 64, 1, 0,', 'pw2_fwd': 'gemm_ws_kernel<512,',
    'pw2_dgrad': 'gemm_ws_kernel<256,', 'pw1_dgrad': 'gemm_rows_kernel<128, 3, 5>',
}


def pmc_traffic(label):
    """HBM bytes per launch of the kernel behind `label`, from the committed rocprofv3
    PMC passes (profiles/r01_pmc_hbm_traffic.json: FETCH_SIZE x2 on gfx950 + WRITE_SIZE,
    separate passes, same bench command). bench.py cannot run the profiler on itself;
    None when the file or the kernel is missing."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles',
                        'r01_pmc_hbm_traffic.json')
    key = KERNEL_OF_LABEL.get(label)
    if key is None or not os.path.exists(path):
        return None
    with open(path) as f:
        kernels = json.load(f)['kernels']
    for name, v in kernels.items():
        if key in name:
            return v['hbm_traffic_MB']*1e6
    return None


def kernel_roofline(model, batches, scaler, steps=3):
    """Event-timed steps (HIP events around every launch, on the launch stream):
    roofline of the kernel with the largest total time."""
    lib = hip.lib()
    lib.brv_prof_enable(1)
    for i in range(steps):
        batch, lengths = batches[i % len(batches)]
        model.train_step(batch, lengths, True, scaler)
    torch.cuda.synchronize()
    prof = hip.profile_collect()
    lib.brv_prof_enable(0)
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
    roof = {
        'kernel': label,
        'bound': 'hbm' if hbm_bound else 'mfma',
        'achieved': gbs if hbm_bound else tfs,
        'peak': PEAK_HBM_GBS if hbm_bound else PEAK_MFMA_TFLOPS,
        'unit': 'GB/s' if hbm_bound else 'TFLOP/s',
        'frac': (gbs/PEAK_HBM_GBS) if hbm_bound else (tfs/PEAK_MFMA_TFLOPS),
        'traffic': pmc_traffic(label),
        'traffic_source': 'profiles/r01_pmc_hbm_traffic.json (rocprofv3 --pmc, bytes per launch)',
        'avg_launch_us': avg_s*1e6,
        'launches_per_step': top['calls']/steps,
        'algorithmic_bytes_per_launch': nbytes,
        'algorithmic_flops_per_launch': flops,
        'tflops': tfs, 'gbs': gbs,
        'share_of_kernel_time': top['ms']/sum(v['ms'] for v in prof.values()),
    }
    table = {k: {'calls_per_step': v['calls']/steps,
                 'ms_per_step': v['ms']/steps,
                 'gbs': v['bytes']/max(v['ms'], 1e-9)/1e6,
                 'tflops': v['flops']/max(v['ms'], 1e-9)/1e9}
             for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])}
    return roof, table


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--kernel-table', action='store_true',
                    help='also print the per-kernel table (stderr)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run for --gpus > 1')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm device (no CPU fallback)')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=device)

    torch.manual_seed(0)
    model = ConvTasNet().to(device)          # defaults = BASELINE config
    if world > 1:
        broadcast_parameters(model)
        GradSynchronizer(model)
    scaler = torch.amp.GradScaler('cuda', enabled=False)   # bf16: no loss scaling
    batches = make_batches(4, rank, device)

    def step(i):
        batch, lengths = batches[i % len(batches)]
        return model.train_step(batch, lengths, True, scaler)

    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    final_loss = float(loss)

    roof, table = kernel_roofline(model, batches, scaler)
    if world > 1:
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
            'final_loss': final_loss,
            'whole_step_mfma_frac': value/world*FLOP_PER_UTT_TRAIN/(PEAK_MFMA_TFLOPS*1e12),
            'roofline': roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline()
        if args.kernel_table:
            for k, v in table.items():
                print(f'{k:18s} {v["calls_per_step"]:6.1f} calls/step '
                      f'{v["ms_per_step"]:8.3f} ms/step {v["gbs"]:8.1f} GB/s '
                      f'{v["tflops"]:8.1f} TFLOP/s', file=sys.stderr)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
