"""Evaluation entry point, same command line as the reference (scripts/test_model.py:35-317):

    python scripts/test_model.py -i models/<id> [...] -t <test set> [...] [--cuda]
        [--metrics pesq stoi estoi snr sisnr] [--best METRIC] [--batch_size 20]
        [--workers 0] [--ddp] [-f] [--no_train_check] [--output_dir DIR]

Every mixture of a test set is enhanced (``model.enhance(mixture, use_amp=cfg.trainer.use_amp)``)
and the input and the output are scored against the clean target with the registered metrics;
the scores go to ``<model>/scores.hdf5`` under ``<checkpoint>/<test set>`` as a
``[mixture, metric, {input, output}]`` array with the reference's dimension scales -- through ``h5py``
when importable, else through the HDF5 C library (``brever_amd/h5lite.py``); only without both to
``<model>/scores.npz`` with the same keys (``metrics``, ``which``,
``<checkpoint>/<test set>``). With ``--ddp`` (one process per GPU) the batches of the sorted
sampler are sharded over the ranks and gathered on rank 0 with ``gather_object``. A test set may
be ``synthetic:<items>:<seconds>``. The HIP models have no CPU path: pass ``--cuda``.
``--output_dir`` writes ``NNNNN_{input,output}.flac`` as the reference does (mono 16-bit FLAC through the native
encoder ``brv_flac_encode16`` instead of ``torchaudio.save``; ``BRV_OUTPUT_WAV=1``: 32-bit float WAV)."""
import argparse
import logging
import os

# read when the HIP runtime loads (torch import): see brever_amd/__init__.py
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import pprint
import re
import struct

import numpy as np
import torch
import torch.distributed as dist

from _common import ROOT, is_synthetic, make_dataset  # noqa: F401

from brever_amd.batching import DistributedBatchSamplerWrapper, SortedBatchSampler
from brever_amd.config import get_config
from brever_amd.data import BreverDataLoader, write_flac
from brever_amd.inspect import Path
from brever_amd.logger import set_logger
from brever_amd.metrics import MetricRegistry
from brever_amd.models import ModelRegistry

from brever_amd import h5lite

try:
    import h5py
except ImportError:
    h5py = None


def find_best_checkpoint(ckpt_dir, metric):
    """Checkpoint file whose name carries the highest ``_<metric>=<value>``
    (scripts/test_model.py:266-277)."""
    pattern = re.compile(rf'^.*?_{re.escape(metric)}=(-?\d+\.\d+(?:e(?:\+|-)\d+)?).*?\.ckpt$')
    scored = [(float(m.group(1)), os.path.join(ckpt_dir, m.group(0)))
              for m in map(pattern.match, os.listdir(ckpt_dir)) if m]
    if not scored:
        raise FileNotFoundError(f'no checkpoint with a {metric} score in {ckpt_dir}')
    return max(scored)[1]


def write_wav(path, x, fs):
    """Mono 32-bit float WAV."""
    data = np.asarray(x, dtype='<f4').tobytes()
    with open(path, 'wb') as f:
        f.write(b'RIFF' + struct.pack('<I', 36 + len(data)) + b'WAVE')
        f.write(b'fmt ' + struct.pack('<IHHIIHH', 16, 3, 1, fs, fs*4, 4, 32))
        f.write(b'data' + struct.pack('<I', len(data)) + data)


class ScoreFile:
    """``scores.hdf5`` with the reference's layout (scripts/test_model.py:245-263): through ``h5py`` when
    it is importable, else through the HDF5 C library itself (``brever_amd.h5lite``: the same H5D / H5DS
    calls ``h5py`` makes); only when neither is there, ``scores.npz`` with the same keys."""

    def __init__(self, model_dir):
        self.backend = 'h5py' if h5py is not None else ('h5lite' if h5lite.available() else 'npz')
        self.path = os.path.join(model_dir, 'scores.npz' if self.backend == 'npz' else 'scores.hdf5')

    def _load_npz(self):
        if not os.path.exists(self.path):
            return {}
        with np.load(self.path, allow_pickle=False) as f:
            return {k: f[k] for k in f.files}

    def contains(self, key):
        if not os.path.exists(self.path):
            return False
        if self.backend == 'h5py':
            with h5py.File(self.path, 'r') as f:
                return key in f
        if self.backend == 'h5lite':
            with h5lite.File(self.path, 'r') as f:
                return key in f
        return key in self._load_npz()

    def write(self, key, scores, metrics):
        if self.backend == 'npz':
            content = self._load_npz()
            content.update({key: scores, 'metrics': np.array(metrics),
                            'which': np.array(['input', 'output'])})
            np.savez(self.path, **content)
            return
        new = not os.path.exists(self.path)
        labels = ('mixture', 'metric', 'which')
        if self.backend == 'h5lite':
            with h5lite.File(self.path, 'w' if new else 'a') as f:
                if new:
                    f.write_strings('metrics', list(metrics))
                    f.write_strings('which', ['input', 'output'])
                f.write_array(key, scores)
                f.set_dims(key, labels, {1: 'metrics', 2: 'which'})
            return
        with h5py.File(self.path, 'w' if new else 'a') as f:
            if new:
                f['metrics'] = list(metrics)
                f['which'] = ['input', 'output']
            if key in f:
                f[key][...] = scores
                dset = f[key]
            else:
                dset = f.create_dataset(key, data=scores)
            for axis, label in enumerate(labels):
                dset.dims[axis].label = label
            dset.dims[1].attach_scale(f['metrics'])
            dset.dims[2].attach_scale(f['which'])


@torch.no_grad()
def test_model(i_test, model, cfg, test_path, scores, checkpoint_path, rank, device, args):
    progress = f'[{i_test}/{len(args.tests)}]'
    key = f'{os.path.basename(checkpoint_path)}/' \
          f'{os.path.basename(os.path.normpath(test_path))}'
    if scores.contains(key) and not args.force:
        if rank == 0:
            logging.info(f'Model already tested on {test_path} {progress}')
        return
    if rank == 0:
        logging.info(f'Evaluating on {test_path} {progress}')

    dataset = make_dataset(test_path, cfg.dataset.fs, transform=None, seed=20_000,
                           **({} if is_synthetic(test_path)
                              else dict(segment_length=0.0, sources=cfg.dataset.sources)))
    # sorted by decreasing length: least padding per batch (test_model.py:131-140)
    sampler = SortedBatchSampler(dataset, batch_size=args.batch_size, shuffle=False,
                                 dynamic=True, reverse=True, fs=cfg.dataset.fs)
    if dist.is_initialized():
        sampler = DistributedBatchSamplerWrapper(sampler)
    loader = BreverDataLoader(dataset=dataset, num_workers=args.workers, batch_sampler=sampler)

    n_mix = sum(len(b) for b in sampler) if dist.is_initialized() else len(dataset)
    dset_scores = np.empty((n_mix, len(args.metrics), 2))
    avg_delta = {metric: 0.0 for metric in args.metrics}
    i_mix = 0
    for batch, lengths in loader:
        batch, lengths = batch.to(device), lengths.to(device)
        input_, target = batch[:, 0], batch[:, 1:]
        output = model.enhance(input_, use_amp=cfg.trainer.use_amp)
        if output.ndim == 3:
            output = output[:, 0]              # first separated source (test_model.py:176-178)
        target = target[:, 0].mean(-2)         # first target source, left / right averaged
        input_ = input_.mean(-2)
        n = batch.shape[0]
        for j, metric in enumerate(args.metrics):
            fn = MetricRegistry.get(metric)
            in_score = np.asarray(torch.as_tensor(fn(input_, target, lengths=lengths)).cpu(), dtype=float)
            out_score = np.asarray(torch.as_tensor(fn(output, target, lengths=lengths)).cpu(), dtype=float)
            dset_scores[i_mix:i_mix + n, j, 0] = in_score
            dset_scores[i_mix:i_mix + n, j, 1] = out_score
            avg_delta[metric] = (float((out_score - in_score).sum()) + avg_delta[metric]*i_mix)/(i_mix + n)
        if args.output_dir is not None:
            os.makedirs(args.output_dir, exist_ok=True)
            for x, name in ((input_, 'input'), (output, 'output')):
                for i in range(n):
                    sig = x[i, :int(lengths[i])].float().cpu().numpy()
                    stem = os.path.join(args.output_dir, f'{i_mix + i:05d}_{name}')
                    if os.environ.get('BRV_OUTPUT_WAV') == '1':
                        write_wav(stem + '.wav', sig, cfg.dataset.fs)
                    else:                      # scripts/test_model.py:201-209 of the reference
                        write_flac(stem + '.flac', sig, cfg.dataset.fs)
        i_mix += n

    if dist.is_initialized():
        world = dist.get_world_size()
        score_parts = [None]*world if rank == 0 else None
        delta_parts = [None]*world if rank == 0 else None
        dist.gather_object(dset_scores, object_gather_list=score_parts)
        dist.gather_object(avg_delta, object_gather_list=delta_parts)
        if rank == 0:
            dset_scores = np.concatenate(score_parts, axis=0)
            avg_delta = {m: sum(d[m] for d in delta_parts)/world for m in args.metrics}
    if rank == 0:
        logging.info('Average delta scores:')
        for metric, delta in avg_delta.items():
            logging.info(f'{metric}: {delta:.2e}')
        scores.write(key, dset_scores, args.metrics)


def main(i_input, input_, args):
    progress = f'[{i_input}/{len(args.inputs)}]'
    if not os.path.exists(input_):
        print(f'Model {input_} does not exist {progress}')
        return
    if input_.endswith('.ckpt'):
        model_dir = os.path.dirname(os.path.dirname(input_))
        checkpoint_path = input_
    else:
        model_dir = input_
        checkpoint_path = os.path.join(input_, 'checkpoints', 'last.ckpt')
    if args.best is not None:
        checkpoint_path = find_best_checkpoint(os.path.join(model_dir, 'checkpoints'), args.best)
    if not os.path.exists(os.path.join(model_dir, 'losses.npz')) and not args.no_train_check:
        print(f'Model {input_} is not trained {progress}')
        return
    cfg = get_config(os.path.join(model_dir, 'config.yaml'))

    rank = 0
    device = 'cuda' if args.cuda else 'cpu'
    if args.ddp and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        from brever_amd.parallel import init_process_group
        init_process_group('nccl' if args.cuda else 'gloo')
    if dist.is_initialized():
        rank = dist.get_rank()
        if args.cuda:
            device = rank % torch.cuda.device_count()
            torch.cuda.set_device(device)

    set_logger(os.path.join(model_dir, 'log_test.log'), args.ddp, rank)
    if rank == 0:
        logging.info(f'Testing {checkpoint_path} {progress}')
        logging.debug(f'Configuration: \n {pprint.pformat(cfg.to_dict())}')

    model = ModelRegistry.get(cfg.arch)(**cfg.model.to_dict()).to(device)
    map_location = f'cuda:{device}' if isinstance(device, int) else device
    state = torch.load(checkpoint_path, map_location=map_location, weights_only=False)
    model.load_state_dict(state['model'])
    if 'ema' in state:
        from brever_amd.training import ExponentialMovingAverage
        ema = ExponentialMovingAverage(model.parameters(), decay=cfg.trainer.ema_decay)
        ema.load_state_dict(state['ema'])
        ema.copy_to()
        if hasattr(model, 'mark_params_changed'):
            model.mark_params_changed()
    torch.set_grad_enabled(False)
    model.eval()

    scores = ScoreFile(model_dir)
    for i, test_path in enumerate(args.tests):
        test_model(i, model, cfg, test_path, scores, checkpoint_path, rank, device, args)
        if dist.is_initialized():
            dist.barrier()


if __name__ == '__main__':
    parser = argparse.ArgumentParser(description='test a model')
    parser.add_argument('-i', '--inputs', nargs='+', required=True,
                        help='model directories or checkpoints')
    parser.add_argument('-t', '--tests', type=Path, nargs='+', required=True,
                        help='test dataset paths')
    parser.add_argument('-f', '--force', action='store_true', help='test even if already tested')
    parser.add_argument('--output_dir', help='where to write signals')
    parser.add_argument('--cuda', action='store_true', help='run on GPU')
    parser.add_argument('--metrics', nargs='+',
                        default=['pesq', 'stoi', 'estoi', 'snr', 'sisnr'],
                        help='metrics to evaluate with')
    parser.add_argument('--no_train_check', action='store_true',
                        help='test even if model is not trained')
    parser.add_argument('--best', help='metric to use for checkpoint selection')
    parser.add_argument('--batch_size', type=int, default=20, help='batch size')
    parser.add_argument('--workers', type=int, default=0, help='number of workers')
    parser.add_argument('--ddp', action='store_true', help='use DDP')
    cli = parser.parse_args()
    if cli.output_dir is not None and cli.ddp:
        raise ValueError('cannot use DDP with output_dir')
    for i_in, model_input in enumerate(cli.inputs):
        main(i_in, model_input, cli)
    if dist.is_initialized():
        dist.destroy_process_group()
