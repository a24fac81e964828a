"""Evaluation entry point: ``python scripts/test_model.py -i models/<id> -t <test set>``
(reference: scripts/test_model.py:35-317): enhance every mixture of the test set and
score input and output against the clean target with the registered metrics; prints
the improvements (SI-SNRi ...) and writes ``scores.npz`` next to the checkpoint
(h5py is not available here; same ``[mixture, metric, {input, output}]`` layout)."""
import argparse
import logging
import os
import re

import numpy as np
import torch

from _common import ROOT, make_dataset  # noqa: F401

from brever_amd.batching import SortedBatchSampler
from brever_amd.config import get_config
from brever_amd.data import BreverDataLoader
from brever_amd.logger import set_logger
from brever_amd.metrics import MetricRegistry
from brever_amd.models import ModelRegistry


def find_best_checkpoint(ckpt_dir, metric):
    pattern = re.compile(rf'epoch=(\d+)_{re.escape(metric)}=(.+)\.ckpt')
    found = [f for f in os.listdir(ckpt_dir) if pattern.fullmatch(f)]
    if len(found) != 1:
        raise FileNotFoundError(f'expected one best checkpoint for {metric} in '
                                f'{ckpt_dir}, found {found}')
    return os.path.join(ckpt_dir, found[0])


@torch.no_grad()
def test_model(model, cfg, test_spec, metrics, batch_seconds, device):
    dataset = make_dataset(test_spec, cfg.dataset.fs, transform=None, seed=20_000)
    sampler = SortedBatchSampler(dataset, batch_seconds, dynamic=True, shuffle=False,
                                 reverse=True, fs=cfg.dataset.fs)
    loader = BreverDataLoader(dataset, batch_sampler=sampler)
    scores = np.empty((len(dataset), len(metrics), 2))
    for indices, (batch, lengths) in zip(sampler, loader):
        batch, lengths = batch.to(device), lengths.to(device)
        mixture = batch[:, 0]                      # (B, 2, L)
        output = model.enhance(mixture, use_amp=cfg.trainer.use_amp)
        if output.ndim == 3:
            output = output[:, 0]
        target = batch[:, 1].mean(-2)
        noisy = mixture.mean(-2)
        for j, name in enumerate(metrics):
            fn = MetricRegistry.get(name)
            scores[indices, j, 0] = fn(noisy, target, lengths=lengths).cpu().numpy()
            scores[indices, j, 1] = fn(output, target, lengths=lengths).cpu().numpy()
    return scores


def main():
    parser = argparse.ArgumentParser(description='test a model')
    parser.add_argument('-i', '--inputs', nargs='+', required=True)
    parser.add_argument('-t', '--tests', nargs='+', required=True)
    parser.add_argument('--metrics', default='snr,sisnr')
    parser.add_argument('--best', default=None, help='use best checkpoint for metric')
    parser.add_argument('--batch_size', type=float, default=64.0, help='seconds')
    parser.add_argument('--device', default='cuda')
    args = parser.parse_args()
    set_logger()
    metrics = [m for m in args.metrics.split(',') if m]
    for input_ in args.inputs:
        ckpt_dir = os.path.join(input_, 'checkpoints')
        ckpt = os.path.join(ckpt_dir, 'last.ckpt') if args.best is None \
            else find_best_checkpoint(ckpt_dir, args.best)
        cfg = get_config(os.path.join(input_, 'config.yaml'))
        model = ModelRegistry.get(cfg.arch)(**cfg.model.to_dict()).to(args.device)
        state = torch.load(ckpt, map_location=args.device, weights_only=False)
        model.load_state_dict(state['model'])
        model.eval()
        for test in args.tests:
            scores = test_model(model, cfg, test, metrics, args.batch_size, args.device)
            out = os.path.join(input_, 'scores.npz')
            np.savez(out, scores=scores, metrics=np.array(metrics),
                     which=np.array(['input', 'output']), test=np.array(test))
            for j, name in enumerate(metrics):
                imp = (scores[:, j, 1] - scores[:, j, 0]).mean()
                logging.info(f'{input_} on {test}: {name} in '
                             f'{scores[:, j, 0].mean():.2f} out '
                             f'{scores[:, j, 1].mean():.2f} improvement {imp:+.2f} dB')


if __name__ == '__main__':
    main()
