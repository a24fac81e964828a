"""Test doubles shared by the CPU and GPU suites (mirrors the seam the reference
tests use: tests/utils.py:9-86 -- an in-memory dataset and a 1x1-conv model)."""
import random

import torch

from brever_amd.models.base import BreverBaseModel
from oracle import criterion as ref_criterion


class DummyDataset(torch.utils.data.Dataset):
    """Seeded random-length items; same construction recipe (hence the same
    lengths and samples) as the reference's test dataset."""

    def __init__(self, n_examples, n_sources, n_channels, min_length, max_length,
                 transform=None):
        rng = random.Random(42)
        self._segment_info = [(i, (0, rng.randint(min_length, max_length)))
                              for i in range(n_examples)]
        g = torch.Generator().manual_seed(42)
        self.sources = [
            torch.randn((n_sources, n_channels, self._segment_info[i][1][1]),
                        generator=g)
            for i in range(n_examples)
        ]
        self.n_examples = n_examples
        self.transform = transform
        self.preloaded_data = None
        self._duration = sum(x[1][1] for x in self._segment_info)/16000
        self._effective_duration = self._duration
        self.segment_strategy = 'pass'
        self.rmm_dset = None

    def __getitem__(self, index):
        if self.preloaded_data is not None:
            return self.preloaded_data[index]
        item = self.sources[index]
        if self.transform is not None:
            item = self.transform(item)
        return item

    def __len__(self):
        return self.n_examples

    def get_segment_length(self, i):
        return self._segment_info[i][1][1]

    def get_max_segment_length(self):
        return max(end - start for _, (start, end) in self._segment_info)

    def preload(self, device, tqdm_desc=None):
        self.preloaded_data = [self[i].to(device) for i in range(len(self))]

    def set_epoch(self, epoch):
        pass


class DummyModel(BreverBaseModel):
    """1x1 conv over channels, oracle SNR criterion (CPU-capable)."""

    def __init__(self, channels=2, output_sources=1, criterion='snr'):
        super().__init__(criterion=ref_criterion.CRITERIA[criterion])
        self.conv = torch.nn.Conv1d(channels, channels*output_sources, 1)
        self.output_sources = output_sources
        self.channels = channels
        self.optimizer = torch.optim.Adam(self.parameters())

    def forward(self, x):
        x = self.conv(x)
        return x.reshape(x.shape[0], self.output_sources, self.channels,
                         x.shape[-1])

    def loss(self, batch, lengths, use_amp):
        inputs, labels = batch[:, 0], batch[:, 1:]
        return self.criterion(self(inputs), labels, lengths).mean()

    def _enhance(self, x, use_amp):
        return x.mean(-2)


def sgmse_case(golden, tag):
    """Oracle network, SDE, sampler and replayed noise for one case of tests/golden/sgmse.npz;
    the weights come from the seeded construction the golden script used."""
    import functools
    import json

    import scipy.signal

    from brever_amd.models import ModelRegistry
    from oracle import sgmse as osg
    cfg = json.loads(str(golden[f'{tag}_config']))
    torch.manual_seed(3)
    model = ModelRegistry.get(str(golden[f'{tag}_arch']))(**cfg)
    gen = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if 'norm' in name:
                p.add_(0.1*torch.randn(p.shape, generator=gen))
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    assert torch.equal(flat, torch.from_numpy(golden[f'{tag}_params']))
    edm = tag == 'edm'
    net = osg.Net(model.state_dict(), 'model.net.', skip_scale=0.5**0.5,
                  block_type='adm' if edm else 'ncsn')
    sde = osg.OUCosine() if edm else osg.RichterOUVE()
    kw = dict(precond='edm' if edm else 'richter')
    draws = [torch.from_numpy(golden[f'{tag}_noise_{i}'])
             for i in range(int(golden[f'{tag}_n_noise']))]

    def noise(shape, complex_, it=iter(draws)):
        d = next(it)
        assert tuple(d.shape) == tuple(shape) and d.is_complex() == complex_
        return d
    if edm:
        sampler = functools.partial(osg.edm_sample, noise=noise, num_steps=3, schurn=1.0,
                                    smin=0.0, smax=float('inf'), snoise=1.0, **kw)
    else:
        sampler = functools.partial(osg.pc_sample, noise=noise,
                                    num_steps=cfg['solver_num_steps'], corrector_steps=1,
                                    corrector_snr=0.5, **kw)
    window = scipy.signal.get_window('hann', cfg['stft_frame_length'])
    return model, net, sde, kw, sampler, window, draws


def run_entry_points(tmp_path, arch, model_args=(), trainer_args=(), train='synthetic:8:0.5',
                     val='synthetic:4:0.5', test='synthetic:3:0.5', metrics=('snr', 'sisnr'),
                     extra_test_args=()):
    """init_model.py -> train_model.py -> test_model.py with the reference's command lines
    (dataset / trainer options before the architecture, model options after it; ``--cuda``,
    ``--metrics a b``). Returns (model_dir, losses npz, scores array [mixture, metric, 2])."""
    import os
    import subprocess
    import sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(*a):
        return subprocess.run([sys.executable, *a], capture_output=True, text=True, cwd=root)

    models = os.path.join(str(tmp_path), 'models')
    os.makedirs(models, exist_ok=True)
    out = run('scripts/init_model.py', '--train_path', str(train), '--val_path', str(val),
              '--preload', 'true', '--workers', '0', *trainer_args, '--models_dir', models,
              arch, *model_args)
    assert out.returncode == 0, out.stderr[-3000:]
    model_dir = os.path.join(models, os.listdir(models)[0])
    out = run('scripts/train_model.py', model_dir)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    losses = np.load(os.path.join(model_dir, 'losses.npz'))
    out2 = run('scripts/train_model.py', model_dir)
    assert out2.returncode != 0 and 'training already done' in out2.stderr
    out = run('scripts/test_model.py', '-i', model_dir, '-t', str(test), '--cuda',
              '--metrics', *metrics, *extra_test_args)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    name = 'scores.hdf5' if os.path.exists(os.path.join(model_dir, 'scores.hdf5')) else 'scores.npz'
    key = 'last.ckpt/' + os.path.basename(os.path.normpath(str(test)))
    if name.endswith('.npz'):
        f = np.load(os.path.join(model_dir, name))
        assert list(f['metrics']) == list(metrics) and list(f['which']) == ['input', 'output']
        scores = f[key]
    else:
        import h5py
        with h5py.File(os.path.join(model_dir, name), 'r') as f:
            scores = f[key][...]
    return model_dir, losses, scores
