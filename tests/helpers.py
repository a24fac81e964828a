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
        try:
            import h5py
            with h5py.File(os.path.join(model_dir, name), 'r') as f:
                scores = f[key][...]
        except ImportError:
            from brever_amd import h5lite
            with h5lite.File(os.path.join(model_dir, name), 'r') as f:
                assert f.read_strings('metrics') == list(metrics)
                assert f.read_strings('which') == ['input', 'output']
                scores = f.read_array(key)
    return model_dir, losses, scores


# ---- a small FLAC ENCODER (test infrastructure only) ---------------------------------------------
# Writes RFC 9639 streams that exercise the decoder of csrc/flac.hip: fixed predictors of every
# order, LPC with quantised coefficients, verbatim and constant subframes, wasted bits,
# left/side / side/right / mid/side stereo, partitioned Rice residuals with both parameter
# widths and escape partitions, fixed- and variable-length last blocks.
class _BitWriter:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def write(self, value, bits):
        if bits == 0:
            return
        self.acc = (self.acc << bits) | (value & ((1 << bits) - 1))
        self.n += bits
        while self.n >= 8:
            self.n -= 8
            self.out.append((self.acc >> self.n) & 0xff)
        self.acc &= (1 << self.n) - 1

    def unary(self, q):
        while q >= 32:
            self.write(0, 32)
            q -= 32
        self.write(1, q + 1)

    def align(self):
        if self.n:
            self.write(0, 8 - self.n)


def _crc(data, poly, width):
    top, mask, c = 1 << (width - 1), (1 << width) - 1, 0
    for b in data:
        c ^= b << (width - 8)
        for _ in range(8):
            c = ((c << 1) ^ poly) & mask if c & top else (c << 1) & mask
    return c


def _flac_residual(bw, res, blocksize, order, porder, five_bit, escape):
    bw.write(1 if five_bit else 0, 2)
    bw.write(porder, 4)
    pbits = 5 if five_bit else 4
    pos = 0
    for part in range(1 << porder):
        count = blocksize - order if porder == 0 else (blocksize >> porder) - (order if part == 0 else 0)
        chunk = res[pos:pos + count]
        pos += count
        if escape and part % 2 == 1:
            nb = max([1] + [int(v).bit_length() + 1 for v in chunk])
            bw.write((1 << pbits) - 1, pbits)
            bw.write(nb, 5)
            for v in chunk:
                bw.write(int(v), nb)
            continue
        mean = sum(abs(int(v)) for v in chunk)/max(len(chunk), 1)
        k = min(max(int(mean).bit_length(), 0), (1 << pbits) - 2)
        bw.write(k, pbits)
        for v in chunk:
            v = int(v)
            u = (v << 1) if v >= 0 else ((-v << 1) - 1)
            bw.unary(u >> k)
            bw.write(u & ((1 << k) - 1), k)


def _flac_subframe(bw, x, bps, kind, order, porder, five_bit, escape):
    x = [int(v) for v in x]
    n = len(x)
    wasted = 0
    if kind != 'constant' and any(x):
        while all(v % (1 << (wasted + 1)) == 0 for v in x) and wasted < bps - 1:
            wasted += 1
    if wasted:
        x = [v >> wasted for v in x]
        bps -= wasted
    if kind == 'constant':
        assert len(set(x)) == 1
        bw.write(0, 8)
        bw.write(x[0], bps)
        return
    type_bits = {'verbatim': 1, 'fixed': 8 + order, 'lpc': 31 + order}[kind]
    bw.write(0, 1); bw.write(type_bits, 6); bw.write(1 if wasted else 0, 1)
    if wasted:
        bw.unary(wasted - 1)
    if kind == 'verbatim':
        for v in x:
            bw.write(v, bps)
        return
    for v in x[:order]:
        bw.write(v, bps)
    if kind == 'fixed':
        coef = [[], [1], [2, -1], [3, -3, 1], [4, -6, 4, -1]][order]
        shift = 0
    else:
        import numpy as np
        a = np.asarray(x, dtype=float)
        # least-squares predictor, quantised to 12 bits
        rows = np.stack([a[order - 1 - j:n - 1 - j] for j in range(order)], axis=1)
        sol = np.linalg.lstsq(rows, a[order:], rcond=None)[0] if n > 2*order else np.zeros(order)
        precision, shift = 12, 9
        coef = [int(max(-2048, min(2047, round(c*(1 << shift))))) for c in sol]
        bw.write(precision - 1, 4)
        bw.write(shift, 5)
        for c in coef:
            bw.write(c, precision)
    res = []
    for i in range(order, n):
        pred = sum(c*x[i - 1 - j] for j, c in enumerate(coef)) >> shift
        res.append(x[i] - pred)
    _flac_residual(bw, res, n, order, porder, five_bit, escape)


def flac_encode(pcm, rate=16000, bps=16, blocksize=1024, kind='fixed', order=2, stereo='independent',
                porder=2, five_bit=False, escape=False):
    """``pcm``: int array (frames, channels). Returns the bytes of a FLAC stream."""
    import numpy as np
    pcm = np.asarray(pcm, dtype=np.int64)
    if pcm.ndim == 1:
        pcm = pcm[:, None]
    frames, C = pcm.shape
    out = bytearray(b'fLaC')
    info = _BitWriter()
    info.write(blocksize, 16); info.write(blocksize, 16); info.write(0, 24); info.write(0, 24)
    info.write(rate, 20); info.write(C - 1, 3); info.write(bps - 1, 5); info.write(frames, 36)
    info.write(0, 128)
    out += bytes([0x80, 0, 0, 34]) + bytes(info.out)
    ch_code = {'independent': C - 1, 'left_side': 8, 'side_right': 9, 'mid_side': 10}[stereo]
    for fno, start in enumerate(range(0, frames, blocksize)):
        block = pcm[start:start + blocksize]
        n = len(block)
        bw = _BitWriter()
        bw.write(0xfff8 >> 2, 14); bw.write(0, 1); bw.write(0, 1)
        bw.write(7, 4)                                   # 16-bit (blocksize - 1) follows
        bw.write(0, 4)                                   # sample rate from STREAMINFO
        bw.write(ch_code, 4)
        bw.write({8: 1, 12: 2, 16: 4, 20: 5, 24: 6}[bps], 3); bw.write(0, 1)
        assert fno < 0x800
        if fno < 0x80:
            bw.write(fno, 8)
        else:
            bw.write(0xc0 | (fno >> 6), 8); bw.write(0x80 | (fno & 0x3f), 8)
        bw.write(n - 1, 16)
        bw.write(_crc(bytes(bw.out), 0x07, 8), 8)
        chans = [block[:, c] for c in range(C)]
        widths = [bps]*C
        if stereo != 'independent':
            left, right = chans
            side = left - right
            if stereo == 'left_side':
                chans, widths = [left, side], [bps, bps + 1]
            elif stereo == 'side_right':
                chans, widths = [side, right], [bps + 1, bps]
            else:
                chans, widths = [(left + right) >> 1, side], [bps, bps + 1]
        po = porder
        while po > 0 and (n % (1 << po) or (n >> po) <= order):
            po -= 1
        for x, w in zip(chans, widths):
            k = kind
            if k != 'verbatim' and len(set(int(v) for v in x)) == 1:
                k = 'constant'
            elif k in ('fixed', 'lpc') and n <= order:
                k = 'verbatim'
            _flac_subframe(bw, x, w, k, order, po, five_bit, escape)
        bw.align()
        bw.write(_crc(bytes(bw.out), 0x8005, 16), 16)
        out += bytes(bw.out)
    return bytes(out)
