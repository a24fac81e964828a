"""ORACLE (test infrastructure, not product code).

CPU restatement in plain PyTorch of the reference's feed-forward mask model,
brever/models/ffnn/ffnn.py:15-203, with the pieces of brever/modules it calls:
STFT.forward/backward (modules/stft.py:59-138), MelFilterbank (:152-198) and the
``logfbe`` feature (modules/features.py:142-199). Only ``tests/`` import this module.

Pinning: tests/golden/ffnn.npz, produced from the imported reference by
tests/golden/make_golden.py (transform output, forward/loss/gradients at fixed weights
with dropout 0, enhance output, parameter count).
"""
import math

import numpy as np
import scipy.signal
import torch
import torch.nn as nn

FEPS = torch.finfo().eps          # features.py:10
DEPS = np.finfo(float).eps        # ffnn.py:12


def _stft(x, n=512, hop=256):
    """STFT(frame_length=n, hop_length=hop, window='hann', normalized=True)."""
    w = torch.from_numpy(scipy.signal.get_window('hann', n)).to(x.dtype)
    L = x.shape[-1]
    frames = math.ceil(max(L - n, 0)/hop) + 1
    x = torch.nn.functional.pad(x, (0, (frames - 1)*hop + n - L))
    lead = x.shape[:-1]
    X = torch.stft(x.reshape(-1, x.shape[-1]), n_fft=n, hop_length=hop, window=w, center=True,
                   pad_mode='constant', normalized=False, onesided=True, return_complex=True)
    X = X/w.pow(2).sum().sqrt()                                   # stft.py:82-83
    return X.view(*lead, *X.shape[-2:])


def _istft(X, n=512, hop=256):
    w = torch.from_numpy(scipy.signal.get_window('hann', n)).to(X.real.dtype)
    X = X*w.pow(2).sum().sqrt()                                   # stft.py:121-122
    lead = X.shape[:-2]
    y = torch.istft(X.reshape(-1, *X.shape[-2:]), n_fft=n, hop_length=hop, window=w,
                    center=True, normalized=False, onesided=True)
    return y.view(*lead, -1)


def mel_filters(n_filters=64, n_fft=512, fs=16e3, fmin=50, fmax=8000):
    """MelFilterbank.calc_filterbank (stft.py:163-178)."""
    to_mel = lambda f: 2595*math.log10(1 + f/700)                 # noqa: E731
    mel = torch.linspace(to_mel(fmin), to_mel(fmax), n_filters + 2)
    fc = 700*(10**(mel/2595) - 1)
    f = torch.arange(n_fft//2 + 1).float()*fs/n_fft
    filters = torch.zeros((n_filters, len(f)))
    for i in range(1, n_filters + 1):
        rise = (fc[i - 1] <= f) & (f <= fc[i])
        filters[i - 1, rise] = (f[rise] - fc[i - 1])/(fc[i] - fc[i - 1])
        fall = (fc[i] <= f) & (f <= fc[i + 1])
        filters[i - 1, fall] = (fc[i + 1] - f[fall])/(fc[i + 1] - fc[i])
    scaling = filters.sum(axis=1, keepdims=True)
    return filters/scaling, scaling


class OracleFFNN(nn.Module):
    def __init__(self, stacks=5, decimation=1, n=512, hop=256, mel=64, hidden_layers=(1024, 1024),
                 dropout=0.2):
        super().__init__()
        self.stacks, self.decimation, self.n, self.hop = stacks, decimation, n, hop
        self.filters, self.scaling = mel_filters(mel, n)
        self.input_size = mel*(stacks + 1)
        layers, start = [], self.input_size
        for end in hidden_layers:                                 # ffnn.py:158-164
            layers += [nn.Linear(start, end), nn.ReLU(), nn.Dropout(dropout)]
            start = end
        layers += [nn.Linear(start, mel), nn.Sigmoid()]
        self.module_list = nn.ModuleList(layers)
        self.register_buffer('mean', torch.zeros(self.input_size, 1))
        self.register_buffer('std', torch.ones(self.input_size, 1))

    def logfbe(self, X):                                          # features.py:185-195
        out = X.abs().pow(2).mean(-3)
        return torch.log(self.filters @ out + FEPS)

    def stack(self, data):                                        # ffnn.py:130-140
        out = [data]
        for i in range(self.stacks):
            rolled = data.roll(i + 1, -1)
            rolled[..., :i + 1] = data[..., :1]
            out.append(rolled)
        return torch.cat(out, dim=0 if data.ndim == 2 else 1)

    def transform(self, sources):                                 # ffnn.py:77-92
        X = _stft(sources, self.n, self.hop)
        mix, fg = X
        bg = mix - fg
        x = self.stack(self.logfbe(mix))[..., ::self.decimation]
        fgp = self.filters @ fg.abs().pow(2).mean(0)
        bgp = self.filters @ bg.abs().pow(2).mean(0)
        irm = (1 + bgp/(fgp + DEPS)).pow(-0.5)
        return torch.cat([x, irm[..., ::self.decimation]])

    def forward(self, x):                                         # ffnn.py:72-75,166-171
        x = ((x - self.mean)/self.std).transpose(1, 2)
        for m in self.module_list:
            x = m(x)
        return x.transpose(1, 2)

    def loss(self, batch, lengths):                               # ffnn.py:94-99, mse
        from .criterion import mse
        out = self(batch[:, :self.input_size])
        return mse(out, batch[:, self.input_size:], lengths).mean()

    def enhance(self, x):                                         # ffnn.py:101-114
        length = x.shape[-1]
        X = _stft(x, self.n, self.hop)
        feats = self.stack(self.logfbe(X))
        mask = self(feats)
        mask_ext = (self.filters*self.scaling).T @ mask           # stft.py:196-198
        y = _istft(X.mean(1)*mask_ext, self.n, self.hop)
        return y[..., :length]
