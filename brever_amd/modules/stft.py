"""STFT / inverse STFT / mel filterbank on the HIP path.

Same constructor arguments, padding arithmetic, normalisation, magnitude
compression, scaling and return types as the reference wrappers around
``torch.stft`` / ``torch.istft`` (brever/modules/stft.py:12-198, contract in
SURVEY.md App. A.1/A.2):

* ``frames = ceil(max(L - N, 0)/H) + 1`` non-centred frames, right zero padding to
  whole frames, plus ``N/2`` zeros on both sides (``center=True``, constant mode);
* ``X /= sqrt(sum w^2)`` iff ``normalized``; ``X <- abs(X)^c e^{j angle X}``;
  ``X *= scale``; the inverse undoes these in reverse order and runs the
  window-envelope-normalised overlap-add of ``torch.istft``.

The transforms run in ``libbrever_hip.so`` as fp32 DFT-GEMMs on the exact-fp32 MFMA
(``brv_stft_forward`` / ``brv_istft_backward``); the window-weighted DFT bases are
built here once per instance in float64 and cached per device. Limits of the HIP
path for now: one-sided spectra, ``center=True``, constant padding, hop dividing the
frame length, ``n_fft == frame_length``. ``STFT.forward`` is differentiable when
``compression_factor == 1`` (``brv_stft_adjoint``: the transposed DFT-GEMM followed by
a plain overlap-add); ``STFT.backward`` and the filterbank give values only.
"""
import functools
import math

import numpy as np
import scipy.signal
import torch

from .. import hip


def fft_freqs(fs=16e3, n_fft=512, onesided=True):
    """FFT bin frequencies (brever/utils.py:40-66)."""
    freqs = np.arange(n_fft)*fs/n_fft
    mask = freqs > fs/2
    if onesided:
        return freqs[~mask]
    freqs[mask] = freqs[mask] - fs
    return freqs


class _StftFunction(torch.autograd.Function):
    """x (rows, L) fp32 -> spec (rows, bins, F) complex64; gradient through
    ``brv_stft_adjoint`` (compression 1 only)."""

    @staticmethod
    def forward(ctx, x2, basis, frame_length, hop_length, compression, scale):
        lib = hip.lib()
        rows, L = x2.shape
        F = lib.brv_stft_frames(L, frame_length, hop_length)
        bins = frame_length//2 + 1
        spec = torch.empty(rows, bins, F, 2, dtype=torch.float32, device=x2.device)
        hip.check(lib.brv_stft_forward(
            hip.ptr(x2), hip.ptr(basis), hip.ptr(spec), rows, L, frame_length,
            hop_length, float(compression), float(scale), hip.stream()), 'brv_stft_forward')
        ctx.save_for_backward(basis)
        ctx.geom = (rows, L, frame_length, hop_length, F, float(compression), float(scale))
        return torch.view_as_complex(spec)

    @staticmethod
    def backward(ctx, grad):
        basis, = ctx.saved_tensors
        rows, L, n, hop, F, compression, scale = ctx.geom
        if compression != 1:
            raise NotImplementedError('gradient of the magnitude compression is not built yet '
                                      'on the HIP path')
        dspec = torch.view_as_real(grad.to(torch.complex64).contiguous())
        dx = stft_adjoint(dspec, basis, rows, L, n, hop, F, scale)
        return dx, None, None, None, None, None


class _IstftFunction(torch.autograd.Function):
    """spec (rows, bins, F) complex64 -> y (rows, hop*(F-1)): ``brv_istft_backward``; the
    gradient (compression 1 only) is the window-envelope division followed by a framed DFT
    with the transposed inverse basis."""

    @staticmethod
    def forward(ctx, spec, inv, win, inv_t, frame_length, hop_length, compression, scale):
        rows, bins, F = spec.shape
        spec_r = torch.view_as_real(spec.contiguous())
        scratch = torch.empty(rows, F, frame_length, dtype=torch.float32, device=spec.device)
        out_len = hop_length*(F - 1)
        y = torch.empty(rows, out_len, dtype=torch.float32, device=spec.device)
        hip.check(hip.lib().brv_istft_backward(
            hip.ptr(spec_r), hip.ptr(inv), hip.ptr(win), hip.ptr(scratch), hip.ptr(y), rows, F,
            frame_length, hop_length, float(compression), float(scale), hip.stream()),
            'brv_istft_backward')
        ctx.save_for_backward(win, inv_t)
        ctx.geom = (rows, bins, F, frame_length, hop_length, float(compression), float(scale))
        return y

    @staticmethod
    def backward(ctx, dy):
        win, inv_t = ctx.saved_tensors
        rows, bins, F, n, hop, compression, scale = ctx.geom
        if compression != 1:
            raise NotImplementedError('gradient of the magnitude compression is not built yet '
                                      'on the HIP path')
        lib = hip.lib()
        dy = dy.float().contiguous()
        out_len = dy.shape[-1]
        u = torch.empty_like(dy)
        hip.check(lib.brv_istft_env_divide(hip.ptr(dy), hip.ptr(win), hip.ptr(u), rows, out_len,
                                           n, hop, F, hip.stream()), 'brv_istft_env_divide')
        dspec = torch.empty(rows, bins, F, 2, dtype=torch.float32, device=dy.device)
        hip.check(lib.brv_framed_dft_forward(
            hip.ptr(u), hip.ptr(inv_t), hip.ptr(dspec), rows, out_len, n, hop, n//2, F, 1.0,
            1.0/scale, hip.stream()), 'brv_framed_dft_forward')
        return torch.view_as_complex(dspec), None, None, None, None, None, None, None


def stft_adjoint(dspec, basis, rows, L, frame_length, hop_length, frames, scale):
    """dx (rows, L) = adjoint of the framed DFT applied to dspec (rows, bins, F, 2)."""
    scratch = torch.empty(rows, frames, frame_length, dtype=torch.float32, device=dspec.device)
    dx = torch.empty(rows, L, dtype=torch.float32, device=dspec.device)
    hip.check(hip.lib().brv_stft_adjoint(
        hip.ptr(dspec), hip.ptr(basis), hip.ptr(scratch), hip.ptr(dx), rows, L,
        frame_length, hop_length, float(scale), hip.stream()), 'brv_stft_adjoint')
    return dx


class STFT:
    def __init__(self, frame_length=512, hop_length=256, window='hann',
                 center=True, pad_mode='constant', normalized=True,
                 onesided=True, compression_factor=1, scale_factor=1,
                 n_fft=None):
        self.frame_length = frame_length
        self.hop_length = hop_length
        self.center = center
        self.pad_mode = pad_mode
        self.normalized = normalized
        self.onesided = onesided
        self.compression_factor = compression_factor
        self.scale_factor = scale_factor
        self.n_fft = frame_length if n_fft is None else n_fft
        if window is None:
            window = 'boxcar'
        if isinstance(window, str):
            window = functools.partial(scipy.signal.get_window, window)
        if callable(window):
            window = window(frame_length)
        if isinstance(window, np.ndarray):
            window = torch.from_numpy(window)
        self.window = window
        unsupported = []
        if not center:
            unsupported.append('center=False')
        if pad_mode != 'constant':
            unsupported.append(f"pad_mode='{pad_mode}'")
        if not onesided:
            unsupported.append('onesided=False')
        if self.n_fft != frame_length:
            unsupported.append('n_fft != frame_length')
        if frame_length % hop_length != 0 or frame_length % 2 != 0:
            unsupported.append('hop_length not dividing an even frame_length')
        if unsupported:
            raise NotImplementedError(
                'not built yet on the HIP path: ' + ', '.join(unsupported))
        self._tables = {}

    # -- host-side constant tables (float64 -> fp32), cached per device ----------
    def _get_tables(self, device):
        key = str(device)
        if key not in self._tables:
            n = self.frame_length
            w = self.window.double().cpu().numpy()
            norm = 1.0/math.sqrt(float((w**2).sum())) if self.normalized else 1.0
            k = np.arange(n//2 + 1)[:, None]
            m = np.arange(n)[None, :]
            ang = 2.0*np.pi*k*m/n
            basis = np.empty((2*(n//2 + 1), n))
            basis[0::2] = np.cos(ang)*w[None, :]*norm
            basis[1::2] = -np.sin(ang)*w[None, :]*norm
            # inverse real DFT of a one-sided spectrum, windowed, normalisation undone
            eps = np.full(n//2 + 1, 2.0)
            eps[0] = 1.0
            eps[-1] = 1.0
            inv = np.empty((n, 2*(n//2 + 1)))
            inv[:, 0::2] = (np.cos(ang)*eps[:, None]).T
            inv[:, 1::2] = (-np.sin(ang)*eps[:, None]).T
            inv *= w[:, None]/(n*norm)
            self._tables[key] = (
                torch.from_numpy(basis).float().to(device).contiguous(),
                torch.from_numpy(inv).float().to(device).contiguous(),
                self.window.float().to(device).contiguous(),
            )
            # transposed inverse basis: the adjoint of the inverse transform is a framed DFT
            self._inv_t = getattr(self, '_inv_t', {})
            self._inv_t[key] = torch.from_numpy(np.ascontiguousarray(inv.T)).float().to(device)
        return self._tables[key]

    def __call__(self, x, return_type='complex'):
        return self.forward(x, return_type=return_type)

    def frame_count(self, samples):
        """Frames WITHOUT the n/2 centre padding (stft.py:146-149)."""
        return math.ceil(max(samples - self.frame_length, 0)/self.hop_length) + 1

    def pad(self, x):
        frames = self.frame_count(x.shape[-1])
        padding = (frames - 1)*self.hop_length + self.frame_length - x.shape[-1]
        return torch.nn.functional.pad(x, (0, padding), mode=self.pad_mode)

    def forward(self, x, return_type='complex'):
        hip.require_device(x)
        lib = hip.lib()
        basis, _, _ = self._get_tables(x.device)
        lead, L = x.shape[:-1], x.shape[-1]
        rows = int(np.prod(lead)) if lead else 1
        x2 = x.reshape(rows, L).float().contiguous()
        F = lib.brv_stft_frames(L, self.frame_length, self.hop_length)
        bins = self.frame_length//2 + 1
        out = _StftFunction.apply(x2, basis, self.frame_length, self.hop_length,
                                  self.compression_factor, self.scale_factor)
        out = out.view(*lead, bins, F)
        if return_type == 'complex':
            return out
        if return_type == 'real_imag':
            return out.real, out.imag
        if return_type == 'mag_phase':
            return out.abs(), out.angle()
        raise ValueError('return_type must be complex, real_imag or '
                         f'mag_phase, got {return_type}')

    def backward(self, x, input_type='complex'):
        if input_type == 'real_imag':
            x = torch.complex(*x)
        elif input_type == 'mag_phase':
            mag, phase = x
            x = mag*torch.exp(1j*phase)
        elif input_type != 'complex':
            raise ValueError('input_type must be complex, real_imag or '
                             f'mag_phase, got {input_type}')
        hip.require_device(x)
        lib = hip.lib()
        _, inv, win = self._get_tables(x.device)
        lead, (bins, F) = x.shape[:-2], x.shape[-2:]
        if bins != self.frame_length//2 + 1:
            raise ValueError(f'expected {self.frame_length//2 + 1} bins, got {bins}')
        rows = int(np.prod(lead)) if lead else 1
        spec = x.reshape(rows, bins, F).to(torch.complex64)
        out_len = self.hop_length*(F - 1)
        y = _IstftFunction.apply(spec, inv, win,
                                 self._inv_t[str(x.device)], self.frame_length, self.hop_length,
                                 self.compression_factor, self.scale_factor)
        return y.view(*lead, out_len)


class ConvSTFT:
    """STFT as a strided convolution with sqrt-window DFT rows and its transposed-convolution
    inverse (brever/modules/stft.py:201-319), same arguments and return types. Forward and
    backward are the framed DFT-GEMMs ``brv_framed_dft_forward / _transpose``."""

    def __init__(self, frame_length=512, hop_length=256, window='hann',
                 compression_factor=1, scale_factor=1, normalized=True):
        self.frame_length = frame_length
        self.hop_length = hop_length
        self.compression_factor = compression_factor
        self.scale_factor = scale_factor
        self.normalized = normalized
        if isinstance(window, str):
            window = scipy.signal.get_window(window, frame_length)**0.5
        if isinstance(window, np.ndarray):
            window = torch.from_numpy(window)
        self.window = window
        self._normalization_factor = 0.5*frame_length/hop_length**0.5
        n = frame_length
        k = np.arange(n//2 + 1)[:, None]
        m = np.arange(n)[None, :]
        ang = 2.0*np.pi*k*m/n
        re, im = np.cos(ang), -np.sin(ang)                 # rows of fft(eye(n))
        re[0] /= 2**0.5
        im[0] /= 2**0.5
        if normalized:
            re, im = re/self._normalization_factor, im/self._normalization_factor
        w = self.window.double().numpy()[None, :]
        basis = np.empty((2*(n//2 + 1), n))
        basis[0::2] = re*w
        basis[1::2] = im*w
        self._basis_host = torch.from_numpy(basis).float()
        # the reference's (2*bins, 1, n) convolution filters: real rows then imaginary rows
        self.filters = torch.cat([torch.from_numpy(re*w), torch.from_numpy(im*w)]) \
            .unsqueeze(1).float()
        self._basis = {}

    def _get_basis(self, device):
        key = str(device)
        if key not in self._basis:
            self._basis[key] = self._basis_host.to(device).contiguous()
        return self._basis[key]

    def __call__(self, x, return_type='complex'):
        return self.forward(x, return_type=return_type)

    def frame_count(self, samples):
        return math.ceil(max(samples - self.frame_length, 0)/self.hop_length) + 1

    def pad(self, x):
        frames = self.frame_count(x.shape[-1])
        padding = (frames - 1)*self.hop_length + self.frame_length - x.shape[-1]
        x = torch.nn.functional.pad(x, (0, padding))
        padding = self.frame_length - self.hop_length
        return torch.nn.functional.pad(x, (padding, padding))

    def forward(self, x, return_type='complex'):
        hip.require_device(x)
        lead, L = x.shape[:-1], x.shape[-1]
        rows = int(np.prod(lead)) if lead else 1
        n, hop = self.frame_length, self.hop_length
        side = n - hop
        padded = (self.frame_count(L) - 1)*hop + n + 2*side     # length after ConvSTFT.pad
        F = (padded - n)//hop + 1
        bins = n//2 + 1
        x2 = x.reshape(rows, L).float().contiguous()
        spec = torch.empty(rows, bins, F, 2, dtype=torch.float32, device=x.device)
        hip.check(hip.lib().brv_framed_dft_forward(
            hip.ptr(x2), hip.ptr(self._get_basis(x.device)), hip.ptr(spec), rows, L, n, hop,
            side, F, float(self.compression_factor), float(self.scale_factor), hip.stream()),
            'brv_framed_dft_forward')
        out = torch.view_as_complex(spec).view(*lead, bins, F)
        if return_type == 'complex':
            return out
        if return_type == 'real_imag':
            return out.real, out.imag
        if return_type == 'mag_phase':
            return out.abs(), out.angle()
        raise ValueError('return_type must be complex, real_imag or '
                         f'mag_phase, got {return_type}')

    def backward(self, x, input_type='complex'):
        if input_type == 'real_imag':
            x = torch.complex(*x)
        elif input_type == 'mag_phase':
            mag, phase = x
            x = mag*torch.exp(1j*phase)
        elif input_type != 'complex':
            raise ValueError('input_type must be complex, real_imag or '
                             f'mag_phase, got {input_type}')
        hip.require_device(x)
        lead, (bins, F) = x.shape[:-2], x.shape[-2:]
        rows = int(np.prod(lead)) if lead else 1
        n, hop = self.frame_length, self.hop_length
        side = n - hop
        out_len = (F - 1)*hop + n - 2*side                     # conv_transpose1d, trimmed
        spec = torch.view_as_real(x.reshape(rows, bins, F).to(torch.complex64).contiguous())
        scratch = torch.empty(rows, F, n, dtype=torch.float32, device=x.device)
        y = torch.empty(rows, out_len, dtype=torch.float32, device=x.device)
        hip.check(hip.lib().brv_framed_dft_transpose(
            hip.ptr(spec), hip.ptr(self._get_basis(x.device)), hip.ptr(scratch), hip.ptr(y), rows,
            F, n, hop, side, out_len, float(self.compression_factor), float(self.scale_factor),
            hip.stream()), 'brv_framed_dft_transpose')
        if not self.normalized:
            y = y/self._normalization_factor**2
        return y.view(*lead, out_len)


class MelFilterbank:
    """Triangular HTK-mel filterbank, rows normalised to sum 1
    (brever/modules/stft.py:152-198); ``forward`` / ``backward`` are fp32 GEMMs."""

    def __init__(self, n_filters=64, n_fft=512, fs=16e3, fmin=50, fmax=8000):
        self.n_filters = n_filters
        self.n_fft = n_fft
        self.fs = fs
        self.fmin = fmin
        self.fmax = fmax
        self.filters, self.fc, self.scaling = self.calc_filterbank()

    def calc_filterbank(self):
        mel = torch.linspace(self.freq_to_mel(self.fmin), self.freq_to_mel(self.fmax),
                             self.n_filters + 2)
        fc = self.mel_to_freq(mel)
        f = torch.from_numpy(fft_freqs(self.fs, self.n_fft)).float()
        filters = torch.zeros((self.n_filters, len(f)))
        for i in range(1, self.n_filters + 1):
            rise = (fc[i - 1] <= f) & (f <= fc[i])
            filters[i - 1, rise] = (f[rise] - fc[i - 1])/(fc[i] - fc[i - 1])
            fall = (fc[i] <= f) & (f <= fc[i + 1])
            filters[i - 1, fall] = (fc[i + 1] - f[fall])/(fc[i + 1] - fc[i])
        scaling = filters.sum(axis=1, keepdims=True)
        filters /= scaling
        return filters, fc, scaling

    @staticmethod
    def mel_to_freq(mel):
        return 700*(10**(mel/2595) - 1)

    @staticmethod
    def freq_to_mel(f):
        return 2595*math.log10(1 + f/700)

    @property
    def inverse_filters(self):
        return (self.filters*self.scaling).T

    def _matmul(self, matrix, x):
        hip.require_device(x)
        a = matrix.float().to(x.device).contiguous()
        lead, (K, N) = x.shape[:-2], x.shape[-2:]
        if K != a.shape[1]:
            raise ValueError(f'expected {a.shape[1]} rows, got {K}')
        rows = int(np.prod(lead)) if lead else 1
        x2 = x.reshape(rows, K, N).float().contiguous()
        out = torch.empty(rows, a.shape[0], N, dtype=torch.float32, device=x.device)
        hip.check(hip.lib().brv_matmul_f32(
            hip.ptr(a), hip.ptr(x2), hip.ptr(out), rows, a.shape[0], N, K, 0,
            hip.stream()), 'brv_matmul_f32')
        return out.view(*lead, a.shape[0], N)

    def __call__(self, x):
        return self.forward(x)

    def forward(self, x):
        return self._matmul(self.filters, x)

    def backward(self, x):
        return self._matmul(self.inverse_filters, x)
