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

The transforms run in ``libbrever_hip.so`` as DFT products with the basis in double precision
on the fp64 matrix pipe (``brv_dft64_forward`` / ``brv_dft64_synthesis`` + ``brv_overlap_add``):
data stay fp32 in HBM, the n-term sums accumulate in fp64, so the round trip meets the
reference's own ``atol=1e-6`` (tests/test_modules.py:319-326), which a plain fp32 DFT product
misses by 20x. The window-weighted bases are built here once per instance in float64 and cached
per device. Any hop, ``n_fft >= frame_length``, one- or two-sided spectra, ``center`` on or
off, every ``pad_mode`` of ``F.pad``. Both directions are differentiable, including through the
magnitude compression (``brv_spec_compress_backward``); the gradient through non-constant
padding is not built (``_dft_adjoint`` raises).
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


def _complex_pairs(t):
    """complex64 tensor -> contiguous float32 view (..., 2)."""
    return torch.view_as_real(t.to(torch.complex64).contiguous())


class _SpecCompress(torch.autograd.Function):
    """Y = scale |X|^(c-1) X on complex64 tensors (magnitude compression + scaling,
    stft.py:85-87,114-118) with its gradient, both as HIP kernels."""

    @staticmethod
    def forward(ctx, x, compression, scale):
        xr = _complex_pairs(x)
        y = torch.empty_like(xr)
        hip.check(hip.lib().brv_spec_compress(hip.ptr(xr), hip.ptr(y), x.numel(),
                                              float(compression), float(scale), hip.stream()),
                  'brv_spec_compress')
        ctx.save_for_backward(xr)
        ctx.cs = (float(compression), float(scale))
        return torch.view_as_complex(y)

    @staticmethod
    def backward(ctx, grad):
        xr, = ctx.saved_tensors
        gy = _complex_pairs(grad)
        gx = torch.empty_like(xr)
        hip.check(hip.lib().brv_spec_compress_backward(
            hip.ptr(xr), hip.ptr(gy), hip.ptr(gx), xr.numel()//2, *ctx.cs, hip.stream()),
            'brv_spec_compress_backward')
        return torch.view_as_complex(gx), None, None


class _StftLinear(torch.autograd.Function):
    """x (rows, L) fp32 -> X (rows, bins, F) complex64: the linear part of ``STFT.forward``
    (framing, window, DFT, normalisation; optionally the fused compression / scale when no
    gradient is needed). Gradient = the transposed product + plain overlap-add."""

    @staticmethod
    def forward(ctx, x2, stft, compression, scale):
        spec = stft._dft_forward(x2, stft._tables(x2.device)['basis'], compression, scale)
        ctx.stft, ctx.L = stft, x2.shape[-1]
        ctx.scale = float(scale)
        if compression != 1:
            ctx.mark_non_differentiable(spec)
        return spec

    @staticmethod
    def backward(ctx, grad):
        dx = ctx.stft._dft_adjoint(_complex_pairs(grad), ctx.L, ctx.scale)
        return dx, None, None, None


class _IstftLinear(torch.autograd.Function):
    """X (rows, bins, F) complex64 -> y (rows, out_len): inverse DFT of every frame, window,
    overlap-add divided by the window-square envelope (``torch.istft``). Gradient
    (``center=True``): envelope division, then a framed DFT with the synthesis basis."""

    @staticmethod
    def forward(ctx, spec, stft, compression, scale):
        y = stft._istft(spec, compression, scale)
        ctx.stft, ctx.geom = stft, (spec.shape[-1], y.shape[-1], float(scale))
        if compression != 1:
            ctx.mark_non_differentiable(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        stft = ctx.stft
        F, out_len, scale = ctx.geom
        if not stft.center:
            raise NotImplementedError('gradient of STFT.backward needs center=True')
        lib = hip.lib()
        dy = dy.float().contiguous()
        rows = dy.shape[0]
        tb = stft._tables(dy.device)
        u = torch.empty_like(dy)
        hip.check(lib.brv_istft_env_divide(hip.ptr(dy), hip.ptr(tb['window']), hip.ptr(u), rows,
                                           out_len, stft.n_fft, stft.hop_length, F, hip.stream()),
                  'brv_istft_env_divide')
        dspec = torch.empty(rows, stft.bins, F, 2, dtype=torch.float32, device=dy.device)
        hip.check(lib.brv_dft64_forward(
            hip.ptr(u), hip.ptr(tb['synthesis']), hip.ptr(dspec), rows, out_len, stft.n_fft,
            stft.hop_length, stft.n_fft//2, F, stft.bins, 1.0, 1.0/scale, hip.stream()),
            'brv_dft64_forward')
        return torch.view_as_complex(dspec), None, None, None


class STFT:
    """Same constructor, padding arithmetic, normalisation, compression, scaling and return types
    as the reference (stft.py:32-149). Every option of ``torch.stft`` the reference forwards is
    supported: any ``pad_mode``, any hop, ``n_fft >= frame_length`` (the window is
    zero-padded centrally to ``n_fft`` like torch does), one- and two-sided spectra,
    ``center`` on or off. Both directions are differentiable, through the magnitude
    compression as well."""

    _PAD_MODES = {'constant': 0, 'reflect': 1, 'replicate': 2, 'circular': 3}

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
        if pad_mode not in self._PAD_MODES:
            raise ValueError(f"pad_mode must be one of {sorted(self._PAD_MODES)}, got '{pad_mode}'")
        if self.n_fft < frame_length:
            raise ValueError(f'n_fft ({self.n_fft}) must be >= frame_length ({frame_length})')
        self.bins = self.n_fft//2 + 1 if onesided else self.n_fft
        self._cache = {}

    # -- host-side constant tables (float64), cached per device --------------------
    def _tables(self, device):
        key = str(device)
        if key not in self._cache:
            n, bins = self.n_fft, self.bins
            w = self.window.double().cpu().numpy()
            norm = 1.0/math.sqrt(float((w**2).sum())) if self.normalized else 1.0
            left = (n - self.frame_length)//2             # torch.stft centres the window in n_fft
            wp = np.zeros(n)
            wp[left:left + self.frame_length] = w
            ang = 2.0*np.pi*np.arange(bins)[:, None]*np.arange(n)[None, :]/n
            basis = np.empty((2*bins, n))
            basis[0::2] = np.cos(ang)*wp*norm
            basis[1::2] = -np.sin(ang)*wp*norm
            # frames of the inverse transform: w[m] * irfft(X[:n/2 + 1])[m] -- hermitian completion
            # = every bin except DC and Nyquist counted twice. torch.istft treats a two-sided
            # input the same way: it keeps bins 0..n/2 and ignores the rest. Normalisation undone.
            weight = np.zeros(bins)
            weight[:n//2 + 1] = 2.0
            weight[0] = 1.0
            if n % 2 == 0:
                weight[n//2] = 1.0
            synth = np.empty((2*bins, n))
            synth[0::2] = np.cos(ang)*weight[:, None]
            synth[1::2] = -np.sin(ang)*weight[:, None]
            synth *= wp[None, :]/(n*norm)
            self._cache[key] = {
                'basis': torch.from_numpy(basis).to(device).contiguous(),
                'synthesis': torch.from_numpy(synth).to(device).contiguous(),
                'window': torch.from_numpy(wp).float().to(device).contiguous(),
            }
        return self._cache[key]

    def __call__(self, x, return_type='complex'):
        return self.forward(x, return_type=return_type)

    def frame_count(self, samples):
        """Frames WITHOUT the n_fft/2 centre padding (stft.py:146-149)."""
        return math.ceil(max(samples - self.frame_length, 0)/self.hop_length) + 1

    def pad(self, x):
        frames = self.frame_count(x.shape[-1])
        padding = (frames - 1)*self.hop_length + self.frame_length - x.shape[-1]
        return torch.nn.functional.pad(x, (0, padding), mode=self.pad_mode)

    def _geometry(self, L):
        """(frames, left offset of frame 0) of torch.stft on the right-padded signal."""
        padded = (self.frame_count(L) - 1)*self.hop_length + self.frame_length
        pad_left = self.n_fft//2 if self.center else 0
        total = padded + 2*pad_left
        if total < self.n_fft:
            raise ValueError(f'input of {L} samples is shorter than one n_fft = {self.n_fft} frame')
        return (total - self.n_fft)//self.hop_length + 1, pad_left

    # -- the three products -----------------------------------------------------------
    def _explicit_padding(self, x2):
        """pad_mode != 'constant': the right padding of ``STFT.pad`` and torch.stft's centre
        padding are applied one after the other, each in ``pad_mode`` (zeros need no copy: the
        framed product reads them implicitly). Values only (no gradient through the padding)."""
        lib = hip.lib()
        mode = self._PAD_MODES[self.pad_mode]
        rows, L = x2.shape
        padded = (self.frame_count(L) - 1)*self.hop_length + self.frame_length
        stages = [(0, padded)] if padded > L else []
        if self.center:
            half = self.n_fft//2
            stages.append((half, padded + 2*half))
        for left, out_len in stages:
            y = torch.empty(rows, out_len, dtype=torch.float32, device=x2.device)
            status = lib.brv_pad_signal(hip.ptr(x2), hip.ptr(y), rows, x2.shape[-1], left, out_len,
                                        mode, hip.stream())
            if status == -2:
                raise ValueError(f"pad_mode='{self.pad_mode}' needs padding smaller than the "
                                 f'input ({x2.shape[-1]} samples)')
            hip.check(status, 'brv_pad_signal')
            x2 = y
        return x2

    def _dft_forward(self, x2, basis, compression=1.0, scale=1.0):
        rows, L = x2.shape
        F, pad_left = self._geometry(L)
        if self.pad_mode != 'constant':
            x2 = self._explicit_padding(x2)
            L, pad_left = x2.shape[-1], 0
        spec = torch.empty(rows, self.bins, F, 2, dtype=torch.float32, device=x2.device)
        hip.check(hip.lib().brv_dft64_forward(
            hip.ptr(x2), hip.ptr(basis), hip.ptr(spec), rows, L, self.n_fft, self.hop_length,
            pad_left, F, self.bins, float(compression), float(scale), hip.stream()),
            'brv_dft64_forward')
        return torch.view_as_complex(spec)

    def _dft_adjoint(self, dspec, L, scale=1.0):
        """dx (rows, L): adjoint of ``scale * _dft_forward`` applied to dspec (rows, bins, F, 2)."""
        if self.pad_mode != 'constant':
            raise NotImplementedError("gradient through pad_mode != 'constant' is not built")
        lib = hip.lib()
        rows, F = dspec.shape[0], dspec.shape[2]
        _, pad_left = self._geometry(L)
        frames = torch.empty(rows, F, self.n_fft, dtype=torch.float32, device=dspec.device)
        hip.check(lib.brv_dft64_synthesis(
            hip.ptr(dspec), hip.ptr(self._tables(dspec.device)['basis']), hip.ptr(frames), rows, F,
            self.n_fft, self.bins, 1.0, 1.0/float(scale), hip.stream()), 'brv_dft64_synthesis')
        dx = torch.empty(rows, L, dtype=torch.float32, device=dspec.device)
        hip.check(lib.brv_overlap_add(hip.ptr(frames), None, hip.ptr(dx), rows, F, self.n_fft,
                                      self.hop_length, pad_left, L, hip.stream()),
                  'brv_overlap_add')
        return dx

    def _istft(self, spec, compression=1.0, scale=1.0):
        lib = hip.lib()
        rows, bins, F = spec.shape
        tb = self._tables(spec.device)
        frames = torch.empty(rows, F, self.n_fft, dtype=torch.float32, device=spec.device)
        hip.check(lib.brv_dft64_synthesis(
            hip.ptr(_complex_pairs(spec)), hip.ptr(tb['synthesis']), hip.ptr(frames), rows, F,
            self.n_fft, bins, float(compression), float(scale), hip.stream()),
            'brv_dft64_synthesis')
        pad_left = self.n_fft//2 if self.center else 0
        out_len = self.n_fft + self.hop_length*(F - 1) - 2*pad_left
        y = torch.empty(rows, out_len, dtype=torch.float32, device=spec.device)
        hip.check(lib.brv_overlap_add(hip.ptr(frames), hip.ptr(tb['window']), hip.ptr(y), rows, F,
                                      self.n_fft, self.hop_length, pad_left, out_len, hip.stream()),
                  'brv_overlap_add')
        return y

    # -- public interface --------------------------------------------------------------------
    def forward(self, x, return_type='complex'):
        hip.require_device(x)
        lead, L = x.shape[:-1], x.shape[-1]
        rows = int(np.prod(lead)) if lead else 1
        x2 = x.reshape(rows, L).float().contiguous()
        c, s = self.compression_factor, self.scale_factor
        if c != 1 and x2.requires_grad and torch.is_grad_enabled():
            out = _SpecCompress.apply(_StftLinear.apply(x2, self, 1.0, 1.0), c, s)
        else:
            out = _StftLinear.apply(x2, self, c, s)
        out = out.view(*lead, *out.shape[-2:])
        if return_type == 'complex':
            return out
        if return_type == 'real_imag':
            return out.real, out.imag
        if return_type == 'mag_phase':
            if out.requires_grad:
                return out.abs(), out.angle()
            pairs = _complex_pairs(out)
            mag = torch.empty(out.shape, dtype=torch.float32, device=out.device)
            phase = torch.empty_like(mag)
            hip.check(hip.lib().brv_mag_phase(hip.ptr(pairs), hip.ptr(mag), hip.ptr(phase),
                                              out.numel(), hip.stream()), 'brv_mag_phase')
            return mag, phase
        raise ValueError('return_type must be complex, real_imag or '
                         f'mag_phase, got {return_type}')

    def backward(self, x, input_type='complex'):
        if input_type == 'real_imag':
            x = torch.complex(*x)
        elif input_type == 'mag_phase':
            mag, phase = x
            if mag.requires_grad or phase.requires_grad:
                x = torch.polar(mag, phase)
            else:
                hip.require_device(mag, phase)
                m, ph = mag.float().contiguous(), phase.float().contiguous()
                pairs = torch.empty(*m.shape, 2, dtype=torch.float32, device=m.device)
                hip.check(hip.lib().brv_polar(hip.ptr(m), hip.ptr(ph), hip.ptr(pairs), m.numel(),
                                              hip.stream()), 'brv_polar')
                x = torch.view_as_complex(pairs)
        elif input_type != 'complex':
            raise ValueError('input_type must be complex, real_imag or '
                             f'mag_phase, got {input_type}')
        hip.require_device(x)
        lead, (bins, F) = x.shape[:-2], x.shape[-2:]
        if bins != self.bins:
            raise ValueError(f'expected {self.bins} bins, got {bins}')
        rows = int(np.prod(lead)) if lead else 1
        spec = x.reshape(rows, bins, F).to(torch.complex64)
        c, s = self.compression_factor, self.scale_factor
        if c != 1 and spec.requires_grad and torch.is_grad_enabled():
            # X / scale, then |.|^(1/c): one compression op with exponent 1/c, scale s^(-1/c)
            spec = _SpecCompress.apply(spec, 1.0/c, float(s)**(-1.0/c))
            y = _IstftLinear.apply(spec, self, 1.0, 1.0)
        else:
            y = _IstftLinear.apply(spec, self, c, s)
        return y.view(*lead, y.shape[-1])


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
