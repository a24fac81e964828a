"""ORACLE (test infrastructure, not product code).

NumPy float64 restatement of the reference's STFT wrapper, brever/modules/stft.py
:59-149 (which calls torch.stft / torch.istft with center=True, pad_mode='constant',
normalized=False and applies its own normalisation / compression / scale) and of the
frame arithmetic :140-149, following SURVEY.md App. A.1 / A.2. Only tests import it.

Pinning: tests/golden/stft.npz (reference outputs for several parameter combos on a
seeded signal, frame counts for edge lengths) -- tests/test_oracle.py.
"""
import math

import numpy as np


def frame_count(samples, frame_length, hop_length):
    return math.ceil(max(samples - frame_length, 0)/hop_length) + 1


def stft_frames(samples, frame_length, hop_length):
    """Frames of STFT.forward including the n/2 centre padding."""
    nc = frame_count(samples, frame_length, hop_length)
    padded = (nc - 1)*hop_length + frame_length
    return 1 + padded//hop_length


def stft(x, window, hop_length, normalized=True, compression=1.0, scale=1.0):
    """x (..., L) real -> (..., n/2+1, F) complex."""
    x = np.asarray(x, dtype=np.float64)
    w = np.asarray(window, dtype=np.float64)
    n = len(w)
    L = x.shape[-1]
    nc = frame_count(L, n, hop_length)
    pad_r = (nc - 1)*hop_length + n - L
    xp = np.pad(x, [(0, 0)]*(x.ndim - 1) + [(n//2, pad_r + n//2)])
    F = 1 + (L + pad_r)//hop_length
    idx = np.arange(F)[:, None]*hop_length + np.arange(n)[None, :]
    frames = xp[..., idx]*w                                  # (..., F, n)
    X = np.fft.rfft(frames, axis=-1)                         # (..., F, n/2+1)
    X = np.swapaxes(X, -1, -2)
    if normalized:
        X = X/np.sqrt((w**2).sum())
    if compression != 1:
        X = np.abs(X)**compression*np.exp(1j*np.angle(X))
    return X*scale


def istft(X, window, hop_length, normalized=True, compression=1.0, scale=1.0):
    """(..., n/2+1, F) complex -> (..., hop*(F-1)) real."""
    X = np.asarray(X, dtype=np.complex128)/scale
    w = np.asarray(window, dtype=np.float64)
    n = len(w)
    if compression != 1:
        X = np.abs(X)**(1/compression)*np.exp(1j*np.angle(X))
    if normalized:
        X = X*np.sqrt((w**2).sum())
    F = X.shape[-1]
    frames = np.fft.irfft(np.swapaxes(X, -1, -2), n=n, axis=-1)*w    # (..., F, n)
    total = (F - 1)*hop_length + n
    y = np.zeros(X.shape[:-2] + (total,))
    env = np.zeros(total)
    for t in range(F):
        y[..., t*hop_length:t*hop_length + n] += frames[..., t, :]
        env[t*hop_length:t*hop_length + n] += w**2
    y = y/env
    return y[..., n//2:n//2 + hop_length*(F - 1)]


def _conv_basis(n, hop, window, normalized):
    """ConvSTFT filters (brever/modules/stft.py:219-239): rows of fft(eye(n)), DC row / sqrt 2,
    optional 0.5 n / sqrt(hop) normalisation, times the sqrt-window."""
    k = np.arange(n//2 + 1)[:, None]
    m = np.arange(n)[None, :]
    F = np.exp(-2j*np.pi*k*m/n)
    F[0] /= 2**0.5
    norm = 0.5*n/hop**0.5
    if normalized:
        F = F/norm
    return F*np.asarray(window, dtype=np.float64)[None, :], norm


def conv_stft(x, window, hop_length, normalized=True, compression=1.0, scale=1.0):
    """ConvSTFT.forward (stft.py:244-272): x (..., L) -> (..., n/2+1, F) complex."""
    x = np.asarray(x, dtype=np.float64)
    n = len(window)
    basis, _ = _conv_basis(n, hop_length, window, normalized)
    L = x.shape[-1]
    nc = frame_count(L, n, hop_length)
    side = n - hop_length
    xp = np.pad(x, [(0, 0)]*(x.ndim - 1) + [(side, (nc - 1)*hop_length + n - L + side)])
    F = (xp.shape[-1] - n)//hop_length + 1
    idx = np.arange(F)[:, None]*hop_length + np.arange(n)[None, :]
    X = np.einsum('...fn,kn->...kf', xp[..., idx], basis)
    if compression != 1:
        X = np.abs(X)**compression*np.exp(1j*np.angle(X))
    return X*scale


def conv_istft(X, window, hop_length, normalized=True, compression=1.0, scale=1.0):
    """ConvSTFT.backward (stft.py:274-307): conv_transpose1d with the same filters,
    /norm^2 when not normalised, trimmed by n - hop on both sides."""
    X = np.asarray(X, dtype=np.complex128)/scale
    n = len(window)
    basis, norm = _conv_basis(n, hop_length, window, normalized)
    if compression != 1:
        X = np.abs(X)**(1/compression)*np.exp(1j*np.angle(X))
    frames = np.einsum('...kf,kn->...fn', X.real, basis.real) \
        + np.einsum('...kf,kn->...fn', X.imag, basis.imag)
    F = X.shape[-1]
    total = (F - 1)*hop_length + n
    y = np.zeros(X.shape[:-2] + (total,))
    for t in range(F):
        y[..., t*hop_length:t*hop_length + n] += frames[..., t, :]
    if not normalized:
        y = y/norm**2
    side = n - hop_length
    return y[..., side:total - side]
