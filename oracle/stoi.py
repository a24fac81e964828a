"""ORACLE (test infrastructure, not product code): STOI / ESTOI in NumPy.

The reference computes these metrics with the third-party wheels ``pystoi`` 0.3.3 and
``batch_pystoi`` 0.0.1 (brever/metrics.py:19-45,98-109; requirements.txt), which are NOT in
this image and not under /root/reference: **parity unpinned**. This file restates the
published algorithm they implement -- C. H. Taal et al., "An Algorithm for Intelligibility
Prediction of Time-Frequency Weighted Noisy Speech", IEEE TASLP 2011 (STOI) and J. Jensen,
C. H. Taal, "An Algorithm for Predicting the Intelligibility of Speech Masked by Modulated
Noise Maskers", IEEE/ACM TASLP 2016 (ESTOI) -- with pystoi's constants and processing order:
resampling to 10 kHz with its Octave-style Kaiser polyphase filter, removal of frames more
than 40 dB below the loudest clean frame (256-sample Hann frames, 50 % overlap, overlap-add),
512-point DFT of 256-sample Hann frames, 15 one-third octave bands from 150 Hz, 30-frame
(384 ms) segments, then either clipped, mean-removed, normalised correlations averaged over
bands and segments (STOI, beta = -15 dB) or row- and column-normalised segments (ESTOI).
Anchors that do not need the wheel: d(x, x) = 1, monotone in the SNR of x + noise, batched ==
one-by-one (the reference's own tests/test_metrics.py:13-54 property).
"""
import numpy as np
from scipy.signal import resample_poly

FS = 10000
N_FRAME = 256
NFFT = 512
NUMBAND = 15
MINFREQ = 150
N = 30
BETA = -15.0
DYN_RANGE = 40
EPS = np.finfo('float').eps


def resample_filter(p, q):
    """Kaiser-windowed sinc of Octave's ``resample`` (60 dB rejection), before normalisation."""
    g = np.gcd(p, q)
    p, q = p//g, q//g
    cutoff = 1.0/(2*max(p, q))
    roll_off = cutoff/10
    rejection = 60.0
    L = int(np.ceil((rejection - 8)/(28.714*roll_off)))
    t = np.arange(-L, L + 1)
    ideal = 2*p*cutoff*np.sinc(2*cutoff*t)
    beta = 0.1102*(rejection - 8.7)
    return np.kaiser(2*L + 1, beta)*ideal


def resample_oct(x, p, q):
    h = resample_filter(p, q)
    return resample_poly(x, p, q, window=h/np.sum(h))


def third_octave_matrix(fs=FS, nfft=NFFT, num_bands=NUMBAND, min_freq=MINFREQ):
    f = np.linspace(0, fs, nfft + 1)[:nfft//2 + 1]
    k = np.arange(num_bands, dtype=float)
    lo = min_freq*2.0**((2*k - 1)/6)
    hi = min_freq*2.0**((2*k + 1)/6)
    obm = np.zeros((num_bands, len(f)))
    edges = []
    for i in range(num_bands):
        a = int(np.argmin((f - lo[i])**2))
        b = int(np.argmin((f - hi[i])**2))
        obm[i, a:b] = 1
        edges.append((a, b))
    return obm, edges


def _frames(x, framelen, hop):
    w = np.hanning(framelen + 2)[1:-1]
    starts = range(0, len(x) - framelen, hop)
    return np.array([w*x[i:i + framelen] for i in starts]).reshape(len(starts), framelen)


def _overlap_add(frames, hop):
    n, framelen = frames.shape
    out = np.zeros(framelen + (n - 1)*hop if n > 0 else 0)
    for i in range(n):
        out[i*hop:i*hop + framelen] += frames[i]
    return out


def remove_silent_frames(x, y, dyn_range=DYN_RANGE, framelen=N_FRAME, hop=N_FRAME//2):
    xf, yf = _frames(x, framelen, hop), _frames(y, framelen, hop)
    if len(xf) == 0:
        return np.zeros(0), np.zeros(0)
    energies = 20*np.log10(np.linalg.norm(xf, axis=1) + EPS)
    keep = (np.max(energies) - dyn_range - energies) < 0
    return _overlap_add(xf[keep], hop), _overlap_add(yf[keep], hop)


def _band_spectrogram(x, obm):
    spec = np.fft.rfft(_frames(x, N_FRAME, N_FRAME//2), n=NFFT).T        # (bins, frames)
    return np.sqrt(obm @ np.abs(spec)**2)


def _row_col_normalize(s):
    s = s - s.mean(axis=2, keepdims=True)
    s = s/(np.linalg.norm(s, axis=2, keepdims=True) + EPS)
    s = s - s.mean(axis=1, keepdims=True)
    return s/(np.linalg.norm(s, axis=1, keepdims=True) + EPS)


def stoi(clean, processed, fs_sig, extended=False):
    """STOI (or ESTOI) of ``processed`` against ``clean`` (1-D arrays of equal length)."""
    x, y = np.asarray(clean, dtype=float), np.asarray(processed, dtype=float)
    if x.shape != y.shape:
        raise ValueError('clean and processed must have the same shape')
    if fs_sig != FS:
        x, y = resample_oct(x, FS, fs_sig), resample_oct(y, FS, fs_sig)
    x, y = remove_silent_frames(x, y)
    obm, _ = third_octave_matrix()
    if len(x) <= N_FRAME:
        return 1e-5
    xb, yb = _band_spectrogram(x, obm), _band_spectrogram(y, obm)
    if xb.shape[1] < N:
        return 1e-5                      # pystoi: "not enough frames", with a warning
    xs = np.array([xb[:, m - N:m] for m in range(N, xb.shape[1] + 1)])
    ys = np.array([yb[:, m - N:m] for m in range(N, xb.shape[1] + 1)])
    if extended:
        return float(np.sum(_row_col_normalize(xs)*_row_col_normalize(ys)/N)/xs.shape[0])
    scale = np.linalg.norm(xs, axis=2, keepdims=True)/(np.linalg.norm(ys, axis=2, keepdims=True) + EPS)
    yp = np.minimum(ys*scale, xs*(1 + 10**(-BETA/20)))
    yp = yp - yp.mean(axis=2, keepdims=True)
    xs = xs - xs.mean(axis=2, keepdims=True)
    yp = yp/(np.linalg.norm(yp, axis=2, keepdims=True) + EPS)
    xs = xs/(np.linalg.norm(xs, axis=2, keepdims=True) + EPS)
    return float(np.sum(yp*xs)/(xs.shape[0]*xs.shape[1]))
