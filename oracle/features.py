"""ORACLE (test infrastructure, not product code).

NumPy float64 restatement of the reference's FeatureExtractor family
(brever/modules/features.py:142-262): filterbank energies with optional 'pdf' normalisation,
log / cubic compression, DCT-II + delta rows, and the interaural level / phase differences.
Pinned by tests/golden/features.npz (reference outputs on a seeded spectrum).
"""
import numpy as np
import scipy.fft

EPS = float(np.finfo(np.float32).eps)          # features.py:10 (torch.finfo().eps)


def fbe(spec, filters, normalize=False, compression='none', dct=False, n_dct=14):
    """spec (B, channels, bins, frames) complex, filters (n_filters, bins)."""
    out = (np.abs(spec)**2).mean(1)
    out = np.einsum('mk,bkt->bmt', filters, out)
    if normalize:
        out = out/(out.sum(1, keepdims=True) + EPS)
    if compression == 'log':
        out = np.log(out + EPS)
    elif compression == 'cubic':
        out = np.cbrt(out)
    if dct:
        cep = scipy.fft.dct(out, axis=1, type=2, norm='ortho')[:, 1:n_dct]
        d1 = np.zeros_like(cep); d1[..., 1:] = np.diff(cep, n=1, axis=2)
        d2 = np.zeros_like(cep); d2[..., 2:] = np.diff(cep, n=2, axis=2)
        out = np.concatenate([cep, d1, d2], axis=1)
    return out


def ild(spec, filters):
    mag = np.abs(spec)
    return np.einsum('mk,bkt->bmt', filters, 20*np.log10((mag[:, 1] + EPS)/(mag[:, 0] + EPS)))


def ipd(spec, filters):
    ph = np.angle(spec)
    return np.einsum('mk,bkt->bmt', filters, ph[:, 1] - ph[:, 0])


def ic(spec, filters, hop_length=256, fs=16e3, tau=10e-3):
    """Interaural coherence (features.py:263-293). PARITY UNPINNED: the reference calls
    torchaudio.functional.lfilter (torchaudio 2.1.1, requirements.txt; absent from this image),
    restated here from its documented behaviour: y[t] = b0 x[t] + b1 x[t-1] - a1 y[t-1] with
    a = [1, -alpha], b = [1 - alpha, 0], the OUTPUT clamped to [-1, 1] (``clamp=True`` default)."""
    import scipy.signal
    alpha = np.exp(-hop_length/(tau*fs))
    mag, ph = np.abs(spec), np.angle(spec)
    x_lr = mag[:, 0]*mag[:, 1]*np.exp(1j*(ph[:, 0] - ph[:, 1]))
    chans = np.stack([mag[:, 0]**2, mag[:, 1]**2, x_lr.real, x_lr.imag])
    phi = np.clip(scipy.signal.lfilter([1 - alpha, 0.0], [1.0, -alpha], chans, axis=-1), -1.0, 1.0)
    coh = (phi[2]**2 + phi[3]**2)/(phi[0]*phi[1])
    return np.sqrt(np.einsum('mk,bkt->bmt', filters, coh))


FEATURES = {
    'fbe': lambda s, f: fbe(s, f), 'logfbe': lambda s, f: fbe(s, f, compression='log'),
    'cubicfbe': lambda s, f: fbe(s, f, compression='cubic'),
    'pdf': lambda s, f: fbe(s, f, normalize=True),
    'logpdf': lambda s, f: fbe(s, f, normalize=True, compression='log'),
    'cubicpdf': lambda s, f: fbe(s, f, normalize=True, compression='cubic'),
    'mfcc': lambda s, f: fbe(s, f, compression='log', dct=True),
    'cubicmfcc': lambda s, f: fbe(s, f, compression='cubic', dct=True),
    'pdfcc': lambda s, f: fbe(s, f, normalize=True, compression='log', dct=True),
    'ild': ild, 'ipd': ipd,
}
