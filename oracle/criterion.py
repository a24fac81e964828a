"""ORACLE (test infrastructure, not product code).

CPU restatement in plain PyTorch of the length-masked losses of the reference,
brever/criterion.py:21-234. Only ``tests/``, ``__graft_entry__.smoke``
and the ``cpu_baseline`` leg of ``bench.py`` may import this module; the
product path (``brever_amd``) never does.

Pinning: checked against ``tests/golden/losses.npz`` produced from the imported
reference (``tests/golden/make_golden.py``) and against the reference's own
"batched == per-item" property (tests/test_losses.py:13-57).
"""
from itertools import permutations

import torch

EPS = torch.finfo(torch.float32).eps   # brever/criterion.py:9


def length_mask(x, lengths):
    """0/1 mask over the last dim, 1 where index < lengths[b]
    (brever/criterion.py:229-234, without the per-item Python loop)."""
    idx = torch.arange(x.shape[-1], device=x.device)
    mask = idx.view(*[1]*(x.ndim - 1), -1) < lengths.view(-1, *[1]*(x.ndim - 1))
    return mask.to(torch.float32)


def snr(x, y, lengths):
    """-mean_src 10 log10(sum y^2 / (sum (y-x)^2 + eps) + eps) -> (B,)
    (brever/criterion.py:75-101)."""
    assert x.shape == y.shape and x.ndim >= 2
    m = length_mask(x, lengths)
    x, y = x*m, y*m
    ratio = y.pow(2).sum(-1)/((y - x).pow(2).sum(-1) + EPS)
    val = 10*torch.log10(ratio + EPS)
    return -val.mean(tuple(range(1, x.ndim - 1)))


def sisnr(x, y, lengths):
    """PIT SI-SNR, (B, S, L) x2 -> (B,) (brever/criterion.py:21-72)."""
    assert x.shape == y.shape and x.ndim == 3
    m = length_mask(x, lengths)
    n = lengths.view(-1, 1, 1)
    x, y = x*m, y*m
    x = (x - x.sum(2, keepdim=True)/n)*m
    y = (y - y.sum(2, keepdim=True)/n)*m
    est = x.unsqueeze(1)                       # (B, 1, S, L)
    ref = y.unsqueeze(2)                       # (B, S, 1, L)
    proj = (est*ref).sum(3, keepdim=True)*ref/ref.pow(2).sum(3, keepdim=True)
    noise = est - proj
    val = proj.pow(2).sum(3)/(noise.pow(2).sum(3) + EPS)
    val = 10*torch.log10(val + EPS)            # (B, S_ref, S_est)
    S = x.shape[1]
    best = None
    for perm in permutations(range(S)):
        # reference pairs ref i with est perm[i] through a one-hot einsum
        total = sum(val[:, i, perm[i]] for i in range(S))
        best = total if best is None else torch.maximum(best, total)
    return -(best/S)


def mse(x, y, lengths, weight=None):
    """masked sum |x-y|^2 / length, mean over middle dims
    (brever/criterion.py:104-132)."""
    assert x.shape == y.shape and x.ndim >= 2
    m = length_mask(x, lengths)
    err = ((x - y)*m).abs().pow(2).sum(-1)
    err = err/lengths.view(-1, *[1]*(x.ndim - 2))
    if weight is not None:
        err = err*weight.view(-1, *[1]*(x.ndim - 2))
    return err.mean(tuple(range(1, x.ndim - 1)))


def _boxcar_stft(x, frame_length, hop_length):
    """STFT(window=None, normalized=False) of the reference (brever/modules/stft.py:59-99,
    140-149): right-pad to whole frames, then torch.stft(center=True, constant padding)."""
    L = x.shape[-1]
    frames = -(-max(L - frame_length, 0)//hop_length) + 1
    pad = (frames - 1)*hop_length + frame_length - L
    x = torch.nn.functional.pad(x, (0, pad))
    lead = x.shape[:-1]
    out = torch.stft(x.reshape(-1, x.shape[-1]), n_fft=frame_length, hop_length=hop_length,
                     window=torch.ones(frame_length, dtype=x.dtype), center=True,
                     pad_mode='constant', normalized=False, onesided=True, return_complex=True)
    return out.view(*lead, *out.shape[-2:])


def multiresyu(x, y, lengths, frame_lengths=(512,), hop_lengths=None, time_domain_weight=0.5,
               spectral_weight=0.5, scale_invariant=False):
    """Time-domain L1 + multi-resolution STFT-magnitude L1, divided by the item length
    (brever/criterion.py:193-226); scale_invariant: the estimate is first multiplied by
    <x, y>/(<x, x> + eps) per row (:207-212)."""
    assert x.shape == y.shape
    if hop_lengths is None:
        hop_lengths = [n//2 for n in frame_lengths]
    m = length_mask(x, lengths)
    x, y = x*m, y*m
    if scale_invariant:
        x = x*((x*y).sum(-1, keepdim=True)/(x.pow(2).sum(-1, keepdim=True) + EPS))
    out = time_domain_weight*(x - y).abs().sum(-1)
    for n, h in zip(frame_lengths, hop_lengths):
        xm, ym = _boxcar_stft(x, n, h).abs(), _boxcar_stft(y, n, h).abs()
        out = out + spectral_weight*(xm - ym).abs().sum((-2, -1))/len(frame_lengths)
    out = out/lengths.view(-1, *[1]*(x.ndim - 2))
    return out.mean(tuple(range(1, x.ndim - 1)))


CRITERIA = {'snr': snr, 'sisnr': sisnr, 'mse': mse, 'multiresyu': multiresyu}
