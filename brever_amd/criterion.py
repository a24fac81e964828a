"""Length-masked criteria on the HIP path.

Same registry keys, argument meaning and assertions as the reference
(brever/criterion.py:14-132): ``snr``, ``sisnr``, ``mse`` take
``(x, y, lengths)`` with ``x, y`` of shape ``(batch, ..., length)`` and return
a ``(batch,)`` loss. The masking of ``apply_mask`` (criterion.py:229-234) is
folded into the reduction kernels: ``lengths`` stays on the device and there
is no per-item host synchronisation.

The arithmetic runs in ``libbrever_hip.so`` (``brv_snr_forward`` ...); CPU
tensors are rejected (no fallback) -- the CPU restatement used for checking
lives in ``oracle/criterion.py`` and is never imported from here.
"""
import inspect

import torch

from . import hip
from .registry import Registry

eps = torch.finfo(torch.float32).eps

CriterionRegistry = Registry('criterion')


def init_criterion(name, **kwargs):
    criterion = CriterionRegistry.get(name)
    if inspect.isclass(criterion):
        criterion = criterion(**kwargs)
    return criterion


def _rows(x, y, lengths):
    """Flatten ``(B, ..., L)`` to ``(B, S, L)`` contiguous fp32 rows."""
    hip.require_device(x, y, lengths)
    B, L = x.shape[0], x.shape[-1]
    x2 = x.reshape(B, -1, L).float().contiguous()
    y2 = y.reshape(B, -1, L).float().contiguous()
    lengths = lengths.to(device=x.device, dtype=torch.int64).contiguous()
    return x2, y2, lengths, B, x2.shape[1], L


def _scratch(B, S, device):
    n = hip.lib().brv_loss_scratch_bytes(B, S)
    return torch.empty(n, dtype=torch.uint8, device=device)


class _SnrFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, lengths):
        x2, y2, lengths, B, S, L = _rows(x, y, lengths)
        scratch = _scratch(B, S, x.device)
        loss = torch.empty(B, dtype=torch.float32, device=x.device)
        hip.check(hip.lib().brv_snr_forward(
            hip.ptr(x2), hip.ptr(y2), hip.ptr(lengths), B, S, L, L,
            hip.ptr(scratch), hip.ptr(loss), hip.stream()), 'brv_snr_forward')
        ctx.save_for_backward(x2, y2, lengths, scratch)
        ctx.shape = x.shape
        ctx.in_dtype = x.dtype
        return loss

    @staticmethod
    def backward(ctx, grad):
        x2, y2, lengths, scratch = ctx.saved_tensors
        B, S, L = x2.shape
        dx = torch.empty_like(x2)
        g = grad.float().contiguous()
        hip.check(hip.lib().brv_snr_backward(
            hip.ptr(x2), hip.ptr(y2), hip.ptr(lengths), B, S, L, L,
            hip.ptr(scratch), hip.ptr(g), hip.ptr(dx), hip.stream()),
            'brv_snr_backward')
        return dx.view(ctx.shape).to(ctx.in_dtype), None, None


class _SisnrFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, lengths):
        x2, y2, lengths, B, S, L = _rows(x, y, lengths)
        if S > 4:
            raise ValueError('sisnr supports at most 4 sources on the HIP path')
        scratch = _scratch(B, S, x.device)
        loss = torch.empty(B, dtype=torch.float32, device=x.device)
        hip.check(hip.lib().brv_sisnr_forward(
            hip.ptr(x2), hip.ptr(y2), hip.ptr(lengths), B, S, L, L,
            hip.ptr(scratch), hip.ptr(loss), hip.stream()), 'brv_sisnr_forward')
        ctx.save_for_backward(x2, y2, lengths, scratch)
        ctx.shape = x.shape
        ctx.in_dtype = x.dtype
        return loss

    @staticmethod
    def backward(ctx, grad):
        x2, y2, lengths, scratch = ctx.saved_tensors
        B, S, L = x2.shape
        dx = torch.empty_like(x2)
        g = grad.float().contiguous()
        hip.check(hip.lib().brv_sisnr_backward(
            hip.ptr(x2), hip.ptr(y2), hip.ptr(lengths), B, S, L, L,
            hip.ptr(scratch), hip.ptr(g), hip.ptr(dx), hip.stream()),
            'brv_sisnr_backward')
        return dx.view(ctx.shape).to(ctx.in_dtype), None, None


class _MseFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, lengths, weight):
        x2, y2, lengths, B, S, L = _rows(x, y, lengths)
        scratch = _scratch(B, S, x.device)
        loss = torch.empty(B, dtype=torch.float32, device=x.device)
        w = None
        if weight is not None:
            hip.require_device(weight)
            w = weight.float().contiguous()
        hip.check(hip.lib().brv_mse_forward(
            hip.ptr(x2), hip.ptr(y2), hip.ptr(lengths), hip.ptr(w), B, S, L, L,
            hip.ptr(scratch), hip.ptr(loss), hip.stream()), 'brv_mse_forward')
        ctx.save_for_backward(x2, y2, lengths)
        ctx.weight = w
        ctx.shape = x.shape
        ctx.in_dtype = x.dtype
        return loss

    @staticmethod
    def backward(ctx, grad):
        x2, y2, lengths = ctx.saved_tensors
        B, S, L = x2.shape
        dx = torch.empty_like(x2)
        g = grad.float().contiguous()
        hip.check(hip.lib().brv_mse_backward(
            hip.ptr(x2), hip.ptr(y2), hip.ptr(lengths), hip.ptr(ctx.weight), B, S,
            L, L, hip.ptr(g), hip.ptr(dx), hip.stream()), 'brv_mse_backward')
        return dx.view(ctx.shape).to(ctx.in_dtype), None, None, None


class _MultiResYuFunction(torch.autograd.Function):
    """Masked time-domain L1 + multi-resolution STFT-magnitude L1, per (item, source)
    row divided by the item length (brever/criterion.py:193-226, scale_invariant=False)."""

    @staticmethod
    def forward(ctx, x, y, lengths, stfts, time_w, spec_w):
        lib = hip.lib()
        x2, y2, lengths, B, S, L = _rows(x, y, lengths)
        rows = B*S
        xm, ym = torch.empty_like(x2), torch.empty_like(y2)
        for src, dst in ((x2, xm), (y2, ym)):
            hip.check(lib.brv_apply_mask(hip.ptr(src), hip.ptr(lengths), hip.ptr(dst), B, S, L,
                                         hip.stream()), 'brv_apply_mask')
        sums = torch.empty(rows, dtype=torch.float64, device=x.device)
        hip.check(lib.brv_l1_forward(hip.ptr(xm), hip.ptr(ym), hip.ptr(sums), rows, L,
                                     hip.stream()), 'brv_l1_forward')
        total = time_w*sums
        specs = []
        for stft in stfts:
            basis = stft._tables(x.device)['basis']
            with torch.no_grad():
                X = stft._dft_forward(xm.view(rows, L), basis, 1.0, stft.scale_factor)
                Y = stft._dft_forward(ym.view(rows, L), basis, 1.0, stft.scale_factor)
            n = X.shape[-2]*X.shape[-1]
            ssum = torch.empty(rows, dtype=torch.float64, device=x.device)
            hip.check(lib.brv_mag_l1_forward(
                hip.ptr(torch.view_as_real(X)), hip.ptr(torch.view_as_real(Y)), hip.ptr(ssum),
                rows, n, hip.stream()), 'brv_mag_l1_forward')
            total = total + (spec_w/len(stfts))*ssum
            specs.append((X, Y))
        total = total.view(B, S)/lengths.view(B, 1).double()
        ctx.save_for_backward(xm, ym, lengths, *[t for pair in specs for t in pair])
        ctx.meta = (stfts, time_w, spec_w, x.shape, x.dtype)
        return total.mean(1).float()

    @staticmethod
    def backward(ctx, grad):
        lib = hip.lib()
        xm, ym, lengths, *flat = ctx.saved_tensors
        stfts, time_w, spec_w, shape, in_dtype = ctx.meta
        B, S, L = xm.shape
        rows = B*S
        # d loss[b] / d row (b, s) total = 1/(S*len_b)
        grow = (grad.float()/(S*lengths.float())).repeat_interleave(S).contiguous()
        dx = torch.empty_like(xm)
        g_t = (time_w*grow).contiguous()
        hip.check(lib.brv_l1_backward(hip.ptr(xm), hip.ptr(ym), hip.ptr(g_t), hip.ptr(dx), rows,
                                      L, 0, hip.stream()), 'brv_l1_backward')
        for k, stft in enumerate(stfts):
            X, Y = flat[2*k], flat[2*k + 1]
            n = X.shape[-2]*X.shape[-1]
            g_s = (spec_w/len(stfts)*grow).contiguous()
            dX = torch.empty(*X.shape, 2, dtype=torch.float32, device=X.device)
            hip.check(lib.brv_mag_l1_backward(
                hip.ptr(torch.view_as_real(X)), hip.ptr(torch.view_as_real(Y)), hip.ptr(g_s),
                hip.ptr(dX), rows, n, hip.stream()), 'brv_mag_l1_backward')
            # adjoint of X = scale * DFT(x): scale * DFT^T
            dx += stft._dft_adjoint(dX, L, stft.scale_factor).view(B, S, L)
        out = torch.empty_like(dx)
        hip.check(lib.brv_apply_mask(hip.ptr(dx), hip.ptr(lengths), hip.ptr(out), B, S, L,
                                     hip.stream()), 'brv_apply_mask')
        return out.view(shape).to(in_dtype), None, None, None, None, None


class _ScaleInvariantFunction(torch.autograd.Function):
    """x -> alpha*x with alpha = <x, y>/(<x, x> + eps) per (item, source) row over the samples
    below the item length (brever/criterion.py:207-212); zero beyond the length."""

    @staticmethod
    def forward(ctx, x, y, lengths):
        x2, y2, lengths, B, S, L = _rows(x, y, lengths)
        out = torch.empty_like(x2)
        stats = torch.empty(B*S, 2, dtype=torch.float64, device=x2.device)
        hip.check(hip.lib().brv_si_scale_forward(
            hip.ptr(x2), hip.ptr(y2), hip.ptr(lengths), hip.ptr(out), hip.ptr(stats), B, S, L,
            float(eps), hip.stream()), 'brv_si_scale_forward')
        ctx.save_for_backward(x2, y2, lengths, stats)
        ctx.shape = x.shape
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        x2, y2, lengths, stats = ctx.saved_tensors
        B, S, L = x2.shape
        g2 = g.reshape(B, S, L).float().contiguous()
        dx = torch.empty_like(x2)
        hip.check(hip.lib().brv_si_scale_backward(
            hip.ptr(g2), hip.ptr(x2), hip.ptr(y2), hip.ptr(lengths), hip.ptr(stats), hip.ptr(dx), B,
            S, L, hip.stream()), 'brv_si_scale_backward')
        return dx.view(ctx.shape), None, None


@CriterionRegistry.register('multiresyu')
class MultiResYuLoss:
    """Multi-resolution STFT magnitude + L1 time-domain loss
    (brever/criterion.py:135-226): boxcar, un-normalised STFTs of ``frame_lengths`` /
    ``hop_lengths`` (default: half the frame), ``(B, ..., L)`` -> ``(B,)``.
    ``scale_invariant=True`` first rescales each estimate by its least-squares gain towards the
    target (``_ScaleInvariantFunction``)."""

    def __init__(self, frame_lengths=[512], hop_lengths=None, time_domain_weight=0.5,
                 spectral_weight=0.5, scale_invariant=False):
        from .modules.stft import STFT
        if hop_lengths is None:
            hop_lengths = [x // 2 for x in frame_lengths]
        self.stfts = [STFT(frame_length=n, hop_length=h, window=None, normalized=False)
                      for n, h in zip(frame_lengths, hop_lengths)]
        self.time_domain_weight = time_domain_weight
        self.spectral_weight = spectral_weight
        self.scale_invariant = scale_invariant

    def __call__(self, x, y, lengths):
        assert x.shape == y.shape
        if self.scale_invariant:
            x = _ScaleInvariantFunction.apply(x, y, lengths)
        return _MultiResYuFunction.apply(x, y, lengths, self.stfts,
                                         float(self.time_domain_weight),
                                         float(self.spectral_weight))


@CriterionRegistry.register('sisnr')
def sisnr(x, y, lengths):
    """PIT scale-invariant SNR, ``(B, S, L)`` -> ``(B,)``
    (brever/criterion.py:21-72); at most 4 sources on the HIP path."""
    assert x.shape == y.shape
    assert x.ndim == 3
    return _SisnrFunction.apply(x, y, lengths)


@CriterionRegistry.register('snr')
def snr(x, y, lengths):
    """SNR without PIT, ``(B, ..., L)`` -> ``(B,)`` (brever/criterion.py:75-101)."""
    assert x.shape == y.shape
    assert x.ndim >= 2
    return _SnrFunction.apply(x, y, lengths)


@CriterionRegistry.register('mse')
def mse(x, y, lengths, weight=None):
    """Masked mean squared error, ``(B, ..., L)`` -> ``(B,)``
    (brever/criterion.py:104-132); real or complex inputs."""
    assert x.shape == y.shape
    assert x.ndim >= 2
    if x.is_complex() or y.is_complex():
        # |x - y|^2 = re^2 + im^2: the real / imaginary parts become an extra middle axis of
        # size 2 (whose mean halves the sum) in front of the masked time axis
        xr = torch.view_as_real(x.to(torch.complex64)).movedim(-1, -2).contiguous()
        yr = torch.view_as_real(y.to(torch.complex64)).movedim(-1, -2).contiguous()
        w = weight.reshape(-1) if weight is not None else None
        return 2.0*_MseFunction.apply(xr, yr, lengths, w)
    return _MseFunction.apply(x, y, lengths, weight.reshape(-1) if weight is not None else None)
