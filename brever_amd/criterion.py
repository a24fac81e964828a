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
    (brever/criterion.py:104-132); real inputs (complex spectrograms: not built)."""
    assert x.shape == y.shape
    assert x.ndim >= 2
    if x.is_complex() or y.is_complex():
        raise NotImplementedError('complex mse is not built yet on the HIP path')
    return _MseFunction.apply(x, y, lengths, weight)
