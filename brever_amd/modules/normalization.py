"""Causal (cumulative) normalisation modules on the HIP path.

Same classes, arguments, parameter names (``gain``, ``bias``) and semantics as the reference
(brever/modules/normalization.py:5-71): every frame is normalised with the statistics of its
channel group over all frames up to and including it. Forward and backward are
``brv_causal_groupnorm_forward / _backward`` (fp32); a ``time_dim`` other than the last is
handled by a transposed copy around the kernels.
"""
import torch
import torch.nn as nn

from .. import hip


class _CausalGroupNormFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gain, bias, groups, eps):
        hip.require_device(x)
        lib = hip.lib()
        x = x.float().contiguous()
        B, C, T = x.shape[0], x.shape[1], x.shape[-1]
        inner = x.numel()//(B*C*T)
        y = torch.empty_like(x)
        stats = torch.empty(B*groups, T, 2, dtype=torch.float32, device=x.device)
        scratch = torch.empty(lib.brv_causal_groupnorm_scratch_bytes(B, groups, T), dtype=torch.uint8,
                              device=x.device)
        hip.check(lib.brv_causal_groupnorm_forward(
            hip.ptr(x), hip.ptr(gain), hip.ptr(bias), hip.ptr(y), hip.ptr(stats), hip.ptr(scratch),
            B, C, inner, T, groups, float(eps), hip.stream()), 'brv_causal_groupnorm_forward')
        ctx.save_for_backward(x, gain, stats)
        ctx.cfg = (groups, inner)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gain, stats = ctx.saved_tensors
        groups, inner = ctx.cfg
        lib = hip.lib()
        dy = dy.float().contiguous()
        B, C, T = x.shape[0], x.shape[1], x.shape[-1]
        dx = torch.empty_like(x)
        dgain, dbias = torch.empty_like(gain), torch.empty_like(gain)
        scratch = torch.empty(lib.brv_causal_groupnorm_scratch_bytes(B, groups, T), dtype=torch.uint8,
                              device=x.device)
        uv = torch.empty(B*groups, T, 2, dtype=torch.float32, device=x.device)
        hip.check(lib.brv_causal_groupnorm_backward(
            hip.ptr(x), hip.ptr(dy), hip.ptr(gain), hip.ptr(stats), hip.ptr(dx), hip.ptr(dgain),
            hip.ptr(dbias), hip.ptr(scratch), hip.ptr(uv), B, C, inner, T, groups, hip.stream()),
            'brv_causal_groupnorm_backward')
        return dx, dgain, dbias, None, None


class CausalGroupNorm(nn.Module):
    def __init__(self, num_channels, num_groups, time_dim=-1, eps=1e-10):
        super().__init__()
        if num_channels % num_groups != 0:
            raise ValueError('num_channels must be divisible by num_groups')
        self._check_time_dim(time_dim)
        self.num_groups = num_groups
        self.time_dim = time_dim
        self.eps = eps
        self.gain = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))

    def forward(self, x):
        time_dim = list(range(x.ndim))[self.time_dim]
        self._check_time_dim(time_dim)
        last = x.ndim - 1
        if time_dim != last:
            x = x.transpose(time_dim, last)
        y = _CausalGroupNormFunction.apply(x, self.gain, self.bias, self.num_groups, self.eps)
        return y.transpose(time_dim, last) if time_dim != last else y

    @staticmethod
    def _check_time_dim(time_dim):
        if time_dim == 0:
            raise ValueError('time_dim cannot be the batch dimension (0)')
        elif time_dim == 1:
            raise ValueError('time_dim cannot be the channel dimension (1)')


class CausalLayerNorm(CausalGroupNorm):
    def __init__(self, num_channels, time_dim=-1, eps=1e-10):
        super().__init__(num_channels=num_channels, num_groups=1, time_dim=time_dim, eps=eps)


class CausalInstanceNorm(CausalGroupNorm):
    def __init__(self, num_channels, time_dim=-1, eps=1e-10):
        super().__init__(num_channels=num_channels, num_groups=num_channels, time_dim=time_dim,
                         eps=eps)
