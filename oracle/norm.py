"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's causal normalisations
(brever/modules/normalization.py:5-46 CausalGroupNorm; :54-72 the LayerNorm / InstanceNorm
shorthands, groups = 1 / groups = channels). Pinned by tests/golden/norms.npz (generated from
the imported reference by tests/golden/make_golden.py). Nothing on the product path imports
this file.

Restated as running moments: with the input viewed as (batch, group, rest, frame), frame t is
normalised by the mean and the biased variance of everything its group holds in frames 0..t
(normalization.py:34-41), then scaled per channel (normalization.py:45-46)."""
import torch


def causal_group_norm(x, gain, bias, groups, time_dim=-1, eps=1e-10):
    """x (B, C, ...) -> same shape; differentiable (torch autograd) so the test can pin the
    gradients as well. `time_dim` may not be 0 or 1 (normalization.py:48-52)."""
    t_ax = range(x.ndim)[time_dim]
    if t_ax in (0, 1):
        raise ValueError('time_dim cannot be the batch or the channel dimension')
    B, C = x.shape[:2]
    if C % groups:
        raise ValueError('num_channels must be divisible by num_groups')
    z = x.movedim(t_ax, -1)                                   # frames last
    moved = z.shape
    T = moved[-1]
    z = z.reshape(B, groups, -1, T)                           # (B, G, rest, T)
    seen = z.shape[2]*torch.arange(1, T + 1, dtype=x.dtype)   # elements seen up to each frame
    m1 = z.sum(2).cumsum(-1)/seen                             # running mean (B, G, T)
    m2 = (z*z).sum(2).cumsum(-1)/seen                         # running mean of squares
    spread = (m2 - m1*m1 + eps).sqrt()
    z = (z - m1[:, :, None])/spread[:, :, None]
    z = z.reshape(moved).movedim(-1, t_ax)
    per_channel = [1, C] + [1]*(x.ndim - 2)
    return z*gain.view(per_channel) + bias.view(per_channel)
