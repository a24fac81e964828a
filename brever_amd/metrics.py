"""Evaluation metrics of the hot path: ``snr`` and ``sisnr`` (= minus the
criteria), reference brever/metrics.py:112-150. PESQ / STOI / ESTOI wrap
third-party C / NumPy wheels that are absent here and are out of scope for this
round."""
import torch

from .criterion import CriterionRegistry
from .registry import Registry

MetricRegistry = Registry('metric')


def _check_input(x, y, lengths):
    """Add batch / source dims, default and validate ``lengths``
    (brever/metrics.py:126-150)."""
    if x.shape != y.shape:
        raise ValueError('inputs must have same shape, got '
                         f'{x.shape} and {y.shape}')
    unbatched = x.ndim == 1
    if unbatched:
        x, y = x.unsqueeze(0), y.unsqueeze(0)
    if x.ndim != 2:
        raise ValueError(f'input must be 1 or 2 dimensional, got {x.ndim}')
    x, y = x.unsqueeze(1), y.unsqueeze(1)
    if lengths is None:
        lengths = torch.full((x.shape[0],), x.shape[-1], device=x.device)
    else:
        if len(lengths) != x.shape[0]:
            raise ValueError('lengths must have same length as batch size, '
                             f'got {len(lengths)} and {x.shape[0]}')
        if bool((torch.as_tensor(lengths) > x.shape[-1]).any()):
            raise ValueError('lengths items must be smaller than input '
                             f'length, got lengths={lengths} and '
                             f'input.shape={x.shape}')
    return x, y, lengths, unbatched


def _negated(name):
    def metric(x, y, lengths=None):
        x, y, lengths, unbatched = _check_input(x, y, lengths)
        with torch.no_grad():
            value = -CriterionRegistry.get(name)(x, y, lengths)
        return value.item() if unbatched else value
    metric.__name__ = name
    return metric


snr = MetricRegistry.register('snr')(_negated('snr'))
sisnr = MetricRegistry.register('sisnr')(_negated('sisnr'))
