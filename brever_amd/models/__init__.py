"""Model zoo of the HIP path. Importing this package populates ``ModelRegistry``
(reference: brever/models/__init__.py:1-37)."""
import torch

from .base import BreverBaseModel, ModelRegistry  # noqa: F401
from .convtasnet import ConvTasNet  # noqa: F401
from .dccrn import DCCRN  # noqa: F401
from .ffnn import FFNN  # noqa: F401
from .sgmse import IDMSE, SGMSEp, SGMSEpM  # noqa: F401
from .tfgridnet import TFGridNet  # noqa: F401


def count_params(model):
    """Number of trainable parameters."""
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


@torch.no_grad()
def set_all_weights(model, val=1e-3, buffers=False):
    """Fill every parameter (and optionally every buffer) with a constant."""
    for p in model.parameters():
        p.fill_(val)
    if buffers:
        for b in model.buffers():
            b.fill_(val)
    if hasattr(model, 'mark_params_changed'):
        model.mark_params_changed()
