"""Model plugin surface: the drop-in boundary on the Python side.

Mirrors ``BreverBaseModel`` / ``ModelRegistry`` of the reference
(brever/models/base.py:9-358): a model is an ``nn.Module`` registered with
``@ModelRegistry.register(name)`` that provides ``loss`` and ``_enhance`` and
inherits ``transform / enhance / train_step / val_step / update / pre_train /
on_validate / optimizers / compile``. ``BreverTrainer`` and
``scripts/test_model.py`` only ever talk to these methods.
"""
from typing import Callable

import torch
import torch.nn as nn

from ..criterion import init_criterion
from ..registry import Registry

ModelRegistry = Registry('model')


class BreverBaseModel(nn.Module):
    """Base class of every model.

    Parameters
    ----------
    criterion : callable or str or None
        If a string, resolved through ``CriterionRegistry`` and stored as
        ``self.criterion`` (brever/models/base.py:43-53).
    """

    _is_submodel = False

    def __init__(
        self,
        criterion: Callable[..., torch.Tensor] | str | None = None,
    ):
        super().__init__()
        if criterion is not None:
            if isinstance(criterion, str):
                criterion = init_criterion(criterion)
            self.criterion = criterion
        self._compiled_call_impl = None

    # -- optimizers ----------------------------------------------------------
    def init_optimizer(self, optimizer, net=None, **kwargs):
        """``optimizer`` is a ``torch.optim`` class or its name
        (brever/models/base.py:55-79)."""
        if isinstance(optimizer, str):
            optimizer = getattr(torch.optim, optimizer)
        target = self if net is None else net
        return optimizer(target.parameters(), **kwargs)

    def optimizers(self):
        return self.optimizer

    # -- data interface ------------------------------------------------------
    def transform(self, sources):
        """``(n_sources, 2, n_samples)`` -> model inputs; identity by default.
        Must work on whatever device ``sources`` lives on."""
        return sources

    def enhance(self, x, use_amp=False):
        """``(2, L)`` -> ``(L,)``/``(S, L)`` or ``(B, 2, L)`` -> ``(B, L)``/
        ``(B, S, L)``; any other rank is a ``ValueError``
        (brever/models/base.py:122-155)."""
        if x.ndim == 2:
            return self._enhance(x.unsqueeze(0), use_amp).squeeze(0)
        if x.ndim != 3:
            raise ValueError(f'input must be 2 or 3 dimensional, got {x.ndim}')
        return self._enhance(x, use_amp)

    def _enhance(self, x, use_amp):
        raise NotImplementedError

    # -- optimisation steps --------------------------------------------------
    def train_step(self, batch, lengths, use_amp, scaler):
        self.optimizer.zero_grad()
        loss = self.loss(batch, lengths, use_amp)
        self.update(loss, scaler)
        return loss

    def val_step(self, batch, lengths, use_amp):
        return self.loss(batch, lengths, use_amp)

    def loss(self, batch, lengths, use_amp):
        raise NotImplementedError

    def update(self, loss, scaler, net=None, optimizer=None, grad_clip=0.0,
               retain_graph=None):
        """backward -> [unscale + clip_grad_norm_] -> optimizer step -> scaler
        update (brever/models/base.py:270-301)."""
        net = self if net is None else net
        optimizer = self.optimizer if optimizer is None else optimizer
        scaler.scale(loss).backward(retain_graph=retain_graph)
        if grad_clip != 0.0:
            scaler.unscale_(optimizer)
            torch.nn.utils.clip_grad_norm_(net.parameters(), grad_clip)
        scaler.step(optimizer)
        scaler.update()

    # -- hooks ---------------------------------------------------------------
    def pre_train(self, dataset, dataloader, epochs):
        pass

    def on_validate(self, val_loss):
        pass

    # -- in-place compilation (kept for API compatibility) ------------------
    def compile(self, *args, **kwargs):
        self._compiled_call_impl = torch.compile(self._call_impl, *args,
                                                 **kwargs)

    def __call__(self, *args, **kwargs):
        impl = self._compiled_call_impl
        if impl is None:
            impl = self._call_impl
        return impl(*args, **kwargs)

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop('_compiled_call_impl', None)
        return state
