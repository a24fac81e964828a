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
        (brever/models/base.py:55-79). Plain Adam over the whole model becomes ``FlatAdam``: the
        parameters are packed into ONE flat fp32 buffer (every ``nn.Parameter`` a view of it, same
        ``state_dict``) and ``update`` runs ``clip_grad_norm_`` + ``Adam.step`` as the two launches of
        ``brv_clip_adam_step2`` instead of PyTorch's multi-tensor kernels (K11; every model, not only
        Conv-TasNet). Anything else (other optimizers, options FlatAdam does not implement, a
        sub-network) is constructed as the reference does."""
        if isinstance(optimizer, str):
            optimizer = getattr(torch.optim, optimizer)
        if self._fused_adam and optimizer is torch.optim.Adam and net is None \
                and set(kwargs) <= {'lr', 'betas', 'eps'} \
                and all(p.dtype == torch.float32 for p in self.parameters()) \
                and any(True for _ in self.parameters()):
            from ..optim import FlatAdam
            self._flat_base = True
            self._flatten_base()
            return FlatAdam(self.parameters(), owner=self, **kwargs)
        target = self if net is None else net
        return optimizer(target.parameters(), **kwargs)

    # flat parameter storage of the generic FlatAdam path (Conv-TasNet brings its own: models/convtasnet.py).
    # Opt-in per model class: the HIP models (which only step on a ROCm device) set `_fused_adam = True`
    _fused_adam = False
    _flat_base = False
    _grad_sync = None

    def set_grad_sync(self, fn):
        """Data parallelism (brever_amd.parallel.GradSynchronizer; the reference wraps any model in DDP,
        brever/training.py:62-63). ``fn(flat_grad)`` sums the buffer over ranks and returns the factor that turns
        the sum into the mean. Every ``update`` of this model calls it after backward has ended (and after
        whatever side streams the backward pass used were joined): ONE collective per step on the flat gradient
        instead of one blocking collective per parameter from inside backward. ``False`` = this model keeps
        separate parameters (no flat buffer): the caller installs per-parameter hooks instead."""
        if not self._flat_base:
            return False
        self._grad_sync = fn
        return True

    def _flatten_base(self):
        params = list(self.parameters())
        total = sum(p.numel() for p in params)
        flat = torch.empty(total, dtype=torch.float32, device=params[0].device)
        offsets, off = [], 0
        for p in params:
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            offsets.append((p, off))
            off += n
        self._flat, self._offsets, self._flat_grad = flat, offsets, None

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if self._flat_base:
            self._flatten_base()             # (.to(device) gave every parameter a storage of its own again)
        return out

    def flat_params(self):
        """The one buffer every parameter lives in, or None for a model that keeps separate parameters."""
        return self._flat if self._flat_base else None

    def param_offsets(self):
        return self._offsets

    def mark_params_changed(self):
        pass

    def _after_backward(self):
        """Hook of ``update`` between ``backward`` and the first read of a gradient (DCCRN joins its side stream)."""

    def gather_grads(self):
        """Flat gradient for ``FlatAdam``: the ``.grad`` tensors autograd left (separate allocations) are
        copied into one buffer by a single multi-tensor copy; parameters without a gradient count as zero."""
        if self._flat_grad is None or self._flat_grad.device != self._flat.device:
            self._flat_grad = torch.zeros_like(self._flat)
            self._grad_slices = [self._flat_grad[off:off + p.numel()].view(p.shape) for p, off in self._offsets]
        have = [(v, p.grad) for v, (p, _) in zip(self._grad_slices, self._offsets) if p.grad is not None]
        if len(have) != len(self._offsets):
            # (differs from torch.optim.Adam, which SKIPS a parameter without a gradient: here it steps on a zero
            # gradient, i.e. its moments decay and it keeps moving on old momentum -- ADVICE r4; said once)
            if not getattr(self, '_warned_missing_grads', False):
                import logging
                logging.getLogger(__name__).warning(
                    '%d of %d parameters have no gradient: FlatAdam treats them as zero gradients (torch.optim.Adam '
                    'would skip them)', len(self._offsets) - len(have), len(self._offsets))
                self._warned_missing_grads = True
            self._flat_grad.zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        return self._flat_grad

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
        whole = net is None
        net = self if net is None else net
        optimizer = self.optimizer if optimizer is None else optimizer
        scaler.scale(loss).backward(retain_graph=retain_graph)
        self._after_backward()
        from ..optim import FlatAdam
        sync = self._grad_sync
        if whole and isinstance(optimizer, FlatAdam) and getattr(optimizer, '_owner', None) is self \
                and not scaler.is_enabled():
            # clip_grad_norm_(grad_clip) + Adam.step fused on the flat buffers (two HIP launches); under a process
            # group the flat gradient is summed over ranks first and the mean taken inside the kernel
            if sync is None:
                optimizer.step(max_norm=float(grad_clip))
            else:
                grads = self.gather_grads()
                optimizer.step(max_norm=float(grad_clip), grad_scale=sync(grads), grads=grads)
            return
        if sync is not None:
            # a sub-network, another optimizer or a live scaler: the same mean, written back to the .grad tensors
            from ..parallel import all_reduce_mean_grads
            all_reduce_mean_grads(net.parameters(), sync)
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
