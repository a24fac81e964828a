"""Conv-TasNet on the MI355X HIP path.

Plugin-compatible with the reference model (registry key ``convtasnet``, same
constructor signature and defaults, same ``state_dict`` keys and shapes, same
seeded initialisation order -- brever/models/convtasnet/convtasnet.py:19-97 and
SURVEY.md App. A.3) but *none* of its layers is ever executed by PyTorch: the
``nn.Conv1d`` / ``nn.GroupNorm`` / ``nn.PReLU`` objects below are parameter
containers only. All parameters are views of one flat fp32 buffer that the C
ABI (``brv_ctn_forward`` / ``brv_ctn_backward``, ``include/brever_hip.h``)
consumes; activations live in a caller-allocated workspace in HBM.

Two ways in:

* ``forward(x)`` is differentiable (``torch.autograd.Function``), so the
  reference's generic ``loss`` -> ``update`` sequence works unchanged;
* ``train_step`` short-circuits autograd: forward -> SNR loss -> backward ->
  [gradient all-reduce hook] -> fused clip + Adam, all on flat buffers.

Compute dtype follows ``use_amp`` like the reference's autocast switch
(convtasnet.py:78-97): ``use_amp=True`` = bf16 storage / MFMA operands with fp32
accumulation, statistics and master weights (``brv_ctn_*``; the reference's GPU
branch would use fp16 + GradScaler, convtasnet.py:81 -- bf16 needs no scaler);
``use_amp=False`` = fp32 activations and products of fp32 accuracy (fp32 MFMA / split-bf16)
(``brv_ctn_f32_*``), the precision of ``enhance(x)`` in scripts/test_model.py
and of ``BreverTrainer(use_amp=False)``. A bare ``model(x)`` is fp32 unless it
runs under ``torch.autocast``.
"""
import os

import torch
import torch.nn as nn

from .. import hip
from ..optim import FlatAdam
from .base import BreverBaseModel, ModelRegistry


class _ParamOnly(nn.Module):
    """Container whose children hold parameters but are never called."""

    def forward(self, *args, **kwargs):
        raise RuntimeError('this module only stores parameters; the compute '
                           'runs in libbrever_hip.so')


class _CausalLayerNorm(_ParamOnly):
    """Parameter container of the cumulative layer norm: ``gain`` / ``bias`` as in the
    reference (brever/modules/normalization.py:17-18), so causal state dicts load."""

    def __init__(self, channels):
        super().__init__()
        self.gain = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))


def _norm(causal, channels):
    if causal:
        return _CausalLayerNorm(channels)
    return nn.GroupNorm(1, channels, eps=1e-8)


class _Encoder(_ParamOnly):
    def __init__(self, filters, filter_length):
        super().__init__()
        self.filter_length = filter_length
        self.stride = filter_length//2
        self.conv = nn.Conv1d(1, filters, filter_length, stride=self.stride,
                              bias=False)


class _Decoder(_ParamOnly):
    def __init__(self, filters, filter_length):
        super().__init__()
        self.filter_length = filter_length
        self.stride = filter_length//2
        self.trans_conv = nn.ConvTranspose1d(filters, 1, filter_length,
                                             stride=self.stride, bias=False)


class _Block(_ParamOnly):
    def __init__(self, bn, hidden, skip, kernel_size, dilation, causal, last):
        super().__init__()
        self.conv = nn.Conv1d(bn, hidden, 1)
        self.d_conv = nn.Conv1d(hidden, hidden, kernel_size, dilation=dilation,
                                groups=hidden)
        self.res_conv = None if last else nn.Conv1d(hidden, bn, 1)
        self.skip_conv = nn.Conv1d(hidden, skip, 1)
        self.norm_1 = _norm(causal, hidden)
        self.norm_2 = _norm(causal, hidden)
        self.prelu_1 = nn.PReLU()
        self.prelu_2 = nn.PReLU()


class _TCN(_ParamOnly):
    def __init__(self, filters, bn, hidden, skip, kernel_size, layers, repeats,
                 sources, causal):
        super().__init__()
        self.sources = sources
        self.layer_norm = _norm(causal, filters)
        self.bottleneck_conv = nn.Conv1d(filters, bn, 1)
        self.conv_blocks = nn.ModuleList()
        for r in range(repeats):
            for i in range(layers):
                last = r == repeats - 1 and i == layers - 1
                self.conv_blocks.append(
                    _Block(bn, hidden, skip, kernel_size, 2**i, causal, last))
        self.prelu = nn.PReLU()
        self.output_conv = nn.Conv1d(skip, filters*sources, 1)


_warned_queues = False


def _queues_ok():
    import brever_amd
    return brever_amd.HW_QUEUES_OK


def _process_group():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


class _ConvTasNetFunction(torch.autograd.Function):
    """wave (B, L) -> (B, S, L) with gradients for every parameter."""

    @staticmethod
    def forward(ctx, model, wave, amp, *params):
        out = model._hip_forward(wave, amp)
        ctx.model = model
        ctx.wave = wave
        ctx.amp = amp
        ctx.version = model._ws_version[amp]
        return out

    @staticmethod
    def backward(ctx, d_out):
        model = ctx.model
        if ctx.version != model._ws_version[ctx.amp]:
            raise RuntimeError(
                'the activation workspace was overwritten by a later forward; '
                'call backward before running the model again'
            )
        flat_grad = model._autograd_buffer()
        model._hip_backward(ctx.wave, d_out, flat_grad, ctx.amp)
        grads = tuple(flat_grad[off:off + p.numel()].view(p.shape)
                      for p, off in model._offsets)
        return (None, None, None) + grads


@ModelRegistry.register('convtasnet')
class ConvTasNet(BreverBaseModel):
    def __init__(
        self,
        filters: int = 512,
        filter_length: int = 32,
        bottleneck_channels: int = 128,
        hidden_channels: int = 512,
        skip_channels: int = 128,
        kernel_size: int = 3,
        layers: int = 8,
        repeats: int = 3,
        output_sources: int = 1,
        causal: bool = False,
        criterion: str = 'snr',
        optimizer: str = 'Adam',
        learning_rate: float = 0.001,
        grad_clip: float = 5.0,
    ):
        super().__init__(criterion=criterion)
        self._criterion_name = criterion if isinstance(criterion, str) else None
        self.cfg = hip.CtnConfig(
            filters, filter_length, bottleneck_channels, hidden_channels,
            skip_channels, kernel_size, layers, repeats, output_sources,
            int(bool(causal)))
        self.output_sources = output_sources
        # parameter containers, constructed in the reference's RNG order
        self.encoder = _Encoder(filters, filter_length)
        self.decoder = _Decoder(filters, filter_length)
        self.tcn = _TCN(filters, bottleneck_channels, hidden_channels,
                        skip_channels, kernel_size, layers, repeats,
                        output_sources, causal)

        self._flat = None
        self._flat_grad = None
        self._offsets = []
        self._prepared = None
        self._prepared_dirty = True
        self._workspace = {}
        self._ws_key = {}
        self._ws_version = {True: 0, False: 0}
        self._grad_sync = None
        self._amp = False
        self._step_bufs = None
        self._two = None
        self._flatten()

        self.optimizer = self.init_optimizer(optimizer, lr=learning_rate)
        self.grad_clip = grad_clip

    # ---- flat parameter storage -------------------------------------------------
    def _flatten(self):
        """(Re)pack all parameters, in ``parameters()`` order, into one buffer and
        turn every ``nn.Parameter`` into a view of it."""
        params = list(self.parameters())
        total = sum(p.numel() for p in params)
        flat = torch.empty(total, dtype=torch.float32, device=params[0].device)
        offsets, off = [], 0
        for p in params:
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            p.grad = None
            offsets.append((p, off))
            off += n
        self._flat = flat
        self._flat_grad = None
        self._offsets = offsets
        self._prepared = None
        self._prepared_dirty = True
        self._workspace = {}
        self._ws_key = {}
        self._step_bufs = None
        self._two = None

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._flatten()
        return out

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)   # copies in place
        self._prepared_dirty = True
        return out

    def flat_params(self):
        return self._flat

    def param_offsets(self):
        return self._offsets

    def flat_grads(self):
        """Flat gradient buffer with every ``p.grad`` bound to a view of it."""
        if self._flat_grad is None or self._flat_grad.device != self._flat.device:
            self._flat_grad = torch.zeros_like(self._flat)
        g = self._flat_grad
        views = getattr(self, '_grad_views', None)
        if views is None or views[0] is not g:
            views = (g, [g[off:off + p.numel()].view(p.shape) for p, off in self._offsets])
            self._grad_views = views
        # re-bind only what something else replaced (zero_grad(set_to_none=True), autograd)
        for (p, _), view in zip(self._offsets, views[1]):
            if p.grad is not view:
                p.grad = view
        return g

    def gather_grads(self):
        """Flat gradient for the optimizer: zero-copy when the ``.grad`` tensors
        already are views of one buffer (both entry paths of this model),
        otherwise gathered."""
        first = self._offsets[0][0].grad
        if first is not None:
            base = first.data_ptr()
            contiguous = all(
                p.grad is not None and p.grad.is_contiguous()
                and p.grad.data_ptr() == base + 4*off
                for p, off in self._offsets)
            if contiguous:
                storage = first.untyped_storage()
                start = (base - storage.data_ptr())//4
                flat = torch.empty(0, dtype=torch.float32, device=first.device)
                flat.set_(storage, start, (self._flat.numel(),))
                return flat
        g = torch.zeros_like(self._flat)
        for p, off in self._offsets:
            if p.grad is not None:
                g[off:off + p.numel()].copy_(p.grad.reshape(-1))
        return g

    def mark_params_changed(self):
        self._prepared_dirty = True

    def set_grad_sync(self, fn):
        """``fn(flat_grad) -> grad_scale`` is called between backward and the
        optimizer step (data-parallel all-reduce). If ``fn`` also has ``nparts > 1``,
        ``bucket(part, grad_slice)`` and ``finish() -> grad_scale``, the fused
        ``train_step`` runs the backward pass in ``nparts`` parts and hands each
        finished gradient slice to ``bucket`` right away, so that its all-reduce
        overlaps the rest of backward (brever_amd.parallel.GradSynchronizer)."""
        self._grad_sync = fn

    def init_optimizer(self, optimizer, net=None, **kwargs):
        if optimizer == 'Adam' and net is None:
            return FlatAdam(self.parameters(), owner=self, **kwargs)
        return super().init_optimizer(optimizer, net=net, **kwargs)

    # ---- HIP calls ------------------------------------------------------------------
    def _cfg_ptr(self):
        import ctypes
        return ctypes.byref(self.cfg)

    def frames(self, length):
        return hip.lib().brv_ctn_frames(self._cfg_ptr(), int(length))

    def activation_bytes_per_second(self, fs=16000):
        """HBM kept resident per second of input audio (for the dynamic batcher)."""
        return hip.lib().brv_ctn_workspace_bytes(self._cfg_ptr(), 1, fs)

    def _check_layout(self):
        lib = hip.lib()
        n = lib.brv_ctn_param_count(self._cfg_ptr())
        if n < 0:
            hip.check(int(n), 'brv_ctn_param_count')
        if n != self._flat.numel():
            raise RuntimeError(f'parameter layout mismatch: library expects {n} '
                               f'floats, module holds {self._flat.numel()}')

    def _prepare(self):
        lib = hip.lib()
        hip.require_device(self._flat)
        if self._prepared is None:
            self._check_layout()
            nbytes = lib.brv_ctn_prepared_bytes(self._cfg_ptr())
            self._prepared = torch.empty(nbytes, dtype=torch.uint8,
                                         device=self._flat.device)
            self._prepared_dirty = True
        # the prepared operands depend on the mode switches of the call (fused forward: gamma-folded
        # [res | skip] weights, else the plain ones): a switch toggled on a live model prepares again
        opts = hip.launch_opts()
        if self._prepared_dirty or getattr(self, '_prepared_flags', None) != opts.flags:
            hip.check(lib.brv_ctn_prepare(
                self._cfg_ptr(), hip.ptr(self._flat), hip.ptr(self._prepared),
                hip.opts_ptr(opts), hip.stream()), 'brv_ctn_prepare')
            self._prepared_dirty = False
            self._prepared_flags = opts.flags

    def _get_workspace(self, B, L, amp=True):
        """Activation workspace of one precision (``amp``: bf16 path, else fp32)."""
        amp = bool(amp)
        key = (B, L, self._flat.device)
        if self._ws_key.get(amp) != key:
            fn = hip.lib().brv_ctn_workspace_bytes if amp \
                else hip.lib().brv_ctn_f32_workspace_bytes
            nbytes = fn(self._cfg_ptr(), B, L)
            if nbytes < 0:
                hip.check(int(nbytes), 'brv_ctn_workspace_bytes')
            ws = self._workspace.get(amp)
            if ws is None or ws.numel() < nbytes \
                    or ws.device != self._flat.device:
                self._workspace[amp] = torch.empty(
                    nbytes, dtype=torch.uint8, device=self._flat.device)
            self._ws_key[amp] = key
        return self._workspace[amp]

    def _autograd_buffer(self):
        """Fresh zeroed flat buffer for the gradients autograd hands back. Never reused: the tensors
        returned by ``backward`` are views of it, and a caller of ``torch.autograd.grad`` (or of a
        second backward with ``retain_graph``) may keep them across the next backward -- a recycled
        buffer silently overwrote them. The caching allocator recycles the memory once
        nothing refers to it; the hot path (``train_step``) does not come through here."""
        return torch.zeros_like(self._flat)

    @staticmethod
    def _rows(x, amp):
        """``(tensor, row stride)`` of a (batch, length) fp32 input for the kernels: the bf16 path reads
        rows in place at any row stride (``batch[:, 0]`` of the trainer's (B, 1 + S, L) tensor: no strided
        torch copy kernel in front of every step); the fp32 path and
        anything else get a contiguous copy."""
        if amp and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 \
                and x.stride(0) >= x.shape[1]:
            return x, x.stride(0)
        x = x.float().contiguous()
        return x, x.shape[1]

    @staticmethod
    def _label_rows(labels):
        """``(tensor, item stride, source stride)`` of the (batch, sources, length) labels, in place when
        the samples are contiguous fp32 (``batch[:, 1:]``), else of a contiguous copy."""
        if labels.dtype == torch.float32 and labels.dim() == 3 and labels.stride(2) == 1:
            return labels, labels.stride(0), labels.stride(1)
        labels = labels.float().contiguous()
        return labels, labels.stride(0), labels.stride(1)

    def _hip_forward(self, wave, amp=True):
        hip.require_device(wave, self._flat)
        if wave.ndim != 2:
            raise ValueError(f'input must be (batch, length), got {wave.shape}')
        amp = bool(amp)
        wave, wstride = self._rows(wave, amp)
        B, L = wave.shape
        ws = self._get_workspace(B, L, amp)
        out = torch.empty(B, self.output_sources, L, dtype=torch.float32,
                          device=wave.device)
        if amp:
            self._prepare()
            opts = hip.launch_opts()
            hip.check(hip.lib().brv_ctn_forward(
                self._cfg_ptr(), hip.ptr(self._flat), hip.ptr(self._prepared),
                hip.ptr(ws), hip.ptr(wave), wstride, hip.ptr(out), B, L, hip.opts_ptr(opts),
                hip.stream()), 'brv_ctn_forward')
        else:
            self._check_layout()
            hip.check(hip.lib().brv_ctn_f32_forward(
                self._cfg_ptr(), hip.ptr(self._flat), hip.ptr(ws),
                hip.ptr(wave), hip.ptr(out), B, L, hip.stream()),
                'brv_ctn_f32_forward')
        self._ws_version[amp] += 1
        return out

    def grad_buckets(self, nparts):
        """``[(offset, count)]``: the slice of the flat gradient that is final after
        part ``p`` of an ``nparts``-part backward (``brv_ctn_grad_bucket``)."""
        import ctypes
        out = []
        for part in range(nparts):
            off, cnt = ctypes.c_int64(0), ctypes.c_int64(0)
            hip.check(hip.lib().brv_ctn_grad_bucket(
                self._cfg_ptr(), part, nparts, ctypes.byref(off), ctypes.byref(cnt)),
                'brv_ctn_grad_bucket')
            out.append((off.value, cnt.value))
        return out

    def _hip_backward(self, wave, d_out, flat_grad, amp=True, nparts=1, after_part=None):
        """Backward into ``flat_grad``. With ``nparts > 1`` the pass runs in parts and
        ``after_part(part, flat_grad[offset:offset + count])`` is called after each one
        with the gradient slice that part finished (bucketed all-reduce hook)."""
        amp = bool(amp)
        wave, wstride = self._rows(wave, amp)
        d_out = d_out.float().contiguous()
        B, L = wave.shape
        ws = self._get_workspace(B, L, amp)
        buckets = self.grad_buckets(nparts) if after_part is not None else None
        lib = hip.lib()
        opts = hip.launch_opts()
        for part in range(nparts):
            if amp:
                hip.check(lib.brv_ctn_backward_part(
                    self._cfg_ptr(), hip.ptr(self._flat), hip.ptr(self._prepared),
                    hip.ptr(ws), hip.ptr(wave), wstride, hip.ptr(d_out), hip.ptr(flat_grad),
                    B, L, part, nparts, hip.opts_ptr(opts), hip.stream()), 'brv_ctn_backward_part')
            else:
                hip.check(lib.brv_ctn_f32_backward_part(
                    self._cfg_ptr(), hip.ptr(self._flat), hip.ptr(ws),
                    hip.ptr(wave), hip.ptr(d_out), hip.ptr(flat_grad), B, L,
                    part, nparts, hip.stream()), 'brv_ctn_f32_backward_part')
            if after_part is not None:
                off, cnt = buckets[part]
                if cnt:
                    after_part(part, flat_grad[off:off + cnt])

    def workspace_tensor(self, name, index, B, L, shape, dtype):
        """View of a saved activation of the bf16 path (tests / profiling)."""
        off = hip.lib().brv_ctn_workspace_offset(
            self._cfg_ptr(), B, L, name.encode(), index)
        if off < 0:
            hip.check(int(off), 'brv_ctn_workspace_offset')
        n = 1
        for s in shape:
            n *= s
        itemsize = torch.empty(0, dtype=dtype).element_size()
        raw = self._workspace[True][off:off + n*itemsize]
        return raw.view(dtype).view(*shape)

    # ---- plugin surface ---------------------------------------------------------------
    def _use_amp(self):
        return bool(self._amp) or torch.is_autocast_enabled()

    def forward(self, x):
        amp = self._use_amp()
        if torch.is_grad_enabled() and any(p.requires_grad for p, _ in self._offsets):
            return _ConvTasNetFunction.apply(self, x, amp,
                                             *[p for p, _ in self._offsets])
        return self._hip_forward(x, amp)

    def transform(self, sources):
        return sources.mean(axis=-2)      # mono; runs wherever `sources` lives

    def loss(self, batch, lengths, use_amp):
        inputs, labels = batch[:, 0], batch[:, 1:]
        prev, self._amp = self._amp, bool(use_amp)     # convtasnet.py:78-86 autocast switch
        try:
            outputs = self(inputs)
        finally:
            self._amp = prev
        loss = self.criterion(outputs, labels, lengths)
        return loss.mean()

    def update(self, loss, scaler):
        """backward -> [gradient all-reduce hook] -> clip -> step (base.py:270-301).
        The data-parallel hook runs here too, so every criterion / optimizer is
        synchronised, not only the fused ``train_step``."""
        scaler.scale(loss).backward()
        scaler.unscale_(self.optimizer)
        flat_opt = isinstance(self.optimizer, FlatAdam)
        grad_scale = 1.0
        if self._grad_sync is not None:
            grads = self.gather_grads()
            grad_scale = self._grad_sync(grads)
            if not flat_opt or scaler.is_enabled():
                grads.mul_(grad_scale)
                grad_scale = 1.0
            self._scatter_grads(grads)       # no-op when the .grad tensors are views of `grads`
        if flat_opt and not scaler.is_enabled():
            self.optimizer.step(max_norm=self.grad_clip, grad_scale=grad_scale)
            return
        if self.grad_clip != 0.0:
            torch.nn.utils.clip_grad_norm_(self.parameters(), self.grad_clip)
        scaler.step(self.optimizer)
        scaler.update()

    def _scatter_grads(self, flat):
        """Write a gathered (copied) flat gradient back to the ``.grad`` tensors."""
        for p, off in self._offsets:
            if p.grad is not None and p.grad.data_ptr() != flat.data_ptr() + 4*off:
                p.grad.copy_(flat[off:off + p.numel()].view(p.shape))

    def _enhance(self, x, use_amp):
        return self._hip_forward(x.mean(axis=-2), use_amp)

    def train_step(self, batch, lengths, use_amp, scaler):
        """Fused step when the criterion is the HIP ``snr`` and the optimizer the
        flat Adam; otherwise the reference's generic sequence
        (brever/models/base.py:178-210)."""
        fused = (self._criterion_name == 'snr'
                 and isinstance(self.optimizer, FlatAdam))
        if not fused:
            return super().train_step(batch, lengths, use_amp, scaler)
        # neither bf16 nor fp32 needs loss scaling: the GradScaler is left untouched
        lib = hip.lib()
        inputs, labels = batch[:, 0], batch[:, 1:]
        hip.require_device(inputs, lengths)
        B, L = inputs.shape
        S = self.output_sources
        amp = bool(use_amp)
        sync = self._grad_sync
        if self.uses_two_chains(B, amp):
            return self._train_step_two_chains(inputs, labels, lengths)
        with torch.no_grad():
            out = self._hip_forward(inputs, amp)
            labels, ybs, yss = self._label_rows(labels)
            lengths = lengths.to(torch.int64).contiguous()
            scratch, loss_b, gscale, d_out = self._step_buffers(B, S, L, out.device)
            hip.check(lib.brv_snr_forward_strided(
                hip.ptr(out), hip.ptr(labels), ybs, yss, hip.ptr(lengths), B, S, L, L,
                hip.ptr(scratch), hip.ptr(loss_b), hip.stream()),
                'brv_snr_forward_strided')
            hip.check(lib.brv_snr_backward_strided(
                hip.ptr(out), hip.ptr(labels), ybs, yss, hip.ptr(lengths), B, S, L, L,
                hip.ptr(scratch), hip.ptr(gscale), hip.ptr(d_out), hip.stream()),
                'brv_snr_backward_strided')
            grads = self.flat_grads()
            hip.check(lib.brv_memset_zero(hip.ptr(grads), 4*grads.numel(), hip.stream()), 'brv_memset_zero')
            sync = self._grad_sync
            grad_scale = 1.0
            if sync is not None and getattr(sync, 'nparts', 1) > 1:
                # bucketed: each part's slice is all-reduced while the next part computes
                self._hip_backward(inputs, d_out, grads, amp, nparts=sync.nparts,
                                   after_part=sync.bucket)
                grad_scale = sync.finish()
            else:
                self._hip_backward(inputs, d_out, grads, amp)
                if sync is not None:
                    grad_scale = sync(grads)
            self.optimizer.step(max_norm=self.grad_clip, grad_scale=grad_scale)
            loss = torch.empty((), dtype=torch.float32, device=out.device)
            hip.check(lib.brv_mean_f32(hip.ptr(loss_b), B, hip.ptr(loss), hip.stream()), 'brv_mean_f32')
            return loss

    @staticmethod
    def uses_two_chains(B, amp):
        """Whether the fused bf16 step of a batch of ``B`` runs as two half-batch kernel chains on two
        streams: batches of >= 8 (odd ones split 1 + B//2 | B//2), unless ``BRV_CTN_STREAMS=1`` -- and never next
        to an initialised process group when the HIP runtime cannot have seen >= 8 hardware queues
        (``brever_amd.HW_QUEUES_OK``: torch imported before the package and ``GPU_MAX_HW_QUEUES`` not
        exported): with the default 4 queues RCCL's streams and the two chains share queues and the
        step is slower than one chain (9.6 vs 7.7 ms)."""
        if not (bool(amp) and B >= 8 and os.environ.get('BRV_CTN_STREAMS', '2') != '1'):
            return False
        if _queues_ok() or not _process_group():
            return True
        # next to a process group without >= 8 hardware queues. An EXPORTED value below 8 is a
        # configuration error of the launch (a 20 % slower step): fail loudly; BRV_CTN_STREAMS=1 opts
        # into the one-chain step. A default that came too late (torch imported before this package)
        # falls back to one chain with a warning, once.
        import brever_amd
        exported = getattr(brever_amd, '_exported', None)
        if exported is not None:
            raise RuntimeError(
                f'GPU_MAX_HW_QUEUES={exported} is exported, but the two-chain Conv-TasNet step next to RCCL '
                'needs >= 8 hardware queues (DESIGN.md 5h): export GPU_MAX_HW_QUEUES=8 (or unset it before '
                'importing brever_amd), or set BRV_CTN_STREAMS=1 to run the one-chain step')
        global _warned_queues
        if not _warned_queues:
            import warnings
            warnings.warn('brever_amd was imported after torch and GPU_MAX_HW_QUEUES is not exported: the HIP '
                          'runtime has 4 hardware queues, the Conv-TasNet step runs ONE kernel chain next to '
                          'the process group (export GPU_MAX_HW_QUEUES=8 for the two-chain step)')
            _warned_queues = True
        return False

    def _train_step_two_chains(self, inputs, labels, lengths):
        """The fused bf16 step as TWO independent half-batch chains on two streams. Every launch of
        the TCN depends on the one before it (a layer norm over the whole item sits between them), so
        a single chain leaves the chip idle for the ~4.5 us between dependent launches and in the tail
        of every persistent kernel: 0.6 ms of a 7.9 ms step (rocprofv3 kernel trace,
        tools/trace_gaps.py). Items are independent until the weight gradients are summed, so the two
        halves of the batch run as separate chains whose workgroups fill each other's gaps. Same
        arithmetic per item; the weight gradient is g(first half) + g(second half). Measured
        7.97 -> 7.65 ms per step (16 x 4 s), 7.45 ms with the persistent kernels at 7/8 of the CUs
        meanwhile; four chains: 10.4 ms. ``BRV_CTN_STREAMS=1``: one chain."""
        lib = hip.lib()
        B, L = inputs.shape
        S = self.output_sources
        nB = self._chain_split(B)                 # items per chain (odd batches: the first takes one more)
        dev = inputs.device
        with torch.no_grad():
            wave, wstride = self._rows(inputs, True)        # read in place: no strided copies
            labels, ybs, yss = self._label_rows(labels)
            lengths = lengths.to(torch.int64).contiguous()
            self._prepare()
            t = self._two_chain_buffers(B, S, L, dev)
            grads = self.flat_grads()
            # (no PyTorch kernel between here and the end of the step: memsets, the sum of the two
            # chains' gradients, the loss mean and the optimizer are library launches)
            main, side = torch.cuda.current_stream(dev), t['side']
            side.wait_stream(main)
            # persistent kernels at 7/8 of the CUs while two chains share the chip (csrc: num_cus) --
            # an option of these calls, not a state of the library
            opts = hip.launch_opts(cu_eighths=int(os.environ.get('BRV_CTN_CU_EIGHTHS', '7')))
            po = hip.opts_ptr(opts)
            streams = (main, side)
            flat, prep, cfg = hip.ptr(self._flat), hip.ptr(self._prepared), self._cfg_ptr()

            def half(h):
                sl = slice(0, nB[0]) if h == 0 else slice(nB[0], B)
                return (wave[sl], labels[sl], lengths[sl], t['out'][sl], t['d_out'][sl], t['loss'][sl],
                        t['gscale'][sl])
            for h in (0, 1):                     # forward + loss of both halves, then both backwards
                x, y, ln, out, d_out, loss_b, gscale = half(h)
                with torch.cuda.stream(streams[h]):
                    st = hip.stream()
                    if h == 1:
                        if not t['grad2_zero']:
                            hip.check(lib.brv_memset_zero(hip.ptr(t['grad2']), 4*t['grad2'].numel(), st),
                                      'brv_memset_zero')
                        t['grad2_zero'] = False        # (until this step's sum pass has re-zeroed it)
                    hip.check(lib.brv_ctn_forward(cfg, flat, prep, hip.ptr(t['ws'][h]), hip.ptr(x), wstride,
                                                  hip.ptr(out), nB[h], L, po, st), 'brv_ctn_forward')
                    hip.check(lib.brv_snr_forward_strided(
                        hip.ptr(out), hip.ptr(y), ybs, yss, hip.ptr(ln), nB[h], S, L, L,
                        hip.ptr(t['scratch'][h]), hip.ptr(loss_b), st), 'brv_snr_forward_strided')
                    hip.check(lib.brv_snr_backward_strided(
                        hip.ptr(out), hip.ptr(y), ybs, yss, hip.ptr(ln), nB[h], S, L, L,
                        hip.ptr(t['scratch'][h]), hip.ptr(gscale), hip.ptr(d_out), st),
                        'brv_snr_backward_strided')
                    if h == 0:
                        # the first chain's gradient buffer, zeroed behind ITS forward (the other chain's kernels
                        # fill the chip meanwhile) instead of in the serial section in front of both chains
                        hip.check(lib.brv_memset_zero(hip.ptr(grads), 4*grads.numel(), st), 'brv_memset_zero')
            sync = self._grad_sync
            nparts = getattr(sync, 'nparts', 1) if sync is not None else 1
            buckets = self.grad_buckets(nparts) if nparts > 1 else [(0, grads.numel())]
            for part in range(nparts):
                for h in (0, 1):
                    x, y, ln, out, d_out, loss_b, gscale = half(h)
                    with torch.cuda.stream(streams[h]):
                        hip.check(lib.brv_ctn_backward_part(
                            cfg, flat, prep, hip.ptr(t['ws'][h]), hip.ptr(x), wstride, hip.ptr(d_out),
                            hip.ptr(grads if h == 0 else t['grad2']), nB[h], L, part, nparts, po, hip.stream()),
                            'brv_ctn_backward_part')
                # this part's slice of the gradient is final in both halves
                main.wait_stream(side)
                off, cnt = buckets[part]
                if cnt and nparts > 1:
                    # bucketed all-reduce: sum the slice now and hand it over while the next part's
                    # chains run (the second buffer is re-zeroed at the next step's start)
                    hip.check(lib.brv_axpby(hip.ptr(grads[off:]), 1.0, hip.ptr(t['grad2'][off:]), 1.0,
                                            hip.ptr(grads[off:]), cnt, hip.stream()), 'brv_axpby')
                    sync.bucket(part, grads[off:off + cnt])
            if nparts > 1:
                grad_scale, second = sync.finish(), None
                t['grad2_zero'] = False
            else:
                # one pass: grads += grad2, grad2 = 0, squared norm -- then [all-reduce] clip + Adam
                second = t['grad2']
                if sync is None:
                    grad_scale = 1.0
                else:
                    hip.check(lib.brv_axpby(hip.ptr(grads), 1.0, hip.ptr(second), 1.0, hip.ptr(grads),
                                            grads.numel(), hip.stream()), 'brv_axpby')
                    hip.check(lib.brv_memset_zero(hip.ptr(second), 4*second.numel(), hip.stream()),
                              'brv_memset_zero')
                    grad_scale, second = sync(grads), None
            self.optimizer.step(max_norm=self.grad_clip, grad_scale=grad_scale, grads2=second)
            # (only now: a failed launch above must not leave the buffer marked as zeroed)
            t['grad2_zero'] = nparts == 1
            loss = torch.empty((), dtype=torch.float32, device=dev)
            hip.check(lib.brv_mean_f32(hip.ptr(t['loss']), B, hip.ptr(loss), hip.stream()), 'brv_mean_f32')
            return loss

    @staticmethod
    def _chain_split(B):
        """Items of the two chains: halves. ``BRV_CTN_SPLIT=n`` (A/B runs): n items on the first chain -- measured at
        16 x 4 s (round 5, profiles/r05_chain_split.txt): uneven chains do not interleave better than halves."""
        n = os.environ.get('BRV_CTN_SPLIT')
        if n is not None and 1 <= int(n) < B:
            return (int(n), B - int(n))
        return ((B + 1)//2, B//2)

    def _two_chain_buffers(self, B, S, L, dev):
        """Buffers of the two-chain step. GROW-ONLY (the trainer's dynamic batches change (B, L) almost
        every step): the two half-batch activation workspaces are slices of the model's one bf16
        workspace -- the buffer validation / ``enhance`` / odd batches use as a whole -- and the
        output / loss buffers are views of flat allocations that are replaced only by larger ones (the
        old reference is dropped first: no 2x peak). The side stream is created once."""
        lib = hip.lib()
        Bh = max(self._chain_split(B))             # the larger part sizes both workspaces
        nws = lib.brv_ctn_workspace_bytes(self._cfg_ptr(), Bh, L)
        if nws < 0:
            hip.check(int(nws), 'brv_ctn_workspace_bytes')
        nws = (int(nws) + 255)//256*256
        nscr = (int(lib.brv_loss_scratch_bytes(Bh, S)) + 255)//256*256
        t = self._two
        if t is None or t['dev'] != dev:
            t = self._two = dict(dev=dev, side=torch.cuda.Stream(device=dev), cap_out=0, cap_scr=0, cap_b=0,
                                 out_flat=None, d_out_flat=None, scr_flat=None, loss_flat=None,
                                 gscale_flat=None, grad2=torch.zeros_like(self._flat), grad2_zero=True)
        # both halves inside the shared workspace (invalidates a saved autograd forward, like any forward)
        ws = self._workspace.get(True)
        if ws is None or ws.numel() < 2*nws or ws.device != dev:
            self._workspace[True] = None
            ws = None
            self._workspace[True] = ws = torch.empty(2*nws, dtype=torch.uint8, device=dev)
        self._ws_key[True] = None
        self._ws_version[True] += 1
        t['ws'] = [ws[:nws], ws[nws:2*nws]]
        n_out = B*S*L
        if t['cap_out'] < n_out:
            t['out_flat'] = t['d_out_flat'] = None
            t['out_flat'] = torch.empty(n_out, dtype=torch.float32, device=dev)
            t['d_out_flat'] = torch.empty(n_out, dtype=torch.float32, device=dev)
            t['cap_out'] = n_out
        if t['cap_scr'] < 2*nscr:
            t['scr_flat'] = None
            t['scr_flat'] = torch.empty(2*nscr, dtype=torch.uint8, device=dev)
            t['cap_scr'] = 2*nscr
        if t['cap_b'] < B:
            t['loss_flat'] = torch.empty(B, dtype=torch.float32, device=dev)
            t['gscale_flat'] = torch.empty(B, dtype=torch.float32, device=dev)
            t['cap_b'] = B
            t['gscale_B'] = None
        if t.get('gscale_B') != B:
            t['gscale_flat'][:B].fill_(1.0/B)
            t['gscale_B'] = B
        t['out'] = t['out_flat'][:n_out].view(B, S, L)
        t['d_out'] = t['d_out_flat'][:n_out].view(B, S, L)
        t['scratch'] = [t['scr_flat'][:nscr], t['scr_flat'][nscr:2*nscr]]
        t['loss'] = t['loss_flat'][:B]
        t['gscale'] = t['gscale_flat'][:B]
        if t['grad2'].device != dev or t['grad2'].numel() != self._flat.numel():
            t['grad2'] = torch.zeros_like(self._flat)
            t['grad2_zero'] = True
        return t

    def _step_buffers(self, B, S, L, device):
        """Loss scratch, per-item losses, gradient scales and d_out of the fused
        step, allocated once per (B, S, L) instead of every step."""
        key = (B, S, L, device)
        if self._step_bufs is None or self._step_bufs[0] != key:
            n = hip.lib().brv_loss_scratch_bytes(B, S)
            self._step_bufs = (key, (
                torch.empty(n, dtype=torch.uint8, device=device),
                torch.empty(B, dtype=torch.float32, device=device),
                torch.full((B,), 1.0/B, dtype=torch.float32, device=device),
                torch.empty(B, S, L, dtype=torch.float32, device=device)))
        return self._step_bufs[1]
