"""Data parallelism: one process per GPU, gradients all-reduced every step.

RCCL over xGMI on MI355X (``torch.distributed`` backend ``nccl``); gloo in the
CPU tests. The reference only *wraps* its model in DistributedDataParallel and
never arms the reducer (brever/training.py:62-63,
326-329); the intended semantics -- mean gradient over ranks -- are implemented
here explicitly:

* parameters (and buffers) are broadcast once from rank 0;
* models that expose ``set_grad_sync`` (the HIP Conv-TasNet, one flat fp32 gradient
  buffer of 19.7 MB) run their backward pass in ``nparts`` parts (3 by default =
  one per TCN repeat, 6.6 MB each); the slice of the flat gradient a part finishes is
  all-reduced right away with ``async_op=True``: the collective runs on RCCL's own
  stream, ordered behind the kernels of that part, while the compute stream goes on
  with the next part. The compute stream waits for the collectives once, before the
  fused clip + Adam (``grad_scale = 1/world``). Buckets are views of the flat buffer:
  no packing copies. The generic ``loss -> update`` sequence of the same model (other
  criteria / optimizers) calls the hook once with the whole buffer;
* every other model whose parameters live in one flat buffer (``BreverBaseModel._flat_base``: DCCRN, TF-GridNet,
  SGMSE+, FFNN under plain Adam) hands its flat gradient to ``__call__`` from ``update``, AFTER backward has ended
  and after the side streams of that backward pass were joined (DCCRN's weight gradients): one summing all-reduce
  per step, the mean taken inside the fused clip + Adam kernel. No collective is issued from inside backward;
* a model without a flat buffer (another optimizer, a foreign ``nn.Module``) gets post-accumulate-grad hooks that
  average each ``.grad`` over ranks during backward, i.e. before clipping and the optimizer step. Kernels that
  produce parameter gradients off the main stream look for such hooks and stay in order (models/dccrn.py).
"""
import datetime
import os

import torch
import torch.distributed as dist


def init_process_group(backend, timeout_s=None, **kwargs):
    """``dist.init_process_group`` on 127.0.0.1 unless MASTER_ADDR says otherwise (one node).

    ``timeout_s`` is the timeout of EVERY collective of the group (NCCL watchdog, gloo operations), not only
    of the rendezvous: the train / test scripts leave it at torch's defaults (10 min NCCL, 30 min gloo, as the
    reference) -- a rank that scores its whole test shard before the final gather, or rank 0 writing a
    checkpoint before a barrier, may keep the others waiting for minutes. ``bench.py`` passes a short one
    (120 s, or ``BRV_DIST_TIMEOUT_S``) so that a bad rendezvous on a fresh node exits non-zero instead of
    hanging."""
    if timeout_s is None and 'BRV_DIST_TIMEOUT_S' in os.environ:
        timeout_s = float(os.environ['BRV_DIST_TIMEOUT_S'])
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if timeout_s is not None:
        kwargs['timeout'] = datetime.timedelta(seconds=float(timeout_s))
    return dist.init_process_group(backend, **kwargs)


def broadcast_parameters(model, src=0):
    """Make every rank start from rank ``src``'s weights (what DDP's constructor
    does in the reference, brever/training.py:63)."""
    with torch.no_grad():
        flat = getattr(model, 'flat_params', None)
        flat = flat() if callable(flat) else None
        if flat is not None:
            dist.broadcast(flat, src)
            model.mark_params_changed()
        else:
            for p in model.parameters():
                dist.broadcast(p.data, src)
        for b in model.buffers():
            dist.broadcast(b.data, src)


class GradSynchronizer:
    """``nparts``: gradient buckets of a flat-gradient model (1 = one all-reduce after
    the last backward kernel). ``exposed_ms()`` reports how long the compute stream
    waited for collectives in ``finish`` (GPU only; what overlap did not hide)."""

    def __init__(self, model, nparts=3):
        self.world = dist.get_world_size()
        self.nparts = max(1, int(nparts))
        self._pending = []
        self._waits = []
        self.calls = 0                     # collectives issued so far (tests, bench line)
        setter = getattr(model, 'set_grad_sync', None)
        # (Conv-TasNet's setter returns None; the base class answers False when there is no flat buffer)
        self.flat_model = callable(setter) and setter(self) is not False
        if not self.flat_model:
            self._install_hooks(model)

    # -- flat-gradient models -------------------------------------------------
    def __call__(self, flat_grad):
        """Single-bucket hook: sum over ranks, the mean is taken inside the optimizer
        kernel (``grad_scale``). Nothing overlaps this collective: all of it is exposed, and it is
        timed with the same event pair as ``finish`` so that ``exposed_ms`` says so."""
        timed = torch.cuda.is_available() and flat_grad.is_cuda and dist.get_backend() == 'nccl'
        if timed:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        dist.all_reduce(flat_grad)
        self.calls += 1
        if timed:
            b.record()
            self._waits.append((a, b))
        return 1.0/self.world

    def bucket(self, part, grad_slice):
        """Called after backward part ``part``: start the all-reduce of its slice."""
        self._pending.append(dist.all_reduce(grad_slice, async_op=True))
        self.calls += 1

    def finish(self):
        """Make the compute stream wait for the outstanding collectives."""
        timed = self._pending and torch.cuda.is_available() \
            and dist.get_backend() == 'nccl'
        if timed:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        for work in self._pending:
            work.wait()
        if timed:
            b.record()
            self._waits.append((a, b))
        self._pending = []
        return 1.0/self.world

    def exposed_ms(self, reset=True):
        """Mean time per step the compute stream spent waiting in ``finish``."""
        if not self._waits:
            return 0.0
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self._waits)/len(self._waits)
        if reset:
            self._waits = []
        return ms

    def _install_hooks(self, model):
        """Generic path: average each gradient over ranks as soon as autograd has
        accumulated it, i.e. before clip_grad_norm_ and the optimizer step see it."""
        world = self.world

        def hook(param):
            dist.all_reduce(param.grad)
            self.calls += 1
            param.grad /= world

        for p in model.parameters():
            if p.requires_grad:
                p.register_post_accumulate_grad_hook(hook)

    def train_step(self, model, batch, lengths, use_amp, scaler):
        return model.train_step(batch, lengths, use_amp, scaler)


def all_reduce_mean_grads(params, sync):
    """Mean over ranks of the ``.grad`` tensors of ``params`` as ONE collective on a packed copy (the generic
    branch of ``BreverBaseModel.update`` for a flat model: a sub-network, another optimizer). ``sync`` is the
    ``GradSynchronizer`` (sums, returns 1/world).

    EVERY parameter that requires a gradient takes its slot of the packed buffer, in ``params`` order, zeros where
    this rank has no ``.grad`` (a data-dependent branch, a sub-network unused on this rank): the ranks' buffers then
    have the same length and layout whatever their None sets are -- packing only the existing gradients made
    ``all_reduce`` hang or sum misaligned slices. A parameter without a gradient here receives the mean
    of the others' (as DDP does with ``find_unused_parameters``) when any rank had one."""
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    ref = next((p.grad for p in params if p.grad is not None), None)
    dtype = ref.dtype if ref is not None else params[0].dtype
    device = ref.device if ref is not None else params[0].device
    flat = torch.zeros(sum(p.numel() for p in params), dtype=dtype, device=device)
    off = 0
    have = torch.zeros(len(params), dtype=dtype, device=device)
    for i, p in enumerate(params):
        n = p.numel()
        if p.grad is not None:
            flat[off:off + n].copy_(p.grad.reshape(-1))
            have[i] = 1
        off += n
    packed = torch.cat([flat, have])             # (the presence flags ride in the same collective)
    packed.mul_(sync(packed))
    any_rank = (packed[flat.numel():] > 0).tolist()
    off = 0
    for i, p in enumerate(params):
        n = p.numel()
        if p.grad is not None:
            p.grad.copy_(packed[off:off + n].view(p.shape))
        elif any_rank[i]:
            p.grad = packed[off:off + n].view(p.shape).clone()
        off += n
