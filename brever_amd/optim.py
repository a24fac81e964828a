"""Flat-buffer Adam: clip_grad_norm_ + Adam.step as two HIP kernels.

Replaces, for models whose parameters are views of one flat fp32 buffer, the
per-tensor sequence of the reference's ``BreverBaseModel.update``
(brever/models/base.py:296-301): ``clip_grad_norm_(parameters, grad_clip)``
followed by ``torch.optim.Adam.step`` (343 tensors for Conv-TasNet). The
arithmetic runs in ``brv_clip_adam_step`` (``include/brever_hip.h``).

The class stays a ``torch.optim.Adam`` so that ``state_dict`` /
``load_state_dict`` keep the format the trainer checkpoints
(brever/training.py:414-416,440-441): per-parameter ``step``, ``exp_avg``,
``exp_avg_sq`` -- here views of two flat moment buffers.
"""
import torch

from . import hip


class FlatAdam(torch.optim.Adam):
    def __init__(self, params, owner, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps)
        self._owner = owner          # exposes flat_params() / flat_grads()
        self._exp_avg = None
        self._exp_avg_sq = None
        self._step_count = 0
        self._scratch = None
        self._slot = 0
        self.last_grad_norm = None

    # -- flat state ------------------------------------------------------------
    def _ensure_state(self):
        flat = self._owner.flat_params()
        if self._exp_avg is not None and self._exp_avg.device == flat.device \
                and self._exp_avg.numel() == flat.numel():
            return
        old_m, old_v = self._exp_avg, self._exp_avg_sq
        self._exp_avg = torch.zeros_like(flat)
        self._exp_avg_sq = torch.zeros_like(flat)
        if old_m is not None and old_m.numel() == flat.numel():
            self._exp_avg.copy_(old_m)
            self._exp_avg_sq.copy_(old_v)
        self._scratch = torch.zeros(64, dtype=torch.uint8, device=flat.device)
        self._slot = 0
        self.last_grad_norm = torch.zeros(1, dtype=torch.float32,
                                          device=flat.device)
        self._bind_state()

    def _bind_state(self):
        step = torch.tensor(float(self._step_count))
        for p, off in self._owner.param_offsets():
            n = p.numel()
            self.state[p] = {
                'step': step,
                'exp_avg': self._exp_avg[off:off + n].view(p.shape),
                'exp_avg_sq': self._exp_avg_sq[off:off + n].view(p.shape),
            }

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        flat = self._owner.flat_params()
        self._exp_avg = torch.zeros_like(flat)
        self._exp_avg_sq = torch.zeros_like(flat)
        self._step_count = 0
        for p, off in self._owner.param_offsets():
            st = self.state.get(p)
            if not st:
                continue
            n = p.numel()
            self._exp_avg[off:off + n].copy_(st['exp_avg'].reshape(-1))
            self._exp_avg_sq[off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
            self._step_count = int(float(st['step']))
        self._scratch = torch.zeros(64, dtype=torch.uint8, device=flat.device)
        self._slot = 0
        self.last_grad_norm = torch.zeros(1, dtype=torch.float32,
                                          device=flat.device)
        self._bind_state()

    # -- step --------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None, max_norm=0.0, grad_scale=1.0, grads2=None, grads=None):
        """One Adam step on the flat buffers.

        ``max_norm > 0`` folds ``clip_grad_norm_(max_norm)`` into the same
        launch; ``grad_scale`` multiplies the gradient first (``1/world_size``
        after a summing all-reduce); ``grads2``: a second flat gradient buffer (the
        other kernel chain's) that is added in the norm pass and left zeroed; ``grads``: the
        flat gradient if the caller has gathered (and all-reduced) it already.
        Two launches in all (``brv_clip_adam_step2``).
        """
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        flat = self._owner.flat_params()
        hip.require_device(flat)
        self._ensure_state()
        if grads is None:
            grads = self._owner.gather_grads()
        group = self.param_groups[0]
        beta1, beta2 = group['betas']
        slot = self._slot
        step = self._step_count + 1      # (committed after the launch: a failed step retried keeps its bias correction)
        try:
            hip.check(hip.lib().brv_clip_adam_step2(
                hip.ptr(flat), hip.ptr(grads), hip.ptr(grads2), hip.ptr(self._exp_avg),
                hip.ptr(self._exp_avg_sq), flat.numel(), float(grad_scale),
                float(max_norm), float(group['lr']), float(beta1), float(beta2),
                float(group['eps']), step, hip.ptr(self._scratch), slot,
                hip.ptr(self.last_grad_norm), hip.stream()), 'brv_clip_adam_step2')
        except Exception:
            # the accumulators' zero / non-zero state is unknown after a failed launch: start over
            self._scratch.zero_()
            self._slot = 0
            raise
        self._step_count = step
        next(iter(self.state.values()))['step'].fill_(float(step))
        self._slot = 1 - slot           # the kernel zeroed the other accumulator for the next call
        self._owner.mark_params_changed()
        return loss
