"""Parameter averaging: the traditional EMA and the post-hoc EMA of Karras et al.

Same classes, constructor arguments, methods and ``state_dict`` layout as the reference
(brever/modules/ema.py:32-274: ``EMA(model, beta)``, ``EMAKarras(model, sigma_rels)`` with
``update / store / restore / apply / state_dict / load_state_dict / post_hoc_ema`` and the
static ``sigma_rel_to_gamma`` / ``solve_weights``). The running averages of parameters that
live on a ROCm device are updated by ``brv_ema_update`` (rounded operation by operation like
the reference's tensor expression, so the averages are bit-identical); averages of CPU
parameters (a model that has not been moved yet) use the same expression in torch. The
post-hoc reconstruction is host-side linear algebra on a handful of scalars (float64) followed
by weighted sums of checkpointed tensors.
"""
import os

import numpy as np
import torch

from .. import hip


def _ema_step(ema_param, param, beta):
    if ema_param.is_cuda and ema_param.dtype == torch.float32 and ema_param.is_contiguous():
        src = param.detach().contiguous()
        hip.check(hip.lib().brv_ema_update(hip.ptr(ema_param), hip.ptr(src), float(1 - beta),
                                           ema_param.numel(), hip.stream()), 'brv_ema_update')
    else:
        ema_param += (1 - beta)*(param.detach() - ema_param)


class _Averager:
    def _params(self):
        return list(self.model.parameters())

    def _update(self, ema_params, beta):
        with torch.no_grad():
            for param, ema_param in zip(self._params(), ema_params):
                _ema_step(ema_param, param, beta)

    def store(self):
        self.stored_params = [p.detach().clone() for p in self._params()]

    def restore(self):
        if self.stored_params is None:
            raise RuntimeError('no stored parameters')
        self._assign(self.stored_params)
        self.stored_params = None

    def _assign(self, tensors):
        with torch.no_grad():
            for param, src in zip(self._params(), tensors):
                param.data.copy_(src.data)
        if hasattr(self.model, 'mark_params_changed'):      # flat-parameter HIP models
            self.model.mark_params_changed()

    def state_dict(self):
        return {name: getattr(self, name) for name in self._state_dict_attrs}

    def load_state_dict(self, state_dict):
        assert set(state_dict) == set(self._state_dict_attrs)
        for name, value in state_dict.items():
            setattr(self, name, value)


class EMA(_Averager):
    """Traditional exponential moving average with a fixed ``beta``."""

    def __init__(self, model, beta=0.999):
        assert 0.0 < beta < 1.0
        self.model = model
        self.beta = beta
        self.ema_params = [p.detach().clone() for p in model.parameters()]
        self.stored_params = None
        self._state_dict_attrs = ['ema_params']

    def update(self):
        self._update(self.ema_params, self.beta)

    def apply(self):
        self._assign(self.ema_params)


class EMAKarras(_Averager):
    """Power-function EMA profiles that can be recombined after training into any other
    profile (T. Karras et al., "Analyzing and Improving the Training Dynamics of Diffusion
    Models", 2023: algorithms 2 and 3)."""

    def __init__(self, model, sigma_rels=[0.05, 0.1]):
        assert all(0.0 < s < 1.0 for s in sigma_rels)
        self.model = model
        self.sigma_rels = sigma_rels
        self.ema_params = {s: [p.detach().clone() for p in model.parameters()]
                           for s in sigma_rels}
        self.stored_params = None
        self._num_updates = 0
        self._gammas = {s: self.sigma_rel_to_gamma(s) for s in sigma_rels}
        self._state_dict_attrs = ['ema_params', '_num_updates', '_gammas']

    def update(self):
        self._num_updates += 1
        for s in self.sigma_rels:
            beta = (1 - 1/self._num_updates)**(self._gammas[s] + 1)
            self._update(self.ema_params[s], beta)

    def apply(self, sigma_rel):
        self._assign(self.ema_params[sigma_rel])

    @staticmethod
    def sigma_rel_to_gamma(sigma_rel):
        """Exponent of the power-function profile with relative standard deviation
        ``sigma_rel``: largest real root of g^3 + 7 g^2 + (16 - t) g + (12 - t), t = sigma_rel^-2."""
        t = sigma_rel**-2
        return np.roots([1, 7, 16 - t, 12 - t]).real.max()

    @staticmethod
    def solve_weights(t_i, gamma_i, t_r, gamma_r):
        """Least-squares weights that express the target profiles (t_r, gamma_r) in the stored
        ones (t_i, gamma_i): solve A X = B with the profiles' inner products."""
        def inner(t_a, g_a, t_b, g_b):
            expo = np.where(t_a < t_b, g_b, -g_a)
            return (g_a + 1)*(g_b + 1)*(t_a/t_b)**expo/((g_a + g_b + 1)*np.maximum(t_a, t_b))
        col = lambda v: np.float64(v).reshape(-1, 1)       # noqa: E731
        row = lambda v: np.float64(v).reshape(1, -1)       # noqa: E731
        A = inner(col(t_i), col(gamma_i), row(t_i), row(gamma_i))
        B = inner(col(t_i), col(gamma_i), row(t_r), row(gamma_r))
        return np.linalg.solve(A, B)

    def post_hoc_ema(self, ckpts_or_ckpt_dir, sigma_rel_r, t_r=None, extension='.ckpt',
                     state_dict_key=None, apply=True):
        """Averaged parameters for target profile(s) ``sigma_rel_r`` (at update step(s) ``t_r``,
        default: the latest) from the profiles stored in checkpoints; see the reference's
        docstring (ema.py:160-190) for the argument conventions, which are kept."""
        if isinstance(ckpts_or_ckpt_dir, str):
            ckpts = [os.path.join(ckpts_or_ckpt_dir, f) for f in os.listdir(ckpts_or_ckpt_dir)
                     if f.endswith(extension)]
            if not ckpts:
                raise ValueError(f'no {extension} file in {ckpts_or_ckpt_dir}')
        else:
            ckpts = ckpts_or_ckpt_dir
        sigma_was_list, t_was_list = isinstance(sigma_rel_r, list), isinstance(t_r, list)
        if not sigma_was_list:
            sigma_rel_r = [sigma_rel_r]*(len(t_r) if t_was_list else 1)
        if not all(isinstance(s, float) for s in sigma_rel_r):
            raise TypeError('sigma_rel_r must be a float or a list of floats')
        if not all(0.0 < s < 1.0 for s in sigma_rel_r):
            raise ValueError('sigma_rel_r values must be strictly in [0, 1]')
        if t_r is not None:
            if not t_was_list:
                t_r = [t_r]*len(sigma_rel_r)
            if not all(isinstance(t, int) for t in t_r):
                raise TypeError('t_r must be an int or a list of ints')
            if len(t_r) != len(sigma_rel_r):
                raise ValueError('gamma_r and t_r must have the same length')
        if apply and len(sigma_rel_r) > 1:
            raise ValueError('cannot apply multiple EMA profiles to the model')
        stored, t_i, gamma_i = [], [], []
        for ckpt in ckpts:
            state = torch.load(ckpt, weights_only=False)
            if state_dict_key is not None:
                if state_dict_key not in state:
                    raise ValueError(f"no '{state_dict_key}' key in {ckpt}")
                state = state[state_dict_key]
            for s in self.sigma_rels:
                if s not in state['ema_params']:
                    raise ValueError(f'no averaged parameters for sigma_rel={s} in {ckpt}')
                stored.append(state['ema_params'][s])
                t_i.append(state['_num_updates'])
                gamma_i.append(state['_gammas'][s])
        if t_r is None:
            t_r = [max(t_i)]*len(sigma_rel_r)
        X = self.solve_weights(t_i, gamma_i, t_r, [self.sigma_rel_to_gamma(s) for s in sigma_rel_r])
        with torch.no_grad():
            out = [[sum(x.item()*p for x, p in zip(X[:, k], tensors)) for tensors in zip(*stored)]
                   for k in range(X.shape[1])]
        if apply:
            self._assign(out[0])
        return out if (sigma_was_list or t_was_list) else out[0]
