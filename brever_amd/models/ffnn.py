"""Feed-forward mask estimator on log-mel features, HIP path.

Same constructor signature, registry key (``ffnn``), state-dict names, ``transform`` /
``loss`` / ``_enhance`` / ``pre_train`` semantics as the reference
(brever/models/ffnn/ffnn.py:15-203). The STFT, the mel filterbank, the feature
compression, frame stacking, the ideal-ratio-mask labels, the input normalisers and the
MLP (Linear -> ReLU -> Dropout ... -> Linear -> Sigmoid) with its gradients run in
``libbrever_hip.so`` (``brv_stft_forward``, ``brv_matmul_f32``, ``brv_gemm_f32``,
``brv_fbe_power`` ...). PyTorch supplies device memory, the parameter containers
(``nn.Linear`` modules created in the reference's order, so a seeded init is identical)
and the dropout keep-masks (``torch.bernoulli`` on the device generator).

``transform`` computes on the device of its input; CPU tensors are moved to the model's
ROCm device and the result is returned on the caller's device (the reference allows
dataset workers to call it on CPU tensors, base.py:100-104).
"""
import logging

import numpy as np
import torch
import torch.nn as nn

from .. import hip
from ..modules.features import FeatureExtractor
from ..modules.stft import STFT, MelFilterbank
from .base import BreverBaseModel, ModelRegistry

eps = np.finfo(float).eps        # ffnn.py:12


def _gemm(a, b, d, batch, M, N, K, lda, ldb, ldd, a_bs, b_bs, d_bs, trans_a=0, trans_b=0,
          kbatch=1, a_kbs=0, b_kbs=0, bias=None, accumulate=0):
    hip.check(hip.lib().brv_gemm_f32(
        hip.ptr(a), hip.ptr(b), hip.ptr(d), batch, M, N, K, lda, ldb, ldd, a_bs, b_bs, d_bs,
        trans_a, trans_b, kbatch, a_kbs, b_kbs, hip.ptr(bias), accumulate, hip.stream()),
        'brv_gemm_f32')


class _LinearFunction(torch.autograd.Function):
    """y[b] (O, T) = W (O, I) @ x[b] (I, T) + bias[:, None]: nn.Linear on the feature axis
    of a (B, features, frames) tensor without the two transposes of ffnn.py:167-171."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        B, I, T = x.shape
        O = weight.shape[0]
        x = x.contiguous()
        y = torch.empty(B, O, T, dtype=torch.float32, device=x.device)
        _gemm(weight, x, y, B, O, T, I, I, T, T, 0, I*T, O*T, bias=bias)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        B, I, T = x.shape
        O = weight.shape[0]
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)                       # W^T (I, O) @ dy[b] (O, T)
            _gemm(weight, dy, dx, B, I, T, O, I, T, T, 0, O*T, I*T, trans_a=1)
        dw = torch.empty_like(weight)                      # sum_b dy[b] (O, T) @ x[b]^T (T, I)
        _gemm(dy, x, dw, 1, O, I, T, T, T, I, 0, 0, 0, trans_b=1, kbatch=B, a_kbs=O*T, b_kbs=I*T)
        db = torch.empty(O, dtype=torch.float32, device=x.device)
        hip.check(hip.lib().brv_row_sum(hip.ptr(dy), hip.ptr(db), B, O, T, hip.stream()),
                  'brv_row_sum')
        return dx, dw, db


class _ReluDropoutFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask, scale):
        x = x.contiguous()
        out = torch.empty_like(x)
        hip.check(hip.lib().brv_relu_dropout_forward(hip.ptr(x), hip.ptr(mask), hip.ptr(out),
                                                     x.numel(), float(scale), hip.stream()),
                  'brv_relu_dropout_forward')
        ctx.save_for_backward(x, mask)
        ctx.scale = float(scale)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, mask = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        hip.check(hip.lib().brv_relu_dropout_backward(hip.ptr(x), hip.ptr(mask), hip.ptr(dy),
                                                      hip.ptr(dx), x.numel(), ctx.scale,
                                                      hip.stream()), 'brv_relu_dropout_backward')
        return dx, None, None


class _SigmoidFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        hip.check(hip.lib().brv_sigmoid_forward(hip.ptr(x), hip.ptr(y), x.numel(), hip.stream()),
                  'brv_sigmoid_forward')
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        hip.check(hip.lib().brv_sigmoid_backward(hip.ptr(y), hip.ptr(dy), hip.ptr(dx), y.numel(),
                                                 hip.stream()), 'brv_sigmoid_backward')
        return dx


class _FFNN(nn.Module):
    """Parameter layout of the reference (ffnn.py:151-171): ``module_list`` holds
    Linear / ReLU / Dropout triples and a final Linear / Sigmoid, so state-dict keys are
    ``module_list.0.weight`` ... as in the reference."""

    def __init__(self, input_size, output_size, hidden_layers=[1024, 1024], dropout=0.2):
        super().__init__()
        self.input_size = input_size
        self.output_size = output_size
        self.dropout = dropout
        self.module_list = nn.ModuleList()
        start_size = input_size
        for end_size in hidden_layers:
            self.module_list.append(nn.Linear(start_size, end_size))
            self.module_list.append(nn.ReLU())
            self.module_list.append(nn.Dropout(dropout))
            start_size = end_size
        self.module_list.append(nn.Linear(start_size, output_size))
        self.module_list.append(nn.Sigmoid())

    def forward(self, x):
        hip.require_device(x)
        x = x.float()
        linears = [m for m in self.module_list if isinstance(m, nn.Linear)]
        for lin in linears[:-1]:
            x = _LinearFunction.apply(x, lin.weight, lin.bias)
            mask, scale = None, 1.0
            if self.training and self.dropout > 0:
                keep = 1.0 - self.dropout
                mask = torch.empty_like(x).bernoulli_(keep)
                scale = 1.0/keep
            x = _ReluDropoutFunction.apply(x, mask, scale)
        x = _LinearFunction.apply(x, linears[-1].weight, linears[-1].bias)
        return _SigmoidFunction.apply(x)


class StaticNormalizer(nn.Module):
    def __init__(self, input_size):
        super().__init__()
        self.register_buffer('mean', torch.zeros((input_size, 1)))
        self.register_buffer('std', torch.ones((input_size, 1)))

    def set_statistics(self, mean, std):
        self.mean[:], self.std[:] = mean, std

    def forward(self, x):
        hip.require_device(x)
        unbatched = x.ndim == 2
        x3 = (x.unsqueeze(0) if unbatched else x).float().contiguous()
        B, R, T = x3.shape
        out = torch.empty_like(x3)
        # locals keep both (possibly converted) tensors alive until the launch is queued
        mean, std = self.mean.float().contiguous(), self.std.float().contiguous()
        hip.check(hip.lib().brv_static_norm(hip.ptr(x3), hip.ptr(mean), hip.ptr(std), hip.ptr(out), B,
                                            R, T, hip.stream()), 'brv_static_norm')
        return out.squeeze(0) if unbatched else out


class CumulativeNormalizer(nn.Module):
    def __init__(self, eps=1e-4):
        super().__init__()
        self.eps = eps

    def forward(self, x):
        hip.require_device(x)
        x3 = x.float().contiguous()
        out = torch.empty_like(x3)
        T = x3.shape[-1]
        hip.check(hip.lib().brv_cumulative_norm(hip.ptr(x3), hip.ptr(out), x3.numel()//T, T,
                                                float(self.eps), hip.stream()),
                  'brv_cumulative_norm')
        return out


@ModelRegistry.register('ffnn')
class FFNN(BreverBaseModel):
    _fused_adam = True       # clip + Adam as brv_clip_adam_step2 on one flat buffer (models/base.py)

    def __init__(
        self,
        fs: int = 16000,
        features: set[str] = {'logfbe'},
        stacks: int = 5,
        decimation: int = 1,
        stft_frame_length: int = 512,
        stft_hop_length: int = 256,
        stft_window: str = 'hann',
        mel_filters: int = 64,
        hidden_layers: list[int] = [1024, 1024],
        dropout: float = 0.2,
        normalization: str = 'static',
        criterion: str = 'mse',
        optimizer: str = 'Adam',
        learning_rate: float = 0.0001,
    ):
        super().__init__(criterion=criterion)
        self.stacks = stacks
        self.decimation = decimation
        self.stft = STFT(frame_length=stft_frame_length, hop_length=stft_hop_length,
                         window=stft_window)
        self.mel_fb = MelFilterbank(n_filters=mel_filters, n_fft=stft_frame_length, fs=fs)
        self.feature_extractor = FeatureExtractor(features=features, mel_fb=self.mel_fb,
                                                  hop_length=stft_hop_length, fs=fs)
        input_size = self.feature_extractor.n_features*(stacks + 1)
        self.ffnn = _FFNN(input_size=input_size, output_size=mel_filters,
                          hidden_layers=hidden_layers, dropout=dropout)
        if normalization == 'static':
            self.normalization = StaticNormalizer(input_size)
        elif normalization == 'cumulative':
            self.normalization = CumulativeNormalizer()
        else:
            raise ValueError(f'unrecognized normalization type, got {normalization}')
        self.optimizer = self.init_optimizer(optimizer, lr=learning_rate)

    def forward(self, x):
        return self.ffnn(self.normalization(x))

    # -- device plumbing: compute where the HIP kernels can run --------------------------
    def _compute_device(self, x):
        if x.is_cuda:
            return x.device
        dev = next(self.parameters()).device
        if dev.type != 'cuda':
            if not torch.cuda.is_available():
                raise RuntimeError('the HIP path needs a ROCm device (no CPU fallback)')
            dev = torch.device('cuda', torch.cuda.current_device())
        return dev

    def transform(self, sources):
        assert sources.shape[0] == 2  # mixture, foreground
        home = sources.device
        sources = sources.to(self._compute_device(sources))
        spec = self.stft(sources)                      # (2, channels, bins, frames)
        mix, foreground = spec[0], spec[1]
        background = mix - foreground
        x = self.decimate(self.stack(self.feature_extractor(mix)))
        labels = self.decimate(self.irm(foreground, background))
        return torch.cat([x, labels]).to(home)

    def loss(self, batch, lengths, use_amp):
        inputs = batch[:, :self.ffnn.input_size]
        labels = batch[:, self.ffnn.input_size:]
        outputs = self(inputs)
        return self.criterion(outputs, labels, lengths).mean()

    def _enhance(self, x, use_amp):
        length = x.shape[-1]
        x = self.stft(x)                               # (B, channels, bins, frames)
        features = self.stack(self.feature_extractor(x))
        mask = self.ffnn(self.normalization(features))
        mask_extrapolated = self.mel_fb.backward(mask)
        B, C, bins, F = x.shape
        spec = torch.view_as_real(x.to(torch.complex64).contiguous())
        out = torch.empty(B, bins, F, 2, dtype=torch.float32, device=x.device)
        hip.check(hip.lib().brv_masked_mean_spec(
            hip.ptr(spec), hip.ptr(mask_extrapolated.float().contiguous()), hip.ptr(out), B, C,
            bins*F, hip.stream()), 'brv_masked_mean_spec')
        y = self.stft.backward(torch.view_as_complex(out))
        return y[..., :length]

    def irm(self, foreground, background):
        """(channels, bins, frames) complex spectra -> (mel_filters, frames) ideal ratio mask
        (ffnn.py:121-128)."""
        lib = hip.lib()
        C, bins, F = foreground.shape
        powers = []
        for s in (foreground, background):
            spec = torch.view_as_real(s.to(torch.complex64).contiguous())
            p = torch.empty(1, bins, F, dtype=torch.float32, device=s.device)
            hip.check(lib.brv_fbe_power(hip.ptr(spec), hip.ptr(p), 1, C, bins*F, hip.stream()),
                      'brv_fbe_power')
            powers.append(self.mel_fb(p[0]))
        out = torch.empty_like(powers[0])
        hip.check(lib.brv_irm(hip.ptr(powers[0]), hip.ptr(powers[1]), hip.ptr(out), out.numel(),
                              float(eps), hip.stream()), 'brv_irm')
        return out

    def stack(self, data):
        hip.require_device(data)
        unbatched = data.ndim == 2
        d3 = (data.unsqueeze(0) if unbatched else data).float().contiguous()
        B, nf, T = d3.shape
        out = torch.empty(B, (self.stacks + 1)*nf, T, dtype=torch.float32, device=data.device)
        hip.check(hip.lib().brv_stack_frames(hip.ptr(d3), hip.ptr(out), B, nf, T, self.stacks,
                                             hip.stream()), 'brv_stack_frames')
        return out.squeeze(0) if unbatched else out

    def decimate(self, data):
        return data[..., ::self.decimation]

    def pre_train(self, dataset, dataloader, epochs):
        if isinstance(self.normalization, StaticNormalizer):
            logging.info('Calculating training statistics')
            mean, var = 0, 0
            for i in range(len(dataset)):
                data = dataset[i]
                inputs = data[:self.ffnn.input_size]
                mean += inputs.mean(-1, keepdim=True)
                var += inputs.pow(2).mean(-1, keepdim=True)
            mean, var = mean/len(dataset), var/len(dataset)
            var -= mean.pow(2)
            self.normalization.set_statistics(mean, var.sqrt())
