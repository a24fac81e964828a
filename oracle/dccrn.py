"""ORACLE (test infrastructure, not product code).

CPU restatement in plain PyTorch of the reference DCCRN, brever/models/dccrn/dccrn.py:28-358
(forward / apply_mask / mask network / complex wrappers / LSTM block), with STFT.forward /
backward of brever/modules/stft.py:59-138 (hann, normalized). Only ``tests/`` import it.

Pinning: tests/golden/dccrn.npz from the imported reference (tests/golden/make_golden.py):
parameter count, forward output in train mode (batch statistics, running estimates after
the step) and in eval mode on a seeded input.

``OracleDCCRN(emulate_bf16=True)`` restates the arithmetic of the HIP path under ``use_amp`` (brever_amd/models/
dccrn.py): the operands of the convolutions' three matrix products (y from x and W, dx from dy and W, dW from dy and
x) and of the LSTM input projections (gates from x and W_ih, dx and dW_ih from the gate gradients) are rounded to
bf16, and for lstm_channels = 128 (the width the HIP path runs on the bf16 matrix pipe) the hidden state and W_hh in
the recurrent product of every step and the gate gradients and W_hh in its adjoint; everything else -- accumulation, bias
sums, batch norms, the gate math and cell state, the recurrences of other widths, the Linear layers -- stays fp32. It is
the yardstick of tests/test_gpu_sizes.py (HIP error <= 1.5 x this emulation's own error, per tensor); with the flag off the module
is the pinned fp32 restatement, bit for bit.
"""
import math

import scipy.signal
import torch
import torch.nn as nn
import torch.nn.functional as F


def _stft(x, n, hop):
    w = torch.from_numpy(scipy.signal.get_window('hann', n)).to(x.dtype)
    L = x.shape[-1]
    frames = math.ceil(max(L - n, 0)/hop) + 1
    x = F.pad(x, (0, (frames - 1)*hop + n - L))
    X = torch.stft(x, n_fft=n, hop_length=hop, window=w, center=True, pad_mode='constant',
                   normalized=False, onesided=True, return_complex=True)
    return X/w.pow(2).sum().sqrt()


def _istft(X, n, hop):
    w = torch.from_numpy(scipy.signal.get_window('hann', n)).to(X.real.dtype)
    return torch.istft(X*w.pow(2).sum().sqrt(), n_fft=n, hop_length=hop, window=w, center=True,
                       normalized=False, onesided=True)


class _RoundForward(torch.autograd.Function):
    """bf16 rounding of a matrix-product operand; the gradient passes unchanged."""
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBackward(torch.autograd.Function):
    """Identity whose incoming gradient is rounded to bf16 (the dy operand of the two backward products)."""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


_qf, _qb = _RoundForward.apply, _RoundBackward.apply


class ComplexWrapper(nn.Module):                                   # dccrn.py:221-231
    emulate_bf16 = False
    round_output = False        # emulation: the product + bias is stored as bf16 (round 6: the blocks whose batch norm reads
                                # a bf16 convolution output on the HIP path -- what torch.autocast makes of a convolution)

    def __init__(self, module_cls, *args, **kwargs):
        super().__init__()
        self.module_real = module_cls(*args, **kwargs)
        self.module_imag = module_cls(*args, **kwargs)

    def _emulated(self, x):
        """The four real convolutions with bf16-rounded operands; the biases join in fp32 behind the point where
        the gradient is rounded (the bias gradient of the HIP path sums the unrounded dy)."""
        mr, mi = self.module_real, self.module_imag
        xr, xi = (_qf(t) for t in torch.chunk(x, 2, dim=1))
        wr, wi = _qf(mr.weight), _qf(mi.weight)
        if isinstance(mr, nn.ConvTranspose2d):
            def f(t, w):
                return F.conv_transpose2d(t, w, None, mr.stride, mr.padding, mr.output_padding)
        else:
            def f(t, w):
                return F.conv2d(t, w, None, mr.stride, mr.padding)
        prod = _qb(torch.cat([f(xr, wr) - f(xi, wi), f(xi, wr) + f(xr, wi)], dim=1))
        bias = torch.cat([mr.bias - mi.bias, mr.bias + mi.bias]).view(1, -1, 1, 1)
        return _qf(prod + bias) if self.round_output else prod + bias

    def forward(self, x):
        if self.emulate_bf16:
            return self._emulated(x)
        in_real, in_imag = torch.chunk(x, 2, dim=1)
        out_real = self.module_real(in_real) - self.module_imag(in_imag)
        out_imag = self.module_real(in_imag) + self.module_imag(in_real)
        return torch.cat([out_real, out_imag], dim=1)


class ComplexBatchNorm2d(nn.Module):
    """complex_batchnorm.py:29-215: centre, whiten with the inverse square root of the 2x2
    covariance of (real, imaginary) per channel, then a 2x2 affine map."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.eps, self.momentum = eps, momentum
        self.weight = nn.Parameter(torch.tensor([[1.0], [0.0], [1.0]]).repeat(1, num_features))
        self.bias = nn.Parameter(torch.zeros(2, num_features))
        self.register_buffer('running_mean', torch.zeros(2, num_features))
        self.register_buffer('running_var', torch.eye(2).unsqueeze(-1).repeat(1, 1, num_features))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))

    def forward(self, x):
        xr, xi = torch.chunk(x, 2, dim=1)
        t = torch.stack([xr, xi])                                  # (2, B, C, H, W)
        ax = (1, 3, 4)
        if self.training:
            self.num_batches_tracked += 1
            mean = t.mean(dim=ax)
            self.running_mean += self.momentum*(mean.detach() - self.running_mean)
        else:
            mean = self.running_mean
        t = t - mean[:, None, :, None, None]
        if self.training:
            var = (t*t).mean(dim=ax) + self.eps
            vrr, vii = var[0], var[1]
            vri = (t[0]*t[1]).mean(dim=(0, 2, 3))
            cov = torch.stack([vrr, vri, vri, vii]).detach().reshape(2, 2, -1)
            self.running_var += self.momentum*(cov - self.running_var)
        else:
            vrr, vri, _, vii = self.running_var.reshape(4, -1)
        s = torch.sqrt(vrr*vii - vri*vri)
        den = torch.sqrt(vrr + vii + 2*s)*s
        p, q, r, s2 = (vii + s)/den, -vri/den, -vri/den, (vrr + s)/den
        sh = (1, -1, 1, 1)
        z0 = t[0]*p.view(sh) + t[1]*r.view(sh)
        z1 = t[0]*q.view(sh) + t[1]*s2.view(sh)
        w, b = self.weight, self.bias
        return torch.cat([z0*w[0].view(sh) + z1*w[1].view(sh) + b[0].view(sh),
                          z0*w[1].view(sh) + z1*w[2].view(sh) + b[1].view(sh)], dim=1)


class _Block(nn.Module):                                           # dccrn.py:234-290
    def __init__(self, cls, cin, cout, norm=True, activation=True, complex_bn=False, **kw):
        super().__init__()
        self.conv = ComplexWrapper(cls, in_channels=cin, out_channels=cout, **kw)
        self.norm = (ComplexBatchNorm2d(cout) if complex_bn else nn.BatchNorm2d(2*cout)) \
            if norm else None
        self.activation = nn.PReLU() if activation else None

    def forward(self, x):
        x = self.conv(x)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


class _ComplexLSTMLayer(ComplexWrapper):                           # dccrn.py:330-358
    @staticmethod
    def _lstm_emulated(m, x):
        """nn.LSTM whose input projection runs on bf16-rounded operands: the projection is formed here and handed
        to the same LSTM through an identity input matrix (exact in fp32: one non-zero term per sum)."""
        gates = _qb(_qf(x) @ _qf(m.weight_ih_l0).t())
        if m.hidden_size == 128:
            # H = 128 is the width whose recurrent product the HIP path runs on the bf16 matrix pipe under use_amp
            # (csrc/dccrn.hip lstm_fwd_mv_kernel / lstm_bwd_mv_kernel, round 5): the hidden state and W_hh are
            # rounded as operands of the step's product, the gate gradients and W_hh in its adjoint; accumulation,
            # gate math and the cell state stay fp32. An explicit loop (the same one tests/test_gpu.py checks the
            # kernels against), differentiated by autograd: _qb rounds the gradient that enters the product.
            B, T, H = x.shape[0], x.shape[1], m.hidden_size
            w = _qf(m.weight_hh_l0)
            bias = m.bias_ih_l0 + m.bias_hh_l0
            h = gates.new_zeros(B, H)
            c = gates.new_zeros(B, H)
            ys = []
            for t in range(T):
                pre = gates[:, t] + bias + _qb(_qf(h) @ w.t())
                i, f, g, o = pre.chunk(4, dim=-1)
                c = torch.sigmoid(f)*c + torch.sigmoid(i)*torch.tanh(g)
                h = torch.sigmoid(o)*torch.tanh(c)
                ys.append(h)
            return torch.stack(ys, dim=1)
        eye = torch.eye(gates.shape[-1], dtype=gates.dtype)
        out, _, _ = torch._VF.lstm(gates, (gates.new_zeros(1, x.shape[0], m.hidden_size),
                                        gates.new_zeros(1, x.shape[0], m.hidden_size)),
                                [eye, m.weight_hh_l0, m.bias_ih_l0, m.bias_hh_l0], True, 1, 0.0, m.training, False,
                                True)
        return out

    def forward(self, real, imag):
        if self.emulate_bf16:
            mr, mi, f = self.module_real, self.module_imag, self._lstm_emulated
            return f(mr, real) - f(mi, imag), f(mr, imag) + f(mi, real)
        rr, _ = self.module_real(real)
        ii, _ = self.module_imag(imag)
        ri, _ = self.module_real(imag)
        ir, _ = self.module_imag(real)
        return rr - ii, ri + ir


class _ComplexLSTM(nn.Module):
    def __init__(self, input_size, hidden_size, num_layers):
        super().__init__()
        self.layers = nn.ModuleList(
            _ComplexLSTMLayer(nn.LSTM, input_size=input_size if i == 0 else hidden_size,
                              hidden_size=hidden_size, batch_first=True, bidirectional=False)
            for i in range(num_layers))


class _LSTMBlock(nn.Module):                                       # dccrn.py:293-311
    def __init__(self, input_size, hidden_size, num_layers):
        super().__init__()
        self.lstm = _ComplexLSTM(input_size, hidden_size, num_layers)
        self.linear_r = nn.Linear(hidden_size, input_size)
        self.linear_i = nn.Linear(hidden_size, input_size)

    emulate_bf16 = False        # the two output projections on bf16-rounded operands (round 6: the HIP path runs them on
                                # the bf16 MFMA under use_amp, as torch.autocast runs nn.Linear)

    def _linear(self, m, x):
        if not self.emulate_bf16:
            return m(x)
        return _qb(_qf(x) @ _qf(m.weight).t()) + m.bias

    def forward(self, x):
        real, imag = torch.chunk(x, 2, dim=-1)
        for layer in self.lstm.layers:
            real, imag = layer(real, imag)
        return torch.cat([self._linear(self.linear_r, real), self._linear(self.linear_i, imag)], dim=-1)


class _MaskNet(nn.Module):                                         # dccrn.py:145-218
    def __init__(self, input_dim, channels, kernel_size, stride, padding, output_padding,
                 lstm_channels, lstm_layers, complex_bn=False):
        super().__init__()
        kw = dict(kernel_size=kernel_size, stride=stride, padding=padding)
        self.encoder = nn.ModuleList(
            _Block(nn.Conv2d, 1 if i == 0 else channels[i - 1], channels[i], complex_bn=complex_bn,
                   **kw)
            for i in range(len(channels)))
        self.decoder = nn.ModuleList(
            _Block(nn.ConvTranspose2d, channels[i]*2, 1 if i == 0 else channels[i - 1],
                   norm=i != 0, activation=i != 0, output_padding=output_padding,
                   complex_bn=complex_bn, **kw)
            for i in range(len(channels) - 1, -1, -1))
        dim = input_dim
        for _ in channels:
            dim = (dim + 2*padding[0] - kernel_size[0])//stride[0] + 1
        self.lstm = _LSTMBlock(channels[-1]*dim, lstm_channels, lstm_layers)

    round_grads = False     # emulation: the gradient with respect to a bf16 activation is bf16 at each of its consumers (round 6)

    def forward(self, x):
        q = _qb if self.round_grads else (lambda t: t)
        outs = []
        for k, blk in enumerate(self.encoder):
            x = blk(q(x) if k > 0 else x)          # (the inputs of encoder blocks 2.. are bf16 activations)
            outs.append(x)
        x = x.permute(0, x.ndim - 1, *range(1, x.ndim - 1))
        x = self.lstm(x.reshape(*x.shape[:2], -1)).reshape(*x.shape)
        x = x.permute(0, *range(2, x.ndim), 1)
        for k, (blk, enc) in enumerate(zip(self.decoder, reversed(outs))):
            if k > 0:                               # (the first decoder block reads fp32 tensors: the recurrent block's
                x, enc = q(x), q(enc)               # output and the last encoder block's)
            real, imag = x.chunk(2, dim=1)
            sr, si = enc.chunk(2, dim=1)
            x = blk(torch.cat([real, sr, imag, si], dim=1))
        return x


class OracleDCCRN(nn.Module):
    def __init__(self, n=512, hop=128, channels=(16, 32, 64, 128, 128, 128), kernel_size=(5, 2),
                 stride=(2, 1), padding=(2, 0), output_padding=(1, 0), lstm_channels=128,
                 lstm_layers=2, use_complex_batchnorm=False, emulate_bf16=False):
        super().__init__()
        self.n, self.hop = n, hop
        self.emulate_bf16 = emulate_bf16
        self.mask_net = _MaskNet(n//2, list(channels), kernel_size, stride, padding,
                                 output_padding, lstm_channels, lstm_layers,
                                 complex_bn=use_complex_batchnorm)

    @staticmethod
    def apply_mask(x, mask):                                       # dccrn.py:96-109
        in_real, in_imag = torch.chunk(x, 2, dim=1)
        in_mag = (in_real**2 + in_imag**2).sqrt()
        in_phase = torch.atan2(in_imag, in_real)
        mask_real, mask_imag = torch.chunk(mask, 2, dim=1)
        mask_mag = (mask_real**2 + mask_imag**2 + 1e-7).sqrt().tanh()
        mask_real = mask_real + (mask_real == 0)*1e-7
        mask_phase = torch.atan2(mask_imag, mask_real)
        out_mag = in_mag*mask_mag
        out_phase = in_phase + mask_phase
        return torch.complex(out_mag*out_phase.cos(), out_mag*out_phase.sin())

    def forward(self, x):                                          # dccrn.py:83-94
        for m in self.modules():
            if isinstance(m, ComplexWrapper):
                m.emulate_bf16 = self.emulate_bf16
        # the HIP path (brever_amd/models/dccrn.py: _BlockFunction) stores the convolution output of a block as bf16 where
        # the block's batch norm writes bf16 too: every block with a norm but the first encoder block (its fp32 output
        # feeds the column-matrix weight gradient) and the last encoder block (fp32 for the recurrent block)
        self.mask_net.lstm.emulate_bf16 = bool(self.emulate_bf16)
        self.mask_net.round_grads = bool(self.emulate_bf16)
        enc, dec = self.mask_net.encoder, self.mask_net.decoder
        for k, blk in enumerate(enc):
            blk.conv.round_output = bool(self.emulate_bf16) and 0 < k < len(enc) - 1
        for blk in dec:
            blk.conv.round_output = bool(self.emulate_bf16) and getattr(blk, 'norm', None) is not None
        length = x.shape[-1]
        x = _stft(x, self.n, self.hop)[..., 1:, :]
        x = torch.stack([x.real, x.imag], dim=1)
        x = self.apply_mask(x, self.mask_net(x)).squeeze(1)
        x = F.pad(x, (0, 0, 1, 0))
        return _istft(x, self.n, self.hop)[..., :length]
