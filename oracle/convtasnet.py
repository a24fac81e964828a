"""ORACLE (test infrastructure, not product code).

CPU restatement in plain PyTorch of the reference Conv-TasNet,
brever/models/convtasnet/convtasnet.py:19-268, written against ``torch`` ops
only. It serves three purposes: (1) checker for the HIP path in ``tests/`` and
``__graft_entry__.smoke()``; (2) the ``cpu_baseline`` leg of ``bench.py``;
(3) the CPU model used to pin the host-side trainer against the reference's
post-training golden parameters. The product path (``brever_amd``) never
imports it.

Pinning: ``tests/golden/convtasnet_*.npz`` (forward output, loss and gradients
of the imported reference for seeded weights/inputs; the tiny 2-epoch training
golden of the reference's tests/test_training.py:83-94) -- see
``tests/golden/make_golden.py`` and ``tests/test_oracle.py``.

Contract notes:
* construction order = RNG consumption order of the reference:
  encoder.conv, decoder.trans_conv, tcn.layer_norm, tcn.bottleneck_conv, per
  block conv, d_conv, res_conv, skip_conv, norm_1, norm_2, prelu_1, prelu_2,
  then tcn.prelu, tcn.output_conv (convtasnet.py:49-61,159-186,209-238);
* ``state_dict`` keys/shapes equal the reference's (343 tensors at defaults);
* gLN statistics include the zero-padded tail of shorter batch items.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from brever_amd.models.base import BreverBaseModel

from . import criterion as ref_criterion


def _rb(x, on):
    """bf16 rounding point (value stays fp32) used by the bf16-emulating mode."""
    return x.bfloat16().float() if on else x


class CumulativeLayerNorm(nn.Module):
    """cLN: statistics over channels and all frames up to t
    (brever/modules/normalization.py:5-62 with num_groups=1)."""

    def __init__(self, channels, eps=1e-8):
        super().__init__()
        self.eps = eps
        self.gain = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))

    def forward(self, x):                      # (B, C, T)
        C = x.shape[1]
        count = C*torch.arange(1, x.shape[-1] + 1, device=x.device,
                               dtype=x.dtype).view(1, 1, -1)
        mean = x.sum(1, keepdim=True).cumsum(-1)/count
        var = x.pow(2).sum(1, keepdim=True).cumsum(-1)/count - mean.pow(2)
        x = (x - mean)/(var + self.eps).sqrt()
        return x*self.gain.view(1, -1, 1) + self.bias.view(1, -1, 1)


def _norm(causal, channels):
    if causal:
        return CumulativeLayerNorm(channels, eps=1e-8)
    return nn.GroupNorm(1, channels, eps=1e-8)


class _Encoder(nn.Module):
    def __init__(self, filters, filter_length):
        super().__init__()
        self.filter_length = filter_length
        self.stride = filter_length//2
        self.conv = nn.Conv1d(1, filters, filter_length, stride=self.stride,
                              bias=False)


class _Decoder(nn.Module):
    def __init__(self, filters, filter_length):
        super().__init__()
        self.filter_length = filter_length
        self.stride = filter_length//2
        self.trans_conv = nn.ConvTranspose1d(filters, 1, filter_length,
                                             stride=self.stride, bias=False)


class _Block(nn.Module):
    def __init__(self, bn, hidden, skip, kernel_size, dilation, causal, last):
        super().__init__()
        self.kernel_size, self.dilation, self.causal = \
            kernel_size, dilation, causal
        self.conv = nn.Conv1d(bn, hidden, 1)
        self.d_conv = nn.Conv1d(hidden, hidden, kernel_size,
                                dilation=dilation, groups=hidden)
        self.res_conv = None if last else nn.Conv1d(hidden, bn, 1)
        self.skip_conv = nn.Conv1d(hidden, skip, 1)
        self.norm_1 = _norm(causal, hidden)
        self.norm_2 = _norm(causal, hidden)
        self.prelu_1 = nn.PReLU()
        self.prelu_2 = nn.PReLU()


class _TCN(nn.Module):
    def __init__(self, filters, bn, hidden, skip, kernel_size, layers,
                 repeats, sources, causal):
        super().__init__()
        self.sources = sources
        self.layer_norm = _norm(causal, filters)
        self.bottleneck_conv = nn.Conv1d(filters, bn, 1)
        self.conv_blocks = nn.ModuleList()
        for r in range(repeats):
            for i in range(layers):
                last = r == repeats - 1 and i == layers - 1
                self.conv_blocks.append(_Block(bn, hidden, skip, kernel_size,
                                               2**i, causal, last))
        self.prelu = nn.PReLU()
        self.output_conv = nn.Conv1d(skip, filters*sources, 1)


class OracleConvTasNet(BreverBaseModel):
    """Same constructor signature, parameter names and numerics as the
    reference ``ConvTasNet`` (convtasnet.py:30-64).

    ``emulate_bf16=True`` (``'fused'``: the rounding points of the fused forward, the
    default for the default widths) inserts bf16 rounding at the points where the HIP
    path stores bf16 tensors or feeds bf16 MFMA operands (DESIGN.md "numerics")
    so the kernels can be compared at tight tolerance; it is off for everything
    that pins the oracle against the reference.
    """

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
        emulate_bf16: bool = False,
    ):
        if isinstance(criterion, str):
            criterion = ref_criterion.CRITERIA[criterion]
        super().__init__(criterion=criterion)
        self.encoder = _Encoder(filters, filter_length)
        self.decoder = _Decoder(filters, filter_length)
        self.tcn = _TCN(filters, bottleneck_channels, hidden_channels,
                        skip_channels, kernel_size, layers, repeats,
                        output_sources, causal)
        self.optimizer = self.init_optimizer(optimizer, lr=learning_rate)
        self.grad_clip = grad_clip
        self.emulate_bf16 = emulate_bf16
        self.trace = None          # set to a dict to capture intermediates

    def _tap(self, name, value):
        if self.trace is not None:
            self.trace[name] = value.detach()

    # ---- forward -----------------------------------------------------------
    def encode(self, x):
        K, hop = self.encoder.filter_length, self.encoder.stride
        x = F.pad(x, (0, (K - x.shape[-1]) % hop))          # convtasnet.py:115-120
        e = self.emulate_bf16
        w = F.conv1d(_rb(x, e).unsqueeze(1), _rb(self.encoder.conv.weight, e),
                     stride=hop)
        w = _rb(w, e)
        self._tap('w', w)
        return w

    def _block(self, blk, x, index=0):
        e = self.emulate_bf16
        z1 = _rb(F.conv1d(x, _rb(blk.conv.weight, e), blk.conv.bias), e)
        self._tap(f'z1.{index}', z1)
        # the causal HIP path keeps the cLN output in HBM as bf16; the non-causal one applies
        # the norm in fp32 while loading z1
        h = _rb(blk.norm_1(blk.prelu_1(z1)), e and blk.causal)
        pad = (blk.kernel_size - 1)*blk.dilation
        left = pad if blk.causal else pad//2                # convtasnet.py:244-251
        h = F.pad(h, (left, pad - left))
        z2 = _rb(F.conv1d(h, blk.d_conv.weight, blk.d_conv.bias,
                          dilation=blk.dilation, groups=h.shape[1]), e)
        self._tap(f'z2.{index}', z2)
        if e == 'fused' and not blk.causal:
            return self._block_tail_fused(blk, x, z2)
        h = _rb(blk.norm_2(blk.prelu_2(z2)), e)
        out = None
        if blk.res_conv is not None:
            out = _rb(x + F.conv1d(h, _rb(blk.res_conv.weight, e),
                                   blk.res_conv.bias), e)
        skip = F.conv1d(h, _rb(blk.skip_conv.weight, e), blk.skip_conv.bias)
        return out, skip

    def _block_tail_fused(self, blk, x, z2):
        """Rounding points of the fused forward (csrc/dwpw2_fused.cuh + the lazy A staging of
        gemm_ws.cuh): W gLN(p) = rstd (W gamma) p + (b + W beta - mean rstd (W gamma) 1); the
        product runs on bf16 p with bf16 gamma-folded weights and is stored as bf16 (u); the
        consumers apply the per-item scale and offset in fp32."""
        p = blk.prelu_2(z2)
        n2 = blk.norm_2
        mean = p.mean((1, 2), keepdim=True)
        rstd = (p.var((1, 2), unbiased=False, keepdim=True) + n2.eps).rsqrt()
        outs = []
        for conv in (blk.res_conv, blk.skip_conv):
            if conv is None:
                outs.append(None)
                continue
            wg = _rb(conv.weight*n2.weight.view(1, -1, 1), True)
            u = _rb(F.conv1d(_rb(p, True), wg), True)
            v0 = conv.bias + conv.weight[:, :, 0] @ n2.bias
            v1 = wg.sum((1, 2))
            outs.append(rstd*u + (v0.view(1, -1, 1) - mean*rstd*v1.view(1, -1, 1)))
        out = None if outs[0] is None else _rb(x + outs[0], True)
        return out, outs[1]

    def separate(self, w):
        e = self.emulate_bf16
        tcn = self.tcn
        x = _rb(tcn.layer_norm(w), e)
        x = _rb(F.conv1d(x, _rb(tcn.bottleneck_conv.weight, e),
                         tcn.bottleneck_conv.bias), e)
        self._tap('x.0', x)
        skip_sum = 0
        for i, blk in enumerate(tcn.conv_blocks):
            x, skip = self._block(blk, x, i)
            if x is not None:
                self._tap(f'x.{i + 1}', x)
            skip_sum = skip_sum + skip
        self._tap('skip', skip_sum)
        h = _rb(tcn.prelu(skip_sum), e)
        m = torch.sigmoid(F.conv1d(h, _rb(tcn.output_conv.weight, e),
                                   tcn.output_conv.bias))
        self._tap('m', m)
        return m.view(w.shape[0], tcn.sources, w.shape[1], w.shape[2])

    def decode(self, w, masks):
        e = self.emulate_bf16
        B, S, C, T = masks.shape
        y = _rb(w.unsqueeze(1)*masks, e).view(B*S, C, T)
        out = F.conv_transpose1d(y, _rb(self.decoder.trans_conv.weight, e),
                                 stride=self.decoder.stride)
        return out.view(B, S, -1)

    def forward(self, x):
        length = x.shape[-1]
        w = self.encode(x)
        masks = self.separate(w)
        return self.decode(w, masks)[:, :, :length]

    # ---- plugin surface ----------------------------------------------------
    def transform(self, sources):
        return sources.mean(axis=-2)

    def loss(self, batch, lengths, use_amp):
        inputs, labels = batch[:, 0], batch[:, 1:]
        device = batch.device.type
        dtype = torch.bfloat16 if device == 'cpu' else torch.float16
        with torch.autocast(device_type=device, dtype=dtype, enabled=use_amp):
            outputs = self(inputs)
            loss = self.criterion(outputs, labels, lengths)
        return loss.mean()

    def update(self, loss, scaler):
        super().update(loss, scaler, grad_clip=self.grad_clip)

    def _enhance(self, x, use_amp):
        x = x.mean(axis=-2)
        device = x.device.type
        dtype = torch.bfloat16 if device == 'cpu' else torch.float16
        with torch.autocast(device_type=device, dtype=dtype, enabled=use_amp):
            return self.forward(x)
