"""DCCRN (deep complex convolution recurrent network) on the HIP path -- forward values.

Same constructor signature, registry key (``dccrn``), module tree / state-dict names and
seeded initialisation as the reference (brever/models/dccrn/dccrn.py:28-358): the
``nn.Conv2d`` / ``nn.ConvTranspose2d`` / ``nn.BatchNorm2d`` / ``nn.PReLU`` / ``nn.LSTM`` /
``nn.Linear`` objects are created in the reference's order but only hold parameters; the
arithmetic (STFT, the four real convolutions of every complex layer, batch norm + PReLU, the
complex LSTM, the mask application, iSTFT) runs in ``libbrever_hip.so``.

Forward and backward are ``torch.autograd.Function`` pieces whose two sides call the HIP
kernels (``brv_conv2d_* / brv_conv_transpose2d_forward / brv_conv2d_wgrad``,
``brv_batchnorm2d_*``, ``brv_lstm_recurrent_*``, ``brv_gemm_f32``, ``brv_dccrn_apply_mask*``,
``brv_stft_forward`` / ``brv_istft_backward`` and its adjoint); torch only concatenates, slices
and transposes between them. fp32 throughout: correctness-first direct convolutions, not yet
tuned (the reference's autocast has no counterpart here, ``use_amp`` is ignored).
"""
import torch
import torch.nn as nn

from .. import hip
from ..modules.stft import STFT
from .base import BreverBaseModel, ModelRegistry


class _ParamOnly(nn.Module):
    def forward(self, *args, **kwargs):
        raise RuntimeError('this module only stores parameters; the compute runs in '
                           'libbrever_hip.so')


class ComplexWrapper(_ParamOnly):
    def __init__(self, module_cls, *args, **kwargs):
        super().__init__()
        self.module_real = module_cls(*args, **kwargs)
        self.module_imag = module_cls(*args, **kwargs)


class EncoderBlock(_ParamOnly):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding,
                 use_complex_batchnorm):
        super().__init__()
        self.conv = ComplexWrapper(nn.Conv2d, in_channels=in_channels, out_channels=out_channels,
                                   kernel_size=kernel_size, stride=stride, padding=padding)
        if use_complex_batchnorm:
            raise NotImplementedError('use_complex_batchnorm=True is not built yet on the HIP path')
        self.norm = nn.BatchNorm2d(2*out_channels)
        self.activation = nn.PReLU()


class DecoderBlock(_ParamOnly):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding,
                 use_complex_batchnorm, output_padding, norm=True, activation=True):
        super().__init__()
        self.conv = ComplexWrapper(nn.ConvTranspose2d, in_channels=in_channels,
                                   out_channels=out_channels, kernel_size=kernel_size,
                                   stride=stride, padding=padding, output_padding=output_padding)
        self.norm, self.activation = None, None
        if norm:
            if use_complex_batchnorm:
                raise NotImplementedError('use_complex_batchnorm=True is not built yet on the '
                                          'HIP path')
            self.norm = nn.BatchNorm2d(2*out_channels)
        if activation:
            self.activation = nn.PReLU()


class SingleLayerComplexLSTM(ComplexWrapper):
    def __init__(self, *args, **kwargs):
        super().__init__(nn.LSTM, *args, **kwargs)


class ComplexLSTM(_ParamOnly):
    def __init__(self, input_size, hidden_size, num_layers=1, **kwargs):
        super().__init__()
        self.layers = nn.ModuleList()
        for i in range(num_layers):
            self.layers.append(SingleLayerComplexLSTM(
                input_size=input_size if i == 0 else hidden_size, hidden_size=hidden_size,
                batch_first=True, bidirectional=False))


class LSTMBlock(_ParamOnly):
    def __init__(self, input_size, hidden_size, num_layers):
        super().__init__()
        self.lstm = ComplexLSTM(input_size=input_size, hidden_size=hidden_size,
                                num_layers=num_layers, batch_first=True, bidirectional=False)
        self.linear_r = nn.Linear(hidden_size, input_size)
        self.linear_i = nn.Linear(hidden_size, input_size)


class DCCRNMaskNet(_ParamOnly):
    def __init__(self, input_dim, channels, kernel_size, stride, padding, output_padding,
                 lstm_channels, lstm_layers, use_complex_batchnorm):
        super().__init__()
        self.geom = (tuple(kernel_size), tuple(stride), tuple(padding), tuple(output_padding))
        self.encoder = nn.ModuleList()
        for i in range(len(channels)):
            self.encoder.append(EncoderBlock(1 if i == 0 else channels[i - 1], channels[i],
                                             kernel_size, stride, padding, use_complex_batchnorm))
        self.decoder = nn.ModuleList()
        for i in range(len(channels) - 1, -1, -1):
            self.decoder.append(DecoderBlock(channels[i]*2, 1 if i == 0 else channels[i - 1],
                                             kernel_size, stride, padding, use_complex_batchnorm,
                                             output_padding, norm=i != 0, activation=i != 0))
        enc_out_dim = input_dim
        for _ in channels:
            enc_out_dim = (enc_out_dim + 2*padding[0] - kernel_size[0])//stride[0] + 1
        self.lstm = LSTMBlock(input_size=channels[-1]*enc_out_dim, hidden_size=lstm_channels,
                              num_layers=lstm_layers)


def _gemm(a, b, d, batch, M, N, K, lda, ldb, ldd, a_bs, b_bs, d_bs, trans_a=0, trans_b=0,
          bias=None):
    hip.check(hip.lib().brv_gemm_f32(
        hip.ptr(a), hip.ptr(b), hip.ptr(d), batch, M, N, K, lda, ldb, ldd, a_bs, b_bs, d_bs,
        trans_a, trans_b, 1, 0, 0, hip.ptr(bias), 0, hip.stream()), 'brv_gemm_f32')


def _conv(x, w, bias, y, geom, transpose, acc, sign, out_pad=(0, 0)):
    """One real (transposed) convolution: y (+)= sign*(op(x, w) + bias); x, y contiguous."""
    lib = hip.lib()
    (kh, kw), (sh, sw), (ph, pw) = geom
    B, Cin, H, W = x.shape
    Cout = y.shape[1]
    args = [hip.ptr(x), hip.ptr(w), hip.ptr(bias), hip.ptr(y), B, Cin, H, W, Cout, kh, kw, sh, sw,
            ph, pw]
    if transpose:
        hip.check(lib.brv_conv_transpose2d_forward(*args, out_pad[0], out_pad[1], Cin*H*W,
                                                   y[0].numel(), acc, sign, hip.stream()),
                  'brv_conv_transpose2d_forward')
    else:
        hip.check(lib.brv_conv2d_forward(*args, Cin*H*W, y[0].numel(), acc, sign, hip.stream()),
                  'brv_conv2d_forward')


class _ComplexConvFunction(torch.autograd.Function):
    """ComplexWrapper(nn.Conv2d | nn.ConvTranspose2d) (dccrn.py:221-231) on (B, 2*Cin, H, W)
    with the real half first: real = M_r(x_r) - M_i(x_i), imag = M_r(x_i) + M_i(x_r)."""

    @staticmethod
    def forward(ctx, x, wr, br, wi, bi, geom4, transpose):
        (kh, kw), (sh, sw), (ph, pw), (oph, opw) = geom4
        geom = geom4[:3]
        B, C2, H, W = x.shape
        xr, xi = (t.contiguous() for t in x.chunk(2, dim=1))
        if transpose:
            Cout = wr.shape[1]
            Ho, Wo = (H - 1)*sh - 2*ph + kh + oph, (W - 1)*sw - 2*pw + kw + opw
        else:
            Cout = wr.shape[0]
            Ho, Wo = (H + 2*ph - kh)//sh + 1, (W + 2*pw - kw)//sw + 1
        yr = torch.empty(B, Cout, Ho, Wo, dtype=torch.float32, device=x.device)
        yi = torch.empty_like(yr)
        op = (oph, opw)
        _conv(xr, wr, br, yr, geom, transpose, 0, 1.0, op)
        _conv(xi, wi, bi, yr, geom, transpose, 1, -1.0, op)
        _conv(xi, wr, br, yi, geom, transpose, 0, 1.0, op)
        _conv(xr, wi, bi, yi, geom, transpose, 1, 1.0, op)
        ctx.save_for_backward(xr, xi, wr, wi)
        ctx.cfg = (geom, transpose, (H, W), (Ho, Wo))
        return torch.cat([yr, yi], dim=1)

    @staticmethod
    def backward(ctx, dy):
        lib = hip.lib()
        xr, xi, wr, wi = ctx.saved_tensors
        geom, transpose, (H, W), (Ho, Wo) = ctx.cfg
        (kh, kw), (sh, sw), (ph, pw) = geom
        dr, di = (t.contiguous() for t in dy.chunk(2, dim=1))
        B, Cin = xr.shape[:2]
        Cout = dr.shape[1]
        # data gradients: the opposite operation with the same weights, no bias
        dxr, dxi = torch.empty_like(xr), torch.empty_like(xi)
        if transpose:
            back = dict(transpose=False)
        else:
            back = dict(transpose=True,
                        out_pad=(H - ((Ho - 1)*sh - 2*ph + kh), W - ((Wo - 1)*sw - 2*pw + kw)))
        _conv(dr, wr, None, dxr, geom, acc=0, sign=1.0, **back)       # dx_r = Mr^T dr + Mi^T di
        _conv(di, wi, None, dxr, geom, acc=1, sign=1.0, **back)
        _conv(di, wr, None, dxi, geom, acc=0, sign=1.0, **back)       # dx_i = Mr^T di - Mi^T dr
        _conv(dr, wi, None, dxi, geom, acc=1, sign=-1.0, **back)
        dwr, dwi = torch.empty_like(wr), torch.empty_like(wi)
        dbr = torch.empty(Cout, dtype=torch.float32, device=dy.device)
        dbi = torch.empty_like(dbr)

        def wgrad(inp, grad, dw, db, acc, sign):
            if transpose:      # weight (Cin, Cout, kh, kw): roles of input and gradient swap
                hip.check(lib.brv_conv2d_wgrad(
                    hip.ptr(grad), hip.ptr(inp), hip.ptr(dw), None, B, Cout, Ho, Wo, Cin, H, W,
                    kh, kw, sh, sw, ph, pw, Cout*Ho*Wo, Cin*H*W, acc, sign, hip.stream()),
                    'brv_conv2d_wgrad')
            else:
                hip.check(lib.brv_conv2d_wgrad(
                    hip.ptr(inp), hip.ptr(grad), hip.ptr(dw), None, B, Cin, H, W, Cout, Ho, Wo,
                    kh, kw, sh, sw, ph, pw, Cin*H*W, Cout*Ho*Wo, acc, sign, hip.stream()),
                    'brv_conv2d_wgrad')
        wgrad(xr, dr, dwr, dbr, 0, 1.0)            # dWr = x_r*dr + x_i*di
        wgrad(xi, di, dwr, dbr, 1, 1.0)
        wgrad(xr, di, dwi, dbi, 0, 1.0)            # dWi = x_r*di - x_i*dr
        wgrad(xi, dr, dwi, dbi, 1, -1.0)
        # bias gradients: channel sums of the output gradient
        sr, si = torch.empty_like(dbr), torch.empty_like(dbr)
        for src, dst in ((dr, sr), (di, si)):
            hip.check(lib.brv_row_sum(hip.ptr(src), hip.ptr(dst), B, Cout, Ho*Wo, hip.stream()),
                      'brv_row_sum')
        dbr = _CombineFunction.apply(sr, si, 1.0)
        dbi = _CombineFunction.apply(si, sr, -1.0)
        return torch.cat([dxr, dxi], dim=1), dwr, dbr, dwi, dbi, None, None


class _CombineFunction(torch.autograd.Function):
    """a + sign*b."""

    @staticmethod
    def forward(ctx, a, b, sign):
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        hip.check(hip.lib().brv_combine(hip.ptr(a), hip.ptr(b), hip.ptr(out), a.numel(),
                                        float(sign), hip.stream()), 'brv_combine')
        ctx.sign = float(sign)
        return out

    @staticmethod
    def backward(ctx, g):
        return g, ctx.sign*g if ctx.sign != 1.0 else g, None


class _BatchNormActFunction(torch.autograd.Function):
    """nn.BatchNorm2d followed by an optional scalar nn.PReLU (dccrn.py:251-254, 284-289)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, slope, norm, training):
        x = x.contiguous()
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        invstd = torch.empty_like(mean)
        hip.check(hip.lib().brv_batchnorm2d_forward(
            hip.ptr(x), hip.ptr(gamma), hip.ptr(beta), hip.ptr(norm.running_mean),
            hip.ptr(norm.running_var), hip.ptr(slope), hip.ptr(y), hip.ptr(mean), hip.ptr(invstd),
            B, C, H*W, float(norm.eps), float(norm.momentum), int(training), hip.stream()),
            'brv_batchnorm2d_forward')
        if training:
            norm.num_batches_tracked += 1
        ctx.save_for_backward(x, gamma, beta, slope if slope is not None else gamma.new_zeros(0),
                              mean, invstd)
        ctx.training = training
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, slope, mean, invstd = ctx.saved_tensors
        if not ctx.training:
            raise NotImplementedError('gradient of eval-mode batch norm is not built on the HIP '
                                      'path')
        has_slope = slope.numel() > 0
        dy = dy.contiguous()
        B, C, H, W = x.shape
        dx = torch.empty_like(x)
        dgamma, dbeta, dsl = (torch.empty(C, dtype=torch.float32, device=x.device)
                              for _ in range(3))
        hip.check(hip.lib().brv_batchnorm2d_backward(
            hip.ptr(x), hip.ptr(dy), hip.ptr(mean), hip.ptr(invstd), hip.ptr(gamma), hip.ptr(beta),
            hip.ptr(slope) if has_slope else None, hip.ptr(dx), hip.ptr(dgamma), hip.ptr(dbeta),
            hip.ptr(dsl), B, C, H*W, hip.stream()), 'brv_batchnorm2d_backward')
        dslope = None
        if has_slope:
            tot = torch.empty(1, dtype=torch.float32, device=x.device)
            hip.check(hip.lib().brv_row_sum(hip.ptr(dsl), hip.ptr(tot), 1, 1, C, hip.stream()),
                      'brv_row_sum')
            dslope = tot
        return dx, dgamma, dbeta, dslope, None, None


class _LSTMFunction(torch.autograd.Function):
    """Single-layer unidirectional nn.LSTM, batch_first, zero initial state: x (B, T, I) ->
    hidden states (B, T, H)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh):
        lib = hip.lib()
        x = x.contiguous()
        B, T, I = x.shape
        H = w_hh.shape[1]
        gates = torch.empty(B, T, 4*H, dtype=torch.float32, device=x.device)
        _gemm(x, w_ih, gates, 1, B*T, 4*H, I, I, I, 4*H, 0, 0, 0, trans_b=1)
        bias = _CombineFunction.apply(b_ih.detach(), b_hh.detach(), 1.0)
        y = torch.empty(B, T, H, dtype=torch.float32, device=x.device)
        act = torch.empty(B, T, 4*H, dtype=torch.float32, device=x.device)
        cs = torch.empty(B, T, H, dtype=torch.float32, device=x.device)
        hip.check(lib.brv_lstm_recurrent_forward(hip.ptr(gates), hip.ptr(w_hh), hip.ptr(bias),
                                                 hip.ptr(y), hip.ptr(act), hip.ptr(cs), B, T, H,
                                                 hip.stream()), 'brv_lstm_recurrent_forward')
        ctx.save_for_backward(x, w_ih, w_hh, y, act, cs)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = hip.lib()
        x, w_ih, w_hh, y, act, cs = ctx.saved_tensors
        B, T, I = x.shape
        H = w_hh.shape[1]
        dy = dy.contiguous()
        dg = torch.empty(B, T, 4*H, dtype=torch.float32, device=x.device)
        hip.check(lib.brv_lstm_recurrent_backward(hip.ptr(act), hip.ptr(cs), hip.ptr(w_hh),
                                                  hip.ptr(dy), hip.ptr(dg), B, T, H, hip.stream()),
                  'brv_lstm_recurrent_backward')
        dx = torch.empty_like(x)                                   # dg (BT, 4H) @ W_ih (4H, I)
        _gemm(dg, w_ih, dx, 1, B*T, I, 4*H, 4*H, I, I, 0, 0, 0)
        dw_ih = torch.empty_like(w_ih)                             # dg^T (4H, BT) @ x (BT, I)
        _gemm(dg, x, dw_ih, 1, 4*H, I, B*T, 4*H, I, I, 0, 0, 0, trans_a=1)
        h_prev = torch.zeros_like(y)                               # hidden state entering step t
        h_prev[:, 1:] = y[:, :-1]
        dw_hh = torch.empty_like(w_hh)
        _gemm(dg, h_prev, dw_hh, 1, 4*H, H, B*T, 4*H, H, H, 0, 0, 0, trans_a=1)
        db = torch.empty(4*H, dtype=torch.float32, device=x.device)
        # column sums of dg (BT, 4H): row_sum over the transposed view (1, 4H, BT) needs a
        # contiguous (4H, BT) copy
        dgt = dg.view(B*T, 4*H).t().contiguous()
        hip.check(lib.brv_row_sum(hip.ptr(dgt), hip.ptr(db), 1, 4*H, B*T, hip.stream()),
                  'brv_row_sum')
        return dx, dw_ih, dw_hh, db, db.clone()


class _ApplyMaskFunction(torch.autograd.Function):
    """DCCRN.apply_mask (dccrn.py:96-109): x, mask (B, 2, Fq, T) -> complex (B, 1, Fq, T)."""

    @staticmethod
    def forward(ctx, x, mask):
        x, mask = x.contiguous(), mask.contiguous()
        B, _, Fq, T = x.shape
        n = Fq*T
        out = torch.empty(B, n, 2, dtype=torch.float32, device=x.device)
        x2, m2 = x.view(B, 2*n), mask.view(B, 2*n)
        for b in range(B):
            hip.check(hip.lib().brv_dccrn_apply_mask(
                hip.ptr(x2[b]), hip.ptr(x2[b, n:]), hip.ptr(m2[b]), hip.ptr(m2[b, n:]),
                hip.ptr(out[b]), n, hip.stream()), 'brv_dccrn_apply_mask')
        ctx.save_for_backward(x, mask)
        return torch.view_as_complex(out.view(B, 1, Fq, T, 2))

    @staticmethod
    def backward(ctx, g):
        x, mask = ctx.saved_tensors
        B, _, Fq, T = x.shape
        n = Fq*T
        gr = torch.view_as_real(g.to(torch.complex64).contiguous()).view(B, n, 2)
        dm = torch.empty_like(mask)
        x2, m2, d2 = x.view(B, 2*n), mask.view(B, 2*n), dm.view(B, 2*n)
        for b in range(B):
            hip.check(hip.lib().brv_dccrn_apply_mask_backward(
                hip.ptr(x2[b]), hip.ptr(x2[b, n:]), hip.ptr(m2[b]), hip.ptr(m2[b, n:]),
                hip.ptr(gr[b]), hip.ptr(d2[b]), hip.ptr(d2[b, n:]), n, hip.stream()),
                'brv_dccrn_apply_mask_backward')
        return None, dm


@ModelRegistry.register('dccrn')
class DCCRN(BreverBaseModel):
    def __init__(
        self,
        stft_frame_length: int = 512,
        stft_hop_length: int = 128,
        stft_window: str = 'hann',
        channels: list[int] = [16, 32, 64, 128, 128, 128],
        kernel_size: tuple[int, int] = (5, 2),
        stride: tuple[int, int] = (2, 1),
        padding: tuple[int, int] = (2, 0),
        output_padding: tuple[int, int] = (1, 0),
        lstm_channels: int = 128,
        lstm_layers: int = 2,
        use_complex_batchnorm: bool = False,
        criterion: str = 'snr',
        optimizer: str = 'Adam',
        learning_rate: float = 0.0001,
    ):
        super().__init__(criterion=criterion)
        self.kernel_size = kernel_size
        self.stride = stride
        self.channels = channels
        self.stft = STFT(frame_length=stft_frame_length, hop_length=stft_hop_length,
                         window=stft_window)
        self.mask_net = DCCRNMaskNet(
            input_dim=self.stft.frame_length//2, channels=channels, kernel_size=kernel_size,
            stride=stride, padding=padding, output_padding=output_padding,
            lstm_channels=lstm_channels, lstm_layers=lstm_layers,
            use_complex_batchnorm=use_complex_batchnorm)
        self.optimizer = self.init_optimizer(optimizer, lr=learning_rate)

    # ---- layers ------------------------------------------------------------------------
    def _complex_conv(self, x, wrap, transpose):
        mr, mi = wrap.module_real, wrap.module_imag
        return _ComplexConvFunction.apply(x, mr.weight, mr.bias, mi.weight, mi.bias,
                                          self.mask_net.geom, transpose)

    def _norm_act(self, x, norm, act):
        if norm is None:
            return x
        training = norm.training and norm.track_running_stats
        return _BatchNormActFunction.apply(x, norm.weight, norm.bias,
                                           act.weight if act is not None else None, norm, training)

    @staticmethod
    def _lstm(lstm, x):
        return _LSTMFunction.apply(x, lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0,
                                   lstm.bias_hh_l0)

    def _lstm_block(self, x):
        """LSTMBlock (dccrn.py:293-311) on the encoder output (B, 2C, Fq, T) -> same shape."""
        from .ffnn import _LinearFunction
        blk = self.mask_net.lstm
        B, C2, Fq, T = x.shape
        rows = x.reshape(B, C2*Fq, T).transpose(1, 2)        # (B, T, features): real then imag
        real, imag = rows.chunk(2, dim=-1)
        for layer in blk.lstm.layers:
            rr = self._lstm(layer.module_real, real)
            ii = self._lstm(layer.module_imag, imag)
            ri = self._lstm(layer.module_real, imag)
            ir = self._lstm(layer.module_imag, real)
            real, imag = _CombineFunction.apply(rr, ii, -1.0), _CombineFunction.apply(ri, ir, 1.0)
        # Linear applied on the feature axis of (B, features, T): the output is already in the
        # (channels*freqs, frames) layout of the decoder input
        out_r = _LinearFunction.apply(real.transpose(1, 2).contiguous(), blk.linear_r.weight,
                                      blk.linear_r.bias)
        out_i = _LinearFunction.apply(imag.transpose(1, 2).contiguous(), blk.linear_i.weight,
                                      blk.linear_i.bias)
        return torch.cat([out_r, out_i], dim=1).view(B, C2, Fq, T)

    def _mask_net(self, x):
        net = self.mask_net
        encoder_outputs = []
        for blk in net.encoder:
            x = self._norm_act(self._complex_conv(x, blk.conv, False), blk.norm, blk.activation)
            encoder_outputs.append(x)
        x = self._lstm_block(x)
        for blk, enc in zip(net.decoder, reversed(encoder_outputs)):
            real, imag = x.chunk(2, dim=1)
            skip_real, skip_imag = enc.chunk(2, dim=1)
            x = torch.cat([real, skip_real, imag, skip_imag], dim=1)
            x = self._complex_conv(x, blk.conv, True)
            x = self._norm_act(x, blk.norm, blk.activation)
        return x

    def forward(self, x):
        hip.require_device(x)
        length = x.shape[-1]
        with torch.no_grad():                         # the input carries no gradient
            spec = self.stft(x.float())               # (B, 257, F) complex
            spec = spec[..., 1:, :]                   # remove the DC component
            xin = torch.stack([spec.real, spec.imag], dim=1).contiguous()
        mask = self._mask_net(xin)
        out = _ApplyMaskFunction.apply(xin, mask).squeeze(1)
        out = torch.nn.functional.pad(out, (0, 0, 1, 0))      # put the DC row back (zeros)
        y = self.stft.backward(out)
        return y[..., :length]

    def transform(self, sources):
        assert sources.shape[0] == 2  # mixture, foreground
        return sources.mean(axis=-2)

    def loss(self, batch, lengths, use_amp):
        inputs, labels = batch[:, 0], batch[:, 1]
        outputs = self(inputs)                        # fp32 kernels; use_amp has no effect yet
        return self.criterion(outputs, labels, lengths).mean()

    def update(self, loss, scaler):
        super().update(loss, scaler, grad_clip=5.0)

    def _enhance(self, x, use_amp):
        with torch.no_grad():
            return self.forward(x.mean(axis=-2))
