"""DCCRN (deep complex convolution recurrent network) on the HIP path -- forward values.

Same constructor signature, registry key (``dccrn``), module tree / state-dict names and
seeded initialisation as the reference (brever/models/dccrn/dccrn.py:28-358): the
``nn.Conv2d`` / ``nn.ConvTranspose2d`` / ``nn.BatchNorm2d`` / ``nn.PReLU`` / ``nn.LSTM`` /
``nn.Linear`` objects are created in the reference's order but only hold parameters; the
arithmetic (STFT, the four real convolutions of every complex layer, batch norm + PReLU, the
complex LSTM, the mask application, iSTFT) runs in ``libbrever_hip.so``.

Forward and backward are ``torch.autograd.Function`` pieces whose two sides call the HIP
kernels (``brv_im2col / brv_col2im / brv_complex_weight_pack`` around ``brv_gemm_f32``,
``brv_batchnorm2d_*``, ``brv_lstm_recurrent_*``, ``brv_gemm_f32``, ``brv_dccrn_apply_mask*``,
``brv_stft_forward`` / ``brv_istft_backward`` and its adjoint); torch only concatenates, slices
and transposes between them. fp32 activations throughout; ``use_amp`` runs the matrix products (convolutions, LSTM input
projections) with bf16 operands and fp32 accumulation, otherwise on the exact-fp32 MFMA.
"""
import os

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


class ComplexBatchNorm2d(_ParamOnly):
    """Parameter / buffer container of the reference's ComplexBatchNorm2d
    (complex_batchnorm.py:29-74): weight (3, C) = W_rr, W_ri, W_ii, bias (2, C), running mean
    (2, C) and running 2x2 covariance (2, 2, C)."""

    def __init__(self, num_features, eps=1e-05, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.track_running_stats = True
        self.weight = nn.Parameter(torch.empty(3, num_features))
        self.bias = nn.Parameter(torch.empty(2, num_features))
        self.register_buffer('running_mean', torch.zeros(2, num_features))
        self.register_buffer('running_var', torch.eye(2, 2).unsqueeze(-1).repeat(1, 1, num_features))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))
        with torch.no_grad():
            self.weight.copy_(torch.tensor([[1.0], [0.0], [1.0]]))
            self.bias.zero_()


class EncoderBlock(_ParamOnly):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding,
                 use_complex_batchnorm):
        super().__init__()
        self.conv = ComplexWrapper(nn.Conv2d, in_channels=in_channels, out_channels=out_channels,
                                   kernel_size=kernel_size, stride=stride, padding=padding)
        self.norm = ComplexBatchNorm2d(out_channels) if use_complex_batchnorm \
            else nn.BatchNorm2d(2*out_channels)
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
            self.norm = ComplexBatchNorm2d(out_channels) if use_complex_batchnorm \
                else nn.BatchNorm2d(2*out_channels)
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


_AMP = {'on': False}      # set by DCCRN.loss / _enhance around the forward pass (use_amp)


def _gemm(a, b, d, batch, M, N, K, lda, ldb, ldd, a_bs, b_bs, d_bs, trans_a=0, trans_b=0,
          kbatch=1, a_kbs=0, b_kbs=0, bias=None, lowp=False):
    """fp32 matrices in HBM; ``lowp``: operands rounded to bf16 inside the kernel, fp32
    accumulation (``use_amp``), else the exact-fp32 MFMA. A bf16 ``b`` or ``d`` tensor (the
    column matrices of the use_amp convolutions) selects ``brv_gemm_bf16_mixed``."""
    flags = int(b.dtype == torch.bfloat16) | int(d.dtype == torch.bfloat16) << 1
    if flags:
        assert lowp
        hip.check(hip.lib().brv_gemm_bf16_mixed(
            hip.ptr(a), hip.ptr(b), hip.ptr(d), batch, M, N, K, lda, ldb, ldd, a_bs, b_bs, d_bs,
            trans_a, trans_b, kbatch, a_kbs, b_kbs, hip.ptr(bias), 0, flags, hip.stream()),
            'brv_gemm_bf16_mixed')
        return
    if not lowp:
        hip.gemm_f32(a, b, d, batch, M, N, K, lda, ldb, ldd, a_bs, b_bs, d_bs, trans_a, trans_b, kbatch, a_kbs,
                     b_kbs, bias, 0)
        return
    hip.check(hip.lib().brv_gemm_bf16(
        hip.ptr(a), hip.ptr(b), hip.ptr(d), batch, M, N, K, lda, ldb, ldd, a_bs, b_bs, d_bs,
        trans_a, trans_b, kbatch, a_kbs, b_kbs, hip.ptr(bias), 0, hip.stream()), 'brv_gemm_bf16')


def _gemm_conv(a, img, d, batch, M, N, K, lda, ldd, a_bs, img_bs, d_bs, mode, image, geom, grid,
               trans_b=0, kbatch=1, a_kbs=0, img_kbs=0, bias=None):
    """``_gemm(lowp=True)`` whose B operand is the column matrix of ``img`` -- never written out
    (``brv_gemm_bf16_conv``): ``image`` = (C, H, W) of ``img``, ``grid`` = the pixel grid of the
    columns; mode 1 = im2col, mode 2 = the gather form of col2im."""
    (kh, kw), (sh, sw), (ph, pw) = geom
    hip.check(hip.lib().brv_gemm_bf16_conv(
        hip.ptr(a), hip.ptr(img), hip.ptr(d), batch, M, N, K, lda, ldd, a_bs, img_bs, d_bs, 0, trans_b,
        kbatch, a_kbs, img_kbs, hip.ptr(bias), 0, mode, image[0], image[1], image[2], kh, kw, sh, sw,
        ph, pw, grid[0], grid[1], hip.stream()), 'brv_gemm_bf16_conv')


_IMPLICIT = os.environ.get('BRV_DCCRN_IM2COL', '0') != '1'     # use_amp: implicit GEMM convolutions
_COL_KEEP_BYTES = int(float(os.environ.get('BRV_DCCRN_COL_KEEP_GB', '8'))*2**30)   # per column matrix kept for backward


_ROWS = os.environ.get('BRV_DCCRN_ROWS', '1') != '0'      # use_amp: one-launch row convolutions (csrc/cconv.hip)
_ROWS_GEOM = ((5, 2), (2, 1), (2, 0), (1, 0))


def _cconv_rows(x, wc, bias, M, m_stride, c_stride, transposed, x2=None, split_out=False, wp=None):
    """``brv_cconv_rows``: the (5, 2) / (2, 1) / (2, 0) convolution (``transposed`` = 0) or transposed
    convolution (1) of ``x`` (B, C, H, W) with W[m][c][i][j] = wc.flat[m*m_stride + c*c_stride + 2i + j].
    ``x2``: the input is the skip concatenation [x[:, :s], x2[:, :s], x[:, s:], x2[:, s:]] read from its two
    sources; ``split_out``: the M output channels are dealt the same way to two (B, M/2, ..) tensors."""
    lib = hip.lib()
    B, C, H, W = x.shape
    seg = 0
    if x2 is not None:
        assert x2.shape == x.shape and C % 16 == 0
        seg, C = C//2, 2*C
    if wp is None:       # (``wp``: the fragments of this reading of ``wc``, packed already -- _pack_complex_layer)
        wp = torch.empty(lib.brv_cconv_packed_bytes(M, C), dtype=torch.uint8, device=x.device)
        hip.check(lib.brv_cconv_pack(hip.ptr(wc), hip.ptr(wp), M, C, m_stride, c_stride, hip.stream()),
                  'brv_cconv_pack')
    shape = (B, M//2 if split_out else M) + ((2*H, W + 1) if transposed else (H//2, W - 1))
    out = torch.empty(shape, dtype=torch.float32, device=x.device)
    out2 = torch.empty_like(out) if split_out else None
    hip.check(lib.brv_cconv_rows(hip.ptr(x), hip.ptr(x2), seg, hip.ptr(wp), hip.ptr(bias), hip.ptr(out),
                                 hip.ptr(out2), M//4 if split_out else 0, B, C, M, H, W, int(transposed),
                                 hip.stream()), 'brv_cconv_rows')
    return (out, out2) if split_out else out


def _pack_complex_layer(wr, wi, br, bi, sign, fwd, bwd):
    """One launch (``brv_cconv_pack_complex``) for the packed real matrix, the packed bias and the operand fragments of
    the forward reading ``fwd`` = (M, C, m_stride, c_stride) of it and, if asked for, the data gradient's ``bwd``."""
    lib = hip.lib()
    R, Cw = wr.shape[0], wr[0].numel()
    dev = wr.device
    wc = torch.empty(2*R, 2*Cw, dtype=torch.float32, device=dev)
    bias = torch.empty(2*br.numel(), dtype=torch.float32, device=dev)
    wp1 = torch.empty(lib.brv_cconv_packed_bytes(fwd[0], fwd[1]), dtype=torch.uint8, device=dev)
    wp2 = torch.empty(lib.brv_cconv_packed_bytes(bwd[0], bwd[1]), dtype=torch.uint8, device=dev) if bwd else None
    wr_c, wi_c, br_c, bi_c = wr.contiguous(), wi.contiguous(), br.contiguous(), bi.contiguous()
    hip.check(lib.brv_cconv_pack_complex(
        hip.ptr(wr_c), hip.ptr(wi_c), hip.ptr(br_c), hip.ptr(bi_c), R, Cw, br.numel(), float(sign), hip.ptr(wc),
        hip.ptr(bias), hip.ptr(wp1), *fwd, hip.ptr(wp2), *(bwd or (0, 0, 0, 0)), hip.stream()),
        'brv_cconv_pack_complex')
    return wc, bias, wp1, wp2


def _cconv_wgrad(small, big, small2=None):
    """``brv_cconv_wgrad``: (A, 10 C) weight-gradient matrix of the row convolutions; ``small2``: the second
    source of a skip concatenation that was never materialised."""
    B, C, Hb, Wb = big.shape
    seg = small.shape[1]//2 if small2 is not None else 0
    A = 4*seg if seg else small.shape[1]
    Hs, Ws = small.shape[2:]
    assert Hb == 2*Hs and Wb == Ws + 1
    out = torch.empty(A, 10*C, dtype=torch.float32, device=big.device)
    ws = torch.empty(hip.lib().brv_cconv_wgrad_workspace_bytes(B, A, C, Hs), dtype=torch.uint8, device=big.device)
    lowp = big.dtype == torch.bfloat16          # (all three images alike: _BlockFunction)
    assert small.dtype == big.dtype and (small2 is None or small2.dtype == big.dtype)
    # (the LDS-DMA kernel fetches whole 16-byte pieces around the images' ends: include/brever_hip.h)
    assert not lowp or all(_has_slack(t) for t in (small, small2, big) if t is not None), 'bf16 image without slack'
    fn, name = (hip.lib().brv_cconv_wgrad_bf16, 'brv_cconv_wgrad_bf16') if lowp else \
        (hip.lib().brv_cconv_wgrad, 'brv_cconv_wgrad')
    hip.check(fn(hip.ptr(small), hip.ptr(small2), hip.ptr(big), hip.ptr(out), hip.ptr(ws),
                 B, A, C, Hs, Ws, seg, hip.stream()), name)
    return out


# Weight and bias gradients of the row convolutions on a side stream (use_amp training): nothing on the way back to
# the input depends on them, and the backward pass has a long stretch -- the recurrences' BPTT, 64 chains = a quarter
# of the CUs for 0.9 ms -- that the decoder's weight gradients fill. BRV_DCCRN_WGRAD_SIDE=0: everything in order.
_WGRAD_SIDE = os.environ.get('BRV_DCCRN_WGRAD_SIDE', '1') != '0'
# BRV_LSTM_MV=0: the fp32 recurrence kernels under use_amp too (round 4)
_LSTM_MV = os.environ.get('BRV_LSTM_MV', '1') != '0'
_side = {'streams': {}, 'pending': {}}     # both keyed by device index


def _side_stream(device):
    st = _side['streams'].get(device.index)
    if st is None:
        st = _side['streams'][device.index] = torch.cuda.Stream(device)
    return st


def _join_side(device):
    """The stream the gradients are consumed on waits for the side stream. Queued once per backward pass (autograd
    engine callback, runs when the pass ends) AND called by ``DCCRN.gather_grads`` / ``update`` before anything
    reads a gradient -- the callback alone is not enough: a backward pass that raises never runs it."""
    _side['pending'][device.index] = False
    st = _side['streams'].get(device.index)
    if st is not None:
        torch.cuda.current_stream(device).wait_stream(st)


def _side_allowed(params):
    """Parameter gradients may only come off the side stream when nothing looks at them before the end of the
    backward pass: no ``.grad`` to accumulate into in place (``zero_grad(set_to_none=False)``, gradient
    accumulation) and no post-accumulate hook (the per-parameter all-reduce of ``GradSynchronizer`` for models
    without a flat buffer runs on the main stream DURING backward)."""
    if not _WGRAD_SIDE:
        return False
    return all(p.grad is None and not getattr(p, '_post_accumulate_grad_hooks', None) for p in params)


def _im2col(x, geom, grid, lowp=False):
    """``lowp``: the column matrix in bf16 (half the bytes of the largest tensor of the layer)."""
    (kh, kw), (sh, sw), (ph, pw) = geom
    B, C, H, W = x.shape
    col = torch.empty(B, C*kh*kw, grid[0]*grid[1], dtype=torch.bfloat16 if lowp else torch.float32,
                      device=x.device)
    fn, name = (hip.lib().brv_im2col_bf16, 'brv_im2col_bf16') if lowp else \
        (hip.lib().brv_im2col, 'brv_im2col')
    hip.check(fn(hip.ptr(x), hip.ptr(col), B, C, H, W, kh, kw, sh, sw, ph, pw, grid[0], grid[1],
                 hip.stream()), name)
    return col


def _col2im(col, bias, C, image, geom, grid):
    (kh, kw), (sh, sw), (ph, pw) = geom
    B = col.shape[0]
    y = torch.empty(B, C, image[0], image[1], dtype=torch.float32, device=col.device)
    fn, name = (hip.lib().brv_col2im_bf16, 'brv_col2im_bf16') if col.dtype == torch.bfloat16 else \
        (hip.lib().brv_col2im, 'brv_col2im')
    hip.check(fn(hip.ptr(col), hip.ptr(bias), hip.ptr(y), B, C, image[0], image[1], kh, kw, sh, sw,
                 ph, pw, grid[0], grid[1], hip.stream()), name)
    return y


def _combine(a, b, sign):
    out = torch.empty_like(a)
    hip.check(hip.lib().brv_combine(hip.ptr(a), hip.ptr(b), hip.ptr(out), a.numel(), float(sign),
                                    hip.stream()), 'brv_combine')
    return out


class _ComplexConvFunction(torch.autograd.Function):
    """ComplexWrapper(nn.Conv2d | nn.ConvTranspose2d) (dccrn.py:221-231) on (B, 2*Cin, H, W)
    with the real half first: real = M_r(x_r) - M_i(x_i), imag = M_r(x_i) + M_i(x_r).

    The four real convolutions are one matrix product per batch item on the exact-fp32 MFMA:
    the complex weight becomes the real matrix [[Wr, -Wi], [Wi, Wr]] (``brv_complex_weight_pack``)
    applied to the column matrix of the whole (real | imaginary) input (``brv_im2col``); the
    transposed convolution applies the transposed matrix and scatters back (``brv_col2im``).
    The backward pass is the same three pieces with the roles swapped; the column matrix is
    rebuilt there instead of being kept."""

    @staticmethod
    def forward(ctx, x, wr, br, wi, bi, geom4, transpose, skip=None):
        """``skip``: the convolution input is torch.cat([x_real, skip_real, x_imag, skip_imag], dim=1)
        (dccrn.py:213-217); the row kernels read the two tensors in place."""
        (kh, kw), (sh, sw), (ph, pw), (oph, opw) = geom4
        geom = geom4[:3]
        lib = hip.lib()
        lowp = ctx.lowp = _AMP['on']
        ctx.side_ok = _side_allowed((wr, br, wi, bi))
        kept_col = None
        x = x.contiguous()
        B, C2, H, W = x.shape
        rows = ctx.rows = bool(lowp and _ROWS and tuple(map(tuple, geom4)) == _ROWS_GEOM
                               and (transpose or (H % 2 == 0 and W >= 2)))
        ctx.seg = 0
        if skip is not None:
            ctx.seg = C2//2
            if rows and transpose and ctx.seg % 8 == 0:
                skip = skip.contiguous()
            else:                                  # the other paths work on the concatenated tensor
                s = ctx.seg
                x, skip = torch.cat([x[:, :s], skip[:, :s], x[:, s:], skip[:, s:]], dim=1), None
            C2 *= 2
        Cin = C2//2
        R = wr.shape[0]
        Cw = wr[0].numel()
        khw = kh*kw
        ctx.wp_bwd = None
        if rows:
            # packed matrix, bias, and the fragments of both readings of the matrix (this forward's, the data
            # gradient's) in one launch; the data gradient's only if something upstream wants it
            Cout = wr.shape[1] if transpose else R
            need_dx = ctx.needs_input_grad[0] or (skip is not None and ctx.needs_input_grad[7]) or bool(ctx.seg)
            fwd = (2*Cout, C2, khw, 2*Cw) if transpose else (2*Cout, C2, 2*Cw, khw)
            bwd = ((2*Cin, 2*Cout, 2*Cw, khw) if transpose else (2*Cin, 2*Cout, khw, 2*Cw)) if need_dx else None
            wc, bias, wp_fwd, ctx.wp_bwd = _pack_complex_layer(wr, wi, br, bi, -1.0 if transpose else 1.0, fwd, bwd)
        else:
            wc = torch.empty(2*R, 2*Cw, dtype=torch.float32, device=x.device)
            wr_c, wi_c = wr.contiguous(), wi.contiguous()      # alive until the launch is queued
            hip.check(lib.brv_complex_weight_pack(hip.ptr(wr_c), hip.ptr(wi_c),
                                                  hip.ptr(wc), R, Cw, -1.0 if transpose else 1.0,
                                                  hip.stream()), 'brv_complex_weight_pack')
            bias = torch.empty(2*br.numel(), dtype=torch.float32, device=x.device)
            br_c, bi_c = br.contiguous(), bi.contiguous()
            hip.check(lib.brv_complex_bias_pack(hip.ptr(br_c), hip.ptr(bi_c), hip.ptr(bias), br.numel(),
                                                hip.stream()), 'brv_complex_bias_pack')
        if rows and not transpose:
            Ho, Wo = H//2, W - 1
            y = _cconv_rows(x, wc, bias, 2*Cout, 2*Cw, khw, 0, wp=wp_fwd)
        elif rows:
            Ho, Wo = 2*H, W + 1
            y = _cconv_rows(x, wc, bias, 2*Cout, khw, 2*Cw, 1, x2=skip, wp=wp_fwd)
        elif not transpose and lowp and _IMPLICIT:
            # the column matrix of x is read in place (brv_gemm_bf16_conv): no im2col pass, no 10x copy
            Cout = R
            Ho, Wo = (H + 2*ph - kh)//sh + 1, (W + 2*pw - kw)//sw + 1
            y = torch.empty(B, 2*Cout, Ho, Wo, dtype=torch.float32, device=x.device)
            _gemm_conv(wc, x, y, B, 2*Cout, Ho*Wo, 2*Cw, 2*Cw, Ho*Wo, 0, 2*Cin*H*W, 2*Cout*Ho*Wo, 1,
                       (2*Cin, H, W), geom, (Ho, Wo), bias=bias)
        elif transpose:
            Cout = wr.shape[1]
            Ho, Wo = (H - 1)*sh - 2*ph + kh + oph, (W - 1)*sw - 2*pw + kw + opw
            col = torch.empty(B, 2*Cw, H*W, dtype=torch.bfloat16 if lowp else torch.float32,
                              device=x.device)
            _gemm(wc, x, col, B, 2*Cw, H*W, 2*Cin, 2*Cw, H*W, H*W, 0, 2*Cin*H*W, 2*Cw*H*W,
                  trans_a=1, lowp=lowp)
            y = _col2im(col, bias, 2*Cout, (Ho, Wo), geom, (H, W))
        else:
            Cout = R
            Ho, Wo = (H + 2*ph - kh)//sh + 1, (W + 2*pw - kw)//sw + 1
            col = _im2col(x, geom, (Ho, Wo), lowp)
            y = torch.empty(B, 2*Cout, Ho, Wo, dtype=torch.float32, device=x.device)
            _gemm(wc, col, y, B, 2*Cout, Ho*Wo, 2*Cw, 2*Cw, Ho*Wo, Ho*Wo, 0, 2*Cw*Ho*Wo,
                  2*Cout*Ho*Wo, bias=bias, lowp=lowp)
            # kept for the weight gradient (no second im2col in backward; 3 GB over the six encoder layers at
            # 16 x 4 s -- the device has 288), unless one matrix alone is out of proportion
            if any(ctx.needs_input_grad[:5]) and col.numel()*col.element_size() <= _COL_KEEP_BYTES:
                kept_col = col
        ctx.two = skip is not None
        ctx.col = kept_col
        ctx.save_for_backward(x, wc, *((skip,) if ctx.two else ()))
        ctx.cfg = (geom, transpose, (H, W), (Ho, Wo), Cin, Cout, R, Cw, wr.shape)
        return y

    @staticmethod
    def _unpack_param_grads(dwc, dy, wshape, R, Cw, Cout, B, HoWo, transpose):
        """dwc (2R, 2Cw) -> (d Wr, d Wi); channel sums of dy -> (d br, d bi) (bias = [br - bi | br + bi])."""
        lib = hip.lib()
        dwr = torch.empty(wshape, dtype=torch.float32, device=dy.device)
        dwi = torch.empty_like(dwr)
        hip.check(lib.brv_complex_weight_unpack(hip.ptr(dwc), hip.ptr(dwr), hip.ptr(dwi), R, Cw,
                                                -1.0 if transpose else 1.0, hip.stream()),
                  'brv_complex_weight_unpack')
        sums = torch.empty(2*Cout, dtype=torch.float32, device=dy.device)
        hip.check(lib.brv_row_sum(hip.ptr(dy), hip.ptr(sums), B, 2*Cout, HoWo, hip.stream()),
                  'brv_row_sum')
        dbr = torch.empty(Cout, dtype=torch.float32, device=dy.device)
        dbi = torch.empty_like(dbr)
        hip.check(lib.brv_complex_bias_unpack(hip.ptr(sums), hip.ptr(dbr), hip.ptr(dbi), Cout, hip.stream()),
                  'brv_complex_bias_unpack')
        return dwr, dwi, dbr, dbi

    @staticmethod
    def backward(ctx, dy):
        lib = hip.lib()
        x, wc = ctx.saved_tensors[:2]
        skip = ctx.saved_tensors[2] if ctx.two else None
        geom, transpose, (H, W), (Ho, Wo), Cin, Cout, R, Cw, wshape = ctx.cfg
        dy = dy.contiguous()
        B = x.shape[0]
        lowp = ctx.lowp
        khw = geom[0][0]*geom[0][1]
        dwc = None if ctx.rows else torch.empty_like(wc)
        dskip = None
        if ctx.rows:
            def param_grads():
                """dW (complex pair) and the bias gradients: only dy, x (and skip) go in."""
                if transpose:
                    dwc_ = _cconv_wgrad(x, dy, small2=skip)
                elif 2*Cin >= 8:
                    dwc_ = _cconv_wgrad(dy, x)
                else:        # (2 real input channels of the first encoder: the column-matrix kernel wastes less there)
                    dwc_ = torch.empty_like(wc)
                    _gemm_conv(dy, x, dwc_, 1, 2*Cout, 2*Cw, Ho*Wo, Ho*Wo, 2*Cw, 0, 0, 0, 1, (2*Cin, H, W), geom,
                               (Ho, Wo), trans_b=1, kbatch=B, a_kbs=2*Cout*Ho*Wo, img_kbs=2*Cin*H*W)
                return _ComplexConvFunction._unpack_param_grads(dwc_, dy, wshape, R, Cw, Cout, B, Ho*Wo, transpose)
            side = _side_stream(dy.device) if ctx.side_ok else None
            if side is not None:
                side.wait_stream(torch.cuda.current_stream(dy.device))      # dy, x exist; dx is not waited for
            # (the input of the first encoder layer -- the spectrogram, built under no_grad -- needs no
            # gradient: its transposed row convolution at the largest resolution, H = 256, was computed and dropped)
            need_dx = ctx.needs_input_grad[0] or (ctx.two and ctx.needs_input_grad[7]) or bool(ctx.seg)
            dx = None
            if not need_dx:
                pass
            elif transpose:
                dx = _cconv_rows(dy, wc, None, 2*Cin, 2*Cw, khw, 0, split_out=ctx.two, wp=ctx.wp_bwd)
                if ctx.two:
                    dx, dskip = dx
            else:
                dx = _cconv_rows(dy, wc, None, 2*Cin, khw, 2*Cw, 1, wp=ctx.wp_bwd)
            if side is None:
                dwr, dwi, dbr, dbi = param_grads()
            else:
                with torch.cuda.stream(side):
                    dwr, dwi, dbr, dbi = param_grads()
                for t in (x, dy, skip):
                    if t is not None:
                        t.record_stream(side)
                if not _side['pending'].get(dy.device.index):
                    _side['pending'][dy.device.index] = True
                    dev = dy.device
                    torch.autograd.Variable._execution_engine.queue_callback(lambda: _join_side(dev))
            if ctx.seg and not ctx.two and dx is not None:      # the concatenation was materialised (segments off the chunk of 8)
                s = ctx.seg
                dx, dskip = (torch.cat([dx[:, :s], dx[:, 2*s:3*s]], dim=1),
                             torch.cat([dx[:, s:2*s], dx[:, 3*s:]], dim=1))
            return dx, dwr, dbr, dwi, dbi, None, None, dskip
        elif transpose and lowp and _IMPLICIT:
            dx = torch.empty_like(x)
            _gemm_conv(wc, dy, dx, B, 2*Cin, H*W, 2*Cw, 2*Cw, H*W, 0, 2*Cout*Ho*Wo, 2*Cin*H*W, 1,
                       (2*Cout, Ho, Wo), geom, (H, W))
            _gemm_conv(x, dy, dwc, 1, 2*Cin, 2*Cw, H*W, H*W, 2*Cw, 0, 0, 0, 1, (2*Cout, Ho, Wo), geom,
                       (H, W), trans_b=1, kbatch=B, a_kbs=2*Cin*H*W, img_kbs=2*Cout*Ho*Wo)
        elif not transpose and lowp and _IMPLICIT:
            _gemm_conv(dy, x, dwc, 1, 2*Cout, 2*Cw, Ho*Wo, Ho*Wo, 2*Cw, 0, 0, 0, 1, (2*Cin, H, W), geom,
                       (Ho, Wo), trans_b=1, kbatch=B, a_kbs=2*Cout*Ho*Wo, img_kbs=2*Cin*H*W)
            # data gradient: product + scatter (the generic gather form doubles the matrix work at
            # stride 2 and shrinks M to 2*Cin: measured 3.8x slower than this pair)
            col = torch.empty(B, 2*Cw, Ho*Wo, dtype=torch.bfloat16, device=x.device)
            _gemm(wc, dy, col, B, 2*Cw, Ho*Wo, 2*Cout, 2*Cw, Ho*Wo, Ho*Wo, 0, 2*Cout*Ho*Wo,
                  2*Cw*Ho*Wo, trans_a=1, lowp=lowp)
            dx = _col2im(col, None, 2*Cin, (H, W), geom, (Ho, Wo))
        elif transpose:
            dcol = _im2col(dy, geom, (H, W), lowp)                 # (B, 2*Cw, H*W)
            dx = torch.empty_like(x)
            _gemm(wc, dcol, dx, B, 2*Cin, H*W, 2*Cw, 2*Cw, H*W, H*W, 0, 2*Cw*H*W, 2*Cin*H*W,
                  lowp=lowp)
            _gemm(x, dcol, dwc, 1, 2*Cin, 2*Cw, H*W, H*W, H*W, 2*Cw, 0, 0, 0, trans_b=1,
                  kbatch=B, a_kbs=2*Cin*H*W, b_kbs=2*Cw*H*W, lowp=lowp)
        else:
            col, ctx.col = ctx.col, None                           # the forward pass's column matrix, if kept
            if col is None:
                col = _im2col(x, geom, (Ho, Wo), lowp)             # (B, 2*Cw, Ho*Wo)
            _gemm(dy, col, dwc, 1, 2*Cout, 2*Cw, Ho*Wo, Ho*Wo, Ho*Wo, 2*Cw, 0, 0, 0, trans_b=1,
                  kbatch=B, a_kbs=2*Cout*Ho*Wo, b_kbs=2*Cw*Ho*Wo, lowp=lowp)
            _gemm(wc, dy, col, B, 2*Cw, Ho*Wo, 2*Cout, 2*Cw, Ho*Wo, Ho*Wo, 0, 2*Cout*Ho*Wo,
                  2*Cw*Ho*Wo, trans_a=1, lowp=lowp)                           # the buffer now holds dcol
            dx = _col2im(col, None, 2*Cin, (H, W), geom, (Ho, Wo))
        dwr, dwi, dbr, dbi = _ComplexConvFunction._unpack_param_grads(dwc, dy, wshape, R, Cw, Cout, B, Ho*Wo,
                                                                      transpose)
        if ctx.seg and not ctx.two:          # the concatenation was materialised: deal its gradient back
            s = ctx.seg
            dx, dskip = (torch.cat([dx[:, :s], dx[:, 2*s:3*s]], dim=1),
                         torch.cat([dx[:, s:2*s], dx[:, 3*s:]], dim=1))
        return dx, dwr, dbr, dwi, dbi, None, None, dskip


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


class _ComplexMixFunction(torch.autograd.Function):
    """(2 modules, 2 B, T, H) outputs of the two single-layer LSTMs of a ComplexLSTM layer on [real; imag] ->
    (real, imag) = (real(real) - imag(imag), real(imag) + imag(real)) (dccrn.py:330-358): one launch each way."""

    @staticmethod
    def forward(ctx, out):
        out = out.contiguous()
        G, B2, T, H = out.shape
        assert G == 2 and B2 % 2 == 0
        real = torch.empty(B2//2, T, H, dtype=torch.float32, device=out.device)
        imag = torch.empty_like(real)
        hip.check(hip.lib().brv_complex_mix_forward(hip.ptr(out), hip.ptr(real), hip.ptr(imag), real.numel(),
                                                    hip.stream()), 'brv_complex_mix_forward')
        ctx.shape = out.shape
        return real, imag

    @staticmethod
    def backward(ctx, greal, gimag):
        greal, gimag = greal.contiguous(), gimag.contiguous()
        dout = torch.empty(ctx.shape, dtype=torch.float32, device=greal.device)
        hip.check(hip.lib().brv_complex_mix_backward(hip.ptr(greal), hip.ptr(gimag), hip.ptr(dout), greal.numel(),
                                                     hip.stream()), 'brv_complex_mix_backward')
        return dout


class _ForkFunction(torch.autograd.Function):
    """One tensor, two consumers (an encoder output feeds the next block AND the decoder's skip input,
    dccrn.py:205-217): the two gradients are summed by ``brv_combine`` instead of the autograd engine's
    ``at::add``."""

    @staticmethod
    def forward(ctx, x):
        return x.view(x.shape), x.view(x.shape)

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None or g2 is None:
            return g1 if g2 is None else g2
        return _combine(g1.contiguous(), g2.contiguous(), 1.0)


# nn.BatchNorm2d.num_batches_tracked of the layers a training forward has passed: one multi-tensor add at the end of the
# mask network instead of a one-element launch per layer (eleven in the default network)
_BN_COUNTERS = []


def _flush_bn_counters():
    if _BN_COUNTERS:
        with torch.no_grad():
            torch._foreach_add_(list(_BN_COUNTERS), 1)
        _BN_COUNTERS.clear()


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
            _BN_COUNTERS.append(norm.num_batches_tracked)      # (+= 1 for all layers in one launch: _flush_bn_counters)
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


# use_amp with the row kernels: the activations BETWEEN the layers live in HBM as bf16 -- the tensors the reference's
# autocast holds there too. Exactly the values the fp32 form of the row kernels rounds its operands to on the way in,
# so the products are bit-identical; what changes is the bytes: the convolutions and their weight gradients are bound
# by what a CU takes in per cycle (DESIGN.md 5b "Round 6, DCCRN"). BRV_DCCRN_BF16_ACT=0: fp32 tensors, round-5 path.
_BF16_ACT = os.environ.get('BRV_DCCRN_BF16_ACT', '1') != '0'
# ... and so does the convolution output of a block whose batch norm writes bf16 (every block with a norm but the first
# and the last encoder block): what torch.autocast makes of a convolution in the reference; a rounding point the fp32-
# output form does not have -- oracle/dccrn.py emulates it (ComplexWrapper.round_output). BRV_DCCRN_BF16_Y=0: fp32.
_BF16_Y = os.environ.get('BRV_DCCRN_BF16_Y', '1') != '0'
# use_amp: the output projections of the recurrent block on the bf16 MFMA (BRV_DCCRN_LINEAR_LOWP=0: exact-fp32 products)
_LINEAR_LOWP = os.environ.get('BRV_DCCRN_LINEAR_LOWP', '1') != '0'
# the gradient with respect to a bf16 activation is bf16 as well (what autocast hands backward; halves the bytes of the
# norm's backward passes and of the data-gradient kernels' stores): the token of a bf16 output is a bf16 tensor, so the
# engine expects -- and the consuming block returns -- a bf16 gradient. BRV_DCCRN_BF16_GRAD=0: fp32 gradients.
_BF16_GRAD = os.environ.get('BRV_DCCRN_BF16_GRAD', '1') != '0'
_LINEAR_FUSED = os.environ.get('BRV_DCCRN_LINEAR_FUSED', '1') != '0'   # (0: two linear nodes between transposed copies and a concatenation)
_TWO_TOKENS = os.environ.get('BRV_DCCRN_TWO_TOKENS', '1') != '0'     # (0: the two gradients of an encoder output summed by a pass)


def _token(shape, device, dtype=torch.float32):
    """What autograd tracks in place of a bf16 activation: an fp32 tensor of the logical shape that owns ONE element
    (stride 0). The engine checks gradients against the dtype and shape of the forward tensor -- a bf16 output would
    have its fp32 gradient cast to bf16 -- so every block returns (token, bf16 data) and reads its inputs' data from
    the second member; only gradients travel along the first."""
    return torch.empty(1, dtype=dtype, device=device).expand(shape)


def _bf16_empty(shape, device):
    """A bf16 image for the row kernels: 16 readable bytes in front of and behind it (their LDS-DMA descriptors reach
    that far: csrc/cconv_dma.cuh, include/brever_hip.h)."""
    n = 1
    for d in shape:
        n *= d
    return torch.empty(n + 16, dtype=torch.bfloat16, device=device)[8:8 + n].view(shape)


def _has_slack(t):
    """``t`` (bf16) sits at least 8 elements inside its storage on both sides (``_bf16_empty``)."""
    return t.storage_offset() >= 8 and t.untyped_storage().nbytes()//2 >= t.storage_offset() + t.numel() + 8


def _as_bf16(t):
    with torch.no_grad():
        out = _bf16_empty(t.shape, t.device)
        out.copy_(t)
        return out


class _BlockFunction(torch.autograd.Function):
    """EncoderBlock / DecoderBlock (dccrn.py:238-292) as ONE autograd node under ``use_amp``: complex (transposed)
    convolution by the row kernels on bf16 images, nn.BatchNorm2d (+ scalar nn.PReLU) with the bf16 output the next
    block reads; backward: the norm's pass writes the gradient with respect to the convolution output as bf16 (and
    its channel sums = the bias gradient), the data / weight gradient kernels read it. Arithmetic and rounding points
    are those of ``_ComplexConvFunction`` + ``_BatchNormActFunction`` on the rows path."""

    @staticmethod
    def forward(ctx, x, x16, skip, skip16, wr, br, wi, bi, gamma, beta, slope, norm, training, geom4, transpose,
                out_bf16, two_out=False):
        """``two_out``: a second token for the block's second consumer (an encoder output feeds the next block AND the
        decoder's skip input, dccrn.py:205-217): the two gradients arrive separately and the norm's backward pass adds
        them on the fly (``brv_batchnorm2d_backward_ex``: no pass that sums them)."""
        lib = hip.lib()
        (kh, kw) = geom4[0]
        khw = kh*kw
        dev = x.device
        # (the engine would hand backward a dense tensor of zeros for every output without a gradient -- the bf16 data
        # output of each block, 131 MB at the first levels: 0.27 ms of fills per step)
        ctx.set_materialize_grads(False)
        ctx.side_ok = _side_allowed((wr, br, wi, bi))
        B, C2, H, W = x.shape
        ctx.x_f32 = None
        first = x16 is None and not transpose and C2 < 8      # the first encoder block: its fp32 image as it is
        if x16 is None:
            x = x.contiguous()
            # (its weight gradient runs on the column-matrix kernel, which wants the fp32 image too)
            if first and any(ctx.needs_input_grad[4:8]):
                ctx.x_f32 = x
            x16 = x if first else _as_bf16(x)
        two = skip is not None
        if two and skip16 is None:
            skip16 = _as_bf16(skip)
        seg = C2//2 if two else 0
        Cin2 = 2*C2 if two else C2                 # real input channels of the convolution
        R, Cw = wr.shape[0], wr[0].numel()
        Cout = wr.shape[1] if transpose else R
        need_dx = ctx.needs_input_grad[0] or (two and ctx.needs_input_grad[2])
        fwd = (2*Cout, Cin2, khw, 2*Cw) if transpose else (2*Cout, Cin2, 2*Cw, khw)
        bwd = ((Cin2, 2*Cout, 2*Cw, khw) if transpose else (Cin2, 2*Cout, khw, 2*Cw)) if need_dx else None
        wc, bias, wp_fwd, ctx.wp_bwd = _pack_complex_layer(wr, wi, br, bi, -1.0 if transpose else 1.0, fwd, bwd)
        Ho, Wo = (2*H, W + 1) if transpose else (H//2, W - 1)
        y16 = ctx.y16 = bool(_BF16_Y and norm is not None and out_bf16 and not first)
        y = torch.empty(B, 2*Cout, Ho, Wo, dtype=torch.bfloat16 if y16 else torch.float32, device=dev)
        hip.check(lib.brv_cconv_rows_ex(hip.ptr(x16), hip.ptr(skip16), seg, hip.ptr(wp_fwd), hip.ptr(bias),
                                        hip.ptr(y), None, 0, B, Cin2, 2*Cout, H, W, int(transpose), int(not first),
                                        int(y16), hip.stream()), 'brv_cconv_rows_ex')
        assert first or (_has_slack(x16) and (skip16 is None or _has_slack(skip16))), 'bf16 image without slack'
        ctx.dx_bf16 = x.dtype == torch.bfloat16          # (the input is the bf16 token of a block: its gradient is bf16)
        assert not two or skip.dtype == x.dtype
        ctx.cfg = (geom4, transpose, (H, W), (Ho, Wo), Cin2, Cout, R, Cw, wr.shape, seg, two)
        ctx.has_norm = norm is not None
        ctx.training = training
        if norm is None:
            ctx.save_for_backward(x16, wc, *((skip16,) if two else ()))
            ctx.has_slope = False
            return y, None
        mean = torch.empty(2*Cout, dtype=torch.float32, device=dev)
        invstd = torch.empty_like(mean)
        args = (hip.ptr(y), hip.ptr(gamma), hip.ptr(beta), hip.ptr(norm.running_mean), hip.ptr(norm.running_var),
                hip.ptr(slope))
        tail = (hip.ptr(mean), hip.ptr(invstd), B, 2*Cout, Ho*Wo, float(norm.eps), float(norm.momentum),
                int(training), hip.stream())
        if out_bf16:
            a16 = _bf16_empty((B, 2*Cout, Ho, Wo), dev)
            fwd_fn = lib.brv_batchnorm2d_forward_bf16io if y16 else lib.brv_batchnorm2d_forward_bf16
            hip.check(fwd_fn(*args, hip.ptr(a16), *tail), 'brv_batchnorm2d_forward_bf16[io]')
            # (bf16 gradients need every encoder output to hand out one token per consumer: a forked token's two
            # gradients are summed by an fp32 pass)
            tdt = torch.bfloat16 if (_BF16_GRAD and _TWO_TOKENS) else torch.float32
            out = (_token(y.shape, dev, tdt), a16) + ((_token(y.shape, dev, tdt),) if two_out else ())
            ctx.mark_non_differentiable(a16)
        else:
            a = torch.empty_like(y)
            hip.check(lib.brv_batchnorm2d_forward(*args, hip.ptr(a), *tail), 'brv_batchnorm2d_forward')
            out = (a, None)
        if training:
            _BN_COUNTERS.append(norm.num_batches_tracked)
        ctx.has_slope = slope is not None
        ctx.save_for_backward(x16, wc, *((skip16,) if two else ()), y, gamma, beta, mean, invstd,
                              *((slope,) if slope is not None else ()))
        return out

    @staticmethod
    def backward(ctx, g, _g16=None, g2=None):
        lib = hip.lib()
        if g is None:
            g, g2 = g2, None
        if g is None:                       # nothing downstream asked for a gradient
            return (None,)*17
        geom4, transpose, (H, W), (Ho, Wo), Cin2, Cout, R, Cw, wshape, seg, two = ctx.cfg
        saved = list(ctx.saved_tensors)
        x16, wc = saved[:2]
        skip16 = saved[2] if two else None
        rest = saved[2 + int(two):]
        khw = geom4[0][0]*geom4[0][1]
        dev = g.device
        g = g.contiguous()
        g2 = g2.contiguous() if g2 is not None else None
        B = x16.shape[0]
        small_cin = ctx.x_f32 is not None
        dgamma = dbeta = dslope = None
        sums = torch.empty(2*Cout, dtype=torch.float32, device=dev)       # channel sums of d(conv output)
        if ctx.has_norm:
            if not ctx.training:
                raise NotImplementedError('gradient of eval-mode batch norm is not built on the HIP path')
            y, gamma, beta, mean, invstd = rest[:5]
            slope = rest[5] if ctx.has_slope else None
            dgamma, dbeta, dsl = (torch.empty(2*Cout, dtype=torch.float32, device=dev) for _ in range(3))
            dy = torch.empty_like(y) if small_cin else None
            dy16 = None if small_cin else _bf16_empty(y.shape, dev)
            assert g2 is None or g2.dtype == g.dtype
            hip.check(lib.brv_batchnorm2d_backward_ex(
                hip.ptr(y), int(ctx.y16), hip.ptr(g), hip.ptr(g2), int(g.dtype == torch.bfloat16), hip.ptr(mean),
                hip.ptr(invstd), hip.ptr(gamma),
                hip.ptr(beta), hip.ptr(slope), hip.ptr(dy if small_cin else dy16), int(not small_cin), hip.ptr(dgamma),
                hip.ptr(dbeta), hip.ptr(dsl), None if small_cin else hip.ptr(sums), B, 2*Cout, Ho*Wo, hip.stream()),
                'brv_batchnorm2d_backward_ex')
            if ctx.has_slope:
                dslope = torch.empty(1, dtype=torch.float32, device=dev)
                hip.check(lib.brv_row_sum(hip.ptr(dsl), hip.ptr(dslope), 1, 1, 2*Cout, hip.stream()), 'brv_row_sum')
        else:
            dy, dy16 = g, _as_bf16(g)

        def param_grads():
            """dW (complex pair) and the bias gradients: only dy, x (and skip) go in."""
            if small_cin:
                geom = geom4[:3]
                dwc = torch.empty_like(wc)
                _gemm_conv(dy, ctx.x_f32, dwc, 1, 2*Cout, 2*Cw, Ho*Wo, Ho*Wo, 2*Cw, 0, 0, 0, 1, (Cin2, H, W), geom,
                           (Ho, Wo), trans_b=1, kbatch=B, a_kbs=2*Cout*Ho*Wo, img_kbs=Cin2*H*W)
                return _ComplexConvFunction._unpack_param_grads(dwc, dy, wshape, R, Cw, Cout, B, Ho*Wo, transpose)
            if transpose:
                dwc = _cconv_wgrad(x16, dy16, small2=skip16)
            else:
                dwc = _cconv_wgrad(dy16, x16)
            if dy is not None:          # (no norm behind the convolution: the sums of the fp32 gradient)
                hip.check(lib.brv_row_sum(hip.ptr(dy), hip.ptr(sums), B, 2*Cout, Ho*Wo, hip.stream()), 'brv_row_sum')
            dwr = torch.empty(wshape, dtype=torch.float32, device=dev)
            dwi = torch.empty_like(dwr)
            hip.check(lib.brv_complex_weight_unpack(hip.ptr(dwc), hip.ptr(dwr), hip.ptr(dwi), R, Cw,
                                                    -1.0 if transpose else 1.0, hip.stream()),
                      'brv_complex_weight_unpack')
            dbr = torch.empty(Cout, dtype=torch.float32, device=dev)
            dbi = torch.empty_like(dbr)
            hip.check(lib.brv_complex_bias_unpack(hip.ptr(sums), hip.ptr(dbr), hip.ptr(dbi), Cout, hip.stream()),
                      'brv_complex_bias_unpack')
            return dwr, dwi, dbr, dbi

        side = _side_stream(dev) if ctx.side_ok else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(dev))
        need_dx = ctx.needs_input_grad[0] or (two and ctx.needs_input_grad[2])
        dx = dskip = None
        if need_dx:
            src = dy16 if dy16 is not None else _as_bf16(dy)
            assert _has_slack(src), 'bf16 image without slack'
            M = Cin2
            shape = (B, M//2 if two else M) + ((H, W))
            dx = torch.empty(shape, dtype=torch.bfloat16 if ctx.dx_bf16 else torch.float32, device=dev)
            dskip = torch.empty_like(dx) if two else None
            # the data gradient of a (transposed) convolution is the other form with the same weights
            hip.check(lib.brv_cconv_rows_ex(hip.ptr(src), None, 0, hip.ptr(ctx.wp_bwd), None, hip.ptr(dx),
                                            hip.ptr(dskip), M//4 if two else 0, B, 2*Cout, M, Ho, Wo,
                                            int(not transpose), 1, int(ctx.dx_bf16), hip.stream()), 'brv_cconv_rows_ex')
        if side is None:
            dwr, dwi, dbr, dbi = param_grads()
        else:
            with torch.cuda.stream(side):
                dwr, dwi, dbr, dbi = param_grads()
            for t in (x16, skip16, dy, dy16, sums, ctx.x_f32):
                if t is not None:
                    t.record_stream(side)
            if not _side['pending'].get(dev.index):
                _side['pending'][dev.index] = True
                torch.autograd.Variable._execution_engine.queue_callback(lambda: _join_side(dev))
        return (dx, None, dskip, None, dwr, dbr, dwi, dbi, dgamma, dbeta, dslope, None, None, None, None, None, None)


class _CplxMomentsFunction(torch.autograd.Function):
    """x (B, 2C, H, W) -> (5, C): per-channel means of xr, xi, xr^2, xi^2, xr*xi."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, C2, H, W = x.shape
        m = torch.empty(5, C2//2, dtype=torch.float32, device=x.device)
        hip.check(hip.lib().brv_cplx_moments(hip.ptr(x), hip.ptr(m), B, C2//2, H*W, hip.stream()),
                  'brv_cplx_moments')
        ctx.save_for_backward(x)
        return m

    @staticmethod
    def backward(ctx, gm):
        x, = ctx.saved_tensors
        B, C2, H, W = x.shape
        gm = (gm.float()/(B*H*W)).contiguous()
        dx = torch.empty_like(x)
        hip.check(hip.lib().brv_cplx_moments_backward(hip.ptr(x), hip.ptr(gm), hip.ptr(dx), B, C2//2,
                                                      H*W, hip.stream()), 'brv_cplx_moments_backward')
        return dx


class _CplxAffineFunction(torch.autograd.Function):
    """y = A x + o per complex channel (A (4, C), o (2, C)) then the optional scalar PReLU."""

    @staticmethod
    def forward(ctx, x, A, o, slope):
        x, A, o = x.contiguous(), A.contiguous(), o.contiguous()
        B, C2, H, W = x.shape
        y = torch.empty_like(x)
        hip.check(hip.lib().brv_cplx_affine_forward(hip.ptr(x), hip.ptr(A), hip.ptr(o), hip.ptr(slope),
                                                    hip.ptr(y), B, C2//2, H*W, hip.stream()),
                  'brv_cplx_affine_forward')
        ctx.save_for_backward(x, A, o, slope if slope is not None else A.new_zeros(0))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, A, o, slope = ctx.saved_tensors
        has_slope = slope.numel() > 0
        dy = dy.contiguous()
        B, C2, H, W = x.shape
        C = C2//2
        dx = torch.empty_like(x)
        dA, do = torch.empty_like(A), torch.empty_like(o)
        dsl = torch.empty(C, dtype=torch.float32, device=x.device)
        hip.check(hip.lib().brv_cplx_affine_backward(
            hip.ptr(x), hip.ptr(dy), hip.ptr(A), hip.ptr(o), hip.ptr(slope) if has_slope else None,
            hip.ptr(dx), hip.ptr(dA), hip.ptr(do), hip.ptr(dsl), B, C, H*W, hip.stream()),
            'brv_cplx_affine_backward')
        dslope = None
        if has_slope:
            dslope = torch.empty(1, dtype=torch.float32, device=x.device)
            hip.check(hip.lib().brv_row_sum(hip.ptr(dsl), hip.ptr(dslope), 1, 1, C, hip.stream()),
                      'brv_row_sum')
        return dx, dA, do, dslope


def _complex_batch_norm(x, norm, slope):
    """ComplexBatchNorm2d.forward (complex_batchnorm.py:76-215) + the block's PReLU. The reductions
    over the tensor and the normalisation itself are HIP kernels; the 2x2 inverse square root of
    the covariance and its composition with the affine weights are per-channel scalars (a few
    element-wise torch operations on (C,) vectors that also carry their gradients)."""
    training = norm.training
    if training:
        m = _CplxMomentsFunction.apply(x)
        mean = m[:2]
        vrr, vii = m[2] - m[0]*m[0] + norm.eps, m[3] - m[1]*m[1] + norm.eps
        vri = m[4] - m[0]*m[1]
        with torch.no_grad():
            norm.num_batches_tracked += 1
            norm.running_mean += norm.momentum*(mean - norm.running_mean)
            cov = torch.stack([vrr, vri, vri, vii]).reshape(2, 2, -1)
            norm.running_var += norm.momentum*(cov - norm.running_var)
    else:
        mean = norm.running_mean
        vrr, vri, _, vii = norm.running_var.reshape(4, -1)
    s = torch.sqrt(vrr*vii - vri*vri)
    t = torch.sqrt(vrr + vii + 2*s)
    denom = t*s
    p, q, r, s2 = (vii + s)/denom, -vri/denom, -vri/denom, (vrr + s)/denom
    w0, w1, w2 = norm.weight
    a_rr, a_ri = p*w0 + q*w1, r*w0 + s2*w1
    a_ir, a_ii = p*w1 + q*w2, r*w1 + s2*w2
    A = torch.stack([a_rr, a_ri, a_ir, a_ii])
    o = torch.stack([norm.bias[0] - (a_rr*mean[0] + a_ri*mean[1]),
                     norm.bias[1] - (a_ir*mean[0] + a_ii*mean[1])])
    return _CplxAffineFunction.apply(x, A, o, slope)


class _LSTMFunction(torch.autograd.Function):
    """G independent single-layer unidirectional nn.LSTMs (batch_first, zero initial state) in
    one set of launches: x (G, B, T, I) -> hidden states (G, B, T, H), parameters stacked on a
    leading group axis (w_ih (G, 4H, I), w_hh (G, 4H, H), b_ih / b_hh (G, 4H)). The input
    projections and their gradients are batched matrix products over G; the recurrence runs
    all G*B chains concurrently, one workgroup each."""

    TILE_MIN_CHAINS = 256       # below this the one-chain-per-workgroup kernels have more parallelism
    # use_amp: the tile kernels multiply on the bf16 MFMA (a step of 16 chains = 16 instructions per wave); DCCRN's 64
    # long chains then run as 4 workgroups
    TILE_MIN_CHAINS_LOWP = int(os.environ.get('BRV_LSTM_TILE_MIN_LOWP', '256'))

    @staticmethod
    def _interleave(w, H):
        """Rows gate*H + unit -> 4*unit + gate (the gate layout of brv_lstm_tile_*)."""
        G = w.shape[0]
        return w.view(G, 4, H, *w.shape[2:]).transpose(1, 2).reshape(w.shape).contiguous()

    @staticmethod
    def _deinterleave(w, H):
        G = w.shape[0]
        return w.view(G, H, 4, *w.shape[2:]).transpose(1, 2).reshape(w.shape).contiguous()

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh):
        lib = hip.lib()
        x, w_ih, w_hh = x.contiguous(), w_ih.contiguous(), w_hh.contiguous()
        G, B, T, I = x.shape
        # x of shape (1, B, T, I) with several weight sets: ONE input shared by all groups (batch stride 0 in the
        # products; its gradient is the sum over the groups) -- DCCRN's complex LSTM feeds [real; imag] to both modules
        shared = ctx.shared = G == 1 and w_hh.shape[0] > 1
        if shared:
            G = w_hh.shape[0]
        H = w_hh.shape[-1]
        lowp = ctx.lowp = _AMP['on']
        # many short chains (TF-GridNet): 16 chains per workgroup on the exact-fp32 MFMA, gates
        # interleaved; few long chains (DCCRN): one workgroup per chain
        tiled = ctx.tiled = bool(lib.brv_lstm_tile_supported(H)) and \
            G*B >= (_LSTMFunction.TILE_MIN_CHAINS_LOWP if lowp else _LSTMFunction.TILE_MIN_CHAINS)
        if tiled:
            w_ih = _LSTMFunction._interleave(w_ih, H)
        gates = torch.empty(G, B, T, 4*H, dtype=torch.float32, device=x.device)
        _gemm(x, w_ih, gates, G, B*T, 4*H, I, I, I, 4*H, 0 if shared else B*T*I, 4*H*I, B*T*4*H, trans_b=1,
              lowp=lowp)
        bias = _combine(b_ih.detach().contiguous(), b_hh.detach().contiguous(), 1.0)
        y = torch.empty(G, B, T, H, dtype=torch.float32, device=x.device)
        act = torch.empty(G, B, T, 4*H, dtype=torch.float32, device=x.device)
        cs = torch.empty(G, B, T, H, dtype=torch.float32, device=x.device)
        # use_amp, few long chains: the step's matrix-vector product on the bf16 MFMA (csrc/dccrn.hip lstm_*_mv_kernel)
        mv = ctx.mv = bool(lowp and not tiled and _LSTM_MV and lib.brv_lstm_recurrent_bf16_supported(H))
        fn, name = (lib.brv_lstm_tile_forward, 'brv_lstm_tile_forward') if tiled else \
            (lib.brv_lstm_recurrent_forward_bf16, 'brv_lstm_recurrent_forward_bf16') if mv else \
            (lib.brv_lstm_recurrent_forward, 'brv_lstm_recurrent_forward')
        extra = (0, H, B*T*H, int(lowp)) if tiled else ()   # no reversed group, (G, B, T, H) output
        hip.check(fn(hip.ptr(gates), hip.ptr(w_hh), hip.ptr(bias), hip.ptr(y), hip.ptr(act),
                     hip.ptr(cs), G*B, T, H, G, *extra, hip.stream()), name)
        ctx.save_for_backward(x, w_ih, w_hh, y, act, cs)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = hip.lib()
        x, w_ih, w_hh, y, act, cs = ctx.saved_tensors          # w_ih interleaved when tiled
        G, B, T, I = x.shape
        shared = ctx.shared
        if shared:
            G = w_hh.shape[0]
        H = w_hh.shape[-1]
        lowp = ctx.lowp
        dy = dy.contiguous()
        dg = torch.empty(G, B, T, 4*H, dtype=torch.float32, device=x.device)
        fn, name = (lib.brv_lstm_tile_backward, 'brv_lstm_tile_backward') if ctx.tiled else \
            (lib.brv_lstm_recurrent_backward_bf16, 'brv_lstm_recurrent_backward_bf16') if ctx.mv else \
            (lib.brv_lstm_recurrent_backward, 'brv_lstm_recurrent_backward')
        extra = (0, H, B*T*H, int(lowp)) if ctx.tiled else ()
        hip.check(fn(hip.ptr(act), hip.ptr(cs), hip.ptr(w_hh), hip.ptr(dy), hip.ptr(dg), G*B, T, H,
                     G, *extra, hip.stream()), name)
        BT = B*T
        dx = torch.empty_like(x)                                   # dg (BT, 4H) @ W_ih (4H, I)
        if shared:                                                 # (summed over the groups: they are the k-batches)
            _gemm(dg, w_ih, dx, 1, BT, I, 4*H, 4*H, I, I, 0, 0, 0, kbatch=G, a_kbs=BT*4*H, b_kbs=4*H*I, lowp=lowp)
        else:
            _gemm(dg, w_ih, dx, G, BT, I, 4*H, 4*H, I, I, BT*4*H, 4*H*I, BT*I, lowp=lowp)
        dw_ih = torch.empty_like(w_ih)                             # dg^T (4H, BT) @ x (BT, I)
        _gemm(dg, x, dw_ih, G, 4*H, I, BT, 4*H, I, I, BT*4*H, 0 if shared else BT*I, 4*H*I, trans_a=1, lowp=lowp)
        # recurrent weight gradient: the gate gradients of frames 1.. against the hidden states of frames 0..T-2 of the
        # same chain, through shifted views (round 5: a zero-filled shifted copy of y was built per call -- a 16 MB fill
        # and a 16 MB copy in front of every product; the chains are the k-batches of the product instead)
        dw_hh = torch.zeros_like(w_hh) if T == 1 else torch.empty_like(w_hh)
        if T > 1:
            # (one launch: the groups are the batch, the chains of a group the k-batches)
            _gemm(dg.view(-1)[4*H:], y, dw_hh, G, 4*H, H, T - 1, 4*H, H, H, B*T*4*H, B*T*H, 4*H*H, trans_a=1,
                  kbatch=B, a_kbs=T*4*H, b_kbs=T*H, lowp=lowp)
        # bias gradients: column sums of dg (BT, 4H) per group, in a fixed order (no transposed copy)
        db = torch.empty(G, 4*H, dtype=torch.float32, device=x.device)
        scratch = torch.empty(lib.brv_col_sum_scratch_bytes(G, 4*H), dtype=torch.uint8, device=x.device)
        hip.check(lib.brv_col_sum(hip.ptr(dg), hip.ptr(db), hip.ptr(scratch), G, BT, 4*H, hip.stream()),
                  'brv_col_sum')
        if ctx.tiled:
            dw_ih, dw_hh, db = (_LSTMFunction._deinterleave(t, H) for t in (dw_ih, dw_hh, db))
        return dx, dw_ih, dw_hh, db, db.clone()


class _LinearLowpFunction(torch.autograd.Function):
    """``ffnn._LinearFunction`` (nn.Linear on the feature axis of (B, features, frames): LSTMBlock.linear_r / linear_i,
    dccrn.py:293-311) with the three products on bf16-rounded operands, fp32 accumulation: what torch.autocast makes of
    nn.Linear; ``use_amp`` only (oracle/dccrn.py: _LSTMBlock._linear emulates it)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        B, I, T = x.shape
        O = weight.shape[0]
        x = x.contiguous()
        y = torch.empty(B, O, T, dtype=torch.float32, device=x.device)
        _gemm(weight, x, y, B, O, T, I, I, T, T, 0, I*T, O*T, bias=bias, lowp=True)
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
            _gemm(weight, dy, dx, B, I, T, O, I, T, T, 0, O*T, I*T, trans_a=1, lowp=True)
        dw = torch.empty_like(weight)                      # sum_b dy[b] (O, T) @ x[b]^T (T, I)
        _gemm(dy, x, dw, 1, O, I, T, T, T, I, 0, 0, 0, trans_b=1, kbatch=B, a_kbs=O*T, b_kbs=I*T, lowp=True)
        db = torch.empty(O, dtype=torch.float32, device=x.device)
        hip.check(hip.lib().brv_row_sum(hip.ptr(dy), hip.ptr(db), B, O, T, hip.stream()), 'brv_row_sum')
        return dx, dw, db


class _ComplexLinearFunction(torch.autograd.Function):
    """LSTMBlock.linear_r / linear_i (dccrn.py:293-311) on the recurrent block's (B, T, H) outputs, written straight into
    the two halves of the (B, 2 F, T) decoder input: the products read their operands transposed in place (no
    ``transpose().contiguous()`` copies in front, no concatenation behind, none of their gradients' copies)."""

    @staticmethod
    def forward(ctx, real, imag, wr, br, wi, bi, lowp):
        real, imag = real.contiguous(), imag.contiguous()
        B, T, H = real.shape
        F_ = wr.shape[0]
        out = torch.empty(B, 2*F_, T, dtype=torch.float32, device=real.device)
        for k, (x, w, b) in enumerate(((real, wr, br), (imag, wi, bi))):
            # out[b][k F : (k + 1) F] (F x T) = W (F x H) @ x[b]^T (H x T)
            _gemm(w, x, out[:, k*F_:], B, F_, T, H, H, H, T, 0, T*H, 2*F_*T, trans_b=1, bias=b, lowp=lowp)
        ctx.save_for_backward(real, imag, wr, wi)
        ctx.lowp = lowp
        return out

    @staticmethod
    def backward(ctx, dy):
        real, imag, wr, wi = ctx.saved_tensors
        B, T, H = real.shape
        F_ = wr.shape[0]
        dy = dy.contiguous()
        lowp = ctx.lowp
        grads = []
        for k, (x, w) in enumerate(((real, wr), (imag, wi))):
            d = dy[:, k*F_:]                                   # (B, F, T) view: rows T apart, items 2 F T apart
            dx = torch.empty_like(x)                           # dx[b] (T x H) = d[b]^T (T x F) @ W (F x H)
            _gemm(d, w, dx, B, T, H, F_, T, H, H, 2*F_*T, 0, T*H, trans_a=1, lowp=lowp)
            dw = torch.empty_like(w)                           # sum_b d[b] (F x T) @ x[b] (T x H)
            _gemm(d, x, dw, 1, F_, H, T, T, H, H, 0, 0, 0, kbatch=B, a_kbs=2*F_*T, b_kbs=T*H, lowp=lowp)
            grads.append((dx, dw))
        sums = torch.empty(2*F_, dtype=torch.float32, device=dy.device)
        hip.check(hip.lib().brv_row_sum(hip.ptr(dy), hip.ptr(sums), B, 2*F_, T, hip.stream()), 'brv_row_sum')
        return grads[0][0], grads[1][0], grads[0][1], sums[:F_], grads[1][1], sums[F_:], None


class _ApplyMaskFunction(torch.autograd.Function):
    """DCCRN.apply_mask (dccrn.py:96-109): x, mask (B, 2, Fq, T) -> complex (B, 1, Fq, T)."""

    @staticmethod
    def forward(ctx, x, mask):
        x, mask = x.contiguous(), mask.contiguous()
        B, _, Fq, T = x.shape
        n = Fq*T
        out = torch.empty(B, n, 2, dtype=torch.float32, device=x.device)
        hip.check(hip.lib().brv_dccrn_apply_mask_batched(hip.ptr(x), hip.ptr(mask), hip.ptr(out), B, n,
                                                         hip.stream()), 'brv_dccrn_apply_mask_batched')
        ctx.save_for_backward(x, mask)
        return torch.view_as_complex(out.view(B, 1, Fq, T, 2))

    @staticmethod
    def backward(ctx, g):
        x, mask = ctx.saved_tensors
        B, _, Fq, T = x.shape
        n = Fq*T
        gr = torch.view_as_real(g.to(torch.complex64).contiguous()).view(B, n, 2)
        dm = torch.empty_like(mask)
        hip.check(hip.lib().brv_dccrn_apply_mask_backward_batched(
            hip.ptr(x), hip.ptr(mask), hip.ptr(gr), hip.ptr(dm), B, n, hip.stream()),
            'brv_dccrn_apply_mask_backward_batched')
        return None, dm


@ModelRegistry.register('dccrn')
class DCCRN(BreverBaseModel):
    _fused_adam = True       # clip + Adam as brv_clip_adam_step2 on one flat buffer (models/base.py)

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
    def _complex_conv(self, x, wrap, transpose, skip=None):
        mr, mi = wrap.module_real, wrap.module_imag
        return _ComplexConvFunction.apply(x, mr.weight, mr.bias, mi.weight, mi.bias,
                                          self.mask_net.geom, transpose, skip)

    def _norm_act(self, x, norm, act):
        if norm is None:
            return x
        if isinstance(norm, ComplexBatchNorm2d):
            return _complex_batch_norm(x, norm, act.weight if act is not None else None)
        training = norm.training and norm.track_running_stats
        return _BatchNormActFunction.apply(x, norm.weight, norm.bias,
                                           act.weight if act is not None else None, norm, training)

    @staticmethod
    def _lstm(lstms, xs):
        """The single-layer LSTMs ``lstms`` on their inputs ``xs`` (same shapes), concurrently."""
        stack = lambda name: torch.stack([getattr(m, name) for m in lstms])  # noqa: E731
        return _LSTMFunction.apply(xs if torch.is_tensor(xs) else torch.stack(xs), stack('weight_ih_l0'), stack('weight_hh_l0'),
                                   stack('bias_ih_l0'), stack('bias_hh_l0'))

    def _lstm_block(self, x):
        """LSTMBlock (dccrn.py:293-311) on the encoder output (B, 2C, Fq, T) -> same shape."""
        from .ffnn import _LinearFunction
        blk = self.mask_net.lstm
        B, C2, Fq, T = x.shape
        rows = x.reshape(B, C2*Fq, T).transpose(1, 2)        # (B, T, features): real then imag
        real, imag = rows.chunk(2, dim=-1)
        for layer in blk.lstm.layers:
            # each module sees both halves; both modules run in one set of launches (4B chains)
            # (round 5: ONE concatenation [real; imag] shared by both modules instead of two concatenations and a stack)
            out = self._lstm((layer.module_real, layer.module_imag), torch.cat([real, imag], dim=0).unsqueeze(0))
            if out.is_cuda and out.numel() % 16 == 0:          # (each of the four parts in whole 16-byte pieces)
                real, imag = _ComplexMixFunction.apply(out)
            else:
                (rr, ri), (ir, ii) = out[0].chunk(2, dim=0), out[1].chunk(2, dim=0)
                real, imag = _CombineFunction.apply(rr, ii, -1.0), _CombineFunction.apply(ri, ir, 1.0)
        # Linear applied on the feature axis of (B, features, T): the output is already in the
        # (channels*freqs, frames) layout of the decoder input
        if _LINEAR_FUSED and real.is_cuda:
            out = _ComplexLinearFunction.apply(real, imag, blk.linear_r.weight, blk.linear_r.bias, blk.linear_i.weight,
                                               blk.linear_i.bias, bool(_AMP['on'] and _LINEAR_LOWP))
            return out.view(B, C2, Fq, T)
        linear = _LinearLowpFunction if (_AMP['on'] and _LINEAR_LOWP) else _LinearFunction
        out_r = linear.apply(real.transpose(1, 2).contiguous(), blk.linear_r.weight, blk.linear_r.bias)
        out_i = linear.apply(imag.transpose(1, 2).contiguous(), blk.linear_i.weight, blk.linear_i.bias)
        return torch.cat([out_r, out_i], dim=1).view(B, C2, Fq, T)

    def _blocks_ok(self, x):
        """Every encoder / decoder block on the bf16-activation path (``_BlockFunction``): use_amp, the row kernels'
        geometry, plain batch norms, skip segments in whole chunks of 8 channels, planes the 16-byte norm passes take."""
        net = self.mask_net
        if not (_AMP['on'] and _ROWS and _BF16_ACT and x.is_cuda and tuple(map(tuple, net.geom)) == _ROWS_GEOM):
            return False
        H, W = x.shape[-2:]
        for blk in net.encoder:
            if isinstance(blk.norm, ComplexBatchNorm2d) or H % 2 or W < 2:
                return False
            H, W = H//2, W - 1
            if (H*W) % 4 or (blk.norm.num_features//2) % 8:
                return False
        for blk in net.decoder:
            H, W = 2*H, W + 1
            if isinstance(blk.norm, ComplexBatchNorm2d) or (blk.norm is not None and (H*W) % 4):
                return False
        return net.decoder[-1].norm is None       # (the mask leaves the network as fp32 data)

    def _block(self, x, x16, skip, skip16, blk, transpose, out_bf16, two_out=False):
        mr, mi = blk.conv.module_real, blk.conv.module_imag
        norm, act = blk.norm, blk.activation
        training = norm is not None and norm.training and norm.track_running_stats
        return _BlockFunction.apply(x, x16, skip, skip16, mr.weight, mr.bias, mi.weight, mi.bias,
                                    norm.weight if norm is not None else None, norm.bias if norm is not None else None,
                                    act.weight if act is not None else None, norm, training, self.mask_net.geom,
                                    transpose, out_bf16, two_out)

    def _mask_net_blocks(self, x):
        """``_mask_net`` with bf16 activations between the blocks: every block returns (token, data) -- ``_token``."""
        net = self.mask_net
        _side['pending'][x.device.index] = False
        _BN_COUNTERS.clear()
        tok, a16 = x, None
        skips = []
        for k, blk in enumerate(net.encoder):
            if k + 1 < len(net.encoder) and _TWO_TOKENS:     # bf16 output, one token per consumer (next block, decoder's skip input)
                tok, a16, s_tok = self._block(tok, a16, None, None, blk, False, True, two_out=True)
            else:                            # the last encoder output feeds the recurrent block: fp32 data, forked
                tok, a16 = self._block(tok, a16, None, None, blk, False, k + 1 < len(net.encoder))
                s_tok = tok
                if tok.requires_grad:
                    tok, s_tok = _ForkFunction.apply(tok)
            skips.append((s_tok, a16))
        tok, a16 = self._lstm_block(tok), None
        for blk, (s_tok, s16) in zip(net.decoder, reversed(skips)):
            tok, a16 = self._block(tok, a16, s_tok, s16, blk, True, blk.norm is not None)
        _flush_bn_counters()
        return tok

    def _mask_net(self, x):
        net = self.mask_net
        if self._blocks_ok(x):
            return self._mask_net_blocks(x)
        if x.is_cuda:
            # a backward pass that raised never ran its end-of-pass callback: the "join queued" mark of this device must
            # not survive into the next pass, or that pass would queue no join at all
            _side['pending'][x.device.index] = False
        _BN_COUNTERS.clear()                          # (a forward that raised: its counts are dropped)
        encoder_outputs = []
        for blk in net.encoder:
            x = self._norm_act(self._complex_conv(x, blk.conv, False), blk.norm, blk.activation)
            skip = x
            if x.requires_grad:
                x, skip = _ForkFunction.apply(x)
            encoder_outputs.append(skip)
        x = self._lstm_block(x)
        for blk, enc in zip(net.decoder, reversed(encoder_outputs)):
            # torch.cat([real, skip_real, imag, skip_imag], dim=1) (dccrn.py:213-217) inside the function
            x = self._complex_conv(x, blk.conv, True, skip=enc)
            x = self._norm_act(x, blk.norm, blk.activation)
        _flush_bn_counters()
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
        _AMP['on'] = bool(use_amp)      # bf16-operand matrix products (the reference autocasts)
        try:
            outputs = self(inputs)
        finally:
            _AMP['on'] = False
        return self.criterion(outputs, labels, lengths).mean()

    def gather_grads(self):
        dev = next(self.parameters()).device
        if dev.type == 'cuda':
            _join_side(dev)              # (unconditional: cheap, and right after a backward pass that raised)
        return super().gather_grads()

    def update(self, loss, scaler, **kwargs):
        kwargs.setdefault('grad_clip', 5.0)
        super().update(loss, scaler, **kwargs)

    def _after_backward(self):
        """Called by ``BreverBaseModel.update`` right behind ``backward``: whichever branch reads the gradients next
        (the flat gather, the packed all-reduce of a sub-network optimizer, ``clip_grad_norm_``) finds the side
        stream's weight gradients complete -- also after a backward pass whose engine callback never ran."""
        dev = next(self.parameters()).device
        if dev.type == 'cuda':
            _join_side(dev)

    @property
    def latency(self):
        """Algorithmic latency in samples (dccrn.py:136-142): one STFT frame plus the look-ahead of
        the encoder / decoder convolutions along the frame axis."""
        _, kernel_size = self.kernel_size
        _, stride = self.stride
        enc_dec = (kernel_size - 1)*sum(stride**i for i in range(len(self.channels)))
        return self.stft.frame_length + enc_dec*self.stft.hop_length

    def _enhance(self, x, use_amp):
        _AMP['on'] = bool(use_amp)
        try:
            with torch.no_grad():
                return self.forward(x.mean(axis=-2))
        finally:
            _AMP['on'] = False
