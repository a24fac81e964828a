"""Differentiable forward of the SGMSE+ score network for training (``SGMSEp.loss``).

The inference path of ``sgmse.py`` launches bare kernels; here the same operators are
``torch.autograd.Function`` pairs whose two sides call ``libbrever_hip.so`` (fp32: column
matrix + exact-fp32 MFMA products for the convolutions and their weight / data gradients,
``brv_groupnorm_fold / brv_affine_act / brv_groupnorm_backward``, ``brv_silu*``,
``brv_fir_resample2d`` in both directions, ``brv_softmax_rows*``). The block structure follows
brever/models/sgmse/net.py:232-452 exactly as the inference path does; torch only
concatenates, slices and carries the autograd graph.
"""
import os

import torch

from .. import hip


AMP = {'on': False}       # set by SGMSEp.loss around the forward pass (use_amp)


def _gemm(a, b, d, batch, M, N, K, lda, ldb, ldd, a_bs, b_bs, d_bs, trans_a=0, trans_b=0,
          kbatch=1, a_kbs=0, b_kbs=0, bias=None, lowp=False):
    """``lowp``: bf16 operands with fp32 accumulation (the convolutions under ``use_amp``); a bf16 ``b`` or ``d``
    tensor (the column matrices of those convolutions) selects ``brv_gemm_bf16_mixed``."""
    flags = int(b.dtype == torch.bfloat16) | int(d.dtype == torch.bfloat16) << 1
    if flags:
        assert lowp
        hip.check(hip.lib().brv_gemm_bf16_mixed(
            hip.ptr(a), hip.ptr(b), hip.ptr(d), batch, M, N, K, lda, ldb, ldd, a_bs, b_bs, d_bs,
            trans_a, trans_b, kbatch, a_kbs, b_kbs, hip.ptr(bias), 0, flags, hip.stream()), 'brv_gemm_bf16_mixed')
        return
    if not lowp:
        hip.gemm_f32(a, b, d, batch, M, N, K, lda, ldb, ldd, a_bs, b_bs, d_bs, trans_a, trans_b, kbatch, a_kbs,
                     b_kbs, bias, 0)
        return
    hip.check(hip.lib().brv_gemm_bf16(
        hip.ptr(a), hip.ptr(b), hip.ptr(d), batch, M, N, K, lda, ldb, ldd, a_bs, b_bs, d_bs,
        trans_a, trans_b, kbatch, a_kbs, b_kbs, hip.ptr(bias), 0, hip.stream()), 'brv_gemm_bf16')


# use_amp: the explicit column matrices (and the column-matrix gradient) in bf16 -- the products round them to bf16
# anyway; half the bytes (and half the kept memory) of the largest tensors of a convolution. Off by default: measured
# 65.9 against 61.9 ms per step at 4 x 1 s (the mixed-type product's loader costs more than the bytes it saves)
_COL_BF16 = os.environ.get('BRV_SGMSE_COL_BF16', '0') == '1'

# use_amp convolutions with the column matrix read in place (no 9x copy in HBM: memory for larger batches). Off by
# default: measured 73.3 against 69.7 ms per step at 4 x 1 s -- the virtual-column loader of gemm_bf16_kernel costs
# more than the im2col / col2im passes it removes (155 us per product against 59 + 59)
_IMPLICIT = os.environ.get('BRV_SGMSE_TRAIN_IMPLICIT', '0') == '1'


def _gemm_conv(a, img, d, batch, M, N, K, lda, ldd, a_bs, img_bs, d_bs, C, H, W, k, trans_b=0, kbatch=1, a_kbs=0,
               img_kbs=0, bias=None):
    """``_gemm(lowp=True)`` whose B operand is the im2col matrix of ``img`` (C, H, W per item; k x k window,
    stride 1, padding k//2) -- never written out (``brv_gemm_bf16_conv``)."""
    hip.check(hip.lib().brv_gemm_bf16_conv(
        hip.ptr(a), hip.ptr(img), hip.ptr(d), batch, M, N, K, lda, ldd, a_bs, img_bs, d_bs, 0, trans_b,
        kbatch, a_kbs, img_kbs, hip.ptr(bias), 0, 1, C, H, W, k, k, 1, 1, k//2, k//2, H, W, hip.stream()),
        'brv_gemm_bf16_conv')


# Column matrices kept from the forward pass for the weight gradient (instead of a second im2col in backward): up to
# this many bytes per network evaluation -- the device has 288 GB, the default network at 4 x 1 s keeps 9 GB; past
# the budget (long inputs, large batches) a convolution falls back to rebuilding its column matrix
_COL_BUDGET = int(float(os.environ.get('BRV_SGMSE_COL_CACHE_GB', '48'))*2**30)
_col_kept = [0]


def _empty(*shape, like):
    return torch.empty(*shape, dtype=torch.float32, device=like.device)


class ConvFn(torch.autograd.Function):
    """nn.Conv2d (stride 1, 'same' padding k//2): y_b = W (Cout x Cin*k*k) @ col_b + bias."""

    @staticmethod
    def forward(ctx, x, w, bias):
        x = x.contiguous()
        B, Cin, H, W = x.shape
        Cout, _, k, _ = w.shape
        K, HW = Cin*k*k, H*W
        y = _empty(B, Cout, H, W, like=x)
        ctx.lowp = AMP['on']
        ctx.save_for_backward(x, w)
        if ctx.lowp and k > 1 and _IMPLICIT:
            _gemm_conv(w, x, y, B, Cout, HW, K, K, HW, 0, Cin*HW, Cout*HW, Cin, H, W, k, bias=bias)
            return y
        col = ConvFn._col(x, k, ctx.lowp)
        _gemm(w, col, y, B, Cout, HW, K, K, HW, HW, 0, K*HW, Cout*HW, bias=bias, lowp=ctx.lowp)
        ctx.col = None
        if k > 1 and any(ctx.needs_input_grad[:2]):
            nbytes = col.numel()*col.element_size()
            if _col_kept[0] + nbytes <= _COL_BUDGET:
                _col_kept[0] += nbytes
                ctx.col = col
        return y

    @staticmethod
    def _col(x, k, lowp=False):
        if k == 1:
            return x
        B, Cin, H, W = x.shape
        half = lowp and _COL_BF16
        col = torch.empty(B, Cin*k*k, H*W, dtype=torch.bfloat16 if half else torch.float32, device=x.device)
        fn, name = (hip.lib().brv_im2col_bf16, 'brv_im2col_bf16') if half else (hip.lib().brv_im2col, 'brv_im2col')
        hip.check(fn(hip.ptr(x), hip.ptr(col), B, Cin, H, W, k, k, 1, 1, k//2, k//2, H, W, hip.stream()), name)
        return col

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        B, Cin, H, W = x.shape
        Cout, _, k, _ = w.shape
        K, HW = Cin*k*k, H*W
        db = _empty(Cout, like=x)
        hip.check(hip.lib().brv_row_sum(hip.ptr(dy), hip.ptr(db), B, Cout, HW, hip.stream()),
                  'brv_row_sum')
        if ctx.lowp and k > 1 and _IMPLICIT:
            # both gradients as products with a column matrix read in place: dW = dy @ col(x)^T summed over the
            # batch; dx = the same convolution of dy with the window rotated by 180 degrees and (Cout, Cin) swapped
            dw = torch.empty_like(w)
            _gemm_conv(dy, x, dw, 1, Cout, K, HW, HW, K, 0, 0, 0, Cin, H, W, k, trans_b=1, kbatch=B,
                       a_kbs=Cout*HW, img_kbs=Cin*HW)
            dx = None
            if ctx.needs_input_grad[0]:
                wt = w.flip(2, 3).transpose(0, 1).reshape(Cin, Cout*k*k).contiguous()
                dx = torch.empty_like(x)
                _gemm_conv(wt, dy, dx, B, Cin, HW, Cout*k*k, Cout*k*k, HW, 0, Cout*HW, Cin*HW, Cout, H, W, k)
            return dx, dw, db
        col = getattr(ctx, 'col', None)
        if col is None:
            col = ConvFn._col(x, k, ctx.lowp)
        else:
            ctx.col = None                       # (its memory becomes dcol below and is released with this call)
        dw = torch.empty_like(w)
        _gemm(dy, col, dw, 1, Cout, K, HW, HW, HW, K, 0, 0, 0, trans_b=1, kbatch=B,
              a_kbs=Cout*HW, b_kbs=K*HW, lowp=ctx.lowp)
        dx = None
        if ctx.needs_input_grad[0]:
            dcol = col if k != 1 else _empty(B, K, HW, like=x)
            _gemm(w, dy, dcol, B, K, HW, Cout, K, HW, HW, 0, Cout*HW, K*HW, trans_a=1,
                  lowp=ctx.lowp)
            if k == 1:
                dx = dcol.view(B, Cin, H, W)
            else:
                dx = torch.empty_like(x)
                fn, name = (hip.lib().brv_col2im_bf16, 'brv_col2im_bf16') if dcol.dtype == torch.bfloat16 else \
                    (hip.lib().brv_col2im, 'brv_col2im')
                hip.check(fn(hip.ptr(dcol), None, hip.ptr(dx), B, Cin, H, W, k, k, 1, 1, k//2, k//2, H, W,
                             hip.stream()), name)
        return dx, dw, db


class GroupNormFn(torch.autograd.Function):
    """act(GroupNorm(x + add[:, :, None, None])), act = SiLU or identity."""

    @staticmethod
    def forward(ctx, x, add, gamma, beta, groups, eps, silu):
        x = x.contiguous()
        B, C, H, W = x.shape
        lib = hip.lib()
        scratch = torch.zeros(lib.brv_groupnorm_scratch_bytes(B, groups), dtype=torch.uint8,
                              device=x.device)
        scale, shift, mu, rstd = (_empty(B, C, like=x) for _ in range(4))
        addc = add.contiguous() if add is not None else None
        hip.check(lib.brv_groupnorm_fold(
            hip.ptr(x), hip.ptr(addc), hip.ptr(gamma), hip.ptr(beta), None, None, hip.ptr(scratch),
            hip.ptr(scale), hip.ptr(shift), hip.ptr(mu), hip.ptr(rstd), B, C, H*W, groups,
            float(eps), hip.stream()), 'brv_groupnorm_fold')
        y = torch.empty_like(x)
        hip.check(lib.brv_affine_act(hip.ptr(x), hip.ptr(scale), hip.ptr(shift), hip.ptr(y), B, C,
                                     H*W, int(silu), hip.stream()), 'brv_affine_act')
        ctx.save_for_backward(x, gamma, scale, shift, mu, rstd)
        ctx.cfg = (groups, bool(silu), add is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, scale, shift, mu, rstd = ctx.saved_tensors
        groups, silu, has_add = ctx.cfg
        dy = dy.contiguous()
        B, C, H, W = x.shape
        dx = torch.empty_like(x)
        s1, s2, dadd = (_empty(B, C, like=x) for _ in range(3))
        coef = _empty(3*B*C, like=x)
        hip.check(hip.lib().brv_groupnorm_backward(
            hip.ptr(x), hip.ptr(dy), hip.ptr(scale), hip.ptr(shift), hip.ptr(mu), hip.ptr(rstd),
            hip.ptr(gamma), hip.ptr(dx), hip.ptr(s1), hip.ptr(s2), hip.ptr(dadd), hip.ptr(coef), B,
            C, H*W, groups, int(silu), hip.stream()), 'brv_groupnorm_backward')
        # d gamma / d beta: (B, C) -> (C,), a handful of values
        return dx, (dadd if has_add else None), s2.sum(0), s1.sum(0), None, None, None


class AffineActFn(torch.autograd.Function):
    """act(scale[:, :, None, None]*x + shift[:, :, None, None]) with per-(item, channel) scale and
    shift that carry gradients (the ADM modulation (1 + m)*norm + s followed by SiLU)."""

    @staticmethod
    def forward(ctx, x, scale, shift, silu):
        x, scale, shift = x.contiguous(), scale.contiguous(), shift.contiguous()
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        hip.check(hip.lib().brv_affine_act(hip.ptr(x), hip.ptr(scale), hip.ptr(shift), hip.ptr(y), B,
                                           C, H*W, int(silu), hip.stream()), 'brv_affine_act')
        ctx.save_for_backward(x, scale, shift)
        ctx.silu = bool(silu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, scale, shift = ctx.saved_tensors
        dy = dy.contiguous()
        B, C, H, W = x.shape
        dx = torch.empty_like(x)
        dscale, dshift = torch.empty_like(scale), torch.empty_like(shift)
        zeros, ones = torch.zeros_like(scale), torch.ones_like(scale)
        hip.check(hip.lib().brv_affine_act_backward(
            hip.ptr(x), hip.ptr(dy), hip.ptr(scale), hip.ptr(shift), hip.ptr(zeros), hip.ptr(ones),
            hip.ptr(dx), hip.ptr(dscale), hip.ptr(dshift), B, C, H*W, int(ctx.silu), hip.stream()),
            'brv_affine_act_backward')
        return dx, dscale, dshift, None


class SiluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        hip.check(hip.lib().brv_silu(hip.ptr(x), hip.ptr(y), x.numel(), hip.stream()), 'brv_silu')
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        hip.check(hip.lib().brv_silu_backward(hip.ptr(x), hip.ptr(dy), hip.ptr(dx), x.numel(),
                                              hip.stream()), 'brv_silu_backward')
        return dx


def _axpby_raw(a, alpha, b, beta):
    out = torch.empty_like(a)
    hip.check(hip.lib().brv_axpby(hip.ptr(a), float(alpha), hip.ptr(b), float(beta), hip.ptr(out),
                                  a.numel(), hip.stream()), 'brv_axpby')
    return out


class AxpbyFn(torch.autograd.Function):
    """alpha*a + beta*b (b may be None)."""

    @staticmethod
    def forward(ctx, a, alpha, b, beta):
        ctx.coef = (float(alpha), float(beta), b is not None)
        return _axpby_raw(a.contiguous(), alpha, b.contiguous() if b is not None else None, beta)

    @staticmethod
    def backward(ctx, g):
        alpha, beta, has_b = ctx.coef
        g = g.contiguous()
        return (_axpby_raw(g, alpha, None, 0.0), None,
                _axpby_raw(g, beta, None, 0.0) if has_b else None, None)


class ResampleFn(torch.autograd.Function):
    """Resample.forward with an explicit plan (padding, output size); the backward pass is the
    adjoint FIR operator (down <-> up) with the same padding."""

    @staticmethod
    def forward(ctx, x, kernel, up, pad, out_hw):
        x = x.contiguous()
        B, C, H, W = x.shape
        K = kernel.shape[-1]
        y = _empty(B, C, out_hw[0], out_hw[1], like=x)
        hip.check(hip.lib().brv_fir_resample2d(
            hip.ptr(x), hip.ptr(kernel), hip.ptr(y), B*C, H, W, out_hw[0], out_hw[1], K, pad[0],
            pad[1], int(up), 4.0 if up else 1.0, hip.stream()), 'brv_fir_resample2d')
        ctx.save_for_backward(kernel)
        ctx.cfg = (bool(up), pad, (H, W))
        return y

    @staticmethod
    def backward(ctx, dy):
        kernel, = ctx.saved_tensors
        up, pad, (H, W) = ctx.cfg
        dy = dy.contiguous()
        B, C, Ho, Wo = dy.shape
        K = kernel.shape[-1]
        dx = _empty(B, C, H, W, like=dy)
        if up:       # adjoint of the transposed convolution with 4*kernel: strided FIR with 4*kernel
            k4 = (4.0*kernel).contiguous()
            hip.check(hip.lib().brv_fir_resample2d(
                hip.ptr(dy), hip.ptr(k4), hip.ptr(dx), B*C, Ho, Wo, H, W, K, pad[0], pad[1], 0, 1.0,
                hip.stream()), 'brv_fir_resample2d')
        else:        # adjoint of the strided FIR: transposed convolution onto the input grid
            hip.check(hip.lib().brv_fir_resample2d(
                hip.ptr(dy), hip.ptr(kernel), hip.ptr(dx), B*C, Ho, Wo, H, W, K, pad[0], pad[1], 1,
                1.0, hip.stream()), 'brv_fir_resample2d')
        return dx, None, None, None, None


class LinearFn(torch.autograd.Function):
    """(N, I) -> (N, O) = x @ W^T + b."""

    @staticmethod
    def forward(ctx, x, w, bias):
        x = x.contiguous()
        N, I = x.shape
        O = w.shape[0]
        yt = _empty(O, N, like=x)
        _gemm(w, x, yt, 1, O, N, I, I, I, N, 0, 0, 0, trans_b=1, bias=bias)
        ctx.save_for_backward(x, w)
        return yt.t().contiguous()

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        N, I = x.shape
        O = w.shape[0]
        dx = _empty(N, I, like=x)
        _gemm(dy, w, dx, 1, N, I, O, O, I, I, 0, 0, 0)
        dw = torch.empty_like(w)
        _gemm(dy, x, dw, 1, O, I, N, O, I, I, 0, 0, 0, trans_a=1)
        return dx, dw, dy.sum(0)


class AttentionCoreFn(torch.autograd.Function):
    """softmax(q^T k / sqrt(C)) applied to v, on (N, C, L) maps (net.py:432-443)."""

    @staticmethod
    def forward(ctx, q, k, v):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        N, C, L = q.shape
        lib = hip.lib()
        w = _empty(N, L, L, like=q)
        _gemm(q, k, w, N, L, L, C, L, L, L, C*L, C*L, L*L, trans_a=1)
        w = _axpby_raw(w, 1.0/C**0.5, None, 0.0)
        p = torch.empty_like(w)
        hip.check(lib.brv_softmax_rows(hip.ptr(w), hip.ptr(p), N*L, L, hip.stream()),
                  'brv_softmax_rows')
        a = _empty(N, C, L, like=q)
        _gemm(v, p, a, N, C, L, L, L, L, L, C*L, L*L, C*L, trans_b=1)
        ctx.save_for_backward(q, k, v, p)
        return a

    @staticmethod
    def backward(ctx, da):
        q, k, v, p = ctx.saved_tensors
        da = da.contiguous()
        N, C, L = q.shape
        dv = torch.empty_like(v)                      # da (C x L) @ P (L x L)
        _gemm(da, p, dv, N, C, L, L, L, L, L, C*L, L*L, C*L)
        dp = torch.empty_like(p)                      # da^T (L x C) @ v (C x L)
        _gemm(da, v, dp, N, L, L, C, L, L, L, C*L, C*L, L*L, trans_a=1)
        dw = torch.empty_like(p)
        hip.check(hip.lib().brv_softmax_rows_backward(hip.ptr(p), hip.ptr(dp), hip.ptr(dw), N*L, L,
                                                      hip.stream()), 'brv_softmax_rows_backward')
        dw = _axpby_raw(dw, 1.0/C**0.5, None, 0.0)
        dq = torch.empty_like(q)                      # k (C x L) @ dW^T
        _gemm(k, dw, dq, N, C, L, L, L, L, L, C*L, L*L, C*L, trans_b=1)
        dk = torch.empty_like(k)                      # q (C x L) @ dW
        _gemm(q, dw, dk, N, C, L, L, L, L, L, C*L, L*L, C*L)
        return dq, dk, dv


# ---------------------------------------------------------------------------------------------
# the network, block by block (same order of operations as sgmse.py / net.py)
# ---------------------------------------------------------------------------------------------
def conv(x, mod):
    return ConvFn.apply(x, mod.weight, mod.bias)


def group_norm(x, mod, add=None, silu=False):
    return GroupNormFn.apply(x, add, mod.weight, mod.bias, mod.num_groups, mod.eps, silu)


def resample(resampler, x, direction):
    padding, out_hw, up = resampler.plan(x.shape, direction)
    return ResampleFn.apply(x, resampler.kernel.float().contiguous(), up, padding, out_hw)


def attention(blk, x, out_scale):
    N, C, H, W = x.shape
    xn = group_norm(x, blk.norm)
    q, k, v = (conv(xn, m).view(N, C, H*W) for m in (blk.conv_query, blk.conv_key, blk.conv_value))
    a = AttentionCoreFn.apply(q, k, v).view(N, C, H, W)
    return AxpbyFn.apply(x, out_scale, conv(a, blk.conv_out), out_scale)


class DropoutFn(torch.autograd.Function):
    """nn.Dropout (UNetBlock.dropout, net.py:409): the keep mask is drawn on the device
    generator, mask and 1/keep are applied by ``brv_dropout_apply`` in both directions."""

    @staticmethod
    def forward(ctx, x, p):
        keep = 1.0 - p
        mask = torch.empty_like(x).bernoulli_(keep)
        out = torch.empty_like(x)
        hip.check(hip.lib().brv_dropout_apply(hip.ptr(x.contiguous()), hip.ptr(mask), hip.ptr(out),
                                              x.numel(), 1.0/keep, hip.stream()), 'brv_dropout_apply')
        ctx.save_for_backward(mask)
        ctx.scale = 1.0/keep
        return out

    @staticmethod
    def backward(ctx, dy):
        mask, = ctx.saved_tensors
        dx = torch.empty_like(mask)
        hip.check(hip.lib().brv_dropout_apply(hip.ptr(dy.contiguous()), hip.ptr(mask), hip.ptr(dx),
                                              mask.numel(), ctx.scale, hip.stream()),
                  'brv_dropout_apply')
        return dx, None


def unet_block(blk, x, emb):
    h = group_norm(x, blk.norm_1, silu=True)
    if blk.resampler is not None:
        h = resample(blk.resampler, h, blk.up_or_down)
        x = resample(blk.resampler, x, blk.up_or_down)
    h = conv(h, blk.conv_1)
    e = LinearFn.apply(emb, blk.linear.weight, blk.linear.bias)
    if e.shape[0] != h.shape[0]:
        e = e.expand(h.shape[0], -1)
    if blk.block_type == 'adm':       # silu((scale + 1)*norm(h) + shift), net.py:405-407
        m, sh = e.chunk(2, dim=1)
        h = AffineActFn.apply(group_norm(h, blk.norm_2), AxpbyFn.apply(m, 1.0, torch.ones_like(m), 1.0),
                              sh, True)
    else:
        h = group_norm(h, blk.norm_2, add=e, silu=True)
    if blk.dropout.p > 0 and blk.training:
        h = DropoutFn.apply(h, blk.dropout.p)
    h = conv(h, blk.conv_2)
    if blk.skip_conv is not None:
        x = conv(x, blk.skip_conv)
    x = AxpbyFn.apply(x, blk.skip_scale, h, blk.skip_scale)
    if blk.attn is not None:
        x = attention(blk.attn, x, blk.skip_scale)
    return x


def unet(net, x, sigma):
    """DiffusionUNet.forward (net.py:232-262) with gradients."""
    from .sgmse import AuxiliaryDown, AuxiliaryUp  # noqa: F401
    _col_kept[0] = 0                             # column-matrix budget of this evaluation (ConvFn.forward)
    emb = net.emb.fourier_proj(sigma.reshape(-1))
    emb = SiluFn.apply(LinearFn.apply(emb, net.emb.linear_1.weight, net.emb.linear_1.bias))
    emb = SiluFn.apply(LinearFn.apply(emb, net.emb.linear_2.weight, net.emb.linear_2.bias))
    aux = x
    x = conv(x, net.input_conv)
    skips = [x]
    for enc, aux_block in zip(net.encoder, net.aux_downs):
        n = len(enc.unet_blocks)
        for i, blk in enumerate(enc.unet_blocks):
            x = unet_block(blk, x, emb)
            if i != n - 1:
                skips.append(x)
        if aux_block is not None:
            aux = resample(aux_block.resampler, aux, 'down')
            x = AxpbyFn.apply(x, 1.0, conv(aux, aux_block.conv), 1.0)
            if aux_block.type_ == 'residual':
                aux = x = AxpbyFn.apply(x, aux_block.skip_scale, None, 0.0)
        skips.append(x)
    x = unet_block(net.bottleneck_block_1, x, emb)
    x = unet_block(net.bottleneck_block_2, x, emb)
    aux = None
    for dec, aux_block in zip(net.decoder, net.aux_ups):
        for blk in dec.unet_blocks:
            if blk.resampler is None:
                x = torch.cat([x, skips.pop()], dim=1)
            x = unet_block(blk, x, emb)
        if aux_block is not None:
            if aux_block.resampler is not None:
                aux = resample(aux_block.resampler, aux, 'up')
            if aux_block.type_ == 'skip' or aux_block.resampler is None:
                h = conv(group_norm(x, aux_block.norm, silu=True), aux_block.conv)
                aux = h if aux is None else AxpbyFn.apply(aux, 1.0, h, 1.0)
            else:
                x = aux = AxpbyFn.apply(x, 1.0, conv(aux, aux_block.conv), 1.0)
    if aux is None:
        aux = x
    if isinstance(net.output_conv, torch.nn.Conv2d):
        return conv(aux, net.output_conv)
    return conv(group_norm(aux, net.output_conv[0]), net.output_conv[1])
