"""TF-GridNet on the HIP path (reference: brever/models/tfgridnet/tfgridnet.py:28-415).

The ``nn`` modules below only hold parameters (same names, shapes, construction order and
therefore the same default initialisation and ``state_dict`` as the reference); every arithmetic
operation runs in ``libbrever_hip.so`` through ``torch.autograd.Function`` pairs:

* RMS normalisation by the unbiased standard deviation and its reversal: ``brv_row_std`` /
  ``brv_row_scale`` (tfgridnet.py:108-109,128);
* STFT / inverse STFT: the differentiable DFT-GEMM kernels of ``modules/stft.py``;
* the 3x3 input convolution and the output transposed convolution (= a convolution with the
  flipped, transposed kernel): column matrix + MFMA product (``sgmse_train.ConvFn``);
  ``nn.GroupNorm(1, C)``: ``brv_groupnorm_*``;
* the grid blocks work channels-last, (B, T, Q, C): ``nn.LayerNorm(C)``,
  ``LayerNormalization4DCF`` and ``AllHeadPReLULayerNormalization4DCF`` are one row-norm operator
  (``brv_rownorm_*``) on rows laid out so that the normalised axes are contiguous; the
  bidirectional LSTMs run both directions of all sequences concurrently in the register-resident
  recurrence kernels (``dccrn._LSTMFunction``, two groups: the sequence and its reversal); the
  linear layers and 1x1 convolutions are MFMA products over the channel axis; the attention is
  two batched products around ``brv_softmax_rows``.

torch only pads, slices, permutes, concatenates and carries the autograd graph.
"""
import math

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip
from ..modules.stft import STFT
from . import sgmse_train as T
from .base import BreverBaseModel, ModelRegistry
from .dccrn import _AMP, _LSTMFunction, _ParamOnly


class LayerNormalization4DCF(_ParamOnly):
    """Parameters of tfgridnet.py:356-380: gamma / beta (1, C, 1, F)."""

    def __init__(self, input_dimension, eps=1e-5):
        super().__init__()
        assert len(input_dimension) == 2
        size = [1, input_dimension[0], 1, input_dimension[1]]
        self.gamma = nn.Parameter(torch.ones(*size))
        self.beta = nn.Parameter(torch.zeros(*size))
        self.eps = eps


class AllHeadPReLULayerNormalization4DCF(_ParamOnly):
    """Parameters of tfgridnet.py:383-415: gamma / beta (1, H, E, 1, F), one PReLU slope per
    head."""

    def __init__(self, input_dimension, eps=1e-5):
        super().__init__()
        assert len(input_dimension) == 3
        H, E, n_freqs = input_dimension
        self.gamma = nn.Parameter(torch.ones(1, H, E, 1, n_freqs))
        self.beta = nn.Parameter(torch.zeros(1, H, E, 1, n_freqs))
        self.act = nn.PReLU(num_parameters=H, init=0.25)
        self.eps, self.H, self.E, self.n_freqs = eps, H, E, n_freqs


class GridNetV2Block(_ParamOnly):
    """Parameters of one grid block (tfgridnet.py:176-253)."""

    def __init__(self, emb_dim, emb_ks, emb_hs, n_freqs, hidden_channels, n_head=4,
                 approx_qk_dim=512, activation='PReLU', eps=1e-5):
        super().__init__()
        in_channels = emb_dim*emb_ks
        self.intra_norm = nn.LayerNorm(emb_dim, eps=eps)
        self.intra_rnn = nn.LSTM(in_channels, hidden_channels, 1, batch_first=True,
                                 bidirectional=True)
        if emb_ks == emb_hs:
            self.intra_linear = nn.Linear(hidden_channels*2, in_channels)
        else:
            self.intra_linear = nn.ConvTranspose1d(hidden_channels*2, emb_dim, emb_ks, stride=emb_hs)
        self.inter_norm = nn.LayerNorm(emb_dim, eps=eps)
        self.inter_rnn = nn.LSTM(in_channels, hidden_channels, 1, batch_first=True,
                                 bidirectional=True)
        if emb_ks == emb_hs:
            self.inter_linear = nn.Linear(hidden_channels*2, in_channels)
        else:
            self.inter_linear = nn.ConvTranspose1d(hidden_channels*2, emb_dim, emb_ks, stride=emb_hs)
        E = math.ceil(approx_qk_dim*1.0/n_freqs)
        assert emb_dim % n_head == 0
        self.attn_conv_Q = nn.Conv2d(emb_dim, n_head*E, 1)
        self.attn_norm_Q = AllHeadPReLULayerNormalization4DCF((n_head, E, n_freqs), eps=eps)
        self.attn_conv_K = nn.Conv2d(emb_dim, n_head*E, 1)
        self.attn_norm_K = AllHeadPReLULayerNormalization4DCF((n_head, E, n_freqs), eps=eps)
        self.attn_conv_V = nn.Conv2d(emb_dim, n_head*emb_dim//n_head, 1)
        self.attn_norm_V = AllHeadPReLULayerNormalization4DCF((n_head, emb_dim//n_head, n_freqs),
                                                              eps=eps)
        self.attn_concat_proj = nn.Sequential(
            nn.Conv2d(emb_dim, emb_dim, 1),
            getattr(nn, activation)(),
            LayerNormalization4DCF((emb_dim, n_freqs), eps=eps),
        )
        self.emb_dim, self.emb_ks, self.emb_hs, self.n_head = emb_dim, emb_ks, emb_hs, n_head


# ---------------------------------------------------------------------------------------------
class _RowNormFn(torch.autograd.Function):
    """y = (prelu(x) - mean_row) * rstd_row * gain[g] + bias[g] on rows (R, n); the group of row r
    is (r // inner) % G; ``slope`` (G,) or None."""

    @staticmethod
    def forward(ctx, x, slope, gain, bias, inner, eps):
        x, gain, bias = x.contiguous(), gain.contiguous(), bias.contiguous()
        R, n = x.shape
        G = gain.shape[0]
        y = torch.empty_like(x)
        stats = torch.empty(R, 2, dtype=torch.float32, device=x.device)
        sl = slope.contiguous() if slope is not None else None
        hip.check(hip.lib().brv_rownorm_forward(hip.ptr(x), hip.ptr(sl), hip.ptr(gain), hip.ptr(bias),
                                                hip.ptr(y), hip.ptr(stats), R, n, inner, G,
                                                float(eps), hip.stream()), 'brv_rownorm_forward')
        ctx.save_for_backward(x, sl, gain, stats)
        ctx.cfg = (inner, G, slope is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, sl, gain, stats = ctx.saved_tensors
        inner, G, has_slope = ctx.cfg
        lib = hip.lib()
        R, n = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dgain, dbias = torch.empty_like(gain), torch.empty_like(gain)
        rows = torch.empty(R, dtype=torch.float32, device=x.device) if has_slope else None
        scratch = torch.empty(lib.brv_rownorm_scratch_bytes(n, G), dtype=torch.uint8,
                              device=x.device)
        hip.check(lib.brv_rownorm_backward(
            hip.ptr(x), hip.ptr(dy), hip.ptr(sl), hip.ptr(gain), hip.ptr(stats), hip.ptr(dx),
            hip.ptr(dgain), hip.ptr(dbias), hip.ptr(rows), hip.ptr(scratch), R, n, inner, G,
            hip.stream()), 'brv_rownorm_backward')
        dslope = None
        if has_slope:                                   # rows are (outer, G, inner)
            dslope = torch.empty(G, dtype=torch.float32, device=x.device)
            hip.check(lib.brv_row_sum(hip.ptr(rows), hip.ptr(dslope), R//(G*inner), G, inner,
                                      hip.stream()), 'brv_row_sum')
        return dx, dslope, dgain, dbias, None, None


class _RowScaleFn(torch.autograd.Function):
    """y[r] = x[r] * s[r] (``divide`` False) or x[r] / s[r] on (R, n); gradient wrt x only."""

    @staticmethod
    def forward(ctx, x, s, divide):
        x, s = x.contiguous(), s.contiguous()
        y = torch.empty_like(x)
        hip.check(hip.lib().brv_row_scale(hip.ptr(x), hip.ptr(s), hip.ptr(y), x.shape[0], x.shape[1],
                                          int(divide), hip.stream()), 'brv_row_scale')
        ctx.save_for_backward(s)
        ctx.divide = divide
        return y

    @staticmethod
    def backward(ctx, dy):
        s, = ctx.saved_tensors
        return _RowScaleFn.apply(dy, s, ctx.divide), None, None


class _AttentionFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(D)) v on q, k (N, L, D), v (N, L, Dv) (tfgridnet.py:325-327)."""

    @staticmethod
    def forward(ctx, q, k, v):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        N, L, D = q.shape
        Dv = v.shape[-1]
        lowp = ctx.lowp = _AMP['on']
        w = T._empty(N, L, L, like=q)
        T._gemm(q, k, w, N, L, L, D, D, D, L, L*D, L*D, L*L, trans_b=1, lowp=lowp)
        w = T._axpby_raw(w, 1.0/D**0.5, None, 0.0)
        p = torch.empty_like(w)
        hip.check(hip.lib().brv_softmax_rows(hip.ptr(w), hip.ptr(p), N*L, L, hip.stream()),
                  'brv_softmax_rows')
        a = T._empty(N, L, Dv, like=q)
        T._gemm(p, v, a, N, L, Dv, L, L, Dv, Dv, L*L, L*Dv, L*Dv, lowp=lowp)
        ctx.save_for_backward(q, k, v, p)
        return a

    @staticmethod
    def backward(ctx, da):
        q, k, v, p = ctx.saved_tensors
        da = da.contiguous()
        N, L, D = q.shape
        Dv = v.shape[-1]
        lowp = ctx.lowp
        dv = torch.empty_like(v)                       # P^T (L x L) @ da (L x Dv)
        T._gemm(p, da, dv, N, L, Dv, L, L, Dv, Dv, L*L, L*Dv, L*Dv, trans_a=1, lowp=lowp)
        dp = torch.empty_like(p)                       # da (L x Dv) @ v^T
        T._gemm(da, v, dp, N, L, L, Dv, Dv, Dv, L, L*Dv, L*Dv, L*L, trans_b=1, lowp=lowp)
        dw = torch.empty_like(p)
        hip.check(hip.lib().brv_softmax_rows_backward(hip.ptr(p), hip.ptr(dp), hip.ptr(dw), N*L, L,
                                                      hip.stream()), 'brv_softmax_rows_backward')
        dw = T._axpby_raw(dw, 1.0/D**0.5, None, 0.0)
        dq = torch.empty_like(q)                       # dW (L x L) @ k (L x D)
        T._gemm(dw, k, dq, N, L, D, L, L, D, D, L*L, L*D, L*D, lowp=lowp)
        dk = torch.empty_like(k)                       # dW^T @ q
        T._gemm(dw, q, dk, N, L, D, L, L, D, D, L*L, L*D, L*D, trans_a=1, lowp=lowp)
        return dq, dk, dv


_SMALL = os.environ.get('BRV_TFG_LINEAR_SMALL', '1') != '0'     # narrow linear layers on brv_linear_small
_SMALL_SCRATCH = {}
# NOTE: the two narrow-layer paths below are taken BEFORE ``lowp`` is looked at, so under use_amp the
# layers with K, N <= 64 (and their weight gradients) run in full fp32 while every other product rounds its operands
# to bf16: closer to the fp32 oracle than the bf16 emulation assumes, never further (the use_amp tolerances of
# tests/test_gpu.py are upper bounds on the distance from the fp32 reference). The cut-over is a size rule of the
# library (`brv_linear_small_supported`), so results change at that size by bf16-rounding level, not more.


def _gemm(a, b, d, batch, M, N, K, lda, ldb, ldd, a_bs=0, b_bs=0, d_bs=0, trans_a=0, trans_b=0,
          kbatch=1, a_kbs=0, b_kbs=0, bias=None, mode=0, lowp=False):
    """brv_gemm_f32 / brv_gemm_bf16 (bf16 operands, fp32 accumulation: ``use_amp``); ``mode`` 1
    adds to d, 2 reads ``bias`` per output column; a bf16 ``d`` tensor is written directly."""
    half = torch.bfloat16
    flags = int(b.dtype == half) | int(d.dtype == half) << 1 | int(a.dtype == half) << 2
    if _SMALL and not flags and batch == 1 and kbatch == 1 and not trans_a and (bias is None or mode == 2) \
            and lda % 4 == 0 and ldd % 4 == 0 and (a.data_ptr() | d.data_ptr()) % 16 == 0 \
            and hip.lib().brv_linear_small_supported(M, N, K):
        # narrow layers (K, N <= 64 over ~2.6e5 rows): one thread per row in fp32 instead of a 128 x 128 MFMA tile
        hip.check(hip.lib().brv_linear_small(hip.ptr(a), hip.ptr(b), hip.ptr(bias), hip.ptr(d), M, N, K, lda, ldb,
                                             ldd, trans_b, int(mode == 1), hip.stream()), 'brv_linear_small')
        return
    if _SMALL and not flags and batch == 1 and kbatch == 1 and trans_a and not trans_b and bias is None and mode == 0 \
            and lda % 4 == 0 and ldb % 4 == 0 and (a.data_ptr() | b.data_ptr()) % 16 == 0 \
            and hip.lib().brv_linear_small_wgrad_supported(K, M, N):
        # their weight gradients: (rows x M)^T (rows x N) over ~2.6e5 rows, slices added in a fixed order
        lib = hip.lib()
        # (one scratch buffer per device and size, not an allocation per call: up to 48 calls per step -- ADVICE r4; the
        # launches of a stream run in order, so consecutive products may share it)
        nbytes = lib.brv_linear_small_wgrad_scratch_bytes(M, N)
        key = (d.device.index, torch.cuda.current_stream(d.device).cuda_stream)
        scratch = _SMALL_SCRATCH.get(key)
        if scratch is None or scratch.numel() < nbytes:
            scratch = _SMALL_SCRATCH[key] = torch.empty(nbytes, dtype=torch.uint8, device=d.device)
        hip.check(lib.brv_linear_small_wgrad(hip.ptr(a), hip.ptr(b), hip.ptr(d), hip.ptr(scratch), K, M, N, lda, ldb,
                                             ldd, hip.stream()), 'brv_linear_small_wgrad')
        return
    if flags:
        hip.check(hip.lib().brv_gemm_bf16_mixed(
            hip.ptr(a), hip.ptr(b), hip.ptr(d), batch, M, N, K, lda, ldb, ldd, a_bs, b_bs, d_bs,
            trans_a, trans_b, kbatch, a_kbs, b_kbs, hip.ptr(bias), mode, flags, hip.stream()),
            'brv_gemm_bf16_mixed')
        return
    if not lowp:
        hip.gemm_f32(a, b, d, batch, M, N, K, lda, ldb, ldd, a_bs, b_bs, d_bs, trans_a, trans_b, kbatch, a_kbs,
                     b_kbs, bias, mode)
        return
    hip.check(hip.lib().brv_gemm_bf16(
        hip.ptr(a), hip.ptr(b), hip.ptr(d), batch, M, N, K, lda, ldb, ldd, a_bs, b_bs, d_bs,
        trans_a, trans_b, kbatch, a_kbs, b_kbs, hip.ptr(bias), mode, hip.stream()), 'brv_gemm_bf16')


def _column_sums(x, rows, cols, batch=1, lowp=False):
    """(batch, rows, cols) -> (batch, cols). fp32 tensors: ``brv_col_sum`` (fixed order, 16-byte loads: 17 us for
    258 516 x 32 -- also under ``lowp``, where round 3 used a ones-vector product with split atomics); bf16
    tensors: a ones-vector product on the bf16 MFMA."""
    out = torch.empty(batch, cols, dtype=torch.float32, device=x.device)
    if x.dtype == torch.float32:
        lib = hip.lib()
        scratch = torch.empty(lib.brv_col_sum_scratch_bytes(batch, cols), dtype=torch.uint8,
                              device=x.device)
        hip.check(lib.brv_col_sum(hip.ptr(x), hip.ptr(out), hip.ptr(scratch), batch, rows, cols,
                                  hip.stream()), 'brv_col_sum')
        return out
    if x.dtype == torch.bfloat16 and cols % 8 == 0 and x.data_ptr() % 16 == 0:
        lib = hip.lib()
        scratch = torch.empty(lib.brv_col_sum_scratch_bytes(batch, cols), dtype=torch.uint8,
                              device=x.device)
        hip.check(lib.brv_col_sum_bf16(hip.ptr(x), hip.ptr(out), hip.ptr(scratch), batch, rows, cols,
                                       hip.stream()), 'brv_col_sum_bf16')
        return out
    ones = torch.ones(rows, dtype=torch.float32, device=x.device)
    _gemm(ones, x, out, batch, 1, cols, rows, rows, cols, cols, 0, rows*cols, cols, lowp=True)
    return out


class _LinearFn(torch.autograd.Function):
    """(N, I) -> (N, O) = x @ W^T + b, row-major in and out (no transposed copies)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        x, w = x.contiguous(), w.contiguous()
        N, I = x.shape
        O = w.shape[0]
        lowp = ctx.lowp = _AMP['on']
        y = T._empty(N, O, like=x)
        _gemm(x, w, y, 1, N, O, I, I, I, O, trans_b=1, bias=bias.contiguous(), mode=2, lowp=lowp)
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        N, I = x.shape
        O = w.shape[0]
        dx = T._empty(N, I, like=x)
        _gemm(dy, w, dx, 1, N, I, O, O, I, I, lowp=ctx.lowp)
        dw = torch.empty_like(w)
        _gemm(dy, x, dw, 1, O, I, N, O, I, I, trans_a=1, lowp=ctx.lowp)
        return dx, dw, _column_sums(dy, N, O, lowp=ctx.lowp)[0]


def _linear(x, mod):
    """nn.Linear / 1x1 convolution over the last axis of a channels-last tensor."""
    w = mod.weight.reshape(mod.weight.shape[0], -1)
    return _LinearFn.apply(x.reshape(-1, x.shape[-1]), w, mod.bias).view(*x.shape[:-1], w.shape[0])


class _BiLSTMFn(torch.autograd.Function):
    """Bidirectional single-layer nn.LSTM (batch_first, zero initial state) on x (N, S, I) ->
    (N, S, 2H) with the tiled recurrence kernels: both directions are groups of ONE launch, the
    second walking the frames backwards in place (no flipped copies), both writing their half of
    the output rows. Input projections, their gradients and the bias gradients are batched
    products; the recurrent weight gradient pairs every gate gradient with the previous hidden
    state of its own direction through shifted views (no shifted copy)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh):
        lib = hip.lib()
        x, w_hh = x.contiguous(), w_hh.contiguous()
        N, S, I = x.shape
        H = w_hh.shape[-1]
        lowp = ctx.lowp = _AMP['on']
        w_ih = _LSTMFunction._interleave(w_ih.contiguous(), H)
        # use_amp: the input projection and the saved gate activations are bf16 tensors (the
        # recurrence kernel is HBM-bound on exactly these two streams)
        io = torch.bfloat16 if lowp else torch.float32
        gates = torch.empty(2, N, S, 4*H, dtype=io, device=x.device)
        _gemm(x, w_ih, gates, 2, N*S, 4*H, I, I, I, 4*H, 0, 4*H*I, N*S*4*H, trans_b=1, lowp=lowp)
        bias = T._axpby_raw(b_ih.detach().contiguous(), 1.0, b_hh.detach().contiguous(), 1.0)
        y = T._empty(N, S, 2*H, like=x)
        act, cs = torch.empty(2, N, S, 4*H, dtype=io, device=x.device), T._empty(2, N, S, H, like=x)
        hip.check(lib.brv_lstm_tile_forward(hip.ptr(gates), hip.ptr(w_hh), hip.ptr(bias), hip.ptr(y),
                                            hip.ptr(act), hip.ptr(cs), 2*N, S, H, 2, 2, 2*H, H,
                                            2*int(lowp), hip.stream()), 'brv_lstm_tile_forward')
        ctx.save_for_backward(x, w_ih, w_hh, y, act, cs)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = hip.lib()
        x, w_ih, w_hh, y, act, cs = ctx.saved_tensors            # w_ih interleaved
        N, S, I = x.shape
        H = w_hh.shape[-1]
        lowp, NS = ctx.lowp, N*S
        dy = dy.contiguous()
        dg = torch.empty(2, N, S, 4*H, dtype=act.dtype, device=x.device)     # bf16 under use_amp
        hip.check(lib.brv_lstm_tile_backward(hip.ptr(act), hip.ptr(cs), hip.ptr(w_hh), hip.ptr(dy),
                                             hip.ptr(dg), 2*N, S, H, 2, 2, 2*H, H, 2*int(lowp),
                                             hip.stream()),
                  'brv_lstm_tile_backward')
        dx = torch.empty_like(x)                      # sum over both directions: dg_g @ W_ih_g
        _gemm(dg, w_ih, dx, 1, NS, I, 4*H, 4*H, I, I, kbatch=2, a_kbs=NS*4*H, b_kbs=4*H*I,
              lowp=lowp)
        dw_ih = torch.empty_like(w_ih)                # dg_g^T @ x
        _gemm(dg, x, dw_ih, 2, 4*H, I, NS, 4*H, I, I, NS*4*H, 0, 4*H*I, trans_a=1, lowp=lowp)
        dw_hh = torch.zeros_like(w_hh) if S == 1 else torch.empty_like(w_hh)
        if S > 1:
            # forward direction: gate gradients of frames 1.. against hidden states of frames 0..;
            # backward direction: frames ..S-2 against hidden states of frames 1.. -- ONE launch: the directions are the
            # batch (from the forward direction's operands the backward one's lie NS 4H - 4H / 3H elements further)
            _gemm(dg.view(-1)[4*H:], y.view(-1), dw_hh, 2, 4*H, H, S - 1, 4*H, 2*H, H, NS*4*H - 4*H, 3*H, 4*H*H,
                  trans_a=1, kbatch=N, a_kbs=S*4*H, b_kbs=S*2*H, lowp=lowp)
        db = _column_sums(dg, NS, 4*H, batch=2)
        dw_ih, dw_hh, db = (_LSTMFunction._deinterleave(t, H) for t in (dw_ih, dw_hh, db))
        return dx, dw_ih, dw_hh, db, db.clone()


class _WindowFn(torch.autograd.Function):
    """F.unfold of (N, C, S) with windows of ``ks`` every ``hs`` along S -> (N, C*ks, n)
    (``fold`` False: brv_im2col, gradient brv_col2im) or its adjoint, the overlap-add of
    (N, C*ks, n) windows onto (N, C, S) plus a per-channel bias (``fold`` True: what a
    ConvTranspose1d does after its channel product)."""

    @staticmethod
    def _geom(C, S, ks, hs):
        return (C, S, 1, ks, 1, hs, 1, 0, 0, (S - ks)//hs + 1, 1)

    @staticmethod
    def _unfold(x, C, S, ks, hs):
        N = x.shape[0]
        n = (S - ks)//hs + 1
        col = T._empty(N, C*ks, n, like=x)
        hip.check(hip.lib().brv_im2col(hip.ptr(x), hip.ptr(col), N, *_WindowFn._geom(C, S, ks, hs),
                                       hip.stream()), 'brv_im2col')
        return col

    @staticmethod
    def _fold(col, bias, C, S, ks, hs):
        N = col.shape[0]
        out = T._empty(N, C, S, like=col)
        hip.check(hip.lib().brv_col2im(hip.ptr(col), hip.ptr(bias), hip.ptr(out), N,
                                       *_WindowFn._geom(C, S, ks, hs), hip.stream()), 'brv_col2im')
        return out

    @staticmethod
    def forward(ctx, x, bias, C, S, ks, hs, fold):
        ctx.cfg = (C, S, ks, hs, fold, bias is not None)
        x = x.contiguous()
        if fold:
            return _WindowFn._fold(x, bias.contiguous() if bias is not None else None, C, S, ks, hs)
        return _WindowFn._unfold(x, C, S, ks, hs)

    @staticmethod
    def backward(ctx, g):
        C, S, ks, hs, fold, has_bias = ctx.cfg
        g = g.contiguous()
        if not fold:
            return _WindowFn._fold(g, None, C, S, ks, hs), None, None, None, None, None, None
        db = None
        if has_bias:
            db = T._empty(C, like=g)
            hip.check(hip.lib().brv_row_sum(hip.ptr(g), hip.ptr(db), g.shape[0], C, S, hip.stream()),
                      'brv_row_sum')
        return _WindowFn._unfold(g, C, S, ks, hs), db, None, None, None, None, None


def _add(a, b):
    return T.AxpbyFn.apply(a, 1.0, b, 1.0)


_HEAD_PERMUTE = os.environ.get('BRV_TFG_HEAD_PERMUTE', '1') != '0'     # 0: torch permute + copy (round 4)


class _HeadPermuteFn(torch.autograd.Function):
    """``merge`` False: x (B, T, F, H*E) channels-last -> (B, H, T, E, F); True: x (B, H, T, E, F) -> (B, T, F, H*E).
    ``brv_head_permute`` (one launch through LDS) where a torch permute + copy ran a five-dimensional strided copy;
    the gradient is the opposite permutation."""

    @staticmethod
    def _run(x, B, Tn, Fq, H, E, merge):
        x = x.contiguous()
        out = torch.empty((B, Tn, Fq, H*E) if merge else (B, H, Tn, E, Fq), dtype=torch.float32, device=x.device)
        hip.check(hip.lib().brv_head_permute(hip.ptr(x), hip.ptr(out), B, Tn, Fq, H, E, int(merge), hip.stream()),
                  'brv_head_permute')
        return out

    @staticmethod
    def forward(ctx, x, H, E, merge):
        if merge:
            B, _, Tn, _, Fq = x.shape
        else:
            B, Tn, Fq, _ = x.shape
        ctx.cfg = (B, Tn, Fq, H, E, merge)
        return _HeadPermuteFn._run(x, B, Tn, Fq, H, E, merge)

    @staticmethod
    def backward(ctx, g):
        B, Tn, Fq, H, E, merge = ctx.cfg
        return _HeadPermuteFn._run(g, B, Tn, Fq, H, E, not merge), None, None, None


def _head_split(x, H, E):
    """(B, T, F, H*E) -> (B, H, T, E, F)."""
    B, Tn, Fq, _ = x.shape
    if _HEAD_PERMUTE and x.dtype == torch.float32 and hip.lib().brv_head_permute_supported(Fq, H, E):
        return _HeadPermuteFn.apply(x, H, E, False)
    return x.view(B, Tn, Fq, H, E).permute(0, 3, 1, 4, 2).contiguous()


def _head_merge(a, B, H, Tn, E, Fq):
    """(B*H, T, E*F) -> (B, T, F, H*E)."""
    a = a.view(B, H, Tn, E, Fq)
    if _HEAD_PERMUTE and a.dtype == torch.float32 and hip.lib().brv_head_permute_supported(Fq, H, E):
        return _HeadPermuteFn.apply(a, H, E, True)
    return a.permute(0, 2, 4, 1, 3).reshape(B, Tn, Fq, H*E)


def _bilstm(x, rnn):
    """Bidirectional single-layer nn.LSTM (batch_first) on (N, S, I) -> (N, S, 2H): the sequence
    and its reversal are two groups of one launch set."""
    stack = lambda name: torch.stack([getattr(rnn, name + '_l0'),  # noqa: E731
                                      getattr(rnn, name + '_l0_reverse')])
    if hip.lib().brv_lstm_tile_supported(rnn.hidden_size):
        return _BiLSTMFn.apply(x, stack('weight_ih'), stack('weight_hh'), stack('bias_ih'),
                               stack('bias_hh'))
    y = _LSTMFunction.apply(torch.stack([x, x.flip(1)]), stack('weight_ih'), stack('weight_hh'),
                            stack('bias_ih'), stack('bias_hh'))
    return torch.cat([y[0], y[1].flip(1)], dim=-1)


@ModelRegistry.register('tfgridnet')
class TFGridNet(BreverBaseModel):
    _fused_adam = True       # clip + Adam as brv_clip_adam_step2 on one flat buffer (models/base.py)

    def __init__(
        self,
        n_srcs: int = 1,
        n_fft: int = 256,
        stride: int = 128,
        window: str = 'hann',
        n_layers: int = 6,
        lstm_hidden_units: int = 128,
        attn_n_head: int = 4,
        attn_approx_qk_dim: int = 512,
        emb_dim: int = 32,
        emb_ks: int = 4,
        emb_hs: int = 4,
        activation: str = 'PReLU',
        eps: float = 1e-5,
        criterion: str = 'multiresyu',
        optimizer: str = 'Adam',
        learning_rate: float = 0.001,
        grad_clip: float = 1.0,
    ):
        super().__init__(criterion=criterion)
        self.n_srcs = n_srcs
        self.n_layers = n_layers
        n_freqs = n_fft//2 + 1
        self.stft = STFT(frame_length=n_fft, hop_length=stride, window=window, normalized=False)
        n_imics = 2
        ks, padding = (3, 3), (1, 1)
        self.conv = nn.Sequential(
            nn.Conv2d(2*n_imics, emb_dim, ks, padding=padding),
            nn.GroupNorm(1, emb_dim, eps=eps),
        )
        self.blocks = nn.ModuleList([
            GridNetV2Block(emb_dim, emb_ks, emb_hs, n_freqs, lstm_hidden_units, n_head=attn_n_head,
                           approx_qk_dim=attn_approx_qk_dim, activation=activation, eps=eps)
            for _ in range(n_layers)])
        self.deconv = nn.ConvTranspose2d(emb_dim, n_srcs*2, ks, padding=padding)
        self.optimizer = self.init_optimizer(optimizer, lr=learning_rate)
        self.grad_clip = grad_clip
        self.scheduler = self.init_lr_scheduler()

    # ---- the grid block, channels-last -------------------------------------------------------
    @staticmethod
    def _layer_norm(x, norm):
        C = x.shape[-1]
        y = _RowNormFn.apply(x.reshape(-1, C), None, norm.weight.view(1, C), norm.bias.view(1, C), 1,
                             norm.eps)
        return y.view(x.shape)

    @staticmethod
    def _head_norm(x, norm):
        """AllHeadPReLULayerNormalization4DCF on channels-last x (B, T, F, H*E) -> (B*H, T, E*F),
        which is the (items, frames, features) layout the attention products read."""
        B, Tn, Fq, _ = x.shape
        H, E = norm.H, norm.E
        rows = _head_split(x, H, E).view(B*H*Tn, E*Fq)
        y = _RowNormFn.apply(rows, norm.act.weight, norm.gamma.view(H, E*Fq), norm.beta.view(H, E*Fq),
                             Tn, norm.eps)
        return y.view(B*H, Tn, E*Fq)

    def _grid_rnn(self, x, norm, rnn, linear, ks, hs):
        """x (B, A, S, C): layer norm over C, windows of ``ks`` every ``hs`` along S as LSTM
        steps, bidirectional LSTM, back to (B, A, S, C) (tfgridnet.py:268-313)."""
        B, A, S, C = x.shape
        h = self._layer_norm(x, norm)
        if ks == hs:                                   # disjoint windows: plain views
            h = _bilstm(h.view(B*A, S//ks, ks*C), rnn)
            return _linear(h, linear).view(B, A, S, C)
        # overlapping windows: unfold (feature = channel*ks + offset), transposed convolution =
        # channel product + overlap-add
        hc = h.view(B*A, S, C).transpose(1, 2)                               # (BA, C, S)
        col = _WindowFn.apply(hc, None, C, S, ks, hs, False)                 # (BA, C*ks, n)
        h = _bilstm(col.transpose(1, 2).contiguous(), rnn)                   # (BA, n, 2H)
        w = linear.weight.reshape(linear.weight.shape[0], C*ks).t()          # (C*ks, 2H)
        col = _LinearFn.apply(h.reshape(-1, h.shape[-1]), w, torch.zeros(C*ks, device=x.device))
        col = col.view(B*A, -1, C*ks).transpose(1, 2)                        # (BA, C*ks, n)
        out = _WindowFn.apply(col, linear.bias, C, S, ks, hs, True)          # (BA, C, S)
        return out.transpose(1, 2).reshape(B, A, S, C)

    def _block(self, blk, x):
        """x (B, T, Q, C) channels-last -> same shape (tfgridnet.py:255-353)."""
        B, old_T, old_Q, C = x.shape
        ks, hs = blk.emb_ks, blk.emb_hs
        olp = ks - hs
        Tp = math.ceil((old_T + 2*olp - ks)/hs)*hs + ks
        Qp = math.ceil((old_Q + 2*olp - ks)/hs)*hs + ks
        x = F.pad(x, (0, 0, olp, Qp - old_Q - olp, olp, Tp - old_T - olp))
        # intra-frame (full-band) recurrence along the bands
        x = _add(self._grid_rnn(x, blk.intra_norm, blk.intra_rnn, blk.intra_linear, ks, hs), x)
        # sub-band recurrence along the frames
        x = x.transpose(1, 2).contiguous()                                   # (B, Q, T, C)
        x = _add(self._grid_rnn(x, blk.inter_norm, blk.inter_rnn, blk.inter_linear, ks, hs), x)
        x = x.transpose(1, 2)[:, olp:olp + old_T, olp:olp + old_Q].contiguous()   # (B, T, Q, C)
        # full-band self-attention across the frames
        q = self._head_norm(_linear(x, blk.attn_conv_Q), blk.attn_norm_Q)
        k = self._head_norm(_linear(x, blk.attn_conv_K), blk.attn_norm_K)
        v = self._head_norm(_linear(x, blk.attn_conv_V), blk.attn_norm_V)
        a = _AttentionFn.apply(q, k, v)                                      # (B*H, T, Ev*Q)
        H, Ev = blk.n_head, blk.attn_norm_V.E
        a = _head_merge(a, B, H, old_T, Ev, old_Q)                           # (B, T, Q, H*Ev)
        conv, act, norm = blk.attn_concat_proj
        a = _linear(a, conv)                                                 # (B, T, Q, C)
        slope = act.weight if isinstance(act, nn.PReLU) else None
        if slope is None and not isinstance(act, nn.Identity):
            raise NotImplementedError(f'activation {type(act).__name__} is not built on the HIP '
                                      'path (PReLU and Identity are)')
        if slope is None or slope.numel() == 1:
            # the normalisation runs over the whole (channel, band) plane of a frame: the order of the plane inside a
            # row does not matter to its statistics, so the rows stay in the (band, channel) order the projection wrote
            # and the (tiny) gain / bias planes are transposed instead -- round 5: two 33 MB activation copies per block
            # (the transpose in front and the one inside the residual add) and their two mirror images in backward
            rows = a.reshape(B*old_T, old_Q*C)
            gam = norm.gamma.view(C, old_Q).t().reshape(1, old_Q*C)
            bet = norm.beta.view(C, old_Q).t().reshape(1, old_Q*C)
            rows = _RowNormFn.apply(rows, slope, gam, bet, 1, norm.eps)
            return _add(rows.view(B, old_T, old_Q, C), x)
        rows = a.transpose(2, 3).reshape(B*old_T, C*old_Q)                   # (channel, band) rows
        rows = _RowNormFn.apply(rows, slope, norm.gamma.view(1, C*old_Q), norm.beta.view(1, C*old_Q),
                                1, norm.eps)
        a = rows.view(B, old_T, C, old_Q).transpose(2, 3)
        return _add(a, x)

    def forward(self, x):
        hip.require_device(x)
        B, M, L = x.shape
        with torch.no_grad():                          # the mixture carries no gradient
            x = x.float().contiguous()
            std = torch.empty(B, dtype=torch.float32, device=x.device)
            hip.check(hip.lib().brv_row_std(hip.ptr(x), hip.ptr(std), B, M*L, hip.stream()),
                      'brv_row_std')
            xn = _RowScaleFn.apply(x.view(B, M*L), std, True).view(B, M, L)
            spec = self.stft(xn).transpose(2, 3)                              # (B, M, T, F)
            batch = torch.cat((spec.real, spec.imag), dim=1).contiguous()     # (B, 2M, T, F)
        n_frames, n_freqs = batch.shape[2:]
        conv, norm = self.conv
        batch = T.GroupNormFn.apply(T.ConvFn.apply(batch, conv.weight, conv.bias), None, norm.weight,
                                    norm.bias, 1, norm.eps, False)
        batch = batch.permute(0, 2, 3, 1).contiguous()                        # (B, T, F, C)
        for blk in self.blocks:
            batch = self._block(blk, batch)
        batch = batch.permute(0, 3, 1, 2)                                      # (B, C, T, F)
        # transposed convolution, stride 1 = convolution with the flipped, transposed kernel
        w = self.deconv.weight.transpose(0, 1).flip(2, 3)
        batch = T.ConvFn.apply(batch, w.contiguous(), self.deconv.bias)       # (B, 2S, T, F)
        batch = batch.view(B, self.n_srcs, 2, n_frames, n_freqs)
        spec = torch.complex(batch[:, :, 0], batch[:, :, 1]).transpose(2, 3)  # (B, S, F, T)
        y = self.stft.backward(spec)[..., :L]
        S, Ly = y.shape[1:]
        return _RowScaleFn.apply(y.reshape(B, S*Ly), std, False).view(B, S, Ly)

    def loss(self, batch, lengths, use_amp):
        inputs, labels = batch[:, 0], batch[:, 1:]
        # reference signal: the average of the left and right direct-path signals
        # (tfgridnet.py:135-139)
        labels = T._axpby_raw(labels[:, :, 0].contiguous(), 0.5, labels[:, :, 1].contiguous(), 0.5)
        _AMP['on'] = T.AMP['on'] = bool(use_amp)
        try:
            outputs = self(inputs)
        finally:
            _AMP['on'] = T.AMP['on'] = False
        return self.criterion(outputs, labels, lengths).mean()

    def _enhance(self, x, use_amp):
        _AMP['on'] = T.AMP['on'] = bool(use_amp)
        try:
            with torch.no_grad():
                return self.forward(x)
        finally:
            _AMP['on'] = T.AMP['on'] = False

    def update(self, loss, scaler):
        super().update(loss, scaler, grad_clip=self.grad_clip)

    def on_validate(self, val_loss):
        self.scheduler.step(val_loss)

    def state_dict(self):
        return {'net': super().state_dict(), 'scheduler': self.scheduler.state_dict()}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict['net'])
        self.scheduler.load_state_dict(state_dict['scheduler'])

    def init_lr_scheduler(self):
        return torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode='min', factor=0.5,
                                                          patience=3)
