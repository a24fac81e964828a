"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's TF-GridNet
(brever/models/tfgridnet/tfgridnet.py:28-415) as one function of a flat ``{name: tensor}``
parameter table (the reference's ``named_parameters`` keys). Pinned by tests/golden/tfgridnet.npz
(outputs, multiresyu loss and every parameter gradient of two narrow configurations, generated
from the imported reference by tests/golden/make_golden.py). Nothing on the product path imports
this file.

Written channels-last, (B, T, Q, C), with explicit LSTM recurrences, so that it shares no
structure with either the reference (module tree on (B, C, T, Q)) or the HIP path beyond the
mathematics."""
import math

import torch
import torch.nn.functional as F

import scipy.signal

from . import criterion as oc


def _lstm_direction(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of nn.LSTM (gate order i, f, g, o; zero initial state): x (N, S, I)."""
    N, S, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(N, H)
    c = x.new_zeros(N, H)
    pre = x @ w_ih.t() + b_ih + b_hh
    out = [None]*S
    for s in (range(S - 1, -1, -1) if reverse else range(S)):
        i, f, g, o = (pre[:, s] + h @ w_hh.t()).chunk(4, dim=1)
        c = torch.sigmoid(f)*c + torch.sigmoid(i)*torch.tanh(g)
        h = torch.sigmoid(o)*torch.tanh(c)
        out[s] = h
    return torch.stack(out, dim=1)


def _bilstm(x, P, prefix):
    fwd = _lstm_direction(x, P[prefix + 'weight_ih_l0'], P[prefix + 'weight_hh_l0'],
                          P[prefix + 'bias_ih_l0'], P[prefix + 'bias_hh_l0'], False)
    bwd = _lstm_direction(x, P[prefix + 'weight_ih_l0_reverse'], P[prefix + 'weight_hh_l0_reverse'],
                          P[prefix + 'bias_ih_l0_reverse'], P[prefix + 'bias_hh_l0_reverse'], True)
    return torch.cat([fwd, bwd], dim=-1)


def _moments_norm(x, dims, eps):
    mu = x.mean(dim=dims, keepdim=True)
    var = ((x - mu)**2).mean(dim=dims, keepdim=True)
    return (x - mu)/torch.sqrt(var + eps)


def _head_norm(x, P, prefix, H, eps):
    """tfgridnet.py:383-415 on channels-last x (B, T, Q, H*E) -> (B, H, T, E, Q)."""
    B, T, Q, HE = x.shape
    E = HE//H
    x = x.view(B, T, Q, H, E).permute(0, 3, 1, 4, 2)                 # (B, H, T, E, Q)
    slope = P[prefix + 'act.weight'].view(1, H, 1, 1, 1)
    x = torch.where(x > 0, x, slope*x)
    x = _moments_norm(x, (3, 4), eps)
    gamma = P[prefix + 'gamma'][0].permute(0, 2, 1, 3)               # (H, E, 1, Q) -> (H, 1, E, Q)
    beta = P[prefix + 'beta'][0].permute(0, 2, 1, 3)
    return x*gamma + beta


def _grid_rnn(x, P, prefix, ks, eps, hs=None):
    hs = ks if hs is None else hs
    """Layer norm over channels, groups of ``ks`` neighbours along axis 2 as one LSTM step,
    bidirectional LSTM, linear back to ks*C, residual (tfgridnet.py:268-292 with ks == hs)."""
    B, A, S, C = x.shape
    h = F.layer_norm(x, (C,), P[prefix + 'norm.weight'], P[prefix + 'norm.bias'], eps)
    if ks == hs:
        h = _bilstm(h.reshape(B*A, S//ks, ks*C), P, prefix + 'rnn.')
        h = h @ P[prefix + 'linear.weight'].t() + P[prefix + 'linear.bias']
        return x + h.reshape(B, A, S, C)
    # overlapping windows (tfgridnet.py:275-291): every window of ks neighbours (stride hs) is
    # one LSTM step with features ordered channel-major; the ConvTranspose1d spreads each step's
    # output back over its window and overlapping contributions add
    win = h.unfold(2, ks, hs)                                   # (B, A, n, C, ks)
    n = win.shape[2]
    r = _bilstm(win.reshape(B*A, n, C*ks), P, prefix + 'rnn.')  # (BA, n, 2H)
    w = P[prefix + 'linear.weight']                             # (2H, C, ks)
    spread = torch.einsum('bnh,hci->bnci', r, w)
    out = x.new_zeros(B*A, S, C) + P[prefix + 'linear.bias']
    for i in range(ks):
        idx = torch.arange(n)*hs + i
        out = out.index_add(1, idx, spread[..., i])
    return x + out.reshape(B, A, S, C)


def _block(x, P, prefix, cfg):
    """One GridNetV2Block (tfgridnet.py:255-353) on (B, T, Q, C)."""
    ks, hs, H, eps = cfg['emb_ks'], cfg['emb_hs'], cfg['attn_n_head'], cfg['eps']
    B, T0, Q0, C = x.shape
    olp = ks - hs
    T = math.ceil((T0 + 2*olp - ks)/hs)*hs + ks
    Q = math.ceil((Q0 + 2*olp - ks)/hs)*hs + ks
    x = F.pad(x, (0, 0, olp, Q - Q0 - olp, olp, T - T0 - olp))
    x = _grid_rnn(x, P, prefix + 'intra_', ks, eps, hs)                                 # along bands
    x = _grid_rnn(x.transpose(1, 2), P, prefix + 'inter_', ks, eps, hs).transpose(1, 2)   # along frames
    x = x[:, olp:olp + T0, olp:olp + Q0]

    def proj(name):
        w = P[prefix + name + '.weight'].flatten(1)
        return x @ w.t() + P[prefix + name + '.bias']
    q = _head_norm(proj('attn_conv_Q'), P, prefix + 'attn_norm_Q.', H, eps)           # (B,H,T,E,Q)
    k = _head_norm(proj('attn_conv_K'), P, prefix + 'attn_norm_K.', H, eps)
    v = _head_norm(proj('attn_conv_V'), P, prefix + 'attn_norm_V.', H, eps)
    D = q.shape[3]*q.shape[4]
    score = torch.einsum('bhteq,bhseq->bhts', q, k)/D**0.5
    a = torch.einsum('bhts,bhseq->bhteq', torch.softmax(score, dim=-1), v)            # (B,H,T,Ev,Q)
    a = a.permute(0, 2, 4, 1, 3).reshape(B, T0, Q0, -1)                               # (B,T,Q,H*Ev)
    a = a @ P[prefix + 'attn_concat_proj.0.weight'].flatten(1).t() + P[prefix + 'attn_concat_proj.0.bias']
    if prefix + 'attn_concat_proj.1.weight' in P:
        a = torch.where(a > 0, a, P[prefix + 'attn_concat_proj.1.weight']*a)
    a = _moments_norm(a, (2, 3), eps)
    gamma = P[prefix + 'attn_concat_proj.2.gamma'][0].permute(1, 2, 0)                # (1, Q, C)
    beta = P[prefix + 'attn_concat_proj.2.beta'][0].permute(1, 2, 0)
    return a*gamma + beta + x


DEFAULTS = dict(n_srcs=1, n_fft=256, stride=128, n_layers=6, lstm_hidden_units=128, attn_n_head=4,
                attn_approx_qk_dim=512, emb_dim=32, emb_ks=4, emb_hs=4, eps=1e-5)


def forward(P, cfg, x):
    """x (B, 2, L) -> (B, n_srcs, L) (tfgridnet.py:105-130)."""
    cfg = dict(DEFAULTS, **cfg)
    B, M, L = x.shape
    n, hop, S = cfg['n_fft'], cfg['stride'], cfg['n_srcs']
    std = x.reshape(B, -1).std(dim=1).view(B, 1, 1)
    window = torch.from_numpy(scipy.signal.get_window('hann', n)).to(x.dtype)
    frames = math.ceil(max(L - n, 0)/hop) + 1                              # modules/stft.py:140-149
    xp = F.pad(x/std, (0, (frames - 1)*hop + n - L))
    spec = torch.stft(xp.reshape(B*M, -1), n_fft=n, hop_length=hop, window=window, center=True,
                      pad_mode='constant', normalized=False, onesided=True, return_complex=True)
    spec = spec.view(B, M, *spec.shape[1:])                                # (B, M, F, T)
    grid = torch.cat([spec.real, spec.imag], dim=1).transpose(2, 3)        # (B, 2M, T, F)
    grid = F.conv2d(grid, P['conv.0.weight'], P['conv.0.bias'], padding=1)
    grid = F.group_norm(grid, 1, P['conv.1.weight'], P['conv.1.bias'], cfg['eps'])
    grid = grid.permute(0, 2, 3, 1)                                        # (B, T, F, C)
    for i in range(cfg['n_layers']):
        grid = _block(grid, P, f'blocks.{i}.', cfg)
    grid = F.conv_transpose2d(grid.permute(0, 3, 1, 2), P['deconv.weight'], P['deconv.bias'],
                              padding=1)                                   # (B, 2S, T, F)
    grid = grid.reshape(B, S, 2, *grid.shape[2:])
    spec = torch.complex(grid[:, :, 0], grid[:, :, 1]).transpose(2, 3)     # (B, S, F, T)
    y = torch.istft(spec.reshape(B*S, *spec.shape[2:]), n_fft=n, hop_length=hop, window=window,
                    center=True, normalized=False, onesided=True)
    return y.view(B, S, -1)[..., :L]*std


def loss(P, cfg, batch, lengths):
    """tfgridnet.py:132-146 with the default criterion (multiresyu, criterion.py:135-226)."""
    mix, target = batch[:, 0], batch[:, 1:].mean(dim=-2)
    return oc.multiresyu(forward(P, cfg, mix), target, lengths).mean()


def parameter_shapes(cfg):
    """{name: shape} in the reference's ``named_parameters`` order (tfgridnet.py:76-101,193-248)."""
    cfg = dict(DEFAULTS, **cfg)
    C, ks, Hh, H = cfg['emb_dim'], cfg['emb_ks'], cfg['lstm_hidden_units'], cfg['attn_n_head']
    Q = cfg['n_fft']//2 + 1
    E = math.ceil(cfg['attn_approx_qk_dim']/Q)
    sh = {'conv.0.weight': (C, 4, 3, 3), 'conv.0.bias': (C,), 'conv.1.weight': (C,), 'conv.1.bias': (C,)}
    for i in range(cfg['n_layers']):
        p = f'blocks.{i}.'
        for part in ('intra_', 'inter_'):
            sh[p + part + 'norm.weight'] = (C,); sh[p + part + 'norm.bias'] = (C,)
            for d in ('', '_reverse'):
                sh[p + part + 'rnn.weight_ih_l0' + d] = (4*Hh, ks*C)
                sh[p + part + 'rnn.weight_hh_l0' + d] = (4*Hh, Hh)
                sh[p + part + 'rnn.bias_ih_l0' + d] = (4*Hh,)
                sh[p + part + 'rnn.bias_hh_l0' + d] = (4*Hh,)
            if ks == cfg['emb_hs']:
                sh[p + part + 'linear.weight'] = (ks*C, 2*Hh); sh[p + part + 'linear.bias'] = (ks*C,)
            else:
                sh[p + part + 'linear.weight'] = (2*Hh, C, ks); sh[p + part + 'linear.bias'] = (C,)
        for name, e in (('Q', E), ('K', E), ('V', C//H)):
            sh[p + f'attn_conv_{name}.weight'] = (H*e, C, 1, 1); sh[p + f'attn_conv_{name}.bias'] = (H*e,)
            sh[p + f'attn_norm_{name}.gamma'] = (1, H, e, 1, Q); sh[p + f'attn_norm_{name}.beta'] = (1, H, e, 1, Q)
            sh[p + f'attn_norm_{name}.act.weight'] = (H,)
        sh[p + 'attn_concat_proj.0.weight'] = (C, C, 1, 1); sh[p + 'attn_concat_proj.0.bias'] = (C,)
        sh[p + 'attn_concat_proj.1.weight'] = (1,)
        sh[p + 'attn_concat_proj.2.gamma'] = (1, C, 1, Q); sh[p + 'attn_concat_proj.2.beta'] = (1, C, 1, Q)
    sh['deconv.weight'] = (C, 2*cfg['n_srcs'], 3, 3); sh['deconv.bias'] = (2*cfg['n_srcs'],)
    return sh
