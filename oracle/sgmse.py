"""CPU restatement of the SGMSE+ inference path. TEST INFRASTRUCTURE ONLY (tests/, smoke()):
the product path (brever_amd/models/sgmse.py) never imports this file.

Follows brever/models/sgmse/net.py:232-477 (U-Net forward), preconditioning.py:40-58,
sdes.py:11-81 (prior, reverse step, probability flow), solvers.py:21-77 (EDM and
predictor-corrector samplers) and modules/resampling.py:27-61, as plain functions on a
state dict (name -> fp32 CPU tensor); the architecture is read off the parameter names.
Gaussian noise comes from the ``noise`` callable so that tests replay recorded draws.
Pinned by tests/golden/sgmse.npz (generated from the imported reference by
tests/golden/make_golden.py: denoiser output and a full ``enhance`` with recorded noise).
"""
import math

import torch
import torch.nn.functional as F


class Net:
    def __init__(self, state, prefix, skip_scale, block_type='ncsn', requires_grad=False):
        self.sd = {k[len(prefix):]: v.detach().float().cpu().clone() for k, v in state.items()
                   if k.startswith(prefix)}
        if requires_grad:                  # training oracle: leaves of the autograd graph
            for v in self.sd.values():
                if v.is_floating_point():
                    v.requires_grad_(True)
        self.skip_scale = skip_scale
        self.block_type = block_type
        self.paddings = []
        self.kernel = self.sd['resampler.kernel']

    def has(self, key):
        return key in self.sd

    def count(self, prefix):
        idx = {int(k[len(prefix):].split('.')[0]) for k in self.sd if k.startswith(prefix)}
        return max(idx) + 1 if idx else 0

    # -- layers ----------------------------------------------------------------------------
    def conv(self, x, name):
        w = self.sd[name + '.weight']
        return F.conv2d(x, w, self.sd[name + '.bias'], padding=w.shape[-1]//2)

    def norm(self, x, name):
        C = x.shape[1]
        return F.group_norm(x, min(32, C//4), self.sd[name + '.weight'], self.sd[name + '.bias'],
                            1e-6)

    def linear(self, x, name):
        return F.linear(x, self.sd[name + '.weight'], self.sd[name + '.bias'])

    def resample(self, x, direction):
        K = self.kernel.shape[-1]
        kern = self.kernel.tile([x.shape[1], 1, 1, 1])
        if direction == 'down':
            pad = tuple(math.ceil(K/2) - 1 if d % 2 == 0 else math.ceil((K + 1)/2) - 1
                        for d in x.shape[-2:])
            opad = tuple((d + 2*p - K) % 2 for d, p in zip(x.shape[-2:], pad))
            self.paddings.append((pad, opad))
            return F.conv2d(x, kern, padding=pad, groups=x.shape[1], stride=2)
        pad, opad = self.paddings.pop()
        return F.conv_transpose2d(x, kern*4, padding=pad, output_padding=opad,
                                  groups=x.shape[1], stride=2)

    def attention(self, x, p):
        N, C, H, W = x.shape
        xn = self.norm(x, p + '.norm')
        q = self.conv(xn, p + '.conv_query').reshape(N, C, H*W).transpose(1, 2)
        k = self.conv(xn, p + '.conv_key').reshape(N, C, H*W)
        v = self.conv(xn, p + '.conv_value').reshape(N, C, H*W).transpose(1, 2)
        w = torch.bmm(q, k/C**0.5).softmax(dim=-1)
        a = torch.bmm(w, v).transpose(1, 2).reshape(N, C, H, W)
        return x + self.conv(a, p + '.conv_out')

    def unet_block(self, x, emb, p, direction):
        resamples = self.has(p + '.resampler.kernel')
        h = F.silu(self.norm(x, p + '.norm_1'))
        if resamples:
            h = self.resample(h, direction)
            x = self.resample(x, direction)
        h = self.conv(h, p + '.conv_1')
        e = self.linear(emb, p + '.linear')[:, :, None, None]
        if self.block_type == 'adm':
            scale, shift = e.chunk(2, dim=1)
            h = (scale + 1)*self.norm(h, p + '.norm_2') + shift
        else:
            h = self.norm(h + e, p + '.norm_2')
        h = self.conv(F.silu(h), p + '.conv_2')
        if self.has(p + '.skip_conv.weight'):
            x = self.conv(x, p + '.skip_conv')
        x = self.skip_scale*(x + h)
        if self.has(p + '.attn.norm.weight'):
            x = self.skip_scale*self.attention(x, p + '.attn')
        return x

    # -- network ---------------------------------------------------------------------------
    def __call__(self, x, cnoise):
        b = self.sd['emb.fourier_proj.b']
        ang = 2*math.pi*cnoise.reshape(-1).float().outer(b)
        emb = torch.cat([ang.sin(), ang.cos()], dim=-1)
        emb = F.silu(self.linear(emb, 'emb.linear_1'))
        emb = F.silu(self.linear(emb, 'emb.linear_2'))
        aux = x
        x = self.conv(x, 'input_conv')
        skips = [x]
        num_res = self.count('encoder.')
        for i in range(num_res):
            nb = self.count(f'encoder.{i}.unet_blocks.')
            for j in range(nb):
                x = self.unet_block(x, emb, f'encoder.{i}.unet_blocks.{j}', 'down')
                if j != nb - 1:
                    skips.append(x)
            p = f'aux_downs.{i}'
            if self.has(p + '.conv.weight'):
                aux = self.resample(aux, 'down')
                x = x + self.conv(aux, p + '.conv')
                if self.sd[p + '.conv.weight'].shape[-1] == 3:       # 'residual' encoder
                    aux = x = x*self.skip_scale
            skips.append(x)
        x = self.unet_block(x, emb, 'bottleneck_block_1', 'none')
        x = self.unet_block(x, emb, 'bottleneck_block_2', 'none')
        aux = None
        skip_decoder = self.sd['output_conv.weight'].shape[-1] == 1 if self.has('output_conv.weight') \
            else False
        for i in range(num_res):
            for j in range(self.count(f'decoder.{i}.unet_blocks.')):
                p = f'decoder.{i}.unet_blocks.{j}'
                if not self.has(p + '.resampler.kernel'):
                    x = torch.cat([x, skips.pop()], dim=1)
                x = self.unet_block(x, emb, p, 'up')
            p = f'aux_ups.{i}'
            if self.has(p + '.conv.weight'):
                resamples = self.has(p + '.resampler.kernel')
                if resamples:
                    aux = self.resample(aux, 'up')
                if skip_decoder or not resamples:
                    h = self.conv(F.silu(self.norm(x, p + '.norm')), p + '.conv')
                    aux = h if aux is None else aux + h
                else:
                    x = aux = x + self.conv(aux, p + '.conv')
        if aux is None:
            aux = x
        if self.has('output_conv.weight'):
            return self.conv(aux, 'output_conv')
        return self.conv(self.norm(aux, 'output_conv.0'), 'output_conv.1')


def denoise(net, sde, x, y, sigma, t, precond='richter', sigma_data=0.1):
    """Preconditioned denoiser D(x; y, sigma, t) (preconditioning.py:40-55)."""
    scaling = sde.s(t)
    if precond == 'richter':
        cskip, cout, cin, shift, cnoise = 1, -scaling*sigma**2/t, scaling, y, t.log()
    else:
        cskip = sigma_data**2/(sigma**2 + sigma_data**2)
        cout = sigma*sigma_data/(sigma**2 + sigma_data**2)**0.5
        cin, shift, cnoise = 1/(sigma**2 + sigma_data**2)**0.5, 0, sigma.log()/4
    x_in = cin*x + shift
    out = net(torch.cat([x_in.real, x_in.imag, y.real, y.imag], dim=1), cnoise)
    return cskip*x + cout*torch.complex(out[:, 0], out[:, 1]).unsqueeze(1)


def score(net, sde, x, y, sigma, t, **kw):
    return (denoise(net, sde, x, y, sigma, t, **kw) - x)/(sde.s(t)*sigma**2)


def pc_sample(net, sde, y, noise, num_steps, corrector_steps, corrector_snr, on_step=None, **kw):
    """Predictor-corrector sampler (solvers.py:48-77); noise(shape, complex) -> tensor;
    ``on_step(i, x)``: the state after every reverse step (tests)."""
    dt = -1/num_steps
    t = torch.arange(1, 0, dt)
    sigma = sde.sigma(t)
    one = torch.tensor(1)
    x = y + sde.s(one)*sde.sigma(one)*noise(y.shape, True)
    eps = 2*(corrector_snr*sde.s(t)*sigma)**2
    for i in range(num_steps):
        for _ in range(corrector_steps):
            sc = score(net, sde, (x - y)/sde.s(t[i]), y, sigma[i], t[i], **kw)
            x = x + eps[i]*sc + (2*eps[i])**0.5*noise(x.shape, True)
        sc = score(net, sde, (x - y)/sde.s(t[i]), y, sigma[i], t[i], **kw)
        drift = sde.drift_coef(t[i])*(y - x)
        if i < num_steps - 1:
            x = x + (drift - sde.g(t[i])**2*sc)*dt + sde.g(t[i])*(-dt)**0.5*noise(x.shape, False)
        else:
            x = x + dt*(drift - 0.5*sde.g(t[i])**2*sc)
        if on_step is not None:
            on_step(i, x)
    return x


def edm_sample(net, sde, y, noise, num_steps, schurn, smin, smax, snoise, **kw):
    """Heun sampler of Karras et al. (solvers.py:18-45)."""
    gamma_max = min(schurn/num_steps, 2**0.5 - 1)
    t = torch.linspace(1, 0, num_steps + 1)
    sigma = sde.sigma(t)
    one = torch.tensor(1)
    x = y + sde.s(one)*sde.sigma(one)*noise(y.shape, True)

    def flow(xc, tc, sc_):
        sc = score(net, sde, (xc - y)/sde.s(tc), y, sc_, tc, **kw)
        return sde.drift_coef(tc)*(y - xc) - 0.5*sde.g(tc)**2*sc
    for i in range(num_steps):
        eps = snoise*noise(x.shape, True)
        gamma = gamma_max if smin <= sigma[i] <= smax else 0
        sigma_hat = sigma[i]*(1 + gamma)
        t_hat = sde.sigma_inv(sigma_hat)
        x_hat = sde.s(t_hat)/sde.s(t[i])*(x - y) + y \
            + sde.s(t_hat)*(sigma_hat**2 - sigma[i]**2)**0.5*eps
        d_hat = flow(x_hat, t_hat, sigma_hat)
        x = x_hat + (t[i + 1] - t_hat)*d_hat
        if i < num_steps - 1:
            d_next = flow(x, t[i + 1], sigma[i + 1])
            x = x_hat + 0.5*(t[i + 1] - t_hat)*(d_hat + d_next)
    return x


class RichterOUVE:
    """sdes.py:40-81 scalar schedules (host scalars)."""

    def __init__(self, stiffness=1.5, sigma_min=0.05, sigma_max=0.5):
        self.k, self.smin = stiffness, sigma_min
        self.p, self.logp = sigma_max/sigma_min, math.log(sigma_max/sigma_min)

    def s(self, t):
        return (-self.k*t).exp()

    def sigma(self, t):
        return self.smin*(((self.p**t/self.s(t))**2 - 1)/(1 + self.k/self.logp))**0.5

    def g(self, t):
        return self.smin*self.p**t*(2*self.logp)**0.5

    def sigma_inv(self, sigma):
        return 0.5*(1 + (1 + self.k/self.logp)*(sigma/self.smin)**2).log()/(self.k + self.logp)

    def drift_coef(self, t):
        return self.k


class OUCosine:
    """sdes.py:91-172 (VP base + shifted cosine schedule)."""

    def __init__(self, stiffness=1.5, lambda_min=-12.0, lambda_max=float('inf'), shift=3.0,
                 beta_clamp=10.0):
        self.k, self.shift, self.clampv = stiffness, shift, beta_clamp
        inv = lambda lam: 2/math.pi*math.atan(math.exp((shift - lam)/2))  # noqa: E731
        self.t_min, self.t_max = inv(lambda_min + shift), inv(lambda_max + shift)
        self.t_d = self.t_min - self.t_max

    def _a(self, t):
        return math.pi*(self.t_max + self.t_d*t)/2

    def beta(self, t):
        a = self._a(t)
        return (math.pi*self.t_d/a.cos()**2*a.tan()/(math.exp(self.shift) + a.tan()**2)) \
            .clamp(max=self.clampv)

    def sigma(self, t):
        return (-(-2*self._a(t).tan().log() + self.shift)/2).exp()

    def sigma_inv(self, sigma):
        lam = -2*sigma.log()
        return (2/math.pi*((self.shift - lam)/2).exp().atan() - self.t_max)/self.t_d

    def s(self, t):
        return (-self.k*t).exp()/(1 + self.sigma(t)**2)**0.5

    def g(self, t):
        return (-self.k*t).exp()*self.beta(t)**0.5

    def drift_coef(self, t):
        return self.k + 0.5*self.beta(t)


def enhance(net, sde, wav, window, hop_length, sampler, compression=0.5, scale=0.15):
    """SGMSEp._enhance (sgmse.py:182-199): mono mix-down, peak normalisation, compressed
    STFT without the Nyquist bin, reverse sampling, inverse STFT. ``sampler(net, sde, y)``
    is one of ``pc_sample`` / ``edm_sample`` with its options bound."""
    import numpy as np

    from . import stft as ostft
    length = wav.shape[-1]
    x = wav.mean(axis=-2, keepdims=True)
    norm = x.abs().amax(axis=-1, keepdims=True)
    spec = ostft.stft((x/norm).numpy(), window, hop_length, normalized=False,
                      compression=compression, scale=scale)
    y = torch.from_numpy(spec[..., :-1, :]).to(torch.complex64)
    out = sampler(net, sde, y)
    out = np.pad(out.numpy(), [(0, 0)]*(out.ndim - 2) + [(0, 1), (0, 0)])
    wave = ostft.istft(out, window, hop_length, normalized=False, compression=compression,
                       scale=scale)
    return (torch.from_numpy(wave).float()*norm)[..., :length].squeeze(1)


def train_loss(net, sde, batch, lengths, t, noise, precond='richter', sigma_data=0.1):
    """SGMSEp.loss (sgmse.py:163-176) with the draws of t and of the noise given: weighted,
    length-masked complex MSE (criterion.py:104-132) between the denoiser output and the
    clean-minus-noisy target."""
    y, x0 = batch[:, 0].unsqueeze(1), batch[:, 1].unsqueeze(1)
    sigma = sde.sigma(t)
    weight = 1/sigma**2 if precond == 'richter' \
        else (sigma**2 + sigma_data**2)/(sigma*sigma_data)**2
    d = denoise(net, sde, x0 - y + sigma*noise, y, sigma, t, precond=precond)
    frames = torch.arange(d.shape[-1])
    mask = (frames[None, :] < lengths[:, None])[:, None, None, :]
    err = ((d - (x0 - y))*mask).abs().pow(2).sum(-1)/lengths.view(-1, 1, 1)
    if weight is not None:
        err = err*weight.view(-1, 1, 1)
    return err.mean((1, 2)).mean()
