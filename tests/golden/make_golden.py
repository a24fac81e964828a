"""Generate the golden fixtures under tests/golden/ from the *imported reference*.

Runs ONLY in the build container, where the reference checkout is mounted
read-only at /root/reference; nothing of the reference (source or bytecode) is
copied -- the fixtures are inputs-by-seed and outputs. Third-party wheels the
reference imports but that are absent here (torchaudio, soundfile, h5py, wandb,
dotenv, torch_ema, pesq, pystoi, batch_pystoi, sofa) are replaced by empty
stand-in modules *before* import; none of them takes part in
the arithmetic recorded here.

    python tests/golden/make_golden.py [name ...]  # writes *.npz / *.json next to it
"""
import json
import os
import random
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    def missing(*a, **k):
        raise NotImplementedError('stubbed third-party function')

    ta = mod('torchaudio', info=missing, load=missing, save=missing)
    ta.functional = mod('torchaudio.functional', lfilter=missing)
    mod('soundfile', read=missing, write=missing)
    mod('wandb', log=missing, init=missing, login=missing)
    mod('dotenv', load_dotenv=missing)
    mod('h5py')
    mod('sofa')
    mod('pystoi', stoi=missing)
    mod('batch_pystoi', stoi=missing)
    pesq = mod('pesq', pesq=missing)
    pesq._pesq = mod('pesq._pesq', USAGE_BATCH='', _check_fs_mode=missing,
                     _pesq_inner=missing, _processor_mapping=missing)

    class PesqError:
        RAISE_EXCEPTION = 0
        RETURN_VALUES = 1
    pesq.cypesq = mod('pesq.cypesq', PesqError=PesqError)

    class ExponentialMovingAverage:      # functional stand-in; unused here
        def __init__(self, parameters, decay):
            self.params = list(parameters)
        def update(self): pass
        def store(self): pass
        def copy_to(self): pass
        def restore(self): pass
        def state_dict(self): return {}
        def load_state_dict(self, s): pass
    mod('torch_ema', ExponentialMovingAverage=ExponentialMovingAverage)


def lengths_fixture(n=100, lo=1600, hi=160000, seed=42):
    rng = random.Random(seed)
    return [rng.randint(lo, hi) for _ in range(n)]


class LengthOnlyDataset:
    """Implements just what the reference samplers touch."""

    def __init__(self, lengths):
        self._lengths = lengths
        self.rmm_dset = None

    def __len__(self):
        return len(self._lengths)

    def get_segment_length(self, i):
        return self._lengths[i]


def golden_batching(out):
    from brever.batching import (BatchSamplerRegistry,
                                 DistributedBatchSamplerWrapper)
    lengths = lengths_fixture()
    dset = LengthOnlyDataset(lengths)
    cases = []
    for name in ['random', 'sorted', 'bucket']:
        for dynamic, batch_size in [(False, 8), (True, 40.0)]:
            for kwargs in [dict(), dict(drop_last=True), dict(shuffle=False),
                           dict(seed=3)]:
                if name == 'sorted':
                    variants = [dict(kwargs), dict(kwargs, reverse=True)]
                else:
                    variants = [kwargs]
                for kw in variants:
                    sampler = BatchSamplerRegistry.get(name)(
                        dset, batch_size, dynamic=dynamic, **kw)
                    epochs = {}
                    for epoch in range(4):
                        sampler.set_epoch(epoch)
                        if kw.get('shuffle', True) or epoch == 0:
                            epochs[str(epoch)] = list(sampler)
                    cases.append(dict(name=name, dynamic=dynamic,
                                      batch_size=batch_size, kwargs=kw,
                                      epochs=epochs))
    # DDP wrapper, world sizes 2 and 4, fixed-length items (BASELINE config 3)
    fixed = LengthOnlyDataset([64000]*64)
    ddp = []
    for world in [2, 4]:
        for rank in range(world):
            sampler = BatchSamplerRegistry.get('bucket')(fixed, 64.0, dynamic=True)
            wrapper = DistributedBatchSamplerWrapper(sampler, num_replicas=world,
                                                     rank=rank)
            per_epoch = {}
            for epoch in range(3):
                wrapper.set_epoch(epoch)
                per_epoch[str(epoch)] = list(wrapper)
            ddp.append(dict(world=world, rank=rank, epochs=per_epoch))
    sampler = BatchSamplerRegistry.get('bucket')(fixed, 64.0, dynamic=True)
    single = {}
    for epoch in range(2):
        sampler.set_epoch(epoch)
        single[str(epoch)] = list(sampler)
    with open(os.path.join(out, 'batching.json'), 'w') as f:
        json.dump(dict(lengths=lengths, cases=cases, ddp=ddp, fixed_single=single,
                       seed0=sampler._seed), f)


def golden_collate(out):
    from brever.data import BreverDataLoader
    g = torch.Generator().manual_seed(7)
    items = [torch.randn(2, n, generator=g) for n in (5, 3, 4)]
    batch, lengths = BreverDataLoader._collate_fn(items)
    pairs = [(torch.randn(2, n, generator=g), torch.randn(1, generator=g))
             for n in (5, 3, 4)]
    pbatch, plengths = BreverDataLoader._collate_fn(pairs)
    np.savez(os.path.join(out, 'collate.npz'),
             **{f'item{i}': x.numpy() for i, x in enumerate(items)},
             batch=batch.numpy(), lengths=lengths.numpy(),
             **{f'pair{i}_0': a.numpy() for i, (a, b) in enumerate(pairs)},
             **{f'pair{i}_1': b.numpy() for i, (a, b) in enumerate(pairs)},
             pbatch0=pbatch[0].numpy(), pbatch1=pbatch[1].numpy(),
             plengths=plengths.numpy())


def golden_losses(out):
    from brever.criterion import CriterionRegistry
    torch.manual_seed(0)
    B, S, lo, hi = 6, 3, 500, 1200
    lengths = torch.randint(lo, hi, (B,))
    x = torch.randn(B, S, hi)
    y = torch.randn(B, S, hi) + 0.5*x
    res = dict(x=x.numpy(), y=y.numpy(), lengths=lengths.numpy())
    for name in ['snr', 'sisnr', 'mse']:
        res[name] = CriterionRegistry.get(name)(x, y, lengths).numpy()
    w = torch.rand(B)
    res['weight'] = w.numpy()
    res['mse_weighted'] = CriterionRegistry.get('mse')(x, y, lengths, weight=w).numpy()
    xg = x.clone().requires_grad_(True)
    CriterionRegistry.get('snr')(xg, y, lengths).mean().backward()
    res['snr_grad'] = xg.grad.numpy()
    gw = torch.rand(B)                      # non-uniform upstream gradient per item
    res['gweight'] = gw.numpy()
    # (the reference's sisnr divides the amax output in place, criterion.py:70, which
    # autograd of this torch version rejects: its gradient is checked against the
    # oracle's autograd instead, tests/test_gpu.py)
    xg = x.clone().requires_grad_(True)
    (CriterionRegistry.get('mse')(xg, y, lengths)*gw).sum().backward()
    res['mse_grad'] = xg.grad.numpy()
    xg = x.clone().requires_grad_(True)
    (CriterionRegistry.get('mse')(xg, y, lengths, weight=w)*gw).sum().backward()
    res['mse_weighted_grad'] = xg.grad.numpy()
    # MultiResYuLoss (criterion.py:135-226): default single resolution and three resolutions
    from brever.criterion import MultiResYuLoss
    for tag, kw in (('multiresyu', {}),
                    ('multiresyu3', dict(frame_lengths=[512, 256, 128], time_domain_weight=0.3,
                                         spectral_weight=0.7)),
                    ('multiresyu_si', dict(frame_lengths=[256, 128], scale_invariant=True))):
        crit = MultiResYuLoss(**kw)
        res[tag] = crit(x, y, lengths).numpy()
        xg = x.clone().requires_grad_(True)
        (crit(xg, y, lengths)*gw).sum().backward()
        res[tag + '_grad'] = xg.grad.numpy()
    np.savez_compressed(os.path.join(out, 'losses.npz'), **res)


SMALL = dict(filters=48, filter_length=16, bottleneck_channels=24,
             hidden_channels=40, skip_channels=16, kernel_size=3, layers=3,
             repeats=2, output_sources=1)
SMALL2 = dict(SMALL, output_sources=2, kernel_size=2)


def flat_state(model):
    return torch.cat([p.detach().reshape(-1) for p in model.parameters()])


def golden_convtasnet(out):
    from brever.models import ModelRegistry
    for tag, kw, B, L in [('small', SMALL, 3, 1000), ('small2', SMALL2, 2, 777),
                          ('causal', dict(SMALL, causal=True), 3, 1000),
                          ('causal2', dict(SMALL2, causal=True), 2, 777)]:
        torch.manual_seed(0)
        model = ModelRegistry.get('convtasnet')(**kw)
        # de-trivialise the affine / PReLU parameters (they initialise to 1/0/.25)
        g = torch.Generator().manual_seed(1)
        with torch.no_grad():
            for name, p in model.named_parameters():
                if 'norm' in name or 'prelu' in name:
                    p.add_(0.1*torch.randn(p.shape, generator=g))
        S = kw['output_sources']
        batch = 0.3*torch.randn(B, 1 + S, L, generator=g)
        lengths = torch.tensor([L, L - 137, L - 400][:B])
        for b in range(B):
            batch[b, :, lengths[b]:] = 0
        out_t = model(batch[:, 0])
        loss = model.loss(batch, lengths, use_amp=False)
        loss.backward()
        grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
        np.savez_compressed(
            os.path.join(out, f'convtasnet_{tag}.npz'),
            params=flat_state(model).numpy(), batch=batch.numpy(),
            lengths=lengths.numpy(), output=out_t.detach().numpy(),
            loss=loss.detach().numpy(), grads=grads.numpy(),
            config=json.dumps(kw))
    # default architecture: seeded init (reproducible by construction order),
    # short input; outputs + per-tensor gradient norms only
    torch.manual_seed(0)
    model = ModelRegistry.get('convtasnet')()
    g = torch.Generator().manual_seed(2)
    batch = 0.3*torch.randn(1, 2, 4000, generator=g)
    lengths = torch.tensor([4000])
    out_t = model(batch[:, 0])
    loss = model.loss(batch, lengths, use_amp=False)
    loss.backward()
    gnorm = torch.stack([p.grad.norm() for p in model.parameters()])
    flat = flat_state(model)
    np.savez_compressed(
        os.path.join(out, 'convtasnet_default.npz'),
        batch=batch.numpy(), lengths=lengths.numpy(), output=out_t.detach().numpy(),
        loss=loss.detach().numpy(), grad_norms=gnorm.numpy(),
        param_sum=flat.double().sum().numpy(),
        param_abs_sum=flat.double().abs().sum().numpy(),
        first_params=flat[:64].numpy())


def golden_training(out):
    """The 2-epoch flow of the reference's tests/test_training.py for the dummy
    and convtasnet models (val_metrics reduced to {'snr'}: PESQ/ESTOI wheels are
    absent). Records the first 10 parameters after training."""
    sys.path.insert(0, os.path.join(REF, 'tests'))
    from utils import DummyDataset, DummyModel
    from brever.models import ModelRegistry
    from brever.training import BreverTrainer
    FS = 16000
    res = {}
    for tag, ctor, sources in [
        ('dummy', lambda: DummyModel(channels=2, output_sources=2), 3),
        ('convtasnet', lambda: ModelRegistry.get('convtasnet')(
            filters=4, filter_length=2, bottleneck_channels=1, hidden_channels=1,
            skip_channels=1, kernel_size=1, layers=1, repeats=1,
            output_sources=2), 3),
        # reference tests/test_training.py:125-150 ('sgmse'): its random draws (training t and
        # noise, the sampler's noise during validation) all come from the global CPU generator
        ('sgmse', lambda: ModelRegistry.get('sgmsep')(
            stft_frame_length=512, stft_hop_length=256, net_base_channels=4,
            net_channel_mult=[1, 1, 1, 1], net_num_blocks_per_res=1, net_noise_channel_mult=1,
            net_emb_channel_mult=1, net_fir_kernel=[1, 1], net_attn_resolutions=[0],
            net_attn_bottleneck=False, solver_num_steps=1), 2),
        # reference tests/test_training.py:196-217 ('tfgridnet')
        ('tfgridnet', lambda: ModelRegistry.get('tfgridnet')(
            n_srcs=2, n_layers=1, lstm_hidden_units=1, attn_n_head=1, attn_approx_qk_dim=1,
            emb_dim=1), 3),
    ]:
        torch.manual_seed(0)
        random.seed(0)
        np.random.seed(0)
        model = ctor()
        train = DummyDataset(16, sources, 2, int(FS*0.5), FS*4,
                             transform=model.transform)
        val = DummyDataset(4, sources, 2, int(FS*0.5), FS*4)
        with tempfile.TemporaryDirectory() as tmp:
            trainer = BreverTrainer(
                model=model, train_dataset=train, val_dataset=val,
                model_dirpath=tmp, epochs=2, val_period=1, val_metrics={'snr'},
                batch_sampler='bucket', batch_size=8.0, dynamic_batch_size=True,
                ema=True, device='cpu', preload=True)
            trainer.run()
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        res[tag] = flat[:10].numpy()
        res[tag + '_train_loss'] = np.array(
            [float(d['loss']) for d in trainer.loss_logger.train_loss])
        res[tag + '_val_loss'] = np.array(
            [float(d['loss']) for d in trainer.loss_logger.val_loss])
    np.savez(os.path.join(out, 'training.npz'), **res)


def golden_stft(out):
    """STFT.forward / backward of the reference on the seeded signal of its own
    round-trip test (tests/test_modules.py:319-326), frame counts for edge lengths,
    and the default MelFilterbank."""
    from brever.modules import STFT, MelFilterbank
    g = torch.Generator().manual_seed(42)
    x = torch.randn(1, 4096, generator=g)
    res = dict(x=x.numpy())
    combos = [(256, 1.0, 1.0, True), (128, 0.5, 0.15, False), (128, 1.0, 1.0, True),
              (256, 0.5, 0.15, True)]
    res['combos'] = np.array(combos, dtype=np.float64)
    for i, (hop, comp, scale, norm) in enumerate(combos):
        stft = STFT(frame_length=512, hop_length=hop, compression_factor=comp,
                    scale_factor=scale, normalized=norm)
        X = stft(x)
        res[f'spec{i}'] = X.numpy()
        res[f'back{i}'] = stft.backward(X.clone()).numpy()
    odd = STFT(frame_length=512, hop_length=128)
    y = torch.randn(3000, generator=g)
    res['x_odd'] = y.numpy()
    res['spec_odd'] = odd(y).numpy()
    res['back_odd'] = odd.backward(odd(y)).numpy()
    lens = [1, 511, 512, 513, 3000, 63999, 64000, 64001]
    res['lens'] = np.array(lens)
    for hop in (128, 256):
        s = STFT(frame_length=512, hop_length=hop)
        res[f'frames{hop}'] = np.array([s(torch.zeros(n)).shape[-1] for n in lens])
    # ConvSTFT (stft.py:201-319): forward / backward for three parameter combos
    from brever.modules import ConvSTFT
    ccombos = [(512, 256, 1.0, 1.0, True), (256, 64, 0.5, 0.3, True), (128, 64, 1.0, 1.0, False)]
    res['conv_combos'] = np.array(ccombos, dtype=np.float64)
    for i, (n, hop, comp, scale, norm) in enumerate(ccombos):
        cs = ConvSTFT(frame_length=n, hop_length=hop, compression_factor=comp,
                      scale_factor=scale, normalized=norm)
        X = cs(y.unsqueeze(0))
        res[f'conv_spec{i}'] = X.numpy()
        re, im = X.real.clone(), X.imag.clone()
        res[f'conv_back{i}'] = cs.backward((re, im), input_type='real_imag').numpy()
    mel = MelFilterbank()
    res['mel_filters'] = mel.filters.numpy()
    res['mel_fc'] = mel.fc.numpy()
    res['mel_scaling'] = mel.scaling.numpy()
    spec = torch.randn(2, 257, 12, generator=g).abs()
    res['mel_in'] = spec.numpy()
    res['mel_fwd'] = mel(spec).numpy()
    res['mel_bwd'] = mel.backward(mel(spec)).numpy()
    np.savez_compressed(os.path.join(out, 'stft.npz'), **res)


def golden_ffnn(out):
    """FFNN (brever/models/ffnn/ffnn.py): seeded parameters, transform of a seeded item,
    forward / loss / gradients at fixed weights (dropout 0, so no RNG enters), enhance in
    eval mode, default parameter count."""
    from brever.models import FFNN, count_params
    torch.manual_seed(0)
    res = dict(n_params_default=np.array(count_params(FFNN())))
    net = FFNN(hidden_layers=[96, 80], dropout=0.0)
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        net.normalization.set_statistics(torch.randn(384, 1, generator=g),
                                         torch.rand(384, 1, generator=g) + 0.5)
    res['params'] = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy()
    res['mean'] = net.normalization.mean.numpy().copy()
    res['std'] = net.normalization.std.numpy().copy()
    clean = 0.1*torch.randn(2, 9000, generator=g)
    noise = 0.05*torch.randn(2, 9000, generator=g)
    sources = torch.stack([clean + noise, clean])              # (mixture, foreground) x channels
    res['sources'] = sources.numpy()
    item = net.transform(sources)
    res['item'] = item.numpy()
    items = [net.transform(torch.stack([clean[:, :L] + noise[:, :L], clean[:, :L]]))
             for L in (9000, 7000, 4000)]
    T = max(i.shape[-1] for i in items)
    batch = torch.stack([torch.nn.functional.pad(i, (0, T - i.shape[-1])) for i in items])
    lengths = torch.tensor([i.shape[-1] for i in items])
    res['batch'] = batch.numpy(); res['lengths'] = lengths.numpy()
    net.train()
    res['output'] = net(batch[:, :384]).detach().numpy()
    loss = net.loss(batch, lengths, False)
    res['loss'] = loss.detach().numpy()
    loss.backward()
    res['grads'] = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).numpy()
    net.eval()
    mix = torch.stack([(clean + noise)[:, :8000], (clean + noise)[:, :8000]*0.5])   # (B, 2, L)
    with torch.no_grad():
        res['enhance_in'] = mix.numpy()
        res['enhance_out'] = net.enhance(mix).numpy()
    cum = FFNN(hidden_layers=[32], normalization='cumulative', dropout=0.0)
    with torch.no_grad():
        res['cumnorm_out'] = cum.normalization(batch[:, :384]).numpy()
    np.savez_compressed(os.path.join(out, 'ffnn.npz'), **res)


def golden_dccrn(out):
    """DCCRN (brever/models/dccrn/dccrn.py): a narrow configuration at seeded weights,
    forward in train mode (batch statistics; running estimates after the call) and in eval
    mode; default parameter count."""
    from brever.models import DCCRN, count_params
    torch.manual_seed(0)
    res = dict(n_params_default=np.array(count_params(DCCRN())))
    cfg = dict(channels=[4, 8, 8, 16, 16, 16], lstm_channels=24, lstm_layers=2)
    res['config'] = json.dumps(cfg)
    net = DCCRN(**cfg)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if 'norm' in name or 'activation' in name:
                p.add_(0.1*torch.randn(p.shape, generator=g))
    res['params'] = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy()
    x = 0.3*torch.randn(2, 3000, generator=g)
    res['x'] = x.numpy()
    net.train()
    with torch.no_grad():
        res['out_train'] = net(x).numpy()
    res['running'] = torch.cat([b.detach().reshape(-1).float() for n, b in net.named_buffers()
                                if 'running' in n]).numpy()
    # loss and all gradients in train mode (snr criterion, ragged lengths)
    batch = torch.stack([x, 0.7*x + 0.1*torch.randn(2, 3000, generator=g)], dim=1)
    lengths = torch.tensor([3000, 2500])
    res['batch'] = batch.numpy(); res['lengths'] = lengths.numpy()
    net.zero_grad()
    loss = net.loss(batch, lengths, use_amp=False)
    loss.backward()
    res['loss'] = loss.detach().numpy()
    res['grads'] = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).numpy()
    net.eval()
    with torch.no_grad():
        res['out_eval'] = net(x).numpy()
        res['enhance'] = net.enhance(torch.stack([x, 0.5*x], dim=1)).numpy()
    # the complex batch norm variant (complex_batchnorm.py): same recipe, prefixed 'cbn_'
    torch.manual_seed(0)
    cfg2 = dict(cfg, use_complex_batchnorm=True)
    net = DCCRN(**cfg2)
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if 'norm' in name or 'activation' in name:
                p.add_(0.1*torch.randn(p.shape, generator=g))
    res['cbn_config'] = json.dumps(cfg2)
    res['cbn_params'] = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy()
    net.train()
    with torch.no_grad():
        res['cbn_out_train'] = net(x).numpy()
    res['cbn_running'] = torch.cat([b.detach().reshape(-1).float() for n, b in net.named_buffers()
                                    if 'running' in n]).numpy()
    net.zero_grad()
    loss = net.loss(batch, lengths, use_amp=False)
    loss.backward()
    res['cbn_loss'] = loss.detach().numpy()
    res['cbn_grads'] = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).numpy()
    net.eval()
    with torch.no_grad():
        res['cbn_out_eval'] = net(x).numpy()
    np.savez_compressed(os.path.join(out, 'dccrn.npz'), **res)


SGMSE_CASES = {
    # NCSN++ style: skip encoder / skip decoder, attention at one resolution and in the
    # bottleneck, odd frame count (padding stack), predictor-corrector sampler
    'pc': ('sgmsep', dict(stft_frame_length=64, stft_hop_length=16, net_base_channels=8,
                          net_channel_mult=[1, 2, 2], net_num_blocks_per_res=1,
                          net_attn_resolutions=[16], solver_num_steps=3), 600),
    # ADM blocks, standard encoder / decoder, box FIR, cosine SDE + Heun sampler
    'edm': ('idmse', dict(stft_frame_length=64, stft_hop_length=16, net_base_channels=8,
                          net_channel_mult=[1, 2, 2], solver_num_steps=3,
                          solver_edm_schurn=1.0), 560),
    # residual encoder / residual decoder (denoiser only)
    'res': ('sgmsep', dict(stft_frame_length=64, stft_hop_length=16, net_base_channels=8,
                           net_channel_mult=[1, 2], net_num_blocks_per_res=1,
                           net_attn_resolutions=[], net_attn_bottleneck=False,
                           net_encoder_type='residual', net_decoder_type='residual',
                           solver_num_steps=2), 500),
}


def golden_sgmse(out):
    """SGMSE+ (brever/models/sgmse): seeded narrow models; the preconditioned denoiser at one
    noise level and a full ``enhance`` whose Gaussian draws are recorded (in call order) so
    that another implementation can replay them."""
    import brever.models.sgmse.sdes as sdes
    import brever.models.sgmse.solvers as solvers
    from brever.models import ModelRegistry
    res = {}
    for tag, (arch, cfg, length) in SGMSE_CASES.items():
        torch.manual_seed(3)
        net = ModelRegistry.get(arch)(**cfg)
        g = torch.Generator().manual_seed(11)
        with torch.no_grad():
            for name, p in net.named_parameters():
                if 'norm' in name:
                    p.add_(0.1*torch.randn(p.shape, generator=g))
        net.eval()
        res[f'{tag}_arch'] = np.array(arch)
        res[f'{tag}_config'] = np.array(json.dumps(cfg))
        res[f'{tag}_params'] = torch.cat([p.detach().reshape(-1)
                                          for p in net.parameters()]).numpy()
        wav = 0.3*torch.randn(1, 2, length, generator=g)
        res[f'{tag}_wav'] = wav.numpy()
        with torch.no_grad():
            y = net.stft(wav.mean(axis=-2, keepdims=True)/wav.mean(axis=-2).abs().max())
            y = y[..., :-1, :]
            x = y + 0.2*torch.randn(y.shape, generator=g).to(y.dtype)
            t = torch.tensor(0.6)
            sigma = net.sde.sigma(t)
            res[f'{tag}_den_x'] = x.numpy(); res[f'{tag}_den_y'] = y.numpy()
            res[f'{tag}_den_t'] = t.numpy()
            res[f'{tag}_den_out'] = net(x, y, sigma, t).numpy()
            draws = []
            real_randn, real_randn_like = torch.randn, torch.randn_like

            def randn(*a, **k):
                k.pop('device', None)
                d = real_randn(*a, generator=g, **k)
                draws.append(d)
                return d

            def randn_like(ref, **k):
                d = real_randn(ref.shape, generator=g, dtype=ref.dtype)
                draws.append(d)
                return d
            patched = types.SimpleNamespace(**{n: getattr(torch, n) for n in dir(torch)})
            patched.randn, patched.randn_like = randn, randn_like
            sdes.torch, solvers.torch = patched, patched
            try:
                if tag != 'res':
                    res[f'{tag}_enhance'] = net.enhance(wav.clone()).numpy()
            finally:
                sdes.torch, solvers.torch = torch, torch
            for i, d in enumerate(draws):
                res[f'{tag}_noise_{i}'] = d.numpy()
            res[f'{tag}_n_noise'] = np.array(len(draws))
        # training objective (sgmse.py:163-176) on a ragged batch of two items with the draws of
        # t and of the Gaussian noise fixed: loss value and all parameter gradients
        import brever.models.sgmse.sgmse as sg
        items = [net.transform(0.3*torch.randn(2, 2, n, generator=g)) for n in (length, length - 90)]
        lengths = torch.tensor([it.shape[-1] for it in items])
        batch = torch.stack([torch.nn.functional.pad(it, (0, int(lengths.max()) - it.shape[-1]))
                             for it in items])
        t_draw = torch.rand(2, 1, 1, 1, generator=g)
        n_draw = torch.randn(batch[:, 1].unsqueeze(1).shape, generator=g, dtype=batch.dtype)
        patched = types.SimpleNamespace(**{n: getattr(torch, n) for n in dir(torch)})
        patched.rand = lambda *a, **k: t_draw.clone()
        patched.randn_like = lambda ref, **k: n_draw.clone()
        sg.torch = patched
        try:
            net.train()
            net.zero_grad()
            loss = net.loss(batch, lengths, use_amp=False)
            loss.backward()
        finally:
            sg.torch = torch
            net.eval()
        res[f'{tag}_train_batch'] = batch.numpy(); res[f'{tag}_train_lengths'] = lengths.numpy()
        # the reference maps the uniform draw to [t_eps, 1]: the effective t is what is stored
        res[f'{tag}_train_t'] = (t_draw*(1 - net.t_eps) + net.t_eps).numpy()
        res[f'{tag}_train_noise'] = n_draw.numpy()
        res[f'{tag}_train_loss'] = loss.detach().numpy()
        res[f'{tag}_train_grads'] = torch.cat([p.grad.reshape(-1) if p.grad is not None
                                               else torch.zeros(p.numel())
                                               for p in net.parameters()]).numpy()
    np.savez_compressed(os.path.join(out, 'sgmse.npz'), **res)


def golden_features(out):
    """FeatureExtractor (brever/modules/features.py) on a seeded 2-channel spectrum: every
    feature except 'ic' (needs torchaudio.lfilter), one by one and concatenated."""
    from brever.modules import FeatureExtractor, MelFilterbank
    g = torch.Generator().manual_seed(21)
    spec = torch.randn(2, 2, 257, 23, generator=g, dtype=torch.complex64)
    names = ['cubicfbe', 'cubicmfcc', 'cubicpdf', 'fbe', 'ild', 'ipd', 'logfbe', 'logpdf', 'mfcc',
             'pdf', 'pdfcc']
    res = dict(spec=spec.numpy(), names=np.array(names))
    mel = MelFilterbank()
    for name in names:                       # batched, one feature at a time
        res[name] = FeatureExtractor({name}, mel).calc_feature(spec, name).numpy()
    fx = FeatureExtractor(set(names), mel)
    res['all'] = fx(spec[0]).numpy()         # the reference's __call__ takes unbatched input
    res['n_features'] = np.array(fx.n_features)
    np.savez_compressed(os.path.join(out, 'features.npz'), **res)


def golden_ema(out):
    """EMA / EMAKarras (brever/modules/ema.py) on a tiny Linear model through 12 seeded parameter
    updates: running averages, Karras exponents, post-hoc weights and the post-hoc average
    reconstructed from per-step checkpoints."""
    import brever.modules.ema as ema_mod
    from brever.modules import EMA, EMAKarras
    # torch >= 2.6 loads with weights_only=True by default, which rejects the numpy scalars of
    # the reference's own checkpoints: load them the way the pinned torch did
    patched = types.SimpleNamespace(**{n: getattr(torch, n) for n in dir(torch)})
    patched.load = lambda f, **k: torch.load(f, weights_only=False)
    ema_mod.torch = patched
    g = torch.Generator().manual_seed(5)
    model = torch.nn.Linear(6, 5)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(torch.randn(p.shape, generator=g))
    res = dict(init=torch.cat([p.detach().reshape(-1) for p in model.parameters()]).numpy())
    ema = EMA(model, beta=0.97)
    kar = EMAKarras(model, sigma_rels=[0.05, 0.1])
    steps = []
    with tempfile.TemporaryDirectory() as d:
        for i in range(12):
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(0.1*torch.randn(p.shape, generator=g))
            steps.append(torch.cat([p.detach().reshape(-1) for p in model.parameters()]).numpy())
            ema.update(); kar.update()
            torch.save(kar.state_dict(), f'{d}/{i:02d}.ckpt')
        post = kar.post_hoc_ema(d, 0.2, apply=False)
        post2 = kar.post_hoc_ema(d, [0.15, 0.3], t_r=[8, 12], apply=False)
    flat = lambda ps: torch.cat([p.reshape(-1) for p in ps]).numpy()   # noqa: E731
    res.update(params=np.stack(steps), ema=flat(ema.ema_params),
               kar_005=flat(kar.ema_params[0.05]), kar_010=flat(kar.ema_params[0.1]),
               gammas=np.array([kar._gammas[0.05], kar._gammas[0.1]]),
               post=flat(post), post2=np.stack([flat(p) for p in post2]),
               weights=EMAKarras.solve_weights([3, 7, 12], [5.0, 9.0, 5.0], [12, 10], [6.5, 7.0]))
    np.savez_compressed(os.path.join(out, 'ema.npz'), **res)


def golden_tfgridnet(out):
    """TF-GridNet (brever/models/tfgridnet/tfgridnet.py): a narrow configuration at seeded
    weights (norm gains / biases and PReLU slopes perturbed): forward output, multiresyu loss and
    all gradients on a ragged two-item batch, `enhance`, default parameter count; the same for a
    two-source network with odd frame / band counts that need the grid padding."""
    from brever.models import count_params
    from brever.models.tfgridnet import TFGridNet
    torch.manual_seed(0)
    res = dict(n_params_default=np.array(count_params(TFGridNet())))
    cases = {
        'a': (dict(n_fft=32, stride=16, n_layers=2, lstm_hidden_units=16, attn_n_head=2,
                   attn_approx_qk_dim=34, emb_dim=8), 400, 11),
        'b': (dict(n_srcs=2, n_fft=24, stride=8, n_layers=1, lstm_hidden_units=32, attn_n_head=4,
                   attn_approx_qk_dim=20, emb_dim=8, emb_ks=2, emb_hs=2), 333, 12),
        # overlapping windows: unfold + transposed-convolution branch
        'c': (dict(n_fft=24, stride=12, n_layers=1, lstm_hidden_units=16, attn_n_head=2,
                   attn_approx_qk_dim=20, emb_dim=4, emb_ks=4, emb_hs=2), 300, 13),
    }
    for tag, (cfg, L, seed) in cases.items():
        torch.manual_seed(seed)
        net = TFGridNet(**cfg)
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for name, p in net.named_parameters():
                if 'norm' in name or 'act' in name or name.startswith('conv.1') \
                        or 'attn_concat_proj.1' in name or 'attn_concat_proj.2' in name:
                    p.add_(0.1*torch.randn(p.shape, generator=g))
        res[tag + '_config'] = json.dumps(cfg)
        res[tag + '_names'] = json.dumps([n for n, _ in net.named_parameters()])
        res[tag + '_params'] = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy()
        S = cfg.get('n_srcs', 1)
        mix = 0.3*torch.randn(2, 2, L, generator=g)
        tgt = 0.3*torch.randn(2, S, 2, L, generator=g)
        batch = torch.cat([mix[:, None], tgt], dim=1)            # (B, 1 + S, 2, L)
        lengths = torch.tensor([L, L - 57])
        res[tag + '_batch'] = batch.numpy(); res[tag + '_lengths'] = lengths.numpy()
        with torch.no_grad():
            res[tag + '_out'] = net(mix).numpy()
        net.zero_grad()
        loss = net.loss(batch, lengths, use_amp=False)
        loss.backward()
        res[tag + '_loss'] = loss.detach().numpy()
        res[tag + '_grads'] = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).numpy()
        with torch.no_grad():
            res[tag + '_enhance'] = net.enhance(mix).numpy()
    np.savez_compressed(os.path.join(out, 'tfgridnet.npz'), **res)


def golden_norms(out):
    """CausalGroupNorm / LayerNorm / InstanceNorm (brever/modules/normalization.py) on a seeded
    (B, C, F, T) tensor with non-trivial gain / bias: outputs and the gradients wrt the input,
    gain and bias for a random upstream gradient; also with the frames on axis 2."""
    from brever.modules import CausalGroupNorm, CausalInstanceNorm, CausalLayerNorm
    g = torch.Generator().manual_seed(8)
    x = torch.randn(3, 6, 5, 37, generator=g)
    gy = torch.randn(3, 6, 5, 37, generator=g)
    gain = 1 + 0.3*torch.randn(6, generator=g)
    bias = 0.2*torch.randn(6, generator=g)
    res = dict(x=x.numpy(), gy=gy.numpy(), gain=gain.numpy(), bias=bias.numpy())
    for tag, ctor in (('layer', lambda: CausalLayerNorm(6)), ('group', lambda: CausalGroupNorm(6, 2)),
                      ('instance', lambda: CausalInstanceNorm(6)),
                      ('group_t2', lambda: CausalGroupNorm(6, 3, time_dim=2))):
        norm = ctor()
        with torch.no_grad():
            norm.gain.copy_(gain); norm.bias.copy_(bias)
        xg = x.clone().requires_grad_(True)
        y = norm(xg)
        (y*gy).sum().backward()
        res[tag] = y.detach().numpy()
        res[tag + '_dx'] = xg.grad.numpy()
        res[tag + '_dgain'] = norm.gain.grad.numpy()
        res[tag + '_dbias'] = norm.bias.grad.numpy()
    np.savez_compressed(os.path.join(out, 'norms.npz'), **res)


def golden_segments(out):
    """Segment tables of BreverDataset.get_segment_info (brever/data.py:112-210) for seeded
    file lengths x strategies x (segment, overlap, max segment) settings; the file-length
    query (torchaudio.info on FLAC members) is replaced by the given list."""
    from brever.data import BreverDataset
    rng = random.Random(7)
    cases = []
    for _ in range(120):
        lens = [rng.randint(1, 50000) for _ in range(rng.randint(1, 5))]
        seg = rng.choice([0, 100, 1600, 16000, 20000])
        ov = rng.choice([0, 0, 10, 800]) if seg else 0
        if seg and ov >= seg:
            ov = 0
        strat = rng.choice(['drop', 'pass', 'pad', 'overlap', 'random'])
        mx = rng.choice([0, 0, 8000, 30000])
        ref = object.__new__(BreverDataset)
        ref.segment_length, ref.overlap_length, ref.segment_strategy = seg, ov, strat
        ref.max_segment_length, ref.fs, ref.rmm_dset = mx, 16000, None
        ref.get_file_lengths = lambda l=lens: l
        logging_off = __import__('logging')
        logging_off.disable(logging_off.WARNING)
        ref.get_segment_info()
        logging_off.disable(logging_off.NOTSET)
        cases.append(dict(lengths=lens, segment=seg, overlap=ov, strategy=strat, max_segment=mx,
                          segment_after=ref.segment_length,
                          table=[[i, s, e] for i, (s, e) in ref._segment_info]))
    with open(os.path.join(out, 'segments.json'), 'w') as f:
        json.dump(cases, f)


IN_SCOPE_MODELS = ['convtasnet', 'dccrn', 'ffnn', 'sgmsep', 'sgmsepm', 'sgmsepheun',
                   'sgmsepmheun', 'idmse', 'tfgridnet']


def _jsonable(x):
    if isinstance(x, dict):
        return {k: _jsonable(v) for k, v in x.items()}
    if isinstance(x, (set, frozenset)):
        return {'__set__': sorted(_jsonable(v) for v in x)}
    if isinstance(x, tuple):
        return {'__tuple__': [_jsonable(v) for v in x]}
    if isinstance(x, list):
        return [_jsonable(v) for v in x]
    return x


def golden_config(out):
    """Config / CLI contract of the reference (brever/config.py, args.py, inspect.py):
    default config dicts and their ``get_hash()`` for the in-scope models, hashes after 20
    seeded field perturbations, ``get_func_spec`` of BreverDataset / BreverTrainer / the models
    (type and action as names), and the parsed namespace + resulting hash of the command lines
    the reference's own tests/test_args.py builds (``--arg=default`` for every option)."""
    from brever.args import ModelArgParser
    from brever.config import get_model_default_config
    from brever.data import BreverDataset
    from brever.inspect import get_func_spec
    from brever.models import ModelRegistry
    from brever.training import BreverTrainer

    def spec_json(func):
        res = {}
        for arg, item in get_func_spec(func).items():
            action = item['action']
            res[arg] = {
                'type': item['type'].__name__,
                'action': None if action is None else [action.origin.__name__, action.type_.__name__],
                'default': _jsonable(item['default']),
                'required': item['required'],
            }
        return res

    fixture = {'defaults': {}, 'hashes': {}, 'perturbed': [], 'specs': {}, 'commands': {}}
    fixture['specs']['BreverDataset'] = spec_json(BreverDataset)
    fixture['specs']['BreverTrainer'] = spec_json(BreverTrainer)
    rng = random.Random(7)
    for key in IN_SCOPE_MODELS:
        cfg = get_model_default_config(key)
        fixture['defaults'][key] = _jsonable(cfg.to_dict())
        fixture['hashes'][key] = [cfg.get_hash(), cfg.get_hash(length=12)]
        fixture['specs'][key] = spec_json(ModelRegistry.get(key))
        # the command of tests/test_args.py:42-66
        cmd = ['--seed=0', '--train_path=foo', '--val_path=bar']
        for func in (BreverDataset, BreverTrainer):
            for arg, x in get_func_spec(func).items():
                d = x['default']
                cmd.append(f'--{arg}=' + (','.join(str(y) for y in d)
                                         if isinstance(d, (list, tuple, set)) else str(d)))
        cmd.append(key)
        for arg, x in get_func_spec(ModelRegistry.get(key)).items():
            d = x['default']
            cmd.append(f'--{arg}=' + (','.join(str(y) for y in d)
                                     if isinstance(d, (list, tuple, set)) else str(d)))
        parser = ModelArgParser()
        args = parser.parse_args(cmd)
        cfg2 = get_model_default_config(key)
        cfg2.update_from_args(args, parser.arg_map(key))
        fixture['commands'][key] = {'argv': cmd, 'namespace': _jsonable(vars(args)),
                                    'hash': cfg2.get_hash(),
                                    'arg_map': parser.arg_map(key)}
    # seeded perturbations of scalar fields
    for _ in range(20):
        key = rng.choice(IN_SCOPE_MODELS)
        cfg = get_model_default_config(key)
        section = rng.choice(['model', 'trainer', 'dataset'])
        fields = [(k, v) for k, v in getattr(cfg, section).to_dict().items()
                  if isinstance(v, (int, float, bool, str))]
        name, value = rng.choice(fields)
        if isinstance(value, bool):
            new = not value
        elif isinstance(value, int):
            new = value + rng.randint(1, 9)
        elif isinstance(value, float):
            new = value*1.5 + 0.25
        else:
            new = value + '_x'
        cfg.set_field([section, name], new)
        fixture['perturbed'].append({'model': key, 'field': [section, name],
                                     'value': new, 'hash': cfg.get_hash()})
    with open(os.path.join(out, 'config.json'), 'w') as f:
        json.dump(fixture, f, indent=1, sort_keys=True)


def golden_stft_matrix(out):
    """The reference's own STFT test matrix (tests/test_modules.py:300-326: 32 combinations of
    hop / compression / scale / normalized / onesided on randn(4096, seed 42)) plus geometries
    it does not test (n_fft > frame_length, hop not dividing the frame, center=False, two-sided
    with a short n_fft): per case 192 sampled spectrum values (fixed pseudo-random positions),
    the spectrum's energy, the round-trip output and, for 4 cases, gradients of a seeded
    quadratic form through forward (incl. compression) and backward."""
    import itertools
    from brever.modules import STFT
    gen = torch.Generator().manual_seed(42)
    x = torch.randn(4096, generator=gen)
    cases = [dict(frame_length=512, hop_length=h, compression_factor=c, scale_factor=s,
                  normalized=nm, onesided=o)
             for h, c, s, nm, o in itertools.product([256, 128], [1.0, 0.5], [1.0, 0.15],
                                                     [False, True], [False, True])]
    cases += [dict(frame_length=400, hop_length=160, n_fft=512),
              dict(frame_length=400, hop_length=100, n_fft=512, compression_factor=0.5),
              dict(frame_length=512, hop_length=200),
              dict(frame_length=256, hop_length=64, center=False, window='hamming'),
              dict(frame_length=60, hop_length=25, n_fft=64, onesided=False, window='hamming'),
              dict(frame_length=512, hop_length=128, window=None, normalized=False),
              dict(frame_length=512, hop_length=128, pad_mode='reflect'),
              dict(frame_length=400, hop_length=160, n_fft=512, pad_mode='replicate', onesided=False)]
    data = {'x': x.numpy(), 'cases': json.dumps(cases)}
    pick = np.random.default_rng(0)
    for i, kw in enumerate(cases):
        stft = STFT(**kw)
        # (F.pad only reflects / replicates batched input: those cases take x as (1, 4096))
        X = stft(x if kw.get('pad_mode', 'constant') == 'constant' else x.unsqueeze(0))
        flat = X.reshape(-1)
        idx = pick.integers(0, flat.numel(), 192)
        data[f'idx{i}'] = idx
        data[f'val{i}'] = torch.view_as_real(flat[idx]).numpy()
        data[f'shape{i}'] = np.array(X.shape)
        data[f'energy{i}'] = np.array(float((X.abs()**2).sum()))
        data[f'rt{i}'] = stft.backward(X.clone()).numpy()
    # gradients: L = sum Re(conj(G) STFT(x)) and L = sum g * ISTFT(X)
    for j, i in enumerate([0, 5, 12, 33]):
        stft = STFT(**cases[i])
        xg = x.clone().requires_grad_(True)
        X = stft(xg)
        G = torch.randn(X.shape, generator=gen) + 1j*torch.randn(X.shape, generator=gen)
        (X*G.conj()).real.sum().backward()
        data[f'gcase{j}'] = np.array(i)
        data[f'G{j}'] = torch.view_as_real(G).numpy()
        data[f'dx{j}'] = xg.grad.numpy()
        Xg = X.detach().clone().requires_grad_(True)
        y = stft.backward(Xg*1.0)
        g = torch.randn(y.shape, generator=gen)
        (y*g).sum().backward()
        data[f'g{j}'] = g.numpy()
        data[f'dX{j}'] = torch.view_as_real(Xg.grad).numpy()
    np.savez_compressed(os.path.join(out, 'stft_matrix.npz'), **data)


def main():
    install_stubs()
    sys.path.insert(0, REF)
    os.chdir(REF)       # the reference opens config/... relatively
    torch.set_num_threads(4)
    todo = [golden_batching, golden_collate, golden_losses, golden_convtasnet, golden_training,
            golden_stft, golden_ffnn, golden_dccrn, golden_sgmse, golden_segments, golden_features, golden_ema, golden_norms, golden_tfgridnet, golden_config, golden_stft_matrix]
    only = sys.argv[1:]                  # e.g. `make_golden.py sgmse` regenerates one file
    for fn in todo:
        if not only or fn.__name__[len('golden_'):] in only:
            fn(HERE)
    print('golden fixtures written to', HERE)


if __name__ == '__main__':
    main()
