"""The three size gaps VERDICT r03 item 4 names, closed at each configuration's OWN size:

* SGMSE+ (BASELINE config 4): the default 65.6 M-parameter network, one 4 s utterance (256 x 501 spectrogram),
  `use_amp`, the FULL 30-step predictor-corrector `enhance` (60 chained fp16 network evaluations) with the
  Gaussian draws replayed, against oracle/sgmse.py on the CPU -- the waveform and the growth of the
  deviation step by step (reference brever/models/sgmse/sgmse.py:178-193, solvers.py:54-77);
* DCCRN (BASELINE config 3): default size, 2 x 4 s, ALL parameter gradients in fp32 and under `use_amp`
  against the autograd of oracle/dccrn.py (brever/models/dccrn/dccrn.py:145-222);
* Conv-TasNet (BASELINE config 1): default widths, 8 layers x 1 repeat, 10 fused bf16 training steps against
  the oracle under CPU-bf16 autocast AND against the fp32 oracle: the drift is printed and bounded
  (SURVEY.md 8d iii).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cuda():
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a ROCm device')
    return torch.device('cuda:0')


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm()/(b.norm() + 1e-30))


def test_sgmse_default_network_full_30_step_enhance_use_amp():
    """The one slow test of the suite (about 2 - 4 minutes of host time for the 60 oracle evaluations)."""
    import functools

    import scipy.signal

    from brever_amd.models import SGMSEp
    from oracle import sgmse as osg
    dev = _cuda()
    steps = 30
    torch.manual_seed(1)
    model = SGMSEp(solver_num_steps=steps)
    net = osg.Net(model.state_dict(), 'model.net.', skip_scale=0.5**0.5)
    sde = osg.RichterOUVE()
    gen = torch.Generator().manual_seed(21)
    wav = 0.1*torch.randn(1, 2, 64000, generator=gen)

    draws = []

    def draw(shape, complex_, idx):
        g = torch.Generator().manual_seed(7000 + idx)
        return torch.randn(tuple(shape), dtype=torch.complex64 if complex_ else torch.float32, generator=g)

    def oracle_noise(shape, complex_):
        d = draw(shape, complex_, len(draws))
        draws.append((tuple(shape), complex_))
        return d
    states_ref = []
    sampler = functools.partial(osg.pc_sample, noise=oracle_noise, num_steps=steps, corrector_steps=1,
                                corrector_snr=0.5, on_step=lambda i, x: states_ref.append(x.clone()))
    window = scipy.signal.get_window('hann', 512)
    torch.set_num_threads(max(1, min(64, torch.get_num_threads())))
    with torch.no_grad():
        want = osg.enhance(net, sde, wav, window, 128, sampler)
    assert len(states_ref) == steps and states_ref[0].shape[-2:] == (256, 501)

    model = model.to(dev).eval()
    count = [0]

    def hip_noise(shape, complex_):
        k = count[0]
        count[0] += 1
        assert draws[k] == (tuple(shape), complex_), (k, draws[k], shape, complex_)
        return draw(shape, complex_, k)
    states = []
    model._noise_source = hip_noise
    model._step_hook = lambda i, x: states.append(x.detach().clone())
    got = model.enhance(wav.to(dev), use_amp=True)
    assert count[0] == len(draws) == 2*steps            # initial draw + 30 corrector + 29 predictor draws
    growth = [rel(torch.view_as_real(a), torch.view_as_real(b)) for a, b in zip(states, states_ref)]
    e = rel(got, want)
    print('default SGMSE+ 30-step PC enhance, use_amp (60 fp16 network evaluations): waveform rel-L2 %.3e' % e)
    print('   state deviation after step 1, 5, 10, 15, 20, 25, 30:',
          ' '.join('%.2e' % growth[i] for i in (0, 4, 9, 14, 19, 24, 29)))
    assert got.shape == want.shape
    # measured: waveform 2.2e-4; the state's deviation grows from 1.4e-4 (step 1) to 2.8e-4 (step 30) and levels off
    assert e <= 2e-3, e
    assert max(growth) <= 3e-3, max(growth)
    # the fp32 path on the same draws
    count[0] = 0
    states.clear()
    got32 = model.enhance(wav.to(dev), use_amp=False)
    e32 = rel(got32, want)
    print('   fp32 path: waveform rel-L2 %.3e, last state %.3e' % (e32, rel(torch.view_as_real(states[-1]),
                                                                     torch.view_as_real(states_ref[-1]))))
    assert e32 <= 1e-4, e32                             # measured 4.9e-7


def test_dccrn_default_size_gradients_fp32_and_use_amp():
    from brever_amd.models import DCCRN
    from oracle.criterion import snr as osnr
    from oracle.dccrn import OracleDCCRN
    dev = _cuda()
    torch.manual_seed(4)
    net = DCCRN()
    oracle = OracleDCCRN()
    with torch.no_grad():
        flat = torch.cat([p.reshape(-1) for p in net.parameters()])
        o = 0
        for p in oracle.parameters():
            p.copy_(flat[o:o + p.numel()].view_as(p))
            o += p.numel()
    gen = torch.Generator().manual_seed(9)
    B, L = 2, 64000
    batch = 0.1*torch.randn(B, 2, L, generator=gen)
    lengths = torch.tensor([L, L - 5000])
    batch[1, :, lengths[1]:] = 0
    oracle.train()
    torch.set_num_threads(max(1, min(64, torch.get_num_threads())))
    out = oracle(batch[:, 0])
    loss_ref = osnr(out, batch[:, 1], lengths).mean()
    loss_ref.backward()
    g_ref = torch.cat([p.grad.reshape(-1) for p in oracle.parameters()])
    names = [n for n, _ in net.named_parameters()]
    sizes = [p.numel() for p in net.parameters()]
    # the yardstick of the use_amp run: the same oracle with the HIP path's bf16 rounding points
    # (operands of the convolution products and of the LSTM input projections: oracle/dccrn.py) -- what bf16 operands
    # cost by themselves, per tensor. The fp32 bounds stay absolute.
    emu = OracleDCCRN(emulate_bf16=True)
    emu.load_state_dict(oracle.state_dict())
    emu.train()
    loss_emu = osnr(emu(batch[:, 0]), batch[:, 1], lengths).mean()
    loss_emu.backward()
    g_emu = torch.cat([p.grad.reshape(-1) for p in emu.parameters()])
    floor = 1e-3*float(g_ref.norm())

    def per_tensor(g):
        out, o = {}, 0
        for n, k in zip(names, sizes):
            ref = g_ref[o:o + k]
            # (tensors of a handful of elements -- the PReLU slopes are scalars -- are sums with heavy cancellation:
            # their relative error is that of ONE rounding pattern, not an average; they count in the global bound)
            if float(ref.norm()) > floor and k >= 64:
                out[n] = rel(g[o:o + k], ref)
            o += k
        return out
    e_emu, t_emu = rel(g_emu, g_ref), per_tensor(g_emu)
    print(f'bf16-emulating oracle vs fp32 oracle: loss {float(loss_emu):.5f} vs {float(loss_ref):.5f}, gradient rel '
          f'{e_emu:.3e}, worst tensor {max((v, k) for k, v in t_emu.items())}')
    net = net.to(dev).train()
    for amp in (False, True):
        net.zero_grad(set_to_none=True)
        loss = net.loss(batch.to(dev), lengths.to(dev), amp)
        loss.backward()
        got = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu()
        assert torch.isfinite(got).all()
        e, t_hip = rel(got, g_ref), per_tensor(got)
        worst = max((v, k) for k, v in t_hip.items())
        print(f"default DCCRN 2 x 4 s, {'use_amp' if amp else 'fp32'}: loss {float(loss):.5f} vs {float(loss_ref):.5f}, "
              f'gradient rel {e:.3e}, worst tensor {worst}')
        if not amp:
            assert abs(float(loss.detach()) - float(loss_ref.detach())) <= 1e-3, (float(loss), float(loss_ref))
            assert e <= 2e-3, e
            assert worst[0] <= 2e-2, worst
            continue
        # use_amp: no further from the fp32 oracle than 1.5 x the emulation (+ 1e-3), globally and for EVERY tensor of
        # 64 elements or more. Measured (round 5): global 3.34e-2 against the emulation's 3.31e-2; the tensor that sat
        # at 0.20 / 0.12 in round 4 (encoder.5 conv weights) is at 0.1247 against the emulation's 0.1243: what bf16
        # operands cost there by themselves, not a kernel defect; the largest HIP / emulation ratio of any tensor is 1.11
        assert abs(float(loss.detach()) - float(loss_ref.detach())) <= \
            1.5*abs(float(loss_emu.detach()) - float(loss_ref.detach())) + 1e-3, (float(loss), float(loss_emu), float(loss_ref))
        ratios = sorted(((t_hip[k]/(1.5*t_emu[k] + 1e-3), k, t_hip[k], t_emu[k]) for k in t_hip), reverse=True)
        print('   largest HIP / (1.5 emulation + 1e-3) ratios:', [(round(r, 3), k, round(h, 4), round(e_, 4)) for r, k, h, e_ in ratios[:6]])
        assert e <= 1.5*e_emu + 1e-3, (e, e_emu)
        assert ratios[0][0] <= 1.0, ratios[:5]


def test_convtasnet_bf16_ten_step_trajectory_at_default_widths(monkeypatch):
    from brever_amd.models import ConvTasNet
    from oracle.convtasnet import OracleConvTasNet
    dev = _cuda()
    cfg = dict(layers=8, repeats=1)
    B, L, steps = 4, 16000, 10
    gen = torch.Generator().manual_seed(31)
    batches = []
    for _ in range(steps):
        clean = 0.1*torch.randn(B, 1, L, generator=gen)
        batches.append(torch.cat([clean + 0.1*torch.randn(B, 1, L, generator=gen), clean], dim=1))
    lengths = torch.full((B,), L)
    torch.manual_seed(5)
    ref = OracleConvTasNet(**cfg)
    state = {k: v.clone() for k, v in ref.state_dict().items()}
    scaler = torch.amp.GradScaler('cuda', enabled=False)
    torch.set_num_threads(max(1, min(32, torch.get_num_threads())))

    def oracle_run(amp):
        m = OracleConvTasNet(**cfg)
        m.load_state_dict(state)
        return [float(m.train_step(b, lengths, amp, scaler)) for b in batches]
    loss_fp32 = oracle_run(False)
    loss_cpu_bf16 = oracle_run(True)
    net = ConvTasNet(**cfg)
    net.load_state_dict(state)
    net = net.to(dev)
    monkeypatch.setenv('BRV_CTN_STREAMS', '1')
    loss_hip = [float(net.train_step(b.to(dev), lengths.to(dev), True, scaler)) for b in batches]
    d32 = np.abs(np.array(loss_hip) - np.array(loss_fp32))
    d16 = np.abs(np.array(loss_hip) - np.array(loss_cpu_bf16))
    dcpu = np.abs(np.array(loss_cpu_bf16) - np.array(loss_fp32))
    print('bf16 Conv-TasNet, default widths, 8 blocks, 10 steps (Adam 1e-3, clip 5): loss (dB)')
    print('   HIP bf16     ', ' '.join('%.4f' % v for v in loss_hip))
    print('   CPU fp32     ', ' '.join('%.4f' % v for v in loss_fp32))
    print('   CPU bf16 amp ', ' '.join('%.4f' % v for v in loss_cpu_bf16))
    print('   max |HIP - fp32| %.3e, max |HIP - CPU bf16| %.3e, max |CPU bf16 - fp32| %.3e'
          % (d32.max(), d16.max(), dcpu.max()))
    assert np.isfinite(loss_hip).all()
    # the drift of the HIP bf16 run from the fp32 trajectory stays within a small multiple of what the
    # reference's own CPU-bf16 autocast run shows, and small in absolute terms
    assert d32.max() <= max(3.0*dcpu.max(), 2e-2), (d32.max(), dcpu.max())
    assert d32.max() <= 5e-2, d32.max()


@pytest.mark.parametrize('arch,kwargs,clip', [
    ('dccrn', {}, 5.0), ('ffnn', {}, 0.0), ('tfgridnet', dict(n_layers=2, lstm_hidden_units=32, attn_n_head=2), 1.0),
    ('sgmsepm', {}, 0.0)])
def test_fused_clip_adam_of_every_model_matches_torch(arch, kwargs, clip):
    """K11 for every model: DCCRN / FFNN / TF-GridNet / SGMSE+ step through
    `brv_clip_adam_step2` on ONE flat parameter buffer (models/base.py) instead of `clip_grad_norm_` +
    `torch.optim.Adam` (brever/models/base.py:296-301). Three steps on fabricated gradients against torch's own
    clip + Adam on copies; the optimizer `state_dict` keeps torch's format both ways."""
    from brever_amd.models import ModelRegistry
    from brever_amd.optim import FlatAdam
    dev = _cuda()
    torch.manual_seed(0)
    model = ModelRegistry.get(arch)(**kwargs).to(dev)
    assert isinstance(model.optimizer, FlatAdam)
    params = list(model.parameters())
    # every parameter is a view of the flat buffer, in parameters() order, also after .to(device)
    off = 0
    for p in params:
        assert p.data_ptr() == model.flat_params().data_ptr() + 4*off
        off += p.numel()
    ref_params = [p.detach().clone().requires_grad_(True) for p in params]
    ref_opt = torch.optim.Adam(ref_params, lr=model.optimizer.param_groups[0]['lr'])
    gen = torch.Generator(device='cpu').manual_seed(1)
    scaler = torch.amp.GradScaler('cuda', enabled=False)
    for step in range(3):
        grads = [(0.3*torch.randn(p.shape, generator=gen)).to(dev) for p in params]
        for p, q, g in zip(params, ref_params, grads):
            p.grad = g.clone()
            q.grad = g.clone()
        if clip:
            torch.nn.utils.clip_grad_norm_(ref_params, clip)
        ref_opt.step()
        # the model's own update path (autograd leaves separate .grad tensors: gathered by one multi-tensor copy)
        loss = sum((p*0).sum() for p in params[:1])         # a graph whose backward adds zeros to params[0].grad
        model.update(loss, scaler, grad_clip=clip) if arch in ('ffnn', 'sgmsepm') else model.update(loss, scaler)
    for p, q in zip(params, ref_params):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7), float((p - q).abs().max())
    # state_dict: torch's format, loadable by torch.optim.Adam and back
    sd = model.optimizer.state_dict()
    other = torch.optim.Adam([q.detach().clone().requires_grad_(True) for q in ref_params])
    other.load_state_dict(sd)
    ref_sd = ref_opt.state_dict()
    for k in range(len(params)):
        a, b = sd['state'][k], ref_sd['state'][k]
        assert float(a['step']) == float(b['step']) == 3
        assert torch.allclose(a['exp_avg'], b['exp_avg'], rtol=1e-4, atol=1e-6)
        assert torch.allclose(a['exp_avg_sq'], b['exp_avg_sq'], rtol=1e-4, atol=1e-10)     # (fused multiply-adds)
    model.optimizer.load_state_dict(ref_sd)
    assert torch.allclose(model.optimizer._exp_avg[:params[0].numel()].view(params[0].shape), ref_sd['state'][0]['exp_avg'])


def test_fp32_trajectory_20_steps_at_the_default_depth():
    """VERDICT r05 item 6a / SURVEY 8(d)(ii) / north_star "loss curves matching reference to 1e-3": the 20-step
    protocol of tests/test_gpu.py::test_fp32_trajectory_20_steps on the DEFAULT network -- 8 layers x 3 repeats =
    24 blocks, dilations 1 .. 128, 4 935 217 parameters (that test runs 4 blocks) -- with 2 x 1 s ragged items so
    that the CPU fp32 AND fp64 oracles stay within a few minutes. Fused HIP steps (fp32 path, Adam lr 1e-3, clip
    5) against the CPU fp32 oracle from identical init and data order, reference convtasnet.py:78-89.

    The fp64-anchored bound of that test: against the fp64 trajectory (which both fp32 runs approximate) the HIP run
    may be no further away than 3 x the CPU fp32 run's own worst distance so far (+ 1e-4), at every step; against the
    CPU fp32 oracle |d loss| <= 1e-3 dB over the first three steps and no more than 1.5 x the oracle's own distance
    to its fp64 run afterwards (see the comment at the assertions for why not 1e-3 / 3e-3 throughout)."""
    from brever_amd.models import ConvTasNet
    from oracle.convtasnet import OracleConvTasNet
    torch.manual_seed(7)
    oracle = OracleConvTasNet()
    assert sum(p.numel() for p in oracle.parameters()) == 4935217
    o64 = OracleConvTasNet().double()
    o64.load_state_dict({k: v.double() for k, v in oracle.state_dict().items()})
    net = ConvTasNet()
    net.load_state_dict(oracle.state_dict())
    net = net.to(_cuda())
    scaler = torch.amp.GradScaler('cuda', enabled=False)
    gen = torch.Generator().manual_seed(11)
    L = 16000
    d32, d64, c64 = [], [], []
    for step in range(20):
        clean = 0.1*torch.randn(2, L, generator=gen)
        noise = 0.1*torch.randn(2, L, generator=gen)
        snr_db = -5 + 15*torch.rand(2, 1, generator=gen)
        batch = torch.stack([clean + 10**(-snr_db/20)*noise, clean], dim=1)
        lengths = torch.tensor([L, L - 3500])
        batch[1, :, lengths[1]:] = 0
        want = float(oracle.train_step(batch, lengths, False, scaler).detach())
        truth = float(o64.train_step(batch.double(), lengths, False, scaler).detach())
        got = float(net.train_step(batch.cuda(), lengths.cuda(), False, scaler))
        d32.append(abs(got - want)); d64.append(abs(got - truth)); c64.append(abs(want - truth))
    print('default-depth fp32 trajectory |hip - cpu32|:', ['%.1e' % d for d in d32])
    print('                            |cpu32 - cpu64|:', ['%.1e' % d for d in c64])
    print('                              |hip - cpu64|:', ['%.1e' % d for d in d64])
    # Measured on MI355X (round 6): at this depth the CPU fp32 oracle ITSELF leaves its fp64 run by 1.9e-3 at step 3
    # and 1.2e-2 at step 10 (24 blocks amplify a rounding difference far more than the 4 blocks of the other test:
    # 7e-4 there), the HIP run stays within 8.4e-3 of the fp64 run and is closer to it than the CPU fp32 run at 17 of
    # the 20 steps. The absolute 1e-3 / 3e-3 of the 4-block test are therefore asserted where fp32 arithmetic can hold
    # them -- the first three steps -- and beyond that against the oracle's own distance to the truth:
    assert max(d32[:3]) <= 1e-3, d32
    assert max(d32[:10]) <= max(1e-3, 1.5*max(c64[:10])), (d32, c64)
    assert max(d32) <= max(3e-3, 1.5*max(c64)), (d32, c64)
    assert max(d64) <= 3*max(c64) + 1e-4, (d64, c64)
    for i in range(20):                                  # ... step by step, not only at the worst step
        assert d64[i] <= 3*max(c64[:i + 1]) + 1e-4, (i, d64, c64)
    ref = torch.cat([p.detach().reshape(-1) for p in oracle.parameters()])
    assert rel(net.flat_params(), ref) <= 1e-2, rel(net.flat_params(), ref)
