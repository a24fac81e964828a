"""Parity at the shapes that are benchmarked (VERDICT r02 "Next round" item 1).

The small golden configurations take the generic tile kernels; the kernels `bench.py` times are the
default-width ones (persistent weight-stationary GEMMs, the fused forward stage, the comb-tile
depthwise backward whose geometry changes with the dilation, the grouped weight gradients, the
two-chain step). Here they run at dilations 1 .. 128, with 8 and 24 blocks, B = 8 (two chains) and
B = 3 (one chain), ragged lengths, and ONCE at the full BASELINE size (24 blocks, 16 x 64 000), and
every gradient is compared ELEMENTWISE with the CPU oracle:

* vs the bf16-emulating oracle (forward rounding points of the fused forward): global rel-L2 <= 6e-2,
  every tensor with >= 512 elements <= 0.15;
* vs the fp32 oracle (= the reference arithmetic): global rel-L2 <= 8e-2 (bf16 path) / 1e-4 (fp32 path).

Also here: the 16-row-tile variant (PF = 4) of the channels-last 3 x 3 convolution, default-size DCCRN
and the default SGMSE+ score network on a 256 x 501 spectrogram (BASELINE configs 3 and 4).
"""
import math

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


def _detrivialise(model, gen):
    """Norm gains / biases and PReLU slopes away from their (1, 0, 0.25) initial values."""
    with torch.no_grad():
        for name, p in model.named_parameters():
            if 'norm' in name or 'prelu' in name:
                p.add_(0.1*torch.randn(p.shape, generator=gen))


def _ragged_batch(gen, B, L, scale=0.3):
    batch = scale*torch.randn(B, 2, L, generator=gen)
    cut = [0, 300, 1111, 17, 2048, 5000, 1, 777, 4096, 33, 9000, 64, 2, 12345, 600, 31]
    lengths = torch.tensor([max(L - cut[b % len(cut)], L//2) for b in range(B)])
    for b in range(B):
        batch[b, :, lengths[b]:] = 0
    return batch, lengths


def _oracle_grads(oracle, batch, lengths):
    for p in oracle.parameters():
        p.grad = None
    out = oracle(batch[:, 0])
    loss = oracle.criterion(out, batch[:, 1:], lengths).mean()
    loss.backward()
    return out.detach(), float(loss), torch.cat([p.grad.reshape(-1) for p in oracle.parameters()])


def _per_tensor(net, got, want, bound, min_numel=512, min_norm=1e-4):
    worst = (-1.0, '')
    for (name, p), off in zip(net.named_parameters(), [o for _, o in net.param_offsets()]):
        n = p.numel()
        ref = want[off:off + n]
        if n >= min_numel and float(ref.norm()) > min_norm:
            e = rel(got[off:off + n], ref)
            worst = max(worst, (e, name))
            assert e <= bound, (name, e)
    return worst


def _fused_step_grads(net, batch, lengths, amp, streams, monkeypatch):
    """Unclipped flat gradient the fused ``train_step`` produced (grad_clip 0: the clip + Adam
    kernel then leaves the gradient buffer as the backward pass wrote it)."""
    monkeypatch.setenv('BRV_CTN_STREAMS', streams)
    net.grad_clip = 0.0
    scaler = torch.amp.GradScaler('cuda', enabled=False)
    loss = float(net.train_step(batch.cuda(), lengths.cuda(), amp, scaler))
    torch.cuda.synchronize()
    return loss, net.flat_grads().detach().clone().cpu()


@pytest.mark.parametrize('repeats,B', [(1, 8), (1, 3), (3, 8), (3, 3)])
def test_default_width_gradients_at_all_dilations(monkeypatch, repeats, B):
    """Default widths, 8 layers (dilations 1 .. 128) x {1, 3} repeats, L = 16 000 (T = 999 frames:
    tile tails at every dilation), ragged. B = 8 runs the two-chain step, B = 3 one chain. The FUSED
    train_step buffer and the autograd path, elementwise vs both oracles."""
    from brever_amd.criterion import snr
    from brever_amd.models import ConvTasNet
    from oracle.convtasnet import OracleConvTasNet
    L = 16000
    cfg = dict(layers=8, repeats=repeats)
    gen = torch.Generator().manual_seed(100 + 10*repeats + B)
    torch.manual_seed(17)
    emu = OracleConvTasNet(**cfg, emulate_bf16='fused')
    _detrivialise(emu, gen)
    ref = OracleConvTasNet(**cfg)
    ref.load_state_dict(emu.state_dict())
    batch, lengths = _ragged_batch(gen, B, L)
    out_emu, loss_emu, g_emu = _oracle_grads(emu, batch, lengths)
    out_ref, loss_ref, g_ref = _oracle_grads(ref, batch, lengths)

    net = ConvTasNet(**cfg)
    net.load_state_dict(emu.state_dict())
    net = net.to(_cuda())
    # autograd path (one chain, whole batch)
    net._amp = True
    out = net(batch[:, 0].cuda())
    assert rel(out, out_emu) <= 1e-2, rel(out, out_emu)
    loss = snr(out, batch[:, 1:].cuda(), lengths.cuda()).mean()
    assert abs(float(loss) - loss_emu) <= 2e-3
    loss.backward()
    g_auto = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu()
    net.zero_grad(set_to_none=True)
    # fused step (two chains for B = 8)
    loss_f, g_fused = _fused_step_grads(net, batch, lengths, True, '2', monkeypatch)
    assert abs(loss_f - loss_emu) <= 2e-3, (loss_f, loss_emu)
    for tag, got in (('autograd', g_auto), ('fused', g_fused)):
        e_emu, e_ref = rel(got, g_emu), rel(got, g_ref)
        print(f'repeats {repeats} B {B} {tag}: grad rel vs bf16-emulating oracle {e_emu:.3e}, '
              f'vs fp32 oracle {e_ref:.3e} (oracle-emu vs fp32 {rel(g_emu, g_ref):.3e})')
        assert e_emu <= 6e-2, (tag, e_emu)
        assert e_ref <= 8e-2, (tag, e_ref)
        worst = _per_tensor(net, got, g_emu, 0.15)
        print('   worst tensor', worst)
    # both entry paths computed the same thing (summation order / chain split only)
    assert rel(g_fused, g_auto) <= 2e-2, rel(g_fused, g_auto)


def test_default_architecture_gradients_elementwise(golden_dir):
    """The reference golden of the default architecture (seed 0 init, 1 x 4000) only stores gradient
    NORMS; the elementwise check runs against the oracle at the same weights (pinned to that golden by
    tests/test_oracle.py): every tensor, not a 25 % norm band."""
    import os

    from brever_amd.criterion import snr
    from brever_amd.models import ConvTasNet
    from oracle.convtasnet import OracleConvTasNet
    g = np.load(os.path.join(golden_dir, 'convtasnet_default.npz'))
    torch.manual_seed(0)
    ref = OracleConvTasNet()
    net = ConvTasNet()
    net.load_state_dict(ref.state_dict())
    net = net.to(_cuda())
    net._amp = True
    batch = torch.from_numpy(g['batch'])
    lengths = torch.from_numpy(g['lengths'])
    _, loss_ref, g_ref = _oracle_grads(ref, batch, lengths)
    assert abs(loss_ref - float(g['loss'])) <= 1e-5
    out = net(batch[:, 0].cuda())
    loss = snr(out, batch[:, 1:].cuda(), lengths.cuda()).mean()
    loss.backward()
    got = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu()
    assert rel(got, g_ref) <= 8e-2, rel(got, g_ref)
    # at the seed-0 init the gradient is dominated by the last blocks; tensors below 1e-3 of the
    # largest norm are bf16 noise
    gn = torch.from_numpy(g['grad_norms'])
    floor = 1e-3*float(gn.max())
    worst = _per_tensor(net, got, g_ref, 0.2, min_numel=128, min_norm=floor)
    print('default architecture: global', rel(got, g_ref), 'worst tensor', worst)


def test_full_size_gradients_vs_oracle(monkeypatch):
    """ONE full-size check: 24 blocks, 16 x 64 000, ragged. CPU fp32 oracle forward + backward once
    (~45 s of host time on the GPU box), then (a) the bf16 two-chain fused step and (b) the fp32 path,
    all gradients elementwise."""
    from brever_amd.models import ConvTasNet
    from oracle.convtasnet import OracleConvTasNet
    B, L = 16, 64000
    gen = torch.Generator().manual_seed(77)
    torch.manual_seed(0)
    ref = OracleConvTasNet()
    _detrivialise(ref, gen)
    batch, lengths = _ragged_batch(gen, B, L, scale=0.1)
    lengths[0] = L
    batch[0] = 0.1*torch.randn(2, L, generator=gen)
    torch.set_num_threads(max(1, min(64, torch.get_num_threads())))
    _, loss_ref, g_ref = _oracle_grads(ref, batch, lengths)

    net = ConvTasNet()
    net.load_state_dict(ref.state_dict())
    net = net.to(_cuda())
    loss16, g16 = _fused_step_grads(net, batch, lengths, True, '2', monkeypatch)
    assert abs(loss16 - loss_ref) <= 1e-2, (loss16, loss_ref)
    e16 = rel(g16, g_ref)
    print(f'full size bf16 two-chain: loss {loss16:.5f} vs {loss_ref:.5f}, grad rel {e16:.3e}')
    assert e16 <= 1.5e-2, e16                   # 2 x the measured 7.4e-3: a regression of one kernel shows
    worst = _per_tensor(net, g16, g_ref, 0.25)
    print('   worst tensor', worst)
    del net
    torch.cuda.empty_cache()
    # fp32 path from the same weights
    net = ConvTasNet()
    net.load_state_dict(ref.state_dict())
    net = net.to(_cuda())
    loss32, g32 = _fused_step_grads(net, batch, lengths, False, '1', monkeypatch)
    assert abs(loss32 - loss_ref) <= 1e-4, (loss32, loss_ref)
    e32 = rel(g32, g_ref)
    print(f'full size fp32: loss {loss32:.6f} vs {loss_ref:.6f}, grad rel {e32:.3e}')
    assert e32 <= 1e-4, e32
    _per_tensor(net, g32, g_ref, 2e-3)


@pytest.mark.parametrize('c1,c2,co,H,W,B', [(128, 128, 128, 128, 256, 3), (128, 0, 128, 128, 384, 2),
                                            (256, 0, 256, 130, 250, 2)])
def test_conv_nhwc_sixteen_row_tiles(c1, c2, co, H, W, B):
    """`conv_nhwc_kernel<3, FOLD, PF = 4>` (16-row tiles: taken at >= 192 tiles, i.e. by every 256 x 501
    evaluation of BASELINE config 4 and 45 % of the SGMSE+ batch-8 time): plain (FOLD 0), precomputed
    GroupNorm fold (FOLD 1), fold in the kernel's prologue (FOLD 2), concatenated input, residual,
    output scale, and the statistics epilogue, vs torch CPU fp32."""
    import torch.nn.functional as F

    from brever_amd.models import sgmse as M
    dev = _cuda()
    n_cob = (co + 127)//128
    assert B*((H + 15)//16)*((W + 31)//32)*n_cob >= 192       # the launcher picks PF = 4
    g = torch.Generator().manual_seed(c1 + c2 + H)
    ci = c1 + c2
    conv = torch.nn.Conv2d(ci, co, 3, 1, 1)
    gn = M.GroupNorm(ci)
    with torch.no_grad():
        gn.weight.add_(0.2*torch.randn(ci, generator=g))
        gn.bias.add_(0.2*torch.randn(ci, generator=g))
    xx = torch.randn(B, ci, H, W, generator=g) + 0.5
    ee = torch.randn(B, ci, generator=g)
    rr = torch.randn(B, co, H, W, generator=g)
    xh, rh = xx.half().float(), rr.half().float()
    with torch.no_grad():
        ref = 0.7*(conv(F.silu(gn(xh + ee[:, :, None, None]))) + rh)
        ref_plain = conv(xh)
    conv, gn = conv.to(dev), gn.to(dev)
    act = M._h_from_nchw(xx[:, :c1].to(dev))
    if c2:
        act = M._Act(act.t, act.C, second=M._h_from_nchw(xx[:, c1:].to(dev)))
    res = M._h_from_nchw(rr.to(dev))
    # FOLD 0
    plain = M._h_conv3(act, conv)
    assert rel(M._h_to_nchw(plain), ref_plain) <= 2e-3, rel(M._h_to_nchw(plain), ref_plain)
    # statistics epilogue: per-(item, channel) sum and sum of squares of the STORED fp16 values
    stored = M._h_to_nchw(plain).double()
    sums = plain.sums.cpu()
    assert torch.allclose(sums[..., 0], stored.sum((2, 3)).cpu(), rtol=1e-6, atol=1e-3)
    assert torch.allclose(sums[..., 1], stored.pow(2).sum((2, 3)).cpu(), rtol=1e-6, atol=1e-3)
    # FOLD 1: (scale, shift) precomputed
    fold = M._h_gn_fold(act, gn, add=ee.to(dev))
    got1 = M._h_conv3(act, conv, fold=fold, silu=True, res=res, out_scale=0.7)
    assert rel(M._h_to_nchw(got1), ref) <= 2e-3, rel(M._h_to_nchw(got1), ref)
    # FOLD 2: fold in the convolution's prologue from the per-channel sums
    got2 = M._h_conv3(act, conv, norm=gn, add=ee.to(dev), silu=True, res=res, out_scale=0.7)
    assert rel(M._h_to_nchw(got2), ref) <= 2e-3, rel(M._h_to_nchw(got2), ref)
    assert rel(M._h_to_nchw(got2), M._h_to_nchw(got1)) <= 1e-3


def test_dccrn_default_size_matches_oracle():
    """BASELINE config 3 at its own size: default DCCRN (3.67 M parameters), 2 x 4 s, forward in train
    mode + snr loss, fp32 and use_amp, vs oracle/dccrn.py at the same seeded weights."""
    from brever_amd.models import DCCRN
    from oracle.criterion import snr as osnr
    from oracle.dccrn import OracleDCCRN
    dev = _cuda()
    torch.manual_seed(4)
    net = DCCRN()
    oracle = OracleDCCRN()
    with torch.no_grad():
        flat = torch.cat([p.reshape(-1) for p in net.parameters()])
        assert flat.numel() == sum(p.numel() for p in oracle.parameters())
        o = 0
        for p in oracle.parameters():
            p.copy_(flat[o:o + p.numel()].view_as(p))
            o += p.numel()
    gen = torch.Generator().manual_seed(9)
    batch, lengths = _ragged_batch(gen, 2, 64000, scale=0.1)
    oracle.train()
    with torch.no_grad():
        want = oracle(batch[:, 0])
        want_loss = float(osnr(want, batch[:, 1], lengths).mean())
    net = net.to(dev).train()
    with torch.no_grad():
        got = net(batch[:, 0].to(dev))
    assert got.shape == want.shape
    assert rel(got, want) <= 5e-4, rel(got, want)
    # (each loss call is a train-mode forward from the same parameters; running statistics do not
    # enter a train-mode forward)
    loss32 = float(net.loss(batch.to(dev), lengths.to(dev), False))
    assert abs(loss32 - want_loss) <= 1e-3, (loss32, want_loss)
    loss16 = float(net.loss(batch.to(dev), lengths.to(dev), True))
    assert abs(loss16 - want_loss) <= 2e-2, (loss16, want_loss)
    # train-mode output under use_amp (the module-level switch DCCRN.loss / _enhance set)
    from brever_amd.models import dccrn as dccrn_mod
    with torch.no_grad():
        dccrn_mod._AMP['on'] = True
        try:
            got16 = net(batch[:, 0].to(dev))
        finally:
            dccrn_mod._AMP['on'] = False
    e16 = rel(got16, want)
    print('default DCCRN 2 x 4 s: fp32 rel', rel(got, want), 'use_amp rel', e16)
    assert e16 <= 2e-2, e16


def test_sgmse_default_denoiser_full_spectrogram():
    """BASELINE config 4 at its own size: the default 65.6 M-parameter score network on the 256 x 501
    spectrogram of a 4 s utterance, batch 1, use_amp (channels-last fp16 path: 16-row tiles at the top
    resolutions) vs oracle/sgmse.py on the CPU."""
    from brever_amd.models import SGMSEp
    from brever_amd.models.sgmse import hip_autocast
    from oracle import sgmse as osg
    dev = _cuda()
    torch.manual_seed(1)
    model = SGMSEp()
    net = osg.Net(model.state_dict(), 'model.net.', skip_scale=0.5**0.5)
    sde = osg.RichterOUVE()
    g = torch.Generator().manual_seed(2)
    y = 0.3*torch.randn(1, 1, 256, 501, dtype=torch.complex64, generator=g)
    x = y + 0.2*torch.randn(1, 1, 256, 501, dtype=torch.complex64, generator=g)
    t = torch.tensor(0.5)
    with torch.no_grad():
        want = osg.denoise(net, sde, x, y, sde.sigma(t), t)
    model = model.to(dev).eval()
    with hip_autocast(True):
        got16 = model(x.to(dev), y.to(dev), model.sde.sigma(t), t)
    err = rel(torch.view_as_real(got16), torch.view_as_real(want))
    print('default SGMSE+ denoiser 256 x 501 use_amp rel', err)
    assert err <= 5e-3, err


def _variant_library():
    """tools/_v/variants/libbrever_hip.so: the library with the two rejected kernel organisations compiled in
    (`__graft_entry__.build()` makes it; built here when missing -- hipcc cross-compiles in ~30 s)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, 'tools', '_v', 'variants', 'libbrever_hip.so')
    csrc = os.path.join(root, 'brever_amd', 'csrc')
    newest = max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc) if f.endswith(('.hip', '.cuh', '.h')))
    newest = max(newest, os.path.getmtime(os.path.join(root, 'include', 'brever_hip.h')))
    if not os.path.exists(lib) or os.path.getmtime(lib) < newest:      # (stale: built before the last source change)
        subprocess.run(['bash', os.path.join(root, 'tools', 'mkvariant.sh'), 'variants', '-DBRV_WITH_VARIANTS'],
                       check=True, capture_output=True, timeout=900)
    return root, lib


def _variant_check(switch, layers, repeats, B, L):
    import json
    import os
    import subprocess
    import sys
    root, lib = _variant_library()
    env = dict(os.environ, BRV_LIB_PATH=lib)
    out = subprocess.run([sys.executable, os.path.join(root, 'tests', 'variant_check.py'), switch, str(layers), str(repeats),
                          str(B), str(L)], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])


def test_default_library_refuses_the_rejected_kernel_variants(monkeypatch):
    """Round 6: csrc/dwpw2_fused_v2.cuh and csrc/bwd_fused_p.cuh are compiled only with -DBRV_WITH_VARIANTS; the
    default library must fail loudly (no silent fall-through to the default kernel) when their switches are set."""
    from brever_amd.criterion import snr
    from brever_amd.models import ConvTasNet
    gen = torch.Generator().manual_seed(3)
    batch, lengths = _ragged_batch(gen, 2, 4000)
    for switch in ('BRV_DWPW2_V2', 'BRV_BWD_PERSIST'):
        monkeypatch.setenv(switch, '1')
        net = ConvTasNet(layers=2, repeats=1).to(_cuda())
        net._amp = True
        with pytest.raises(RuntimeError, match='not in this build'):
            out = net(batch[:, 0].cuda())
            snr(out, batch[:, 1:].cuda(), lengths.cuda()).mean().backward()
        monkeypatch.delenv(switch)


@pytest.mark.parametrize('layers,repeats,B,L', [(8, 1, 2, 16000), (4, 2, 1, 700), (8, 1, 11, 10500)])
def test_whole_row_fused_forward_variant_against_the_oracle(layers, repeats, B, L):
    """The whole-row organisation of the fused forward stage (csrc/dwpw2_fused_v2.cuh; variant library only, DESIGN
    5m: 64 against 50 us per launch) against the bf16-emulating ORACLE -- output, loss and every gradient of the
    step that follows, with the bounds of the default kernels -- and against the slab form of the same library.
    Items shorter than a tile / than the dilation, ragged lengths, tile counts that do not divide over the XCDs."""
    r = _variant_check('BRV_DWPW2_V2', layers, repeats, B, L)
    print(f'layers {layers} x {repeats}, B {B}, L {L}: whole-row forward', r)
    assert r['finite']
    assert r['out_vs_oracle'] <= 1e-2 and r['loss_vs_oracle'] <= 2e-3 and r['grad_vs_oracle'] <= 6e-2, r
    assert r['out_vs_default'] <= 5e-3 and r['grad_vs_default'] <= 4e-2, r


@pytest.mark.parametrize('layers,repeats,B,L', [(8, 1, 2, 16000), (8, 2, 3, 5000), (4, 2, 1, 700), (8, 1, 5, 2100),
                                                (3, 3, 4, 20000), (8, 1, 2, 300), (8, 1, 1, 64000)])
def test_fused_backward_equals_three_launch_backward(monkeypatch, layers, repeats, B, L):
    """The backward mirror of the fused forward (csrc/bwd_fused.cuh: [res | skip] data gradient + gLN_2 /
    PReLU_2 backward + transposed depthwise stencil in one launch, layer-norm means from <g, u>) against
    the three-launch sequence (BRV_BWD_FUSE=0) from the same forward: every gradient, elementwise. Items
    shorter than the dilation / than a tile, one item, ragged lengths, the block without residual conv."""
    from brever_amd.criterion import snr
    from brever_amd.models import ConvTasNet
    cfg = dict(layers=layers, repeats=repeats)
    gen = torch.Generator().manual_seed(7*layers + B)
    torch.manual_seed(23)
    ref = ConvTasNet(**cfg)
    _detrivialise(ref, gen)
    batch, lengths = _ragged_batch(gen, B, L)
    grads = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('BRV_BWD_FUSE', mode)
        net = ConvTasNet(**cfg)
        net.load_state_dict(ref.state_dict())
        net = net.to(_cuda())
        net._amp = True
        out = net(batch[:, 0].cuda())
        loss = snr(out, batch[:, 1:].cuda(), lengths.cuda()).mean()
        loss.backward()
        grads[mode] = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu()
        assert torch.isfinite(grads[mode]).all()
        if mode == '1':
            e = rel(grads['1'], grads['0'])
            print(f'layers {layers} x {repeats}, B {B}, L {L}: fused vs three-launch backward rel {e:.3e}')
            assert e <= 1.5e-2, e
            worst = _per_tensor(net, grads['1'], grads['0'], 5e-2, min_numel=128, min_norm=1e-5)
            print('   worst tensor', worst)


@pytest.mark.parametrize('layers,repeats,B,L', [(8, 1, 2, 16000), (8, 1, 9, 2100), (8, 1, 2, 300)])
def test_persistent_fused_backward_variant_against_the_oracle(layers, repeats, B, L):
    """csrc/bwd_fused_p.cuh (variant library only, DESIGN 5n: 83 against 73 us per launch; a workgroup walks two
    tiles, per-channel partial sums kept in LDS over the range, atomics once per workgroup) against the
    bf16-emulating ORACLE with the default kernel's bounds, and against the one-tile-per-workgroup kernel of the same
    library (same arithmetic per element, per-channel sums in another order). Odd tile counts, items shorter than
    a tile, tile ranges that cross items."""
    r = _variant_check('BRV_BWD_PERSIST', layers, repeats, B, L)
    print(f'layers {layers} x {repeats}, B {B}, L {L}: persistent backward', r)
    assert r['finite']
    assert r['loss_vs_oracle'] <= 2e-3 and r['grad_vs_oracle'] <= 6e-2, r
    assert r['grad_vs_default'] <= 2e-3, r


@pytest.mark.parametrize('layers,repeats,B,L', [(8, 1, 2, 16000), (8, 2, 3, 5000), (4, 2, 1, 700), (8, 1, 5, 2100),
                                                (8, 1, 2, 300), (2, 1, 1, 64000)])
def test_first_conv_backward_with_recomputed_z1_equals_stored_path(monkeypatch, layers, repeats, B, L):
    """csrc/pw1_bwd.cuh: the first 1x1 convolution's data gradient rebuilds z1 = W1 x + b1 on the matrix pipe
    instead of reading the stored z1 (default: the weight-stationary kernel with specialised waves; it still
    stores dz1 for the weight gradient); BRV_PW1_RC_WGRAD=1: the weight gradient rebuilds dz1 too and dz1 is
    never stored; BRV_PW1_RC_TILES=1: both in their one-tile-per-workgroup form. The recompute repeats
    pw1_fwd's instruction sequence, so z1 is the same bf16 tensor; the first form also keeps the round-3
    arithmetic of dz1 (gradients equal to the order of fp32 partial sums, 1e-7), the default kernels fold it
    into fewer fused multiply-adds (dz1 differs in the last bf16 bit of some elements: 1e-3 per tensor)."""
    from brever_amd.criterion import snr
    from brever_amd.models import ConvTasNet
    cfg = dict(layers=layers, repeats=repeats)
    gen = torch.Generator().manual_seed(11*layers + B)
    torch.manual_seed(29)
    ref = ConvTasNet(**cfg)
    _detrivialise(ref, gen)
    batch, lengths = _ragged_batch(gen, B, L)
    modes = {'stored': {'BRV_PW1_RC': '0'}, 'ws': {}, 'ws+wgrad': {'BRV_PW1_RC_WGRAD': '1'},
             'tiles': {'BRV_PW1_RC_TILES': '1'}}
    grads = {}
    for mode, env in modes.items():
        for k in ('BRV_PW1_RC', 'BRV_PW1_RC_WGRAD', 'BRV_PW1_RC_TILES'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        net = ConvTasNet(**cfg)
        net.load_state_dict(ref.state_dict())
        net = net.to(_cuda())
        net._amp = True
        out = net(batch[:, 0].cuda())
        loss = snr(out, batch[:, 1:].cuda(), lengths.cuda()).mean()
        loss.backward()
        grads[mode] = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu()
        assert torch.isfinite(grads[mode]).all()
        if mode != 'stored':
            e = rel(grads[mode], grads['stored'])
            exact = mode == 'tiles'
            # (scalars -- the PReLU slope gradients -- are sums with heavy cancellation: compared in the global norm only)
            worst = _per_tensor(net, grads[mode], grads['stored'], 1e-3 if exact else 2e-2, min_numel=1 if exact else 128,
                                min_norm=1e-6)
            print(f'layers {layers} x {repeats}, B {B}, L {L}: {mode} vs stored-z1 backward rel {e:.3e}, worst {worst}')
            assert e <= (1e-4 if exact else 3e-3), (mode, e)


def test_two_models_step_from_two_threads():
    """SURVEY 8b "no global mutable state; re-entrant": two models stepped concurrently from two host
    threads on two streams (one taking the two-chain step with its persistent kernels at 7/8 of the CUs,
    the other the one-chain step at 8/8 -- what used to be a process-global switch) end where the same
    steps run one after the other from one thread end."""
    import threading

    from brever_amd.models import ConvTasNet
    dev = _cuda()
    cfg = dict(layers=3, repeats=2)
    gen = torch.Generator().manual_seed(5)
    data = {}
    for name, B in (('a', 8), ('b', 3)):
        batch, lengths = _ragged_batch(gen, B, 6000)
        data[name] = (batch.to(dev), lengths.to(dev))

    def make(seed):
        torch.manual_seed(seed)
        return ConvTasNet(**cfg).to(dev)

    def run(net, name, stream, out):
        batch, lengths = data[name]
        with torch.cuda.stream(stream):
            out[name] = [float(net.train_step(batch, lengths, True, None)) for _ in range(4)]
        stream.synchronize()

    # reference: one thread, one after the other
    ref_nets = {'a': make(1), 'b': make(2)}
    ref = {}
    for name in ('a', 'b'):
        run(ref_nets[name], name, torch.cuda.current_stream(dev), ref)
    torch.cuda.synchronize()
    # two threads, two streams, at the same time
    nets = {'a': make(1), 'b': make(2)}
    got = {}
    streams = {'a': torch.cuda.Stream(device=dev), 'b': torch.cuda.Stream(device=dev)}
    for s in streams.values():
        s.wait_stream(torch.cuda.current_stream(dev))
    threads = [threading.Thread(target=run, args=(nets[n], n, streams[n], got)) for n in ('a', 'b')]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    for name in ('a', 'b'):
        # (not bitwise: the statistics accumulate with atomics, and Adam's first updates lr*sign(g) turn a
        # rounding difference of a near-zero gradient into a 2e-3 step of that weight)
        assert abs(got[name][0] - ref[name][0]) <= 1e-5, (name, got[name], ref[name])
        assert max(abs(x - y) for x, y in zip(got[name], ref[name])) <= 5e-3, (name, got[name], ref[name])
        assert rel(nets[name].flat_params(), ref_nets[name].flat_params()) <= 2e-2, name


def test_two_chain_step_odd_batch(monkeypatch):
    """odd batches take the two-chain step too (5 + 4 items of a batch of 9): same
    losses and the same gradient as the one-chain step from the same weights."""
    from brever_amd.models import ConvTasNet
    cfg = dict(layers=3, repeats=2)
    gen = torch.Generator().manual_seed(31)
    batch, lengths = _ragged_batch(gen, 9, 5000)
    got = {}
    for mode in ('1', '2'):
        torch.manual_seed(5)
        net = ConvTasNet(**cfg).to(_cuda())
        loss, grads = _fused_step_grads(net, batch, lengths, True, mode, monkeypatch)
        got[mode] = (loss, grads)
    assert abs(got['1'][0] - got['2'][0]) <= 1e-5, (got['1'][0], got['2'][0])
    assert rel(got['2'][1], got['1'][1]) <= 1e-4, rel(got['2'][1], got['1'][1])


def test_clip_adam_step2_matches_torch():
    """`brv_clip_adam_step2` (two launches, no memset): clip_grad_norm_ + Adam.step of torch on g + g2
    (`brever/models/base.py:296-301`), the second buffer left zeroed, the two norm accumulators used in
    turn (the Adam kernel of one call zeroes the other call's), with and without a second buffer."""
    from brever_amd import hip
    dev = _cuda()
    n = 100_003
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3)
    p = p0.clone().to(dev)
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    scratch = torch.zeros(64, dtype=torch.uint8, device=dev)          # both accumulators zero
    norm = torch.zeros(1, device=dev)
    for step in range(1, 7):
        ga = torch.randn(n, generator=g)*(10.0 if step % 2 else 0.001)
        gb = torch.randn(n, generator=g)*(3.0 if step % 3 else 0.0005)
        two = step % 3 != 0                                           # every third call: one buffer only
        ref.grad = (ga + gb).clone() if two else ga.clone()
        total = torch.nn.utils.clip_grad_norm_([ref], 5.0)
        opt.step()
        gd, g2 = ga.clone().to(dev), gb.clone().to(dev)
        hip.check(hip.lib().brv_clip_adam_step2(
            hip.ptr(p), hip.ptr(gd), hip.ptr(g2) if two else None, hip.ptr(m), hip.ptr(v), n, 1.0, 5.0,
            1e-3, 0.9, 0.999, 1e-8, step, hip.ptr(scratch), (step - 1) % 2, hip.ptr(norm), hip.stream()),
            'brv_clip_adam_step2')
        assert abs(float(norm) - float(total)) <= 1e-4*float(total), (step, float(norm), float(total))
        assert torch.allclose(gd.cpu(), ref.grad, rtol=1e-5, atol=1e-8)
        assert torch.allclose(p.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
        if two:
            assert float(g2.abs().max()) == 0.0                      # left zeroed for the next step


@pytest.mark.gpu
@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_f32_big_tiles_all_layouts(ta, tb):
    """brv_gemm_f32 on the 256 x 128 / 128 x 256 tile kernel (csrc/gemm_f32_big.hip: 16-byte aligned
    operands, M, N >= 64) vs the float64 product: both storage orders of both operands, row and column
    tails, leading dimensions larger than the extent, batches, a reduction over operand pairs, row / column
    bias, accumulation. Shapes that would need a split reduction (long K, few tiles) stay on the 128 x 128
    kernel at this entry point (its split adds with atomics: not bitwise repeatable, so repeatability is
    asserted for the others only; the fixed-order split of gemm_f32_big is exercised through the fp32
    Conv-TasNet path: test_fp32_training_is_bitwise_repeatable). rel-L2 <= 2e-6."""
    import torch
    from brever_amd import hip
    lib = hip.lib()
    gen = torch.Generator().manual_seed(7 + 2*ta + tb)
    dev = torch.device('cuda')
    #        batch M    N    K     kbatch  bias  acc
    cases = [(1, 1000, 512, 128, 1, 'col', 0),      # 1x1 convolution: 4 row tiles (one partial) x 4 column tiles
             (1, 700, 128, 512, 1, None, 1),        # accumulate into d, N = one tile
             (2, 260, 132, 96, 1, 'row', 0),        # batch, tails in M, N, K tile (96 = 3 x 32)
             (1, 128, 512, 20000, 1, None, 1),      # weight gradient: 128 x 256 tiles, split reduction
             (1, 512, 128, 4100, 3, 'row', 0),      # reduction over 3 operand pairs, split, K tail (4100 = 128 x 32 + 4)
             (3, 64, 64, 4096, 1, None, 0),         # smallest tile shape the kernel takes
             (3, 4096, 256, 64, 1, 'col', 0),       # 192 tiles over a batch: the split-bf16 kernel (not for ta and tb)
             (2, 1300, 1100, 200, 1, 'row', 1)]     # 198 tiles with tails in M, N and the k-tile, accumulate: same
    for batch, M, N, K, kbatch, bias, acc in cases:
        lda = (M if ta else K) + 4
        ldb = (K if tb else N) + 8
        ldd = N + 3
        a = torch.randn(batch, kbatch, (K if ta else M), lda, generator=gen)
        b = torch.randn(batch, kbatch, (N if tb else K), ldb, generator=gen)
        d0 = torch.randn(batch, M, ldd, generator=gen)
        bv = torch.randn(M if bias == 'row' else N, generator=gen) if bias else None
        opa = a[..., :M].transpose(-1, -2) if ta else a[..., :K]
        opb = b[..., :K].transpose(-1, -2) if tb else b[..., :N]
        want = torch.einsum('zkmr,zkrn->zmn', opa.double(), opb.double())
        if bias == 'row':
            want = want + bv.double()[None, :, None]
        elif bias == 'col':
            want = want + bv.double()[None, None, :]
        if acc:
            want = want + d0[..., :N].double()
        ad, bd = a.to(dev), b.to(dev)
        bvd = bv.to(dev) if bias else None
        outs = []
        for rep in range(2):
            d = d0.to(dev).clone()
            hip.check(lib.brv_gemm_f32(
                hip.ptr(ad), hip.ptr(bd), hip.ptr(d), batch, M, N, K, lda, ldb, ldd,
                ad.stride(0), bd.stride(0), d.stride(0), ta, tb, kbatch, ad.stride(1), bd.stride(1),
                hip.ptr(bvd) if bias else None, (2 if bias == 'col' else acc), hip.stream()), 'brv_gemm_f32')
            torch.cuda.synchronize()
            outs.append(d.cpu())
        got = outs[0]
        if K < 4000:                    # (long reductions over few tiles: atomics, see above)
            assert torch.equal(outs[0], outs[1]), (batch, M, N, K, 'not repeatable')
        assert torch.equal(got[..., N:], d0[..., N:]), 'wrote outside the N columns'
        rel = float((got[..., :N].double() - want).norm()/want.norm())
        assert rel <= 2e-6, (batch, M, N, K, kbatch, bias, acc, rel)


@pytest.mark.gpu
@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_f32_with_workspace_splits_long_reductions_in_a_fixed_order(ta, tb):
    """brv_gemm_f32_ws: the same product with scratch of brv_gemm_f32_workspace_bytes -- long reductions over few
    tiles are split over workgroups whose partial tiles are summed in split order (every result bitwise
    repeatable, unlike the atomics of the entry point without scratch), in the split-bf16 form when a is row-major
    and b stored N x K (the weight gradients of the fp32 DCCRN / SGMSE+ convolutions: column matrix and output
    gradient both contiguous along the pixels, summed over the batch). vs float64, rel-L2 <= 2e-6."""
    import torch
    from brever_amd import hip
    lib = hip.lib()
    gen = torch.Generator().manual_seed(17 + 2*ta + tb)
    dev = torch.device('cuda')
    #        batch M    N    K      kbatch bias   acc
    cases = [(1, 64, 320, 32000, 4, None, 0),       # DCCRN enc2 weight gradient over 4 items
             (1, 256, 1280, 7952, 3, None, 0),      # deeper layer: more tiles, shorter rows
             (1, 128, 512, 20000, 1, None, 1),      # accumulate
             (1, 512, 128, 4100, 3, 'row', 0),      # K tail (4100 = 128 x 32 + 4), bias
             (2, 132, 72, 9000, 2, 'col', 0),       # batch, tails in M and N (whole 16-byte pieces: the big kernel's rule)
             (1, 1000, 512, 128, 1, 'col', 0),      # no split needed: workspace unused
             (1, 40, 36, 50000, 1, None, 0)]        # one partial tile, 1563 k-tiles
    used = 0
    for batch, M, N, K, kbatch, bias, acc in cases:
        lda = (M if ta else K) + 4
        ldb = (K if tb else N) + 8
        ldd = N + 3
        a = torch.randn(batch, kbatch, (K if ta else M), lda, generator=gen)
        b = torch.randn(batch, kbatch, (N if tb else K), ldb, generator=gen)
        d0 = torch.randn(batch, M, ldd, generator=gen)
        bv = torch.randn(M if bias == 'row' else N, generator=gen) if bias else None
        opa = a[..., :M].transpose(-1, -2) if ta else a[..., :K]
        opb = b[..., :K].transpose(-1, -2) if tb else b[..., :N]
        want = torch.einsum('zkmr,zkrn->zmn', opa.double(), opb.double())
        if bias == 'row':
            want = want + bv.double()[None, :, None]
        elif bias == 'col':
            want = want + bv.double()[None, None, :]
        if acc:
            want = want + d0[..., :N].double()
        ad, bd = a.to(dev), b.to(dev)
        bvd = bv.to(dev) if bias else None
        nbytes = lib.brv_gemm_f32_workspace_bytes(batch, M, N, K, ta, tb, kbatch)
        assert nbytes >= 0
        used += nbytes > 0
        # poisoned scratch with a guard behind it: partial tiles are written before they are read, and only inside
        ws = torch.full((nbytes//4 + 64,), float('nan'), device=dev)
        outs = []
        for rep in range(2):
            d = d0.to(dev).clone()
            hip.check(lib.brv_gemm_f32_ws(
                hip.ptr(ad), hip.ptr(bd), hip.ptr(d), batch, M, N, K, lda, ldb, ldd,
                ad.stride(0), bd.stride(0), d.stride(0), ta, tb, kbatch, ad.stride(1), bd.stride(1),
                hip.ptr(bvd) if bias else None, (2 if bias == 'col' else acc),
                hip.ptr(ws) if nbytes else None, nbytes, hip.stream()), 'brv_gemm_f32_ws')
            torch.cuda.synchronize()
            outs.append(d.cpu())
        got = outs[0]
        if nbytes or K < 4000:
            assert torch.equal(outs[0], outs[1]), (batch, M, N, K, 'not repeatable')
        assert bool(torch.isnan(ws[nbytes//4:]).all()), 'wrote behind the scratch'
        assert torch.equal(got[..., N:], d0[..., N:]), 'wrote outside the N columns'
        rel = float((got[..., :N].double() - want).norm()/want.norm())
        assert rel <= 2e-6, (batch, M, N, K, kbatch, bias, acc, rel)
    assert used >= 3


@pytest.mark.gpu
def test_fp32_training_is_bitwise_repeatable():
    """the fp32 path sums in fixed orders (split reductions of the weight gradients added
    in split order, per-channel and per-frame sums by slice, no floating-point atomics on a result that
    depends on their order of arrival): the same 6 training steps from the same initial weights give
    bit-identical parameters and losses, default channel widths, ragged batch."""
    import torch
    from brever_amd.models import ConvTasNet
    dev = torch.device('cuda')

    def run():
        torch.manual_seed(3)
        net = ConvTasNet(layers=4, repeats=1).to(dev)
        scaler = torch.amp.GradScaler('cuda', enabled=False)
        gen = torch.Generator().manual_seed(5)
        losses = []
        for _ in range(6):
            batch = (0.1*torch.randn(3, 2, 9000, generator=gen)).to(dev)
            lengths = torch.tensor([9000, 7777, 5003], device=dev)
            for b in range(3):
                batch[b, :, int(lengths[b]):] = 0
            losses.append(float(net.train_step(batch, lengths, False, scaler)))
        return losses, net.flat_params().detach().cpu().clone()

    l0, p0 = run()
    l1, p1 = run()
    assert l0 == l1, (l0, l1)
    assert torch.equal(p0, p1), float((p0 - p1).abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize('causal,kernel_size,widths,L', [
    (False, 3, (64, 32, 72, 36), 2500), (True, 3, (64, 32, 72, 36), 2500), (False, 5, (48, 24, 40, 16), 2500),
    (True, 5, (48, 24, 40, 16), 2500), (False, 3, (128, 128, 320, 128), 2500),
    (False, 3, (128, 128, 320, 136), 60000)])
def test_fp32_fused_kernels_causal_and_tap_counts(causal, kernel_size, widths, L):
    """The fused streaming kernels and operand transforms of the fp32 path (csrc/ctn_f32_fused.cuh: taken
    when every channel count is a multiple of 4) on what the reference goldens do not reach: cumulative
    layer norms (causal: per-frame tables, one-sided stencil), 5 taps (the 7-tap instantiation), channel
    counts that leave lanes idle (72 = 18 float4) or need two groups per lane (320), the [res | skip]
    product with a column split (128 | 128); and, with 3 x 60 000 samples (11 247 frames), the split-bf16
    kernels on partial tiles -- 320 and 136 columns, a last row tile of 111 frames, weight gradients whose
    reduction is split over the chip. Forward, loss and every gradient vs the CPU fp32 oracle
    (reference: brever/models/convtasnet/convtasnet.py:154-281): 1e-5 / 1e-4 like the golden tests."""
    from brever_amd.models import ConvTasNet
    from oracle.convtasnet import OracleConvTasNet
    N, Bn, H, Sc = widths
    cfg = dict(filters=N, filter_length=16, bottleneck_channels=Bn, hidden_channels=H, skip_channels=Sc,
               kernel_size=kernel_size, layers=3, repeats=2, causal=causal)
    torch.manual_seed(21)
    gen = torch.Generator().manual_seed(4)
    oracle = OracleConvTasNet(**cfg)
    _detrivialise(oracle, gen)
    net = ConvTasNet(**cfg)
    net.load_state_dict(oracle.state_dict())
    net = net.cuda()
    batch, lengths = _ragged_batch(gen, 3, L)
    want_out, want_loss, want = _oracle_grads(oracle, batch, lengths)
    net.zero_grad(set_to_none=True)
    out = net(batch[:, 0].cuda())
    loss = net.criterion(out, batch[:, 1:].cuda(), lengths.cuda()).mean()
    loss.backward()
    got = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu()
    assert rel(out.detach().cpu(), want_out) <= 1e-5, rel(out.detach().cpu(), want_out)
    assert abs(float(loss) - want_loss) <= 1e-5*max(1.0, abs(want_loss))
    # Gradients against the SAME network in float64. Rounding level: the other four configurations sit
    # 4e-7 .. 9e-7 from it (the CPU fp32 oracle 2e-7). One configuration shows what a PReLU does to such a
    # comparison: ONE pre-activation within rounding of zero takes the other branch of the derivative
    # (measured: channel 311 of block 3, bias-gradient error 1.2e-5 against 4e-9 on its 319 neighbours) and
    # every gradient upstream of it moves by ~3e-4. So: the median tensor must be at rounding level, no tensor
    # and not the whole vector may be off by more than an isolated branch flip explains.
    o64 = OracleConvTasNet(**cfg).double()
    o64.load_state_dict({k: v.double() for k, v in oracle.state_dict().items()})
    _, _, want64 = _oracle_grads(o64, batch.double(), lengths)
    errs, errs_cpu, off = [], [], 0
    for name, p in net.named_parameters():
        n = p.numel()
        ref = want64[off:off + n]
        if n >= 64 and float(ref.norm()) > 1e-6:
            errs.append(rel(got[off:off + n].double(), ref))
            errs_cpu.append(rel(want[off:off + n].double(), ref))
        off += n
    errs.sort(); errs_cpu.sort()
    med, med_cpu = errs[len(errs)//2], errs_cpu[len(errs_cpu)//2]
    e_hip, e_cpu = rel(got.double(), want64), rel(want.double(), want64)
    print(f'fp32 fused path gradient vs fp64: hip {e_hip:.2e} (median tensor {med:.2e}, worst {errs[-1]:.2e}), '
          f'cpu fp32 oracle {e_cpu:.2e} (median tensor {med_cpu:.2e})')
    # (sums over 11 247 frames: the CPU oracle's own fp32 rounding reaches 6e-5 there -- the bound follows it)
    assert med <= max(5e-6, 1.5*med_cpu), (med, med_cpu)
    assert errs[-1] <= 5e-3 and e_hip <= 1e-3, (errs[-1], e_hip)


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['bf16', 'bf16_three_launch_forward', 'bf16_causal', 'fp32', 'fp32_causal'])
def test_no_path_reads_uninitialised_memory(monkeypatch, mode):
    """The workspaces and prepared operands are `torch.empty` memory. Regression (round 3): the first block's
    skip accumulation of the three-launch forward multiplied the uninitialised skip buffer by zero --
    0 x NaN = NaN whenever the caching allocator handed back freed memory that held NaNs. Every path runs
    forward + backward on freshly POISONED memory (gigabytes of NaN allocated and freed right before the
    buffers are created) and must give finite results equal to its own run on ordinary memory."""
    from brever_amd.criterion import snr
    from brever_amd.models import ConvTasNet
    dev = torch.device('cuda')
    amp = mode.startswith('bf16')
    if mode == 'bf16_three_launch_forward':
        monkeypatch.setenv('BRV_FWD_FUSE', '0')
    cfg = dict(layers=3, repeats=2, causal=mode.endswith('causal'))
    gen = torch.Generator().manual_seed(17)
    batch, lengths = _ragged_batch(gen, 5, 6000)
    batch, lengths = batch.cuda(), lengths.cuda()

    def run(poison):
        torch.manual_seed(2)
        net = ConvTasNet(**cfg).to(dev)
        net._amp = amp
        if poison:
            torch.cuda.empty_cache()
            junk = [torch.full((1 << 27,), float('nan'), device=dev) for _ in range(12)]     # 6 GB of NaN
            del junk                                      # back to the caching allocator, contents intact
        out = net(batch[:, 0])
        loss = snr(out, batch[:, 1:], lengths).mean()
        loss.backward()
        grads = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
        return out.detach().clone(), grads.detach().clone()

    out0, g0 = run(False)
    out1, g1 = run(True)
    assert torch.isfinite(out1).all() and torch.isfinite(g1).all()
    assert rel(out1, out0) <= (2e-2 if amp else 1e-5), rel(out1, out0)
    assert rel(g1, g0) <= (5e-2 if amp else 1e-4), rel(g1, g0)


@pytest.mark.gpu
@pytest.mark.parametrize('arch,use_amp', [('ffnn', False), ('dccrn', False), ('dccrn', True), ('tfgridnet', False),
                                          ('tfgridnet', True), ('sgmsep', False), ('sgmsep', True)])
def test_other_models_do_not_read_uninitialised_memory(arch, use_amp):
    """The same check for the other models on the HIP path (default configurations, short inputs): the
    training loss and every gradient computed on freshly poisoned memory (NaNs allocated and freed right
    before) are finite and equal to the run on ordinary memory; SGMSE+ also enhances one utterance."""
    from brever_amd.models import ModelRegistry
    dev = torch.device('cuda')
    seconds = 1.0
    L = int(seconds*16000)
    gen = torch.Generator().manual_seed(23)
    wav = 0.1*torch.randn(2, 2, 2, L, generator=gen)            # (B, sources, channels, L)

    def run(poison):
        torch.manual_seed(8)
        model = ModelRegistry.get(arch)().to(dev).train()
        items = [model.transform(w) for w in wav.to(dev)]
        x = tuple(torch.stack([it[i] for it in items]) for i in range(len(items[0]))) \
            if isinstance(items[0], (tuple, list)) else torch.stack(items)
        lengths = torch.full((2,), (x[0] if isinstance(x, tuple) else x).shape[-1], device=dev)
        if poison:
            torch.cuda.empty_cache()
            junk = [torch.full((1 << 27,), float('nan'), device=dev) for _ in range(12)]
            del junk
        torch.manual_seed(9)                                   # SGMSE+ draws t and the noise here
        loss = model.loss(x, lengths, use_amp)
        loss.backward()
        grads = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None])
        out = None
        if arch == 'sgmsep' and use_amp:
            model.eval()
            torch.manual_seed(10)
            with torch.no_grad():
                out = type(model)(solver_num_steps=2).to(dev).eval().enhance(wav[:1, 0].to(dev), use_amp=True)
        return float(loss), grads.detach().clone(), out

    l0, g0, o0 = run(False)
    l1, g1, o1 = run(True)
    assert math.isfinite(l1) and torch.isfinite(g1).all()
    # (fp32 runs of models whose weight gradients add with atomics differ from run to run at ~1e-4)
    tol = 5e-2 if use_amp else 1e-3
    assert abs(l1 - l0) <= tol*max(1.0, abs(l0)), (l0, l1)
    assert rel(g1, g0) <= tol, rel(g1, g0)
    if o1 is not None:
        assert torch.isfinite(o1).all()


@pytest.mark.gpu
@pytest.mark.parametrize('amp', [True, False])
def test_strided_and_offset_inputs_equal_contiguous_ones(amp):
    """The batch a caller hands over need not be a fresh contiguous tensor: a window of a longer buffer (row
    stride > length, storage offset not a multiple of 16 bytes), lengths given as int32 (host tensors are
    refused loudly: there is no CPU fallback). The
    fused step and the autograd path must give what they give on a contiguous copy."""
    from brever_amd.criterion import snr
    from brever_amd.models import ConvTasNet
    dev = torch.device('cuda')
    gen = torch.Generator().manual_seed(31)
    B, L = 9, 7001
    big = (0.2*torch.randn(B, 2, L + 37, generator=gen)).to(dev)
    view = big[:, :, 3:3 + L]                            # row stride L + 37, offset 3 floats
    assert not view.is_contiguous()
    lengths = torch.tensor([L - 11*i for i in range(B)])
    for b in range(B):
        big[b, :, 3 + int(lengths[b]):] = 0
    cfg = dict(layers=3, repeats=2)

    def step_params(batch, lens):
        torch.manual_seed(4)
        net = ConvTasNet(**cfg).to(dev)
        scaler = torch.amp.GradScaler('cuda', enabled=False)
        losses = [float(net.train_step(batch, lens, amp, scaler)) for _ in range(2)]
        return losses, net.flat_params().detach().clone()

    def autograd_grads(batch, lens):
        torch.manual_seed(4)
        net = ConvTasNet(**cfg).to(dev)
        net._amp = amp
        out = net(batch[:, 0])
        snr(out, batch[:, 1:], lens).mean().backward()
        return torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()

    ref_l, ref_p = step_params(view.contiguous(), lengths.to(dev))
    tol = 2e-2 if amp else 1e-6
    with pytest.raises(RuntimeError):
        step_params(view, lengths)
    for lens in (lengths.to(dev), lengths.to(dev).int()):
        got_l, got_p = step_params(view, lens)
        assert max(abs(a - b) for a, b in zip(got_l, ref_l)) <= (1e-2 if amp else 1e-5), (got_l, ref_l)
        assert rel(got_p, ref_p) <= tol, rel(got_p, ref_p)
    g_ref = autograd_grads(view.contiguous(), lengths.to(dev))
    g_view = autograd_grads(view, lengths.to(dev))
    assert rel(g_view, g_ref) <= (5e-2 if amp else 1e-6), rel(g_view, g_ref)


@pytest.mark.gpu
@pytest.mark.parametrize('amp', [True, False])
def test_changing_batch_shapes_reuse_the_buffers_correctly(amp):
    """An epoch's batches change shape (dynamic batching, the last partial batch): 16 x 3000, 5 x 5000,
    9 x 2100, 16 x 3000, 3 x 800, 12 x 5000 samples in a row through ONE model -- the workspaces grow and are
    re-carved (two chains for B >= 8, one below) -- against a model whose cached buffers are dropped and whose
    memory is poisoned before every step. Same losses; same parameters at the end (bf16: atomics -> 3e-2)."""
    from brever_amd.models import ConvTasNet
    dev = torch.device('cuda')
    shapes = [(16, 3000), (5, 5000), (9, 2100), (16, 3000), (3, 800), (12, 5000)]
    cfg = dict(layers=3, repeats=2)

    def run(fresh_buffers):
        torch.manual_seed(6)
        net = ConvTasNet(**cfg).to(dev)
        scaler = torch.amp.GradScaler('cuda', enabled=False)
        gen = torch.Generator().manual_seed(40)
        losses = []
        for B, L in shapes:
            batch, lengths = _ragged_batch(gen, B, L)
            if fresh_buffers:
                net._workspace, net._ws_key = {}, {}
                net._two, net._step_bufs = None, None
                torch.cuda.empty_cache()
                junk = [torch.full((1 << 27,), float('nan'), device=dev) for _ in range(8)]
                del junk
            losses.append(float(net.train_step(batch.to(dev), lengths.to(dev), amp, scaler)))
        return losses, net.flat_params().detach().clone()

    l0, p0 = run(False)
    l1, p1 = run(True)
    assert all(math.isfinite(v) for v in l0 + l1), (l0, l1)
    assert max(abs(a - b) for a, b in zip(l0, l1)) <= (3e-2 if amp else 1e-5), (l0, l1)
    assert rel(p0, p1) <= (3e-2 if amp else 1e-6), rel(p0, p1)


def test_sixty_four_ragged_utterances_equal_four_steps_of_sixteen(monkeypatch):
    """A dynamic bucket batch larger than anything `bench.py` steps: 64 ragged utterances
    (0.5 - 4 s) through the two-chain fused step give the loss and the gradient of the same items taken as four
    batches of 16 (the mean over the batch: gradient = mean of the four), summation order aside."""
    from brever_amd.models import ConvTasNet
    gen = torch.Generator().manual_seed(5)
    B, L = 64, 64000
    batch = 0.1*torch.randn(B, 2, L, generator=gen)
    lengths = torch.randint(8000, L + 1, (B,), generator=gen)
    lengths[0] = L
    for b in range(B):
        batch[b, :, lengths[b]:] = 0
    torch.manual_seed(3)
    ref = ConvTasNet()

    def fresh():                       # (train_step also takes the optimizer step: every run starts from `ref`)
        net = ConvTasNet()
        net.load_state_dict(ref.state_dict())
        return net.to(_cuda())
    net = fresh()
    loss64, g64 = _fused_step_grads(net, batch, lengths, True, '2', monkeypatch)
    parts, losses = [], []
    for k in range(4):
        sl = slice(16*k, 16*k + 16)
        lk, gk = _fused_step_grads(fresh(), batch[sl], lengths[sl], True, '2', monkeypatch)
        parts.append(gk)
        losses.append(lk)
    g_ref = torch.stack(parts).mean(0)
    assert abs(loss64 - sum(losses)/4) <= 2e-4, (loss64, losses)
    e = rel(g64, g_ref)
    print(f'64 ragged utterances in one step vs 4 x 16: loss {loss64:.5f}, gradient rel {e:.3e}')
    assert e <= 2e-3, e
    _per_tensor(net, g64, g_ref, 2e-2)


def test_mode_mismatch_between_calls_is_loud(monkeypatch):
    """prepare / forward / backward pick the fused or the three-launch kernels from the options of
    their OWN call. A caller that changes BRV_OPT_NO_FWD_FUSE between forward and backward (workspace filled in
    one mode, read in the other) used to get plausible wrong gradients; the calls now stamp `prepared` and the
    workspace with their mode and poison the results with NaN on a mismatch."""
    from brever_amd.criterion import snr
    from brever_amd.models import ConvTasNet
    gen = torch.Generator().manual_seed(3)
    batch, lengths = _ragged_batch(gen, 2, 4000)
    torch.manual_seed(1)
    net = ConvTasNet(layers=2, repeats=1).to(_cuda())
    net._amp = True
    monkeypatch.delenv('BRV_FWD_FUSE', raising=False)
    out = net(batch[:, 0].cuda())
    assert torch.isfinite(out).all()
    loss = snr(out, batch[:, 1:].cuda(), lengths.cuda()).mean()
    monkeypatch.setenv('BRV_FWD_FUSE', '0')            # the backward call now asks for the other mode
    loss.backward()
    got = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    assert torch.isnan(got).any(), 'a mode mismatch between forward and backward must not pass silently'
    # consistent calls in the other mode are fine (the host prepares again: the cache is keyed on the flags)
    net.zero_grad(set_to_none=True)
    out = net(batch[:, 0].cuda())
    snr(out, batch[:, 1:].cuda(), lengths.cuda()).mean().backward()
    got = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    assert torch.isfinite(got).all()
