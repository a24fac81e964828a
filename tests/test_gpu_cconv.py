"""brv_cconv_rows (csrc/cconv.hip): the DCCRN convolution / transposed convolution in one launch, against
torch's CPU float64 convolutions of the SAME bf16-rounded operands (reference geometry
brever/models/dccrn/dccrn.py:225-292: kernel (5, 2), stride (2, 1), padding (2, 0), output_padding (1, 0)).
With identical operands only the order of the fp32 sums differs: the bound is 2e-5 of the output norm.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cuda():
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a ROCm device')
    return torch.device('cuda:0')


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float64)


def _run(x, w_mcij, bias, transposed):
    """w_mcij: (M, C, 5, 2) logical weight; stored as a (M, C*10) matrix (m_stride = 10 C, c_stride = 10)."""
    from brever_amd.models.dccrn import _cconv_rows
    M, C = w_mcij.shape[:2]
    wc = w_mcij.reshape(M, C*10).contiguous()
    return _cconv_rows(x.contiguous(), wc, bias, M, C*10, 10, transposed)


# (B, C, M, Hin, Win): every workgroup shape of the launcher (M <= 32, <= 64, <= 128, > 128 with few and many
# workgroups), channel counts off the chunk of 8, output rows off the fragment of 32, frame counts around the
# 128- / 256-frame tiles, the smallest images
CASES = [
    (2, 2, 32, 16, 37), (1, 32, 64, 8, 129), (2, 64, 128, 8, 300), (1, 128, 256, 4, 257),
    (16, 24, 256, 8, 131), (1, 5, 3, 4, 9), (1, 13, 70, 2, 2), (3, 40, 130, 6, 513), (1, 256, 512, 4, 64),
    (2, 20, 64, 5, 70), (1, 9, 100, 3, 260),      # odd input heights: first / last row pairs of the pair form
]


@pytest.mark.parametrize('case', CASES, ids=lambda c: 'B%d_C%d_M%d_H%d_W%d' % c)
@pytest.mark.parametrize('transposed', [0, 1])
def test_rows_convolution_equals_float64_convolution_of_the_rounded_operands(case, transposed):
    dev = _cuda()
    B, C, M, H, W = case
    if not transposed and H % 2:
        H += 1                      # the strided form is defined on even heights (brv_cconv_rows)
    g = torch.Generator().manual_seed(B*1000 + C*10 + M + H + W + transposed)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(M, C, 5, 2, generator=g)/(C*10)**0.5
    bias = torch.randn(M, generator=g)
    y = _run(x.to(dev), w.to(dev), bias.to(dev), transposed).cpu().double()
    if transposed:
        ref = torch.nn.functional.conv_transpose2d(_bf(x), _bf(w).transpose(0, 1).contiguous(), bias.double(),
                                                   stride=(2, 1), padding=(2, 0), output_padding=(1, 0))
    else:
        ref = torch.nn.functional.conv2d(_bf(x), _bf(w), bias.double(), stride=(2, 1), padding=(2, 0))
    assert y.shape == ref.shape
    err = float((y - ref).norm()/ref.norm())
    assert err < 2e-5, err
    assert float((y - ref).abs().max()) < 1e-4*float(ref.abs().max()) + 1e-5


def test_rows_convolution_without_bias_and_strided_weight_matrix():
    """The data-gradient call pattern: no bias, W[m][c][i][j] taken from a (C, M*10) matrix (m_stride = 10)."""
    dev = _cuda()
    g = torch.Generator().manual_seed(5)
    B, C, M, H, W = 2, 48, 96, 6, 200
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(M, C, 5, 2, generator=g)/(C*10)**0.5
    wc = w.permute(1, 0, 2, 3).reshape(C, M*10).contiguous()          # [c][(m, i, j)]
    from brever_amd.models.dccrn import _cconv_rows
    for transposed in (0, 1):
        y = _cconv_rows(x.to(dev), wc.to(dev), None, M, 10, M*10, transposed).cpu().double()
        if transposed:
            ref = torch.nn.functional.conv_transpose2d(_bf(x), _bf(w).transpose(0, 1).contiguous(), None,
                                                       stride=(2, 1), padding=(2, 0), output_padding=(1, 0))
        else:
            ref = torch.nn.functional.conv2d(_bf(x), _bf(w), None, stride=(2, 1), padding=(2, 0))
        assert float((y - ref).norm()/ref.norm()) < 2e-5


@pytest.mark.parametrize('shape', [(2, 16, 72, 6, 140), (1, 8, 24, 4, 300), (2, 24, 200, 4, 129), (1, 8, 64, 2, 70),
                                   (1, 40, 520, 2, 33)],
                         ids=lambda c: 'B%d_seg%d_M%d_H%d_W%d' % c)
def test_rows_convolution_reads_and_writes_a_skip_concatenation_in_place(shape):
    """Input from two sources == input concatenated; output dealt to two tensors == output split (every
    workgroup shape: M <= 32 / 64 / 128 / 256 rows, pair and single-row forms)."""
    from brever_amd.models.dccrn import _cconv_rows
    dev = _cuda()
    g = torch.Generator().manual_seed(11)
    B, seg, M, H, W = shape
    x1 = torch.randn(B, 2*seg, H, W, generator=g).to(dev)
    x2 = torch.randn(B, 2*seg, H, W, generator=g).to(dev)
    wc = (torch.randn(M, 4*seg*10, generator=g)/(40*seg)**0.5).to(dev)
    cat = torch.cat([x1[:, :seg], x2[:, :seg], x1[:, seg:], x2[:, seg:]], dim=1).contiguous()
    for transposed in (0, 1):
        one = _cconv_rows(cat, wc, None, M, 4*seg*10, 10, transposed)
        two = _cconv_rows(x1, wc, None, M, 4*seg*10, 10, transposed, x2=x2)
        assert torch.equal(one, two)
        a, b = _cconv_rows(cat, wc, None, M, 4*seg*10, 10, transposed, split_out=True)
        q = M//4
        assert torch.equal(a, torch.cat([one[:, :q], one[:, 2*q:3*q]], dim=1))
        assert torch.equal(b, torch.cat([one[:, q:2*q], one[:, 3*q:]], dim=1))


WG_CASES = [  # (B, A, C, Hs, Ws)
    (2, 32, 2, 8, 37), (1, 64, 32, 4, 129), (3, 128, 64, 2, 63), (2, 256, 40, 3, 64), (1, 130, 33, 1, 1),
    (16, 96, 256, 2, 200), (1, 5, 3, 4, 500),
]


@pytest.mark.parametrize('case', WG_CASES, ids=lambda c: 'B%d_A%d_C%d_H%d_W%d' % c)
def test_rows_weight_gradient_equals_float64_autograd_of_the_rounded_operands(case):
    """brv_cconv_wgrad against d/dw of conv2d(big, w) . small in float64 (bf16-rounded images)."""
    from brever_amd.models.dccrn import _cconv_wgrad
    dev = _cuda()
    B, A, C, Hs, Ws = case
    g = torch.Generator().manual_seed(sum(case))
    small = torch.randn(B, A, Hs, Ws, generator=g)
    big = torch.randn(B, C, 2*Hs, Ws + 1, generator=g)
    out = _cconv_wgrad(small.to(dev), big.to(dev)).cpu().double()
    w = torch.zeros(A, C, 5, 2, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(_bf(big), w, None, stride=(2, 1), padding=(2, 0))
    (ref,) = torch.autograd.grad(y, w, _bf(small))
    ref = ref.reshape(A, C*10)
    assert out.shape == ref.shape
    assert float((out - ref).norm()/ref.norm()) < 3e-5


def test_rows_weight_gradient_reads_a_skip_concatenation_from_its_two_sources():
    from brever_amd.models.dccrn import _cconv_wgrad
    dev = _cuda()
    g = torch.Generator().manual_seed(9)
    B, seg, C, Hs, Ws = 2, 24, 20, 4, 150
    s1 = torch.randn(B, 2*seg, Hs, Ws, generator=g).to(dev)
    s2 = torch.randn(B, 2*seg, Hs, Ws, generator=g).to(dev)
    big = torch.randn(B, C, 2*Hs, Ws + 1, generator=g).to(dev)
    cat = torch.cat([s1[:, :seg], s2[:, :seg], s1[:, seg:], s2[:, seg:]], dim=1).contiguous()
    one = _cconv_wgrad(cat, big)
    two = _cconv_wgrad(s1, big, small2=s2)
    assert float((one - two).norm()/one.norm()) < 1e-6        # same products, atomics in another order


def test_rows_and_column_matrix_paths_give_the_same_dccrn_gradients():
    """use_amp DCCRN step at a small size: loss and every gradient with BRV_DCCRN_ROWS on and off."""
    import brever_amd.models.dccrn as D
    dev = _cuda()
    torch.manual_seed(0)
    model = D.DCCRN(channels=[8, 16, 32], lstm_channels=32).to(dev)
    batch = 0.1*torch.randn(2, 2, 8000, generator=torch.Generator().manual_seed(1)).to(dev)
    lengths = torch.tensor([8000, 7000], device=dev)
    out = {}
    for rows in (True, False):
        D._ROWS = rows
        model.zero_grad(set_to_none=True)
        loss = model.loss(batch, lengths, True)
        loss.backward()
        out[rows] = (float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters()})
    D._ROWS = True
    assert abs(out[True][0] - out[False][0]) < 2e-2*abs(out[False][0]) + 1e-3
    num = sum(float((out[True][1][n].double() - out[False][1][n].double()).norm()**2) for n in out[True][1])
    den = sum(float(out[False][1][n].double().norm()**2) for n in out[True][1])
    assert (num/den)**0.5 < 3e-2, (num/den)**0.5


def test_parameter_gradients_on_the_side_stream_equal_the_in_order_ones():
    """use_amp DCCRN with the weight / bias gradients queued on the side stream against the same work in order on
    one stream: every gradient of one backward pass, and the losses of five training steps. Two in-order runs are
    not bitwise equal themselves (split reductions of the generic products add with atomics: up to 3e-3 of a
    tensor's norm on the recurrent weights between two identical in-order passes, 1e-4 between the two orders on
    the convolution weights, parameters 6e-4 apart after five Adam steps), so the bounds are a multiple of that
    run-to-run spread, not zero: a read of freed or unwritten memory shows up as O(1)."""
    import brever_amd.models.dccrn as D
    dev = _cuda()
    batch = 0.1*torch.randn(3, 2, 12000, generator=torch.Generator().manual_seed(2)).to(dev)
    lengths = torch.tensor([12000, 11000, 9000], device=dev)
    scaler = torch.amp.GradScaler('cuda', enabled=False)
    grads, losses = {}, {}
    for side in (True, False):
        D._WGRAD_SIDE = side
        torch.manual_seed(7)
        model = D.DCCRN(channels=[8, 16, 32, 32], lstm_channels=32).to(dev)
        model.loss(batch, lengths, True).backward()
        grads[side] = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
        model.zero_grad(set_to_none=True)
        losses[side] = [float(model.train_step(batch, lengths, True, scaler).detach()) for _ in range(5)]
        torch.cuda.synchronize()
    D._WGRAD_SIDE = True
    top = max(float(g.double().norm()) for g in grads[False].values())
    for n in grads[True]:          # (the biases in front of a batch norm have gradients of pure rounding noise)
        a, b = grads[True][n].double(), grads[False][n].double()
        assert float((a - b).norm()) <= 1e-2*float(b.norm()) + 1e-6*top, n
    assert max(abs(x - y) for x, y in zip(losses[True], losses[False])) < 2e-3


# ---- bf16 images (round 6): the same products on images stored as bf16 ---------------------------------------------
def _as_bf16(t):
    """bf16 copy with the readable slack around it the LDS-DMA row kernels ask for (include/brever_hip.h) -- the slack
    holds NaNs: what lies there must never reach a sum (0 x NaN is NaN)."""
    n = t.numel()
    flat = torch.full((n + 16,), float('nan'), dtype=torch.bfloat16, device=t.device)
    out = flat[8:8 + n].view(t.shape)
    out.copy_(t)
    return out


def _rows_raw(x, x2, seg, wc, M, C, m_stride, c_stride, transposed, split_out=False):
    """brv_cconv_rows / brv_cconv_rows_bf16 by the dtype of ``x`` (outputs fp32)."""
    from brever_amd import hip
    lib = hip.lib()
    B, _, H, W = x.shape
    wp = torch.empty(lib.brv_cconv_packed_bytes(M, C), dtype=torch.uint8, device=x.device)
    hip.check(lib.brv_cconv_pack(hip.ptr(wc), hip.ptr(wp), M, C, m_stride, c_stride, hip.stream()), 'brv_cconv_pack')
    shape = (B, M//2 if split_out else M) + ((2*H, W + 1) if transposed else (H//2, W - 1))
    out = torch.empty(shape, dtype=torch.float32, device=x.device)
    out2 = torch.empty_like(out) if split_out else None
    fn = lib.brv_cconv_rows_bf16 if x.dtype == torch.bfloat16 else lib.brv_cconv_rows
    hip.check(fn(hip.ptr(x), hip.ptr(x2), seg, hip.ptr(wp), None, hip.ptr(out), hip.ptr(out2), M//4 if split_out else 0,
                 B, C, M, H, W, int(transposed), hip.stream()), 'brv_cconv_rows*')
    return (out, out2) if split_out else out


@pytest.mark.parametrize('case', CASES, ids=lambda c: 'B%d_C%d_M%d_H%d_W%d' % c)
@pytest.mark.parametrize('transposed', [0, 1])
def test_rows_convolution_on_bf16_images_is_bit_identical(case, transposed):
    """An image of bf16-representable values gives the same bits whether it is stored as fp32 or as bf16 (every
    workgroup shape and both loaders; odd frame counts put the bf16 rows on 2-byte boundaries)."""
    dev = _cuda()
    B, C, M, H, W = case
    if not transposed and H % 2:
        H += 1
    g = torch.Generator().manual_seed(B*1000 + C*10 + M + H + W + transposed)
    x16 = _as_bf16(torch.randn(B, C, H, W, generator=g).to(dev))
    wc = (torch.randn(M, C*10, generator=g)/(C*10)**0.5).to(dev)
    a = _rows_raw(x16.float(), None, 0, wc, M, C, C*10, 10, transposed)
    b = _rows_raw(x16, None, 0, wc, M, C, C*10, 10, transposed)
    assert torch.equal(a, b)


@pytest.mark.parametrize('shape', [(2, 16, 72, 6, 140), (1, 8, 24, 4, 301), (2, 24, 200, 4, 129), (1, 8, 64, 2, 70),
                                   (1, 40, 520, 2, 33)],
                         ids=lambda c: 'B%d_seg%d_M%d_H%d_W%d' % c)
def test_rows_convolution_on_bf16_images_two_sources_and_split_output(shape):
    dev = _cuda()
    g = torch.Generator().manual_seed(12)
    B, seg, M, H, W = shape
    x1 = _as_bf16(torch.randn(B, 2*seg, H, W, generator=g).to(dev))
    x2 = _as_bf16(torch.randn(B, 2*seg, H, W, generator=g).to(dev))
    wc = (torch.randn(M, 4*seg*10, generator=g)/(40*seg)**0.5).to(dev)
    for transposed in (0, 1):
        a = _rows_raw(x1.float(), x2.float(), seg, wc, M, 4*seg, 4*seg*10, 10, transposed)
        b = _rows_raw(x1, x2, seg, wc, M, 4*seg, 4*seg*10, 10, transposed)
        assert torch.equal(a, b)
        cat = _as_bf16(torch.cat([x1[:, :seg], x2[:, :seg], x1[:, seg:], x2[:, seg:]], dim=1))
        a1, a2 = _rows_raw(cat.float(), None, 0, wc, M, 4*seg, 4*seg*10, 10, transposed, split_out=True)
        b1, b2 = _rows_raw(cat, None, 0, wc, M, 4*seg, 4*seg*10, 10, transposed, split_out=True)
        assert torch.equal(a1, b1) and torch.equal(a2, b2)


@pytest.mark.parametrize('case', WG_CASES, ids=lambda c: 'B%d_A%d_C%d_H%d_W%d' % c)
def test_rows_weight_gradient_on_bf16_images_is_bit_identical(case):
    from brever_amd.models.dccrn import _cconv_wgrad
    dev = _cuda()
    B, A, C, Hs, Ws = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    small = _as_bf16(torch.randn(B, A, Hs, Ws, generator=g).to(dev))
    big = _as_bf16(torch.randn(B, C, 2*Hs, Ws + 1, generator=g).to(dev))
    assert torch.equal(_cconv_wgrad(small.float(), big.float()), _cconv_wgrad(small, big))
    if A % 4 == 0 and A >= 8:       # the two-source form (segments of 8 channels and more: LDS-DMA; fewer: registers)
        seg = A//4
        s1, s2 = _as_bf16(small[:, :2*seg]), _as_bf16(small[:, 2*seg:])
        assert torch.equal(_cconv_wgrad(s1.float(), big.float(), small2=s2.float()), _cconv_wgrad(s1, big, small2=s2))


@pytest.mark.parametrize('shape', [(2, 6, 8, 10), (3, 16, 4, 125), (1, 5, 2, 2)], ids=lambda c: 'B%d_C%d_H%d_W%d' % c)
@pytest.mark.parametrize('act', [True, False])
def test_batch_norm_passes_with_bf16_output(shape, act):
    """brv_batchnorm2d_forward_bf16 / _backward_bf16: the fp32 passes' element-wise outputs rounded to bf16, identical
    statistics and parameter gradients, and the channel sums of the unrounded dx."""
    from brever_amd import hip
    lib = hip.lib()
    dev = _cuda()
    B, C, H, W = shape
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, C, H, W, generator=g).to(dev)
    dy = torch.randn(B, C, H, W, generator=g).to(dev)
    gamma = (1 + 0.1*torch.randn(C, generator=g)).to(dev)
    beta = (0.1*torch.randn(C, generator=g)).to(dev)
    slope = torch.tensor([0.25], device=dev) if act else None
    outs = {}
    for lowp in (False, True):
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        mean, invstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
        y = torch.empty(B, C, H, W, dtype=torch.bfloat16 if lowp else torch.float32, device=dev)
        fn = lib.brv_batchnorm2d_forward_bf16 if lowp else lib.brv_batchnorm2d_forward
        hip.check(fn(hip.ptr(x), hip.ptr(gamma), hip.ptr(beta), hip.ptr(rm), hip.ptr(rv), hip.ptr(slope), hip.ptr(y),
                     hip.ptr(mean), hip.ptr(invstd), B, C, H*W, 1e-5, 0.1, 1, hip.stream()), 'bn forward')
        dx = torch.empty_like(y)
        dgamma, dbeta, dsl, sums = (torch.empty(C, device=dev) for _ in range(4))
        args = (hip.ptr(x), hip.ptr(dy), hip.ptr(mean), hip.ptr(invstd), hip.ptr(gamma), hip.ptr(beta), hip.ptr(slope),
                hip.ptr(dx), hip.ptr(dgamma), hip.ptr(dbeta), hip.ptr(dsl))
        if lowp:
            hip.check(lib.brv_batchnorm2d_backward_bf16(*args, hip.ptr(sums), B, C, H*W, hip.stream()), 'bn backward')
        else:
            hip.check(lib.brv_batchnorm2d_backward(*args, B, C, H*W, hip.stream()), 'bn backward')
        outs[lowp] = (y, dx, mean, invstd, rm, rv, dgamma, dbeta, dsl, sums)
    f, h = outs[False], outs[True]
    assert torch.equal(f[0].to(torch.bfloat16), h[0]) and torch.equal(f[1].to(torch.bfloat16), h[1])
    for k in range(2, 9):
        assert torch.equal(f[k], h[k]), k
    ref = f[1].double().sum(dim=(0, 2, 3))
    assert float((h[9].double() - ref).abs().max()) <= 1e-5*float(f[1].double().abs().sum(dim=(0, 2, 3)).max()) + 1e-7


def test_batch_norm_bf16_output_refuses_planes_off_the_16_byte_form():
    from brever_amd import hip
    lib = hip.lib()
    dev = _cuda()
    x = torch.randn(1, 2, 3, 3, device=dev)
    t = torch.empty(2, device=dev)
    y = torch.empty(1, 2, 3, 3, dtype=torch.bfloat16, device=dev)
    rc = lib.brv_batchnorm2d_forward_bf16(hip.ptr(x), hip.ptr(t), hip.ptr(t), None, None, None, hip.ptr(y), hip.ptr(t),
                                          hip.ptr(t), 1, 2, 9, 1e-5, 0.1, 1, hip.stream())
    assert rc == -1


@pytest.mark.parametrize('channels', [[8, 16, 32], [16, 32, 64, 128, 128, 128]], ids=['small', 'default'])
def test_bf16_activations_between_the_blocks_change_no_product(channels):
    """use_amp DCCRN with the activations between the blocks stored as bf16 (``_BlockFunction``) against the round-5
    path that keeps them fp32 and rounds inside the kernels (BRV_DCCRN_BF16_ACT=0): the same rounding points and the
    same products (the kernel-level tests above: bit-identical). Two runs of ONE path are not bitwise equal (the
    recurrent block's products add their split reductions with atomics, and a last-bit difference flips bf16 roundings
    downstream), so the yardstick is measured: the round-5 path run twice. The bf16 path may differ from it by three
    times that spread (+ 1e-5 of the norm); the biases in front of a batch norm have gradients of pure rounding noise
    and are held to the top gradient norm."""
    import brever_amd.models.dccrn as D
    dev = _cuda()
    n = 16000
    batch = 0.1*torch.randn(2, 2, n, generator=torch.Generator().manual_seed(4)).to(dev)
    lengths = torch.tensor([n, n - 3000], device=dev)
    out = []
    y16 = D._BF16_Y
    try:
        D._BF16_Y = False           # (the bf16 convolution OUTPUT is a rounding point of its own: tested against the oracle)
        g16, D._BF16_GRAD = D._BF16_GRAD, False       # (and so are the bf16 gradients between the blocks)
        for lowp in (False, False, True):
            D._BF16_ACT = lowp
            torch.manual_seed(3)
            model = D.DCCRN(channels=channels, lstm_channels=32 if len(channels) == 3 else 128).to(dev)
            with torch.no_grad():
                enh = model._enhance(batch[:, 0:1].repeat(1, 2, 1), True)
            loss = model.loss(batch, lengths, True)
            loss.backward()
            torch.cuda.synchronize()
            out.append((float(loss.detach()), {k: p.grad.detach().clone() for k, p in model.named_parameters()}, enh,
                        {k: b.detach().clone() for k, b in model.named_buffers()}))
    finally:
        D._BF16_ACT = True
        D._BF16_Y = y16
        D._BF16_GRAD = g16
    ref, again, new = out

    def dist(a, b):
        return float((a.double() - b.double()).norm())
    # (one pair of runs is a noisy yardstick -- losses 5e-6 to 4e-5 apart over the pairs seen -- so every bound also has
    # a floor of the size of that spread: an O(1) defect, a wrong product or a stale image, is what this test is for)
    assert abs(new[0] - ref[0]) <= 3*abs(again[0] - ref[0]) + 2e-4*abs(ref[0]), (new[0], again[0], ref[0])
    assert dist(new[2], ref[2]) <= 3*dist(again[2], ref[2]) + 5e-3*float(ref[2].norm())
    for k in ref[3]:
        assert dist(new[3][k], ref[3][k]) <= 3*dist(again[3][k], ref[3][k]) + 1e-3*float(ref[3][k].double().norm()) + 1e-9, k
    # gradients: every tensor of 64 elements or more (the scalar PReLU slopes are sums with heavy cancellation: 3e-2 to
    # 2e-1 apart between two runs of one path -- they count in the global bound), and all of them together
    top = max(float(g.double().norm()) for g in ref[1].values())
    for k in ref[1]:
        if ref[1][k].numel() >= 64:
            assert dist(new[1][k], ref[1][k]) <= 3*dist(again[1][k], ref[1][k]) + 1e-2*float(ref[1][k].double().norm()) + 1e-6*top, \
                (k, dist(new[1][k], ref[1][k]), dist(again[1][k], ref[1][k]), float(ref[1][k].double().norm()))
    tot = sum(float(g.double().norm())**2 for g in ref[1].values())**0.5
    d_new = sum(dist(new[1][k], ref[1][k])**2 for k in ref[1])**0.5
    d_again = sum(dist(again[1][k], ref[1][k])**2 for k in ref[1])**0.5
    assert d_new <= 3*d_again + 5e-3*tot, (d_new, d_again, tot)


@pytest.mark.parametrize('case', [(2, 32, 64, 8, 129, 0), (2, 64, 128, 8, 300, 1), (1, 128, 256, 4, 257, 0),
                                  (16, 24, 256, 8, 131, 1), (2, 2, 32, 16, 37, 0)],
                         ids=lambda c: 'B%d_C%d_M%d_H%d_W%d_t%d' % c)
def test_rows_convolution_with_bf16_output(case):
    """brv_cconv_rows_ex(out_bf16 = 1): the fp32 output rounded to bf16, for fp32 and bf16 images, one and two targets."""
    from brever_amd import hip
    lib = hip.lib()
    dev = _cuda()
    B, C, M, H, W, transposed = case
    g = torch.Generator().manual_seed(sum(case))
    x16 = _as_bf16(torch.randn(B, C, H, W, generator=g).to(dev))
    wc = (torch.randn(M, C*10, generator=g)/(C*10)**0.5).to(dev)
    bias = torch.randn(M, generator=g).to(dev)
    wp = torch.empty(lib.brv_cconv_packed_bytes(M, C), dtype=torch.uint8, device=dev)
    hip.check(lib.brv_cconv_pack(hip.ptr(wc), hip.ptr(wp), M, C, C*10, 10, hip.stream()), 'brv_cconv_pack')
    shape = (B, M) + ((2*H, W + 1) if transposed else (H//2, W - 1))
    for split in (False, True):
        if split and M % 4:
            continue
        oshape = (B, M//2) + shape[2:] if split else shape
        for in_bf16 in (0, 1):
            xin = x16 if in_bf16 else x16.float()
            outs = {}
            for ob in (0, 1):
                o1 = torch.empty(oshape, dtype=torch.bfloat16 if ob else torch.float32, device=dev)
                o2 = torch.empty_like(o1) if split else None
                hip.check(lib.brv_cconv_rows_ex(hip.ptr(xin), None, 0, hip.ptr(wp), hip.ptr(bias), hip.ptr(o1), hip.ptr(o2),
                                                M//4 if split else 0, B, C, M, H, W, transposed, in_bf16, ob, hip.stream()),
                          'brv_cconv_rows_ex')
                outs[ob] = (o1, o2)
            assert torch.equal(outs[0][0].to(torch.bfloat16), outs[1][0])
            if split:
                assert torch.equal(outs[0][1].to(torch.bfloat16), outs[1][1])


@pytest.mark.parametrize('shape', [(2, 6, 8, 10), (3, 16, 4, 125)], ids=lambda c: 'B%d_C%d_H%d_W%d' % c)
def test_batch_norm_passes_with_bf16_input_and_output(shape):
    """brv_batchnorm2d_forward_bf16io / _backward_bf16io on a bf16 x: the bf16-output passes on the widened values."""
    from brever_amd import hip
    lib = hip.lib()
    dev = _cuda()
    B, C, H, W = shape
    g = torch.Generator().manual_seed(5)
    x16 = torch.randn(B, C, H, W, generator=g).to(dev).to(torch.bfloat16)
    dy = torch.randn(B, C, H, W, generator=g).to(dev)
    gamma = (1 + 0.1*torch.randn(C, generator=g)).to(dev)
    beta = (0.1*torch.randn(C, generator=g)).to(dev)
    slope = torch.tensor([0.25], device=dev)
    outs = {}
    for io in (False, True):
        x = x16 if io else x16.float()
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        mean, invstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
        y = torch.empty(B, C, H, W, dtype=torch.bfloat16, device=dev)
        fwd = lib.brv_batchnorm2d_forward_bf16io if io else lib.brv_batchnorm2d_forward_bf16
        hip.check(fwd(hip.ptr(x), hip.ptr(gamma), hip.ptr(beta), hip.ptr(rm), hip.ptr(rv), hip.ptr(slope), hip.ptr(y),
                      hip.ptr(mean), hip.ptr(invstd), B, C, H*W, 1e-5, 0.1, 1, hip.stream()), 'bn forward')
        dx = torch.empty_like(y)
        dgamma, dbeta, dsl, sums = (torch.empty(C, device=dev) for _ in range(4))
        bwd = lib.brv_batchnorm2d_backward_bf16io if io else lib.brv_batchnorm2d_backward_bf16
        hip.check(bwd(hip.ptr(x), hip.ptr(dy), hip.ptr(mean), hip.ptr(invstd), hip.ptr(gamma), hip.ptr(beta), hip.ptr(slope),
                      hip.ptr(dx), hip.ptr(dgamma), hip.ptr(dbeta), hip.ptr(dsl), hip.ptr(sums), B, C, H*W, hip.stream()),
                  'bn backward')
        outs[io] = (y, dx, mean, invstd, rm, rv, dgamma, dbeta, dsl, sums)
    for k in range(6):
        assert torch.equal(outs[False][k], outs[True][k]), k
    for k in range(6, 10):      # (reductions: the two instantiations may contract their multiply-adds differently)
        a, b = outs[False][k].double(), outs[True][k].double()
        assert float((a - b).abs().max()) <= 1e-5*float(b.abs().max()) + 1e-6, k


@pytest.mark.parametrize('types', [(0, 0), (0, 1), (1, 1)], ids=['f32_f32', 'f32_bf16', 'bf16_bf16'])
def test_batch_norm_backward_adds_a_second_gradient_on_the_fly(types):
    """brv_batchnorm2d_backward_ex(dy, dy2) == the same pass on dy + dy2 (an encoder block's two consumers)."""
    from brever_amd import hip
    lib = hip.lib()
    dev = _cuda()
    x_bf16, dx_bf16 = types
    B, C, H, W = 3, 16, 4, 125
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, C, H, W, generator=g).to(dev)
    x = x.to(torch.bfloat16) if x_bf16 else x
    dy, dy2 = torch.randn(B, C, H, W, generator=g).to(dev), torch.randn(B, C, H, W, generator=g).to(dev)
    gamma = (1 + 0.1*torch.randn(C, generator=g)).to(dev)
    beta = (0.1*torch.randn(C, generator=g)).to(dev)
    slope = torch.tensor([0.25], device=dev)
    xf = x.float()
    mean = xf.mean(dim=(0, 2, 3)).contiguous()
    invstd = (xf.var(dim=(0, 2, 3), unbiased=False) + 1e-5).rsqrt().contiguous()
    outs = []
    for a, b in ((dy + dy2, None), (dy, dy2)):
        dx = torch.empty(B, C, H, W, dtype=torch.bfloat16 if dx_bf16 else torch.float32, device=dev)
        dgamma, dbeta, dsl, sums = (torch.empty(C, device=dev) for _ in range(4))
        hip.check(lib.brv_batchnorm2d_backward_ex(hip.ptr(x), x_bf16, hip.ptr(a), hip.ptr(b), 0, hip.ptr(mean), hip.ptr(invstd),
                                                  hip.ptr(gamma), hip.ptr(beta), hip.ptr(slope), hip.ptr(dx), dx_bf16,
                                                  hip.ptr(dgamma), hip.ptr(dbeta), hip.ptr(dsl), hip.ptr(sums), B, C, H*W,
                                                  hip.stream()), 'brv_batchnorm2d_backward_ex')
        outs.append((dx, dgamma, dbeta, dsl, sums))
    assert torch.equal(outs[0][0], outs[1][0])
    for k in range(1, 5):
        a, b = outs[0][k].double(), outs[1][k].double()
        assert float((a - b).abs().max()) <= 1e-5*float(b.abs().max()) + 1e-6, k


def test_bf16_block_path_in_eval_mode_follows_the_emulating_oracle():
    """``enhance`` under use_amp (batch norms on their running statistics, the bf16-in / bf16-out norm pass with
    ``training = 0``) after a few training steps, against OracleDCCRN(emulate_bf16=True) in eval mode at the same
    parameters and running statistics; and the gradient of an eval-mode norm stays refused."""
    import brever_amd.models.dccrn as D
    from oracle.dccrn import OracleDCCRN
    dev = _cuda()
    torch.manual_seed(11)
    kw = dict(channels=[8, 16, 32], lstm_channels=32)
    net = D.DCCRN(**kw).to(dev)
    n = 16000
    batch = 0.1*torch.randn(2, 2, n, generator=torch.Generator().manual_seed(12)).to(dev)
    lengths = torch.tensor([n, n - 2000], device=dev)
    scaler = torch.amp.GradScaler('cuda', enabled=False)
    for _ in range(3):                                  # running statistics away from their initial values
        net.train_step(batch, lengths, True, scaler)
    oracle = OracleDCCRN(**kw, emulate_bf16=True)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items() if k.startswith('mask_net.')}
    missing, unexpected = oracle.load_state_dict(sd, strict=False)
    assert not unexpected and all('mask_net' not in k or 'num_batches' in k for k in missing), (missing, unexpected)
    oracle.eval()
    net.eval()
    x = batch[:, 0].cpu()
    with torch.no_grad():
        want = oracle(x)
        got = net._enhance(batch[:, 0:1].repeat(1, 2, 1), True).cpu()
        plain = OracleDCCRN(**kw)
        plain.load_state_dict(oracle.state_dict())
        plain.eval()
        ref32 = plain(x)
    err, emu = float((got - want).norm()/want.norm()), float((want - ref32).norm()/ref32.norm())
    print(f'eval-mode use_amp enhance: HIP vs emulating oracle {err:.3e}; emulation vs fp32 oracle {emu:.3e}')
    assert err <= 0.5*emu + 2e-3, (err, emu)
    with pytest.raises(NotImplementedError):
        net.loss(batch, lengths, True).backward()


@pytest.mark.parametrize('flag,tol', [('_TWO_TOKENS', 5e-2), ('_LINEAR_FUSED', 5e-3), ('_LINEAR_LOWP', 5e-2), ('_BF16_Y', 5e-2),
                                      ('_BF16_GRAD', 5e-2),
                                      ('_BF16_ACT', 5e-2), ('_WGRAD_SIDE', 5e-3)])
def test_every_host_side_switch_of_the_use_amp_path_still_runs_and_agrees(flag, tol):
    """The A/B switches of models/dccrn.py are code paths of their own: each one turned off gives the
    loss and the gradients of the default path -- to the run-to-run spread where only the data movement differs, to the
    bf16 rounding it adds or removes otherwise (global relative gradient distance)."""
    import brever_amd.models.dccrn as D
    dev = _cuda()
    n = 12000
    batch = 0.1*torch.randn(3, 2, n, generator=torch.Generator().manual_seed(21)).to(dev)
    lengths = torch.tensor([n, n - 1000, n - 4000], device=dev)
    out = {}
    old = getattr(D, flag)
    try:
        for value in (True, False):
            setattr(D, flag, value)
            torch.manual_seed(5)
            model = D.DCCRN(channels=[8, 16, 32, 32], lstm_channels=32).to(dev)
            loss = model.loss(batch, lengths, True)
            loss.backward()
            torch.cuda.synchronize()
            out[value] = (float(loss.detach()), torch.cat([p.grad.reshape(-1) for p in model.parameters()]).double())
    finally:
        setattr(D, flag, old)
    assert abs(out[True][0] - out[False][0]) <= tol*abs(out[True][0]) + 1e-4, (out[True][0], out[False][0])
    d = float((out[True][1] - out[False][1]).norm()/out[True][1].norm())
    assert d <= tol + 3e-3, d


def test_batch_norm_backward_reads_bf16_gradients():
    """brv_batchnorm2d_backward_ex(dy_bf16 = 1) == the same pass on the widened gradients (one and two of them)."""
    from brever_amd import hip
    lib = hip.lib()
    dev = _cuda()
    B, C, H, W = 3, 16, 4, 125
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, C, H, W, generator=g).to(dev).to(torch.bfloat16)
    dy, dy2 = (torch.randn(B, C, H, W, generator=g).to(dev).to(torch.bfloat16) for _ in range(2))
    gamma = (1 + 0.1*torch.randn(C, generator=g)).to(dev)
    beta = (0.1*torch.randn(C, generator=g)).to(dev)
    slope = torch.tensor([0.25], device=dev)
    xf = x.float()
    mean = xf.mean(dim=(0, 2, 3)).contiguous()
    invstd = (xf.var(dim=(0, 2, 3), unbiased=False) + 1e-5).rsqrt().contiguous()
    for second in (None, dy2):
        outs = []
        for lowp in (0, 1):
            a = dy if lowp else dy.float()
            b = None if second is None else (second if lowp else second.float())
            dx = torch.empty(B, C, H, W, dtype=torch.bfloat16, device=dev)
            dgamma, dbeta, dsl, sums = (torch.empty(C, device=dev) for _ in range(4))
            hip.check(lib.brv_batchnorm2d_backward_ex(hip.ptr(x), 1, hip.ptr(a), hip.ptr(b), lowp, hip.ptr(mean),
                                                      hip.ptr(invstd), hip.ptr(gamma), hip.ptr(beta), hip.ptr(slope),
                                                      hip.ptr(dx), 1, hip.ptr(dgamma), hip.ptr(dbeta), hip.ptr(dsl),
                                                      hip.ptr(sums), B, C, H*W, hip.stream()), 'brv_batchnorm2d_backward_ex')
            outs.append((dx, dgamma, dbeta, dsl, sums))
        assert torch.equal(outs[0][0], outs[1][0])
        for k in range(1, 5):
            p, q = outs[0][k].double(), outs[1][k].double()
            assert float((p - q).abs().max()) <= 1e-5*float(q.abs().max()) + 1e-6, k
