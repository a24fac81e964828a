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
