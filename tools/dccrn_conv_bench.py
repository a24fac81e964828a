"""Forward / backward time of each complex convolution of the default DCCRN (B = 16, 4 s, use_amp):

    python tools/dccrn_conv_bench.py [B]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brever_amd.models import dccrn as D  # noqa: E402

CH = [16, 32, 64, 128, 128, 128]
GEOM = ((5, 2), (2, 1), (2, 0), (1, 0))


def timed(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = torch.device('cuda', 0)
    D._AMP['on'] = True
    layers = []
    H, W = 256, 501
    shapes = []
    for i, c in enumerate(CH):
        cin = 1 if i == 0 else CH[i - 1]
        layers.append(('enc%d' % (i + 1), False, cin, c, H, W))
        shapes.append((H, W))
        H, W = (H + 4 - 5)//2 + 1, W - 1
    for i in range(len(CH) - 1, -1, -1):
        cout = 1 if i == 0 else CH[i - 1]
        layers.append(('dec%d' % (i + 1), True, 2*CH[i], cout, H, W))
        H, W = (H - 1)*2 - 4 + 5 + 1, W + 1
    tot_f = tot_b = 0.0
    for name, tr, cin, cout, H, W in layers:
        skip = None
        if tr and os.environ.get('SKIP'):        # decoder input read from its two sources (no concatenation)
            x = torch.randn(B, cin, H, W, device=dev, requires_grad=True)
            skip = torch.randn(B, cin, H, W, device=dev, requires_grad=True)
        else:
            x = torch.randn(B, 2*cin, H, W, device=dev, requires_grad=True)
        wshape = (cin, cout, 5, 2) if tr else (cout, cin, 5, 2)
        wr = (0.05*torch.randn(wshape, device=dev)).requires_grad_()
        wi = (0.05*torch.randn(wshape, device=dev)).requires_grad_()
        br = torch.zeros(cout, device=dev, requires_grad=True)
        bi = torch.zeros(cout, device=dev, requires_grad=True)
        y = D._ComplexConvFunction.apply(x, wr, br, wi, bi, GEOM, tr, skip)
        dy = torch.randn_like(y)
        f = timed(lambda: D._ComplexConvFunction.apply(x, wr, br, wi, bi, GEOM, tr, skip))
        wrt = (x, wr, wi, br, bi) + ((skip,) if skip is not None else ())
        b = timed(lambda: torch.autograd.grad(y, wrt, dy, retain_graph=True))
        flops = 2.0*B*(2*cin)*(2*cout)*10*(y.shape[2]*y.shape[3] if not tr else H*W)
        if os.environ.get('WGRAD'):
            # the two weight-gradient paths alone
            geom = GEOM[:3]
            Cw = 10*(cout if tr else cin)
            dwc = torch.empty(2*(cin if tr else cout), 2*Cw, device=dev)
            xd, Ho, Wo = x.detach(), y.shape[2], y.shape[3]
            if tr:
                new = timed(lambda: D._cconv_wgrad(xd, dy))
                old = timed(lambda: D._gemm_conv(xd, dy, dwc, 1, 2*cin, 2*Cw, H*W, H*W, 2*Cw, 0, 0, 0, 1, (2*cout, Ho, Wo),
                                                 geom, (H, W), trans_b=1, kbatch=B, a_kbs=2*cin*H*W, img_kbs=2*cout*Ho*Wo))
            else:
                new = timed(lambda: D._cconv_wgrad(dy, xd))
                old = timed(lambda: D._gemm_conv(dy, xd, dwc, 1, 2*cout, 2*Cw, Ho*Wo, Ho*Wo, 2*Cw, 0, 0, 0, 1, (2*cin, H, W),
                                                 geom, (Ho, Wo), trans_b=1, kbatch=B, a_kbs=2*cout*Ho*Wo, img_kbs=2*cin*H*W))
            print(f'{name} wgrad rows {new:7.1f} us ({flops/new/1e6:6.1f} TF/s)   column-matrix {old:7.1f} us', flush=True)
            continue
        tot_f += f
        tot_b += b
        print(f'{name} {2*cin:4d}->{2*cout:4d} in {H:3d}x{W:3d} out {y.shape[2]:3d}x{y.shape[3]:3d}: fwd {f:7.1f} us '
              f'({flops/f/1e6:6.1f} TF/s)  bwd {b:7.1f} us ({2*flops/b/1e6:6.1f} TF/s)', flush=True)
    print(f'total fwd {tot_f/1e3:.2f} ms  bwd {tot_b/1e3:.2f} ms')


if __name__ == '__main__':
    main()
