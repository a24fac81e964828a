"""Compares the res/skip weight gradients of the full-width kernel with the generic path."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if len(sys.argv) > 1:
    from brever_amd.models import ConvTasNet
    from brever_amd.criterion import snr
    torch.manual_seed(0)
    net = ConvTasNet().cuda()
    B, L = 3, 16000 + 777
    x = 0.1*torch.randn(B, 3, L, device='cuda')
    lengths = torch.tensor([L, L - 500, L - 3000], device='cuda')
    out = net(x[:, 0])
    loss = snr(out, 0.1*torch.randn_like(out), lengths).mean()
    loss.backward()
    torch.save({n: p.grad.cpu() for n, p in net.named_parameters()}, sys.argv[1])
    sys.exit(0)
env = dict(os.environ)
subprocess.check_call([sys.executable, __file__, '/tmp/g_full.pt'], env=env)
env['BRV_NO_WGRAD_FULL'] = '1'
subprocess.check_call([sys.executable, __file__, '/tmp/g_ref.pt'], env=env)
a, b = torch.load('/tmp/g_full.pt'), torch.load('/tmp/g_ref.pt')
for n in a:
    if 'res_conv' in n or 'skip_conv' in n:
        d = (a[n] - b[n]).abs().max().item(); r = b[n].abs().max().item()
        bad = (~torch.isfinite(a[n])).sum().item()
        if d > 1e-3*r or bad:
            da = (a[n] - b[n]).abs()
            rows = da.reshape(da.shape[0], -1).max(dim=1).values
            cols = da.reshape(da.shape[0], -1).max(dim=0).values if a[n].ndim > 1 else rows
            print('%-40s maxdiff %.3e ref %.3e nonfinite %d  bad rows %s bad cols %s' % (
                n, d, r, bad, (rows > 1e-3*r).nonzero().flatten()[:12].tolist(),
                (cols > 1e-3*r).nonzero().flatten()[:12].tolist()))
print('done')
