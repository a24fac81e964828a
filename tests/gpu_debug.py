"""GPU diagnostic (not a pytest): per-tensor error report of the HIP Conv-TasNet
against the bf16-emulating oracle, for a golden small config. Usage on the GPU box:
    python tests/gpu_debug.py [small|small2]
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from brever_amd.models import ConvTasNet          # noqa: E402
from oracle.convtasnet import OracleConvTasNet     # noqa: E402


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm()/(b.norm() + 1e-30)), float((a - b).abs().max())


def main(tag='small'):
    g = np.load(os.path.join(ROOT, 'tests', 'golden', f'convtasnet_{tag}.npz'))
    cfg = json.loads(str(g['config']))
    batch = torch.from_numpy(g['batch'])
    lengths = torch.from_numpy(g['lengths'])
    B, _, L = batch.shape
    S = cfg['output_sources']
    oracle = OracleConvTasNet(**cfg, emulate_bf16=True)
    off = 0
    with torch.no_grad():
        for p in oracle.parameters():
            n = p.numel()
            p.copy_(torch.from_numpy(g['params'][off:off + n]).view(p.shape))
            off += n
    oracle.trace = {}
    ref_out = oracle(batch[:, 0])
    loss = oracle.loss(batch, lengths, False)
    loss.backward()
    trace = oracle.trace

    net = ConvTasNet(**cfg)
    net.load_state_dict(oracle.state_dict())
    net = net.cuda()
    x = batch[:, 0].cuda()
    out = net(x)
    torch.cuda.synchronize()
    T = net.frames(L)
    Np = (cfg['filters'] + 63)//64*64
    Bnp = (cfg['bottleneck_channels'] + 63)//64*64
    Hp = (cfg['hidden_channels'] + 63)//64*64
    Scp = (cfg['skip_channels'] + 63)//64*64
    nb = cfg['layers']*cfg['repeats']

    def ws(name, idx, shape, dtype, C):
        t = net.workspace_tensor(name, idx, B, L, shape, dtype).float()
        pad = t[..., C:]
        return t[..., :C], float(pad.abs().max()) if pad.numel() else 0.0

    def report(name, got, want, padmax=0.0):
        r, m = rel(got, want)
        print(f'{name:10s} rel {r:9.2e}  maxabs {m:9.2e}  pad {padmax:.1e}  '
              f'|ref| {float(want.abs().max()):.3e}')

    got, pm = ws('w', 0, (B, T, Np), torch.bfloat16, cfg['filters'])
    report('w', got, trace['w'].transpose(1, 2), pm)
    for i in range(nb):
        got, pm = ws('x', i, (B, T, Bnp), torch.bfloat16, cfg['bottleneck_channels'])
        report(f'x.{i}', got, trace[f'x.{i}'].transpose(1, 2), pm)
        got, pm = ws('z1', i, (B, T, Hp), torch.bfloat16, cfg['hidden_channels'])
        report(f'z1.{i}', got, trace[f'z1.{i}'].transpose(1, 2), pm)
        got, pm = ws('z2', i, (B, T, Hp), torch.bfloat16, cfg['hidden_channels'])
        report(f'z2.{i}', got, trace[f'z2.{i}'].transpose(1, 2), pm)
    got, pm = ws('skip', 0, (B, T, Scp), torch.float32, cfg['skip_channels'])
    report('skip', got, trace['skip'].transpose(1, 2), pm)
    got, pm = ws('m', 0, (B*S, T, Np), torch.bfloat16, cfg['filters'])
    want = trace['m'].view(B, S, cfg['filters'], T).permute(0, 1, 3, 2).reshape(B*S, T, -1)
    report('m', got, want, pm)
    report('out(emu)', out, ref_out.detach())
    report('out(fp32)', out, torch.from_numpy(g['output']))

    # backward through the generic autograd path
    from brever_amd.criterion import snr
    lo = snr(out, batch[:, 1:].cuda(), lengths.cuda()).mean()
    print('loss hip', float(lo), 'oracle(emu)', float(loss), 'ref fp32', float(g['loss']))
    lo.backward()
    torch.cuda.synchronize()
    ref_grads = {n: p.grad for n, p in oracle.named_parameters()}
    gold = torch.from_numpy(g['grads'])
    worst = []
    off = 0
    for (name, p) in net.named_parameters():
        n = p.numel()
        r, m = rel(p.grad, ref_grads[name])
        r2, _ = rel(p.grad, gold[off:off + n].view(p.shape))
        off += n
        worst.append((r, name, m, r2, float(ref_grads[name].norm())))
    worst.sort(reverse=True)
    for r, name, m, r2, nrm in worst[:25]:
        print(f'grad {name:45s} rel(emu) {r:9.2e} rel(fp32) {r2:9.2e} maxabs {m:9.2e} |ref| {nrm:.2e}')
    allg = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu().double()
    print('global grad rel err vs fp32 golden:',
          float((allg - gold.double()).norm()/gold.double().norm()))


if __name__ == '__main__':
    main(*(sys.argv[1:2]))
