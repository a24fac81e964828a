"""Helper of tests/test_gpu_shapes.py (not a test module): runs in a SUBPROCESS whose library is the variant build
(`BRV_LIB_PATH=tools/_v/variants/libbrever_hip.so`, tools/mkvariant.sh variants -DBRV_WITH_VARIANTS) -- the two
kernel organisations that were measured slower and left the default library in round 6 (csrc/dwpw2_fused_v2.cuh,
csrc/bwd_fused_p.cuh). Each is compared with the bf16-emulating ORACLE (the yardstick of the default kernels,
tests/test_gpu_shapes.py::test_default_width_gradients_at_all_dilations) and with the default kernel of the same
library.

    python tests/variant_check.py <BRV_DWPW2_V2|BRV_BWD_PERSIST> layers repeats B L   -> one JSON line
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch      # noqa: E402


def main():
    switch, layers, repeats, B, L = sys.argv[1], *map(int, sys.argv[2:6])
    from test_gpu_shapes import _detrivialise, _oracle_grads, _ragged_batch, rel
    from brever_amd.criterion import snr
    from brever_amd.models import ConvTasNet
    from oracle.convtasnet import OracleConvTasNet
    cfg = dict(layers=layers, repeats=repeats)
    gen = torch.Generator().manual_seed(11*layers + B)
    torch.manual_seed(29)
    emu = OracleConvTasNet(**cfg, emulate_bf16='fused')
    _detrivialise(emu, gen)
    batch, lengths = _ragged_batch(gen, B, L)
    out_emu, loss_emu, g_emu = _oracle_grads(emu, batch, lengths)
    dev = torch.device('cuda', 0)
    res = {}
    for mode in ('0', '1'):
        os.environ[switch] = mode
        net = ConvTasNet(**cfg)
        net.load_state_dict(emu.state_dict())
        net = net.to(dev)
        net._amp = True
        out = net(batch[:, 0].to(dev))
        loss = snr(out, batch[:, 1:].to(dev), lengths.to(dev)).mean()
        loss.backward()
        g = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu()
        res[mode] = (out.detach().float().cpu(), float(loss), g)
    o0, l0, g0 = res['0']
    o1, l1, g1 = res['1']
    print(json.dumps({
        'finite': bool(torch.isfinite(g1).all() and torch.isfinite(o1).all()),
        'out_vs_oracle': rel(o1, out_emu), 'loss_vs_oracle': abs(l1 - loss_emu), 'grad_vs_oracle': rel(g1, g_emu),
        'default_out_vs_oracle': rel(o0, out_emu), 'default_grad_vs_oracle': rel(g0, g_emu),
        'out_vs_default': rel(o1, o0), 'grad_vs_default': rel(g1, g0)}))


if __name__ == '__main__':
    main()
