"""Diagnostic build only (tools/_libs/diag): ablation of the fused forward kernels through BRV_DBG
(1 no stores, 2 no epilogue, 4 no MFMA, 8 no A loads). Outputs are wrong by construction."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brever_amd.hip as hip
hip.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_libs', 'diag', 'libbrever_hip.so')
import torch
from brever_amd.models import ConvTasNet
torch.manual_seed(0)
net = ConvTasNet().cuda(); net._amp = True
x = 0.1*torch.randn(16, 64000, device='cuda')
for flags in [0, 1, 2, 4, 8, 12, 13]:
    os.environ['BRV_DBG'] = str(flags)
    with torch.no_grad():
        for _ in range(2):
            net(x)
        torch.cuda.synchronize()
        hip.prof_enable(1)
        for _ in range(3):
            net(x)
        torch.cuda.synchronize()
    prof = hip.profile_collect()
    hip.prof_enable(0)
    row = ' '.join(f'{k}={prof[k]["ms"]/prof[k]["calls"]*1e3:6.1f}us' for k in prof if k in ('pw1_fwd', 'dwpw2_fwd', 'pw2_fwd', 'dwconv_fwd'))
    print(f'dbg={flags:2d}: {row}')
