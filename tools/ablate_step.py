"""Diagnostic build (BRV_LIB_PATH=tools/_libs/diag/libbrever_hip.so): per-label times of one training
step with parts of the persistent GEMMs disabled through BRV_DBG (1 no epilogue stores, 2 no
epilogue, 4 no MFMA, 8 no A loads, 1024 no statistics atomics). Results are wrong by construction."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brever_amd import hip
from brever_amd.models import ConvTasNet
torch.manual_seed(0)
net = ConvTasNet().cuda()
batch = 0.1*torch.randn(16, 2, 64000, device='cuda')
lengths = torch.full((16,), 64000, device='cuda')
labels = ('pw1_fwd', 'dwconv_fwd', 'pw2_fwd', 'pw2_dgrad', 'dwconv_bwd', 'pw1_dgrad', 'pw2_wgrad', 'pw1_wgrad')
for flags in [0, 1, 2, 4, 8, 12, 1024]:
    os.environ['BRV_DBG'] = str(flags)
    for _ in range(2):
        net.train_step(batch, lengths, True, None)
    torch.cuda.synchronize()
    hip.prof_enable(1)
    for _ in range(3):
        net.train_step(batch, lengths, True, None)
    torch.cuda.synchronize()
    prof = hip.profile_collect()
    hip.prof_enable(0)
    print(f'dbg={flags:4d}: ' + ' '.join(f'{k}={prof[k]["ms"]/prof[k]["calls"]*1e3:6.1f}' for k in labels if k in prof))
