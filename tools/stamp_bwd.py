"""Where a workgroup of the fused backward stage (csrc/bwd_fused.cuh) spends its lifetime: a diagnostic build
(-DBRV_DIAG -DBF_STAMP, on the GPU box) stamps s_memtime at the phase boundaries of every workgroup of the LAST
launch; shares of the lifetime per phase (median over workgroups). The stamped build is not timed.

    python tools/stamp_bwd.py
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from variant_bench import build          # noqa: E402

lib = build('bfstamp', ['-DBRV_DIAG', '-DBF_STAMP'])
code = f'''
import os, sys, ctypes
sys.path.insert(0, {ROOT!r})
os.environ['BRV_CTN_STREAMS'] = '1'
os.environ['BRV_LIB_PATH'] = {lib!r}
import numpy as np, torch
import brever_amd.hip as hip
from brever_amd.models import ConvTasNet
torch.manual_seed(0)
net = ConvTasNet(layers=1, repeats=2).cuda()        # two blocks of dilation 1: the last launch is block 0's
g = torch.Generator().manual_seed(1)
batch = (0.1*torch.randn(16, 2, 64000, generator=g)).cuda()
lengths = torch.full((16,), 64000).cuda()
for _ in range(3):
    net.train_step(batch, lengths, True, None)
torch.cuda.synchronize()
n = 2048*8
buf = (ctypes.c_longlong*n)()
hip.lib().brv_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int64]
hip.lib().brv_debug_read(ctypes.cast(buf, ctypes.c_void_p), n)
a = np.array(buf[:], dtype=np.int64).reshape(-1, 8)
a = a[a[:, 0] != 0]
names = ['phase 0 (g loads, W^T g on the matrix pipe)', 'barrier + parameter table', 'phase 1 (z2 loads, dz2 in the window)',
         'folds + atomics + barrier', 'phase 2 (z1 loads, stencil, e1 stores)', 'folds + atomics (end)']
d = np.diff(a[:, :7], axis=1).astype(np.float64)
life = d.sum(axis=1)
print(f'{{len(a)}} workgroups stamped, median lifetime {{np.median(life):.0f}} cycles')
for i, nm in enumerate(names):
    print(f'  {{np.median(d[:, i]):8.0f}} cycles  {{np.median(d[:, i]/life)*100:5.1f}} %  {{nm}}')
span = a[:, 6].max() - a[:, 0].min()
print(f'launch span {{span}} cycles; sum of lifetimes / span / 256 CUs = {{life.sum()/span/256:.2f}} resident workgroups per CU')
'''
r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
print(r.stdout or r.stderr[-2000:])
