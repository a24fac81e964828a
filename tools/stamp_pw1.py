"""Where the producer / consumer waves of the weight-stationary first-conv data gradient (csrc/pw1_bwd.cuh:
pw1_dgrad_ws_kernel) spend a tile: a diagnostic build (-DWSD_STAMP, on the GPU box) stamps s_memtime at the top of
every iteration, in front of the barrier (after the staged operands have arrived) and behind it, for wave 0 and
wave 4 of every workgroup of the LAST launch (-DBRV_DIAG -DWSD_STAMP). Cycles are s_memtime ticks (100 MHz constant clock).

    python tools/stamp_pw1.py
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from variant_bench import build          # noqa: E402

lib = build('wsdstamp', ['-DBRV_DIAG', '-DWSD_STAMP'])
code = f'''
import os, sys, ctypes
sys.path.insert(0, {ROOT!r})
os.environ['BRV_CTN_STREAMS'] = '1'
os.environ['BRV_LIB_PATH'] = {lib!r}
import numpy as np, torch
import brever_amd.hip as hip
from brever_amd.models import ConvTasNet
torch.manual_seed(0)
net = ConvTasNet(layers=1, repeats=2).cuda()
g = torch.Generator().manual_seed(1)
batch = (0.1*torch.randn(16, 2, 64000, generator=g)).cuda()
lengths = torch.full((16,), 64000).cuda()
for _ in range(3):
    net.train_step(batch, lengths, True, None)
torch.cuda.synchronize()
n = 256*64
buf = (ctypes.c_longlong*n)()
hip.lib().brv_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int64]
hip.lib().brv_debug_read(ctypes.cast(buf, ctypes.c_void_p), n)
a = np.array(buf[:], dtype=np.int64).reshape(-1, 2, 32)
a = a[a[:, 0, 0] != 0]
print(len(a), 'workgroups stamped')
for role, name in ((0, 'producer wave 0'), (1, 'consumer wave 4')):
    s = a[:, role, :].astype(np.float64)
    nit = int(((s[:, 2:] != 0).sum(axis=1).min())//3)
    print(name, 'iterations stamped', nit, ' prologue (start -> first iteration top)', np.median(s[:, 2] - s[:, 0]))
    tops = s[:, 2:2 + 3*nit:3]; pre = s[:, 3:3 + 3*nit:3]; post = s[:, 4:4 + 3*nit:3]
    print('   median per iteration: top -> operands in LDS', np.median(pre - tops), ' barrier wait', np.median(post - pre),
          ' barrier -> next top (compute)', np.median(tops[:, 1:] - post[:, :-1]))
    print('   per iteration (median over workgroups): wait-for-operands', np.round(np.median(pre - tops, axis=0)), ' barrier', np.round(np.median(post - pre, axis=0)))
    print('   lifetime', np.median(post[:, -1] - s[:, 0]))
'''
r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
print(r.stdout or r.stderr[-2000:])
