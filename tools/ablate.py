"""Needs a diagnostic build: make -C brever_amd/csrc clean && make -C brever_amd/csrc DIAG=1.
Ablation of the persistent GEMM: times the forward pass kernels with parts of the
kernel disabled through BRV_DBG (1 no stores, 2 no epilogue, 4 no MFMA, 8 no A loads).
Outputs are wrong by construction; only the timings matter."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brever_amd import hip
from brever_amd.models import ConvTasNet

torch.manual_seed(0)
net = ConvTasNet().cuda()
x = 0.1*torch.randn(16, 64000, device='cuda')
for flags in [0, 1024]:
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
    row = ' '.join(f'{k}={prof[k]["ms"]/prof[k]["calls"]*1e3:6.1f}us' for k in ('pw1_fwd', 'pw2_fwd', 'dwconv_fwd'))
    print(f'dbg={flags:2d}: {row}')

# cycle stamps of the last persistent-GEMM launches of one forward (dbg 64)
import ctypes, numpy as np
os.environ['BRV_DBG'] = os.environ.get('ABL_STAMP', '64')
buf = (ctypes.c_longlong*(256*8*4))()
with torch.no_grad():
    net(x)
hip.lib().brv_debug_read(ctypes.cast(buf, ctypes.c_void_p), ctypes.c_int64(256*8*4))   # clears
# only the final forward GEMM launches write after this: run the network once more, the
# last writer wins per slot
with torch.no_grad():
    net(x)
torch.cuda.synchronize()
hip.lib().brv_debug_read(ctypes.cast(buf, ctypes.c_void_p), ctypes.c_int64(256*8*4))
a = np.array(buf).reshape(-1, 8, 4)
a = a[a[:, 0, 3] > 0]
print('stamps of the LAST ws launch (cycles, median over workgroups; wave 0): stage %d mfma %d epilogue %d total %d over %d WGs'
      % (np.median(a[:, 0, 0]), np.median(a[:, 0, 1]), np.median(a[:, 0, 2]), np.median(a[:, 0, 3]), len(a)))

# wave 1: core-clock ticks and 100 MHz real-time ticks over the whole workgroup
w = a[:, 1, :]
print('WG duration: %d core ticks = %d realtime ticks (100 MHz) -> %.2f us at core clock %.0f MHz'
      % (np.median(w[:, 0]), np.median(w[:, 1]), np.median(w[:, 1])/100.0, np.median(w[:, 0]/w[:, 1])*100))
r0 = w[:, 2].min()
print('realtime entry offsets p50/p100: %.2f/%.2f us; exit p50/p100: %.2f/%.2f us'
      % (np.median(w[:, 2] - r0)/100, (w[:, 2].max() - r0)/100, np.median(w[:, 3] - r0)/100, (w[:, 3].max() - r0)/100))
d = (w[:, 3] - w[:, 2])/100.0
np.set_printoptions(linewidth=200, precision=0, suppress=True)
print('per-WG duration (us) by blockIdx.x:')
print(d)
print('percentiles 10/50/90/99/100:', np.percentile(d, [10, 50, 90, 99, 100]))
ph = a[:, 0, :]
ids = a[:, 2, :]
import collections
print('row ids (KP, gridDim.x, EM, tiles):', collections.Counter(map(tuple, ids.tolist())))
for key in sorted(set(map(tuple, ids.tolist()))):
    sel = (ids == np.array(key)).all(axis=1)
    print(key, 'n=%d' % sel.sum(), 'dur p10/50/90/100 us:', np.percentile(d[sel], [10, 50, 90, 100]),
          'stage %d mfma %d epi %d total %d' % tuple(np.median(ph[sel, k]) for k in range(4)))

# launch boundaries of the last stamped launch (BRV_DBG & 512): realtime ticks (10 ns)
os.environ['BRV_DBG'] = os.environ.get('ABL_STAMP2', '576')
with torch.no_grad():
    net(x)
torch.cuda.synchronize()
big = (ctypes.c_longlong*(131072))()
hip.lib().brv_debug_read(ctypes.cast(big, ctypes.c_void_p), ctypes.c_int64(131072))
rows = np.array(big[:256*8*4]).reshape(-1, 8, 4)
rows = rows[rows[:, 1, 3] > 0][:, 1, :]
t0, t3, ta, tb = big[131000], big[131001], rows[:, 2].min(), rows[:, 3].max()
print('stamp kernel end -> first WG entry: %.2f us; first entry -> last exit: %.2f us; last exit -> next kernel start: %.2f us'
      % ((ta - t0)/100, (tb - ta)/100, (t3 - tb)/100))
