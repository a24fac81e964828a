"""Launch time of the DCCRN use_amp recurrence kernels alone (64 chains of 495 steps, H = 128), HIP events:
   python tools/lstm_mv_bench.py     (BRV_LIB_PATH selects a variant library: tools/mkvariant.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brever_amd import hip
lib = hip.lib()
dev = torch.device('cuda', 0)
G, B, T, H = 4, 16, 495, 128
torch.manual_seed(0)
gates = torch.randn(G, B, T, 4*H, device=dev)
w_hh = torch.randn(G, 4*H, H, device=dev)/H**0.5
bias = torch.zeros(G, 4*H, device=dev)
y = torch.empty(G, B, T, H, device=dev); act = torch.empty(G, B, T, 4*H, device=dev); cs = torch.empty_like(y)
dy = torch.randn_like(y); dg = torch.empty_like(act)


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3


for suffix in ('_bf16', ''):
    f = getattr(lib, 'brv_lstm_recurrent_forward' + suffix); b = getattr(lib, 'brv_lstm_recurrent_backward' + suffix)
    tf = timeit(lambda: hip.check(f(hip.ptr(gates), hip.ptr(w_hh), hip.ptr(bias), hip.ptr(y), hip.ptr(act), hip.ptr(cs),
                                    G*B, T, H, G, hip.stream()), 'fwd'))
    tb = timeit(lambda: hip.check(b(hip.ptr(act), hip.ptr(cs), hip.ptr(w_hh), hip.ptr(dy), hip.ptr(dg), G*B, T, H, G,
                                    hip.stream()), 'bwd'))
    print(f"{'bf16 MFMA' if suffix else 'fp32     '}: forward {tf:7.1f} us ({tf/T*1e3:5.0f} ns/step)   backward {tb:7.1f} us "
          f"({tb/T*1e3:5.0f} ns/step)")
