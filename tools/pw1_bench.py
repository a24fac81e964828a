"""The SGMSE+ 1x1 skip convolution alone at its full-resolution shapes (batch 8: 8 x 256 x 251 pixels), HIP events:
   [BRV_LIB_PATH=tools/_v/<tag>/libbrever_hip.so] python tools/pw1_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brever_amd import hip
lib = hip.lib()
dev = torch.device('cuda', 0)
torch.manual_seed(0)
npx = 8*256*251
out = []
for C1, C2, Cout in ((128, 0, 128), (128, 128, 128), (128, 0, 256), (256, 128, 128), (256, 256, 256)):
    x1 = torch.randn(npx, C1, device=dev).half()
    x2 = torch.randn(npx, C2, device=dev).half() if C2 else None
    w = torch.randn(Cout, C1 + C2, device=dev)/16
    bias = torch.randn(Cout, device=dev)
    wp = torch.empty(lib.brv_nhwc_conv1x1_packed_size(Cout, C1, C2), dtype=torch.float16, device=dev)
    hip.check(lib.brv_nhwc_conv1x1_pack(hip.ptr(w), hip.ptr(wp), Cout, C1, C2, hip.stream()), 'pack')
    y = torch.empty(npx, Cout, dtype=torch.float16, device=dev)
    run = lambda: hip.check(lib.brv_nhwc_conv1x1_forward(hip.ptr(x1), C1, C1, hip.ptr(x2), C2, C2, hip.ptr(wp), hip.ptr(bias),
                                                         hip.ptr(y), Cout, npx, Cout, 1.0, hip.stream()), 'fwd')
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)/10*1e3
    mb = npx*(C1 + C2 + Cout)*2/1e6
    out.append(f'{C1}+{C2}->{Cout}: {us:6.1f} us {mb/us*1e-3*1e3:5.2f} TB/s')
print(os.environ.get('BRV_LIB_PATH', 'default'), ' | '.join(out))
