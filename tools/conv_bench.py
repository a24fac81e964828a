"""Time brv_conv2d_mfma_forward on the shapes of the default SGMSE+ score network.

    python tools/conv_bench.py            # prints us and TFLOP/s per shape
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brever_amd import hip  # noqa: E402

SHAPES = [  # (Cin, Cout, k, H, W)
    (128, 128, 3, 256, 501), (256, 128, 3, 256, 501), (128, 128, 3, 128, 251),
    (256, 256, 3, 64, 126), (512, 256, 3, 64, 126), (256, 256, 3, 32, 63), (256, 256, 3, 16, 32),
    (512, 256, 3, 8, 16), (256, 256, 3, 4, 8), (128, 128, 1, 256, 501), (256, 256, 1, 16, 32),
]


def main():
    dev = torch.device('cuda', 0)
    lib = hip.lib()
    for (ci, co, k, H, W) in SHAPES:
        x = torch.randn(1, ci, H, W, device=dev)
        w = torch.randn(co, ci, k, k, device=dev)*0.05
        bias = torch.zeros(co, device=dev)
        y = torch.empty(1, co, H, W, device=dev)
        wp = torch.empty(lib.brv_conv2d_packed_size(co, ci, k), dtype=torch.float16, device=dev)
        hip.check(lib.brv_conv2d_pack_f16(hip.ptr(w), hip.ptr(wp), co, ci, k, hip.stream()), 'pack')

        def run():
            hip.check(lib.brv_conv2d_mfma_forward(
                hip.ptr(x), hip.ptr(wp), hip.ptr(bias), None, None, None, 0, hip.ptr(y), 1, ci, H, W,
                co, k, ci*H*W, co*H*W, 1.0, hip.stream()), 'conv')
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1)/20*1e3
        flops = 2.0*ci*co*k*k*H*W
        print(f'{ci:4d}->{co:4d} k{k} {H:3d}x{W:3d}: {us:8.1f} us  {flops/us/1e6:7.1f} TFLOP/s')


if __name__ == '__main__':
    main()
