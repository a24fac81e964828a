"""Time of every row-convolution launch of the default DCCRN step (B = 16, 4 s) by itself, bf16 or fp32 images:

    python tools/cconv_bench.py [bf16|f32] [B]

forward and data-gradient launches of brv_cconv_rows[_bf16] and the weight-gradient launches of brv_cconv_wgrad[_bf16]
with the shapes models/dccrn.py issues (decoder inputs from their two sources). BRV_LIB_PATH selects a variant library
(tools/mkvariant.sh <tag> cconv.hip -DCC_ABL=<bits>: 1 no image loads, 2 no weight loads, 4 no MFMAs, 8 no stores).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brever_amd import hip  # noqa: E402

CH = [16, 32, 64, 128, 128, 128]


ONLY = os.environ.get('LAYER')          # e.g. LAYER=dec4: that layer only (profiling runs)


def timed(fn, n=10):
    if ONLY:
        n = 2
    for _ in range(1 if ONLY else 3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3


def main():
    lowp = (sys.argv[1] if len(sys.argv) > 1 else 'bf16') == 'bf16'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    dev = torch.device('cuda', 0)
    lib = hip.lib()
    dt = torch.bfloat16 if lowp else torch.float32
    rows = lib.brv_cconv_rows_bf16 if lowp else lib.brv_cconv_rows
    wgrad = lib.brv_cconv_wgrad_bf16 if lowp else lib.brv_cconv_wgrad

    def img(*shape):
        if not lowp:
            return torch.randn(*shape, device=dev)
        from brever_amd.models.dccrn import _as_bf16      # (with the readable slack the LDS-DMA kernels ask for)
        return _as_bf16(torch.randn(*shape, device=dev))

    def conv(x, x2, seg, M, C, transposed, split):
        Bn, _, H, W = x.shape
        wp = torch.zeros(lib.brv_cconv_packed_bytes(M, C), dtype=torch.uint8, device=dev)
        shape = (Bn, M//2 if split else M) + ((2*H, W + 1) if transposed else (H//2, W - 1))
        out = torch.empty(shape, device=dev)
        out2 = torch.empty_like(out) if split else None
        return timed(lambda: hip.check(rows(hip.ptr(x), hip.ptr(x2), seg, hip.ptr(wp), None, hip.ptr(out), hip.ptr(out2),
                                            M//4 if split else 0, Bn, C, M, H, W, int(transposed), hip.stream()), 'rows'))

    def wg(small, small2, big):
        Bn, C, Hb, Wb = big.shape
        seg = small.shape[1]//2 if small2 is not None else 0
        A = 4*seg if seg else small.shape[1]
        Hs, Ws = small.shape[2:]
        out = torch.empty(A, 10*C, device=dev)
        ws = torch.empty(lib.brv_cconv_wgrad_workspace_bytes(Bn, A, C, Hs), dtype=torch.uint8, device=dev)
        return timed(lambda: hip.check(wgrad(hip.ptr(small), hip.ptr(small2), hip.ptr(big), hip.ptr(out), hip.ptr(ws),
                                             Bn, A, C, Hs, Ws, seg, hip.stream()), 'wgrad'))

    H, W = 256, 501
    tot = [0.0, 0.0, 0.0]
    enc = []
    for i, c in enumerate(CH):
        cin = 2 if i == 0 else 2*CH[i - 1]
        Ho, Wo = H//2, W - 1
        if ONLY and ONLY != 'enc%d' % (i + 1):
            enc.append((2*c, Ho, Wo))
            H, W = Ho, Wo
            continue
        x = img(B, cin, H, W)
        dy = img(B, 2*c, Ho, Wo)
        f = conv(x, None, 0, 2*c, cin, 0, False)
        d = conv(dy, None, 0, cin, 2*c, 1, False) if i else 0.0
        w = wg(dy, None, x) if cin >= 8 else 0.0
        print(f'enc{i + 1} ({cin:3d} -> {2*c:3d}, {H:3d} x {W})  fwd {f:7.1f}  dgrad {d:7.1f}  wgrad {w:7.1f} us', flush=True)
        tot = [tot[0] + f, tot[1] + d, tot[2] + w]
        enc.append((2*c, Ho, Wo))
        H, W = Ho, Wo
    for i in range(len(CH) - 1, -1, -1):
        cout = 2 if i == 0 else 2*CH[i - 1]
        c2, H, W = enc[i]
        if ONLY and ONLY != 'dec%d' % (i + 1):
            continue
        x, skip = img(B, c2, H, W), img(B, c2, H, W)
        dy = img(B, cout, 2*H, W + 1)
        f = conv(x, skip, c2//2, cout, 2*c2, 1, False)
        d = conv(dy, None, 0, 2*c2, cout, 0, True)
        w = wg(x, skip, dy)
        print(f'dec{i + 1} ({2*c2:3d} -> {cout:3d}, {H:3d} x {W})  fwd {f:7.1f}  dgrad {d:7.1f}  wgrad {w:7.1f} us', flush=True)
        tot = [tot[0] + f, tot[1] + d, tot[2] + w]
    print(f'total fwd {tot[0]:.0f}  dgrad {tot[1]:.0f}  wgrad {tot[2]:.0f}  all {sum(tot):.0f} us', flush=True)


if __name__ == '__main__':
    main()
