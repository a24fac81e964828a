"""The fp32 products of one TCN block (16 x 4 s: 63 984 frames) on brv_gemm_f32, one by one: microseconds
and TFLOP/s per shape, for the library in BRV_LIB_PATH (default: the in-tree one) and for variants of
csrc/gemm_f32_big.hip built here with -DBRV_BIG_ABL=<mask> (1 no MFMA, 2 no global loads, 4 no epilogue
stores):

    python tools/gemm_f32_bench.py            # the library as built
    python tools/gemm_f32_bench.py 1 2 4 6    # + ablations
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, 'brever_amd', 'csrc')


def build(mask):
    out = os.path.join(ROOT, 'tools', '_libs', f'big{mask}')
    os.makedirs(out, exist_ok=True)
    obj = os.path.join(out, 'gemm_f32_big.o')
    subprocess.run(['/opt/rocm/bin/hipcc', f'-DBRV_BIG_ABL={mask}', '-O3', '-std=c++17', '-fPIC',
                    '--offload-arch=gfx950', '-c', os.path.join(CSRC, 'gemm_f32_big.hip'), '-o', obj], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    others = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.o') and f != 'gemm_f32_big.o']
    lib = os.path.join(out, 'libbrever_hip.so')
    subprocess.run(['/opt/rocm/bin/hipcc', '-shared', '--offload-arch=gfx950', '-o', lib, obj] + others, check=True)
    return lib


def run():
    import torch
    from brever_amd import hip
    lib = hip.lib()
    dev = torch.device('cuda')
    BT = 16*3999
    #        name                M    N    K    ta tb  bias acc
    shapes = [('pw1 fwd  128->512', BT, 512, 128, 0, 1, 'col', 0),
              ('res fwd  512->128', BT, 128, 512, 0, 1, None, 1),
              ('skip dgrad 128->512', BT, 512, 128, 0, 0, None, 0),
              ('pw1 dgrad 512->128', BT, 128, 512, 0, 0, None, 1),
              ('pw1 wgrad 512x128', 512, 128, BT, 1, 0, None, 1),
              ('skip wgrad 128x512', 128, 512, BT, 1, 0, None, 1)]
    for name, M, N, K, ta, tb, bias, acc in shapes:
        a = torch.randn((K, M) if ta else (M, K), device=dev)
        b = torch.randn((N, K) if tb else (K, N), device=dev)
        d = torch.zeros(M, N, device=dev)
        bv = torch.randn(N, device=dev)
        def call():
            hip.check(lib.brv_gemm_f32(hip.ptr(a), hip.ptr(b), hip.ptr(d), 1, M, N, K, a.shape[1], b.shape[1], N,
                                       0, 0, 0, ta, tb, 1, 0, 0, hip.ptr(bv) if bias else None,
                                       2 if bias else acc, hip.stream()), 'brv_gemm_f32')
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1)/20*1e3
        print(f'  {name:22s} {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s', flush=True)


if __name__ == '__main__':
    if os.environ.get('BRV_GEMM_BENCH_CHILD'):
        run()
        sys.exit(0)
    variants = [('as built', os.environ.get('BRV_LIB_PATH'))]
    for m in sys.argv[1:]:
        variants.append((f'BRV_BIG_ABL={m}', build(int(m))))
    for tag, lib in variants:
        env = dict(os.environ, BRV_GEMM_BENCH_CHILD='1')
        if lib:
            env['BRV_LIB_PATH'] = lib
        print(tag, flush=True)
        subprocess.run([sys.executable, __file__], env=env, check=False)
