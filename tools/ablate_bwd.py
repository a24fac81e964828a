"""Where the fused backward kernel (csrc/bwd_fused.cuh) spends its time: per-dilation launch times of the
fused kernel and of the three-launch pair it replaces, then compile-time ablations (-DBF_ABL=bits, results
wrong by construction) built on the GPU box into tools/_libs/abl<bits>/.

    python tools/ablate_bwd.py [bits ...]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, 'brever_amd', 'csrc')


def build(bits):
    out = os.path.join(ROOT, 'tools', '_libs', f'abl{bits}')
    os.makedirs(out, exist_ok=True)
    obj = os.path.join(out, 'convtasnet.o')
    subprocess.run(['/opt/rocm/bin/hipcc', f'-DBF_ABL={bits}', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950',
                    '-c', os.path.join(CSRC, 'convtasnet.hip'), '-o', obj], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    others = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.o') and f != 'convtasnet.o']
    lib = os.path.join(out, 'libbrever_hip.so')
    subprocess.run(['/opt/rocm/bin/hipcc', '-shared', '--offload-arch=gfx950', '-o', lib, obj] + others, check=True)
    return lib


def measure(lib, fuse, by_dil):
    code = f'''
import os, sys
sys.path.insert(0, {ROOT!r})
os.environ['BRV_CTN_STREAMS'] = '1'
os.environ['BRV_BWD_FUSE'] = {fuse!r}
import brever_amd.hip as hip
hip.LIB_PATH = {lib!r}
import torch
from brever_amd.models import ConvTasNet
torch.manual_seed(0)
net = ConvTasNet().cuda()
g = torch.Generator().manual_seed(1)
batch = (0.1*torch.randn(16, 2, 64000, generator=g)).cuda()
lengths = torch.full((16,), 64000).cuda()
for _ in range(3):
    net.train_step(batch, lengths, True, None)
torch.cuda.synchronize()
hip.prof_enable({2 if by_dil else 1})
for _ in range(4):
    net.train_step(batch, lengths, True, None)
torch.cuda.synchronize()
prof = hip.profile_collect()
hip.prof_enable(0)
keys = [k for k in prof if k.startswith(('dwpw2_bwd', 'dwconv_bwd', 'pw2_dgrad', 'pw1_dgrad', 'gu_dots'))]
print(' '.join(f"{{k}}={{prof[k]['ms']/prof[k]['calls']*1e3:.1f}}" for k in sorted(keys)))
'''
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
    return r.stdout.strip() or r.stderr[-400:]


if __name__ == '__main__':
    base = os.path.join(CSRC, 'libbrever_hip.so')
    print('three launches, per dilation:', measure(base, '0', True))
    print('fused, per dilation        :', measure(base, '1', True))
    for bits in [int(a) for a in sys.argv[1:]]:
        print(f'BF_ABL={bits:2d}:', measure(build(bits), '1', True))
