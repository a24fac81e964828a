"""Compile-time ablations of the weight-stationary first-conv data gradient (csrc/pw1_bwd.cuh, -DWSD_ABL=bits;
results wrong by construction), built on the GPU box into tools/_libs/wsd<bits>/ and timed with tools/ab_step.py.

    python tools/ablate_pw1.py 0 1 2 4 8 16 32 [ENV=VALUE ...]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'brever_amd', 'csrc')


def build(bits):
    out = os.path.join(ROOT, 'tools', '_libs', f'wsd{bits}')
    os.makedirs(out, exist_ok=True)
    obj = os.path.join(out, 'convtasnet.o')
    subprocess.run(['/opt/rocm/bin/hipcc', f'-DWSD_ABL={bits}', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950',
                    '-c', os.path.join(CSRC, 'convtasnet.hip'), '-o', obj], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    others = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.o') and f != 'convtasnet.o']
    lib = os.path.join(out, 'libbrever_hip.so')
    subprocess.run(['/opt/rocm/bin/hipcc', '-shared', '--offload-arch=gfx950', '-o', lib, obj] + others, check=True)
    return lib


if __name__ == '__main__':
    bits = [a for a in sys.argv[1:] if '=' not in a]
    env = ','.join(a for a in sys.argv[1:] if '=' in a)
    args = []
    for b in bits:
        lib = build(int(b))
        args.append(f'abl{b}=BRV_LIB_PATH={lib}' + (',' + env if env else ''))
    subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'ab_step.py')] + args + ['--labels', 'pw1_dgrad,pw1_wgrad'])
