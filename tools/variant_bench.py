"""Build libbrever_hip.so variants with extra -D flags for csrc/convtasnet.hip (on the GPU box, into
tools/_libs/<tag>/) and run bench.py against each (BRV_LIB_PATH):

    python tools/variant_bench.py base= wt16=-DBRV_STORE_AUX=16 nt=-DBRV_STORE_AUX=2
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'brever_amd', 'csrc')


def build(tag, flags):
    out = os.path.join(ROOT, 'tools', '_libs', tag)
    os.makedirs(out, exist_ok=True)
    obj = os.path.join(out, 'convtasnet.o')
    subprocess.run(['/opt/rocm/bin/hipcc'] + flags + ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-c',
                    os.path.join(CSRC, 'convtasnet.hip'), '-o', obj], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    others = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.o') and f != 'convtasnet.o']
    lib = os.path.join(out, 'libbrever_hip.so')
    subprocess.run(['/opt/rocm/bin/hipcc', '-shared', '--offload-arch=gfx950', '-o', lib, obj] + others, check=True)
    return lib


if __name__ == '__main__':
    args = sys.argv[1:]
    if args and args[0] == '--ab':          # launch times per label through tools/ab_step.py instead of bench.py
        labels = args[1]
        ab = []
        for arg in args[2:]:
            tag, _, flags = arg.partition('=')
            ab.append(f'{tag}=BRV_LIB_PATH={build(tag, flags.split() if flags else [])}')
        subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'ab_step.py')] + ab + ['--labels', labels])
        sys.exit(0)
    for arg in args:
        tag, _, flags = arg.partition('=')
        lib = build(tag, flags.split() if flags else [])
        env = dict(os.environ, BRV_LIB_PATH=lib)
        for rep in range(2):
            r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '30', '--warmup', '8', '--no-cpu-baseline', '--no-through-trainer'],
                               capture_output=True, text=True, env=env)
            try:
                d = json.loads(r.stdout.strip().splitlines()[-1])
                print(f'{tag:10s} {d["value"]:8.1f} utt/s  {d["ms_per_step"]:.3f} ms', flush=True)
            except Exception:
                print(tag, 'failed', r.stderr[-300:], flush=True)
