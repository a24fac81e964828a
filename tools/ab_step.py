"""A/B of environment switches on the BASELINE step (16 x 4 s): per-label launch times of the one-chain
step (HIP events) and the wall time of the default two-chain step, each variant in its own process.

    python tools/ab_step.py base= rc0=BRV_PW1_RC=0 [name=VAR=value,VAR2=value ...] [--labels pw1_dgrad,pw1_wgrad]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import os, sys, time
sys.path.insert(0, %(root)r)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch
import brever_amd.hip as hip
from brever_amd.models import ConvTasNet
torch.manual_seed(0)
net = ConvTasNet().cuda()
g = torch.Generator().manual_seed(1)
batch = (0.1*torch.randn(16, 2, 64000, generator=g)).cuda()
lengths = torch.full((16,), 64000).cuda()
scaler = torch.amp.GradScaler('cuda', enabled=False)
def run(n):
    for _ in range(n):
        loss = net.train_step(batch, lengths, True, scaler)
    torch.cuda.synchronize()
    return float(loss)
run(5)
t0 = time.perf_counter(); run(30); two = (time.perf_counter() - t0)/30*1e3
os.environ['BRV_CTN_STREAMS'] = '1'
run(3)
t0 = time.perf_counter(); loss = run(20); one = (time.perf_counter() - t0)/20*1e3
hip.prof_enable(1)
run(4)
prof = hip.profile_collect()
hip.prof_enable(0)
labels = %(labels)r
tot = sum(v['ms'] for v in prof.values())/4
keys = [k for k in prof if (not labels or k in labels)]
keys.sort(key=lambda k: -prof[k]['ms'])
print(f"two-chain {two:.3f} ms ({16/two*1e3:.0f} utt/s)  one-chain {one:.3f} ms  kernels {tot:.3f} ms  loss {loss:.5f}")
print('   ' + '  '.join(f"{k}={prof[k]['ms']/prof[k]['calls']*1e3:.1f}us" for k in keys[:12]))
'''


def main():
    labels = []
    variants = []
    repeat = 1
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == '--labels':
            labels = args.pop(0).split(',')
        elif a == '--repeat':
            repeat = int(args.pop(0))
        else:
            name, _, env = a.partition('=')
            variants.append((name, dict(kv.split('=', 1) for kv in env.split(',') if kv)))
    for rep in range(repeat):                 # interleaved: box drift hits every variant alike
        for name, env in variants:
            e = dict(os.environ)
            e.update(env)
            r = subprocess.run([sys.executable, '-c', CODE % {'root': ROOT, 'labels': labels}], env=e,
                               capture_output=True, text=True)
            print(f'[{name}] ' + (r.stdout.strip() or r.stderr[-600:]), flush=True)


if __name__ == '__main__':
    main()
