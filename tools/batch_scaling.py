"""Fixed cost of the Conv-TasNet training step: ms per step against the batch size (4 s utterances), one kernel
chain and the default two chains, and the straight-line fit t(B) = a + b B over B >= 16. `a` is what a step pays
whatever its size -- ~100 launches' boundaries, drains and per-workgroup prologues (weights, tables) -- i.e. the
most a persistent, flag-synchronised form of the step could remove; `b` is the marginal cost
of an utterance. Also per-label launch times at B = 16 and B = 64 (one chain).

    python tools/batch_scaling.py [--sizes 4,8,16,32,48,64]
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch                                        # noqa: E402

import brever_amd.hip as hip                        # noqa: E402
from brever_amd.models import ConvTasNet            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sizes', default='4,8,16,32,48,64')
    args = ap.parse_args()
    sizes = [int(v) for v in args.sizes.split(',')]
    torch.manual_seed(0)
    net = ConvTasNet().cuda()
    scaler = torch.amp.GradScaler('cuda', enabled=False)
    g = torch.Generator().manual_seed(1)
    out = {'one_chain': {}, 'two_chains': {}, 'labels': {}}

    def run(batch, lengths, n):
        for _ in range(n):
            net.train_step(batch, lengths, True, scaler)
        torch.cuda.synchronize()

    for B in sizes:
        batch = (0.1*torch.randn(B, 2, 64000, generator=g)).cuda()
        lengths = torch.full((B,), 64000).cuda()
        for mode, streams in (('two_chains', None), ('one_chain', '1')):
            if streams is None:
                os.environ.pop('BRV_CTN_STREAMS', None)
            else:
                os.environ['BRV_CTN_STREAMS'] = streams
            run(batch, lengths, 4)
            n = max(6, int(0.6/(0.0004*B + 0.001)))
            t0 = time.perf_counter()
            run(batch, lengths, n)
            out[mode][B] = (time.perf_counter() - t0)/n*1e3
        if B in (16, 64):
            hip.prof_enable(1)
            run(batch, lengths, 3)
            prof = hip.profile_collect()
            hip.prof_enable(0)
            out['labels'][B] = {k: v['ms']/v['calls']*1e3 for k, v in prof.items()
                                if k in ('dwpw2_bwd', 'pw1_dgrad', 'dwpw2_fwd', 'pw1_fwd', 'pw2_wgrad', 'pw1_wgrad')}
        os.environ.pop('BRV_CTN_STREAMS', None)
        del batch
        torch.cuda.empty_cache()

    def fit(d):
        xs = [b for b in d if b >= 16]
        n = len(xs)
        mx, my = sum(xs)/n, sum(d[b] for b in xs)/n
        slope = sum((b - mx)*(d[b] - my) for b in xs)/sum((b - mx)**2 for b in xs)
        return {'fixed_ms_per_step': my - slope*mx, 'ms_per_utterance': slope,
                'marginal_utt_per_s': 1e3/slope}
    out['fit_one_chain'] = fit(out['one_chain'])
    out['fit_two_chains'] = fit(out['two_chains'])
    for mode in ('one_chain', 'two_chains'):
        print(mode, '  '.join(f'B={b}: {t:.3f} ms ({b/t*1e3:.0f} utt/s)' for b, t in out[mode].items()))
    print('fit one chain :', out['fit_one_chain'])
    print('fit two chains:', out['fit_two_chains'])
    for B, lab in out['labels'].items():
        print(f'labels B={B} (us per launch):', '  '.join(f'{k}={v:.1f}' for k, v in lab.items()))
    print(json.dumps(out))


if __name__ == '__main__':
    main()
