"""Time SGMSE+ inference (BASELINE.json configs[4]: 30-step reverse SDE) on one GPU.

    python tools/sgmse_bench.py [--seconds 4] [--steps 30] [--nfe-only N]
Prints ms per network evaluation and seconds per utterance (PC sampler, 1 corrector step:
2 network evaluations per step)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=4.0)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--nfe-only', type=int, default=0)
    ap.add_argument('--arch', default='sgmsep')
    ap.add_argument('--amp', type=int, default=1)
    ap.add_argument('--batch', type=int, default=1)
    args = ap.parse_args()
    from brever_amd.models import ModelRegistry
    from brever_amd.models.sgmse import hip_autocast
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = ModelRegistry.get(args.arch)(solver_num_steps=args.steps).to(dev).eval()
    L = int(args.seconds*16000)
    wav = 0.1*torch.randn(1, 2, L, device=dev)
    if args.nfe_only:
        T = L//128 + 1
        y = 0.3*torch.randn(args.batch, 1, 256, T, dtype=torch.complex64, device=dev)
        t = torch.tensor(0.5)
        sigma = model.sde.sigma(t)
        with hip_autocast(args.amp):
            model(y, y, sigma, t)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.nfe_only):
                model(y, y, sigma, t)
            host = (time.perf_counter() - t0)/args.nfe_only*1e3     # enqueue time only
            torch.cuda.synchronize()
        ms = (time.perf_counter() - t0)/args.nfe_only*1e3
        print(json.dumps({'ms_per_nfe': ms, 'host_ms_per_nfe': host, 'frames': T}))
        return
    model.enhance(wav[..., :16000], use_amp=bool(args.amp))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.enhance(wav, use_amp=bool(args.amp))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nfe = args.steps*2
    print(json.dumps({'s_per_utt': dt, 'ms_per_nfe': dt/nfe*1e3, 'nfe': nfe,
                      'rtf': dt/args.seconds, 'finite': bool(torch.isfinite(out).all())}))


if __name__ == '__main__':
    main()
