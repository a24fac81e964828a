"""Per-step wall time of the default DCCRN use_amp train step (16 x 4 s), synchronised after every step: min / median /
p90 over N steps -- the median does not move with the occasional slow step a short back-to-back run averages in.
    python tools/dccrn_step_times.py [steps]        (A/B: set the BRV_DCCRN_* switches in the environment)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brever_amd.models import ModelRegistry  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = ModelRegistry.get('dccrn')().to(dev).train()
    wav = 0.1*torch.randn(16, 2, 2, 64000, device=dev)
    x = torch.stack([model.transform(w) for w in wav])
    lengths = torch.full((16,), 64000, device=dev)
    scaler = torch.amp.GradScaler('cuda', enabled=False)
    for _ in range(8):
        model.train_step(x, lengths, True, scaler)
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        model.train_step(x, lengths, True, scaler)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0)*1e3)
    ts.sort()
    med = ts[len(ts)//2]
    print('dccrn use_amp step, %d steps: min %.2f median %.2f p90 %.2f max %.2f ms -> %.0f utt/s at the median'
          % (n, ts[0], med, ts[int(0.9*len(ts))], ts[-1], 16e3/med))


if __name__ == '__main__':
    main()
