"""Is a row's training step bound by the host? Time to ENQUEUE a step (train_step returns, no synchronise) against the time
of the synchronised step.   python tools/host_bound.py tfgridnet|dccrn [amp|fp32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brever_amd.models import ModelRegistry
arch = sys.argv[1]
amp = len(sys.argv) < 3 or sys.argv[2] != 'fp32'
batch = {'tfgridnet': 4, 'dccrn': 16}[arch]
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = ModelRegistry.get(arch)().to(dev).train()
wav = 0.1*torch.randn(batch, 2, 2, 64000, device=dev)
x = torch.stack([model.transform(w) for w in wav])
lengths = torch.full((batch,), x.shape[-1], device=dev)
scaler = torch.amp.GradScaler('cuda', enabled=False)
for _ in range(3):
    model.train_step(x, lengths, amp, scaler)
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.train_step(x, lengths, amp, scaler)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0)*1e3); tot.append((t2 - t0)*1e3)
t0 = time.perf_counter()
for _ in range(8):
    model.train_step(x, lengths, amp, scaler)
torch.cuda.synchronize()
back = (time.perf_counter() - t0)/8*1e3
print(f'{arch} amp={amp}: enqueue {min(enq):.2f} ms (median {sorted(enq)[4]:.2f}), synchronised step {min(tot):.2f} ms, '
      f'back-to-back {back:.2f} ms per step')
