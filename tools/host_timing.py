"""Host-side time per train_step (no synchronisation inside the loop) vs GPU time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brever_amd.models import ConvTasNet
import brever_amd.models.convtasnet as ct

torch.manual_seed(0)
net = ConvTasNet().cuda()
batch = 0.1*torch.randn(16, 2, 64000, device='cuda')
lengths = torch.full((16,), 64000, device='cuda')
scaler = torch.amp.GradScaler('cuda', enabled=False)
for _ in range(5):
    net.train_step(batch, lengths, True, scaler)
torch.cuda.synchronize()
N = 20
host = []
t0 = time.perf_counter()
for _ in range(N):
    a = time.perf_counter()
    net.train_step(batch, lengths, True, scaler)
    host.append(time.perf_counter() - a)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('host time per step: mean %.2f ms (min %.2f max %.2f); all issued after %.1f ms; GPU done after %.1f ms (%.2f ms/step)'
      % (1e3*sum(host)/N, 1e3*min(host), 1e3*max(host), 1e3*t_issue, 1e3*t_all, 1e3*t_all/N))
# where does the host time go? instrument the pieces of one step
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    net.train_step(batch, lengths, True, scaler)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
