"""Per-kernel event timings of one fused training step at several input lengths
(tools only; prints the table of bench.py's kernel_roofline for each length)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brever_amd import hip
from brever_amd.models import ConvTasNet

torch.manual_seed(0)
net = ConvTasNet().cuda()
for L in [int(x) for x in sys.argv[1:]] or [5200, 64000]:
    batch = 0.1*torch.randn(16, 2, L, device='cuda')
    lengths = torch.full((16,), L, device='cuda')
    for _ in range(3):
        net.train_step(batch, lengths, True, None)
    torch.cuda.synchronize()
    hip.prof_enable(1)
    for _ in range(3):
        net.train_step(batch, lengths, True, None)
    torch.cuda.synchronize()
    prof = hip.profile_collect()
    hip.prof_enable(0)
    print(f'--- L={L} T={net.frames(L)}')
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])[:12]:
        print(f'{k:18s} {v["ms"]/v["calls"]*1e3:8.1f} us/call  {v["bytes"]/v["ms"]/1e6:8.1f} GB/s')
