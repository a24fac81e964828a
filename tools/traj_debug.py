"""fp32 trajectory sensitivity: HIP fp32 vs CPU fp32 oracle vs CPU fp64 oracle (debug tool)."""
import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brever_amd.models import ConvTasNet
from oracle.convtasnet import OracleConvTasNet

cfg = dict(layers=2, repeats=2)
torch.manual_seed(7)
o32 = OracleConvTasNet(**cfg)
o64 = OracleConvTasNet(**cfg).double()
o64.load_state_dict({k: v.double() for k, v in o32.state_dict().items()})
net = ConvTasNet(**cfg); net.load_state_dict(o32.state_dict()); net = net.cuda()
scaler = torch.amp.GradScaler('cuda', enabled=False)
gen = torch.Generator().manual_seed(11)
for step in range(20):
    clean = 0.1*torch.randn(4, 3000, generator=gen)
    noise = 0.1*torch.randn(4, 3000, generator=gen)
    snr_db = -5 + 15*torch.rand(4, 1, generator=gen)
    batch = torch.stack([clean + 10**(-snr_db/20)*noise, clean], dim=1)
    lengths = torch.tensor([3000, 2500, 2000, 1600])
    for b in range(4):
        batch[b, :, lengths[b]:] = 0
    a = float(o32.train_step(batch, lengths, False, scaler))
    d = float(o64.train_step(batch.double(), lengths, False, scaler))
    h = float(net.train_step(batch.cuda(), lengths.cuda(), False, scaler))
    print(f'{step:2d} cpu32 {a:.6f} cpu64 {d:.6f} hip32 {h:.6f} |hip-cpu32| {abs(h-a):.2e} |cpu32-cpu64| {abs(a-d):.2e} |hip-cpu64| {abs(h-d):.2e}')
