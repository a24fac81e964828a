"""A few default Conv-TasNet train steps (16 x 4 s) for `rocprofv3 --kernel-trace`: tools/trace_step.sh lists the launches of
the last step in start order (which launches the rocclr copy / fill kernels sit between)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brever_amd.models import ConvTasNet
torch.manual_seed(0)
net = ConvTasNet().cuda()
batch = (0.1*torch.randn(16, 2, 64000)).cuda()
lengths = torch.full((16,), 64000).cuda()
scaler = torch.amp.GradScaler('cuda', enabled=False)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    loss = net.train_step(batch, lengths, True, scaler)
torch.cuda.synchronize()
