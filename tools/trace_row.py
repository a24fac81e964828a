"""A few training steps of a widened row for `rocprofv3 --kernel-trace`: tools/trace_row.sh lists the launches of the last
step in start order.   python tools/trace_row.py tfgridnet|dccrn [amp|fp32] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brever_amd.models import ModelRegistry
arch = sys.argv[1]
amp = len(sys.argv) < 3 or sys.argv[2] != 'fp32'
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
batch = {'tfgridnet': 4, 'dccrn': 16}[arch]
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = ModelRegistry.get(arch)().to(dev).train()
wav = 0.1*torch.randn(batch, 2, 2, 64000, device=dev)
x = torch.stack([model.transform(w) for w in wav])
lengths = torch.full((batch,), x.shape[-1], device=dev)
scaler = torch.amp.GradScaler('cuda', enabled=False)
for _ in range(steps):
    model.train_step(x, lengths, amp, scaler)
    torch.cuda.synchronize()
