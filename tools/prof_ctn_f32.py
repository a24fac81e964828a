"""Kernel profile target: fp32 Conv-TasNet training steps (use_amp=False, BASELINE size), python3 tools/prof_ctn_f32.py"""
import sys
import torch
sys.path.insert(0, __file__.rsplit('/', 2)[0])
from brever_amd.models import ConvTasNet
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = ConvTasNet().to(dev)
batch = 0.1*torch.randn(16, 2, 64000, device=dev)
lengths = torch.full((16,), 64000, device=dev)
scaler = torch.amp.GradScaler('cuda', enabled=False)
for _ in range(3):
    model.train_step(batch, lengths, False, scaler)
torch.cuda.synchronize()
