"""Kernel profile target: DCCRN training steps (BASELINE config 3), python3 tools/prof_dccrn.py [amp]"""
import sys
import torch
sys.path.insert(0, __file__.rsplit('/', 2)[0])
from brever_amd.models import ModelRegistry
amp = len(sys.argv) > 1 and sys.argv[1] == '1'
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = ModelRegistry.get('dccrn')().to(dev).train()
batch = 0.1*torch.randn(16, 2, 64000, device=dev)
lengths = torch.full((16,), 64000, device=dev)
scaler = torch.amp.GradScaler('cuda', enabled=False)
for _ in range(4):
    model.train_step(batch, lengths, amp, scaler)
torch.cuda.synchronize()
