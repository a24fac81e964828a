import sys, torch
sys.path.insert(0, '/root/repo')
from brever_amd.models import ModelRegistry
B = int(sys.argv[1])
dev = torch.device('cuda', 0)
torch.manual_seed(0)
import os
os.environ['BRV_NO_GRAPH'] = '1'
model = ModelRegistry.get('sgmsep')(solver_num_steps=3).to(dev).eval()
wav = 0.1*torch.randn(B, 2, 64000, device=dev)
model.enhance(wav, use_amp=True)
torch.cuda.synchronize()
model.enhance(wav, use_amp=True)
torch.cuda.synchronize()
