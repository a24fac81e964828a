"""Two evaluations of the default SGMSE+ score network at batch B under use_amp (no graph replay): the workload of the
SQ-counter passes of tools/profile_sq_rows.sh (the sampler of tools/prof_sgmse.py takes too long under --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ['BRV_NO_GRAPH'] = '1'
from brever_amd.models import ModelRegistry
from brever_amd.models.sgmse import hip_autocast
B = int(sys.argv[1])
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = ModelRegistry.get('sgmsep')(solver_num_steps=30).to(dev).eval()
y = 0.3*torch.randn(B, 1, 256, 501, dtype=torch.complex64, device=dev)
t = torch.tensor(0.5)
with torch.no_grad(), hip_autocast(True):
    for _ in range(3):
        model(y, y, model.sde.sigma(t), t)
torch.cuda.synchronize()
