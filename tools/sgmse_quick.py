"""SGMSE+ use_amp enhance at batch 1 and 8 (10-step sampler = 20 network evaluations; ms per evaluation is what counts):
   [BRV_LIB_PATH=tools/_v/<tag>/libbrever_hip.so] python tools/sgmse_quick.py     (variants: tools/mkvariant.sh <tag> conv_nhwc.hip -DBRV_CONV_MIN_TILES=n)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brever_amd.models import ModelRegistry
dev = torch.device('cuda', 0)
torch.manual_seed(0)
steps = 10
model = ModelRegistry.get('sgmsep')(solver_num_steps=steps).to(dev).eval()
res = []
for batch in (1, 8):
    wav = 0.1*torch.randn(batch, 2, 64000, device=dev)
    model.enhance(wav, use_amp=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        model.enhance(wav, use_amp=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0)/2
    res.append(f'b{batch}: {dt/(2*steps)*1e3:6.2f} ms per evaluation')
print(os.environ.get('BRV_LIB_PATH', 'default library'), ' | '.join(res))
