"""torch.profiler over SGMSE+ `enhance` (use_amp, HIP graph off): the aten operators (copies, fills, gathers) that still
run between the HIP launches of a network evaluation, with shapes and call sites.   python tools/prof_torch_sgmse.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['BRV_NO_GRAPH'] = '1'
import torch
from torch.profiler import profile, ProfilerActivity
from brever_amd.models import ModelRegistry
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = ModelRegistry.get('sgmsep')(solver_num_steps=2).to(dev).eval()
wav = 0.1*torch.randn(B, 2, 64000, device=dev)
model.enhance(wav, use_amp=True)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    model.enhance(wav, use_amp=True)
    torch.cuda.synchronize()
ev = prof.key_averages(group_by_input_shape=True, group_by_stack_n=6)
rows = [e for e in ev if e.device_time_total > 0 and e.key.startswith('aten::')]
rows.sort(key=lambda e: -e.device_time_total)
print(f'sgmsep enhance batch {B}, 4 network evaluations: aten operators with device time, us total / calls / shapes / stack')
for e in rows[:40]:
    st = [s for s in e.stack if 'brever_amd' in s][:3]
    print(f'{e.device_time_total:9.0f} {e.count:5d} {e.key:24s} {str(e.input_shapes)[:60]:60s} {" <- ".join(s.split("/")[-1][:48] for s in st)}')
