"""torch.profiler over the training step of a widened row: which aten operators (copies, permutes, fills, adds) still run
between the HIP launches, with shapes and Python call sites.   python tools/prof_torch.py tfgridnet|dccrn [amp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from brever_amd.models import ModelRegistry
arch = sys.argv[1]
amp = len(sys.argv) < 3 or sys.argv[2] != 'fp32'
batch = {'tfgridnet': 4, 'dccrn': 16}[arch]
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = ModelRegistry.get(arch)().to(dev).train()
wav = 0.1*torch.randn(batch, 2, 2, 64000, device=dev)
x = torch.stack([model.transform(w) for w in wav])
lengths = torch.full((batch,), x.shape[-1], device=dev)
scaler = torch.amp.GradScaler('cuda', enabled=False)
for _ in range(3):
    model.train_step(x, lengths, amp, scaler)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(2):
        model.train_step(x, lengths, amp, scaler)
    torch.cuda.synchronize()
ev = prof.key_averages(group_by_input_shape=True, group_by_stack_n=4)
rows = [e for e in ev if e.device_time_total > 0 and e.key.startswith('aten::')]
rows.sort(key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in prof.key_averages() if e.device_type is not None and not e.key.startswith('aten::') and e.device_time_total > 0)
print(f'{arch} amp={amp}: aten operators with device time (2 steps), us total / calls / shapes / stack')
for e in rows[:40]:
    st = [s for s in e.stack if 'brever_amd' in s][:2]
    print(f'{e.device_time_total:9.0f} {e.count:5d} {e.key:28s} {str(e.input_shapes)[:70]:70s} {" <- ".join(s.split("/")[-1][:60] for s in st)}')
