"""The N > 1 step shape on ONE GPU: a world-1 RCCL process group, so that the backward pass runs in
`--buckets` parts with one all-reduce per part (bench.py only does this for world > 1).
usage: python tools/bench_parts.py [--buckets 3] [--steps 20] [--kernel-table]"""
import argparse, os, sys, time
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')     # before torch loads the HIP runtime (brever_amd/__init__.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
from brever_amd import hip
from brever_amd.models import ConvTasNet
from brever_amd.parallel import GradSynchronizer, broadcast_parameters

ap = argparse.ArgumentParser()
ap.add_argument('--buckets', type=int, default=3)
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--warmup', type=int, default=5)
ap.add_argument('--kernel-table', action='store_true')
args = ap.parse_args()
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
device = torch.device('cuda', 0)
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=device)
torch.manual_seed(0)
model = ConvTasNet().to(device)
broadcast_parameters(model)
sync = GradSynchronizer(model, nparts=args.buckets) if args.buckets > 0 else None
scaler = torch.amp.GradScaler('cuda', enabled=False)
batches = bench.make_batches(4, 0, device)
def step(i):
    b, l = batches[i % len(batches)]
    return model.train_step(b, l, True, scaler)
for i in range(args.warmup):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(args.steps):
    step(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0)/args.steps
print(f'buckets {args.buckets}: {dt*1e3:.3f} ms/step, {16/dt:.1f} utt/s')
if args.kernel_table:
    hip.prof_enable(1)
    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    prof = hip.profile_collect()
    hip.prof_enable(0)
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])[:10]:
        print(f"{k:20s} {v['calls']/5:5.1f} calls/step {v['ms']/5:7.3f} ms/step")
dist.destroy_process_group()
