import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brever_amd.batching import BucketBatchSampler
from brever_amd.data import BreverDataLoader, SyntheticMixtureDataset, DevicePrefetcher
print('threads', torch.get_num_threads())
dset = SyntheticMixtureDataset(16*12, 64000, transform=lambda s: s.mean(axis=-2), seed=100)
dset.preload('cpu')
sampler = BucketBatchSampler(dset, batch_size=64.0, dynamic=True, fs=16000)
loader = BreverDataLoader(dataset=dset, batch_sampler=sampler, num_workers=0)
loader.set_epoch(0) if hasattr(loader, 'set_epoch') else None
t0=time.perf_counter()
for b,l in loader: pass
print('loader only: ms/batch', (time.perf_counter()-t0)/len(loader)*1e3)
loader.set_epoch(1)
t0=time.perf_counter(); n=0
for b,l in DevicePrefetcher(loader, 'cuda'):
    n+=1
torch.cuda.synchronize()
print('loader+prefetch: ms/batch', (time.perf_counter()-t0)/n*1e3)
x=torch.randn(16,2,64000)
p=torch.empty(x.numel()).pin_memory()
t0=time.perf_counter()
for _ in range(10): p.view(x.shape).copy_(x)
print('pin copy ms', (time.perf_counter()-t0)/10*1e3)
t0=time.perf_counter()
for _ in range(10): y=p.view(x.shape).to('cuda', non_blocking=True)
torch.cuda.synchronize()
print('h2d ms', (time.perf_counter()-t0)/10*1e3)
torch.set_num_threads(8)
loader.set_epoch(2)
t0=time.perf_counter()
for b,l in loader: pass
print('loader only, 8 threads: ms/batch', (time.perf_counter()-t0)/len(loader)*1e3)

loader.set_epoch(3)
t0=time.perf_counter(); n=0
for b,l in DevicePrefetcher(loader, 'cuda'):
    n+=1
torch.cuda.synchronize()
print('loader+prefetch, 8 threads: ms/batch', (time.perf_counter()-t0)/n*1e3)
torch.set_num_threads(1)
loader.set_epoch(4)
t0=time.perf_counter(); n=0
for b,l in DevicePrefetcher(loader, 'cuda'):
    n+=1
torch.cuda.synchronize()
print('loader+prefetch, 1 thread: ms/batch', (time.perf_counter()-t0)/n*1e3)
