"""Data-parallel path on CPU: 2 processes, gloo backend (RCCL's stand-in), 127.0.0.1.

The reference's DDP never synchronises gradients, so
parity is defined against a single process on the union batch."""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import DummyDataset, DummyModel


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _spawn(worker, world, tmp):
    """mp.spawn with a fresh rendezvous port; one retry (a probed free port can be taken by
    another process between the probe and the bind)."""
    for attempt in range(2):
        try:
            mp.spawn(worker, args=(world, _free_port(), tmp), nprocs=world, join=True)
            return
        except Exception:
            if attempt == 1:
                raise


def _init(rank, world, port):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)


def _make_batch():
    g = torch.Generator().manual_seed(3)
    batch = torch.randn(4, 3, 2, 500, generator=g)      # (B, sources, channels, L)
    lengths = torch.tensor([500, 500, 500, 500])
    return batch, lengths


def _worker_generic(rank, world, port, out_dir):
    from brever_amd.parallel import GradSynchronizer, broadcast_parameters
    _init(rank, world, port)
    torch.manual_seed(100 + rank)                        # different init per rank ...
    model = DummyModel(channels=2, output_sources=2)
    broadcast_parameters(model)                          # ... until the broadcast
    sync = GradSynchronizer(model)
    assert not sync.flat_model
    batch, lengths = _make_batch()
    lo, hi = rank*2, rank*2 + 2
    scaler = torch.amp.GradScaler('cuda', enabled=False)
    for _ in range(3):
        sync.train_step(model, batch[lo:hi], lengths[lo:hi], False, scaler)
    torch.save([p.detach().clone() for p in model.parameters()],
               os.path.join(out_dir, f'params{rank}.pt'))
    dist.destroy_process_group()


def test_gradient_allreduce_equals_union_batch():
    world = 2
    with tempfile.TemporaryDirectory() as tmp:
        _spawn(_worker_generic, world, tmp)
        p0 = torch.load(os.path.join(tmp, 'params0.pt'))
        p1 = torch.load(os.path.join(tmp, 'params1.pt'))
    # single process, union batch, same initial weights as rank 0
    torch.manual_seed(100)
    model = DummyModel(channels=2, output_sources=2)
    batch, lengths = _make_batch()
    scaler = torch.amp.GradScaler('cuda', enabled=False)
    for _ in range(3):
        model.train_step(batch, lengths, False, scaler)
    for a, b, ref in zip(p0, p1, model.parameters()):
        assert torch.equal(a, b)                         # ranks stay in lock-step
        assert torch.allclose(a, ref.detach(), rtol=1e-5, atol=1e-6)


def _worker_flat_base_generic(rank, world, port, out_dir):
    """A base-class model WITH a flat buffer whose optimizer is not FlatAdam: `set_grad_sync` accepts, no hooks,
    and the generic branch of `update` averages the .grad tensors through ONE packed collective per step."""
    from brever_amd.parallel import GradSynchronizer, broadcast_parameters
    _init(rank, world, port)
    torch.manual_seed(100 + rank)
    model = DummyModel(channels=2, output_sources=2)
    model._flat_base = True
    model._flatten_base()
    model.optimizer = torch.optim.Adam(model.parameters())
    broadcast_parameters(model)                          # (one broadcast of the flat buffer)
    sync = GradSynchronizer(model)
    assert sync.flat_model and model._grad_sync is sync
    assert not any(getattr(p, '_post_accumulate_grad_hooks', None) for p in model.parameters())
    batch, lengths = _make_batch()
    lo, hi = rank*2, rank*2 + 2
    scaler = torch.amp.GradScaler('cuda', enabled=False)
    for _ in range(3):
        sync.train_step(model, batch[lo:hi], lengths[lo:hi], False, scaler)
    assert sync.calls == 3
    torch.save([p.detach().clone() for p in model.parameters()],
               os.path.join(out_dir, f'params{rank}.pt'))
    dist.destroy_process_group()


def test_flat_base_model_generic_branch_equals_union_batch():
    """models with `_flat_base` get one collective per step, not one per parameter."""
    world = 2
    with tempfile.TemporaryDirectory() as tmp:
        _spawn(_worker_flat_base_generic, world, tmp)
        p0 = torch.load(os.path.join(tmp, 'params0.pt'))
        p1 = torch.load(os.path.join(tmp, 'params1.pt'))
    torch.manual_seed(100)
    model = DummyModel(channels=2, output_sources=2)
    batch, lengths = _make_batch()
    scaler = torch.amp.GradScaler('cuda', enabled=False)
    for _ in range(3):
        model.train_step(batch, lengths, False, scaler)
    for a, b, ref in zip(p0, p1, model.parameters()):
        assert torch.equal(a, b)
        assert torch.allclose(a, ref.detach(), rtol=1e-5, atol=1e-6)


class _FlatStub(torch.nn.Module):
    """Mimics the flat-gradient protocol of the HIP Conv-TasNet on CPU."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(7))
        self._sync = None
        self.flat = torch.zeros(7)

    def set_grad_sync(self, fn):
        self._sync = fn

    def flat_params(self):
        return self.w.data

    def mark_params_changed(self):
        pass

    def train_step(self, batch, lengths, use_amp, scaler):
        """Same protocol as ConvTasNet.train_step: backward in ``nparts`` parts with the
        bucket hook after each one (here part p "computes" its slice of the gradient)."""
        sync = self._sync
        self.flat.zero_()
        nparts = getattr(sync, 'nparts', 1)
        if nparts > 1:
            bounds = [7*p//nparts for p in range(nparts + 1)]
            for part in range(nparts):          # backward order: last slice first
                lo, hi = bounds[nparts - 1 - part], bounds[nparts - part]
                self.flat[lo:hi] = batch[lo:hi]
                if hi > lo:
                    sync.bucket(part, self.flat[lo:hi])
            scale = sync.finish()
        else:
            self.flat.copy_(batch)
            scale = sync(self.flat)
        return self.flat*scale


def _worker_flat(rank, world, port, out_dir):
    from brever_amd.parallel import GradSynchronizer, broadcast_parameters
    _init(rank, world, port)
    model = _FlatStub()
    with torch.no_grad():
        model.w.fill_(float(rank + 1))
    broadcast_parameters(model)
    assert torch.all(model.w == 1.0)
    g = torch.Generator().manual_seed(17 + rank)
    grad = torch.randn(7, generator=g)
    results = []
    for nparts in (1, 3, 7, 9):                  # single buffer, buckets, more parts than blocks
        sync = GradSynchronizer(model, nparts=nparts)
        assert sync.flat_model and sync.nparts == nparts
        results.append(sync.train_step(model, grad, None, False, None).clone())
    both = [torch.randn(7, generator=torch.Generator().manual_seed(17 + r)) for r in range(world)]
    assert torch.equal(results[0], (both[0] + both[1])*0.5)
    for r in results[1:]:
        assert torch.equal(r, results[0])        # bucketed == single buffer, bit for bit
    torch.save(results[0], os.path.join(out_dir, f'mean{rank}.pt'))
    dist.destroy_process_group()


def test_flat_gradient_hook_sums_and_scales():
    """Single-bucket and bucketed (overlapped) all-reduce of a flat gradient: the mean over
    ranks, identical bit for bit whatever the number of buckets, identical on both ranks."""
    world = 2
    with tempfile.TemporaryDirectory() as tmp:
        _spawn(_worker_flat, world, tmp)
        assert torch.equal(torch.load(os.path.join(tmp, 'mean0.pt')),
                           torch.load(os.path.join(tmp, 'mean1.pt')))


def _worker_trainer(rank, world, port, out_dir):
    from brever_amd.training import BreverTrainer
    _init(rank, world, port)
    torch.manual_seed(rank)
    model = DummyModel(channels=2, output_sources=2)
    FS = 16000
    train = DummyDataset(16, 3, 2, FS//2, FS*2, transform=model.transform)
    val = DummyDataset(4, 3, 2, FS//2, FS*2)
    trainer = BreverTrainer(
        model=model, train_dataset=train, val_dataset=val,
        model_dirpath=os.path.join(out_dir, 'model'), epochs=2, val_period=1,
        val_metrics=set(), batch_sampler='bucket', batch_size=4.0,
        dynamic_batch_size=True, device='cpu', preload=True, ddp=True, rank=rank)
    trainer.run()
    torch.save([p.detach().clone() for p in model.parameters()],
               os.path.join(out_dir, f'params{rank}.pt'))
    if rank == 0:
        assert len(trainer.loss_logger.train_loss) == 2
        assert os.path.exists(os.path.join(out_dir, 'model', 'checkpoints', 'last.ckpt'))
    dist.destroy_process_group()


def _worker_trainer_numpy_metric(rank, world, port, out_dir):
    """A validation metric that hands back a NumPy array, as stoi / estoi do (the packed
    all-reduce of the logged values called .detach() on np.float64 and lost a default `ddp: true`
    run at its first validation epoch)."""
    import numpy as np

    from brever_amd.metrics import MetricRegistry
    from brever_amd.training import BreverTrainer
    _init(rank, world, port)

    @MetricRegistry.register('np_energy')
    def np_energy(x, y, lengths=None):
        return np.asarray([(float(rank) + 1.0)*float(x[i].pow(2).mean()) for i in range(x.shape[0])])

    torch.manual_seed(rank)
    model = DummyModel(channels=2, output_sources=2)
    FS = 16000
    train = DummyDataset(8, 3, 2, FS//2, FS, transform=model.transform)
    val = DummyDataset(4, 3, 2, FS//2, FS, transform=model.transform)
    trainer = BreverTrainer(
        model=model, train_dataset=train, val_dataset=val,
        model_dirpath=os.path.join(out_dir, 'model'), epochs=1, val_period=1,
        val_metrics={'np_energy'}, batch_sampler='bucket', batch_size=4.0,
        dynamic_batch_size=True, device='cpu', preload=True, ddp=True, rank=rank)
    trainer.run()
    if rank == 0:
        logged = trainer.loss_logger.val_metrics[-1]
        assert 'np_energy' in logged and torch.isfinite(torch.as_tensor(logged['np_energy'])), logged
    dist.destroy_process_group()


def test_trainer_two_ranks_numpy_metric():
    with tempfile.TemporaryDirectory() as tmp:
        _spawn(_worker_trainer_numpy_metric, 2, tmp)


def test_trainer_two_ranks():
    world = 2
    with tempfile.TemporaryDirectory() as tmp:
        _spawn(_worker_trainer, world, tmp)
        p0 = torch.load(os.path.join(tmp, 'params0.pt'))
        p1 = torch.load(os.path.join(tmp, 'params1.pt'))
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)          # broadcast + synchronised gradients


def test_distributed_sampler_partitions_batches():
    from brever_amd.batching import BucketBatchSampler, DistributedBatchSamplerWrapper
    dset = DummyDataset(40, 1, 1, 1000, 8000)
    seen = []
    for rank in range(2):
        sampler = BucketBatchSampler(dset, 1.0, dynamic=True)
        wrapper = DistributedBatchSamplerWrapper(sampler, num_replicas=2, rank=rank)
        wrapper.set_epoch(1)
        got = list(wrapper)
        assert len(got) == len(wrapper)
        seen.append(got)
    union = sorted(i for part in seen for b in part for i in b)
    assert set(union) == set(range(40))
    assert len(seen[0]) == len(seen[1])   # equal step counts (padding by repetition)


def _worker_bench_loop(rank, world, port, out_dir):
    """bench.py's N > 1 data path on two gloo ranks: the batches each rank's loader yields."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from brever_amd.data import BreverDataLoader
    _init(rank, world, port)
    dset, sampler = bench.trainer_sampler(rank, world, 5)
    batches = [list(b) for b in sampler]
    dset.preload_indices(sorted({i for b in batches for i in b}))
    getattr(sampler, 'sampler', sampler)._previous_epoch = None      # as bench.through_trainer: same epoch again
    loader = BreverDataLoader(dataset=dset, batch_sampler=sampler, num_workers=0)
    shapes = [(tuple(batch.shape), lengths.tolist()) for batch, lengths in loader]
    n = torch.tensor([len(batches)])
    lo, hi = n.clone(), n.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    torch.save({'batches': batches, 'shapes': shapes, 'lo': int(lo), 'hi': int(hi)},
               os.path.join(out_dir, f'bench_{rank}.pt'))
    dist.destroy_process_group()


def test_bench_multi_gpu_loop_takes_disjoint_batches_of_one_dataset():
    """for world > 1 `bench.py`'s trainer path feeds every rank from ONE dataset through
    BucketBatchSampler -> DistributedBatchSamplerWrapper (BASELINE config 3's path, brever/training.py:119-125,
    batching.py:279-290): the ranks' batches are disjoint, cover the batch list, and every rank runs the same
    number of steps of 16 x 4 s."""
    world = 2
    with tempfile.TemporaryDirectory() as tmp:
        _spawn(_worker_bench_loop, world, tmp)
        res = [torch.load(os.path.join(tmp, f'bench_{r}.pt')) for r in range(world)]
    assert res[0]['lo'] == res[0]['hi'] == len(res[0]['batches']) == len(res[1]['batches']) == 5
    items = [sorted(i for b in r['batches'] for i in b) for r in res]
    assert not set(items[0]) & set(items[1])
    assert sorted(items[0] + items[1]) == list(range(world*16*5))
    for r in res:
        for shape, lengths in r['shapes']:
            assert shape == (16, 2, 64000) and lengths == [64000]*16
