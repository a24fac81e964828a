"""Trainer (host side) for the model plugin surface.

Keeps the constructor signature, checkpoint format and epoch loop of the
reference ``BreverTrainer`` (brever/training.py:25-461) so that
``scripts/train_model.py`` and existing ``config.yaml`` files drive it
unchanged, with two deliberate differences:

* **data parallelism really synchronises gradients.** The reference wraps the
  model in ``DistributedDataParallel`` but then calls ``train_step`` on the
  unwrapped module, so its reducer never runs.
  Here there is no DDP wrapper: parameters are broadcast once from rank 0 and
  ``GradSynchronizer`` all-reduces the (flat) gradient between backward and the
  optimizer step -- RCCL over xGMI on MI355X (backend ``nccl``), gloo in the
  CPU tests. Parity is defined against a single process on the union batch.
* EMA is implemented locally (``torch_ema`` is an absent third-party wheel;
  its arithmetic is "parity unpinned").
"""
import itertools
import logging
import operator
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from .batching import BatchSamplerRegistry, DistributedBatchSamplerWrapper
from .data import BreverDataLoader, DevicePrefetcher
from .inspect import NoParse, Parse
from .metrics import MetricRegistry
from .models import count_params
from .models.base import BreverBaseModel
from .parallel import GradSynchronizer, broadcast_parameters


class ExponentialMovingAverage:
    """Shadow parameters ``s <- d*s + (1-d)*p`` with the warm-up decay
    ``min(decay, (1+n)/(10+n))``; ``store``/``copy_to``/``restore`` swap them in
    for validation (usage: brever/training.py:141-146,312-314,331,357-358)."""

    def __init__(self, parameters, decay, use_num_updates=True):
        self.params = [p for p in parameters if p.requires_grad]
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.shadow = [p.detach().clone() for p in self.params]
        self.backup = None

    @torch.no_grad()
    def update(self):
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates)/(10 + self.num_updates))
        for s, p in zip(self.shadow, self.params):
            if s.device != p.device:
                s.data = s.data.to(p.device)
            s.sub_((1.0 - decay)*(s - p.detach()))

    def store(self):
        self.backup = [p.detach().clone() for p in self.params]

    @torch.no_grad()
    def copy_to(self):
        for s, p in zip(self.shadow, self.params):
            p.copy_(s.to(p.device))

    @torch.no_grad()
    def restore(self):
        for b, p in zip(self.backup, self.params):
            p.copy_(b)
        self.backup = None

    def state_dict(self):
        """Same keys as torch_ema 0.3's ``state_dict`` (``collected_params`` = the stored
        copy, ``None`` outside a store / restore pair), so that the ``ema`` entry of a
        checkpoint loads on either side."""
        return dict(decay=self.decay, num_updates=self.num_updates,
                    shadow_params=self.shadow, collected_params=self.backup)

    def load_state_dict(self, state):
        self.decay = state['decay']
        self.num_updates = state['num_updates']
        self.shadow = [s.clone() for s in state['shadow_params']]
        collected = state.get('collected_params')
        self.backup = None if collected is None else [c.clone() for c in collected]


class BreverTrainer:
    def __init__(
        self,
        model: NoParse[BreverBaseModel],
        train_dataset: NoParse[object],
        val_dataset: NoParse[object],
        model_dirpath: NoParse[str],
        workers: int = 0,
        epochs: int = 100,
        device: int | Parse[str] = 'cuda',
        batch_sampler: str = 'bucket',
        batch_size: int = 32,
        num_buckets: int = 10,
        dynamic_batch_size: bool = True,
        fs: int = 16000,
        ema: bool = False,
        ema_decay: float = 0.999,
        ignore_checkpoint: bool = False,
        preload: bool = False,
        ddp: bool = False,
        rank: int = 0,
        use_wandb: bool = False,
        profile: bool = False,
        val_metrics: set[str] = {'pesq', 'estoi', 'snr'},
        val_period: int = 10,
        use_amp: bool = False,
        compile: bool = False,
        save_on_epochs: list[int] = [],
    ):
        if preload and workers > 0:
            logging.warning('Cannot use workers > 0 with preload=True. '
                            'Forcing workers=0.')
            workers = 0
        if compile:
            logging.warning('compile=True is ignored: the hot path already '
                            'runs in hand-written HIP kernels')
        if use_wandb:
            logging.warning('use_wandb=True is ignored (wandb is not available)')

        # metrics whose third-party backend is missing (pesq: ITU-T P.862 C wheel) would only
        # fail at the first validation, after a full epoch: resolve every name now, drop
        # what cannot run with a warning (unknown names stay a KeyError, as in the reference)
        from .metrics import metric_available
        for name in sorted(val_metrics):
            MetricRegistry.get(name)
        missing = {name for name in val_metrics if not metric_available(name)}
        if missing:
            logging.warning(f'val_metrics {sorted(missing)} need packages that are not '
                            'installed: dropped from validation')
            val_metrics = set(val_metrics) - missing

        if str(device) != 'cpu' and torch.cuda.is_available():
            # On a GPU the host only collates batches and queues kernels. PyTorch's default of
            # one OpenMP thread per core makes those small copies SLOWER and the spinning
            # workers delay the HIP runtime calls: measured on the 128-thread MI355X host,
            # collate + pinned staging of a 16 x 4 s batch 58 ms at 128 threads, 2.3 ms at 1
            # (tools/trainer_path_debug.py). BREVER_HOST_THREADS overrides.
            host_threads = int(os.environ.get('BREVER_HOST_THREADS', '4'))
            if torch.get_num_threads() > host_threads:
                torch.set_num_threads(host_threads)

        self.model = model.to(device)
        self.train_dataset = train_dataset
        self.val_dataset = val_dataset
        self.model_dirpath = model_dirpath
        self.epochs = epochs
        self.device = device
        self.ignore_checkpoint = ignore_checkpoint
        self.preload = preload
        self.rank = rank
        self.profile = profile
        self.val_metrics = val_metrics
        self.val_period = val_period
        self.save_on_epochs = save_on_epochs
        self.use_amp = use_amp

        self.checkpoints_dir = os.path.join(model_dirpath, 'checkpoints')
        self.last_ckpt_path = os.path.join(self.checkpoints_dir, 'last.ckpt')
        self.epochs_ran = 0
        self.max_memory_allocated = 0
        self._profiler = None

        # data parallelism: broadcast once, all-reduce gradients every step
        self.grad_sync = None
        if ddp or dist.is_initialized():
            if not dist.is_initialized():
                raise ValueError('ddp=True requires an initialised process group')
            broadcast_parameters(self.model)
            self.grad_sync = GradSynchronizer(self.model)

        # samplers. NOTE the reference compares the sampler *class* to the string
        # 'bucket' (brever/training.py:96), so num_buckets never reaches the
        # sampler there; kept for batch-composition parity.
        sampler_cls = BatchSamplerRegistry.get(batch_sampler)
        if dynamic_batch_size and batch_size == 0:
            # `batch_size: 0` = size the dynamic batches from the HBM of this GPU: the seconds
            # of audio whose saved activations fit (288 GB on MI355X: a throughput knob, not
            # a memory limit -- batching.hbm_batch_seconds)
            batch_size = self.auto_batch_seconds(self.model, device, fs)
            logging.info(f'dynamic batch size set from HBM capacity: {batch_size:.1f} s')
        self.batch_size = batch_size
        self.train_batch_sampler = sampler_cls(
            dataset=train_dataset, batch_size=batch_size,
            dynamic=dynamic_batch_size, fs=fs)
        if dynamic_batch_size:
            val_batch_size = batch_size
        else:
            val_batch_size = \
                batch_size*train_dataset.get_max_segment_length()/fs
        self.val_batch_sampler = BatchSamplerRegistry.get('sorted')(
            dataset=val_dataset, batch_size=val_batch_size, dynamic=True, fs=fs)
        if dist.is_initialized():
            self.train_batch_sampler = DistributedBatchSamplerWrapper(
                self.train_batch_sampler)
            self.val_batch_sampler = DistributedBatchSamplerWrapper(
                self.val_batch_sampler)

        self.train_dataloader = BreverDataLoader(
            dataset=train_dataset, batch_sampler=self.train_batch_sampler,
            num_workers=workers)
        self.val_dataloader = BreverDataLoader(
            dataset=val_dataset, batch_sampler=self.val_batch_sampler,
            num_workers=workers)

        self.ema = ExponentialMovingAverage(self.model.parameters(), ema_decay) \
            if ema else None
        self.loss_logger = LossLogger(model_dirpath)
        self.checkpoint_saver = CheckpointSaver(self.checkpoints_dir,
                                                self.save_checkpoint)
        self.timer = TrainingTimer(epochs, val_period)
        # kept for API / checkpoint compatibility (brever/training.py:148): the HIP paths
        # compute in bf16 or fp32 with fp32 accumulation and never need loss scaling
        self.scaler = torch.amp.GradScaler('cuda', enabled=False)

    @staticmethod
    def auto_batch_seconds(model, device, fs, cap=4096.0):
        """Dynamic batch size (seconds of audio) whose saved activations fit this GPU's HBM,
        for models that report ``activation_bytes_per_second`` (the HIP Conv-TasNet)."""
        from .batching import hbm_batch_seconds
        per_second = getattr(model, 'activation_bytes_per_second', None)
        if not callable(per_second):
            raise ValueError('batch_size=0 (automatic) needs a model with '
                             'activation_bytes_per_second()')
        hbm = 288e9
        if torch.cuda.is_available() and str(device) != 'cpu':
            hbm = float(torch.cuda.get_device_properties(torch.device(
                'cuda' if isinstance(device, str) else f'cuda:{device}')).total_memory)
        return min(cap, hbm_batch_seconds(per_second(fs), hbm_bytes=hbm))

    # -- helpers ---------------------------------------------------------------
    def get_model(self):
        return self.model

    def optimizers(self):
        opts = self.model.optimizers()
        if isinstance(opts, torch.optim.Optimizer):
            return [opts]
        if not isinstance(opts, (tuple, list)):
            raise ValueError('the model `optimizers` method must return a '
                             f'{torch.optim.Optimizer.__name__} or a sequence, '
                             f'got {opts.__class__.__name__}')
        return opts

    def _log0(self, msg):
        if self.rank == 0:
            logging.info(msg)

    # -- run -------------------------------------------------------------------
    def run(self):
        self._log0(f'Number of parameters: '
                   f'{round(count_params(self.model))/1e6:.2f} M')
        for dset, name in [(self.train_dataset, 'Training dataset'),
                           (self.val_dataset, 'Validation dataset')]:
            for value, kind in [(dset._duration, 'duration'),
                                (dset._effective_duration,
                                 'effective duration')]:
                if value == float('inf'):
                    text = 'inf'
                else:
                    h, rem = divmod(int(value), 3600)
                    mnt, sec = divmod(rem, 60)
                    text = f'{h} h {mnt} m {sec} s'
                self._log0(f'{name} {kind}: {text}')

        resumed = False
        if not self.ignore_checkpoint and os.path.exists(self.last_ckpt_path):
            self._log0('Checkpoint found')
            self.load_checkpoint()
            if self.epochs_ran >= self.epochs:
                self._log0('Model is already trained')
                return
            self._log0(f'Resuming training at epoch {self.epochs_ran}')
            resumed = True

        if self.preload:
            self._log0('Preloading data')
            self.train_dataset.preload(self.device, tqdm_desc='train')
            self.val_dataset.preload(self.device, tqdm_desc='  val')

        if not resumed:
            self._log0('Pre-training model instructions')
            self.model.pre_train(self.train_dataset, self.train_dataloader,
                                 self.epochs)

        if self.profile:
            self._log0('Starting profiler')
            self._profiler = torch.profiler.profile(
                activities=[torch.profiler.ProfilerActivity.CPU,
                            torch.profiler.ProfilerActivity.CUDA],
                schedule=torch.profiler.schedule(wait=1, warmup=1, active=2,
                                                 repeat=1),
                on_trace_ready=lambda prof: self._log0(
                    '\n' + prof.key_averages().table(
                        sort_by='self_cuda_time_total', row_limit=-1)))
            self._profiler.start()
        try:
            self._log0('Starting training loop')
            self.training_loop()
        finally:
            if self._profiler is not None:
                self._profiler.stop()

    def training_loop(self):
        if self.rank == 0:
            self.timer.start()
        for epoch in range(self.epochs_ran, self.epochs):
            self.train_dataloader.set_epoch(epoch)
            self.val_dataloader.set_epoch(epoch)
            train_loss = self.routine(epoch, train=True)
            if self.val_period != 0 and epoch % self.val_period == 0:
                with torch.no_grad():
                    val_loss, val_metrics = self.routine(epoch, train=False)
                self.model.on_validate(
                    val_loss if len(val_loss) > 1
                    else next(iter(val_loss.values())))
            else:
                val_loss, val_metrics = {}, {}
            if dist.is_initialized():
                self.reduce(train_loss, val_loss, val_metrics)
            self.epochs_ran += 1
            if self.rank == 0:
                self.loss_logger.add(train_loss, val_loss, val_metrics)
                self.loss_logger.log(epoch)
                self.checkpoint_saver(epoch, val_loss, val_metrics)
                self.save_checkpoint()
                if epoch in self.save_on_epochs:
                    self.save_checkpoint(os.path.join(
                        self.checkpoints_dir, f'epoch={epoch}.ckpt'))
            if dist.is_initialized():
                dist.barrier()
        if self.rank == 0:
            self.timer.final_log()
            self.loss_logger.plot_and_save()

    def routine(self, epoch, train=True):
        model = self.model
        if train:
            model.train()
            dataloader = self.train_dataloader
        else:
            model.eval()
            dataloader = self.val_dataloader
            if self.ema is not None:
                self.ema.store()
                self.ema.copy_to()
                if hasattr(model, 'mark_params_changed'):
                    model.mark_params_changed()
            avg_metrics = MathDict()
        avg_loss = MathDict()
        use_amp = self.use_amp
        # pinned, double-buffered async H2D: the next batch is copied on a side stream
        # while this one computes (the reference's loop does a synchronous .to per batch,
        # training.py:310-314)
        for batch, lengths in DevicePrefetcher(dataloader, self.device):
            if train:
                loss = self._train_step(model, batch, lengths, use_amp)
                if self.ema is not None:
                    self.ema.update()
            else:
                # the validation set yields raw waveforms: transform + re-collate
                transformed, trans_lengths = BreverDataLoader._collate_fn([
                    model.transform(x[..., :n]) for x, n in zip(batch, lengths)
                ])
                loss = model.val_step(transformed, trans_lengths, use_amp)
                avg_metrics += self.compute_metrics(batch, lengths, use_amp)
            if isinstance(loss, torch.Tensor):
                loss = {'loss': loss}
            elif not isinstance(loss, dict):
                raise ValueError('train_step and val_step must return a tensor '
                                 f'or a dict, got {loss.__class__.__name__}')
            avg_loss += {k: v.detach() for k, v in loss.items()}
            if self._profiler is not None:
                self._profiler.step()
        avg_loss /= len(dataloader)
        if train:
            output = avg_loss
        else:
            if self.ema is not None:
                self.ema.restore()
                if hasattr(model, 'mark_params_changed'):
                    model.mark_params_changed()
            avg_metrics /= len(dataloader)
            output = avg_loss, avg_metrics
        if dist.is_initialized():
            dist.barrier()
        if self.rank == 0:
            self.timer.step(is_validation_step=not train)
            self.timer.log()
        return output

    def _train_step(self, model, batch, lengths, use_amp):
        if self.grad_sync is None:
            return model.train_step(batch, lengths, use_amp, self.scaler)
        return self.grad_sync.train_step(model, batch, lengths, use_amp,
                                         self.scaler)

    def reduce(self, *tensor_dicts):
        """Mean over ranks of every logged scalar, in one packed all-reduce
        (the reference issues one dist.reduce per scalar, training.py:369-373)."""
        items = [(d, k) for d in tensor_dicts for k in d]
        if not items:
            return
        packed = torch.stack([torch.as_tensor(d[k], device=self.device).detach().float().reshape(())
                              for d, k in items])
        dist.all_reduce(packed)
        packed /= dist.get_world_size()
        for i, (d, k) in enumerate(items):
            d[k] = packed[i]

    def compute_metrics(self, batch, lengths, use_amp):
        if not self.val_metrics:
            return {}
        input_, target = batch[:, 0], batch[:, 1:]
        output = self.model.enhance(input_, use_amp=use_amp)
        if output.ndim == 3:
            output = output[:, 0]
        target = target[:, 0].mean(-2)
        metrics = {}
        for name in self.val_metrics:
            values = MetricRegistry.get(name)(output, target, lengths=lengths)
            # (stoi / estoi hand back NumPy arrays: every logged value is a device tensor, as
            # `reduce` and the loss logger expect)
            metrics[name] = torch.as_tensor(values, dtype=torch.float32, device=self.device).mean()
        return metrics

    # -- checkpoints -------------------------------------------------------------
    def save_checkpoint(self, path=None):
        if path is None:
            path = self.last_ckpt_path
        os.makedirs(os.path.dirname(path), exist_ok=True)
        if self.rank != 0:
            return
        peak = torch.cuda.max_memory_allocated() \
            if torch.cuda.is_available() else 0
        state = {
            'epochs': self.epochs_ran,
            'model': self.model.state_dict(),
            'optimizers': [opt.state_dict() for opt in self.optimizers()],
            'scaler': self.scaler.state_dict(),
            'losses': {'train': self.loss_logger.train_loss,
                       'val': self.loss_logger.val_loss},
            'max_memory_allocated': max(peak, self.max_memory_allocated),
            'timer': self.timer.state_dict(),
            'best_ckpts': self.checkpoint_saver.best,
        }
        if self.ema is not None:
            state['ema'] = self.ema.state_dict()
        torch.save(state, path)

    def load_checkpoint(self):
        map_location = f'cuda:{self.device}' if isinstance(self.device, int) \
            else self.device
        state = torch.load(self.last_ckpt_path, map_location=map_location,
                           weights_only=False)
        self.model.load_state_dict(state['model'])
        for opt, sub in zip(self.optimizers(), state['optimizers']):
            opt.load_state_dict(sub)
        self.scaler.load_state_dict(state['scaler'])
        self.loss_logger.train_loss = state['losses']['train']
        self.loss_logger.val_loss = state['losses']['val']
        self.epochs_ran = state['epochs']
        self.max_memory_allocated = state['max_memory_allocated']
        self.timer.load_state_dict(state['timer'])
        self.checkpoint_saver.best = state['best_ckpts']
        if self.ema is not None:
            if 'ema' not in state:
                raise ValueError('exponential moving average state not found '
                                 'in state dict')
            self.ema.load_state_dict(state['ema'])


class TrainingTimer:
    """Average epoch / validation durations and ETA, resumable
    (brever/training.py:464-595)."""

    _FIELDS = ('train_steps_taken', 'val_steps_taken', 'train_steps_measured',
               'val_steps_measured', 'avg_train_duration', 'avg_val_duration')

    def __init__(self, epochs, val_period):
        self.epochs = epochs
        self.val_period = val_period
        self.train_steps_taken = self.val_steps_taken = 0
        self.train_steps_measured = self.val_steps_measured = 0
        self.avg_train_duration = None
        self.avg_val_duration = 0 if val_period == 0 else None
        self.start_time = self.step_start_time = None
        self.resume_offset = 0
        self.first_session_step = True

    def state_dict(self):
        state = {k: getattr(self, k) for k in self._FIELDS}
        state['resume_offset'] = self.total_elapsed_time
        return state

    def load_state_dict(self, state):
        for k in self._FIELDS + ('resume_offset',):
            setattr(self, k, state[k])

    def start(self):
        self.start_time = self.step_start_time = time.time()

    def step(self, is_validation_step=False):
        now = time.time()
        duration = now - self.step_start_time
        kind = 'val' if is_validation_step else 'train'
        if not self.first_session_step:
            n = getattr(self, f'{kind}_steps_measured')
            avg = getattr(self, f'avg_{kind}_duration')
            avg = duration if not n or avg is None else (avg*n + duration)/(n + 1)
            setattr(self, f'avg_{kind}_duration', avg)
            setattr(self, f'{kind}_steps_measured', n + 1)
        setattr(self, f'{kind}_steps_taken',
                getattr(self, f'{kind}_steps_taken') + 1)
        self.first_session_step = False
        self.step_start_time = now

    @staticmethod
    def fmt_time(t):
        if t is None:
            return '--'
        h, mnt, sec = int(t//3600), int((t % 3600)//60), int(t % 60)
        text = f'{sec} s'
        if t >= 60:
            text = f'{mnt} m {text}'
        if t >= 3600:
            text = f'{h} h {text}'
        return text

    def log(self):
        logging.info(', '.join([
            f'Avg train time: {self.fmt_time(self.avg_train_duration)}',
            f'Avg val time: {self.fmt_time(self.avg_val_duration)}',
            f'ETA: {self.fmt_time(self.estimated_time_left)}',
        ]))

    def final_log(self):
        t = self.total_elapsed_time
        logging.info(f'Time spent: {int(t/3600)} h {int(t % 3600/60)} m '
                     f'{int(t % 60)} s')

    @property
    def total_elapsed_time(self):
        return time.time() - self.start_time + self.resume_offset

    @property
    def estimated_time_left(self):
        if self.avg_train_duration is None or self.avg_val_duration is None:
            return None
        val_steps = 0 if self.val_period == 0 else self.epochs//self.val_period
        return self.avg_train_duration*(self.epochs - self.train_steps_taken) \
            + self.avg_val_duration*(val_steps - self.val_steps_taken)


class LossLogger:
    def __init__(self, dirpath):
        self.train_loss, self.val_loss, self.val_metrics = [], [], []
        self.dirpath = dirpath

    def add(self, train_loss, val_loss, val_metrics):
        self.train_loss.append(train_loss)
        self.val_loss.append(val_loss)
        self.val_metrics.append(val_metrics)

    def log(self, epoch):
        parts = itertools.chain(
            (f'train_{k}: {v:.2e}' for k, v in self.train_loss[-1].items()),
            (f'val_{k}: {v:.2e}' for k, v in self.val_loss[-1].items()),
            (f'val_{k}: {v:.2e}' for k, v in self.val_metrics[-1].items()),
        )
        logging.info(f'Epoch {epoch}: ' + ', '.join(parts))

    def to_numpy(self):
        out = {}
        for series, tag in [(self.train_loss, 'train'), (self.val_loss, 'val'),
                            (self.val_metrics, 'metrics')]:
            for epoch, d in enumerate(series):
                for k, v in d.items():
                    out.setdefault(f'{tag}_{k}', []).append((epoch, float(v)))
        return {k: np.array(v) for k, v in out.items()}

    def plot_and_save(self):
        losses = self.to_numpy()
        np.savez(os.path.join(self.dirpath, 'losses.npz'), **losses)
        try:
            import matplotlib
            matplotlib.use('Agg')
            import matplotlib.pyplot as plt
        except Exception:       # plotting is cosmetic
            return
        fig, ax = plt.subplots()
        for k, v in losses.items():
            if not k.startswith('metrics'):
                ax.plot(v[:, 0], v[:, 1], label=k)
        ax.legend()
        ax.set_xlabel('epoch')
        ax.set_ylabel('error')
        ax.grid(True)
        fig.tight_layout()
        fig.savefig(os.path.join(self.dirpath, 'training_curve.png'))
        plt.close(fig)


class CheckpointSaver:
    """Keeps one best checkpoint per logged loss (lower is better) and metric
    (higher is better), named ``epoch=N_<name>=<value>.ckpt``
    (brever/training.py:668-699)."""

    def __init__(self, dirpath, save_func):
        self.dirpath = dirpath
        self.save_func = save_func
        self.best = {}

    def __call__(self, epoch, loss, metrics):
        for values, better in [(loss, operator.lt), (metrics, operator.gt)]:
            for name, val in values.items():
                known = name in self.best
                if known and not better(val, self.best[name]['val']):
                    continue
                path = os.path.join(
                    self.dirpath, f'epoch={epoch}_{name}={self._fmt(val)}.ckpt')
                self.save_func(path)
                logging.info(f'New best {name}, saving {path}')
                if known:
                    old = self.best[name]['filepath']
                    if os.path.exists(old):
                        os.remove(old)
                    else:
                        logging.warning(f'Previous best {name} checkpoint {old} '
                                        'does not exist. Skipping removal.')
                self.best[name] = {'val': val, 'filepath': path}

    @staticmethod
    def _fmt(x):
        return f'{x:.2e}' if abs(x) < 0.1 or abs(x) >= 100 else f'{x:.2f}'


class MathDict(dict):
    """dict with element-wise arithmetic against dicts and scalars."""

    @staticmethod
    def _apply(target, other, op, default):
        if isinstance(other, dict):
            for key, value in other.items():
                target[key] = op(target.get(key, default), value)
        elif isinstance(other, (int, float)):
            for key in target:
                target[key] = op(target[key], other)
        return target

    def __add__(self, o): return self._apply(MathDict(self), o, operator.add, 0)
    def __sub__(self, o): return self._apply(MathDict(self), o, operator.sub, 0)
    def __mul__(self, o): return self._apply(MathDict(self), o, operator.mul, 1)
    def __truediv__(self, o):
        return self._apply(MathDict(self), o, operator.truediv, 1)
    def __iadd__(self, o): return self._apply(self, o, operator.add, 0)
    def __isub__(self, o): return self._apply(self, o, operator.sub, 0)
    def __imul__(self, o): return self._apply(self, o, operator.mul, 1)
    def __itruediv__(self, o): return self._apply(self, o, operator.truediv, 1)
