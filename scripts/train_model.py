"""Training entry point, same command line as the reference (scripts/train_model.py:21-181):

    python scripts/train_model.py models/<id>/ [-f] [--epochs 10 --workers 0 ...]

Every dataset / trainer option of the ``config.yaml`` can be superseded on the command line.
With ``ddp: true`` launch one process per GPU (``torchrun``): gradients are all-reduced over
RCCL every step, overlapped with the backward pass (brever_amd/parallel.py). ``train_path`` /
``val_path`` may be ``synthetic:<items>:<seconds>[:<min_seconds>]`` instead of a dataset
directory. ``trainer.batch_size: 0`` with ``dynamic_batch_size`` sizes the batches from the
HBM capacity (``batching.hbm_batch_seconds``)."""
import argparse
import logging
import os

# read when the HIP runtime loads (torch import): see brever_amd/__init__.py
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import pprint
import random

import numpy as np
import torch
import torch.distributed as dist

from _common import ROOT, is_synthetic, make_dataset  # noqa: F401

from brever_amd.args import ModelArgParser
from brever_amd.config import get_config
from brever_amd.logger import set_logger
from brever_amd.models import ModelRegistry
from brever_amd.training import BreverTrainer


def check_datasets(train_path, val_path):
    """Warn when the training and validation sets were synthesised from the same seed and
    files (reference scripts/train_model.py:140-162; needs the datasets' own config.yaml)."""
    paths = [os.path.join(p, 'config.yaml') for p in (train_path, val_path)]
    if not all(os.path.exists(p) for p in paths):
        logging.warning(f'Could not find {paths[0]} or {paths[1]}. Skipping dataset check.')
        return
    train_cfg, val_cfg = (get_config(p) for p in paths)
    fields = ('seed', 'speakers', 'noises', 'rooms', 'speech_files', 'noise_files',
              'room_files')
    try:
        same = all(getattr(train_cfg.rmm, f) == getattr(val_cfg.rmm, f) for f in fields)
    except AttributeError:
        logging.warning('Dataset configs have no random-mixture-maker section. '
                        'Skipping dataset check.')
        return
    if same:
        logging.warning(
            'Training and validation datasets have the same seed and the same '
            'same speech, noise and room files. They might be the same or too '
            'similar for the validation to be meaningful.')


def main(args):
    loss_path = os.path.join(args.input, 'losses.npz')
    if os.path.exists(loss_path) and not args.force:
        raise FileExistsError(f'training already done: {loss_path}')

    cfg = get_config(os.path.join(args.input, 'config.yaml'))
    cfg.update_from_args(args, ModelArgParser.trainer_arg_map())

    trainer_kwargs = cfg.trainer.to_dict()
    rank = trainer_kwargs.pop('rank')
    device = trainer_kwargs.pop('device')
    if cfg.trainer.ddp:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'   # nccl == RCCL on ROCm
        from brever_amd.parallel import init_process_group
        init_process_group(backend)
        rank = dist.get_rank()
        if torch.cuda.is_available():
            device = rank % torch.cuda.device_count()
            torch.cuda.set_device(device)

    set_logger(os.path.join(args.input, 'log_train.log'), cfg.trainer.ddp, rank)
    if rank == 0:
        logging.info(f'Training {args.input}')
        logging.info(f'Configuration: \n {pprint.pformat(cfg.to_dict())}')

    random.seed(cfg.seed)
    np.random.seed(cfg.seed)
    torch.manual_seed(cfg.seed)

    model = ModelRegistry.get(cfg.arch)(**cfg.model.to_dict())

    if rank == 0 and not is_synthetic(cfg.train_path):
        check_datasets(cfg.train_path, cfg.val_path)
    ds = cfg.dataset
    max_segment_length = ds.max_segment_length
    if cfg.trainer.dynamic_batch_size and max_segment_length == 0:
        # a batch must be able to hold at least one segment (train_model.py:96-100)
        max_segment_length = float(cfg.trainer.batch_size)
    train_dataset = make_dataset(
        cfg.train_path, ds.fs, transform=model.transform, seed=0,
        segment_length=ds.segment_length, overlap_length=ds.overlap_length,
        sources=ds.sources, segment_strategy=ds.segment_strategy,
        max_segment_length=max_segment_length, tar=ds.tar,
        dynamic_mixing=ds.dynamic_mixing,
        dynamic_mixtures_per_epoch=ds.dynamic_mixtures_per_epoch)
    val_dataset = make_dataset(
        cfg.val_path, ds.fs, transform=None, seed=10_000,
        segment_length=0.0, overlap_length=0.0, sources=ds.sources,
        segment_strategy='pass', max_segment_length=max_segment_length, tar=ds.tar,
        dynamic_mixing=False)

    ignore_checkpoint = trainer_kwargs.pop('ignore_checkpoint')
    trainer = BreverTrainer(
        model=model, train_dataset=train_dataset, val_dataset=val_dataset,
        model_dirpath=args.input, device=device, rank=rank,
        ignore_checkpoint=ignore_checkpoint or args.force, **trainer_kwargs)
    trainer.run()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    parser = argparse.ArgumentParser(description='train a model',
                                     conflict_handler='resolve')
    parser.add_argument('input', help='model directory')
    parser.add_argument('-f', '--force', action='store_true',
                        help='train even if already trained')
    parser.add_argument('--wandb_run_id', help='accepted for compatibility (wandb is absent)')
    group = parser.add_argument_group('the following options supersede the config file')
    ModelArgParser.add_dataset_args(group, new_group=False)
    ModelArgParser.add_trainer_args(group, new_group=False)
    ModelArgParser.add_extra_args(group, new_group=False)
    main(parser.parse_args())
