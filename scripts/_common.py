"""Shared helpers of the entry-point scripts."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def str2bool(x):
    if x.lower() in ('1', 'true', 'yes', 'y'):
        return True
    if x.lower() in ('0', 'false', 'no', 'n'):
        return False
    raise argparse.ArgumentTypeError(f'expected a boolean, got {x}')


def add_override_flags(parser, defaults, prefix=''):
    """One optional flag per scalar / list default (the reference generates its CLI
    from type hints the same way, brever/args.py:82-143)."""
    arg_map = {}
    for name, value in defaults.items():
        flag = f'--{prefix}{name}'
        dest = f'{prefix}{name}'.replace('-', '_')
        if isinstance(value, bool):
            parser.add_argument(flag, type=str2bool, default=None, dest=dest)
        elif isinstance(value, (int, float, str)):
            parser.add_argument(flag, type=type(value), default=None, dest=dest)
        elif isinstance(value, (set, frozenset)):
            parser.add_argument(flag, default=None, dest=dest,
                                type=lambda s: set(x for x in s.split(',') if x))
        elif isinstance(value, list):
            parser.add_argument(flag, default=None, dest=dest,
                                type=lambda s: [int(x) for x in s.split(',') if x])
        else:
            continue
        arg_map[dest] = name
    return arg_map


def make_dataset(spec, fs, transform=None, seed=0, **dataset_kwargs):
    """``synthetic:<items>:<seconds>[:<min_seconds>]`` -> in-memory synthetic mixtures; any
    other value is the path of a dataset directory in the reference's layout
    (``audio/NNNNN_<source>.flac|wav`` or ``audio.tar``), read by ``BreverDataset`` with the
    ``dataset`` section of the model config (``tar=False`` is tried when there is no
    ``audio.tar``)."""
    import os

    from brever_amd.data import BreverDataset, SyntheticMixtureDataset
    if not str(spec).startswith('synthetic:'):
        dataset_kwargs.setdefault('tar', os.path.exists(os.path.join(spec, 'audio.tar')))
        return BreverDataset(spec, fs=fs, transform=transform, **dataset_kwargs)
    parts = spec.split(':')[1:]
    n, seconds = int(parts[0]), float(parts[1])
    min_len = int(float(parts[2])*fs) if len(parts) > 2 else None
    return SyntheticMixtureDataset(n, int(seconds*fs), fs=fs, min_length=min_len,
                                   transform=transform, seed=seed)
