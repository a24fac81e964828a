"""Shared helpers of the entry-point scripts."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def is_synthetic(spec):
    return str(spec).startswith('synthetic:')


def make_dataset(spec, fs, transform=None, seed=0, **dataset_kwargs):
    """``synthetic:<items>:<seconds>[:<min_seconds>]`` -> in-memory synthetic mixtures
    (SURVEY.md 8d recipe; there are no corpora in this image); any other value is the path of
    a dataset directory in the reference's layout (``audio/NNNNN_<source>.flac|wav`` or
    ``audio.tar``), read by ``BreverDataset`` with the given ``dataset`` section options
    (``tar`` falls back to ``False`` when the directory has no ``audio.tar``)."""
    from brever_amd.data import BreverDataset, SyntheticMixtureDataset
    if not is_synthetic(spec):
        if dataset_kwargs.get('tar', True) and not dataset_kwargs.get('dynamic_mixing', False) \
                and not os.path.exists(os.path.join(spec, 'audio.tar')):
            dataset_kwargs['tar'] = False
        return BreverDataset(spec, fs=fs, transform=transform, **dataset_kwargs)
    parts = spec.split(':')[1:]
    n, seconds = int(parts[0]), float(parts[1])
    min_len = int(float(parts[2])*fs) if len(parts) > 2 else None
    return SyntheticMixtureDataset(n, int(seconds*fs), fs=fs, min_length=min_len,
                                   transform=transform, seed=seed)
