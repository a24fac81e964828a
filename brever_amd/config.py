"""Configuration objects of the entry points.

Compact equivalent of the reference's config system for the hot path
(brever/config.py:13-136): ``get_config(path)`` loads a ``config.yaml`` into an
immutable nested ``BreverConfig`` with attribute access, ``to_dict()``, an
order-independent ``get_hash()`` (first 8 hex of a sha256, config.py:78-95) and
``update_from_args`` for command-line overrides (config.py:104-123). Default model
hyper-parameters come from the model's ``__init__`` signature (the reference
requires its YAML defaults to equal them, config.py:24-28). Experiment
book-keeping (finders / initialisers) is out of scope.
"""
import hashlib
import inspect
import json

import yaml


class BreverConfig:
    def __init__(self, dict_):
        for key, value in dict_.items():
            if isinstance(value, dict):
                value = BreverConfig(value)
            super().__setattr__(key, value)

    def __setattr__(self, attr, value):
        raise AttributeError(f'{self.__class__.__name__} objects are immutable')

    def __repr__(self):
        return f'BreverConfig({self.to_dict()})'

    def keys(self):
        return self.__dict__.keys()

    def to_dict(self):
        return {k: v.to_dict() if isinstance(v, BreverConfig) else v
                for k, v in self.__dict__.items()}

    def to_json(self):
        def norm(x):
            if isinstance(x, dict):
                return {k: norm(v) for k, v in sorted(x.items())}
            if isinstance(x, (set, frozenset)):
                return sorted(norm(v) for v in x)
            if isinstance(x, (list, tuple)):
                return [norm(v) for v in x]
            return x
        return norm(self.to_dict())

    def get_hash(self, length=8):
        text = json.dumps(self.to_json(), sort_keys=True)
        return hashlib.sha256(text.encode()).hexdigest()[:length]

    def update_from_dict(self, dict_, parent_keys=()):
        for key, value in dict_.items():
            current = getattr(self, key)
            if isinstance(current, BreverConfig):
                current.update_from_dict(value, parent_keys + (key,))
                continue
            if current is not None and value is not None \
                    and type(current) is not type(value) \
                    and not (isinstance(current, float) and isinstance(value, int)):
                raise TypeError(
                    f'type mismatch for {".".join(parent_keys + (key,))}: '
                    f'{type(current).__name__} vs {type(value).__name__}')
            object.__setattr__(self, key, value)

    def update_from_args(self, args, arg_map):
        """``arg_map``: argparse dest -> sequence of nested keys; only options the
        user actually passed (not None) override the file."""
        for dest, keys in arg_map.items():
            value = getattr(args, dest, None)
            if value is None:
                continue
            node = self
            for key in keys[:-1]:
                node = getattr(node, key)
            node.update_from_dict({keys[-1]: value}, tuple(keys[:-1]))


def get_config(path):
    with open(path) as f:
        return BreverConfig(yaml.load(f, Loader=yaml.Loader))


def signature_defaults(func, skip=('self', 'model', 'train_dataset', 'val_dataset',
                                   'model_dirpath')):
    out = {}
    for name, prm in inspect.signature(func).parameters.items():
        if name in skip or prm.default is inspect.Parameter.empty:
            continue
        out[name] = prm.default
    return out


def model_defaults(model_cls):
    """Constructor defaults of a model class; sub-models (``_is_submodel``) inherit their
    parent's and override some (brever/inspect.py:123-126)."""
    out = {}
    if model_cls.__dict__.get('_is_submodel', False):
        out = model_defaults(model_cls.__bases__[0])
    out.update(signature_defaults(model_cls.__init__))
    out.update(model_cls.__dict__.get('_defaults', {}))
    return out


def get_model_default_config(arch):
    """Default ``config.yaml`` content for ``arch`` (model + trainer + dataset)."""
    from .models import ModelRegistry
    from .training import BreverTrainer
    model_cls = ModelRegistry.get(arch)
    trainer = signature_defaults(BreverTrainer.__init__)
    trainer['val_metrics'] = {'snr'}     # pesq / estoi wheels are absent here
    trainer['use_amp'] = True
    return BreverConfig({
        'arch': arch, 'seed': 0, 'train_path': 'none', 'val_path': 'none',
        'dataset': {'fs': 16000, 'sources': ['mixture', 'foreground'],
                    'segment_length': 0.0, 'max_segment_length': 0.0},
        'trainer': trainer,
        'model': model_defaults(model_cls),
    })
