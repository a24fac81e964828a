"""Configuration objects of the entry points (interface of brever/config.py:13-136,263-319).

``get_config(path)`` loads a ``config.yaml`` into an immutable nested ``BreverConfig`` with
attribute access, ``to_dict()`` / ``to_json()``, ``get_field`` / ``set_field`` on key paths,
``update_from_args`` (argparse namespace + the ``arg_map`` of ``brever_amd.args``) and
``get_hash()``. The hash is the model-directory id, so it is reproduced byte for byte: sha256 of
``str(d.items())`` of the recursively key-sorted dict with sets as sorted lists
(config.py:78-95) -- pinned by the reference's hashes of its own default configs and of 20
perturbed ones (tests/golden/config.json). ``get_model_default_config`` reads
``config/models/<arch>.yaml`` and warns when its ``model`` section differs from the model's
signature defaults (config.py:20-31). Dataset / model *finders* (experiment book-keeping) are
out of scope.
"""
import hashlib
import os
import warnings

import yaml

from .inspect import Path, get_func_spec

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _resolve(relpath):
    """The reference opens ``config/...`` relative to the working directory; fall back to the
    files shipped with this package when the caller runs from somewhere else."""
    if os.path.exists(relpath):
        return relpath
    return os.path.join(_ROOT, relpath)


def get_config(path):
    with open(path) as f:
        return BreverConfig(yaml.load(f, Loader=yaml.Loader))


def get_model_default_config(model_key):
    from .models import ModelRegistry
    path = f'config/models/{model_key}.yaml'
    with open(_resolve(path)) as f:
        content = yaml.load(f, Loader=yaml.Loader)
    signature = {name: item['default']
                 for name, item in get_func_spec(ModelRegistry.get(model_key)).items()}
    if content['model'] != signature:
        warnings.warn(f'Default config file {path} does not match default '
                      'arguments from model __init__ signature')
    return BreverConfig(content)


def _canonical(node):
    """Key-sorted copy with sets as sorted lists: what the hash is taken of."""
    out = {}
    for key in sorted(node):
        value = node[key]
        if isinstance(value, dict):
            out[key] = _canonical(value)
        elif isinstance(value, set):
            out[key] = sorted(value)
        else:
            out[key] = value
    return out


class BreverConfig:
    def __init__(self, dict_):
        for key, value in dict_.items():
            object.__setattr__(self, key,
                               BreverConfig(value) if isinstance(value, dict) else value)

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
        out = {}
        for k, v in self.__dict__.items():
            if isinstance(v, BreverConfig):
                out[k] = v.to_json()
            else:
                out[k] = sorted(v) if isinstance(v, set) else v
        return out

    def get_hash(self, length=8):
        text = str(_canonical(self.to_dict()).items())
        return hashlib.sha256(text.encode()).hexdigest()[:length]

    def get_field(self, key_list):
        node = self
        for key in key_list:
            node = getattr(node, key)
        return node

    def set_field(self, key_list, value):
        owner = self.get_field(key_list[:-1])
        key = key_list[-1]
        current = getattr(owner, key)
        if not isinstance(value, type(current)):
            raise TypeError(f'attribute {key} must be {type(current).__name__}, '
                            f'got {type(value).__name__}')
        object.__setattr__(owner, key, value)

    def update_from_args(self, args, arg_map):
        """Options the user did not pass are ``None`` in ``args`` and leave the file's value."""
        for name, key_lists in arg_map.items():
            value = getattr(args, name)
            if value is None:
                continue
            for key_list in key_lists:
                self.set_field(key_list, value)

    def update_from_dict(self, dict_, parent_keys=()):
        for key, value in dict_.items():
            path = list(parent_keys) + [key]
            if isinstance(value, dict):
                self.update_from_dict(value, path)
            else:
                self.set_field(path, value)


class ModelInitializer:
    """``models/<hash>/config.yaml`` writer (config.py:263-308)."""

    def __init__(self, batch_mode=False, models_dir=None):
        self.dir_ = models_dir or get_config(_resolve('config/paths.yaml')).MODELS
        self.batch_mode = batch_mode

    def init_from_args(self, args):
        from .args import ModelArgParser
        config = get_model_default_config(args.arch)
        config.update_from_args(args, ModelArgParser.arg_map(args.arch))
        return self.write_config(config, args.force)

    def get_config_from_kwargs(self, arch, **kwargs):
        from .args import ModelArgParser
        config = get_model_default_config(arch)
        arg_map = ModelArgParser.arg_map(arch)
        for key, value in kwargs.items():
            for key_list in arg_map[key]:
                config.set_field(key_list, value)
        return config

    def init_from_kwargs(self, arch, force=False, model_id=None, **kwargs):
        return self.write_config(self.get_config_from_kwargs(arch, **kwargs), force=force,
                                 model_id=model_id)

    def get_path_from_kwargs(self, arch, **kwargs):
        return Path(os.path.join(self.dir_, self.get_config_from_kwargs(arch, **kwargs).get_hash()))

    def write_config(self, config, force=False, model_id=None):
        model_dir = os.path.join(self.dir_, model_id or config.get_hash())
        os.makedirs(model_dir, exist_ok=True)
        config_path = os.path.join(model_dir, 'config.yaml')
        if os.path.exists(config_path) and not force:
            if not self.batch_mode:
                raise FileExistsError(f'model already exists: {config_path}')
            print(f'model already exists: {config_path}')
        else:
            with open(config_path, 'w') as f:
                yaml.dump(config.to_dict(), f)
            print(f'Initialized {config_path}')
        return Path(model_dir)
