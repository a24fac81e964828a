"""Argument parsers generated from signatures (interface of brever/args.py:82-143).

``ModelArgParser`` gives every argument of ``BreverDataset.__init__``,
``BreverTrainer.__init__`` and of the selected model's ``__init__`` a ``--name`` option and
knows where each lands in a ``config.yaml`` (``arg_map``: option -> list of key paths), so that
``init_model.py --epochs 3 convtasnet --layers 2`` and ``train_model.py models/<id> --workers 0``
work as in the reference. The dataset-creation parser (``DatasetArgParser`` over
``RandomMixtureMaker``) belongs to the dataset synthesis tools, which are out of scope.
"""
import argparse
import warnings

from .inspect import Path, get_func_spec

ALLOWED_DUPLICATE_ARGS = ['fs']      # dataset.fs and trainer.fs are meant to move together


def _sections():
    from .data import BreverDataset
    from .training import BreverTrainer
    return BreverDataset, BreverTrainer


class BaseArgParser(argparse.ArgumentParser):
    extra_args = {}

    @classmethod
    def _add_args(cls, func, parser, add_defaults=False, required=True):
        for name, kw in get_func_spec(func).items():
            kw = dict(kw)
            if not add_defaults:
                kw['default'] = None          # "not given" must stay distinguishable
            if not required:
                kw['required'] = False
            parser.add_argument(f'--{name}', **kw)

    @classmethod
    def add_extra_args(cls, parser, new_group=True, required=False):
        target = parser.add_argument_group('extra options') if new_group else parser
        for name, kw in cls.extra_args.items():
            kw = dict(kw, required=kw.get('required', False) and required)
            target.add_argument(f'--{name}', **kw)

    @classmethod
    def build_argmap(cls, prefixes, classes):
        arg_map = {}
        for prefix, owner in zip(prefixes, classes):
            for name in get_func_spec(owner):
                arg_map.setdefault(name, []).append([prefix, name] if prefix else [name])
        for name, paths in arg_map.items():
            if len(paths) > 1 and name not in ALLOWED_DUPLICATE_ARGS:
                warnings.warn(
                    f'Argument --{name} matches more than one configuration field: '
                    f'{", ".join(".".join(p) for p in paths)}. '
                    'These will be set to the same value.')
        return arg_map


class ModelArgParser(BaseArgParser):
    extra_args = {
        'seed': dict(type=int),
        'train_path': dict(type=Path, required=True),
        'val_path': dict(type=Path, required=True),
    }

    def __init__(self, required=True, *args, **kwargs):
        from .models import ModelRegistry
        super().__init__(*args, conflict_handler='resolve', **kwargs)
        self.add_dataset_args(self, required=required)
        self.add_trainer_args(self, required=required)
        self.add_extra_args(self, required=required)
        subs = self.add_subparsers(help='model architecture', dest='arch',
                                   parser_class=argparse.ArgumentParser,
                                   required=required)
        for key in ModelRegistry.keys():
            self.add_model_args(subs.add_parser(key, conflict_handler='resolve'), key)

    @classmethod
    def add_model_args(cls, parser, model, new_group=True, required=False):
        from .models import ModelRegistry
        target = parser.add_argument_group('model options') if new_group else parser
        cls._add_args(ModelRegistry.get(model), target, required=required)

    @classmethod
    def add_dataset_args(cls, parser, new_group=True, required=False):
        target = parser.add_argument_group('dataset options') if new_group else parser
        cls._add_args(_sections()[0], target, required=required)

    @classmethod
    def add_trainer_args(cls, parser, new_group=True, required=False):
        target = parser.add_argument_group('trainer options') if new_group else parser
        cls._add_args(_sections()[1], target, required=required)

    @classmethod
    def trainer_arg_map(cls):
        dataset, trainer = _sections()
        return {**{name: [[name]] for name in cls.extra_args},
                **cls.build_argmap(['dataset', 'trainer'], [dataset, trainer])}

    @classmethod
    def arg_map(cls, model_key):
        from .models import ModelRegistry
        dataset, trainer = _sections()
        return {**{name: [[name]] for name in cls.extra_args},
                **cls.build_argmap(['dataset', 'trainer', 'model'],
                                   [dataset, trainer, ModelRegistry.get(model_key)])}
