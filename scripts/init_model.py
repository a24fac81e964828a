"""Write models/<hash>/config.yaml from the architecture defaults plus overrides
(reference: scripts/init_model.py:9-29)."""
import argparse
import os

import yaml

from _common import ROOT, add_override_flags  # noqa: F401

from brever_amd.config import get_model_default_config


def main():
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument('arch')
    arch = pre.parse_known_args()[0].arch
    cfg = get_model_default_config(arch)
    parser = argparse.ArgumentParser(description='initialize a model directory')
    parser.add_argument('arch')
    parser.add_argument('--models-dir', default='models')
    parser.add_argument('--train-path', dest='train_path', default=None)
    parser.add_argument('--val-path', dest='val_path', default=None)
    parser.add_argument('--seed', type=int, default=None)
    m_map = add_override_flags(parser, cfg.model.to_dict())
    t_map = add_override_flags(parser, cfg.trainer.to_dict(), prefix='trainer_')
    args = parser.parse_args()
    arg_map = {d: ('model', k) for d, k in m_map.items()}
    arg_map.update({d: ('trainer', k) for d, k in t_map.items()})
    arg_map.update({'train_path': ('train_path',), 'val_path': ('val_path',),
                    'seed': ('seed',)})
    cfg.update_from_args(args, arg_map)
    dirpath = os.path.join(args.models_dir, cfg.get_hash())
    os.makedirs(dirpath, exist_ok=True)
    path = os.path.join(dirpath, 'config.yaml')
    if os.path.exists(path):
        raise FileExistsError(f'model already exists: {path}')
    with open(path, 'w') as f:
        yaml.dump(cfg.to_dict(), f)
    print(f'Initialized {path}')


if __name__ == '__main__':
    main()
