"""Write ``<MODELS>/<id>/config.yaml`` from the architecture defaults plus overrides.

Same command line as the reference (scripts/init_model.py:9-38): dataset / trainer / extra
options first, then the architecture and its own options, e.g.

    python scripts/init_model.py --train_path data/train --val_path data/val \\
        --epochs 50 --use_amp true convtasnet --layers 4

``<id>`` is ``config.get_hash()`` (identical to the reference's ids) unless ``-n NAME``;
``MODELS`` comes from ``config/paths.yaml``. Extension: ``--models_dir DIR`` overrides it."""
import os

# read when the HIP runtime loads (torch import): see brever_amd/__init__.py
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import yaml

from _common import ROOT  # noqa: F401

from brever_amd.args import ModelArgParser
from brever_amd.config import _resolve, get_config, get_model_default_config


def main():
    parser = ModelArgParser(description='initialize a model')
    parser.add_argument('-f', '--force', action='store_true',
                        help='overwrite config file if already exists')
    parser.add_argument('-n', '--name', help='model name')
    parser.add_argument('--models_dir', help='(extension) write here instead of paths.MODELS')
    args = parser.parse_args()

    config = get_model_default_config(args.arch)
    config.update_from_args(args, parser.arg_map(args.arch))
    model_id = args.name if args.name is not None else config.get_hash()
    models_dir = args.models_dir or get_config(_resolve('config/paths.yaml')).MODELS
    model_dir = os.path.join(models_dir, model_id)
    os.makedirs(model_dir, exist_ok=True)
    config_path = os.path.join(model_dir, 'config.yaml')
    if os.path.exists(config_path) and not args.force:
        raise FileExistsError(f'model already exists: {config_path} ')
    with open(config_path, 'w') as f:
        yaml.dump(config.to_dict(), f, sort_keys=False)
    print(f'Initialized {config_path}')


if __name__ == '__main__':
    main()
