"""Entry points and config objects (CPU parts only)."""
import argparse
import os
import subprocess
import sys

import pytest
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_hash_and_immutability():
    from brever_amd.config import BreverConfig
    a = BreverConfig({'x': {'a': 1, 'b': {3, 1}}, 'y': [1, 2]})
    b = BreverConfig({'y': [1, 2], 'x': {'b': {1, 3}, 'a': 1}})
    assert a.get_hash() == b.get_hash() and len(a.get_hash()) == 8
    assert a.get_hash() != BreverConfig({'x': {'a': 2, 'b': {3, 1}}, 'y': [1, 2]}).get_hash()
    with pytest.raises(AttributeError):
        a.y = 3
    ns = argparse.Namespace(a=5, y=None)
    a.update_from_args(ns, {'a': ('x', 'a'), 'y': ('y',)})
    assert a.x.a == 5 and a.y == [1, 2]
    with pytest.raises(TypeError):
        a.update_from_dict({'x': {'a': 'str'}})


def test_default_config_matches_signature():
    from brever_amd.config import get_model_default_config
    cfg = get_model_default_config('convtasnet')
    assert cfg.arch == 'convtasnet'
    assert cfg.model.to_dict() == dict(
        filters=512, filter_length=32, bottleneck_channels=128, hidden_channels=512,
        skip_channels=128, kernel_size=3, layers=8, repeats=3, output_sources=1,
        causal=False, criterion='snr', optimizer='Adam', learning_rate=0.001,
        grad_clip=5.0)                      # reference config/models/convtasnet.yaml:42-56
    assert cfg.trainer.batch_sampler == 'bucket' and cfg.trainer.dynamic_batch_size


def test_init_model_script_writes_config(tmp_path):
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, 'scripts', 'init_model.py'), 'convtasnet',
         '--models-dir', str(tmp_path), '--layers', '2', '--trainer_epochs', '3',
         '--train-path', 'synthetic:8:1.0', '--val-path', 'synthetic:2:1.0'],
        capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    dirs = os.listdir(tmp_path)
    assert len(dirs) == 1
    cfg = yaml.load(open(tmp_path / dirs[0] / 'config.yaml'), Loader=yaml.Loader)
    assert cfg['model']['layers'] == 2 and cfg['trainer']['epochs'] == 3
    assert cfg['train_path'] == 'synthetic:8:1.0'
