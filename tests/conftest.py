import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (ROCm device)')


def pytest_sessionstart(session):
    """A fresh checkout has no libbrever_hip.so (built artefacts are git-ignored): build it once
    when hipcc is available, so that the ABI tests (symbols, signatures) and the GPU tests find
    it. The product path itself never builds anything: it fails loudly when the library is
    missing."""
    import shutil
    import subprocess
    lib = os.path.join(ROOT, 'brever_amd', 'csrc', 'libbrever_hip.so')
    hipcc = shutil.which('hipcc') or ('/opt/rocm/bin/hipcc' if os.path.exists('/opt/rocm/bin/hipcc') else None)
    if not os.path.exists(lib) and hipcc:
        subprocess.run(['make', '-C', os.path.join(ROOT, 'brever_amd', 'csrc')], check=False,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
