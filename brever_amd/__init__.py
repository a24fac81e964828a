"""brever_amd -- MI355X-native hot path behind brever's plugin surface.

Host code in Python on PyTorch-ROCm; compute in hand-written HIP kernels for
gfx950 reached through the C ABI declared in ``include/brever_hip.h``.
"""
import os as _os

# multi-process GPU work on this platform needs dmabuf IPC (RCCL, device tensors shared between
# processes): the legacy IPC mode fails with `hipIpcGetMemHandle: invalid argument`. Only a default --
# an exported value wins -- and it must be in place before the HIP runtime initialises.
_os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
# Conv-TasNet's training step runs two kernel chains on two streams (models/convtasnet.py); next to
# RCCL's streams the default 4 hardware queues are oversubscribed and the chains end up sharing one.
# Read when the HIP runtime loads: effective if this package is imported before torch, as the entry
# points (bench.py, scripts/) do; otherwise export it.
import sys as _sys

_q = _os.environ.get('GPU_MAX_HW_QUEUES')
# False: torch (and with it the HIP runtime) was loaded before this default could be seen; the
# two-chain step then stays off next to an initialised process group (models/convtasnet.py)
HW_QUEUES_OK = (_q is not None and _q.isdigit() and int(_q) >= 8) or 'torch' not in _sys.modules
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

__version__ = '0.1.0'
