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

_exported = _os.environ.get('GPU_MAX_HW_QUEUES')
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
_eff = _os.environ['GPU_MAX_HW_QUEUES']
# True when the runtime does (or will) see >= 8 hardware queues: the effective value must be >= 8, and it
# only takes effect if it was exported by the user or torch (and with it the HIP runtime) has not been
# loaded yet. Otherwise the two-chain step stays off next to an initialised process group
# (models/convtasnet.py): an exported GPU_MAX_HW_QUEUES=4 used to slip through as "ok".
HW_QUEUES_OK = _eff.isdigit() and int(_eff) >= 8 and (_exported is not None or 'torch' not in _sys.modules)

__version__ = '0.1.0'
