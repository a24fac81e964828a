"""brever_amd -- MI355X-native hot path behind brever's plugin surface.

Host code in Python on PyTorch-ROCm; compute in hand-written HIP kernels for
gfx950 reached through the C ABI declared in ``include/brever_hip.h``.
"""
__version__ = '0.1.0'
