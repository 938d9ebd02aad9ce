"""singa_amd — MI355X-native implementation of the SINGA equivariant message-passing + CProMG training hot path.

Host side: Python on PyTorch-ROCm mirroring the reference's module surface (singa_amd.model.*).
Device side: hand-written gfx950 kernels in libsinga_hip.so (singa_amd/csrc, C ABI in include/singa_hip.h).
There is no CPU fallback: importing the ops without the built library, or calling them on CPU tensors, raises.
"""
__version__ = "0.1.0"
