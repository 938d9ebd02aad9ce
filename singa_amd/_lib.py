"""Loader of libsinga_hip.so.  No fallback: if the library is absent or a symbol is missing this raises, and the
ops refuse CPU tensors (the product path is the HIP path)."""
import ctypes
import os

import numpy as np
import torch  # noqa: F401  - loaded before libsinga_hip.so so that both use the HIP runtime PyTorch ships

from . import _capi, so3

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsinga_hip.so")
_lib = None
_inited_devices = set()


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not built - run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        _lib = _capi.bind(LIB_PATH)
    return _lib


def ensure_init(device_index):
    """Upload the J tables to `device_index` once (singa_init is per device: it fills a __device__ array)."""
    if device_index in _inited_devices:
        return
    import torch
    with torch.cuda.device(device_index):
        jd = np.ascontiguousarray(so3.jd_flat(6), dtype=np.float64)
        code = lib().singa_init(jd.ctypes.data_as(ctypes.c_void_p), 6)
        _capi.check(lib(), code, "singa_init")
    _inited_devices.add(device_index)
