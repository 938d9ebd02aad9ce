"""The gfx950 library builds without a GPU (hipcc cross-compiles), loads, and exports every entry point that
include/singa_hip.h declares; the ctypes binding table covers the same set.  No kernel is launched here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="singa_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(singa_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built_lib():
    import __graft_entry__
    __graft_entry__.build()
    return __graft_entry__.LIB


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib)
    names = declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.singa_version.restype = ctypes.c_int
    assert lib.singa_version() >= 100
    lib.singa_last_error_string.restype = ctypes.c_char_p
    assert isinstance(lib.singa_last_error_string(), bytes)


def test_binding_table_matches_header():
    from singa_amd import _capi
    assert sorted(_capi.EXPORTS) == declared_symbols()
    # the test / lab switches live in their own header, outside the drop-in ABI
    assert sorted(_capi.LAB_EXPORTS) == declared_symbols("singa_hip_lab.h")
    assert not set(_capi.LAB_EXPORTS) & set(_capi.EXPORTS)


def test_argument_errors_without_gpu(built_lib):
    """Argument validation happens before any HIP call, so the error convention can be checked on the CPU."""
    from singa_amd import _capi
    lib = _capi.bind(built_lib)
    assert lib.singa_wigner_rows(None, None, 4, 6, 2, None) == -1            # SINGA_E_NULL
    kr, wsz, rr = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert lib.singa_dims(6, 2, ctypes.byref(kr), ctypes.byref(wsz), ctypes.byref(rr)) == 0
    assert (kr.value, wsz.value, rr.value) == (29, 236, 18)      # 235 Wigner entries, padded to 16-byte records
    assert lib.singa_dims(5, 2, None, None, None) == -2                      # SINGA_E_LMAX (built for 2, 4, 6)
    assert b"lmax" in lib.singa_last_error_string()


def test_product_refuses_cpu_tensors():
    import torch
    from singa_amd import ops
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.so3_rmsnorm(torch.zeros(2, 9, 16), torch.ones(3, 16), torch.zeros(16), 2)
