"""Deterministic synthetic weights shared by golden generation, the oracle and the parity tests.

TEST INFRASTRUCTURE (see oracle/README.md). A full SINGA state dict is 107 MB, far too large to commit, and
the reference's own initialisation order cannot be replayed without its code, so golden runs overwrite every
parameter of the reference model with values that any process can regenerate from (name, shape, mean, std):
`numpy.random.RandomState` streams are frozen by NumPy's compatibility policy.
"""
import zlib

import numpy as np
import torch


def synth_tensor(name, shape, mean, std):
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    s = std if std > 0 else 0.05
    return torch.tensor(mean + s * rs.standard_normal(tuple(shape)), dtype=torch.float32)


def synth_state(spec):
    """spec: iterable of (name, shape, mean, std) -> {name: tensor}."""
    return {n: synth_tensor(n, sh, m, s) for n, sh, m, s in spec}


def spec_from_module(module):
    out = []
    for n, p in module.named_parameters():
        out.append((n, tuple(p.shape), float(p.detach().mean()), float(p.detach().std()) if p.numel() > 1 else 0.0))
    return out


def sample_index(n, cap=512):
    """Indices of the gradient elements recorded per parameter in the golden files: up to `cap` evenly spaced positions
    of the flattened tensor (deterministic; the tests recompute the same positions)."""
    import numpy as np
    return np.unique(np.linspace(0, n - 1, min(n, cap)).round().astype(np.int64))
