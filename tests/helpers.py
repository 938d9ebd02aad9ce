"""Shared test helpers (CPU side): golden loading + synthetic weights."""
import os

import numpy as np
import torch

from oracle import weights as W

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["3wi2_4tpp", "4agq_5a7b", "5cp5_4nue"]


def state_from_spec(tag):
    z = np.load(os.path.join(GOLDEN, f"param_spec_{tag}.npz"))
    spec = [(str(n), tuple(int(v) for v in str(s).split(",")) if str(s) else (), float(m), float(sd))
            for n, s, m, sd in zip(z["names"], z["shapes"], z["mean"], z["std"])]
    return W.synth_state(spec)


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def rel_err(a, b):
    a, b = torch.as_tensor(a, dtype=torch.float64), torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm() / (b.norm() + 1e-30))
