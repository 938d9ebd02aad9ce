"""ORACLE (test infrastructure, see oracle/README.md): constant tables of the SO(3) machinery, float64 numpy.

Restates, with citations into /root/reference/model/EF_layers.py ("EF"):
  * coefficient orderings and the l<->m permutation                 EF:1441-1474, 1514-1526
  * rotate_inv rescale                                                 EF:1530-1549
  * S2 grid matrices as built by SO3_Grid from e3nn ToS2Grid/FromS2Grid EF:551-601
  * J matrices (data table shipped with the reference, model/Jd.pt)    EF:2195-2198
Independent of singa_amd/ (the product builds its own tables in singa_amd/so3.py).
"""
import math
import os
from functools import lru_cache

import numpy as np

_JD = None


def jd(l):
    global _JD
    if _JD is None:
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "Jd.npz"))
        _JD = [z[f"J{i}"] for i in range(len(z.files))]
    return _JD[l]


def full_lm(L):
    """(l, m) of every full l-primary coefficient: index l*l + l + m."""
    return [(l, m) for l in range(L + 1) for m in range(-l, l + 1)]


@lru_cache(None)
def reduced_index(L, M):
    """Indices (into the full (L+1)^2 list) of coefficients with |m| <= M, l-primary order. EF:1514-1526."""
    return np.array([l * l + l + m for (l, m) in full_lm(L) if abs(m) <= M], dtype=np.int64)


@lru_cache(None)
def m_primary_perm(L, M):
    """perm[i_m] = reduced l-primary position of the i_m-th m-primary coefficient; m_size[m]. EF:1463-1474."""
    red = [(l, m) for (l, m) in full_lm(L) if abs(m) <= M]
    pos = {lm: i for i, lm in enumerate(red)}
    perm, m_size = [], []
    for m in range(M + 1):
        plus = [pos[(l, m)] for l in range(m, L + 1)]
        perm += plus
        m_size.append(len(plus))
        if m > 0:
            perm += [pos[(l, -m)] for l in range(m, L + 1)]
    return np.array(perm, dtype=np.int64), m_size


@lru_cache(None)
def rotate_inv_rescale(L, M):
    """Per reduced coefficient: sqrt((2l+1)/(2M+1)) for l > M else 1. EF:1539-1547."""
    return np.array([math.sqrt((2 * l + 1) / (2 * M + 1)) if l > M else 1.0
                     for (l, m) in full_lm(L) if abs(m) <= M])


def _assoc_legendre(L, z, y):
    cols = []
    for l in range(L + 1):
        base = np.polynomial.Polynomial([-1.0, 0.0, 1.0]) ** l
        for m in range(-l, l + 1):
            a = abs(m)
            d = base.deriv(l + a) if l + a > 0 else base
            nrm = math.sqrt((2 * l + 1) / (4 * math.pi) * math.factorial(l - a) / math.factorial(l + a))
            cols.append(d(z) * y ** a / (2.0 ** l * math.factorial(l)) * nrm)
    return np.stack(cols, 1)  # [beta, K]


@lru_cache(None)
def s2_grid_mats(L, M):
    """(to_grid[b,a,Kr], from_grid[b,a,Kr]) exactly as SO3_Grid(L, M, normalization='component') registers them."""
    rb = 2 * (L + 1)
    ra = 2 * (M + 1) + 1 if L == M else 2 * M + 1
    beta = (np.arange(rb) + 0.5) / rb * math.pi
    alpha = np.arange(ra) / ra * 2 * math.pi
    leg = _assoc_legendre(L, np.cos(beta), np.abs(np.sin(beta)))  # [b, K]
    lm = full_lm(L)
    az = np.zeros((ra, len(lm)))
    for i, (l, m) in enumerate(lm):
        az[:, i] = 1.0 if m == 0 else math.sqrt(2) * (np.cos(m * alpha) if m > 0 else np.sin(-m * alpha))
    n_to = np.array([math.sqrt(4 * math.pi) / math.sqrt(2 * l + 1) / math.sqrt(L + 1) for (l, m) in lm])
    n_from = np.array([math.sqrt(4 * math.pi) * math.sqrt(2 * l + 1) * math.sqrt(L + 1) for (l, m) in lm])
    b = rb // 2
    k = np.arange(b)
    qw = np.array([(2.0 / b) * math.sin(math.pi * (2 * j + 1) / (4 * b))
                   * (np.sin((2 * j + 1) * (2 * k + 1) * math.pi / (4 * b)) / (2 * k + 1)).sum()
                   for j in range(2 * b)]) / (2.0 * (2 * b) ** 2) * rb ** 2 / ra
    # the e3nn buffers are float32; round the two factors there as the reference does before its einsum
    shb_to = (leg * n_to[None, :]).astype(np.float32).astype(np.float64)
    shb_from = (leg * n_from[None, :] * qw[:, None]).astype(np.float32).astype(np.float64)
    az = az.astype(np.float32).astype(np.float64)
    to = shb_to[:, None, :] * az[None, :, :]
    fr = shb_from[:, None, :] * az[None, :, :]
    if L != M:
        sc = np.array([math.sqrt((2 * l + 1) / (2 * M + 1)) if l > M else 1.0 for (l, m) in lm])
        to, fr = to * sc, fr * sc
    idx = reduced_index(L, M)
    return to[:, :, idx], fr[:, :, idx]
