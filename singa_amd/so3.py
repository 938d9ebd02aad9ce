"""Host-side constant tables for the SO(3) machinery (product code; independent of oracle/).

What the reference builds in Python at module construction (reference model/EF_layers.py, "EF"):
  CoefficientMappingModule (EF:1413-1552)  -> Layout: reduced index list, m-primary permutation, m sizes
  SO3_Grid (EF:531-621; e3nn ToS2Grid/FromS2Grid, component normalisation) -> s2_grid()
  Jd.pt (EF:2195-2198) -> jd_flat()
Everything is computed once in float64 numpy and cached; kernels receive fp32 copies.
"""
import math
import os
from functools import lru_cache

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


@lru_cache(None)
def jd_blocks():
    z = np.load(os.path.join(_DATA, "Jd.npz"))
    return tuple(np.ascontiguousarray(z[f"J{l}"], dtype=np.float64) for l in range(len(z.files)))


def jd_flat(lmax):
    return np.concatenate([jd_blocks()[l].reshape(-1) for l in range(lmax + 1)])


class Layout:
    """Index bookkeeping for one (lmax, mmax)."""

    def __init__(self, L, M):
        self.L, self.M = L, M
        self.K = (L + 1) ** 2
        self.degree = np.repeat(np.arange(L + 1), 2 * np.arange(L + 1) + 1)            # l of each full coefficient
        order = np.concatenate([np.arange(-l, l + 1) for l in range(L + 1)])           # m of each full coefficient
        keep = np.abs(order) <= M
        self.reduced = np.nonzero(keep)[0]                                            # EF:1514-1526
        self.KR = int(keep.sum())
        rl, rm = self.degree[keep], order[keep]
        # m-primary: all m=0 (by l), then for each m>0 the +m rows followed by the -m rows (EF:1463-1474)
        perm = [np.nonzero(rm == 0)[0]]
        self.m_size = [L + 1]
        for m in range(1, M + 1):
            perm += [np.nonzero(rm == m)[0], np.nonzero(rm == -m)[0]]
            self.m_size.append(L - m + 1)
        self.to_m = np.concatenate(perm)                  # m-primary row i  <- reduced l-primary row to_m[i]
        self.rad_rows = sum(self.m_size)
        # floats per edge of the reduced Wigner record, padded to a multiple of 4 (16-byte aligned records: so3_index.h)
        self.WSZ = (int(sum((2 * min(l, M) + 1) * (2 * l + 1) for l in range(L + 1))) + 3) // 4 * 4
        self.seg_rows = [self.m_size[0]] + [2 * s for s in self.m_size[1:]]
        self.seg_start = np.concatenate([[0], np.cumsum(self.seg_rows)]).tolist()


@lru_cache(None)
def layout(L, M):
    return Layout(L, M)


def _legendre(L, ct, st):
    """P~_l^{|m|}: associated Legendre functions in the 'integral' spherical-harmonic normalisation,
    no Condon-Shortley phase, P(l,-m) = P(l,m).  Stable upward recurrences in float64."""
    out = np.zeros((ct.shape[0], (L + 1) ** 2))
    P = {}
    for m in range(L + 1):
        # P_m^m = (2m-1)!! * st^m  (phase-free), P_{m+1}^m = (2m+1) ct P_m^m, then three-term recurrence in l
        pmm = np.ones_like(ct)
        for k in range(1, m + 1):
            pmm = pmm * (2 * k - 1) * st
        P[(m, m)] = pmm
        if m + 1 <= L:
            P[(m + 1, m)] = (2 * m + 1) * ct * pmm
        for l in range(m + 2, L + 1):
            P[(l, m)] = ((2 * l - 1) * ct * P[(l - 1, m)] - (l + m - 1) * P[(l - 2, m)]) / (l - m)
    for l in range(L + 1):
        for m in range(-l, l + 1):
            a = abs(m)
            n = math.sqrt((2 * l + 1) / (4 * math.pi) * math.factorial(l - a) / math.factorial(l + a))
            out[:, l * l + l + m] = n * P[(l, a)]
    return out


@lru_cache(None)
def s2_grid(L, M):
    """(to_grid, from_grid), each [G = res_beta*res_alpha, KR], columns in REDUCED L-PRIMARY order, as
    SO3_Grid(L, M, normalization='component', resolution=None) registers them (EF:551-601)."""
    lay = layout(L, M)
    rb = 2 * (L + 1)
    ra = 2 * (M + 1) + 1 if L == M else 2 * M + 1
    beta = (np.arange(rb) + 0.5) * math.pi / rb
    alpha = 2 * math.pi * np.arange(ra) / ra
    leg = _legendre(L, np.cos(beta), np.abs(np.sin(beta)))
    order = np.concatenate([np.arange(-l, l + 1) for l in range(L + 1)])
    az = np.where(order[None, :] == 0, 1.0,
                  math.sqrt(2) * np.where(order[None, :] > 0, np.cos(order[None, :] * alpha[:, None]),
                                          np.sin(-order[None, :] * alpha[:, None])))
    deg = lay.degree.astype(np.float64)
    n_to = math.sqrt(4 * math.pi) / np.sqrt(2 * deg + 1) / math.sqrt(L + 1)
    n_from = math.sqrt(4 * math.pi) * np.sqrt(2 * deg + 1) * math.sqrt(L + 1)
    b = rb // 2
    j = np.arange(2 * b)[:, None]
    k = np.arange(b)[None, :]
    qw = ((2.0 / b) * np.sin(math.pi * (2 * j[:, 0] + 1) / (4 * b))
          * (np.sin((2 * j + 1) * (2 * k + 1) * math.pi / (4 * b)) / (2 * k + 1)).sum(1)) / (2.0 * (2 * b) ** 2)
    qw = qw * rb ** 2 / ra
    f32 = lambda a: a.astype(np.float32).astype(np.float64)   # the reference holds sha/shb as fp32 buffers
    to = f32(leg * n_to)[:, None, :] * f32(az)[None, :, :]
    fr = f32(leg * n_from * qw[:, None])[:, None, :] * f32(az)[None, :, :]
    if L != M:
        sc = np.where(deg > M, np.sqrt((2 * deg + 1) / (2 * M + 1)), 1.0)
        to, fr = to * sc, fr * sc
    G = rb * ra
    return (np.ascontiguousarray(to.reshape(G, -1)[:, lay.reduced]),
            np.ascontiguousarray(fr.reshape(G, -1)[:, lay.reduced]))


@lru_cache(None)
def s2_grid_factors(L, M, m_primary):
    """Separable form of s2_grid(L, M): to_grid[(b,a), i] = P[b, i] * A[a, mc(i)], from_grid[(b,a), i] = Q[b, i] * A[a, mc(i)]
    (a real spherical-harmonic transform is a Legendre transform in beta followed by a Fourier transform in alpha).
    P, Q: [res_beta, KR] in the requested row order (reduced l-primary, or m-primary); A: [res_alpha, 2M+1] with
    column mc = m + M.  Evaluating the two stages separately costs ~3x fewer FMAs than the dense grid matrices."""
    lay = layout(L, M)
    rb = 2 * (L + 1)
    ra = 2 * (M + 1) + 1 if L == M else 2 * M + 1
    beta = (np.arange(rb) + 0.5) * math.pi / rb
    alpha = 2 * math.pi * np.arange(ra) / ra
    leg = _legendre(L, np.cos(beta), np.abs(np.sin(beta)))
    deg = lay.degree.astype(np.float64)
    n_to = math.sqrt(4 * math.pi) / np.sqrt(2 * deg + 1) / math.sqrt(L + 1)
    n_from = math.sqrt(4 * math.pi) * np.sqrt(2 * deg + 1) * math.sqrt(L + 1)
    b = rb // 2
    j = np.arange(2 * b)[:, None]
    k = np.arange(b)[None, :]
    qw = ((2.0 / b) * np.sin(math.pi * (2 * j[:, 0] + 1) / (4 * b))
          * (np.sin((2 * j + 1) * (2 * k + 1) * math.pi / (4 * b)) / (2 * k + 1)).sum(1)) / (2.0 * (2 * b) ** 2)
    qw = qw * rb ** 2 / ra
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    sc = np.where(deg > M, np.sqrt((2 * deg + 1) / (2 * M + 1)), 1.0) if L != M else np.ones_like(deg)
    P = f32(leg * n_to) * sc
    Q = f32(leg * n_from * qw[:, None]) * sc
    ms = np.arange(-M, M + 1)
    A = f32(np.where(ms[None, :] == 0, 1.0,
                     math.sqrt(2) * np.where(ms[None, :] > 0, np.cos(ms[None, :] * alpha[:, None]),
                                             np.sin(-ms[None, :] * alpha[:, None]))))
    cols = lay.reduced[lay.to_m] if m_primary else lay.reduced
    return np.ascontiguousarray(P[:, cols]), np.ascontiguousarray(Q[:, cols]), np.ascontiguousarray(A)
