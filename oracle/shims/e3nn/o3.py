"""Restatement of the four e3nn.o3 (0.5.1) entry points the reference calls.

Call sites: `o3.xyz_to_angles`, `o3.angles_to_matrix` (reference model/EF_layers.py:510-514);
`ToS2Grid(...).shb/.sha`, `FromS2Grid(...).shb/.sha` (reference model/EF_layers.py:562-587).

Published conventions restated here (e3nn docs, "o3/_rotation.py", "o3/_s2grid.py"):
  * y is the polar axis: beta = acos(y), alpha = atan2(x, z).
  * angles_to_matrix(a, b, c) = Ry(a) Rx(b) Ry(c).
  * S2 grid: beta_i = (i + 1/2) pi / res_beta, alpha_j = 2 pi j / res_alpha.
  * real spherical harmonics split as shb (associated Legendre part, "integral"
    normalised, P(l,-m) = P(l,m), no Condon-Shortley phase) times sha
    ([sqrt2 sin(L a) .. sqrt2 sin(a), 1, sqrt2 cos(a) .. sqrt2 cos(L a)]).
  * 'component' normalisation and Kostelec-Rockmore quadrature weights for FromS2Grid.
No reference test pins these: parity at this boundary is UNPINNED (self-consistency
checks live in tests/test_oracle_conventions.py).
"""
import math

import numpy as np
import torch


def xyz_to_angles(xyz):
    xyz = torch.nn.functional.normalize(xyz, p=2, dim=-1)
    xyz = xyz.clamp(-1, 1)
    beta = torch.acos(xyz[..., 1])
    alpha = torch.atan2(xyz[..., 0], xyz[..., 2])
    return alpha, beta


def _matrix_x(angle):
    c, s = angle.cos(), angle.sin()
    o, z = torch.ones_like(angle), torch.zeros_like(angle)
    return torch.stack([torch.stack([o, z, z], -1), torch.stack([z, c, -s], -1), torch.stack([z, s, c], -1)], -2)


def _matrix_y(angle):
    c, s = angle.cos(), angle.sin()
    o, z = torch.ones_like(angle), torch.zeros_like(angle)
    return torch.stack([torch.stack([c, z, s], -1), torch.stack([z, o, z], -1), torch.stack([-s, z, c], -1)], -2)


def angles_to_matrix(alpha, beta, gamma):
    alpha, beta, gamma = torch.broadcast_tensors(alpha, beta, gamma)
    return _matrix_y(alpha) @ _matrix_x(beta) @ _matrix_y(gamma)


def _legendre_table(lmax, z, y):
    """[b, sum_l (2l+1)] table of P~_l^{|m|}(z, y), m = -l..l, 'integral' normalisation."""
    cols = []
    for l in range(lmax + 1):
        base = np.polynomial.Polynomial([-1.0, 0.0, 1.0]) ** l  # (z^2 - 1)^l
        vals = {}
        for m in range(l + 1):
            d = base.deriv(l + m) if (l + m) > 0 else base
            p = d(z) * (y ** m) / (2.0 ** l * math.factorial(l))
            p = p * math.sqrt((2 * l + 1) / (4 * math.pi) * math.factorial(l - m) / math.factorial(l + m))
            vals[m] = p
        for m in range(-l, l + 1):
            cols.append(vals[abs(m)])
    return np.stack(cols, axis=1)


def _s2_grid(res_beta, res_alpha):
    betas = (np.arange(res_beta) + 0.5) / res_beta * math.pi
    alphas = np.arange(res_alpha) / res_alpha * 2 * math.pi
    return betas, alphas


def _sha(lmax, alphas):
    a = alphas[:, None]
    sin = np.sin(np.arange(lmax, 0, -1)[None, :] * a)
    cos = np.cos(np.arange(1, lmax + 1)[None, :] * a)
    return np.concatenate([math.sqrt(2) * sin, np.ones_like(a), math.sqrt(2) * cos], axis=1)  # [a, 2L+1]


def _expand(lmax):
    m = np.zeros((lmax + 1, 2 * lmax + 1, (lmax + 1) ** 2))
    i = 0
    for l in range(lmax + 1):
        m[l, lmax - l: lmax + l + 1, i: i + 2 * l + 1] = np.eye(2 * l + 1)
        i += 2 * l + 1
    return m


def _quadrature_weights(b):
    k = np.arange(b)
    w = np.array([
        (2.0 / b) * math.sin(math.pi * (2.0 * j + 1.0) / (4.0 * b))
        * ((1.0 / (2 * k + 1)) * np.sin((2 * j + 1) * (2 * k + 1) * math.pi / (4.0 * b))).sum()
        for j in range(2 * b)
    ])
    return w / (2.0 * (2 * b) ** 2)


class ToS2Grid(torch.nn.Module):
    def __init__(self, lmax=None, res=None, normalization="component", dtype=None, device=None):
        super().__init__()
        res_beta, res_alpha = res
        betas, alphas = _s2_grid(res_beta, res_alpha)
        shb = _legendre_table(lmax, np.cos(betas), np.abs(np.sin(betas)))
        sha = _sha(lmax, alphas)
        if normalization == "component":
            n = math.sqrt(4 * math.pi) * np.array([1 / math.sqrt(2 * l + 1) for l in range(lmax + 1)]) / math.sqrt(lmax + 1)
        elif normalization == "norm":
            n = math.sqrt(4 * math.pi) * np.ones(lmax + 1) / math.sqrt(lmax + 1)
        else:
            n = np.ones(lmax + 1)
        m = _expand(lmax)
        shb = np.einsum("lmj,bj,lmi,l->mbi", m, shb, m, n)
        self.register_buffer("sha", torch.tensor(sha, dtype=torch.float32, device=device))
        self.register_buffer("shb", torch.tensor(shb, dtype=torch.float32, device=device))


class FromS2Grid(torch.nn.Module):
    def __init__(self, res=None, lmax=None, normalization="component", dtype=None, device=None):
        super().__init__()
        res_beta, res_alpha = res
        betas, alphas = _s2_grid(res_beta, res_alpha)
        shb = _legendre_table(lmax, np.cos(betas), np.abs(np.sin(betas)))
        sha = _sha(lmax, alphas)
        if normalization == "component":
            n = math.sqrt(4 * math.pi) * np.array([math.sqrt(2 * l + 1) for l in range(lmax + 1)]) * math.sqrt(lmax + 1)
        elif normalization == "norm":
            n = math.sqrt(4 * math.pi) * np.ones(lmax + 1) * math.sqrt(lmax + 1)
        else:
            n = 4 * math.pi * np.ones(lmax + 1)
        m = _expand(lmax)
        assert res_beta % 2 == 0
        qw = _quadrature_weights(res_beta // 2) * res_beta ** 2 / res_alpha
        shb = np.einsum("lmj,bj,lmi,l,b->mbi", m, shb, m, n, qw)
        self.register_buffer("sha", torch.tensor(sha, dtype=torch.float32, device=device))
        self.register_buffer("shb", torch.tensor(shb, dtype=torch.float32, device=device))
