"""Self-consistency of the third-party conventions that sit UNDER the reference and are restated here without the
packages themselves (e3nn 0.5.1 angle / S2-grid conventions, reference model/EF_layers.py:508-601, 2207-2229; SURVEY.md
§8c "identities usable as known-answer tests").  Every golden tensor flowed through oracle/shims, so agreement between
oracle and product alone would not notice a shared wrong convention; these identities tie both table builders
(oracle/so3_tables.py and singa_amd/so3.py), the oracle's Wigner construction from the shipped J matrices and the shim to
an INDEPENDENT definition of the real spherical harmonics (scipy's complex Y_l^m).  CPU only."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import singa_oracle as O
from oracle import so3_tables as T
from singa_amd import so3 as S
from tests.helpers import GOLDEN, NAMES, golden

try:                                       # scipy >= 1.15
    from scipy.special import sph_harm_y

    def _ylm(l, m, polar, azimuth):
        return sph_harm_y(l, m, polar, azimuth)
except ImportError:                        # older scipy
    from scipy.special import sph_harm

    def _ylm(l, m, polar, azimuth):
        return sph_harm(m, l, azimuth, polar)


def real_sh(L, xyz):
    """Real spherical harmonics in the convention the reference inherits from e3nn: y is the polar axis
    (beta = acos y, alpha = atan2(x, z)), 'integral' normalisation, no Condon-Shortley phase, m < 0 <-> sin(|m| alpha),
    m > 0 <-> cos(m alpha) - built from scipy's complex harmonics (which carry the (-1)^m phase)."""
    xyz = np.asarray(xyz, dtype=np.float64)
    r = np.linalg.norm(xyz, axis=1)
    beta = np.arccos(np.clip(xyz[:, 1] / r, -1, 1))
    alpha = np.arctan2(xyz[:, 0], xyz[:, 2])
    cols = []
    for l in range(L + 1):
        for m in range(-l, l + 1):
            y = _ylm(l, abs(m), beta, alpha)
            cols.append(y.real if m == 0 else math.sqrt(2) * (-1) ** m * (y.real if m > 0 else y.imag))
    return np.stack(cols, 1)


def reference_frame_cases(name):
    """(edge vectors, the reference's uniform draws, the frames the reference built) for the three edge types of a
    bundled graph."""
    z, rr = golden(f"embed_L2_{name}.npz"), golden(f"rot_rand_{name}.npz")
    g = O.load_graph_npz(os.path.join(GOLDEN, f"graph_{name}.npz"))
    for k, (ei, ps, pd) in {"pp": ("ei_pp", "pos_p", "pos_p"), "ll": ("ei_ll", "pos_l", "pos_l"),
                            "lp": ("ei_lp", "pos_l", "pos_p")}.items():
        yield g[ps][g[ei][0]] - g[pd][g[ei][1]], torch.as_tensor(rr[k]), torch.as_tensor(z[f"rot_{k}"])


def _frames(n, seed):
    rng = np.random.default_rng(seed)
    vec = torch.tensor(rng.normal(size=(n, 3)), dtype=torch.float64)
    rand = torch.tensor(rng.random((n, 3)), dtype=torch.float64)
    return vec, rand, O.edge_rot_mat(vec, rand)


# ------------------------------------------------------------------------------------------------ edge frames
def test_edge_frame_is_a_rotation_taking_the_edge_to_plus_y():
    vec, _, R = _frames(64, 1)
    eye = torch.eye(3, dtype=torch.float64)
    assert float((R @ R.transpose(1, 2) - eye).abs().max()) < 1e-12
    assert float((torch.linalg.det(R) - 1).abs().max()) < 1e-12
    nx = vec / vec.norm(dim=1, keepdim=True)
    y = (R @ nx.unsqueeze(2)).squeeze(2)
    assert float((y - torch.tensor([0.0, 1.0, 0.0], dtype=torch.float64)).abs().max()) < 1e-12      # R x^ = +y^


@pytest.mark.parametrize("name", NAMES)
def test_edge_frames_match_reference_draws(name):
    """oracle.edge_rot_mat on the reference's own torch.rand_like draws (tests/golden/rot_rand_*.npz, recorded by
    oracle/make_golden.py) reproduces the frames the reference built (rot_pp / rot_ll / rot_lp of embed_L2_*.npz;
    reference model/EF_layers.py:2286-2351).  The product's k1 kernel is held to the same frames in
    tests/test_kernels_emul.py (the kernel source on the CPU) and tests/test_kernels_gpu.py (-m gpu)."""
    for vec, rand, want in reference_frame_cases(name):
        assert float((O.edge_rot_mat(vec, rand) - want).abs().max()) < 1e-6


# ------------------------------------------------------------------------------------------------ Wigner matrices
@pytest.mark.parametrize("L", [2, 4, 6])
def test_wigner_rotates_spherical_harmonics(L):
    """Y(R x) = D(R) Y(x) for the oracle's D = Za J Zb J Zc (EF:2207-2229 with the e3nn angle extraction of EF:508-517)
    against scipy's harmonics: ties the ZYZ angle conventions, the shipped J matrices and the coefficient ordering
    together.  Also: the l = 1 block is R itself, D D^T = I, D(identity) = I."""
    _, _, R = _frames(6, 3)
    D = O.wigner_dense(R.float(), L).double().numpy()
    x = np.random.default_rng(4).normal(size=(7, 3))
    Yx = real_sh(L, x)
    for e in range(R.shape[0]):
        YRx = real_sh(L, x @ R[e].numpy().T)
        assert np.abs(YRx - Yx @ D[e].T).max() < 5e-6
        assert np.abs(D[e][1:4, 1:4] - R[e].numpy()).max() < 1e-6            # (m = -1, 0, +1) <-> (x, y, z)
        assert np.abs(D[e] @ D[e].T - np.eye((L + 1) ** 2)).max() < 2e-6
    D0 = O.wigner_dense(torch.eye(3).unsqueeze(0), L).numpy()[0]
    assert np.abs(D0 - np.eye((L + 1) ** 2)).max() < 1e-6


def test_j_matrices():
    for l in range(7):
        J = T.jd(l)
        assert np.abs(J - J.T).max() < 1e-12 and np.abs(J @ J - np.eye(2 * l + 1)).max() < 1e-10
        assert np.abs(J - S.jd_blocks()[l]).max() == 0.0


# ------------------------------------------------------------------------------------------------ S2 grids
GRIDS = [(2, 2), (4, 2), (6, 2), (4, 4), (6, 6)]


@pytest.mark.parametrize("L,M", GRIDS)
def test_to_grid_is_sh_evaluation(L, M):
    """to_grid[(b, a), i] = n_l * Y_i(grid point (beta_b, alpha_a)) (* the l > M rescale), for both table builders:
    pins the Legendre tables, the alpha-harmonic ordering and the component normalisation sqrt(4 pi / (2l+1) / (L+1))."""
    rb = 2 * (L + 1)
    ra = 2 * (M + 1) + 1 if L == M else 2 * M + 1
    beta = (np.arange(rb) + 0.5) * math.pi / rb
    alpha = 2 * math.pi * np.arange(ra) / ra
    bb, aa = np.meshgrid(beta, alpha, indexing="ij")
    pts = np.stack([np.sin(bb) * np.sin(aa), np.cos(bb), np.sin(bb) * np.cos(aa)], -1).reshape(-1, 3)   # (x, y, z)
    Y = real_sh(L, pts)
    lm = T.full_lm(L)
    n_to = np.array([math.sqrt(4 * math.pi) / math.sqrt(2 * l + 1) / math.sqrt(L + 1) for l, m in lm])
    sc = np.array([math.sqrt((2 * l + 1) / (2 * M + 1)) if (l > M and L != M) else 1.0 for l, m in lm])
    want = (Y * n_to * sc)[:, T.reduced_index(L, M)]
    to_o, fr_o = T.s2_grid_mats(L, M)
    to_p, fr_p = S.s2_grid(L, M)
    G = rb * ra
    assert np.abs(to_o.reshape(G, -1) - want).max() < 2e-6        # the tables are rounded through float32 like e3nn's buffers
    assert np.abs(to_p - want).max() < 2e-6
    # the two builders (polynomial differentiation vs recurrences) agree exactly
    assert np.abs(to_p - to_o.reshape(G, -1)).max() < 1e-12 and np.abs(fr_p - fr_o.reshape(G, -1)).max() < 1e-12


@pytest.mark.parametrize("L,M", GRIDS)
def test_from_grid_inverts_to_grid(L, M):
    """from_grid o to_grid = identity on the band-limited coefficients (Kostelec-Rockmore quadrature is exact there);
    on the reduced [L][M < L] grids both matrices carry the l > M rescale (EF:571-578, 589-596), so the product is
    diag(rescale^2) - that is what the reference computes, and what is pinned here."""
    for to, fr in (tuple(t.reshape(-1, t.shape[-1]) for t in T.s2_grid_mats(L, M)), S.s2_grid(L, M)):
        A = fr.T @ to
        want = np.diag(T.rotate_inv_rescale(L, M) ** 2) if L != M else np.eye(A.shape[0])
        assert np.abs(A - want).max() < 5e-7


@pytest.mark.parametrize("L,M", GRIDS)
def test_separable_factors_rebuild_the_grid_matrices(L, M):
    """singa_amd.so3.s2_grid_factors (what the S2 kernels consume) x-checked against the dense matrices."""
    lay = S.layout(L, M)
    to, fr = S.s2_grid(L, M)
    for m_primary in (False, True):
        P, Q, A = S.s2_grid_factors(L, M, m_primary)
        cols = lay.to_m if m_primary else np.arange(lay.KR)
        order = np.concatenate([np.arange(-l, l + 1) for l in range(L + 1)])[lay.reduced][cols]
        to_sep = (P[:, None, :] * A[:, order + M][None, :, :]).reshape(-1, lay.KR)
        fr_sep = (Q[:, None, :] * A[:, order + M][None, :, :]).reshape(-1, lay.KR)
        assert np.abs(to_sep - to[:, cols]).max() < 1e-12 and np.abs(fr_sep - fr[:, cols]).max() < 1e-12


def test_shim_grids_equal_oracle_tables():
    """oracle/shims/e3nn (what the REFERENCE ran on when the goldens were made) and oracle/so3_tables.py (what the
    oracle runs on) build the same to/from-grid matrices through different code paths (einsum over an expansion tensor
    vs direct columns)."""
    import sys
    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "shims")
    sys.path.insert(0, shim)
    try:
        from e3nn import o3
    finally:
        sys.path.remove(shim)
    for L, M in GRIDS:
        rb = 2 * (L + 1)
        ra = 2 * (M + 1) + 1 if L == M else 2 * M + 1
        tg, fg = o3.ToS2Grid(L, (rb, ra)), o3.FromS2Grid((rb, ra), L)
        to = torch.einsum("mbi,am->bai", tg.shb, tg.sha).double().numpy()          # EF:565-569
        fr = torch.einsum("am,mbi->bai", fg.sha, fg.shb).double().numpy()          # EF:583-587
        if L != M:
            sc = np.array([math.sqrt((2 * l + 1) / (2 * M + 1)) if l > M else 1.0 for l, m in T.full_lm(L)])
            to, fr = to * sc, fr * sc
        idx = T.reduced_index(L, M)
        to_o, fr_o = T.s2_grid_mats(L, M)
        assert np.abs(to[:, :, idx] - to_o).max() < 2e-6 and np.abs(fr[:, :, idx] - fr_o).max() < 2e-5 * np.abs(fr_o).max()
    a, b = o3.xyz_to_angles(torch.tensor([[0.3, -0.5, 0.8]]))
    R = o3.angles_to_matrix(a, b, torch.zeros(1))
    assert float((R @ torch.tensor([0.0, 1.0, 0.0]) - torch.nn.functional.normalize(torch.tensor([[0.3, -0.5, 0.8]]))).abs().max()) < 1e-6


# ------------------------------------------------------------------------------------------------ index algebra
@pytest.mark.parametrize("L", [2, 4, 6])
def test_coefficient_orderings(L):
    """SURVEY A1 (EF:1441-1474, 1514-1526): reduced index list and the m-primary permutation, oracle vs product, and the
    literal lists the survey quotes for L = 6."""
    lay = S.layout(L, 2)
    assert np.array_equal(lay.reduced, T.reduced_index(L, 2))
    perm, m_size = T.m_primary_perm(L, 2)
    assert np.array_equal(lay.to_m, perm) and list(lay.m_size) == list(m_size)
    assert sorted(perm.tolist()) == list(range(lay.KR))                       # to_m is a permutation
    if L == 6:
        assert lay.reduced.tolist() == [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13, 14, 18, 19, 20, 21, 22, 28, 29, 30, 31,
                                        32, 40, 41, 42, 43, 44]
        assert perm.tolist() == [0, 2, 6, 11, 16, 21, 26, 3, 7, 12, 17, 22, 27, 1, 5, 10, 15, 20, 25, 8, 13, 18, 23, 28, 4,
                                 9, 14, 19, 24]
        assert np.allclose(T.rotate_inv_rescale(6, 2)[[9, 14, 19, 24]], [1.1832, 1.3416, 1.4832, 1.6125], atol=5e-5)
