"""Host-side data layer (CPU): collate semantics, synthetic generator determinism, Laplacian PE (single vs batched),
and - only where the reference checkout is mounted - reading its pickled example graphs without torch_geometric."""
import os

import numpy as np
import pytest
import torch

from singa_amd import graph as G
from tests.helpers import GOLDEN, NAMES

REF_EXAMPLE = "/root/reference/example"


def test_collate_offsets_and_ptr():
    gs = [G.load_npz(os.path.join(GOLDEN, f"graph_{n}.npz"), with_lap=False) for n in NAMES]
    b = G.collate(gs)
    np_, nl_ = [g[G.PA]["x"].shape[0] for g in gs], [g[G.LA]["x"].shape[0] for g in gs]
    assert b[G.PA]["ptr"].tolist() == [0, np_[0], np_[0] + np_[1], sum(np_)]
    assert b[G.LA]["batch"].bincount().tolist() == nl_
    lp = b[G.E_LP]["edge_index"]
    e0 = gs[0][G.E_LP]["edge_index"].shape[1]
    # second graph's ligand->protein edges are offset by (ligand count, protein count) of the first graph
    assert torch.equal(lp[:, e0:e0 + 5], gs[1][G.E_LP]["edge_index"][:, :5] + torch.tensor([[nl_[0]], [np_[0]]]))
    assert b["ligand_data"]["smiIndices_input"].shape == (3, 200)
    assert b["ligand_data"]["vina_score"].shape == (3,)
    # PL edges mirror LP edges in the same order (Q5)
    assert torch.equal(b[G.E_PL]["edge_index"], b[G.E_LP]["edge_index"].flip(0))


def test_synthetic_graph_is_deterministic_and_well_formed():
    a, b = G.synthetic_graph(5), G.synthetic_graph(5)
    assert torch.equal(a[G.PA]["pos"], b[G.PA]["pos"]) and torch.equal(a[G.E_PP]["edge_index"], b[G.E_PP]["edge_index"])
    assert a[G.PA]["x"].shape == (200, 59) and a[G.LA]["x"].shape == (30, 59)
    assert a[G.E_PP]["edge_index"].shape == (2, 1700) and a[G.E_LP]["edge_index"].shape == (2, 118)
    pos = torch.cat([a[G.PA]["pos"], a[G.LA]["pos"]])
    d = torch.cdist(pos, pos) + torch.eye(pos.shape[0]) * 10
    assert float(d.min()) >= 1.0 - 1e-5                      # keeps the reference's distance guard quiet (EF:2292)
    ei = a[G.E_PP]["edge_index"]
    key = set(map(tuple, ei.t().tolist()))
    assert all((j, i) in key for i, j in key)                # linked_to is symmetric
    tok = a["ligand_data"]["smiIndices_input"][0]
    assert int(tok[0]) == G.START_TOKEN and int(tok[-1]) == G.PAD_TOKEN
    assert int(a[G.PA]["x"][:, -15:].long().max()) <= 1      # barcode bits (column 51 truncates to 0)


def test_laplacian_pe_batched_matches_per_graph():
    """The batched CPU route and the per-graph routine encode the same graphs.  Eigenvectors of repeated eigenvalues are
    basis-dependent (and a cluster may straddle the 8-column cut), so both are held to the basis-free properties: orthonormal
    columns, an invariant subspace of the graph's Laplacian, Ritz values = eigenvalues 1..8."""
    import numpy as np
    from oracle import singa_oracle as O
    gs = [G.load_npz(os.path.join(GOLDEN, f"graph_{n}.npz")) for n in NAMES]
    b = G.collate(gs)
    pe = G.laplacian_pe_batched(b[G.E_LL]["edge_index"], b[G.LA]["batch"], 3)
    ref = b[G.LA]["lap_pe"]
    for i in range(3):
        m = b[G.LA]["batch"] == i
        lap, w = O.laplacian_spectrum(gs[i][G.E_LL]["edge_index"].numpy(), int(m.sum()))
        for v in (pe[m].double().numpy(), ref[m].double().numpy()):
            assert np.abs(v.T @ v - np.eye(8)).max() < 1e-5
            ritz = v.T @ lap @ v
            assert np.abs(lap @ v - v @ ritz).max() < 1e-5
            assert np.abs(np.linalg.eigvalsh(ritz) - w[1:9]).max() < 1e-5


@pytest.mark.skipif(not os.path.isdir(REF_EXAMPLE), reason="reference checkout not mounted (GPU box)")
def test_load_reference_pt_without_pyg():
    for n in NAMES:
        g = G.load_reference_pt(os.path.join(REF_EXAMPLE, f"{n}.pt"), with_lap=False)
        z = np.load(os.path.join(GOLDEN, f"graph_{n}.npz"))
        assert np.array_equal(g[G.PA]["x"].numpy(), z["x_p"]) and np.array_equal(g[G.E_LP]["edge_index"].numpy(), z["ei_lp"])
        assert np.array_equal(g["atomicnum"][G.LA].numpy(), z["z_l"])
        assert abs(g["ligand_data"]["vina_score"] - float(z["props"][0])) < 1e-6


def test_laplacian_pe_is_canonical_in_repeated_eigenspaces():
    """graph.laplacian_pe on a graph with many connected components (a many-fold zero eigenvalue, as every bonded pocket graph
    has): the encoding must not depend on the basis LAPACK happens to return for a repeated eigenvalue's subspace (it differs
    with the BLAS thread count of the process: round 3's bench 'discrepancy').  _canonical_eigenbasis is a function of the
    subspace: rotating the cluster's eigenvectors by a random orthogonal matrix leaves its result unchanged."""
    import numpy as np
    from singa_amd import graph as G
    rng = np.random.default_rng(0)
    n = 60
    # 12 paths of 5 atoms: 12-fold zero eigenvalue and every other eigenvalue 12-fold too
    src = np.concatenate([np.arange(5 * c, 5 * c + 4) for c in range(12)])
    ei = np.stack([np.concatenate([src, src + 1]), np.concatenate([src + 1, src])])
    a = np.zeros((n, n)); a[ei[0], ei[1]] = 1.0
    dinv = np.clip(a.sum(0), 1, None) ** -0.5
    lap = np.eye(n) - dinv[:, None] * a * dinv[None, :]
    w, v = np.linalg.eigh(lap)
    v2 = v.copy()
    q, _ = np.linalg.qr(rng.standard_normal((12, 12)))
    v2[:, :12] = v[:, :12] @ q                                   # another basis of the zero eigenvalue's subspace
    c1, c2 = G._canonical_eigenbasis(w, v, 9), G._canonical_eigenbasis(w, v2, 9)
    assert np.abs(c1[:, :12] - c2[:, :12]).max() < 1e-12
    assert np.abs(c1[:, :12].T @ c1[:, :12] - np.eye(12)).max() < 1e-12       # still an orthonormal basis ...
    assert np.abs(lap @ c1[:, :12]).max() < 1e-12                              # ... of the same eigenspace
    pe = G.laplacian_pe(ei, n).double().numpy()
    assert np.abs(pe.T @ pe - np.eye(8)).max() < 1e-5 and np.abs(lap @ pe).max() < 1e-5
