"""-m gpu: the product modules (HIP kernels + library GEMMs) against (a) tensors produced by the REFERENCE itself
(tests/golden, oracle/make_golden.py) and (b) the CPU oracle, on the bundled example graphs.  Tolerance 1e-4
relative fp32 (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from oracle import singa_oracle as O
from tests.helpers import NAMES, golden, grad_sample_errors, pinned_relu_ties, product_batch, rel_err, state_from_spec

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build_embedding(L):
    from singa_amd.config import load_config
    from singa_amd.model.Embedding import EquivariantEmbedding
    emb = EquivariantEmbedding(load_config(lmax=L).embedding, device=DEV)
    missing, unexpected = emb.load_state_dict(state_from_spec(f"embed_L{L}"), strict=False)
    assert not unexpected and all("offset" in m for m in missing), (missing, unexpected)
    return emb


@pytest.mark.parametrize("L", [2, 4, 6])
@pytest.mark.parametrize("name", NAMES)
def test_embedding_matches_reference(L, name):
    from singa_amd.graph import LA, PA
    emb = build_embedding(L)
    z = golden(f"embed_L{L}_{name}.npz")
    g = product_batch([name], z)
    out = emb(g)
    st = int(z["node_stride"])
    assert rel_err(out[PA].embedding[::st].detach().cpu(), z["out_p"]) < 1e-4
    assert rel_err(out[LA].embedding.detach().cpu(), z["out_l"]) < 1e-4
    assert rel_err(out["lp_edge"].embedding[::st].detach().cpu(), z["out_lp"]) < 1e-4
    assert rel_err(out["pl_edge"].embedding.detach().cpu(), z["out_pl"]) < 1e-4
    loss = (out[PA].embedding ** 2).sum() + (out[LA].embedding ** 2).sum()
    assert abs(float(loss) - float(z["loss"])) < 1e-4 * float(z["loss"])
    loss.backward()
    params = dict(emb.named_parameters())
    bad = []
    for n, ref in zip(z["grad_names"], z["grad_norms"]):
        gr = params[str(n)].grad
        if ref < 0:
            if gr is not None and float(gr.norm()) != 0.0:
                bad.append((str(n), "unexpected gradient"))
        elif gr is None or abs(float(gr.norm()) - ref) > 2e-3 * ref + 1e-6:
            bad.append((str(n), None if gr is None else float(gr.norm()), float(ref)))
    assert not bad, bad[:8]
    for k in z.files:
        if k.startswith("grad:"):
            assert rel_err(params[k[5:]].grad.cpu(), z[k]) < 2e-3, k


@pytest.mark.parametrize("L", [2, 6])
def test_block0_intermediates_match_reference(L):
    """edge-degree embedding, norm_1, attention, FFN and block output of blocks[0] on the protein pass."""
    from singa_amd.graph import E_PP, PA
    from singa_amd.model.EF_layers import SO3_Embedding
    from singa_amd.model.Embedding import barcode
    name = NAMES[1]
    emb = build_embedding(L)
    z = golden(f"embed_L{L}_{name}.npz")
    g = product_batch([name], z)
    st = int(z["node_stride"])
    zt, pos, ei = g["atomicnum"][PA], g[PA]["pos"], g[E_PP]["edge_index"]
    ev = pos[ei[0]] - pos[ei[1]]
    emb.SO3_rotation[0].set_wigner(g.extras["edge_rot_mat"]["pp"])
    xe = emb._edge_scalars(ev.norm(dim=-1), zt, zt, ei)
    ed = emb.edge_degree_embedding(zt, xe, ei, hetero=False).embedding
    assert rel_err(ed[::st].detach().cpu(), z["edge_degree_pp"]) < 1e-4
    x0 = torch.zeros(zt.shape[0], (L + 1) ** 2, 16, device=DEV)
    x0[:, 0] = (emb.sphere_embedding(zt) + emb.sphere_embedding_2(barcode(g[PA]["x"]))).long().float()
    x = SO3_Embedding(0, [L], 16, torch.float32, DEV, x0 + ed)
    blk = emb.blocks[0]
    xn = blk.norm_1(x.embedding)
    assert rel_err(xn[::st].detach().cpu(), z["b0_norm1_pp"]) < 1e-4
    out = blk(x=x, atomic_numbers=zt, edge_distance=xe, edge_index=ei, batch=len(zt), hetero=False)
    assert rel_err(out.embedding[::st].detach().cpu(), z["b0_out_pp"]) < 1e-4


@pytest.mark.parametrize("L", [2, 4, 6])
def test_singa_step_matches_reference(L):
    from singa_amd.config import load_config
    from singa_amd.model.GAN import SINGA
    sd = state_from_spec(f"singa_L{L}")
    z = golden(f"singa_L{L}_B3.npz")
    model = SINGA(load_config(lmax=L), device=DEV)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(("offset" in m) or m.endswith("pos_emb.pe") or "batch_norm" in m for m in missing), missing
    model.eval()
    g = product_batch(NAMES, z)
    with pinned_relu_ties(L) as pins:          # fp32 ties of the ReLU gates follow the reference's run (see the helper)
        logits = model(g)
        assert rel_err(logits.detach().cpu(), z["logits"]) < 1e-4
        loss = torch.nn.functional.cross_entropy(logits, g["ligand_data"]["smiIndices_tgt"].reshape(-1))
        assert abs(float(loss) - float(z["loss"])) < 1e-4 * float(z["loss"])
        loss.backward()
    assert pins.call == 18 and pins.flipped <= 16, (pins.call, pins.flipped)      # of 10.5 M gates; measured: 0 .. 3
    print(f"L={L}: {pins.flipped} near-zero ReLU gates pinned to the reference's choice")
    tot = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)))
    assert abs(tot - float(z["grad_total"])) < 1e-4 * float(z["grad_total"])      # measured: 1e-6 .. 1e-5 (tools/lab/grad_err_probe.py)
    params = dict(model.named_parameters())
    bad = []
    for n, ref in zip(z["grad_names"], z["grad_norms"]):
        gr = params[str(n)].grad
        if ref < 0:
            if gr is not None and float(gr.norm()) != 0.0:
                bad.append((str(n), "unexpected gradient"))
        # 3e-4: the golden's norms are float32 `p.grad.norm()` of the CPU reference, whose accumulation loses up to 1.4e-4 on
        # the 800 k-element fc_m0 weights (re-running the reference and taking the norm in float64 gives the HIP value to 3e-7)
        elif gr is None or abs(float(gr.norm()) - ref) > 3e-4 * ref + 1e-7:
            bad.append((str(n), None if gr is None else float(gr.norm()), float(ref)))
    assert not bad, bad[:8]
    # element-wise samples of every parameter's gradient against the reference's
    errs = sorted(grad_sample_errors({n: p.grad for n, p in params.items()}, z, 0.0), key=lambda e: -e[1])
    print(f"L={L}: worst element-wise gradient sample errors: " + ", ".join(f"{n} {e:.1e}" for n, e in errs[:3]))
    bad = [e for e in errs if e[1] > 1e-4]            # measured with the ties pinned: <= 3e-7 (3e-3 was needed before)
    assert not bad, bad[:8]


def test_own_knn_graph_matches_oracle_edges():
    """The on-GPU kNN + to_undirected + Laplacian edge builder gives the same edge set as the restated PyG path."""
    from singa_amd.model import CProMG as CP
    g = product_batch(NAMES, None)
    from singa_amd.graph import PA
    pos, batch = g[PA]["pos"], g[PA]["batch"]
    knn = CP.knn_graph(pos, 48, batch, 3)
    smear = CP.GaussianSmearing(stop=15, num_gaussians=64, device=DEV)
    e = CP.KnnEdges(pos, knn, smear)
    knn = knn.cpu()
    knn = knn[:, knn[0] >= 0]                      # absent slots (padding rows) are marked -1
    row, col, ea = O.knn_edges(pos.cpu(), knn, 15.0, 64)
    N = pos.shape[0]
    a = torch.sort(e.row.cpu() * N + e.col.cpu()).values
    b = torch.sort(row * N + col).values
    assert torch.equal(a, b)
    # every node has exactly 48 out-neighbours before symmetrisation
    assert int(torch.bincount(knn[0], minlength=N).min()) == 48
    # and the attributes agree edge by edge
    ia, ib = torch.sort(e.row.cpu() * N + e.col.cpu()).indices, torch.sort(row * N + col).indices
    assert float((e.attr.cpu()[ia] - ea[ib]).abs().max()) < 1e-5


def test_prepare_fills_laplacian_pe_on_gpu():
    """SINGA.prepare computes the per-graph Laplacian PE on the device when the batch does not carry one."""
    from singa_amd.config import load_config
    from singa_amd.graph import LA, PA
    from singa_amd.model.GAN import SINGA
    model = SINGA(load_config(lmax=2), device=DEV)
    g = product_batch(NAMES, None, with_lap=False)
    assert "lap_pe" not in g[PA]
    model.prepare(g)
    assert g[PA]["lap_pe"].shape == (g[PA]["x"].shape[0], 8) and g[LA]["lap_pe"].is_cuda
    logits = model(g)
    assert torch.isfinite(logits).all()
