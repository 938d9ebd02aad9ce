"""-m gpu: BASELINE.json's configurations as parity cases.  Two-graph batches drawn from the SAME generators bench.py
uses (singa_amd.graph.WORKLOADS: config 2 = L 2, config 3 = ragged CrossDocked-shaped graphs at L 4, config 5 = 840
atoms / 8 k edges per graph at L 6) go through the product (HIP kernels, one full training step's forward + CrossEntropy
+ backward) and through the CPU oracle (pinned to the reference by tests/golden): logits within 1e-4 relative
(north_star), and the gradient of EVERY parameter as a whole tensor at 1e-4 (ReLU gates at fp32 ties pinned to the oracle's choice).  Plus the basis-free check of the on-GPU Laplacian
positional encoding (reference model/CProMG.py:562-571)."""
import numpy as np
import pytest
import torch

from oracle import singa_oracle as O
from tests.helpers import oracle_relu_ties, pinned_relu_ties, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("workload", ["cfg2_b32_l2", "cfg3_b128_l4", "cfg5_l6"])
def test_workload_step_matches_oracle(workload):
    from singa_amd import graph as G
    from singa_amd.config import load_config
    from singa_amd.model.GAN import SINGA
    L, kw, _, _ = G.resolve_workload(workload)
    ids = [3, 4]
    graphs = [G.synthetic_graph(i, **G.graph_sizes(i, **kw)) for i in ids]
    cfg = load_config(lmax=L)
    torch.manual_seed(7)
    model = SINGA(cfg, device=DEV)
    model.eval()                                   # dropout off on both sides (Q9)
    sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    # oracle (CPU): its own kNN graphs, frames from the same draws, the same Laplacian encodings
    b, rots, lap_p, lap_l = O.batch_from_graphs(graphs)
    with oracle_relu_ties() as ties:               # the oracle's choice at ReLU gates within fp32 reach of zero ...
        loss_o = O.train_step_loss(sd, b, rots, L, lap_p, lap_l)
    loss_o.backward()
    # product (GPU)
    batch = G.collate(graphs).to(DEV)
    with pinned_relu_ties(records=ties.records) as pins:          # ... is the product's choice too ("ReLU ties")
        logits = model(batch)
        loss = torch.nn.functional.cross_entropy(logits, batch["ligand_data"]["smiIndices_tgt"].reshape(-1))
        loss.backward()
    torch.cuda.synchronize()
    assert pins.call == 18 and pins.flipped <= 16, (pins.call, pins.flipped)
    assert abs(float(loss) - float(loss_o)) < 1e-4 * abs(float(loss_o)), (float(loss), float(loss_o))
    with torch.no_grad():
        bp = torch.repeat_interleave(torch.arange(2), b["ptr_p"][1:] - b["ptr_p"][:-1])
        bl = torch.repeat_interleave(torch.arange(2), b["ptr_l"][1:] - b["ptr_l"][:-1])
        ref = O.singa_forward(sd, b, rots, L, O.knn_graph(b["pos_p"], 48, bp), O.knn_graph(b["pos_l"], 30, bl), lap_p, lap_l)
    assert rel_err(logits.detach().cpu(), ref) < 1e-4
    total = float(torch.sqrt(sum((v.grad.double() ** 2).sum() for v in sd.values() if v.grad is not None)))
    bad, errs, n_checked = [], [], 0
    for name, p in model.named_parameters():
        go = sd[name].grad
        if go is None or float(go.norm()) == 0.0:
            assert p.grad is None or float(p.grad.norm()) == 0.0, name       # Q1 / Q10: no gradient on either side
            continue
        assert p.grad is not None, name
        n_checked += 1
        diff = (p.grad.detach().cpu().double() - go.double())
        err = float(diff.norm() / go.double().norm())
        # 1e-4 on EVERY parameter that carries a gradient (norm above 1e-7 of the total; below: the analytically-zero W_K.bias
        # gradients, rounding noise on both sides).  Measured with the ReLU ties pinned: 1.9e-5 on the smallest bias, <= 5e-6
        # elsewhere; before the ties were understood this needed 2e-3 (6e-3 for pos_ffn.conv1).
        if float(go.norm()) > 1e-7 * total:
            errs.append((err, name, float(go.norm()) / total))
            if err > 1e-4:
                bad.append((name, err, float(go.norm())))
    print(f"{workload}: {pins.flipped} ReLU gates pinned; largest per-parameter gradient errors: "
          + ", ".join(f"{n} {e:.1e} (norm {r:.0e} of total)" for e, n, r in sorted(errs, reverse=True)[:6]))
    assert n_checked == 634, n_checked                    # 634 of the 724 tensors carry gradients
    assert not bad, bad[:8]
    # total gradient norm: 1e-4 (north_star's bar; measured 3e-7 .. 1e-6 on these batches)
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)))
    assert abs(gn - total) < 1e-4 * total, (gn, total)


def test_cfg5_l6_b8_loss_matches_oracle():
    """BASELINE.json configs[4] at full PER-GRAPH size (800 protein + 40 ligand atoms, ~8 k edges per graph, l_max = 6) on an
    8-graph batch: the HIP path's logits and CrossEntropy against the CPU oracle's (forward only: the oracle needs ~20 s
    for it; gradients at this size are covered by the 2-graph case above)."""
    from singa_amd import graph as G
    from singa_amd.config import load_config
    from singa_amd.model.GAN import SINGA
    L, kw, ids, _ = G.resolve_workload("cfg5_l6_b8")
    graphs = [G.synthetic_graph(i, **G.graph_sizes(i, **kw)) for i in ids]
    torch.manual_seed(7)
    model = SINGA(load_config(lmax=L), device=DEV).eval()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    b, rots, lap_p, lap_l = O.batch_from_graphs(graphs)
    with torch.no_grad():
        bp = torch.repeat_interleave(torch.arange(len(ids)), b["ptr_p"][1:] - b["ptr_p"][:-1])
        bl = torch.repeat_interleave(torch.arange(len(ids)), b["ptr_l"][1:] - b["ptr_l"][:-1])
        ref = O.singa_forward(sd, b, rots, L, O.knn_graph(b["pos_p"], 48, bp), O.knn_graph(b["pos_l"], 30, bl), lap_p, lap_l)
        loss_o = torch.nn.functional.cross_entropy(ref, b["tok_tgt"].reshape(-1))
        batch = G.collate(graphs).to(DEV)
        logits = model(batch)
        loss = torch.nn.functional.cross_entropy(logits, batch["ligand_data"]["smiIndices_tgt"].reshape(-1))
    assert rel_err(logits.cpu(), ref) < 1e-4
    assert abs(float(loss) - float(loss_o)) < 1e-4 * abs(float(loss_o)), (float(loss), float(loss_o))


@pytest.mark.parametrize("workload", ["cfg3_b128_l4", "cfg5_l6"])
def test_laplacian_pe_batched_on_gpu_against_oracle_spectrum(workload):
    """graph.laplacian_pe_batched (batched symmetric eigensolve on the GPU, what SINGA.prepare uses when a batch carries
    no encoding) per graph: orthonormal columns, an invariant subspace of the oracle's Laplacian, Ritz values =
    eigenvalues 1..8, fixed sign convention.  Ragged batch (config-3 generator: dozens of small components per pocket, the
    dense route of singa_lap_pe) and config-5 graphs (one 800-atom component: the sparse route, Chebyshev-filtered subspace
    iteration)."""
    from singa_amd import graph as G
    kw = {k: v for k, v in G.WORKLOADS[workload].items() if k not in ("n_graphs", "lmax")}
    graphs = [G.synthetic_graph(i, with_lap=False, **G.graph_sizes(i, **kw)) for i in (11, 12, 13)]
    b = G.collate(graphs).to(DEV)
    for nt, et in ((G.PA, G.E_PP), (G.LA, G.E_LL)):
        pe = G.laplacian_pe_batched(b[et]["edge_index"], b[nt]["batch"], 3).cpu().double().numpy()
        ptr = b[nt]["ptr"].cpu().numpy()
        for i, g in enumerate(graphs):
            v = pe[ptr[i]:ptr[i + 1]]
            n = v.shape[0]
            lap, w = O.laplacian_spectrum(g[et]["edge_index"].numpy(), n)
            assert np.abs(v.T @ v - np.eye(8)).max() < 1e-5
            ritz = v.T @ lap @ v
            assert np.abs(lap @ v - v @ ritz).max() < 1e-5
            assert np.abs(np.linalg.eigvalsh(ritz) - w[1:9]).max() < 1e-5
            # sign convention: the entry of largest magnitude is positive (up to ties between mirror-image atoms)
            assert (v.max(0) >= (-v).max(0) - 1e-6).all()


@pytest.mark.parametrize("workload", ["cfg3_b128_l4", "cfg2_b32_l2"])
def test_eight_graph_step_matches_oracle(workload):
    """The first EIGHT graphs of a bench workload - the sample bench.py's `oracle_check` gates on; graph 7 of config 3 is the
    only one whose ligand (33 atoms) makes the k = 30 kNN a real selection, and the dense layouts are as wide as the widest of
    the eight.  Logits 1e-4, CrossEntropy of the batch and of every graph, total gradient norm 1e-4 (north_star; reference
    train.py:119-124).  Both sides read the same input tensors.  (Round 3's 6.5e-5 / 2.0e-4 gap on this sample was an input
    difference: the Laplacian encodings of graphs with a many-fold zero eigenvalue, computed by numpy in two processes with
    different BLAS thread counts - graph.laplacian_pe is canonical now and bench.py hands the oracle the same tensors.)"""
    import torch.nn.functional as F
    from singa_amd import graph as G
    from singa_amd.config import load_config
    from singa_amd.model.GAN import SINGA
    L, kw, ids, _ = G.resolve_workload(workload)
    ids = ids[:8]
    B = len(ids)
    graphs = [G.synthetic_graph(i, **G.graph_sizes(i, **kw)) for i in ids]
    torch.manual_seed(7)
    model = SINGA(load_config(lmax=L), device=DEV).eval()
    sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    b, rots, lap_p, lap_l = O.batch_from_graphs(graphs)
    bp = torch.repeat_interleave(torch.arange(B), b["ptr_p"][1:] - b["ptr_p"][:-1])
    bl = torch.repeat_interleave(torch.arange(B), b["ptr_l"][1:] - b["ptr_l"][:-1])
    ref = O.singa_forward(sd, b, rots, L, O.knn_graph(b["pos_p"], 48, bp), O.knn_graph(b["pos_l"], 30, bl), lap_p, lap_l)
    tgt = b["tok_tgt"].reshape(-1)
    loss_o = F.cross_entropy(ref, tgt)
    loss_o.backward()
    total = float(torch.sqrt(sum((v.grad.double() ** 2).sum() for v in sd.values() if v.grad is not None)))
    batch = G.collate(graphs).to(DEV)
    logits = model(batch)
    loss = F.cross_entropy(logits, batch["ligand_data"]["smiIndices_tgt"].reshape(-1))
    loss.backward()
    torch.cuda.synchronize()
    lg, rf = logits.detach().cpu(), ref.detach()
    assert rel_err(lg, rf) < 1e-4
    assert abs(float(loss) - float(loss_o)) < 1e-4 * abs(float(loss_o)), (float(loss), float(loss_o))
    T = tgt.numel() // B
    ce = F.cross_entropy(lg, tgt, reduction="none").view(B, T).mean(1)
    ce_o = F.cross_entropy(rf, tgt, reduction="none").view(B, T).mean(1)
    assert float(((ce - ce_o).abs() / ce_o).max()) < 1e-4, (ce.tolist(), ce_o.tolist())
    for i in range(B):                                            # per graph, so that one bad graph cannot hide in the mean
        assert rel_err(lg.view(B, T, -1)[i], rf.view(B, T, -1)[i]) < 1e-4, ids[i]
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)))
    assert abs(gn - total) < 1e-4 * total, (gn, total)
