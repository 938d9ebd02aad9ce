"""-m gpu: (a) a batch padded with inert atoms / edges (graph.pad_batch) gives the same logits, loss and parameter
gradients as the batch itself; (b) the bucketed engine replays ONE captured HIP graph for batches of different sizes
(ragged batches are the normal case: reference utils/Data.py:230, train.py:113-133) and its loss sequence equals the
eager engine's."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
KW = dict(n_ligand=12, e_ll=24, e_x=30)


def _batch(first, n_protein, e_pp, n=2):
    from singa_amd import graph as G
    return G.synthetic_batch(n, first_id=first, n_protein=n_protein, e_pp=e_pp, **KW).to(DEV)


def _model(seed=3):
    from singa_amd.config import load_config
    from singa_amd.model.GAN import SINGA
    torch.manual_seed(seed)
    return SINGA(load_config(lmax=2), device=DEV).eval()


def test_padded_batch_equals_unpadded():
    from singa_amd import graph as G
    from singa_amd.model import EF_layers
    model = _model()
    b = _batch(70, 44, 220)
    r = G.batch_sizes(b)

    def run(batch):
        EF_layers._edge_cache.clear()
        model.zero_grad(set_to_none=True)
        model.prepare(batch)
        logits = model(batch)
        loss = torch.nn.functional.cross_entropy(logits, batch["ligand_data"]["smiIndices_tgt"].reshape(-1))
        loss.backward()
        return logits.detach().clone(), float(loss), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    lo, loss, gr = run(b)
    pb = G.pad_batch(b, r[0] + 70, r[1] + 66, r[2] + 300, r[3] + 40, r[4] + 50)
    pb.extras["pad"].update(mx_p=48, mx_l=16, knn_p=None, knn_l=None)
    lo2, loss2, gr2 = run(pb)
    assert float((lo2 - lo).abs().max()) < 2e-5 * float(lo.abs().max()) and abs(loss2 - loss) < 1e-5 * loss
    assert set(gr) == set(gr2)
    total = float(torch.sqrt(sum((g.double() ** 2).sum() for g in gr.values())))
    # (the key biases of the dense attentions have a mathematically zero gradient - pure rounding noise, hence the floor)
    bad = [(n, float((gr2[n] - g).norm() / (g.norm() + 1e-12))) for n, g in gr.items()
           if float((gr2[n] - g).norm()) > 2e-4 * float(g.norm()) + 1e-7 * total]
    assert not bad, bad[:6]


def test_bucketed_replay_one_capture_for_ragged_batches():
    from singa_amd.engine import TrainStep
    from singa_amd.optim import Adam
    sizes = [(46, 230), (45, 226), (47, 234), (45, 224), (46, 228)]          # (protein atoms per graph, bonded edges)
    batches = [_batch(80 + 3 * i, n, e) for i, (n, e) in enumerate(sizes)]

    def run(**kw):
        model = _model(5)
        eng = TrainStep(model, Adam(model.parameters(), lr=1e-4), None, **kw)
        out = [float(eng.step(b).detach()) for b in batches]
        return out, eng

    eager, _ = run(use_graph=False)
    graph, eng = run(use_graph=True, bucket=True)
    assert eng.captures == 1, eng.captures                  # the first batch opens the class, the others fit in it
    assert all(abs(a - b) < 2e-4 * abs(b) for a, b in zip(graph, eager)), (graph, eager)
    # a much larger batch opens a second class (second capture) and widens the dense layout for good (the widest graph
    # seen so far sets it), so the next small batch is captured once more with the wider layout - after that both
    # classes replay their captures
    big = _batch(200, 80, 400)
    model = _model(5)
    eng = TrainStep(model, Adam(model.parameters(), lr=1e-4), None, use_graph=True, bucket=True)
    seq = [batches[0], big, batches[1], big, batches[2], big, batches[3]]
    got = [float(eng.step(b).detach()) for b in seq]
    assert eng.captures == 3
    model = _model(5)
    ref = TrainStep(model, Adam(model.parameters(), lr=1e-4), None, use_graph=False)
    want = [float(ref.step(b).detach()) for b in seq]
    assert all(abs(a - b) < 2e-4 * abs(b) for a, b in zip(got, want)), (got, want)
