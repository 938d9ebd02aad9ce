"""-m gpu: the HIP-graph replayed training step equals the eager step (same init, same batch, dropout off)."""
import copy

import pytest
import torch

from tests.helpers import NAMES, golden, product_batch, state_from_spec

pytestmark = pytest.mark.gpu


def _run(use_graph, steps=3, torch_adam=False):
    from singa_amd.config import load_config
    from singa_amd.engine import TrainStep
    from singa_amd.model.GAN import SINGA
    L = 2
    model = SINGA(load_config(lmax=L), device="cuda")
    model.load_state_dict(state_from_spec(f"singa_L{L}"), strict=False)
    model.eval()                                            # dropout off so that both runs see the same numbers
    from singa_amd.optim import Adam
    opt = (torch.optim.Adam(model.parameters(), lr=1e-4, betas=(0.99, 0.999)) if torch_adam
           else Adam(model.parameters(), lr=1e-4, betas=(0.99, 0.999)))
    eng = TrainStep(model, opt, None, use_graph=use_graph)
    z = golden(f"singa_L{L}_B3.npz")
    batch = product_batch(NAMES, z)
    losses = []
    for _ in range(steps + (2 if use_graph else 0)):        # graph capture spends 2 real warm-up steps first
        losses.append(float(eng.step(batch).detach()))
    return losses, eng


def test_graph_replay_matches_eager():
    eager, _ = _run(False, steps=5)
    graph, eng = _run(True, steps=3)
    assert eng.captures == 1
    # the graph engine's first call performs 2 un-recorded warm-up steps, so its i-th loss is eager's (i+2)-th
    for a, b in zip(graph[:3], eager[2:5]):
        assert abs(a - b) < 2e-4 * abs(b), (graph, eager)
    assert eager[4] < eager[0]                              # and the step does train


def test_graph_replay_gradients_match_eager_per_parameter():
    """After a few replayed steps, the gradients one more forward+backward replay leaves in .grad equal the eager
    gradients at the same parameters - for every parameter (guards against stale / uninitialised buffers in replay)."""
    from singa_amd.config import load_config
    from singa_amd.engine import TrainStep
    from singa_amd.model.GAN import SINGA
    _, eng = _run(True, steps=2)
    z = golden("singa_L2_B3.npz")
    batch = product_batch(NAMES, z)
    assert eng._load(batch)
    eng.g_fb.replay()
    torch.cuda.synchronize()
    ref = SINGA(load_config(lmax=2), device="cuda")
    ref.load_state_dict(eng.model.state_dict())
    ref.eval()
    ref.prepare(batch)
    logits = ref(batch)
    torch.nn.functional.cross_entropy(logits, batch["ligand_data"]["smiIndices_tgt"].reshape(-1)).backward()
    bad = []
    for (n, p), (_, q) in zip(eng.model.named_parameters(), ref.named_parameters()):
        if q.grad is None:
            assert p.grad is None, n
            continue
        err = float((p.grad - q.grad).norm() / (q.grad.norm() + 1e-12))
        if not (err < 2e-3) and float((p.grad - q.grad).abs().max()) > 1e-9:
            bad.append((n, err, float(p.grad.abs().max()), float(q.grad.abs().max())))
    assert not bad, bad[:8]


def test_fused_adam_matches_torch_adam():
    """The single-launch Adam follows torch.optim.Adam (eager, same init, same batch) step for step."""
    ours, _ = _run(False, steps=5)
    ref, _ = _run(False, steps=5, torch_adam=True)
    for a, b in zip(ours, ref):
        assert abs(a - b) < 1e-4 * abs(b), (ours, ref)


def test_prefetched_steps_equal_plain_steps():
    """TrainStep.prefetch (graph structure of the next batch built on a second stream) changes no number: the loss
    sequence of alternating prefetched batches equals the one of plain steps."""
    from singa_amd import graph as G
    from singa_amd.config import load_config
    from singa_amd.engine import TrainStep
    from singa_amd.model.GAN import SINGA
    from singa_amd.optim import Adam

    def run(prefetch):
        torch.manual_seed(5)
        model = SINGA(load_config(lmax=2), device="cuda").train()
        eng = TrainStep(model, Adam(model.parameters(), lr=1e-4), None, use_graph=True)
        bs = [G.synthetic_batch(2, first_id=40, n_protein=50, n_ligand=12, e_pp=300, e_ll=30, e_x=40).to("cuda") for _ in range(2)]
        out = []
        for i in range(6):
            out.append(float(eng.step(bs[i % 2]).detach()))
            if prefetch:
                eng.prefetch(bs[(i + 1) % 2])
        return out

    a, b = run(False), run(True)
    # not bit-equal even without prefetch: torch's index_add_ (atomics) orders its sums differently from run to run
    assert all(abs(x - y) < 2e-5 * abs(x) for x, y in zip(a, b)), (a, b)
