"""-m gpu: the HIP-graph replayed training step equals the eager step (same init, same batch, dropout off)."""
import copy

import pytest
import torch

from tests.helpers import NAMES, golden, product_batch, state_from_spec

pytestmark = pytest.mark.gpu


def _run(use_graph, steps=3, torch_adam=False, direct_grads=True):
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
    eng = TrainStep(model, opt, None, use_graph=use_graph, direct_grads=direct_grads)
    z = golden(f"singa_L{L}_B3.npz")
    batch = product_batch(NAMES, z)
    losses, norms = [], []
    for _ in range(steps):
        losses.append(float(eng.step(batch).detach()))
        norms.append(float(eng.grad_norm))
    eng.norms = norms
    return losses, eng


def test_graph_replay_matches_eager():
    """Step i of the replayed engine = step i of the eager engine, from the very first step: the capture's warm-up
    steps leave parameters, Adam moments and the step count untouched (one update per batch)."""
    eager, e_eng = _run(False, steps=5)
    graph, eng = _run(True, steps=5)
    assert eng.captures == 1
    for a, b in zip(graph, eager):
        assert abs(a - b) < 2e-4 * abs(b), (graph, eager)
    assert eager[4] < eager[0]                              # and the step does train
    assert float(eng.opt.step_t) == 5.0 and float(e_eng.opt.step_t) == 5.0
    # the total gradient norm (clip_grad_norm_'s return value, train.py:126) is finite and the same replayed and eager;
    # first step: the golden's own total norm
    z = golden("singa_L2_B3.npz")
    assert abs(eng.norms[0] - float(z["grad_total"])) < 1e-4 * float(z["grad_total"]), eng.norms
    for a, b in zip(eng.norms, e_eng.norms):
        assert a == a and abs(a - b) < 2e-3 * abs(b), (eng.norms, e_eng.norms)


def test_recapture_does_not_change_the_trajectory():
    """A batch with other shapes triggers a re-capture; the loss sequence still equals the eager engine's."""
    from singa_amd import graph as G
    from singa_amd.config import load_config
    from singa_amd.engine import TrainStep
    from singa_amd.model.GAN import SINGA
    from singa_amd.optim import Adam
    bs = [G.synthetic_batch(2, first_id=50 + 2 * i, n_protein=40 + 6 * i, n_ligand=12, e_pp=200 + 20 * i, e_ll=24, e_x=30).to("cuda")
          for i in range(3)]

    def run(use_graph):
        torch.manual_seed(3)
        model = SINGA(load_config(lmax=2), device="cuda").eval()
        eng = TrainStep(model, Adam(model.parameters(), lr=1e-4), None, use_graph=use_graph)
        return [float(eng.step(bs[i % 3]).detach()) for i in range(6)], eng

    eager, _ = run(False)
    graph, eng = run(True)
    assert eng.captures == 6                                # every step saw a new shape
    assert all(abs(a - b) < 2e-4 * abs(b) for a, b in zip(graph, eager)), (graph, eager)


def test_adam_state_dict_is_torch_adam_layout():
    """Checkpoints interoperate with torch.optim.Adam in both directions (reference train.py:244-252, gen.py:106-110)."""
    from singa_amd.optim import Adam
    torch.manual_seed(0)
    ws = [torch.nn.Parameter(torch.randn(33, 7, device="cuda")), torch.nn.Parameter(torch.randn(5000, device="cuda")),
          torch.nn.Parameter(torch.randn(4, device="cuda"))]                    # the last one never gets a gradient
    wt = [torch.nn.Parameter(w.detach().clone()) for w in ws]
    ours, ref = Adam(ws, lr=1e-3, betas=(0.99, 0.999)), torch.optim.Adam(wt, lr=1e-3, betas=(0.99, 0.999))
    for k in range(3):
        for a, b in zip(ws[:2], wt[:2]):
            g = torch.randn_like(a)
            a.grad, b.grad = g.clone(), g.clone()
        ours.step()
        ref.step()
    sd, sr = ours.state_dict(), ref.state_dict()
    assert set(sd) == set(sr) and set(sd["state"]) == set(sr["state"]) == {0, 1}
    assert sd["param_groups"][0]["params"] == sr["param_groups"][0]["params"] == [0, 1, 2]
    assert set(sr["param_groups"][0]) <= set(sd["param_groups"][0]) | {"decoupled_weight_decay"}
    for i in (0, 1):
        assert float(sd["state"][i]["step"]) == float(sr["state"][i]["step"]) == 3.0
        for k in ("exp_avg", "exp_avg_sq"):
            assert torch.allclose(sd["state"][i][k], sr["state"][i][k], rtol=1e-4, atol=1e-8)
        assert torch.allclose(ws[i], wt[i], rtol=1e-5, atol=1e-7)
    # torch -> ours and ours -> torch, then one more step on each side
    ws2 = [torch.nn.Parameter(w.detach().clone()) for w in wt]
    ours2 = Adam(ws2, lr=5e-4)
    ours2.load_state_dict(sr)
    assert ours2.param_groups[0]["lr"] == 1e-3 and ours2.param_groups[0]["params"][0] is ws2[0]
    ref2 = torch.optim.Adam([torch.nn.Parameter(w.detach().clone()) for w in ws], lr=5e-4)
    ref2.load_state_dict(sd)
    for a, b in zip(ws2[:2], ref2.param_groups[0]["params"][:2]):
        g = torch.randn_like(a)
        a.grad, b.grad = g.clone(), g.clone()
    ours2.step()
    ref2.step()
    for a, b in zip(ws2[:2], ref2.param_groups[0]["params"][:2]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
    with pytest.raises(ValueError):
        ours2.load_state_dict({"built": True, "param_groups": [{}]})


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


def test_direct_parameter_gradients_equal_autograd_accumulation():
    """TrainStep(direct_grads=True) (bias / affine / split-reduction weight gradients added straight into .grad by one
    multi-job reduction) follows the same trajectory as direct_grads=False (every gradient returned to autograd), eager
    and replayed; and the direct path is the one in use."""
    from singa_amd import ops
    base, e0 = _run(False, steps=4, direct_grads=False)
    assert e0._sink_params is None
    for use_graph in (False, True):
        got, e1 = _run(use_graph, steps=4)
        assert len(e1._sink_params) > 150, len(e1._sink_params)
        for a, b in zip(got, base):
            assert abs(a - b) < 2e-4 * abs(b), (got, base)
        for a, b in zip(e1.norms, e0.norms):
            assert abs(a - b) < 2e-3 * abs(b), (e1.norms, e0.norms)
    assert ops._GradSink.on is False and not ops._GradSink.jobs


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


def test_eager_steps_keep_no_gradient_address_tables():
    """in eager mode the gradients are re-created every step (new addresses whenever the allocator says so);
    the optimizer must not keep one address table per distinct address tuple.  Ragged batches + blocks held between the
    steps force new addresses; only captures pin tables."""
    from singa_amd import graph as G
    from singa_amd.config import load_config
    from singa_amd.engine import TrainStep
    from singa_amd.model.GAN import SINGA
    from singa_amd.optim import Adam
    torch.manual_seed(3)
    model = SINGA(load_config(lmax=2), device="cuda").eval()
    opt = Adam(model.parameters(), lr=1e-4)
    eng = TrainStep(model, opt, None, use_graph=False)
    seen, hold = set(), []
    for i in range(6):
        b = G.synthetic_batch(2, first_id=60 + 2 * i, n_protein=40 + 5 * i, n_ligand=12, e_pp=200 + 16 * i, e_ll=24, e_x=30).to("cuda")
        eng.step(b)
        seen.add(opt._grad_ids)
        hold.append(torch.empty(1 << 18, device="cuda"))      # perturb the allocator between the steps
    assert len(opt._pinned) == 0 and opt._scratch_table is not None
    assert len(seen) > 1, "the test did not manage to move the gradients"
    # a graph engine on the same optimizer pins exactly one table per capture and releases it again
    eng2 = TrainStep(model, opt, None, use_graph=True)
    eng2.step(b)
    eng2.step(b)
    assert len(opt._pinned) == 1
    eng2.release()
    assert len(opt._pinned) == 0


def test_load_state_dict_drops_stale_captures():
    """loading an optimizer state into a running engine must not leave captured optimizer graphs that update
    moment buffers the optimizer no longer owns: after load_state_dict the replayed trajectory equals the eager one."""
    from singa_amd.config import load_config
    from singa_amd.engine import TrainStep
    from singa_amd.model.GAN import SINGA
    from singa_amd.optim import Adam
    z = golden("singa_L2_B3.npz")
    batch = product_batch(NAMES, z)

    def run(use_graph):
        model = SINGA(load_config(lmax=2), device="cuda")
        model.load_state_dict(state_from_spec("singa_L2"), strict=False)
        model.eval()
        opt = Adam(model.parameters(), lr=1e-4, betas=(0.99, 0.999))
        eng = TrainStep(model, opt, None, use_graph=use_graph)
        out = [float(eng.step(batch).detach()) for _ in range(2)]
        sd = copy.deepcopy(opt.state_dict())
        params = [p.detach().clone() for p in model.parameters()]
        out += [float(eng.step(batch).detach()) for _ in range(2)]
        # back to the state after step 2: same moments (copied into the existing buffers) ...
        opt.load_state_dict(sd)
        with torch.no_grad():
            for p, q in zip(model.parameters(), params):
                p.copy_(q)
        out += [float(eng.step(batch).detach()) for _ in range(2)]
        # ... and once more through a REBUILT optimizer state (first entry dropped: another active set)
        first = min(sd["state"])
        sd2 = {"state": {k: v for k, v in sd["state"].items() if k != first}, "param_groups": sd["param_groups"]}
        gen = opt.generation
        opt.load_state_dict(sd2)
        assert opt.generation == gen + 1
        with torch.no_grad():
            for p, q in zip(model.parameters(), params):
                p.copy_(q)
        out += [float(eng.step(batch).detach()) for _ in range(2)]
        return out, eng

    eager, _ = run(False)
    graph, eng = run(True)
    assert all(abs(a - b) < 2e-4 * abs(b) for a, b in zip(graph, eager)), (graph, eager)
    assert abs(eager[4] - eager[2]) < 2e-4 * abs(eager[2]) and abs(eager[5] - eager[3]) < 2e-4 * abs(eager[3])
    assert eng.captures == 2                                # the rebuilt optimizer forced exactly one re-capture


def test_eager_backward_before_capture():
    """An eager forward + backward (autograd history on the current stream) followed by a graph capture: the SO(2) block
    weights cached for ONE forward pass must not keep that pass's autograd graph alive (this crashed inside the capture
    when the cache outlived the pass)."""
    from singa_amd.config import load_config
    from singa_amd.engine import TrainStep
    from singa_amd.model.GAN import SINGA
    from singa_amd.model import EF_layers
    from singa_amd.optim import Adam
    z = golden("singa_L2_B3.npz")
    batch = product_batch(NAMES, z)
    model = SINGA(load_config(lmax=2), device="cuda")
    model.load_state_dict(state_from_spec("singa_L2"), strict=False)
    model.eval()
    # (no reference to the logits is kept: a live autograd graph would itself pin the AccumulateGrad nodes)
    torch.nn.functional.cross_entropy(model(batch), batch["ligand_data"]["smiIndices_tgt"].reshape(-1)).backward()
    assert not EF_layers._bw_cache
    model.zero_grad(set_to_none=True)
    eng = TrainStep(model, Adam(model.parameters(), lr=1e-4, betas=(0.99, 0.999)), None, use_graph=True)
    first = float(eng.step(batch).detach())
    assert abs(first - float(z["loss"])) < 1e-4 * float(z["loss"])
