"""Pin the CPU oracle against tensors produced by the REFERENCE itself (oracle/make_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import singa_oracle as O
from tests.helpers import (BEAM_CASES, NAMES, apply_beam_gains, golden, grad_sample_errors, oracle_pinned_relu_ties, rel_err, smi_voc,
                           state_from_spec)

TOL = 2e-5  # fp32 op-order noise between the reference's einsum/bmm chains and the restatement


def test_wigner_matches_reference():
    z = golden("wigner_L6.npz")
    w = O.wigner_dense(torch.as_tensor(z["rot"]), 6)
    assert rel_err(w, z["wigner"]) < 1e-6


@pytest.mark.parametrize("L", [2, 4, 6])
@pytest.mark.parametrize("name", NAMES)
def test_embedding_forward_and_grads(L, name):
    sd = {k: v.requires_grad_(True) for k, v in state_from_spec(f"embed_L{L}").items()}
    z = golden(f"embed_L{L}_{name}.npz")
    g = O.load_graph_npz(f"tests/golden/graph_{name}.npz")
    rots = {k: torch.as_tensor(z[f"rot_{k}"]) for k in ("pp", "ll", "lp")}
    out = O.embedding_forward(sd, g, rots, L)
    st = int(z["node_stride"])
    assert rel_err(out[O.PA][::st], z["out_p"]) < TOL
    assert rel_err(out[O.LA], z["out_l"]) < TOL
    assert rel_err(out["lp_edge"][::st], z["out_lp"]) < TOL
    assert rel_err(out["pl_edge"], z["out_pl"]) < TOL
    loss = (out[O.PA] ** 2).sum() + (out[O.LA] ** 2).sum()
    assert abs(float(loss) - float(z["loss"])) / float(z["loss"]) < TOL
    loss.backward()
    for n, ref in zip(z["grad_names"], z["grad_norms"]):
        gr = sd[str(n)].grad
        if ref < 0:                      # Q1/Q10: parameters the reference never reaches
            assert gr is None or float(gr.norm()) == 0.0, n
        else:
            assert abs(float(gr.norm()) - ref) <= 1e-3 * ref + 1e-6, (n, float(gr.norm()), ref)
    for k in z.files:
        if k.startswith("grad:"):
            assert rel_err(sd[k[5:]].grad, z[k]) < 1e-3, k


@pytest.mark.parametrize("L", [2, 6])
def test_block0_intermediates(L):
    """Edge-degree embedding, norm_1, radial MLP, attention, FFN and block output of blocks[0] (protein pass)."""
    name = NAMES[1]
    sd = state_from_spec(f"embed_L{L}")
    z = golden(f"embed_L{L}_{name}.npz")
    g = O.load_graph_npz(f"tests/golden/graph_{name}.npz")
    st = int(z["node_stride"])
    hp = O.hyper(sd, "", L)
    fr = O.Frame(torch.as_tensor(z["rot_pp"]), L, 2)
    ei, pos, zt = g["ei_pp"], g["pos_p"], g["z_p"]
    d = (pos[ei[0]] - pos[ei[1]]).norm(dim=-1)
    xe = torch.cat([O.gaussian(d, 10.0, 16, 20.0), sd["source_embedding.weight"][zt[ei[0]]],
                    sd["target_embedding.weight"][zt[ei[1]]]], 1)
    ed = O.edge_degree(sd, "edge_degree_embedding", xe, ei[1], zt.shape[0], fr, 16)
    assert rel_err(ed[::st], z["edge_degree_pp"]) < TOL
    x = torch.zeros(zt.shape[0], (L + 1) ** 2, 16)
    x[:, 0] = (sd["sphere_embedding.weight"][zt] + sd["sphere_embedding_2.weight"][O.barcode(g["x_p"])]).long().float()
    x = x + ed
    xn = O.rms_norm(sd, "blocks.0.norm_1", x, L)
    assert rel_err(xn[::st], z["b0_norm1_pp"]) < TOL
    assert rel_err(O.radial(sd, "blocks.0.ga.so2_conv_1.rad_func", xe)[::st], z["b0_rad_pp"]) < TOL
    ga = O.graph_attention(sd, "blocks.0.ga", xn, xn, xe, ei[0], ei[1], fr, hp)
    assert rel_err(ga[::st], z["b0_ga_pp"]) < TOL
    y = ga + x
    f = O.ffn(sd, "blocks.0.ffn", O.rms_norm(sd, "blocks.0.norm_2", y, L), L)
    assert rel_err(f[::st], z["b0_ffn_pp"]) < TOL
    assert rel_err((f + y)[::st], z["b0_out_pp"]) < TOL


@pytest.mark.parametrize("L", [2, 4, 6])
def test_singa_step(L):
    """Full SINGA forward + CE + backward on the 3-graph batch vs the reference (GAN:25-81, train.py:119-124)."""
    sd = {k: v.requires_grad_(True) for k, v in state_from_spec(f"singa_L{L}").items()}
    z = golden(f"singa_L{L}_B3.npz")
    g = O.collate([O.load_graph_npz(f"tests/golden/graph_{n}.npz") for n in NAMES])
    rots = {k: torch.as_tensor(z[f"rot_{k}"]) for k in ("pp", "ll", "lp")}
    with oracle_pinned_relu_ties(L) as pins:      # fp32-tied ReLU gates follow the reference's run
        logits = O.singa_forward(sd, g, rots, L, torch.as_tensor(z["knn_p"]), torch.as_tensor(z["knn_l"]),
                                 torch.as_tensor(z["lap_p"]), torch.as_tensor(z["lap_l"]))
        assert rel_err(logits, z["logits"]) < 1e-4
        loss = torch.nn.functional.cross_entropy(logits, g["tok_tgt"].reshape(-1))
        assert abs(float(loss) - float(z["loss"])) < 1e-4 * float(z["loss"])
        loss.backward()
    assert pins.call == 18 and pins.flipped <= 16, (pins.call, pins.flipped)
    tot = float(torch.sqrt(sum((v.grad.double() ** 2).sum() for v in sd.values() if v.grad is not None)))
    assert abs(tot - float(z["grad_total"])) < 1e-4 * float(z["grad_total"])
    bad = []
    for n, ref in zip(z["grad_names"], z["grad_norms"]):
        gr = sd[str(n)].grad
        if ref < 0:
            if gr is not None and float(gr.norm()) != 0.0:
                bad.append((str(n), "unexpected grad"))
        elif abs(float(gr.norm()) - ref) > 3e-4 * ref + 1e-7:      # the golden's float32 norms are up to 1.4e-4 low on the largest tensors
            bad.append((str(n), float(gr.norm()), float(ref)))
    assert not bad, bad[:10]
    # element-wise: up to 512 gradient elements of EVERY parameter (a norm cannot see a permuted or sign-flipped block)
    errs = sorted(grad_sample_errors({k: v.grad for k, v in sd.items()}, z, 0.0), key=lambda e: -e[1])
    print(f"L={L}: {pins.flipped} gates pinned; worst element-wise gradient sample errors: " + ", ".join(f"{n} {e:.1e}" for n, e in errs[:4]))
    bad = [e for e in errs if e[1] > 1e-5]          # measured with the ties pinned: 1e-7 (3e-3 was needed before)
    assert not bad, bad[:10]


@pytest.mark.parametrize("L", [2, 4, 6])
def test_relu_tie_fixture_is_consistent(L):
    """The recorded near-zero ReLU gates of the golden step (oracle/make_relu_ties.py): 18 PoswiseFeedForward calls with
    the golden batch's token rows, indices in range, a few hundred entries, about half of them open."""
    t = golden(f"singa_L{L}_B3_relu_ties.npz")
    g = O.collate([O.load_graph_npz(f"tests/golden/graph_{n}.npz") for n in NAMES])
    rows = [int(r) for r in t["rows"]]
    n_p, n_l, B, T = g["x_p"].shape[0], g["x_l"].shape[0], *g["tok_tgt"].shape
    # decoder rows: the property token in front of the tgt_len tokens of each graph (CP:371-420)
    assert len(rows) == 18 and rows[:6] == [n_p] * 6 and rows[6:12] == [n_l] * 6 and rows[12:] == [B * (T + 1)] * 6
    assert len(t["layer"]) == len(t["row"]) == len(t["unit"]) == len(t["on"])
    assert 100 < len(t["layer"]) < 5000 and 0.3 < t["on"].mean() < 0.7
    assert int(t["layer"].min()) >= 0 and int(t["layer"].max()) <= 17 and int(t["unit"].max()) < 1024
    assert all(int(r) < rows[int(l)] for l, r in zip(t["layer"], t["row"]))


@pytest.mark.parametrize("case", BEAM_CASES)
def test_beam_search(case):
    """Restated beam search vs the reference's own (BeamSearch.py:38-175) on reference-embedded proteins: decoded
    tokens, the complete final beam state, every stored hypothesis score, and the first-step log-probabilities."""
    from oracle import beam_oracle as BO
    z = golden(f"beam_{case}.npz")
    sd = state_from_spec("singa_L2")
    apply_beam_gains(sd["model.projection.weight"], z)
    tr = {}
    t = lambda k, dt=torch.float32: torch.as_tensor(z[k]).to(dt)
    B = len(z["names"])
    with torch.no_grad():
        out = BO.beam_search(sd, smi_voc(), int(z["num_beams"]), B, int(z["max_length"]), int(z["topk"]), t("feat"), t("pos"),
                             t("batch", torch.long), t("lap"), t("knn", torch.long), t("prop"), trace=tr)
    assert out.shape == z["decoded"].shape and np.array_equal(out.numpy(), z["decoded"])
    assert np.array_equal(tr["last_beams"].numpy(), z["last_beams"])
    assert rel_err(tr["first_logp"], z["first_logp"]) < TOL
    for b, h in enumerate(tr["hyps"]):
        got = np.array(sorted(s for s, _ in h.items))
        want = z["hyp_scores"][b][: len(got)]
        assert len(got) == int((z["hyp_lens"][b] >= 0).sum())
        assert np.allclose(got, want, rtol=1e-5, atol=1e-6)
        assert sorted(len(x) for _, x in h.items) == [int(v) for v in z["hyp_lens"][b] if v >= 0]
