"""-m gpu: KV-cached beam search (singa_amd/model/BeamSearch.py) against the REFERENCE's own beam search (tests/golden/
beam_*.npz, oracle/make_golden_beam.py): decoded tokens, the complete final beam state, every stored hypothesis
(score and length) and the first-step log-probabilities; and against the CPU oracle on a synthetic protein."""
import numpy as np
import pytest
import torch

from tests.helpers import BEAM_CASES, apply_beam_gains, golden, product_batch, rel_err, smi_voc, state_from_spec

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build_model(z=None):
    from singa_amd.config import Config, load_config
    from singa_amd.model.GAN import SINGA
    model = SINGA(load_config(lmax=2), device=DEV)
    sd = state_from_spec("singa_L2")
    if z is not None:
        apply_beam_gains(sd["model.projection.weight"], z)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected
    model.eval()
    return model, sd, Config


@pytest.mark.parametrize("graph,fused", [(True, True), (False, True), (True, False), (False, False)],
                         ids=["hipgraph-k17", "eager-k17", "hipgraph-library", "eager-library"])
@pytest.mark.parametrize("case", BEAM_CASES)
def test_beam_search_matches_reference(case, graph, fused):
    from singa_amd.model.BeamSearch import beam_search
    z = golden(f"beam_{case}.npz")
    model, _, Config = build_model(z)
    t = lambda k, dt=torch.float32: torch.as_tensor(z[k]).to(dt).to(DEV)
    ex = Config()
    ex.protein_element_batch, ex.protein_atom_feature, ex.protein_pos = t("batch", torch.long), t("feat"), t("pos")
    ex.protein_atom_laplacian, ex.protein_knn = t("lap"), t("knn", torch.long)
    tr = {}
    out = beam_search(model, smi_voc(), int(z["num_beams"]), len(z["names"]), int(z["max_length"]), int(z["topk"]), ex,
                      t("prop"), device=DEV, trace=tr, graph=graph, fused=fused)
    assert rel_err(tr["first_logp"].cpu(), z["first_logp"]) < 1e-4
    assert np.array_equal(tr["last_beams"], z["last_beams"])
    assert out.shape == z["decoded"].shape and np.array_equal(out.cpu().numpy(), z["decoded"])
    for b, h in enumerate(tr["hyps"]):
        n = int((z["hyp_lens"][b] >= 0).sum())
        assert len(h) == n
        assert np.allclose(sorted(s for s, _ in h.beams), z["hyp_scores"][b][:n], rtol=1e-4, atol=1e-5)
        assert sorted(len(x) for _, x in h.beams) == [int(v) for v in z["hyp_lens"][b][:n]]


def test_gen_mode_embedding_matches_reference():
    """`embedding(g, gen_mode=True)` (gen.py:157-160): the protein pass alone gives the features the search starts from."""
    from singa_amd.graph import PA
    z = golden("beam_b1_k20.npz")
    model, _, _ = build_model()
    g = product_batch([str(z["names"][0])], None)
    g.extras["edge_rot_mat"] = {"pp": torch.tensor(z["rot_pp_0"]).to(DEV)}
    with torch.no_grad():
        out = model.embedding(g, gen_mode=True)
    assert list(out.keys()) == [PA]
    assert rel_err(out[PA].embedding.reshape(z["feat"].shape).cpu(), z["feat"]) < 1e-4


def test_beam_search_matches_oracle_on_synthetic_protein():
    """End to end on a synthetic pocket (own kNN graph, GPU Laplacian PE, gen_mode embedding on the GPU), against the CPU
    restatement fed with the same embedded features: same tokens for every returned hypothesis."""
    from oracle import beam_oracle as BO
    from singa_amd import graph as G
    from singa_amd.model.BeamSearch import beam_search
    model, sd, Config = build_model()
    nb, max_len, topk = 5, 12, 2
    gs = [G.synthetic_graph(7 + i, n_protein=60 + 9 * i, n_ligand=12) for i in range(2)]
    b = G.collate(gs).to(DEV)
    model.prepare(b)
    with torch.no_grad():
        feat = model.embedding(b, gen_mode=True)[G.PA].embedding.reshape(b[G.PA]["x"].shape[0], -1)
    from singa_amd.model.CProMG import DenseMap, knn_graph
    batch = b[G.PA]["batch"]
    knn = knn_graph(b[G.PA]["pos"], model.config.model.encoder.knn, batch, 2, DenseMap(batch, 2))
    knn = knn[:, knn[0] >= 0]
    ex = Config()
    ex.protein_element_batch, ex.protein_atom_feature, ex.protein_pos = batch, feat, b[G.PA]["pos"]
    ex.protein_atom_laplacian, ex.protein_knn = b[G.PA]["lap_pe"], knn
    prop = torch.tensor([[1.0, 0.0, 1.0]] * (2 * nb), device=DEV)
    tr, tr_o = {}, {}
    out = beam_search(model, smi_voc(), nb, 2, max_len, topk, ex, prop, device=DEV, trace=tr)
    c = lambda x: x.detach().cpu()
    with torch.no_grad():
        want = BO.beam_search(sd, smi_voc(), nb, 2, max_len, topk, c(feat), c(b[G.PA]["pos"]), c(batch), c(b[G.PA]["lap_pe"]),
                              c(knn), c(prop), trace=tr_o)
    assert rel_err(c(tr["first_logp"]), tr_o["first_logp"]) < 1e-4
    assert np.array_equal(c(out).numpy(), want.numpy())
    assert np.array_equal(tr["last_beams"], tr_o["last_beams"].numpy())


def test_step_kernels_match_library_decoder():
    """k17 (three launches per decoder layer) against the same layers evaluated with library GEMMs, position by position:
    decoder outputs and the appended key / value cache rows."""
    from singa_amd.model.BeamSearch import KVDecoder
    model, _, _ = build_model()
    tf = model.model
    torch.manual_seed(4)
    B, beams, S, P = 2, 3, 37, 12
    enc = torch.randn(B, S, 256, device=DEV)
    pad = torch.zeros(B, 1, S, dtype=torch.bool, device=DEV)
    pad[0, 0, 30:] = True                                   # ragged proteins: the first one has 30 atoms
    with torch.no_grad():
        kv = [KVDecoder(tf.decoder, tf.projection, enc, pad, beams, P, 116, fused=f) for f in (True, False)]
        assert kv[0].fused and not kv[1].fused
        for step in range(P):
            x = torch.randn(B * beams, 256, device=DEV)
            a, b = kv[0].advance(x), kv[1].advance(x)
            assert rel_err(a.cpu(), b.cpu()) < 2e-5, step
            if step % 4 == 3:                               # re-rank the rows as a search step would
                src = torch.randperm(B * beams, device=DEV) % beams + (torch.arange(B * beams, device=DEV) // beams) * beams
                kv[0].follow(src), kv[1].follow(src)
        assert rel_err(kv[0].k[:, :, :, :P].cpu(), kv[1].k[:, :, :, :P].cpu()) < 2e-5
        assert rel_err(kv[0].v[:, :, :, :P].cpu(), kv[1].v[:, :, :, :P].cpu()) < 2e-5
