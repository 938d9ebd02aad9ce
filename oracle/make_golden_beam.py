"""Generate tests/golden/beam_*.npz by running the REFERENCE's own beam search (model/BeamSearch.py:38-175, imported
unmodified from /root/reference under oracle/shims) on the bundled example proteins, the way gen.py:156-196 drives it:
protein pass of the embedding (`gen_mode=True`), encoder 1 only, `num_beams` beams, property prompt, eval mode.

    python oracle/make_golden_beam.py          # writes tests/golden/beam_*.npz

TEST INFRASTRUCTURE.  Only tensors are written: the protein inputs of the search (embedded atom
features, positions, Laplacian PE, the kNN graph the encoder drew, the edge frames of the embedding pass), the decoded
token matrix, the per-hypothesis scores, and the first-step log-probabilities.  Weights are the synthetic ones of
oracle/weights.py (a function of the parameter names), so they are not stored.

Two documented rescalings of the vocabulary projection are applied before a search (the product test applies the same
factors to the same rows): `proj_gain` multiplies the whole matrix and `eos_gain` the '$' row.  With untrained weights
one token dominates every step and the end-of-sequence token never reaches the top of a beam, so the beam reordering
and the hypothesis bookkeeping (BeamSearch.py:107-123) would otherwise stay untested.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (sets sys.path for the shims and the reference)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from easydict import EasyDict  # noqa: E402

PA = MG.PA
CASES = {
    # tag: (graph names, num_beams, max_length, topk, proj_gain, eos_gain)
    "b1_k20": (["4agq_5a7b"], 20, 40, 1, 1.0, 1.0),                    # gen.py's 20 beams, raw synthetic weights
    "b2_k4": (["3wi2_4tpp", "5cp5_4nue"], 4, 24, 2, 1.0, 1.0),
    "b2_k6_eos": (["4agq_5a7b", "3wi2_4tpp"], 6, 32, 3, 1.0, None),    # eos_gain searched below so that '$' wins sometimes
    "b2_k5_flat": (["5cp5_4nue", "4agq_5a7b"], 5, 28, 2, 0.25, None),  # flatter distribution
}


def protein_inputs(model, names, rec, lap_of):
    feats, poss, laps, batches, rots = [], [], [], [], []
    for b, n in enumerate(names):
        g = MG.load_graph(n)
        rec.clear()
        torch.manual_seed(2022)
        with torch.no_grad():
            emb = model.embedding(g, gen_mode=True)
        rots.append(rec.rot[0].numpy())
        f = emb[PA].embedding.reshape(g[PA]["x"].shape[0], -1)
        feats.append(f)
        poss.append(g[PA]["pos"])
        laps.append(lap_of[n])
        batches.append(torch.full((f.shape[0],), b, dtype=torch.long))
    return torch.cat(feats), torch.cat(poss), torch.cat(laps), torch.cat(batches), rots


def main():
    from model.BeamSearch import beam_search
    import model.BeamSearch as BS
    from model.GAN import SINGA
    cfg = MG.config_for(2)
    torch.manual_seed(2022)
    model = SINGA(cfg, device="cpu")
    MG.overwrite_params(model, "singa_L2")
    model.eval()
    rec = MG.Recorder().install()
    voc = list(cfg.model.decoder.smiVoc)
    eos = voc.index("$")
    # Laplacian PEs: the ones recorded for the 3-graph SINGA golden (deterministic inputs of this test)
    sg = np.load(os.path.join(MG.OUT, "singa_L2_B3.npz"))
    n_at = [MG.load_graph(n)[PA]["x"].shape[0] for n in MG.NAMES]
    off = np.concatenate([[0], np.cumsum(n_at)])
    lap_of = {n: torch.from_numpy(sg["lap_p"][off[i]:off[i + 1]]) for i, n in enumerate(MG.NAMES)}
    w0 = model.model.projection.weight.detach().clone()

    for tag, (names, nb, max_len, topk, pgain, gain) in CASES.items():
        feat, pos, lap, batch, rots = protein_inputs(model, names, rec, lap_of)
        B = len(names)
        prop = torch.tensor([[1.0, 1.0, 1.0]] * (B * nb))            # gen.py:169 with generate.prop = [1, 1, 1]
        ex = EasyDict()
        ex.protein_element_batch, ex.protein_atom_feature, ex.protein_pos, ex.protein_atom_laplacian = batch, feat, pos, lap

        def search(g):
            with torch.no_grad():
                model.model.projection.weight.copy_(w0 * pgain)
                model.model.projection.weight[eos] *= g
            rec.clear()
            first, hyps, states = [], [], []
            o_ls, o_bh, o_cat = BS.F.log_softmax, BS.BeamHypotheses, BS.torch.cat

            def cat(ts, dim=0):
                r = o_cat(ts, dim=dim)
                if r.dtype == torch.long:                 # BeamSearch.py:138 - the beam state after each step
                    states.append(r.clone())
                return r

            def ls(x, dim):
                r = o_ls(x, dim=dim)
                if not first:
                    first.append(r.clone())
                return r

            class Hyp(o_bh):
                def __init__(self, *a, **k):
                    super().__init__(*a, **k)
                    hyps.append(self)

            BS.F.log_softmax, BS.BeamHypotheses = ls, Hyp
            BS.torch = type("T", (), {"__getattr__": lambda self, k: cat if k == "cat" else getattr(torch, k)})()
            try:
                with torch.no_grad():
                    out = beam_search(model, voc, nb, B, max_len, topk, ex, prop, device="cpu")
            finally:
                BS.F.log_softmax, BS.BeamHypotheses, BS.torch = o_ls, o_bh, torch
            reorders = sum(not torch.equal(b[:, :-1], a) for a, b in zip(states[:-1], states[1:]))
            return out, first[0], hyps, states[-1] if states else torch.zeros(0, 0, dtype=torch.long), reorders

        if gain is None:
            # first gain of the ladder +-2^k for which some, but not all, hypotheses end with '$' before max_length
            for gain in [s * 1.25 ** k for k in range(0, 16) for s in (1.0, -1.0)]:
                out, first, hyps, last, reorders = search(gain)
                n_short = sum(len(h) < max_len - 1 for hy in hyps for _, h in hy.beams)
                lens = sorted(len(h) for hy in hyps for _, h in hy.beams)
                if n_short >= 2 and len(set(lens)) >= 4:
                    break
            else:
                raise SystemExit(f"{tag}: no eos_gain on the ladder gives hypotheses of four different lengths")
        else:
            out, first, hyps, last, reorders = search(gain)
        d = {"names": np.array(names), "num_beams": np.array(nb), "max_length": np.array(max_len), "topk": np.array(topk),
             "eos_gain": np.array(gain), "proj_gain": np.array(pgain), "feat": feat.numpy(), "pos": pos.numpy(), "lap": lap.numpy(), "batch": batch.numpy(),
             "knn": rec.knn[0].numpy(), "prop": prop.numpy(), "decoded": out.numpy(), "first_logp": first.numpy(),
             "last_beams": last.numpy(), "reorder_steps": np.array(reorders)}
        for b, r in enumerate(rots):
            d[f"rot_pp_{b}"] = r
        sc = [sorted(s for s, _ in hy.beams) for hy in hyps]
        d["hyp_scores"] = np.array([s + [np.nan] * (nb - len(s)) for s in sc])
        d["hyp_lens"] = np.array([sorted(len(h) for _, h in hy.beams) + [-1] * (nb - len(hy.beams)) for hy in hyps])
        np.savez_compressed(os.path.join(MG.OUT, f"beam_{tag}.npz"), **d)
        txt = ["".join(voc[t] for t in row) for row in out.tolist()]
        print(f"beam {tag}: gain {gain:.3f} reordering steps {reorders} last beam state {tuple(last.shape)} decoded {tuple(out.shape)} lens {d['hyp_lens'].tolist()}\n   " + "\n   ".join(txt))


if __name__ == "__main__":
    main()
