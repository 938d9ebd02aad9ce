"""Generate tests/golden/*.npz by running the REFERENCE's own Python modules (imported unmodified from
/root/reference under oracle/shims) on the bundled example graphs.  Run once, in the build container:

    python oracle/make_golden.py            # writes tests/golden/

TEST INFRASTRUCTURE. The reference sources never leave /root/reference; only tensors (inputs and expected
outputs) are written. What is recorded, per SURVEY.md §8c:
  graph_<name>.npz        inputs extracted from example/<name>.pt (reference data files)
  embed_L<L>_<name>.npz   EquivariantEmbedding forward (reference model/Embedding.py:205-480): the three
                          edge_rot_mat draws (Q6), edge-degree output, block-0 intermediates of the protein pass,
                          final embeddings, and parameter-gradient norms of sum-of-squares loss
  rot_rand_<name>.npz     the three torch.rand_like draws the reference's init_edge_rot_mat took (EF:2301) for the
                          frames recorded in embed_L2_<name>.npz (rot_pp / rot_ll / rot_lp)
  wigner_L6.npz           RotationToWignerDMatrix output for 16 edges (reference model/EF_layers.py:508-528)
  singa_L<L>_B3.npz       full SINGA forward + CrossEntropy + backward (reference model/GAN.py:25-81,
                          train.py:119-124) on the 3-graph batch for L = 2, 4, 6, eval-mode dropout, with the kNN
                          graphs and Laplacian PEs that were used recorded as inputs; logits, loss, per-parameter
                          gradient norms and up to 512 gradient elements of every parameter
  param_spec_L<L>.npz     (name, shape, mean, std) of every reference parameter -> oracle/weights.py
"""
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "shims"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, HERE)
warnings.filterwarnings("ignore")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402
from easydict import EasyDict  # noqa: E402
from torch_geometric.data import Batch  # noqa: E402

import weights as W  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
NAMES = ["3wi2_4tpp", "4agq_5a7b", "5cp5_4nue"]
PA, LA = "protein_atoms", "ligand_atoms"
ETYPES = {"pp": (PA, "linked_to", PA), "ll": (LA, "linked_to", LA),
          "lp": (LA, "interact_with", PA), "pl": (PA, "interact_with", LA)}


def load_graph(name):
    g = torch.load(f"/root/reference/example/{name}.pt", map_location="cpu", weights_only=False)
    ld = g["ligand_data"]
    if "vina_score" not in ld:  # older featuriser kept it in y[0] (reference utils/Featuriser.py:155 vs :164)
        ld["vina_score"] = float(g["y"][0])
    # keep only what the hot path reads, so that collate stays simple
    keep = ["sas", "logP", "qed", "weight", "tpsa", "vina_score", "smiIndices_input", "smiIndices_tgt"]
    g._global_store._mapping["ligand_data"] = {k: ld[k] for k in keep}
    g._global_store._mapping.pop("y", None)
    return g


def dump_graph(name, g):
    d = {}
    for nt, key in ((PA, "p"), (LA, "l")):
        d[f"x_{key}"] = g[nt]["x"].numpy()
        d[f"pos_{key}"] = g[nt]["pos"].numpy()
        d[f"z_{key}"] = g["atomicnum"][nt].numpy()
    for k, et in ETYPES.items():
        d[f"ei_{k}"] = g[et]["edge_index"].numpy()
    ld = g["ligand_data"]
    d["props"] = np.array([ld["vina_score"], ld["qed"], ld["sas"]], dtype=np.float64)
    d["tok_in"] = ld["smiIndices_input"].numpy()
    d["tok_tgt"] = ld["smiIndices_tgt"].numpy()
    np.savez_compressed(os.path.join(OUT, f"graph_{name}.npz"), **d)


def config_for(L):
    cfg = EasyDict(yaml.safe_load(open("/root/reference/config/train.yml")))
    cfg.embedding.lmax_list = [L]
    cfg.embedding.mmax_list = [2]
    cfg.model.featurizer_feat_dim = (L + 1) ** 2 * cfg.embedding.sphere_channels
    return cfg


def overwrite_params(module, tag):
    spec = W.spec_from_module(module)
    st = W.synth_state(spec)
    with torch.no_grad():
        for n, p in module.named_parameters():
            p.copy_(st[n])
    np.savez_compressed(
        os.path.join(OUT, f"param_spec_{tag}.npz"),
        names=np.array([s[0] for s in spec]),
        shapes=np.array([",".join(map(str, s[1])) for s in spec]),
        mean=np.array([s[2] for s in spec]), std=np.array([s[3] for s in spec]))


class Recorder:
    """Wraps module-level functions of the imported reference to record their outputs."""

    def __init__(self):
        self.rot, self.knn, self.lap = [], [], []

    def install(self):
        import model.CProMG as CP
        import model.Embedding as EMB
        import model.GAN as GAN
        self._orig = (EMB.init_edge_rot_mat, CP.knn_graph, GAN.lap_pe)
        o_rot, o_knn, o_lap = self._orig

        def rot(*a, **k):
            r = o_rot(*a, **k)
            self.rot.append(r.clone())
            return r

        def knn(*a, **k):
            r = o_knn(*a, **k)
            self.knn.append(r.clone())
            return r

        def lap(*a, **k):
            r = o_lap(*a, **k)
            self.lap.append(r.clone())
            return r

        EMB.init_edge_rot_mat, CP.knn_graph, GAN.lap_pe = rot, knn, lap
        return self

    def clear(self):
        self.rot, self.knn, self.lap = [], [], []


def run_embedding(L, rec):
    from model.Embedding import EquivariantEmbedding
    cfg = config_for(L)
    torch.manual_seed(2022)
    emb = EquivariantEmbedding(cfg.embedding, device="cpu")
    overwrite_params(emb, f"embed_L{L}")
    for name in NAMES:
        g = load_graph(name)
        rec.clear()
        torch.manual_seed(2022)
        inter = {}
        hooks = []

        def save(key):
            def fn(mod, inp, out):
                if key in inter:      # first call only = protein pass
                    return
                t = out.embedding if hasattr(out, "embedding") else out
                inter[key] = t.detach().clone().numpy()
            return fn

        b0 = emb.blocks[0]
        hooks.append(emb.edge_degree_embedding.register_forward_hook(save("edge_degree_pp")))
        hooks.append(b0.norm_1.register_forward_hook(save("b0_norm1_pp")))
        hooks.append(b0.ga.so2_conv_1.rad_func.register_forward_hook(save("b0_rad_pp")))
        hooks.append(b0.ga.register_forward_hook(save("b0_ga_pp")))
        hooks.append(b0.ffn.register_forward_hook(save("b0_ffn_pp")))
        hooks.append(b0.register_forward_hook(save("b0_out_pp")))
        emb.zero_grad()
        draws = []
        orig_rand_like = torch.rand_like

        def rec_rand_like(*a, **k):       # the reference's own draws inside init_edge_rot_mat (EF:2301, Q6)
            r = orig_rand_like(*a, **k)
            draws.append(r.clone())
            return r
        torch.rand_like = rec_rand_like
        try:
            out = emb(g)
        finally:
            torch.rand_like = orig_rand_like
        for h in hooks:
            h.remove()
        assert len(draws) == 3
        if L == 2:      # inputs of the edge-frame pin (tests/test_oracle_conventions.py): draw -> frame, per pass
            np.savez_compressed(os.path.join(OUT, f"rot_rand_{name}.npz"),
                                **{k: draws[i].numpy() for i, k in enumerate(["pp", "ll", "lp"])})
        loss = (out[PA].embedding ** 2).sum() + (out[LA].embedding ** 2).sum()
        loss.backward()
        gn = {n: (float(p.grad.norm()) if p.grad is not None else -1.0) for n, p in emb.named_parameters()}
        d = {f"rot_{k}": rec.rot[i].numpy() for i, k in enumerate(["pp", "ll", "lp"])}
        st = {2: 1, 4: 2, 6: 4}[L]
        d["node_stride"] = np.array(st)
        if name == NAMES[1]:          # block-0 intermediates for the smallest graph only (file size)
            for k, v in inter.items():
                d[k] = v[::st]
        d["out_p"] = out[PA].embedding.detach().numpy()[::st]
        d["out_l"] = out[LA].embedding.detach().numpy()
        d["out_lp"] = out["lp_edge"].embedding.detach().numpy()[::st]
        d["out_pl"] = out["pl_edge"].embedding.detach().numpy()
        d["loss"] = np.array(float(loss))
        d["grad_names"] = np.array(list(gn.keys()))
        d["grad_norms"] = np.array(list(gn.values()))
        for n in ["blocks.0.ga.alpha_dot", "blocks.2.norm_1.affine_weight",
                  "edge_degree_embedding.rad_func.net.0.weight", "blocks.1.ga.proj.bias"]:
            d["grad:" + n] = dict(emb.named_parameters())[n].grad.numpy()
        np.savez_compressed(os.path.join(OUT, f"embed_L{L}_{name}.npz"), **d)
        print(f"embed L={L} {name}: loss {float(loss):.6e}")
        if L == 6 and name == NAMES[1]:
            wig = emb.SO3_rotation[0].RotationToWignerDMatrix(rec.rot[0][:16], 0, 6)
            np.savez_compressed(os.path.join(OUT, "wigner_L6.npz"), rot=rec.rot[0][:16].numpy(), wigner=wig.numpy())


def run_singa(L, rec):
    from model.GAN import SINGA
    cfg = config_for(L)
    torch.manual_seed(2022)
    model = SINGA(cfg, device="cpu")
    overwrite_params(model, f"singa_L{L}")
    model.eval()  # dropout off (PositionalEncoding p=0.1, Q9); BatchNorm layers are never called (Q10)
    batch = Batch.from_data_list([load_graph(n) for n in NAMES])
    rec.clear()
    torch.manual_seed(2022)
    logits = model(batch)
    tgt = batch["ligand_data"]["smiIndices_tgt"].contiguous().view(-1)
    loss = torch.nn.CrossEntropyLoss()(logits, tgt)
    loss.backward()
    gn = {n: (float(p.grad.norm()) if p.grad is not None else -1.0) for n, p in model.named_parameters()}
    tot = float(torch.sqrt(sum(p.grad.norm() ** 2 for p in model.parameters() if p.grad is not None)))
    d = {f"rot_{k}": rec.rot[i].numpy() for i, k in enumerate(["pp", "ll", "lp"])}
    d["knn_p"], d["knn_l"] = rec.knn[0].numpy(), rec.knn[1].numpy()
    d["lap_p"], d["lap_l"] = rec.lap[0].numpy(), rec.lap[1].numpy()
    d["logits"] = logits.detach().numpy()
    d["loss"] = np.array(float(loss))
    d["grad_total"] = np.array(tot)
    d["grad_names"] = np.array(list(gn.keys()))
    d["grad_norms"] = np.array(list(gn.values()))
    # element-wise gradient samples of EVERY parameter (a norm cannot see a permuted or sign-flipped block): up to 512
    # evenly spaced elements of each flattened gradient (W.sample_index), concatenated in parameter order
    samples = [p.grad.reshape(-1)[torch.as_tensor(W.sample_index(p.numel()))].numpy()
               for _, p in model.named_parameters() if p.grad is not None]
    d["grad_samples"] = np.concatenate(samples).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, f"singa_L{L}_B3.npz"), **d)
    print(f"singa L={L}: loss {float(loss):.6f} grad {tot:.4f}")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for n in NAMES:
        dump_graph(n, load_graph(n))
    rec = Recorder().install()
    for L in (2, 4, 6):
        run_embedding(L, rec)
    for L in (2, 4, 6):
        run_singa(L, rec)
