"""SINGA generator assembly for MI355X (reference model/GAN.py:12-81): EquivariantEmbedding + CProMG Transformer.

`SINGA(config, device)(g)` returns logits [B*tgt_len, vocab] exactly as the reference; `g` is a
singa_amd.graph.HeteroGraph batch with the reference's HeteroData field names.  The per-graph Python loops of the
reference (ptr -> batch vectors, GAN:49-55; zip of property lists, GAN:42) are tensor ops; the Laplacian positional
encodings (dgl.lap_pe inside the reference forward, GAN:71,77) are inputs carried by the batch (`lap_pe` on each
node store; SURVEY.md §8c: unpinnable in the reference, computed deterministically per graph by the data pipeline).
"""
import torch
import torch.nn as nn

from ..graph import LA, PA
from .CProMG import Transformer
from .Embedding import EquivariantEmbedding


def lap_pe(data, node_type: str):
    assert node_type in (PA, LA), "Node type not accepted"
    return data[node_type]["lap_pe"]


class SINGA(nn.Module):
    def __init__(self, config, device="cuda"):
        super().__init__()
        self.device = device
        self.config = config
        self.embedding = EquivariantEmbedding(config=self.config.embedding, device=self.device)
        self.model = Transformer(config=self.config.model, protein_atom_feature_dim=self.config.model.featurizer_feat_dim,
                                 num_props=self.config.train.num_props, device=self.device)

    def backward_phases(self):
        """Parameter groups in the order their gradients become final in a backward pass: the transformer (it consumes the
        embedding's output), then the embedding.  The data-parallel step reduces the first group while the second is
        still being computed (engine.TrainStep.two_phase)."""
        return [list(self.model.parameters()), list(self.embedding.parameters())]

    def forward(self, g, boundary=None):
        """boundary: a list - the two embedding outputs are then cut out of the autograd graph where the transformer takes
        them, and the list receives (output, detached twin) pairs: backward() of the loss fills the twins' .grad, and
        torch.autograd.backward(outputs, twin gradients) continues through the embedding."""
        ld = g["ligand_data"]
        if self.config.train.num_props:
            cols = {"vina_score": torch.lt(ld["vina_score"], -7.5), "qed": torch.gt(ld["qed"], 0.6),
                    "sas": torch.lt(ld["sas"], 4.0)}                                         # GAN:38-40
            prop = torch.stack([(cols[p] if p in cols else ld[p]).to(torch.float32) for p in self.config.train.prop], 1)
        else:
            prop = None
        batch, batch_aa = g[PA]["batch"], g[LA]["batch"]
        embed = self.embedding(g)
        xa, xl = embed[PA].embedding, embed[LA].embedding
        if boundary is not None:
            da, dl = xa.detach().requires_grad_(True), xl.detach().requires_grad_(True)
            boundary += [(xa, da), (xl, dl)]
            xa, xl = da, dl
        feat = self.config.model.featurizer_feat_dim
        knn = getattr(g, "extras", {}).get("knn", {})
        prep = getattr(g, "extras", {}).get("prepared")
        return self.model(
            node_attr=xa.reshape(-1, feat), pos=g[PA]["pos"], batch=batch,
            atom_laplacian=lap_pe(g, PA), smiles_index=ld["smiIndices_input"],
            tgt_len=self.config.model.decoder.tgt_len, aa_node_attr=xl.reshape(-1, feat),
            aa_pos=g[LA]["pos"], aa_batch=batch_aa, aa_laplacian=lap_pe(g, LA), prop=prop,
            knn=knn.get(PA), aa_knn=knn.get(LA), prep=prep)

    def prepare(self, g):
        """Per-batch graph structure that does not depend on parameters: destination-sorted edge sets of the four edge
        types, kNN graphs of both encoders, dense-batch maps.  Everything with a host synchronisation lives here, so that
        forward/backward on a prepared batch only enqueue work (and can be captured in a HIP graph)."""
        from .EF_layers import edge_set
        from ..graph import E_LL, E_LP, E_PL, E_PP
        from ..graph import laplacian_pe_batched
        n_p, n_l = g[PA]["x"].shape[0], g[LA]["x"].shape[0]
        B = g.num_graphs
        for nt, et in ((PA, E_PP), (LA, E_LL)):
            if "lap_pe" not in g[nt]:      # deterministic per-graph Laplacian PE, batched eigensolve on the GPU
                g[nt]["lap_pe"] = laplacian_pe_batched(g[et]["edge_index"], g[nt]["batch"], B, self.config.model.encoder.lap_dim)
        knn = getattr(g, "extras", {}).get("knn", {})
        pad = getattr(g, "extras", {}).get("pad") or {}         # fixed capacities of a padded batch (graph.pad_batch)
        nr = pad.get("n_real", {})
        prep = {"p": self.model.encoder.prepare(g[PA]["pos"], g[PA]["batch"], B, knn.get(PA), pad.get("mx_p"),
                                                pad.get("knn_p"), nr.get(PA)),
                "l": self.model.encoder2.prepare(g[LA]["pos"], g[LA]["batch"], B, knn.get(LA), pad.get("mx_l"),
                                                 pad.get("knn_l"), nr.get(LA))}
        prep.update({
                "es": {"pp": edge_set(g[E_PP]["edge_index"], n_p, n_p), "ll": edge_set(g[E_LL]["edge_index"], n_l, n_l),
                       "lp": edge_set(g[E_LP]["edge_index"], n_l, n_p), "pl": edge_set(g[E_PL]["edge_index"], n_p, n_l)}})
        if self.embedding.fuse_homo_passes:
            prep["homo"] = self.embedding.make_homo(g)
        g.extras["prepared"] = prep
        return prep
