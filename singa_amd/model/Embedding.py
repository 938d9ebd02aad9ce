"""EquivariantEmbedding for MI355X (reference model/Embedding.py = "EMB").

Same constructor (`config.embedding`, device), same parameter names, same `forward(g, batch=None, gen_mode=False)`
returning `{'protein_atoms', 'ligand_atoms', 'lp_edge', 'pl_edge'}` of SO3_Embedding.  The four passes
(protein-protein, ligand-ligand, ligand->protein, protein->ligand) and every quirk that changes numbers
 are reproduced; the host-side Python loops of the reference (barcode strings per
node, EMB:250-253) are vectorised tensor ops.
"""
from typing import Dict, Optional

import torch
import torch.nn as nn

from .. import ops
from ..graph import E_LL, E_LP, E_PL, E_PP, LA, PA
from .EF_layers import (CoefficientMappingModule, EdgeDegreeEmbedding, GaussianSmearing, ModuleListInfo, SO3_Embedding,
                        SO3_Grid, SO3_Rotation, TransBlockV2, get_normalization_layer, init_edge_rot_mat,
                        forward_pass)

_AVG_NUM_MODES = 77.81317
_AVG_DEGREE = 23.395238876342773


def barcode(x: torch.Tensor) -> torch.Tensor:
    """Last 15 feature columns truncated to integers and read as a binary number, MSB first (EMB:250-253, Q2)."""
    bits = x[:, -15:].to(torch.long)
    weights = (2 ** torch.arange(14, -1, -1, device=x.device, dtype=torch.long))
    return (bits * weights).sum(1)


class EquivariantEmbedding(nn.Module):
    def __init__(self, config, device: str = "cuda") -> None:
        super().__init__()
        self.device = device
        self.edge_channels = config.edge_channels
        self.sphere_channels = config.sphere_channels
        self.attn_hidden_channels = config.attn_hidden_channels
        self.attn_alpha_channels = config.attn_alpha_channels
        self.attn_value_channels = config.attn_value_channels
        self.ffn_hidden_channels = config.ffn_hidden_channels
        self.lmax_list = [int(i) for i in config.lmax_list]
        self.mmax_list = [int(i) for i in config.mmax_list]
        self.num_resolutions = len(self.lmax_list)
        assert self.num_resolutions == 1
        self.sphere_channels_all = self.num_resolutions * self.sphere_channels
        self.max_num_elements = config.max_num_elements
        self.num_heads = config.num_heads
        self.num_layers = config.num_layers
        self.offset_res = 0
        self.cutoff = config.cutoff
        self.norm_type = config.norm_type
        self.share_atom_edge_embedding = config.share_atom_edge_embedding
        self.use_atom_edge_embedding = config.use_atom_edge_embedding
        assert self.share_atom_edge_embedding and self.use_atom_edge_embedding, "shipped config (config/train.yml:44-45)"
        assert config.alpha_drop == 0.0 and config.proj_drop == 0.0 and config.drop_path_rate == 0.0
        self.block_use_atom_edge_embedding = False
        # when True, hetero layers whose outputs the reference discards only apply their norm_1 side effect (Q4)
        self.skip_dead_hetero_layers = True
        # Parts 1 and 2 (protein-protein and ligand-ligand, same blocks, disjoint node sets) as ONE pass over their union
        self.fuse_homo_passes = True
        import os
        # Part 3's live layer on a second stream beside Part 4 (see _forward); SINGA_OVERLAP_HETERO=0: one stream (lab)
        self.overlap_hetero_passes = os.environ.get("SINGA_OVERLAP_HETERO", "1") == "1"

        self.SO3_rotation = nn.ModuleList([SO3_Rotation(self.lmax_list[0], device=device)])
        self.sphere_embedding = nn.Embedding(self.max_num_elements, self.sphere_channels_all, device=device)
        self.sphere_embedding_2 = nn.Embedding(32767, self.sphere_channels_all, device=device)
        self.distance_expansion = GaussianSmearing(0.0, self.cutoff, self.edge_channels, 20, device=device)
        self.edge_channels_list = [int(self.distance_expansion.num_output)] + [self.edge_channels] * 2
        self.source_embedding = nn.Embedding(self.max_num_elements, self.edge_channels_list[-1], device=device)
        self.target_embedding = nn.Embedding(self.max_num_elements, self.edge_channels_list[-1], device=device)
        self.edge_channels_list[0] = self.edge_channels_list[0] + 2 * self.edge_channels_list[-1]
        self.mappingReduced = CoefficientMappingModule(self.lmax_list, self.mmax_list, device=device)
        self.SO3_grid = ModuleListInfo("({}, {})".format(max(self.lmax_list), max(self.lmax_list)))
        self.edge_degree_embedding = EdgeDegreeEmbedding(
            sphere_channels=self.sphere_channels, lmax_list=self.lmax_list, mmax_list=self.mmax_list,
            SO3_rotation=self.SO3_rotation, mappingReduced=self.mappingReduced, max_num_elements=self.max_num_elements,
            edge_channels_list=self.edge_channels_list, use_atom_edge_embedding=False, rescale_factor=_AVG_DEGREE,
            device=device)
        self.blocks = nn.ModuleList()
        for _ in range(self.num_layers):
            self.blocks.append(TransBlockV2(
                SO3_rotation=self.SO3_rotation, SO3_grid=self.SO3_grid, mappingReduced=self.mappingReduced,
                sphere_channels=self.sphere_channels, attn_hidden_channels=self.attn_hidden_channels,
                attn_alpha_channels=self.attn_alpha_channels, attn_value_channels=self.attn_value_channels,
                ffn_hidden_channels=self.ffn_hidden_channels, output_channels=self.sphere_channels,
                lmax_list=self.lmax_list, mmax_list=self.mmax_list, num_heads=self.num_heads,
                max_num_elements=self.max_num_elements, edge_channels_list=self.edge_channels_list,
                use_atom_edge_embedding=self.block_use_atom_edge_embedding, norm_type=self.norm_type,
                use_m_share_rad=False, use_s2_act_attn=False, use_attn_renorm=True, use_gate_act=False,
                use_grid_mlp=False, use_sep_s2_act=True, alpha_drop=0.0, proj_drop=0.0, drop_path_rate=0.0,
                device=device))
        self.norm = get_normalization_layer(self.norm_type, lmax=max(self.lmax_list), num_channels=self.sphere_channels,
                                            device=device)

    # ------------------------------------------------------------------------------------------------------------
    def _frames(self, g, key: str, vec: torch.Tensor) -> torch.Tensor:
        ex = getattr(g, "extras", {})
        if "edge_rot_mat" in ex and key in ex["edge_rot_mat"]:
            return ex["edge_rot_mat"][key]
        rand = ex["rot_rand"][key] if "rot_rand" in ex and key in ex["rot_rand"] else None
        return init_edge_rot_mat(vec, rand=rand)

    def _edge_scalars(self, dist, z_src, z_dst, ei):
        return torch.cat((self.distance_expansion(dist), ops.embedding(self.source_embedding.weight, z_src[ei[0]]),
                          ops.embedding(self.target_embedding.weight, z_dst[ei[1]])), dim=1)

    def _homo_pass(self, g, node_type, edge_type, key):
        z = g["atomicnum"][node_type]
        x_feat, pos = g[node_type]["x"], g[node_type]["pos"]
        ei = g[edge_type]["edge_index"]
        ev = pos[ei[0]] - pos[ei[1]]
        self.SO3_rotation[0].set_wigner(self._frames(g, key, ev))
        K = (self.lmax_list[0] + 1) ** 2
        # Q1: the reference stores the l=0 initialisation in a LONG tensor: truncation toward zero, no gradient
        with torch.no_grad():
            init = (self.sphere_embedding.weight.index_select(0, z)
                    + self.sphere_embedding_2.weight.index_select(0, barcode(x_feat))).to(torch.long)
        emb = torch.zeros(z.shape[0], K, self.sphere_channels, device=pos.device, dtype=torch.float32)
        emb[:, self.offset_res, :] = init.to(torch.float32)
        edge_distance = self._edge_scalars(ev.norm(dim=-1), z, z, ei)
        edge_degree = self.edge_degree_embedding(z, edge_distance, ei, hetero=False)
        x = SO3_Embedding(0, self.lmax_list, self.sphere_channels, torch.float32, self.device,
                          emb + edge_degree.embedding)
        for i in range(self.num_layers):
            x = self.blocks[i](x=x, atomic_numbers=z, edge_distance=edge_distance, edge_index=ei, batch=len(z),
                               hetero=False)
        x.embedding = self.norm(x.embedding)
        return x

    def _hetero_pass(self, g, x_dict, atomic_numbers, edge_type, source, target, ev_dist, batch, deferred=False):
        """deferred: everything that CHANGES the shared dict (the edge-degree sum and the in-place norm_1 of every layer,
        Q4) happens now; what only reads it - the last layer's attention + feed-forward block and the final norm, i.e. the
        pass's result - is returned as a function to be called later (on another stream: _forward)."""
        ei = g[edge_type]["edge_index"]
        edge_distance = self._edge_scalars(ev_dist, atomic_numbers[source], atomic_numbers[target], ei)
        edge_degree = self.edge_degree_embedding(atomic_numbers, edge_distance, ei, hetero=True,
                                                 source_target=(source, target))
        x_dict[target].embedding = x_dict[target].embedding + edge_degree.embedding
        if deferred and self.skip_dead_hetero_layers:
            for i in range(self.num_layers - 1):
                self.blocks[i].renorm_only(x_dict, (source, target))
            last = self.blocks[self.num_layers - 1]
            x_res = last.renorm_with_residual(x_dict, (source, target))
            mk = lambda t: SO3_Embedding(0, self.lmax_list, self.sphere_channels, torch.float32, self.device, t)
            view = {source: mk(x_dict[source].embedding), target: mk(x_dict[target].embedding)}   # what the last layer reads

            def compute():
                x = last(x=view, atomic_numbers=atomic_numbers, edge_distance=edge_distance, edge_index=ei, batch=batch,
                         hetero=True, source_target=(source, target), renormed_residual=x_res)
                x.embedding = self.norm(x.embedding)
                return x
            return compute
        x = None
        for i in range(self.num_layers):
            if self.skip_dead_hetero_layers and i < self.num_layers - 1:
                self.blocks[i].renorm_only(x_dict, (source, target))
                continue
            x = self.blocks[i](x=x_dict, atomic_numbers=atomic_numbers, edge_distance=edge_distance, edge_index=ei,
                               batch=batch, hetero=True, source_target=(source, target))
        x.embedding = self.norm(x.embedding)
        return x

    def make_homo(self, g):
        """Disjoint union of the two homogeneous graphs (protein-protein and ligand-ligand bonds): Parts 1 and 2 of the
        reference (EMB:226-370) run the SAME blocks with hetero=False on two node sets that never interact, so they are
        one pass over [protein atoms ; ligand atoms] with the ligand edges offset by the protein count.  Halves the
        number of launches of the homogeneous passes and folds the tiny ligand pass (2 k edges) into GEMMs that are
        already large.  Part of `prepare` (no parameters involved)."""
        from .EF_layers import edge_set
        n_p = g[PA]["x"].shape[0]
        n_l = g[LA]["x"].shape[0]
        ei = torch.cat([g[E_PP]["edge_index"], g[E_LL]["edge_index"] + n_p], dim=1).contiguous()
        return {"ei": ei, "n_p": n_p, "n_l": n_l, "es": edge_set(ei, n_p + n_l, n_p + n_l)}

    def _homo_union_pass(self, g, homo):
        n_p = homo["n_p"]
        ei = homo["ei"]
        z = torch.cat([g["atomicnum"][PA], g["atomicnum"][LA]])
        pos = torch.cat([g[PA]["pos"], g[LA]["pos"]])
        code = torch.cat([barcode(g[PA]["x"]), barcode(g[LA]["x"])])
        ev = pos[ei[0]] - pos[ei[1]]
        e_pp = g[E_PP]["edge_index"].shape[1]
        rot = torch.cat([self._frames(g, "pp", ev[:e_pp]), self._frames(g, "ll", ev[e_pp:])], 0)
        self.SO3_rotation[0].set_wigner(rot)
        K = (self.lmax_list[0] + 1) ** 2
        with torch.no_grad():                                  # Q1: long-typed store truncates and cuts autograd
            init = (self.sphere_embedding.weight.index_select(0, z)
                    + self.sphere_embedding_2.weight.index_select(0, code)).to(torch.long)
        emb = torch.zeros(z.shape[0], K, self.sphere_channels, device=pos.device, dtype=torch.float32)
        emb[:, self.offset_res, :] = init.to(torch.float32)
        edge_distance = self._edge_scalars(ev.norm(dim=-1), z, z, ei)
        edge_degree = self.edge_degree_embedding(z, edge_distance, ei, hetero=False)
        x = SO3_Embedding(0, self.lmax_list, self.sphere_channels, torch.float32, self.device, emb + edge_degree.embedding)
        for i in range(self.num_layers):
            x = self.blocks[i](x=x, atomic_numbers=z, edge_distance=edge_distance, edge_index=ei, batch=len(z), hetero=False)
        out = self.norm(x.embedding)
        mk = lambda t: SO3_Embedding(0, self.lmax_list, self.sphere_channels, torch.float32, self.device, t)
        return mk(out[:n_p]), mk(out[n_p:])

    def forward(self, g, batch: Optional[int] = None, gen_mode: bool = False) -> Dict:
        with forward_pass():
            return self._forward(g, batch, gen_mode)

    def _forward(self, g, batch: Optional[int] = None, gen_mode: bool = False) -> Dict:
        x_dict = {}
        if getattr(g, "num_graphs", 1) > 1:
            batch = 64
        if gen_mode:
            x_dict[PA] = self._homo_pass(g, PA, E_PP, "pp")                     # Part 1 only, EMB:297-298
            return x_dict
        if self.fuse_homo_passes:
            prep = getattr(g, "extras", {}).get("prepared") or {}
            x_dict[PA], x_dict[LA] = self._homo_union_pass(g, prep.get("homo") or self.make_homo(g))
        else:
            x_dict[PA] = self._homo_pass(g, PA, E_PP, "pp")                     # Part 1, EMB:226-295
            x_dict[LA] = self._homo_pass(g, LA, E_LL, "ll")                     # Part 2, EMB:305-370
        atomic_numbers = {PA: g["atomicnum"][PA], LA: g["atomicnum"][LA]}
        pos_p, pos_l = g[PA]["pos"], g[LA]["pos"]
        # Part 3: ligand -> protein (EMB:379-428)
        lp_ei = g[E_LP]["edge_index"]
        lp_ev = pos_l[lp_ei[0]] - pos_p[lp_ei[1]]
        self.SO3_rotation[0].set_wigner(self._frames(g, "lp", lp_ev))
        if batch is None:
            batch = len(atomic_numbers[PA])
        pl_ei = g[E_PL]["edge_index"]
        pl_ev = pos_p[pl_ei[0]] - pos_l[pl_ei[1]]
        if self.overlap_hetero_passes and self.skip_dead_hetero_layers and pos_p.is_cuda:
            # Part 4 only depends on what Part 3 does to the shared dict (the edge-degree sum and the norm_1 chain, Q4), not
            # on Part 3's result: Part 3's live layer (attention over the interaction edges + the feed-forward block over
            # all protein atoms) runs on a second stream while Part 4 (the same on the few ligand atoms: latency-bound
            # launches) runs here.  Same arithmetic in the same order inside each pass; autograd mirrors the split in the
            # backward pass and a HIP-graph capture records two parallel branches (as for the two CProMG encoders).
            cur = torch.cuda.current_stream()
            aux = ops.branch_stream(pos_p.device)       # the same second stream the transformer uses: never three branches
            part3 = self._hetero_pass(g, x_dict, atomic_numbers, E_LP, LA, PA, lp_ev.norm(dim=-1), batch, deferred=True)
            fork = torch.cuda.Event()
            fork.record(cur)
            with torch.cuda.stream(aux):
                aux.wait_event(fork)
                x_dict["lp_edge"] = part3()
            # Part 4: protein -> ligand, REUSING the Part-3 frames edge by edge (EMB:437-475, Q5)
            x_dict["pl_edge"] = self._hetero_pass(g, x_dict, atomic_numbers, E_PL, PA, LA, pl_ev.norm(dim=-1), batch)
            cur.wait_stream(aux)
            x_dict["lp_edge"].embedding.record_stream(cur)
        else:
            x_dict["lp_edge"] = self._hetero_pass(g, x_dict, atomic_numbers, E_LP, LA, PA, lp_ev.norm(dim=-1), batch)
            # Part 4: protein -> ligand, REUSING the Part-3 frames edge by edge (EMB:437-475, Q5)
            x_dict["pl_edge"] = self._hetero_pass(g, x_dict, atomic_numbers, E_PL, PA, LA, pl_ev.norm(dim=-1), batch)
        x_dict[PA].embedding = x_dict[PA].embedding + x_dict["lp_edge"].embedding
        x_dict[LA].embedding = x_dict[LA].embedding + x_dict["pl_edge"].embedding
        return x_dict
