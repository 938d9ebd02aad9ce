"""CProMG transformer for MI355X (reference model/CProMG.py = "CP"): kNN-graph attention encoders for protein and
ligand atoms, cross-attention at layers 2 and 5, causal SMILES decoder, vocabulary projection.

Same class names, constructor arguments and parameter names as the reference.  Differences inside:
  * graph construction (kNN, to_undirected(mean), Gaussian smearing, get_laplacian; CP:293-298) is done with
    batched tensor ops on the GPU and the edges are kept sorted by centre node, so that
  * scatter_softmax / scatter_sum (CP:66,74) are the segmented HIP kernels (ops.segment_softmax / segment_wsum);
  * grouped 1x1 Conv1d and position-wise Conv1d layers are evaluated as GEMMs on their own weights;
  * the causal mask is built on the device (the reference builds it with numpy on the host, CP:507-514).
Dense projections run on the library's own f32 MFMA GEMM, the dense attentions on its flash-style MFMA kernel (k19).
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import BatchNorm1d, Conv1d, Dropout, Embedding, LayerNorm, Sequential

from .. import ops
from ..nn import Linear


class ShiftedSoftplus(nn.Module):
    def __init__(self, device="cuda"):
        super().__init__()
        self.shift = math.log(2.0)

    def forward(self, x):
        return F.softplus(x) - self.shift


class GaussianSmearing(nn.Module):
    def __init__(self, start=0.0, stop=10.0, num_gaussians=50, device="cuda"):
        super().__init__()
        offset = torch.linspace(start, stop, num_gaussians, device=device)
        self.coeff = -0.5 / (offset[1] - offset[0]).item() ** 2
        self.register_buffer("offset", offset)

    def forward(self, dist):
        dist = dist.view(-1, 1) - self.offset.view(1, -1)
        return torch.exp(self.coeff * torch.pow(dist, 2))


# ----------------------------------------------------------------------------------------------- graph helpers (GPU)
class DenseMap:
    """to_dense_batch bookkeeping for one node type of one batch: flat positions of the nodes inside the
    padded [B, max_nodes] layout and the validity mask.  Built once per batch (one host sync for max_nodes) and reused
    by every layer, so the forward itself stays free of host synchronisation.  Nodes whose batch id is >= batch_size
    (the inert padding atoms of graph.pad_batch) are outside the layout: `dense` drops them into a trash row behind the
    last real one, `gather` hands them row 0 (their values are never used)."""

    def __init__(self, batch, batch_size, mx=None):
        inside = batch < batch_size
        bc = batch.clamp(max=batch_size)
        num = torch.zeros(batch_size + 1, dtype=torch.int64, device=batch.device).index_add_(0, bc, torch.ones_like(batch))
        num = num[:batch_size]
        self.B = batch_size
        self.mx = int(num.max()) if mx is None else int(mx)
        start = num.cumsum(0) - num
        self.ptr = torch.cat([start, start[-1:] + num[-1:]])      # first node of every graph (+ the end): nodes are contiguous per graph
        trash = batch_size * self.mx
        bi = batch.clamp(max=batch_size - 1)
        self.idx = torch.where(inside, torch.arange(batch.numel(), device=batch.device) - start[bi] + bi * self.mx,
                               torch.full_like(batch, trash))
        self.gidx = torch.where(inside, self.idx, torch.zeros_like(self.idx))
        mask = torch.zeros(trash + 1, dtype=torch.bool, device=batch.device)
        mask[self.idx] = True
        self.mask = mask[:trash].view(batch_size, self.mx)
        self.pad_mask = ~self.mask.unsqueeze(1)

    def tensors(self):
        return [self.idx, self.gidx, self.mask, self.pad_mask]

    def dense(self, x):
        out = x.new_zeros((self.B * self.mx + 1,) + tuple(x.shape[1:]))
        out = out.index_copy(0, self.idx, x)
        return out[:self.B * self.mx].view(self.B, self.mx, *x.shape[1:])

    def gather(self, dense_flat):
        """Per-node rows of a [B * max_nodes, ...] tensor (the inverse of `dense`)."""
        return dense_flat.index_select(0, self.gidx)


def to_dense_batch(x, batch, batch_size):
    dm = DenseMap(batch, batch_size)
    return dm.dense(x), dm.mask, dm.idx


def knn_graph(pos, k, batch, batch_size, dm=None):
    """For every node its k nearest other nodes of the same graph, row = centre (torch_cluster.knn_graph with
    flow='target_to_source', CP:293,330): the library's kernel singa_knn_graph (one wavefront per centre atom, exact fp32
    coordinate differences as torch_cluster forms them, ties to the lower index).  Returns a fixed-size [2, N * k] list in
    which slots that do not exist (graphs with fewer than k+1 nodes, atoms of no graph) hold -1: no data-dependent
    compaction (and no host synchronisation) happens here."""
    dm = dm or DenseMap(batch, batch_size)
    return ops.knn_graph(pos, k, batch, dm.ptr, dm.mx)


class KnnEdges:
    """CP:295-298 on a raw kNN edge list (entries < 0 = absent): edge lengths, to_undirected(reduce='mean'), Gaussian
    smearing and get_laplacian (self-loops appended; 2-D weights, Q12), then sorted by centre node -> CSR row_ptr.
    Coalescing is a radix sort + adjacent-difference + prefix sum (two scalar read-backs for the counts)."""

    def __init__(self, pos, knn_ei, smear, cap=None, n_real=None):
        """cap / n_real: pad the edge list (before the N self loops) with inert edges among the padding atoms
        [n_real, N) up to `cap` edges in total, so that the arrays have a fixed size (graph.pad_batch)."""
        N = pos.shape[0]
        dev = pos.device
        valid = (knn_ei[0] >= 0) & (knn_ei[1] >= 0)
        a, b = knn_ei[0].clamp(min=0), knn_ei[1].clamp(min=0)
        ln = (pos[a] - pos[b]).norm(dim=1)
        sentinel = N * N
        fwd = torch.where(valid, a * N + b, torch.full_like(a, sentinel))
        bwd = torch.where(valid, b * N + a, torch.full_like(a, sentinel))
        skey, perm = torch.sort(torch.cat([fwd, bwd]))
        new = torch.ones_like(skey, dtype=torch.bool)
        new[1:] = skey[1:] != skey[:-1]
        gid = torch.cumsum(new, 0) - 1
        # a kNN list holds every (centre, neighbour) pair once, so a key occurs once (one direction only) or twice (mutual
        # neighbours); `triple` tells whether some key occurs more often (a caller-supplied list with repeats)
        # (the sentinel key of absent slots repeats freely and is dropped below: not counted)
        triple = ((~new[2:] & ~new[1:-1] & (skey[2:] != sentinel)).any() if skey.numel() > 2
                  else torch.zeros((), dtype=torch.bool, device=dev))
        n_groups, last, triple = torch.stack([gid[-1] + 1, skey[-1], triple.to(gid.dtype)]).tolist()   # ONE read-back
        n_edges = n_groups - (1 if last == sentinel else 0)
        ukey = torch.zeros(n_groups, dtype=torch.int64, device=dev).scatter_(0, gid, skey)[:n_edges]
        ln2 = torch.cat([ln, ln])[perm]
        if not triple:
            # mean over the one or two members of each group without atomics (index_add_ on 4.4 M sorted indices is
            # ~9 ms of serialised atomics at the bench size): both members compute the same mean and scatter it
            nxt = torch.cat([ln2[1:], ln2[-1:]])
            prv = torch.cat([ln2[:1], ln2[:-1]])
            has_next = torch.cat([~new[1:], new.new_zeros(1)])
            mean = torch.where(new, torch.where(has_next, 0.5 * (ln2 + nxt), ln2), 0.5 * (prv + ln2))
            ln = torch.zeros(n_groups, device=dev).scatter_(0, gid, mean)[:n_edges]
        else:
            cnt = torch.zeros(n_groups, device=dev).index_add_(0, gid, torch.ones_like(ln2))[:n_edges]
            ln = torch.zeros(n_groups, device=dev).index_add_(0, gid, ln2)[:n_edges] / cnt
        row, col = ukey // N, ukey % N
        fused = ln.is_cuda and smear.offset.numel() == 64     # k-n1: the attribute tensor is written once, in its final layout
        ea = None if fused else smear(ln)
        self.n_edges = n_edges + N                       # undirected kNN edges + self loops, before any padding
        if cap is not None:
            extra = cap - self.n_edges
            nd = N - n_real
            if extra < 0 or nd < 2:
                raise OverflowError(f"KnnEdges: {self.n_edges} edges do not fit the capacity {cap}")
            i = torch.arange(extra, device=dev)
            ps = (i * nd) // max(extra, 1)               # non-decreasing: rows stay sorted (the padding atoms come last)
            row = torch.cat([row, n_real + ps])
            col = torch.cat([col, n_real + (ps + 1 + i % (nd - 1)) % nd])
            if not fused:
                ea = torch.cat([ea, ea.new_zeros(extra, ea.shape[1])])
        loop = torch.arange(N, device=dev)
        bounds = torch.arange(N + 1, device=dev)
        if ln.is_cuda:
            # `row` is sorted already (coalesced keys, then the padding rows) and the self loop of node i goes behind the last
            # edge of row i - what a stable argsort of [row ; loop] gives, without the sort: edge e of row r lands at e + r, the
            # loop of node i at (number of edges with row <= i) + i.  (One 3 M-key radix sort less per encoder and batch.)
            E0 = row.shape[0]
            seg_ptr = torch.searchsorted(row, bounds)                      # first edge of every row; [1:] = edges with row <= i
            order = torch.empty(E0 + N, dtype=torch.int64, device=dev)     # the permutation that stable argsort would return
            order[torch.arange(E0, device=dev) + row] = torch.arange(E0, device=dev)
            order[seg_ptr[1:] + loop] = E0 + loop
            row, col = torch.cat([row, loop]), torch.cat([col, loop])
            self.row, self.col = row[order], col[order]
            seg32 = seg_ptr.to(torch.int32)
            if fused:
                # Gaussian smearing, get_laplacian's -w / degree rows (per-node sums of the edge features, Q12) and the placement
                # in the sorted order as ONE kernel (singa_knn_edge_attr)
                self.attr = ops.knn_edge_attr(ln, seg32, E0, n_edges, smear.offset, smear.coeff)
            else:
                # per-node sum of the edge features (get_laplacian's degree for 2-D weights, Q12).  `row` is sorted, so this is a
                # segment sum - index_add_ spent 9 ms per call on ~2e8 float atomics at the bench size
                deg = ops.segment_sum_rows(ea.contiguous(), seg32)
                self.attr = torch.cat([-ea, deg], 0)[order].contiguous()
        else:
            deg = torch.zeros(N, ea.shape[1], device=dev).index_add_(0, row, ea)
            row, col = torch.cat([row, loop]), torch.cat([col, loop])
            attr = torch.cat([-ea, deg], 0)
            order = torch.argsort(row, stable=True)
            self.row, self.col, self.attr = row[order], col[order], attr[order].contiguous()
        # CSR pointers by binary search in the sorted index lists (no atomics)
        self.row_ptr = torch.searchsorted(self.row, bounds).to(torch.int32)
        self.row32, self.col32 = self.row.to(torch.int32), self.col.to(torch.int32)
        # edges grouped by neighbour (col): needed by the gradients of the gathered key / value rows (32-bit keys: half the
        # radix passes of the 64-bit sort)
        eperm = torch.argsort(self.col32 if ln.is_cuda else self.col, stable=True)
        self.eperm = eperm.to(torch.int32)
        self.col_ptr = torch.searchsorted(self.col[eperm], bounds).to(torch.int32)
        self.N = N

    def tensors(self):
        return [self.row, self.col, self.attr, self.row_ptr, self.row32, self.col32, self.eperm, self.col_ptr]


# ----------------------------------------------------------------------------------------------- attention modules
class MultiHeadAttention(nn.Module):
    """Graph attention over the kNN edges (CP:19-78)."""

    def __init__(self, hidden_channels, edge_channels, key_channels, num_heads=1, device="cuda"):
        super().__init__()
        assert hidden_channels % num_heads == 0 and key_channels % num_heads == 0
        self.num_heads = num_heads
        self.k_lin = Conv1d(hidden_channels, key_channels, 1, groups=num_heads, bias=False, device=device)
        self.q_lin = Conv1d(hidden_channels, key_channels, 1, groups=num_heads, bias=False, device=device)
        self.v_lin = Conv1d(hidden_channels, hidden_channels, 1, groups=num_heads, bias=False, device=device)
        kh, vh = key_channels // num_heads, hidden_channels // num_heads
        self.weight_k_net = Sequential(Linear(edge_channels, kh, device=device), ShiftedSoftplus(), Linear(kh, kh, device=device))
        self.weight_k_lin = Linear(kh, kh, device=device)
        self.weight_v_net = Sequential(Linear(edge_channels, vh, device=device), ShiftedSoftplus(), Linear(vh, vh, device=device))
        self.weight_v_lin = Linear(vh, vh, device=device)
        self.centroid_lin = Linear(hidden_channels, hidden_channels, device=device)
        self.act = ShiftedSoftplus()
        self.out_transform = Linear(hidden_channels, hidden_channels, device=device)
        self.layer_norm = LayerNorm(hidden_channels, device=device)

    def _edge_mlp(self, net, x):
        # first Linear as a plain GEMM, its bias inside the activation kernel (one pass instead of three)
        return net[2](ops.bias_ssp(ops.linear(x, net[0].weight), net[0].bias))

    def forward(self, node_attr, edges: KnnEdges):
        """CP:50-78.  weight_k_lin and weight_v_lin act on the last axis only, so they commute with the per-edge
        scaling and with the alpha-weighted sum: q.(W(w*k)+b) = (qW).(w*k) + q.b and sum_e a_e (W(w_e*v)+b) =
        W sum_e a_e (w_e*v) + b (sum_e a_e = 1).  They are therefore applied to NODE tensors, and the two per-edge
        contractions run as fused gather kernels without any [E,heads,channels] intermediate."""
        N = node_attr.size(0)
        h_keys, h_queries, h_values = ops.grouped_linear3(node_attr, self.k_lin.weight, self.q_lin.weight, self.v_lin.weight,
                                                          self.num_heads)
        scale = 1.0 / math.sqrt(h_keys.size(-1))
        if edges.attr.shape[1] == 64 and self.weight_k_net[0].out_features == 32 and self.weight_v_net[0].out_features == 64:
            W_k, W_v = ops.edge_mlp_pair(edges.attr, (self.weight_k_net[0], self.weight_k_net[2]),
                                         (self.weight_v_net[0], self.weight_v_net[2]))        # [E, 32], [E, 64]: k15c
        else:                                    # other widths: the same two MLPs as library GEMMs + the k15d kernel
            W_k = self._edge_mlp(self.weight_k_net, edges.attr)
            W_v = self._edge_mlp(self.weight_v_net, edges.attr)
        qp = ops.linear_nn(h_queries, self.weight_k_lin.weight)                  # (q W)[n,h,:]
        cterm = ops.rowdot_bias(h_queries, self.weight_k_lin.bias, scale)
        qk_ij = ops.edge_logits(qp, W_k, h_keys, cterm, edges, scale)
        alpha = ops.segment_softmax(qk_ij, edges.row_ptr, 0.0)
        S = ops.gather_wsum(alpha, W_v, h_values, edges)                         # [N, heads, 64]
        aggr_msg = ops.linear(S, self.weight_v_lin.weight, self.weight_v_lin.bias).view(N, -1)
        # centroid_lin(node_attr) + aggr_msg, then ShiftedSoftplus (CP:76-77): the sum rides in the GEMM's epilogue, the
        # Linear's bias in the activation kernel (k15d)
        out = ops.bias_ssp(ops.linear_add(node_attr, self.centroid_lin.weight, None, aggr_msg), self.centroid_lin.bias)
        return ops.layer_norm_residual(self.out_transform(out), None, self.layer_norm)


def _dense_attention(module, Q, K, V, attn_mask, key_channels, hidden_channels):
    B, heads = Q.size(0), module.num_heads
    dk, dv = key_channels // heads, hidden_channels // heads
    scale = 1.0 / math.sqrt(dk)
    if dk == 32 and dv == 64:       # the shipped head geometry: scores never leave the MFMA registers (k19), and the kernel
        # reads q / k / v in the projections' own [B, T, heads, d] layout and writes the context in it: no head transposes.
        # Projections that share their input are ONE launch (W_Q | W_K | W_V of a self attention, W_K | W_V of a cross
        # attention); the kernel reads their column blocks in place
        WQ, WK, WV = ((m.weight, m.bias) for m in (module.W_Q, module.W_K, module.W_V))
        if Q is K and K is V:
            q_s, k_s, v_s = ops.linear_multi(Q, [WQ, WK, WV])
        elif K is V:
            q_s = module.W_Q(Q)
            k_s, v_s = ops.linear_multi(K, [WK, WV])
        else:
            q_s, k_s, v_s = module.W_Q(Q), module.W_K(K), module.W_V(V)
        context = ops.attention(q_s.view(B, -1, heads, dk), k_s.view(B, -1, heads, dk), v_s.view(B, -1, heads, dv), attn_mask,
                                scale, heads, token_major=True)
        return ops.layer_norm_residual(module.linear(context.view(B, -1, hidden_channels)), Q, module.layer_norm)
    q_s = module.W_Q(Q).view(B, -1, heads, dk).transpose(1, 2)
    k_s = module.W_K(K).view(B, -1, heads, dk).transpose(1, 2)
    v_s = module.W_V(V).view(B, -1, heads, dv).transpose(1, 2)
    T, S = q_s.size(2), k_s.size(2)
    flat = lambda t: t.reshape(B * heads, t.size(2), t.size(3))
    probs = ops.masked_softmax(ops.bmm_small(flat(q_s), flat(k_s).transpose(1, 2)), attn_mask, scale, heads)
    context = ops.bmm_small(probs, flat(v_s)).view(B, heads, T, -1)
    context = context.transpose(1, 2).contiguous().view(B, -1, hidden_channels)
    return ops.layer_norm_residual(module.linear(context), Q, module.layer_norm)


class MultiHeadAttention2(nn.Module):
    """Dense cross-attention ligand atoms -> protein atoms (CP:81-105)."""

    def __init__(self, hidden_channels, key_channels, num_heads, device="cuda"):
        super().__init__()
        self.hidden_channls, self.keys_channels, self.num_heads = hidden_channels, key_channels, num_heads
        self.W_Q = Linear(hidden_channels, key_channels, device=device)
        self.W_K = Linear(hidden_channels, key_channels, device=device)
        self.W_V = Linear(hidden_channels, hidden_channels, device=device)
        self.linear = Linear(hidden_channels, hidden_channels, device=device)
        self.layer_norm = LayerNorm(hidden_channels, device=device)

    def forward(self, Q, K, V, attn_mask):
        return _dense_attention(self, Q, K, V, attn_mask, self.keys_channels, self.hidden_channls)


class MultiHeadDeAttention(nn.Module):
    """Decoder self / encoder-decoder attention (CP:134-158)."""

    def __init__(self, hidden_channels, key_channels, num_heads, device="cuda"):
        super().__init__()
        self.hidden_channels, self.key_channels, self.num_heads = hidden_channels, key_channels, num_heads
        self.W_Q = Linear(hidden_channels, key_channels, device=device)
        self.W_K = Linear(hidden_channels, key_channels, device=device)
        self.W_V = Linear(hidden_channels, hidden_channels, device=device)
        self.linear = Linear(hidden_channels, hidden_channels, device=device)
        self.layer_norm = LayerNorm(hidden_channels, device=device)

    def forward(self, Q, K, V, attn_mask):
        return _dense_attention(self, Q, K, V, attn_mask, self.key_channels, self.hidden_channels)


class PoswiseFeedForwardNet(nn.Module):
    """1x1 Conv1d pair + residual LayerNorm on node rows (CP:161-176); batch_norm exists but is never called (Q10)."""

    def __init__(self, hidden_channels, device="cuda"):
        super().__init__()
        self.conv1 = Conv1d(hidden_channels, 1024, 1, device=device)
        self.conv2 = Conv1d(1024, hidden_channels, 1, device=device)
        self.layer_norm = LayerNorm(hidden_channels, device=device)
        self.batch_norm = BatchNorm1d(hidden_channels, device=device)

    def forward(self, inputs):
        h = ops.pos_ffn(inputs, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias)
        return ops.layer_norm_residual(h, inputs, self.layer_norm)


class PoswiseFeedForwardDeNet(nn.Module):
    def __init__(self, hidden_channels, device="cuda"):
        super().__init__()
        self.conv1 = Conv1d(hidden_channels, 1024, 1, device=device)
        self.conv2 = Conv1d(1024, hidden_channels, 1, device=device)
        self.layer_norm = LayerNorm(hidden_channels, device=device)

    def forward(self, inputs):
        h = ops.pos_ffn(inputs, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias)
        return ops.layer_norm_residual(h, inputs, self.layer_norm)


class PositionalEncoding(nn.Module):
    def __init__(self, d_model, dropout=0.1, max_len=5000, device="cuda"):
        super().__init__()
        self.dropout = Dropout(p=dropout)
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0).transpose(0, 1).to(device))

    def forward(self, x):
        """x: [seq_len, batch, d_model]"""
        return self.dropout(x + self.pe[: x.size(0), :])


# ----------------------------------------------------------------------------------------------- encoder / decoder
class EncoderLayer(nn.Module):
    def __init__(self, config, device="cuda"):
        super().__init__()
        self.enc_self_attn = MultiHeadAttention(config.hidden_channels, config.edge_channels, config.key_channels,
                                                config.num_heads, device=device)
        self.pos_ffn = PoswiseFeedForwardNet(config.hidden_channels, device=device)

    def forward(self, node_attr, edges):
        msa_outputs = self.enc_self_attn(node_attr, edges)
        return msa_outputs, self.pos_ffn(msa_outputs)


class EncoderLayer2(nn.Module):
    def __init__(self, config, device="cuda"):
        super().__init__()
        self.enc_self_attn = MultiHeadAttention(config.hidden_channels, config.edge_channels, config.key_channels,
                                                config.num_heads, device=device)
        self.proj = Linear(config.hidden_channels, config.hidden_channels, device=device)
        self.cross_attn = MultiHeadAttention2(config.hidden_channels, config.key_channels, config.num_heads, device=device)
        self.layer_norm = LayerNorm(config.hidden_channels, device=device)
        self.pos_ffn = PoswiseFeedForwardNet(config.hidden_channels, device=device)

    def forward(self, node_attr, edges, idx, atom_msa_outputs, atom_mask, dm, before_cross=None):
        msa_outputs = self.enc_self_attn(node_attr, edges)
        if idx == 2 or idx == 5:                                                   # CP:262
            if before_cross is not None:
                before_cross(idx)
            kv = self.proj(atom_msa_outputs[idx])
            cross = self.cross_attn(dm.dense(msa_outputs), kv, kv, atom_mask)
            msa_outputs = ops.layer_norm_residual(msa_outputs, dm.gather(cross.reshape(-1, cross.size(-1))), self.layer_norm)
        return self.pos_ffn(msa_outputs)


class Encoder(nn.Module):
    def __init__(self, config, protein_atom_feature_dim, device="cuda"):
        super().__init__()
        self.config = config
        self.protein_atom_emb = Linear(protein_atom_feature_dim, config.hidden_channels, device=device)
        self.laplacian_emb = Linear(config.lap_dim, config.hidden_channels, device=device)
        self.layers = nn.ModuleList([EncoderLayer(config, device=device) for _ in range(config.num_interactions)])
        self.distance_expansion = GaussianSmearing(stop=15, num_gaussians=config.edge_channels, device=device)
        self.out = Linear(config.hidden_channels, config.hidden_channels, device=device)       # unused (Q10)
        self.layer_norm = LayerNorm(config.hidden_channels, device=device)                      # unused (Q10)

    def prepare(self, pos, batch, batch_size, knn=None, mx=None, cap=None, n_real=None):
        """Per-batch graph structure (no parameters involved): dense-batch map, kNN edges (CP:293-298).  mx / cap /
        n_real: fixed layout width and edge capacity for padded batches (graph.pad_batch)."""
        dm = DenseMap(batch, batch_size, mx)
        if knn is None:
            knn = knn_graph(pos, self.config.knn, batch, batch_size, dm)
        return {"dense": dm, "edges": KnnEdges(pos, knn, self.distance_expansion, cap, n_real)}

    def forward(self, protein_atom_feature, pos, batch, atom_laplacian, batch_size=None, knn=None, prep=None,
                layer_done=None):
        """`layer_done(idx, dense_msa)`, if given, is called right after layer idx produced its attention output (the
        ligand encoder, running on a second stream, waits on these points - see Transformer.forward)."""
        if prep is None:
            prep = self.prepare(pos, batch, int(batch.max()) + 1 if batch_size is None else batch_size, knn)
        dm, edges = prep["dense"], prep["edges"]
        # protein_atom_emb(x) + laplacian_emb(pe) (CP:300): the sum rides in the second GEMM's epilogue
        node_attr = ops.linear_add(protein_atom_feature, self.protein_atom_emb.weight, self.protein_atom_emb.bias,
                                   self.laplacian_emb(atom_laplacian))
        msa_outputs1 = []
        for idx, layer in enumerate(self.layers):
            msa_outputs, node_attr = layer(node_attr, edges)
            # the ligand encoder's cross attention reads the protein attention outputs of layers 2 and 5 only (CP:262): the
            # other layers' dense copies (a zero fill + an index copy each) are never looked at
            msa_outputs1.append(dm.dense(msa_outputs) if idx in (2, 5) else None)
            if layer_done is not None and msa_outputs1[-1] is not None:
                layer_done(idx, msa_outputs1[-1])
        return dm.dense(node_attr), dm.pad_mask, msa_outputs1


class Encoder2(nn.Module):
    def __init__(self, config, aa_feature_dim, device="cuda"):
        super().__init__()
        self.config = config
        self.aa_emb = Linear(aa_feature_dim, config.hidden_channels, device=device)
        self.laplacian_emb = Linear(config.lap_dim, config.hidden_channels, device=device)
        self.layers = nn.ModuleList([EncoderLayer2(config, device=device) for _ in range(config.num_interactions)])
        self.distance_expansion = GaussianSmearing(stop=25, num_gaussians=config.edge_channels, device=device)
        self.out = Linear(config.hidden_channels, config.hidden_channels, device=device)       # unused (Q10)
        self.layer_norm = LayerNorm(config.hidden_channels, device=device)                      # unused (Q10)

    def prepare(self, aa_pos, aa_batch, batch_size, knn=None, mx=None, cap=None, n_real=None):
        dm = DenseMap(aa_batch, batch_size, mx)
        if knn is None:
            knn = knn_graph(aa_pos, 30, aa_batch, batch_size, dm)                  # CP:330
        return {"dense": dm, "edges": KnnEdges(aa_pos, knn, self.distance_expansion, cap, n_real)}

    def forward(self, aa_feature, aa_pos, aa_batch, aa_laplacian, atom_mask, atom_msa_outputs, batch_size=None, knn=None,
                prep=None, before_cross=None):
        if prep is None:
            prep = self.prepare(aa_pos, aa_batch, int(aa_batch.max()) + 1 if batch_size is None else batch_size, knn)
        dm, edges = prep["dense"], prep["edges"]
        node_attr = ops.linear_add(aa_feature, self.aa_emb.weight, self.aa_emb.bias, self.laplacian_emb(aa_laplacian))
        for idx, layer in enumerate(self.layers):
            node_attr = layer(node_attr, edges, idx, atom_msa_outputs, atom_mask, dm, before_cross)
        return dm.dense(node_attr), dm.pad_mask


class DecoderLayer(nn.Module):
    def __init__(self, config, device="cuda"):
        super().__init__()
        self.dec_self_attn = MultiHeadDeAttention(config.hidden_channels, config.key_channels, config.num_heads, device=device)
        self.dec_enc_attn = MultiHeadDeAttention(config.hidden_channels, config.key_channels, config.num_heads, device=device)
        self.pos_ffn = PoswiseFeedForwardDeNet(config.hidden_channels, device=device)

    def forward(self, dec_inputs, enc_outputs, dec_self_attn_mask, dec_enc_attn_mask):
        dec_outputs = self.dec_self_attn(dec_inputs, dec_inputs, dec_inputs, dec_self_attn_mask)
        dec_outputs = self.dec_enc_attn(dec_outputs, enc_outputs, enc_outputs, dec_enc_attn_mask)
        return self.pos_ffn(dec_outputs)


class Decoder(nn.Module):
    def __init__(self, config, num_props=None, device="cuda"):
        super().__init__()
        self.config, self.device, self.num_props = config, device, num_props
        self.mol_emb = Embedding(len(config.smiVoc), config.hidden_channels, 0, device=device)
        self.pos_emb = PositionalEncoding(config.hidden_channels, device=device)
        self.type_emb = Embedding(2, config.hidden_channels, device=device)
        if self.num_props:
            self.prop_nn = Linear(self.num_props, config.hidden_channels, device=device)
        self.layers = nn.ModuleList([DecoderLayer(config, device=device) for _ in range(config.num_interactions)])
        self.pad_id = list(config.smiVoc).index("^")

    def forward(self, smiles_index, enc_outputs, enc_pad_mask, tgt_len, prop=None):
        b, t = smiles_index.size()
        dev = smiles_index.device
        # nn.Embedding(.., padding_idx=0) of the reference (CP:377, Q9): plain lookup, row 0 gets no gradient
        dec_inputs = ops.embedding(self.mol_emb.weight, smiles_index, self.mol_emb.padding_idx)
        dec_inputs = self.pos_emb(dec_inputs.transpose(0, 1)).transpose(0, 1)
        ids = smiles_index
        num = 0
        if self.num_props:
            assert prop.shape[-1] == self.num_props
            dec_inputs = ops.bias_add(dec_inputs, self.type_emb.weight[1])
            p = ops.bias_add(self.prop_nn(prop.unsqueeze(1)), self.type_emb.weight[0])
            dec_inputs = torch.cat([p, dec_inputs], 1)
            ids = torch.cat([torch.ones(b, 1, dtype=ids.dtype, device=dev), ids], 1)    # property token: id 1, never pad (Q9)
            num = 1
        n = t + num
        pad = ids.eq(self.pad_id).unsqueeze(1).expand(b, n, n)
        causal = torch.triu(torch.ones(n, n, dtype=torch.bool, device=dev), diagonal=1)
        dec_self_attn_mask = pad | causal.unsqueeze(0)
        dec_enc_attn_mask = enc_pad_mask.expand(b, tgt_len + num, enc_pad_mask.size(2))
        for layer in self.layers:
            dec_inputs = layer(dec_inputs, enc_outputs, dec_self_attn_mask, dec_enc_attn_mask)
        return dec_inputs


class Transformer(nn.Module):
    def __init__(self, config, protein_atom_feature_dim, num_props=None, device="cuda"):
        super().__init__()
        self.config, self.num_props, self.device = config, num_props, device
        self.encoder = Encoder(config.encoder, protein_atom_feature_dim, device=device)
        self.encoder2 = Encoder2(config.encoder, protein_atom_feature_dim, device=device)
        self.decoder = Decoder(config.decoder, self.num_props, device=device)
        self.projection = Linear(config.hidden_channels, len(config.decoder.smiVoc), bias=False, device=device)

    overlap_encoders = os.environ.get("SINGA_OVERLAP_ENCODERS", "1") == "1"      # (0: one stream, lab)

    def _encoders_two_streams(self, node_attr, pos, batch, atom_laplacian, aa_node_attr, aa_pos, aa_batch, aa_laplacian,
                              B, knn, aa_knn, prep):
        """The ligand encoder (a few hundred launches on <= 10^3 nodes: latency-bound) runs on a second HIP stream
        beside the protein encoder; it only needs the protein attention outputs of layers 2 and 5 (CP:262), which it
        waits for through events.  Autograd replays the same split in the backward pass, and a HIP-graph capture records
        the two streams as parallel branches.  Same arithmetic, same order within each encoder."""
        cur = torch.cuda.current_stream()
        aux = ops.branch_stream(node_attr.device)
        if prep.get("p") is None:
            prep = dict(prep, p=self.encoder.prepare(pos, batch, B, knn))
        if prep.get("l") is None:
            prep = dict(prep, l=self.encoder2.prepare(aa_pos, aa_batch, B, aa_knn))
        start = torch.cuda.Event()
        start.record(cur)
        ready = {}

        def layer_done(idx, dense_msa):
            if idx in (2, 5):
                dense_msa.record_stream(aux)
                ready[idx] = torch.cuda.Event()
                ready[idx].record(cur)

        enc_outputs1, enc_pad_mask1, msa_outputs = self.encoder(node_attr, pos, batch, atom_laplacian, B, knn, prep["p"],
                                                                layer_done)
        for t in (aa_node_attr, aa_laplacian):
            t.record_stream(aux)
        with torch.cuda.stream(aux):
            aux.wait_event(start)
            enc_outputs2, enc_pad_mask2 = self.encoder2(aa_node_attr, aa_pos, aa_batch, aa_laplacian, enc_pad_mask1,
                                                        msa_outputs, B, aa_knn, prep["l"],
                                                        lambda idx: aux.wait_event(ready[idx]))
        cur.wait_stream(aux)
        enc_outputs2.record_stream(cur)
        return enc_outputs1, enc_pad_mask1, enc_outputs2, enc_pad_mask2

    def forward(self, node_attr, pos, batch, atom_laplacian, smiles_index, tgt_len, aa_node_attr, aa_pos, aa_batch,
                aa_laplacian, prop=None, knn=None, aa_knn=None, prep=None):
        B = smiles_index.shape[0]
        prep = prep or {}
        if not self.overlap_encoders:
            enc_outputs1, enc_pad_mask1, msa_outputs = self.encoder(node_attr, pos, batch, atom_laplacian, B, knn,
                                                                    prep.get("p"))
            enc_outputs2, enc_pad_mask2 = self.encoder2(aa_node_attr, aa_pos, aa_batch, aa_laplacian, enc_pad_mask1,
                                                        msa_outputs, B, aa_knn, prep.get("l"))
        else:
            enc_outputs1, enc_pad_mask1, enc_outputs2, enc_pad_mask2 = self._encoders_two_streams(
                node_attr, pos, batch, atom_laplacian, aa_node_attr, aa_pos, aa_batch, aa_laplacian, B, knn, aa_knn, prep)
        enc_outputs = torch.cat([enc_outputs1, enc_outputs2], dim=1)
        enc_pad_mask = torch.cat([enc_pad_mask1, enc_pad_mask2], dim=2)
        dec_outputs = self.decoder(smiles_index, enc_outputs, enc_pad_mask, tgt_len, prop)
        dec_logits = self.projection(dec_outputs)
        num = 1 if self.num_props else 0
        dec_logits = dec_logits[:, num:, :]
        return dec_logits.reshape(-1, dec_logits.size(-1))
