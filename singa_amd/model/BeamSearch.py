"""Beam-search generation for MI355X (reference model/BeamSearch.py = "BS", driven by gen.py:156-196).

Same entry point and arguments as the reference - `beam_search(model, smiVoc, num_beams, batch_size, max_length, topk,
example, prop, device)` returning the decoded token matrix - and the same selection rules, so the same hypotheses
survive (see `_select` for the rules that are easy to get subtly different).  What is different is where the time goes:

  * the reference re-runs the whole decoder on the growing prefix for every new token (BS:82, O(T^2) positions);
    here each decoder layer keeps its self-attention keys / values per live beam (`KVDecoder`), so a step evaluates ONE
    position per beam, and the caches follow the beams when they are re-ranked;
  * the reference copies the encoder output once per beam (BS:78-79, 135-136) and recomputes its key / value
    projections in every layer at every step; here they are projected once per protein, and the beams of a protein
    form the query rows of one batched matmul against them - nothing is replicated or re-gathered;
  * one device->host transfer per step (the 2*num_beams ranked candidates of every protein) instead of an `.item()`
    per candidate (BS:107-122).

The device work of a step is GEMMs (library calls through ops.linear), softmax / layer norm / top-k (library ops) on
[rows, 256] operands; hypothesis bookkeeping stays on the host, as in the reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


class BeamHypotheses(object):
    """The `num_beams` best finished sequences of one protein (BS:7-35)."""

    def __init__(self, num_beams, max_length, length_penalty):
        self.max_length = max_length - 1
        self.length_penalty = length_penalty
        self.num_beams = num_beams
        self.beams = []
        self.worst_score = 1e9

    def __len__(self):
        return len(self.beams)

    def add(self, hyp, sum_logprobs):
        score = sum_logprobs / len(hyp) ** self.length_penalty
        if len(self) >= self.num_beams and score <= self.worst_score:
            return
        self.beams.append((score, hyp))
        if len(self) > self.num_beams:
            order = sorted((s, i) for i, (s, _) in enumerate(self.beams))
            del self.beams[order[0][1]]
            self.worst_score = order[1][0]
        else:
            self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs, cur_len):
        if len(self) < self.num_beams:
            return False
        return self.worst_score >= best_sum_logprobs / cur_len ** self.length_penalty


class KVDecoder:
    """Incremental evaluation of `Decoder` (CP:385-423): one new position per row and call, self-attention keys / values
    cached per layer, encoder keys / values projected once per protein.

    rows = proteins x beams, protein-major (row r belongs to protein r // beams, as BS:105).  Causality makes the result
    for the newest position identical to the reference's full re-run on the prefix; the padding mask of the self-
    attention (CP:414) never fires for a live beam because a live prefix contains no '^'."""

    def __init__(self, decoder, enc_outputs, enc_pad_mask, beams, max_positions):
        self.dec, self.beams = decoder, beams
        B, S, _ = enc_outputs.shape
        a0 = decoder.layers[0].dec_self_attn
        self.heads = a0.num_heads
        self.dk, self.dv = a0.key_channels // self.heads, a0.hidden_channels // self.heads
        self.B, self.R = B, B * beams
        dev = enc_outputs.device
        self.cross_k, self.cross_v = [], []
        for layer in decoder.layers:
            c = layer.dec_enc_attn
            self.cross_k.append(c.W_K(enc_outputs).view(B, S, self.heads, self.dk).permute(0, 2, 3, 1).contiguous())
            self.cross_v.append(c.W_V(enc_outputs).view(B, S, self.heads, self.dv).transpose(1, 2).contiguous())
        self.cross_mask = enc_pad_mask.view(B, 1, 1, S)
        n = len(decoder.layers)
        self.k = torch.zeros(n, self.R, self.heads, max_positions, self.dk, device=dev)
        self.v = torch.zeros(n, self.R, self.heads, max_positions, self.dv, device=dev)
        self.t = 0

    def follow(self, src_rows):
        """Row r continues the prefix that row src_rows[r] held (BS:134-136)."""
        t = self.t
        self.k[:, :, :, :t] = self.k[:, :, :, :t].index_select(1, src_rows)
        self.v[:, :, :, :t] = self.v[:, :, :, :t].index_select(1, src_rows)

    def token_input(self, tokens, position):
        d = self.dec
        x = d.mol_emb.weight.index_select(0, tokens) + d.pos_emb.pe[position, 0]
        return x + d.type_emb.weight[1] if d.num_props else x

    def prop_input(self, prop):
        d = self.dec
        return d.prop_nn(prop) + d.type_emb.weight[0]

    def advance(self, x):
        """x [rows, hidden]: decoder input at the next position -> decoder output at that position."""
        R, B, H, t = self.R, self.B, self.heads, self.t
        for l, layer in enumerate(self.dec.layers):
            a = layer.dec_self_attn
            self.k[l, :, :, t] = a.W_K(x).view(R, H, self.dk)
            self.v[l, :, :, t] = a.W_V(x).view(R, H, self.dv)
            q = a.W_Q(x).view(R, H, 1, self.dk)
            s = torch.matmul(q, self.k[l, :, :, :t + 1].transpose(-1, -2)) / math.sqrt(self.dk)
            ctx = torch.matmul(torch.softmax(s, dim=-1), self.v[l, :, :, :t + 1]).reshape(R, H * self.dv)
            y = a.layer_norm(a.linear(ctx) + x)
            c = layer.dec_enc_attn
            q = c.W_Q(y).view(B, self.beams, H, self.dk).transpose(1, 2)                   # beams = query rows
            s = (torch.matmul(q, self.cross_k[l]) / math.sqrt(self.dk)).masked_fill(self.cross_mask, -1e9)
            ctx = torch.matmul(torch.softmax(s, dim=-1), self.cross_v[l]).transpose(1, 2).reshape(R, H * self.dv)
            x = layer.pos_ffn(c.layer_norm(c.linear(ctx) + y))
        self.t = t + 1
        return x


def _select(cand_score, cand_flat, prefixes, hyps, done, num_beams, vocab_size, eos, pad, cur_len):
    """Host side of one step (BS:92-125) for all proteins: from the 2*num_beams ranked candidates of each protein pick the
    next live beams and store finished sequences.  Returns (scores, tokens, source rows), one entry per row.

    Rules kept exactly: '$' candidates are stored only when ranked inside the first num_beams, without the '$' and
    scored by the candidate's summed log-probability; `done` is re-evaluated after every candidate except the one that
    fills the beam (the reference's `break` comes first); a finished protein emits (0, '^', row 0)."""
    scores, tokens, rows = [], [], []
    for b in range(len(done)):
        if done[b]:
            scores += [0.0] * num_beams
            tokens += [pad] * num_beams
            rows += [0] * num_beams
            continue
        kept = 0
        best = float(cand_score[b].max())
        for rank in range(2 * num_beams):
            flat = int(cand_flat[b, rank])
            row, tok = b * num_beams + flat // vocab_size, flat % vocab_size
            if tok == eos:
                if rank >= num_beams:
                    continue
                hyps[b].add(prefixes[row].copy(), float(cand_score[b, rank]))
            else:
                scores.append(cand_score[b, rank])
                tokens.append(tok)
                rows.append(row)
                kept += 1
            if kept == num_beams:
                break
            done[b] = done[b] or hyps[b].is_done(best, cur_len)
        assert kept == num_beams, "fewer than num_beams live continuations among 2*num_beams candidates"
    return np.asarray(scores, dtype=np.float32), np.asarray(tokens, dtype=np.int64), np.asarray(rows, dtype=np.int64)


@torch.no_grad()
def beam_search(model, smiVoc, num_beams, batch_size, max_length, topk, example, prop=None, device="cuda", trace=None):
    """BS:38-175.  `model`: SINGA (uses model.model.encoder / decoder / projection); `example`: attribute bag with
    protein_element_batch, protein_atom_feature, protein_pos, protein_atom_laplacian (gen.py:176-181) and, optionally,
    protein_knn (a precomputed [2,E] kNN list; otherwise drawn on the GPU); `prop` [batch_size*num_beams, num_props].
    Returns the decoded int64 token matrix [batch_size*topk, T] on `device`."""
    tf = model.model
    vocab_size = len(smiVoc)
    voc = list(smiVoc)
    sos, eos, pad = voc.index("&"), voc.index("$"), voc.index("^")
    dev = torch.device(device)
    feat = example.protein_atom_feature.float()
    enc_outputs, enc_pad_mask, _ = tf.encoder(feat, example.protein_pos, example.protein_element_batch,
                                              example.protein_atom_laplacian, batch_size,
                                              getattr(example, "protein_knn", None))
    rows = batch_size * num_beams
    num = 1 if tf.decoder.num_props else 0
    kv = KVDecoder(tf.decoder, enc_outputs, enc_pad_mask, num_beams, max_length + num)
    if num:
        kv.advance(kv.prop_input(prop.to(dev).float()))                # position 0 is the property prompt, CP:404-412

    beam_scores = torch.zeros(batch_size, num_beams, device=dev)
    beam_scores[:, 1:] = -1e9                                          # all beams start equal: only beam 0 counts at step 1
    beam_scores = beam_scores.view(-1)
    prefixes = np.full((rows, 1), sos, dtype=np.int64)                 # host copy of input_ids: bookkeeping only
    tokens = torch.full((rows,), sos, dtype=torch.long, device=dev)
    done = [False] * batch_size
    hyps = [BeamHypotheses(num_beams, max_length, length_penalty=0.7) for _ in range(batch_size)]
    cur_len = 1
    while cur_len < max_length:
        out = kv.advance(kv.token_input(tokens, cur_len - 1))
        logp = F.log_softmax(tf.projection(out), dim=-1)
        if trace is not None and "first_logp" not in trace:
            trace["first_logp"] = logp.clone()
        cand = (logp + beam_scores[:, None]).view(batch_size, num_beams * vocab_size)
        cand_score, cand_flat = torch.topk(cand, 2 * num_beams, dim=1, largest=True, sorted=True)
        packed = torch.stack([cand_score.double(), cand_flat.double()]).cpu().numpy()      # the step's one D2H copy
        sc, tk, src = _select(packed[0].astype(np.float32), packed[1].astype(np.int64), prefixes, hyps, done,
                              num_beams, vocab_size, eos, pad, cur_len)
        if all(done):
            break
        step = torch.from_numpy(np.stack([sc.astype(np.float64), tk.astype(np.float64), src.astype(np.float64)])).to(dev)
        beam_scores, tokens, src_rows = step[0].float(), step[1].long(), step[2].long()
        kv.follow(src_rows)
        prefixes = np.concatenate([prefixes[src], tk[:, None]], axis=1)
        cur_len += 1
    if trace is not None:
        trace["last_beams"], trace["hyps"] = prefixes.copy(), hyps

    final = beam_scores.cpu().numpy()
    for b in range(batch_size):
        if not done[b]:                                                # BS:141-149
            for k in range(num_beams):
                r = b * num_beams + k
                hyps[b].add(prefixes[r], float(final[r]))
    best = []
    for h in hyps:
        ranked = sorted(h.beams, key=lambda x: x[0])
        best += [ranked.pop()[1] for _ in range(topk)]
    lens = [len(x) for x in best]
    if min(lens) == max(lens):                                         # BS:164-173
        decoded = np.stack(best)
    else:
        decoded = np.full((len(best), min(max(lens) + 1, max_length)), pad, dtype=np.int64)
        for i, x in enumerate(best):
            decoded[i, :lens[i]] = x
            if lens[i] < max_length:
                decoded[i, lens[i]] = eos
    return torch.from_numpy(decoded).to(dev)
