"""CPU restatement of the reference's beam search (reference model/BeamSearch.py:7-175, driven as in gen.py:156-196).

TEST INFRASTRUCTURE - only tests/ may import this.  Pinned by tests/golden/beam_*.npz, which oracle/make_golden_beam.py
produced by running the reference's own `beam_search` (tests/test_oracle_golden.py).

Like the reference, the decoder is re-run on the whole growing prefix at every step (no KV cache) and all
bookkeeping is host-side.  The behaviours that decide which hypotheses survive are kept as they are:
  * encoder 1 only; the property prompt occupies decoder position 0                       (BS:64-76, CP:404-412)
  * 2*num_beams candidates per sentence, ranked over beams x vocabulary                    (BS:86-90)
  * an end-of-sequence candidate is stored as a hypothesis only when ranked < num_beams; the stored sequence does not
    contain the '$' itself and its score is sum_logprobs / len**0.7                        (BS:107-114, 18-19)
  * `done` is re-evaluated after every candidate EXCEPT the one that fills the beam        (BS:118-123: the `break`
    precedes the update)
  * a finished sentence feeds (score 0, pad, beam 0) to the next step, i.e. row 0 of the WHOLE batch (BS:96,134)
  * unfinished sentences flush their live beams at max_length                              (BS:141-149)
  * output: stacked if all best hypotheses have one length, else padded with '^' and terminated with '$' (BS:164-173)
"""
import torch
import torch.nn.functional as F

from oracle import singa_oracle as SO

LENGTH_PENALTY = 0.7                                   # BS:58


class Hypotheses:
    """BeamHypotheses, BS:7-35: the num_beams best finished sequences of one sentence."""

    def __init__(self, num_beams):
        self.n, self.items, self.worst = num_beams, [], 1e9

    def add(self, seq, sum_logprobs):
        score = sum_logprobs / len(seq) ** LENGTH_PENALTY
        if len(self.items) < self.n or score > self.worst:
            self.items.append((score, seq))
            if len(self.items) > self.n:
                ranked = sorted((s, i) for i, (s, _) in enumerate(self.items))
                del self.items[ranked[0][1]]
                self.worst = ranked[1][0]
            else:
                self.worst = min(score, self.worst)

    def is_done(self, best_sum_logprobs, cur_len):
        return len(self.items) >= self.n and self.worst >= best_sum_logprobs / cur_len ** LENGTH_PENALTY


def beam_search(sd, voc, num_beams, batch_size, max_length, topk, feat, pos, batch, lap, knn, prop, pre="model.",
                trace=None):
    """Returns the decoded token matrix [batch_size*topk, T] (int64), as BS:38-175.  `sd`: SINGA state dict as torch
    tensors; `prop` [batch_size*num_beams, 3].  `trace`, if a dict, receives 'first_logp', 'hyps', 'last_beams'."""
    V = len(voc)
    sos, eos, pad_id = voc.index("&"), voc.index("$"), voc.index("^")
    sd = dict(sd)
    sd.setdefault(pre + "decoder.pos_emb.pe", SO.positional_table(sd[pre + "decoder.mol_emb.weight"].shape[1]))
    enc, pad, _ = SO.encoder1_forward(sd, pre, feat, pos, batch, lap, knn, batch_size)
    enc = enc.repeat_interleave(num_beams, 0)
    pad = pad.repeat_interleave(num_beams, 0)
    R = batch_size * num_beams
    scores = torch.zeros(batch_size, num_beams)
    scores[:, 1:] = -1e9
    scores = scores.view(-1)
    ids = torch.full((R, 1), sos, dtype=torch.long)
    done = [False] * batch_size
    hyps = [Hypotheses(num_beams) for _ in range(batch_size)]
    cur_len = 1
    while cur_len < max_length:
        logits = SO.decoder_forward(sd, pre, ids, prop, enc, pad, pad_id)[:, -1]
        logp = F.log_softmax(logits, dim=-1)
        if trace is not None and "first_logp" not in trace:
            trace["first_logp"] = logp.clone()
        cand = (logp + scores[:, None]).view(batch_size, num_beams * V)
        c_score, c_tok = torch.topk(cand, 2 * num_beams, dim=1, largest=True, sorted=True)
        c_score_h, c_tok_h = c_score.numpy(), c_tok.numpy()
        nxt = []                                        # (score, token, source row) per live beam
        for b in range(batch_size):
            if done[b]:
                nxt += [(0.0, pad_id, 0)] * num_beams
                continue
            mine = []
            for rank in range(2 * num_beams):
                sc, flat = float(c_score_h[b, rank]), int(c_tok_h[b, rank])
                row, tok = b * num_beams + flat // V, flat % V
                if tok == eos:
                    if rank >= num_beams:
                        continue
                    hyps[b].add(ids[row].clone(), sc)
                else:
                    mine.append((sc, tok, row))
                if len(mine) == num_beams:
                    break
                done[b] = done[b] or hyps[b].is_done(float(c_score_h[b].max()), cur_len)
            nxt += mine
        if all(done):
            break
        scores = torch.tensor([x[0] for x in nxt], dtype=torch.float32)
        src = torch.tensor([x[2] for x in nxt], dtype=torch.long)
        ids = torch.cat([ids[src], torch.tensor([x[1] for x in nxt], dtype=torch.long)[:, None]], 1)
        enc, pad = enc[src], pad[src]
        cur_len += 1
        if trace is not None:
            trace["last_beams"] = ids.clone()
    for b in range(batch_size):
        if not done[b]:
            for k in range(num_beams):
                hyps[b].add(ids[b * num_beams + k], float(scores[b * num_beams + k]))
    best = []
    for h in hyps:
        ranked = sorted(h.items, key=lambda x: x[0])
        best += [ranked.pop()[1] for _ in range(topk)]
    lens = [len(x) for x in best]
    if trace is not None:
        trace["hyps"] = hyps
    if min(lens) == max(lens):
        return torch.stack(best).long()
    out = torch.full((len(best), min(max(lens) + 1, max_length)), pad_id, dtype=torch.long)
    for i, x in enumerate(best):
        out[i, :lens[i]] = x
        if lens[i] < max_length:
            out[i, lens[i]] = eos
    return out
