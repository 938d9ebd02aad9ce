#!/usr/bin/env python
"""Generation throughput of the KV-cached beam search in gen.py's configuration: one protein pocket,
num_beams=20, topk=1, max_length = tgt_len + 1 = 201, property prompt - on synthetic pockets with random-init weights.

    python tools/bench_beam.py [--proteins 4] [--max-length 201] [--beams 20] [--no-graph] [--prefix-baseline]

Prints one JSON line: new tokens per second (beams x steps / time), pockets per second, and the per-step time.
`--prefix-baseline` also times the reference's schedule (the whole decoder re-run on the growing prefix for every
token, BeamSearch.py:82) built from the same product modules on the same GPU, so the two differ by the algorithm only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@torch.no_grad()
def prefix_rerun_search(model, voc, num_beams, max_length, ex, prop):
    """The reference's schedule on the product modules: greedy bookkeeping is irrelevant for timing, so every step keeps the
    top num_beams non-eos candidates (same tensor shapes and the same number of decoder evaluations as BS:81-139)."""
    tf = model.model
    V = len(voc)
    eos = voc.index("$")
    enc, mask, _ = tf.encoder(ex.protein_atom_feature, ex.protein_pos, ex.protein_element_batch, ex.protein_atom_laplacian, 1,
                              ex.protein_knn)
    enc, mask = enc.repeat_interleave(num_beams, 0), mask.repeat_interleave(num_beams, 0)
    ids = torch.full((num_beams, 1), voc.index("&"), dtype=torch.long, device=enc.device)
    scores = torch.zeros(num_beams, device=enc.device)
    scores[1:] = -1e9
    for cur_len in range(1, max_length):
        logits = tf.projection(tf.decoder(ids, enc, mask, cur_len, prop))[:, -1]
        logp = torch.log_softmax(logits, -1)
        logp[:, eos] = -1e9
        sc, flat = torch.topk((logp + scores[:, None]).view(-1), num_beams)
        flat = flat.cpu()
        src, tok = (flat // V).to(enc.device), (flat % V).to(enc.device)
        ids = torch.cat([ids[src], tok[:, None]], 1)
        enc, mask, scores = enc[src], mask[src], sc
    return ids


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--proteins", type=int, default=4)
    ap.add_argument("--beams", type=int, default=20)
    ap.add_argument("--max-length", type=int, default=201)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--prefix-baseline", action="store_true")
    args = ap.parse_args()
    import __graft_entry__
    __graft_entry__.build()
    from singa_amd import graph as G
    from singa_amd.config import Config, load_config
    from singa_amd.model.BeamSearch import beam_search
    from singa_amd.model.CProMG import DenseMap, knn_graph
    from singa_amd.model.GAN import SINGA
    dev = torch.device("cuda", 0)
    cfg = load_config(lmax=2)
    torch.manual_seed(cfg.train.seed)
    model = SINGA(cfg, device=dev).eval()
    voc = list(cfg.model.decoder.smiVoc)
    with torch.no_grad():
        # random-init weights: keep '$' from ending the search early, so that every run decodes max_length tokens
        # (channel 0 of the last layer norm gets a +10 offset and is the only input of the '$' logit, weight -3)
        model.model.decoder.layers[-1].pos_ffn.layer_norm.bias[0] = 10.0
        model.model.projection.weight[voc.index("$")] = 0.0
        model.model.projection.weight[voc.index("$"), 0] = -3.0
    prop = torch.ones(args.beams, 3, device=dev)

    def pocket(i):
        b = G.collate([G.synthetic_graph(500 + i)]).to(dev)
        model.prepare(b)
        with torch.no_grad():
            feat = model.embedding(b, gen_mode=True)[G.PA].embedding.reshape(b[G.PA]["x"].shape[0], -1)
        ex = Config()
        batch = b[G.PA]["batch"]
        ex.protein_element_batch, ex.protein_atom_feature, ex.protein_pos = batch, feat, b[G.PA]["pos"]
        ex.protein_atom_laplacian = b[G.PA]["lap_pe"]
        knn = knn_graph(b[G.PA]["pos"], cfg.model.encoder.knn, batch, 1, DenseMap(batch, 1))
        ex.protein_knn = knn[:, knn[0] >= 0]
        return ex

    pockets = [pocket(i) for i in range(args.proteins)]
    run = lambda ex: beam_search(model, voc, args.beams, 1, args.max_length, 1, ex, prop, device=dev, graph=not args.no_graph)
    out = run(pockets[0])                                               # warm-up (library initialisation)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lens = []
    for ex in pockets:
        lens.append(run(ex).shape[1])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = sum(n - 1 for n in lens)
    res = {"metric": "beam_search_new_tokens_per_s", "value": round(steps * args.beams / dt, 1), "unit": "tokens/s",
           "pockets_per_s": round(args.proteins / dt, 3), "ms_per_step": round(dt / steps * 1e3, 4),
           "config": {"workload": "gen.py: 1 pocket (200 atoms), property prompt", "num_beams": args.beams,
                      "max_length": args.max_length, "decoded_lengths": lens,
                      "launch": "eager" if args.no_graph else "hipGraph replay per step"}}
    if args.prefix_baseline:
        prefix_rerun_search(model, voc, args.beams, min(args.max_length, 20), pockets[0], prop)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ids = prefix_rerun_search(model, voc, args.beams, args.max_length, pockets[0], prop)
        torch.cuda.synchronize()
        dtp = time.perf_counter() - t0
        res["prefix_rerun"] = {"value": round((ids.shape[1] - 1) * args.beams / dtp, 1), "unit": "tokens/s",
                               "what": "decoder re-run on the growing prefix per token (the reference's schedule), same GPU modules"}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
