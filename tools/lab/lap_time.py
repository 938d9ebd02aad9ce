"""Lab: graph.laplacian_pe_batched (n2, singa_lap_pe) on config-5 and config-3 batches: ms per call with the sparse route
(Chebyshev-filtered subspace iteration for components >= 384 atoms) and with it switched off (dense Householder route)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import graph as G, _lib

for wl, nb in (("cfg5_l6", 64), ("cfg3_b128_l4", 128)):
    L, kw, ids, _ = G.resolve_workload(wl)
    b = G.synthetic_batch(nb, ids=ids[:nb], with_lap=False, **kw).to("cuda")
    for fsi_min in (384, 2000):
        _lib.lib().singa_lap_pe_fsi_min(fsi_min)
        outs = []
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pe = [G.laplacian_pe_batched(b[et]["edge_index"], b[nt]["batch"], b.num_graphs) for nt, et in ((G.PA, G.E_PP), (G.LA, G.E_LL))]
            torch.cuda.synchronize()
            outs.append((time.perf_counter() - t0) * 1e3)
        print(f"{wl}: fsi_min {fsi_min}: {min(outs):.2f} ms per batch (both node types), finite {all(bool(torch.isfinite(p).all()) for p in pe)}", flush=True)
_lib.lib().singa_lap_pe_fsi_min(384)
