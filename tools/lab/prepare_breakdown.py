"""Lab: where SINGA.prepare spends its time on one config-3 batch (synchronised timers around its pieces)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import graph as G, ops
from singa_amd.config import load_config
from singa_amd.model.GAN import SINGA
from singa_amd.model import CProMG as CP, EF_layers

L, kw, ids, _ = G.resolve_workload("cfg3_b128_l4")
batch = G.synthetic_batch(len(ids), ids=ids, with_lap=False, **kw).to("cuda")
model = SINGA(load_config(lmax=L), device="cuda")
acc = {}


def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        return r
    return w


_ke = CP.KnnEdges.__init__
CP.KnnEdges.__init__ = timed("KnnEdges", _ke)
CP.knn_graph = timed("knn_graph", CP.knn_graph)
ops.EdgeSet.__init__ = timed("EdgeSet (bonded / interaction edges: sort by destination)", ops.EdgeSet.__init__)
G.laplacian_pe_batched = timed("laplacian_pe_batched", G.laplacian_pe_batched)
CP.DenseMap.__init__ = timed("DenseMap", CP.DenseMap.__init__)
reps = 5
for i in range(reps + 1):
    if i == 1:
        acc.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    EF_layers._edge_cache.clear()
    batch.extras.pop("prepared", None)
    for nt in (G.PA, G.LA):
        batch[nt].pop("lap_pe", None) if hasattr(batch[nt], "pop") else None
    model.prepare(batch)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / reps
print(f"prepare: {1e3 * tot:.2f} ms per batch (with synchronising timers inside)")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {1e3 * v / reps:7.2f} ms  {k}")
