"""Lab: kernel-level profile of one KnnEdges construction (protein encoder, config 3) with torch.profiler."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import graph as G, ops
from singa_amd.config import load_config
from singa_amd.model.GAN import SINGA
from singa_amd.model import CProMG as CP
L, kw, ids, _ = G.resolve_workload("cfg3_b128_l4")
batch = G.synthetic_batch(len(ids), ids=ids, with_lap=True, **kw).to("cuda")
model = SINGA(load_config(lmax=L), device="cuda")
enc = model.model.encoder
pos, b = batch[G.PA]["pos"], batch[G.PA]["batch"]
dm = CP.DenseMap(b, batch.num_graphs)
knn = CP.knn_graph(pos, 48, b, batch.num_graphs, dm)
for _ in range(2):
    e = CP.KnnEdges(pos, knn, enc.distance_expansion)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        e = CP.KnnEdges(pos, knn, enc.distance_expansion)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
