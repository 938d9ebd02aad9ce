"""Lab: GPU time of SINGA.prepare (per-batch graph structure) by op, bench workload."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from singa_amd import graph as G
from singa_amd.config import load_config
from singa_amd.model.GAN import SINGA
from singa_amd.model import EF_layers
wl = dict(G.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg3_b128_l4"]); n = wl.pop("n_graphs"); L = wl.pop("lmax")
model = SINGA(load_config(lmax=L), device="cuda")
batch = G.synthetic_batch(n, **wl).to("cuda")
def prep():
    EF_layers._edge_cache.clear(); batch.extras.pop("prepared", None); model.prepare(batch)
for _ in range(3): prep()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    prep(); torch.cuda.synchronize()
rows = [(e.self_device_time_total / 1e3, e.count, e.key, str(e.input_shapes)[:80]) for e in prof.key_averages(group_by_input_shape=True)
        if e.self_device_time_total > 0]
rows.sort(reverse=True)
print(f"total GPU time {sum(r[0] for r in rows):.3f} ms in {sum(r[1] for r in rows)} launches")
for t, c, k, s in rows[:40]: print(f"{t:7.3f} ms {c:4d} x  {k:32s} {s}")
