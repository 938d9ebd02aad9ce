"""Lab: op-level GPU time with shapes (torch.profiler, eager) for one training step at the bench workload."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from singa_amd import graph as G
from singa_amd.config import load_config
from singa_amd.engine import TrainStep
from singa_amd.model.GAN import SINGA
wl = dict(G.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2_b32_l2"]); n = wl.pop("n_graphs"); L = wl.pop("lmax")
cfg = load_config(lmax=L); torch.manual_seed(0)
model = SINGA(cfg, device="cuda").train()
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
eng = TrainStep(model, opt, None, use_graph=False)
batch = G.synthetic_batch(n, **wl).to("cuda")
for _ in range(3): eng.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(2): eng.step(batch)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=40, max_shapes_column_width=70))
