"""Which call sites produce the small launches of one eager step (colsum calls by caller, aten ops by name/shape) - lab probe."""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import graph as G, ops
from singa_amd.config import load_config
from singa_amd.engine import TrainStep
from singa_amd.model.GAN import SINGA
from singa_amd.optim import Adam

wl = dict(G.WORKLOADS["cfg3_b128_l4"]); n = 16; L = wl.pop("lmax"); wl.pop("n_graphs")
cfg = load_config(lmax=L)
torch.manual_seed(0)
model = SINGA(cfg, device="cuda").train()
eng = TrainStep(model, Adam(model.parameters(), lr=1e-4), None, use_graph=False)
batch = G.synthetic_batch(n, **wl).to("cuda")
eng.step(batch)
calls = collections.Counter()
orig = ops.colsum
def counted(t):
    st = traceback.extract_stack(limit=4)
    site = " <- ".join(f"{f.name}:{f.lineno}" for f in reversed(st[:-1]))
    calls[(site, tuple(t.shape))] += 1
    return orig(t)
ops.colsum = counted
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    eng.step(batch)
    torch.cuda.synchronize()
ops.colsum = orig
print("colsum calls:", sum(calls.values()))
for (site, shp), c in calls.most_common(40):
    print(f"{c:5d}  {str(shp):28s} {site}")
print(prof.key_averages(group_by_input_shape=False).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60))
ev = collections.Counter()
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::add", "aten::add_", "aten::copy_", "aten::fill_", "aten::zeros", "aten::cat", "aten::mul", "aten::clone", "aten::contiguous", "aten::zero_", "aten::index_select"):
        ev[(e.key, str(e.input_shapes)[:90])] += e.count
for k, c in ev.most_common(60):
    print(c, k)
