"""Lab: which aten ops are launched most often in one eager step (count, total GPU time, shapes)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from singa_amd import graph as G
from singa_amd.config import load_config
from singa_amd.engine import TrainStep
from singa_amd.model.GAN import SINGA
wl = dict(G.WORKLOADS["cfg2_b32_l2"]); n = wl.pop("n_graphs"); L = wl.pop("lmax")
cfg = load_config(lmax=L); torch.manual_seed(0)
model = SINGA(cfg, device="cuda").train()
from singa_amd.optim import Adam
opt = Adam(model.parameters(), lr=1e-4)
eng = TrainStep(model, opt, None, use_graph=False)
batch = G.synthetic_batch(n, **wl).to("cuda")
for _ in range(3): eng.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    eng.step(batch); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key.startswith("aten::") and e.self_device_time_total > 0:
        rows.append((e.count, e.self_device_time_total / 1e3, e.key, str(e.input_shapes)[:90]))
rows.sort(key=lambda r: -r[1])
print("top by GPU time"); [print(f"{c:5d} calls {t:8.3f} ms  {k:28s} {s}") for c, t, k, s in rows[:40]]
agg = collections.Counter(); tt = collections.Counter()
for c, t, k, s in rows: agg[k] += c; tt[k] += t
print("by op"); [print(f"{agg[k]:6d} calls {tt[k]:8.3f} ms {k}") for k, _ in tt.most_common(25)]
print("all GEMM shapes")
g = [(t, c, k, s) for c, t, k, s in rows if k in ("aten::mm", "aten::bmm", "aten::addmm")]
for t, c, k, s in sorted(g, reverse=True): print(f"{c:5d} calls {t:8.3f} ms {t / c * 1e3:8.1f} us/call {k:12s} {s}")
print("top non-GEMM ops")
g = [(t, c, k, s) for c, t, k, s in rows if k not in ("aten::mm", "aten::bmm", "aten::addmm")]
for t, c, k, s in sorted(g, reverse=True)[:70]: print(f"{c:5d} calls {t:8.3f} ms {t / c * 1e3:8.1f} us/call {k:28s} {s}")
