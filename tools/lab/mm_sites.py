"""Library GEMM calls (aten::mm / addmm / bmm / baddbmm) of one eager config-3 step by shape: count and GPU time (lab probe)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from singa_amd import graph as G
from singa_amd.config import load_config
from singa_amd.engine import TrainStep
from singa_amd.model.GAN import SINGA
from singa_amd.optim import Adam

wl = dict(G.WORKLOADS["cfg3_b128_l4"]); n = int(sys.argv[1]) if len(sys.argv) > 1 else wl["n_graphs"]; L = wl.pop("lmax"); wl.pop("n_graphs")
model = SINGA(load_config(lmax=L), device="cuda").train()
eng = TrainStep(model, Adam(model.parameters(), lr=1e-4), None, use_graph=False)
batch = G.synthetic_batch(n, **wl).to("cuda")
eng.step(batch); eng.step(batch)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    eng.step(batch)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::mm", "aten::addmm", "aten::bmm", "aten::baddbmm", "aten::addmm_", "aten::baddbmm_"):
        rows.append((e.device_time_total / 1e3, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
print(f"library GEMM calls: {sum(r[1] for r in rows)} launches, {sum(r[0] for r in rows):.2f} ms")
for t, c, k, s in rows[:45]:
    print(f"{t:7.3f} ms {c:4d} x {k:14s} {s}")
