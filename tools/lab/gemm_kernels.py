"""Lab: which library kernel serves which GEMM shape in one eager training step (shape, kernel, calls, us/call)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from singa_amd import graph as G
from singa_amd.config import load_config
from singa_amd.engine import TrainStep
from singa_amd.model.GAN import SINGA
from singa_amd.optim import Adam
wl = dict(G.WORKLOADS["cfg2_b32_l2"]); n = wl.pop("n_graphs"); L = wl.pop("lmax")
cfg = load_config(lmax=L); torch.manual_seed(0)
model = SINGA(cfg, device="cuda").train()
eng = TrainStep(model, Adam(model.parameters(), lr=1e-4), None, use_graph=False)
batch = G.synthetic_batch(n, **wl).to("cuda")
for _ in range(3): eng.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    eng.step(batch); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name in ("aten::mm", "aten::bmm", "aten::addmm") and e.kernels:
        for k in e.kernels:
            a = agg[(e.name, str(e.input_shapes), k.name[:60])]
            a[0] += 1; a[1] += k.duration
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print(f"GEMM kernel time {tot / 1e3:.2f} ms")
for (op, shp, kn), (c, t) in rows[:90]:
    print(f"{t / 1e3:7.3f} ms {c:4d} x {t / c:7.1f} us  {op[6:]:6s} {shp[:72]:72s} {kn}")
