"""Lab: which torch (non-library) kernels one eager config-3 training step launches, grouped by operator and input shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import graph as G
from singa_amd.config import load_config
from singa_amd.engine import TrainStep
from singa_amd.model.GAN import SINGA
from singa_amd.optim import Adam
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3_b128_l4"
L, kw, ids, _ = G.resolve_workload(wl)
batch = G.synthetic_batch(len(ids), ids=ids, with_lap=True, **kw).to("cuda")
model = SINGA(load_config(lmax=L), device="cuda")
model.train()
opt = Adam(model.parameters(), lr=1e-4, betas=(0.99, 0.999))
eng = TrainStep(model, opt, None, use_graph=False)
for _ in range(3):
    eng.eager_step(batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    eng.eager_step(batch)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print(f"aten kernels: {tot / 1e3:.2f} ms in {sum(e.count for e in rows)} calls")
for e in rows[:45]:
    print(f"{e.self_device_time_total:9.1f} us  x{e.count:3d}  {e.key:28s} {str(e.input_shapes)[:110]}")
