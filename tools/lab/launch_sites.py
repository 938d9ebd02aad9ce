"""Lab: where do the kernel launches of one eager training step come from?  Every launching aten op is attributed to the
innermost singa_amd source line of its Python stack; ops of the backward pass are attributed to the forward line of
the autograd node they belong to (sequence numbers)."""
import sys, os, collections, bisect
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from singa_amd import graph as G
from singa_amd.config import load_config
from singa_amd.engine import TrainStep
from singa_amd.model.GAN import SINGA
from singa_amd.optim import Adam
L, wl, ids, _ = G.resolve_workload(sys.argv[1] if len(sys.argv) > 1 else "cfg3_b128_l4"); n = len(ids)
cfg = load_config(lmax=L); torch.manual_seed(0)
model = SINGA(cfg, device="cuda").train()
model.model.overlap_encoders = False
eng = TrainStep(model, Adam(model.parameters(), lr=1e-4), None, use_graph=False)
batch = G.synthetic_batch(n, ids=ids, **wl).to("cuda")
for _ in range(3): eng.step(batch)
torch.cuda.synchronize()
model.prepare(batch)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    eng.opt.zero_grad(set_to_none=True)
    eng._fwd_bwd(batch); eng._update(); torch.cuda.synchronize()

def site(stack):
    for fr in stack:
        if "singa_amd/" in fr and "engine.py" not in fr:
            f = fr.split("singa_amd/")[1]
            return f.split(":")[0].replace(".py(", ":").rstrip(")") if "(" in f else f
    return None

evs = [e for e in prof.events() if e.device_type.name == "CPU"]
fwd_site = {}
for e in evs:
    if e.sequence_nr is not None and e.sequence_nr >= 0 and e.stack:
        s = site(e.stack)
        if s and e.sequence_nr not in fwd_site:
            fwd_site[e.sequence_nr] = s
bw = sorted([(e.time_range.start, e.time_range.end, e.sequence_nr, e.name) for e in evs
             if e.name.startswith("autograd::engine::evaluate_function")])
starts = [b[0] for b in bw]
cnt, tim = collections.Counter(), collections.Counter()
seen = set()
for e in evs:
    if not e.kernels or not e.name.startswith("aten::"):
        continue
    ks = tuple((k.name, k.duration) for k in e.kernels)
    key = (e.time_range.start, ks)
    if key in seen: continue
    seen.add(key)
    s = site(e.stack) if e.stack else None
    tag = "fwd"
    if s is None:
        i = bisect.bisect_right(starts, e.time_range.start) - 1
        if i >= 0 and bw[i][1] >= e.time_range.end:
            s = fwd_site.get(bw[i][2], bw[i][3].split(": ")[-1]); tag = "bwd"
    s = f"{tag} {s} {e.name[6:]}"
    cnt[s] += len(e.kernels); tim[s] += sum(k.duration for k in e.kernels)
tot = sum(cnt.values())
print(f"{tot} launches attributed")
for s, c in cnt.most_common(140):
    print(f"{c:5d} launches {tim[s] / 1e3:8.3f} ms  {s}")
