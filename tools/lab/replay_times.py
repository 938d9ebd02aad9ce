"""Lab: host time of a graph replay call vs GPU completion, bench workload."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import graph as G
from singa_amd.config import load_config
from singa_amd.engine import TrainStep
from singa_amd.model.GAN import SINGA
from singa_amd.optim import Adam
wl = dict(G.WORKLOADS["cfg2_b32_l2"]); n = wl.pop("n_graphs"); L = wl.pop("lmax")
for overlap in (True,):
    cfg = load_config(lmax=L); torch.manual_seed(0)
    model = SINGA(cfg, device="cuda").train()
    model.model.overlap_encoders = overlap
    eng = TrainStep(model, Adam(model.parameters(), lr=1e-4), None, use_graph=True)
    batch = G.synthetic_batch(n, **wl).to("cuda")
    for _ in range(3): eng.step(batch)
    torch.cuda.synchronize()
    hs, ts = [], []
    for _ in range(10):
        t0 = time.perf_counter(); eng.g_fb.replay(); eng.g_opt.replay(); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        hs.append(t1 - t0); ts.append(t2 - t0)
    print(f"overlap_encoders={overlap}: replay call returns after {min(hs) * 1e3:.2f} ms (host), GPU done after {min(ts) * 1e3:.2f} ms", flush=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): eng.g_fb.replay(); eng.g_opt.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"   back-to-back replays: {e0.elapsed_time(e1) / 30:.2f} ms per step (events)", flush=True)
    eng.release(); del eng, model
