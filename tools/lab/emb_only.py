"""Lab: replay only the equivariant embedding (forward + backward) 20 times - run under rocprofv3 for a per-kernel view."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import graph as G
from singa_amd.config import load_config
from singa_amd.model.GAN import SINGA
from singa_amd.graph import PA, LA
from singa_amd.model import EF_layers
wl = dict(G.WORKLOADS["cfg2_b32_l2"]); n = wl.pop("n_graphs"); L = wl.pop("lmax")
cfg = load_config(lmax=L); torch.manual_seed(0)
model = SINGA(cfg, device="cuda").train()
batch = G.synthetic_batch(n, **wl).to("cuda")
prep = model.prepare(batch)
for key, et in (("pp", G.E_PP), ("ll", G.E_LL), ("lp", G.E_LP), ("pl", G.E_PL)):
    EF_layers._edge_pinned[batch[et]["edge_index"].data_ptr()] = prep["es"][key]
EF_layers._edge_pinned[prep["homo"]["ei"].data_ptr()] = prep["homo"]["es"]
def fb():
    model.zero_grad(set_to_none=True)
    e = model.embedding(batch)
    ((e[PA].embedding ** 2).sum() + (e[LA].embedding ** 2).sum()).backward()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): fb()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
model.zero_grad(set_to_none=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fb()
for _ in range(20): g.replay()
torch.cuda.synchronize()
