"""Lab: replay time of separately captured pieces of the step (no profiler), bench workload."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import graph as G
from singa_amd.config import load_config
from singa_amd.model.GAN import SINGA, lap_pe
from singa_amd.graph import PA, LA
from singa_amd.model import EF_layers
wl = dict(G.WORKLOADS["cfg2_b32_l2"]); n = wl.pop("n_graphs"); L = wl.pop("lmax")
cfg = load_config(lmax=L); torch.manual_seed(0)
model = SINGA(cfg, device="cuda").train()
batch = G.synthetic_batch(n, **wl).to("cuda")
prep = model.prepare(batch)
for key, et in (("pp", G.E_PP), ("ll", G.E_LL), ("lp", G.E_LP), ("pl", G.E_PL)):
    EF_layers._edge_pinned[batch[et]["edge_index"].data_ptr()] = prep["es"][key]
if "homo" in prep:
    EF_layers._edge_pinned[prep["homo"]["ei"].data_ptr()] = prep["homo"]["es"]
feat = cfg.model.featurizer_feat_dim
ld = batch["ligand_data"]
tgt = ld["smiIndices_tgt"].reshape(-1)

def timed(name, fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    model.zero_grad(set_to_none=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(2): g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:42s} {e0.elapsed_time(e1) / 10:7.2f} ms", flush=True)
    del g
    model.zero_grad(set_to_none=True); torch.cuda.empty_cache()

def emb_fwd():
    with torch.no_grad():
        return model.embedding(batch)
def emb_fwd_bwd():
    model.zero_grad(set_to_none=True)
    e = model.embedding(batch)
    ((e[PA].embedding ** 2).sum() + (e[LA].embedding ** 2).sum()).backward()
emb_const = None
def tf_inputs():
    global emb_const
    if emb_const is None:
        with torch.no_grad():
            e = model.embedding(batch)
        emb_const = (e[PA].embedding.reshape(-1, feat).clone(), e[LA].embedding.reshape(-1, feat).clone())
    return emb_const
def tf(fa, fl):
    prop = torch.ones(n, 3, device="cuda")
    return model.model(node_attr=fa, pos=batch[PA]["pos"], batch=batch[PA]["batch"], atom_laplacian=lap_pe(batch, PA),
                       smiles_index=ld["smiIndices_input"], tgt_len=200, aa_node_attr=fl, aa_pos=batch[LA]["pos"],
                       aa_batch=batch[LA]["batch"], aa_laplacian=lap_pe(batch, LA), prop=prop, prep=prep)
def tf_fwd():
    fa, fl = tf_inputs()
    with torch.no_grad():
        tf(fa, fl)
def tf_fwd_bwd():
    fa, fl = tf_inputs()
    model.zero_grad(set_to_none=True)
    fa = fa.detach().requires_grad_(True); fl = fl.detach().requires_grad_(True)
    torch.nn.functional.cross_entropy(tf(fa, fl), tgt).backward()
def full():
    model.zero_grad(set_to_none=True)
    torch.nn.functional.cross_entropy(model(batch), tgt).backward()
tf_inputs()
if len(sys.argv) > 1 and sys.argv[1] == "dec":
    tfm = model.model
    enc_out = torch.randn(n, 230, 256, device="cuda"); enc_mask = torch.zeros(n, 1, 230, dtype=torch.bool, device="cuda")
    def dec():
        model.zero_grad(set_to_none=True)
        e = enc_out.detach().requires_grad_(True)
        d = tfm.decoder(ld["smiIndices_input"], e, enc_mask, 200, torch.ones(n, 3, device="cuda"))
        torch.nn.functional.cross_entropy(tfm.projection(d)[:, 1:].reshape(-1, 116), tgt).backward()
    timed("decoder + projection + CE fwd+bwd", dec)
    timed("decoder + projection + CE fwd+bwd (again)", dec)
    sys.exit(0)
timed("embedding forward (no grad)", emb_fwd)
timed("embedding forward + backward", emb_fwd_bwd)
timed("transformer forward (no grad)", tf_fwd)
timed("transformer forward + backward", tf_fwd_bwd)
model.model.overlap_encoders = False
timed("transformer fwd + bwd, one stream", tf_fwd_bwd)
model.model.overlap_encoders = True
timed("full forward + backward", full)
