"""Lab: GPU time of the phases of one step (eager, events), bench workload."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import graph as G
from singa_amd.config import load_config
from singa_amd.model.GAN import SINGA, lap_pe
from singa_amd.graph import PA, LA

wlname = sys.argv[1] if len(sys.argv) > 1 else "cfg2_b32_l2"
wl = dict(G.WORKLOADS[wlname]); n = wl.pop("n_graphs"); L = wl.pop("lmax")
if len(sys.argv) > 2: n = int(sys.argv[2])
cfg = load_config(lmax=L)
torch.manual_seed(0)
dev = "cuda"
model = SINGA(cfg, device=dev).train()
batch = G.synthetic_batch(n, **wl).to(dev)
model.prepare(batch)

def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e

def run():
    t = [ev()]
    emb = model.embedding(batch); t.append(ev())
    feat = cfg.model.featurizer_feat_dim
    ld = batch["ligand_data"]
    prop = torch.stack([(ld["vina_score"] < -7.5), (ld["qed"] > 0.6), (ld["sas"] < 4.0)], 1).float()
    logits = model.model(node_attr=emb[PA].embedding.reshape(-1, feat), pos=batch[PA]["pos"], batch=batch[PA]["batch"],
                         atom_laplacian=lap_pe(batch, PA), smiles_index=ld["smiIndices_input"], tgt_len=200,
                         aa_node_attr=emb[LA].embedding.reshape(-1, feat), aa_pos=batch[LA]["pos"], aa_batch=batch[LA]["batch"],
                         aa_laplacian=lap_pe(batch, LA), prop=prop, prep=batch.extras["prepared"])
    t.append(ev())
    loss = torch.nn.functional.cross_entropy(logits, ld["smiIndices_tgt"].reshape(-1))
    model.zero_grad(set_to_none=True)
    loss.backward(); t.append(ev())
    torch.cuda.synchronize()
    return [t[i].elapsed_time(t[i + 1]) for i in range(3)]

for _ in range(3): run()
r = [run() for _ in range(5)]
m = [sum(x[i] for x in r) / len(r) for i in range(3)]
print(f"{wlname} n={n}: embedding fwd {m[0]:.1f} ms | transformer fwd {m[1]:.1f} ms | backward (both) {m[2]:.1f} ms  (eager wall, includes launch gaps)")
# embedding-only backward
def run2():
    a = ev(); emb = model.embedding(batch); b = ev()
    l = (emb[PA].embedding ** 2).sum() + (emb[LA].embedding ** 2).sum()
    model.zero_grad(set_to_none=True); l.backward(); c = ev(); torch.cuda.synchronize()
    return a.elapsed_time(b), b.elapsed_time(c)
for _ in range(2): run2()
r = [run2() for _ in range(5)]
print(f"embedding only: fwd {sum(x[0] for x in r)/5:.1f} ms, bwd {sum(x[1] for x in r)/5:.1f} ms")
