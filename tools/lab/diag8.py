"""Lab: where the HIP path and the CPU oracle part on 8-graph batches.  Per-graph CrossEntropy, the kNN
edge sets of both sides, the product run on the oracle's kNN lists, and the total gradient norm.
    python tools/lab/diag8.py [workload] [n_graphs] [cdist modes ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from oracle import singa_oracle as O
from singa_amd import graph as G
from singa_amd.config import load_config
from singa_amd.model import CProMG, EF_layers
from singa_amd.model.GAN import SINGA

workload = sys.argv[1] if len(sys.argv) > 1 else "cfg3_b128_l4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
modes = sys.argv[3:] or ["kernel"]
dev = "cuda"
L, kw, ids, _ = G.resolve_workload(workload)
ids = ids[:n]
graphs = [G.synthetic_graph(i, **G.graph_sizes(i, **kw)) for i in ids]
torch.manual_seed(7)
model = SINGA(load_config(lmax=L), device=dev).eval()
sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
b, rots, lap_p, lap_l = O.batch_from_graphs(graphs)
B = len(ids)
bp = torch.repeat_interleave(torch.arange(B), b["ptr_p"][1:] - b["ptr_p"][:-1])
bl = torch.repeat_interleave(torch.arange(B), b["ptr_l"][1:] - b["ptr_l"][:-1])
knn_p, knn_l = O.knn_graph(b["pos_p"], 48, bp), O.knn_graph(b["pos_l"], 30, bl)
t0 = time.time()
ref = O.singa_forward(sd, b, rots, L, knn_p, knn_l, lap_p, lap_l)
tgt = b["tok_tgt"].reshape(-1)
loss_o = F.cross_entropy(ref, tgt)
loss_o.backward()
gn_o = float(torch.sqrt(sum((v.grad.double() ** 2).sum() for v in sd.values() if v.grad is not None)))
print(f"{workload} x {B}: oracle loss {float(loss_o):.6f} grad norm {gn_o:.6f} ({time.time() - t0:.1f} s)", flush=True)
T = tgt.numel() // B
ce_o = F.cross_entropy(ref.detach(), tgt, reduction="none").view(B, T).mean(1)


def keyset(ei, N):
    ok = (ei[0] >= 0) & (ei[1] >= 0)
    return set((ei[0][ok] * N + ei[1][ok]).tolist())


def run(tag, knn=None):
    batch = G.collate(graphs)
    if knn is not None:
        batch.extras["knn"] = {G.PA: knn[0], G.LA: knn[1]}
    batch = batch.to(dev)
    EF_layers._edge_cache.clear()
    model.zero_grad(set_to_none=True)
    logits = model(batch)
    loss = F.cross_entropy(logits, batch["ligand_data"]["smiIndices_tgt"].reshape(-1))
    loss.backward()
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)))
    lg = logits.detach().cpu()
    ce = F.cross_entropy(lg, tgt, reduction="none").view(B, T).mean(1)
    print(f"[{tag}] loss {float(loss):.6f} (rel {abs(float(loss) - float(loss_o)) / float(loss_o):.2e}) grad norm {gn:.6f} "
          f"(rel {abs(gn - gn_o) / gn_o:.2e}) logits rel {float((lg.double() - ref.detach().double()).norm() / ref.detach().double().norm()):.2e}")
    for i in range(B):
        r = lg.view(B, T, -1)[i].double() - ref.detach().view(B, T, -1)[i].double()
        print(f"    graph {ids[i]}: CE hip {float(ce[i]):.6f} oracle {float(ce_o[i]):.6f} rel {abs(float(ce[i] - ce_o[i])) / float(ce_o[i]):.2e} "
              f"logits rel {float(r.norm() / ref.detach().view(B, T, -1)[i].double().norm()):.2e}")
    return gn


for mode in modes:
    CProMG.KNN_CDIST_MODE = mode
    for nt, k, pos, bt, ko in ((G.PA, 48, b["pos_p"], bp, knn_p), (G.LA, 30, b["pos_l"], bl, knn_l)):
        own = CProMG.knn_graph(pos.to(dev), k, bt.to(dev), B).cpu()
        N = pos.shape[0]
        so, sp = keyset(ko, N), keyset(own, N)
        only_o, only_p = sorted(so - sp), sorted(sp - so)
        print(f"cdist mode {mode}: {nt} kNN pairs oracle {len(so)} product {len(sp)}; only oracle {len(only_o)}, only product {len(only_p)}")
        for key in only_o[:6] + only_p[:6]:
            i, j = key // N, key % N
            print(f"      pair ({i},{j}) graph {int(bt[i])} dist {float((pos[i].double() - pos[j].double()).norm()):.7f}")
    run(f"own kNN, {mode}")
run("oracle kNN lists", (knn_p, knn_l))
