"""Lab: every parameter gradient of the golden 3-graph step, HIP path vs the CPU oracle (FULL tensors, not the golden's
samples) and both norms vs the golden's recorded norm."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import singa_oracle as O
from tests.helpers import NAMES, golden, product_batch, state_from_spec, pinned_relu_ties
from singa_amd.config import load_config
from singa_amd.model.GAN import SINGA
L = int(sys.argv[1])
sd = state_from_spec(f"singa_L{L}")
z = golden(f"singa_L{L}_B3.npz")
model = SINGA(load_config(lmax=L), device="cuda")
model.load_state_dict(sd, strict=False)
model.eval()
g = product_batch(NAMES, z)
with pinned_relu_ties(L):
    loss = torch.nn.functional.cross_entropy(model(g), g["ligand_data"]["smiIndices_tgt"].reshape(-1))
    loss.backward()
hip = {n: p.grad.detach().cpu().double() for n, p in model.named_parameters() if p.grad is not None}
sdo = {k: v.clone().double().requires_grad_(True) for k, v in state_from_spec(f"singa_L{L}").items()}
go = O.collate([O.load_graph_npz(f"tests/golden/graph_{n}.npz") for n in NAMES])
rots = {k: torch.as_tensor(z[f"rot_{k}"]).double() for k in ("pp", "ll", "lp")}
try:
    logits = O.singa_forward(sdo, go, rots, L, torch.as_tensor(z["knn_p"]), torch.as_tensor(z["knn_l"]),
                             torch.as_tensor(z["lap_p"]).double(), torch.as_tensor(z["lap_l"]).double())
    tag = "float64 oracle"
except Exception as e:                                        # the oracle may insist on float32 somewhere
    print("float64 oracle failed:", repr(e)[:200])
    sdo = {k: v.clone().requires_grad_(True) for k, v in state_from_spec(f"singa_L{L}").items()}
    rots = {k: torch.as_tensor(z[f"rot_{k}"]) for k in ("pp", "ll", "lp")}
    logits = O.singa_forward(sdo, go, rots, L, torch.as_tensor(z["knn_p"]), torch.as_tensor(z["knn_l"]),
                             torch.as_tensor(z["lap_p"]), torch.as_tensor(z["lap_l"]))
    tag = "float32 oracle"
torch.nn.functional.cross_entropy(logits, go["tok_tgt"].reshape(-1)).backward()
gold = {str(n): float(v) for n, v in zip(z["grad_names"], z["grad_norms"])}
rows = []
for n, h in hip.items():
    o = sdo[n].grad
    if o is None or float(o.norm()) < 1e-6:            # (W_K.bias: analytically zero, rounding noise only)
        continue
    o = o.double()
    full = float((h - o).norm() / (o.norm() + 1e-30))
    rows.append((full, n, float(h.norm()), float(o.norm()), gold.get(n, float("nan"))))
rows.sort(reverse=True)
print(tag, "- largest full-tensor relative errors (hip vs oracle), then norms hip / oracle / golden:")
for full, n, nh, no, ng in rows[:12]:
    print(f"  {full:.2e}  {n}: {nh:.7g} / {no:.7g} / {ng:.7g}   norm rel: hip-golden {abs(nh - ng) / ng:.1e}, oracle-golden {abs(no - ng) / ng:.1e}")
print("largest |norm_hip - norm_golden| / norm_golden:")
for full, n, nh, no, ng in sorted(rows, key=lambda r: -abs(r[2] - r[4]) / (r[4] + 1e-30))[:8]:
    print(f"  {abs(nh - ng) / ng:.2e}  {n}: hip {nh:.7g} oracle {no:.7g} golden {ng:.7g}  full err {full:.1e}")
