"""Lab: one SINGA step with torch.empty() filled with NaN (torch.utils.deterministic.fill_uninitialized_memory): any kernel that reads a
buffer it (or its producer) did not fully write turns its consumers NaN - lists the parameters whose gradients are not finite."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.use_deterministic_algorithms(True, warn_only=True)
torch.utils.deterministic.fill_uninitialized_memory = True
import warnings
warnings.filterwarnings("ignore")
from tests.helpers import NAMES, golden, product_batch, state_from_spec
from singa_amd.config import load_config
from singa_amd.model.GAN import SINGA
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sd = state_from_spec(f"singa_L{L}")
z = golden(f"singa_L{L}_B3.npz")
model = SINGA(load_config(lmax=L), device="cuda")
model.load_state_dict(sd, strict=False)
model.eval()
g = product_batch(NAMES, z)
logits = model(g)
print("logits finite:", bool(torch.isfinite(logits).all()), flush=True)
loss = torch.nn.functional.cross_entropy(logits, g["ligand_data"]["smiIndices_tgt"].reshape(-1))
loss.backward()
torch.cuda.synchronize()
bad = [n for n, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
print(f"loss {float(loss):.6f}; parameters with non-finite gradients: {len(bad)}")
for n in bad[:40]:
    p = dict(model.named_parameters())[n]
    print("   ", n, tuple(p.shape), "non-finite elements:", int((~torch.isfinite(p.grad)).sum()))
