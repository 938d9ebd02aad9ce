"""Lab: dump the gradients of the golden 3-graph SINGA step (L given) for the library in SINGA_PROBE_LIB -> a .pt of {name: grad}."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import _lib
if os.environ.get("SINGA_PROBE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SINGA_PROBE_LIB"])
from tests.helpers import NAMES, golden, product_batch, state_from_spec
from singa_amd.config import load_config
from singa_amd.model.GAN import SINGA
L = int(sys.argv[1])
sd = state_from_spec(f"singa_L{L}")
z = golden(f"singa_L{L}_B3.npz")
model = SINGA(load_config(lmax=L), device="cuda")
model.load_state_dict(sd, strict=False)
model.eval()
g = product_batch(NAMES, z)
loss = torch.nn.functional.cross_entropy(model(g), g["ligand_data"]["smiIndices_tgt"].reshape(-1))
loss.backward()
torch.cuda.synchronize()
keep = {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None and ("ffn.so3_linear" in n or "so2_m_conv" in n or "gating" in n or "ffn" in n)}
torch.save(keep, sys.argv[2])
