"""Lab: record, for every ops.ffn_tail call of the golden 3-graph step, h, gate, the output, the output's gradient and h's gradient
-> .pt, for an A / B of two library builds."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import _lib
if os.environ.get("SINGA_PROBE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SINGA_PROBE_LIB"])
from singa_amd import ops
from tests.helpers import NAMES, golden, product_batch, state_from_spec
from singa_amd.config import load_config
from singa_amd.model.GAN import SINGA
L = int(sys.argv[1])
calls = []
_ft = ops.ffn_tail


def ft(h, gate, w, b, L_, addend=None):
    rec = {"h": h.detach().cpu(), "gate": gate.detach().cpu(), "res": addend.detach().cpu() if addend is not None else None}
    out = _ft(h, gate, w, b, L_, addend)
    rec["out"] = out.detach().cpu()
    out.register_hook(lambda g, rec=rec: rec.__setitem__("g_out", g.detach().cpu()))
    h.register_hook(lambda g, rec=rec: rec.__setitem__("g_h", g.detach().cpu()))
    gate.register_hook(lambda g, rec=rec: rec.__setitem__("g_gate", g.detach().cpu()))
    calls.append(rec)
    return out


ops.ffn_tail = ft
sd = state_from_spec(f"singa_L{L}")
z = golden(f"singa_L{L}_B3.npz")
model = SINGA(load_config(lmax=L), device="cuda")
model.load_state_dict(sd, strict=False)
model.eval()
g = product_batch(NAMES, z)
loss = torch.nn.functional.cross_entropy(model(g), g["ligand_data"]["smiIndices_tgt"].reshape(-1))
loss.backward()
torch.cuda.synchronize()
torch.save(calls, sys.argv[2])
