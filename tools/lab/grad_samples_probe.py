"""Which parameters' gradient samples deviate from the reference's (singa_L<L>_B3 golden), under which GEMM path (lab probe)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from oracle import weights as W
from tests.helpers import NAMES, golden, product_batch, state_from_spec
from singa_amd import ops
from singa_amd.config import load_config
from singa_amd.model.GAN import SINGA

L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for own in (True, False):
    ops.USE_OWN_GEMM = own
    sd = state_from_spec(f"singa_L{L}")
    z = golden(f"singa_L{L}_B3.npz")
    model = SINGA(load_config(lmax=L), device="cuda")
    model.load_state_dict(sd, strict=False)
    model.eval()
    g = product_batch(NAMES, z)
    logits = model(g)
    loss = torch.nn.functional.cross_entropy(logits, g["ligand_data"]["smiIndices_tgt"].reshape(-1))
    loss.backward()
    params = dict(model.named_parameters())
    off, rows = 0, []
    for n, ref in zip(z["grad_names"], z["grad_norms"]):
        if ref < 0:
            continue
        gr = params[str(n)].grad
        idx = W.sample_index(gr.numel())
        want = torch.as_tensor(z["grad_samples"][off:off + len(idx)], dtype=torch.float64)
        off += len(idx)
        got = gr.detach().reshape(-1).cpu()[torch.as_tensor(idx)].double()
        rows.append((float((got - want).norm() / (want.norm() + 1e-12)), str(n), float(want.norm()), float(ref),
                     abs(float(gr.norm()) - ref) / ref))
    rows.sort(reverse=True)
    print(f"own={own}: worst sample errors (rel err of samples, name, |samples|, |grad| ref, rel err of norm)")
    for r in rows[:12]:
        print("   %.2e  %-60s %.3e %.3e %.2e" % r)
