"""Lab: worst element-wise gradient sample errors against the reference's goldens (singa_L<L>_B3) for the library given in
SINGA_PROBE_LIB (default: the in-tree build) - A / B of two builds on the parameters whose gradients are ill-conditioned sums.
    [SINGA_PROBE_LIB=tools/lab/build/libsinga_prev.so] python tools/lab/grad_ab_probe.py [L]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import _lib
if os.environ.get("SINGA_PROBE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SINGA_PROBE_LIB"])
from oracle import weights as W
from tests.helpers import NAMES, golden, product_batch, state_from_spec
from singa_amd.config import load_config
from singa_amd.model.GAN import SINGA

L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NOISE = float(os.environ.get("SINGA_PROBE_NORM_NOISE", "0"))
if NOISE:
    # how ill-conditioned are the gradients w.r.t. the norm's rounding?  Multiply every norm output by (1 + NOISE * randn)
    from singa_amd import ops
    _plain, _skip = ops.so3_rmsnorm, ops.so3_rmsnorm_skip
    gen = torch.Generator(device="cuda").manual_seed(1)
    jit = lambda y: y * (1.0 + NOISE * torch.randn(y.shape, device=y.device, generator=gen))
    ops.so3_rmsnorm = lambda *a, **k: jit(_plain(*a, **k))
    def _sk(*a, **k):
        y, s_ = _skip(*a, **k)
        return jit(y), s_
    ops.so3_rmsnorm_skip = _sk
sd = state_from_spec(f"singa_L{L}")
z = golden(f"singa_L{L}_B3.npz")
for rep in range(1):
    model = SINGA(load_config(lmax=L), device="cuda")
    model.load_state_dict(sd, strict=False)
    model.eval()
    if os.environ.get("SINGA_PROBE_ONE_STREAM") == "1":
        model.model.overlap_encoders = False
        model.embedding.overlap_hetero_passes = False
    g = product_batch(NAMES, z)
    logits = model(g)
    loss = torch.nn.functional.cross_entropy(logits, g["ligand_data"]["smiIndices_tgt"].reshape(-1))
    loss.backward()
    torch.cuda.synchronize()
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
        rows.append((float((got - want).norm() / (want.norm() + 1e-12)), str(n), float(want.norm()), abs(float(gr.norm()) - ref) / ref))
    rows = [r for r in rows if r[2] > 1e-6]          # (gradients that are zero up to rounding - the key biases - are noise)
    rows.sort(reverse=True)
    print(f"lib {_lib.LIB_PATH} run {rep}: loss {float(loss):.6f} (ref {float(z['loss']):.6f}); worst sample errors (rel, name, |samples|, rel err of norm)")
    for r in rows[:10]:
        print("   %.2e  %-58s %.3e %.2e" % r)
