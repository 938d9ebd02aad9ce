import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.helpers import NAMES, golden, product_batch, state_from_spec
from singa_amd.config import load_config
from singa_amd.engine import TrainStep
from singa_amd.model.GAN import SINGA
L = 2
model = SINGA(load_config(lmax=L), device="cuda")
model.load_state_dict(state_from_spec(f"singa_L{L}"), strict=False)
model.eval()
cap = os.environ.get("CAPTURABLE", "1") == "1"
kw = dict(fused=True) if os.environ.get("FUSED") == "1" else (dict(foreach=False) if os.environ.get("FOREACH") == "0" else {})
opt = torch.optim.Adam(model.parameters(), lr=1e-4, betas=(0.99, 0.999), capturable=cap, **kw)
eng = TrainStep(model, opt, None, use_graph=os.environ.get("GRAPH", "1") == "1")
z = golden(f"singa_L{L}_B3.npz")
batch = product_batch(NAMES, z)
for i in range(1):
    loss = eng.step(batch)
    torch.cuda.synchronize()
    badg = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    badp = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)))
    print(f"step {i}: loss {float(loss):.6f} gradnorm {gn:.4f} bad grads {len(badg)} {badg[:4]} bad params {len(badp)} {badp[:4]}", flush=True)
print("---- manual replay with checks between fwd/bwd graph and optimizer graph")
for i in range(3):
    eng._load(batch)
    eng.g_fb.replay()
    torch.cuda.synchronize()
    bad = [(n, int((~torch.isfinite(p.grad)).sum()), p.grad.numel()) for n, p in model.named_parameters()
           if p.grad is not None and not torch.isfinite(p.grad).all()]
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)))
    eng.g_opt.replay()
    torch.cuda.synchronize()
    print(f"replay {i}: loss {float(eng.static_loss):.6f} non-finite grads: {len(bad)} {bad[:6]} gradnorm {gn:.5f} engine.grad_norm {float(eng.grad_norm):.5f}", flush=True)
big = [(n, float(p.grad.abs().max()), tuple(p.shape)) for n, p in model.named_parameters() if p.grad is not None and float(p.grad.abs().max()) > 1e6]
print("params with huge grads:", len(big))
for b in big[:40]: print("  ", b)
