"""Lab: so3_rmsnorm forward / backward (plain and with the skip addend) on model-like inputs, saved for an A / B of two library
builds:  [SINGA_PROBE_LIB=...] python tools/lab/norm_ab.py out.pt"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import _lib
if os.environ.get("SINGA_PROBE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SINGA_PROBE_LIB"])
from singa_amd import ops
out = {}
gen = torch.Generator().manual_seed(3)
for L in (2, 4, 6):
    K = (L + 1) ** 2
    for N in (93, 1019, 1112, 4):
        x = torch.randn(N, K, 16, generator=gen) * 0.05
        x[:, 0, :] = torch.randn(N, 16, generator=gen) * 3.0 + 2.0            # a large l = 0 row, small higher degrees
        w = 1 + 0.1 * torch.randn(L + 1, 16, generator=gen)
        b = 0.1 * torch.randn(16, generator=gen)
        g = torch.randn(N, K, 16, generator=gen)
        g2 = torch.randn(N, K, 16, generator=gen)
        xd, wd, bd = (t.cuda().requires_grad_(True) for t in (x, w, b))
        y = ops.so3_rmsnorm(xd, wd, bd, L)
        y.backward(g.cuda())
        out[(L, N, "y")] = y.detach().cpu(); out[(L, N, "gx")] = xd.grad.cpu(); out[(L, N, "gw")] = wd.grad.cpu(); out[(L, N, "gb")] = bd.grad.cpu()
        xs, ws, bs = (t.cuda().requires_grad_(True) for t in (x, w, b))
        y2, sk = ops.so3_rmsnorm_skip(xs, ws, bs, L)
        torch.autograd.backward([y2, sk], [g.cuda(), g2.cuda()])
        out[(L, N, "gx_skip")] = xs.grad.cpu()
        # float64 reference of the same formulas (EF:2155-2192, Q3)
        x64, w64, b64 = (t.double().requires_grad_(True) for t in (x, w, b))
        deg = torch.tensor([l for l in range(L + 1) for _ in range(2 * l + 1)])
        xc = torch.cat([x64[:, :1] - x64[:, :1].mean(2, keepdim=True), x64[:, 1:]], 1)
        bal = (1.0 / ((2 * deg + 1) * (L + 1))).double().view(1, K, 1)
        nrm = ((xc * xc * bal).sum(1, keepdim=True).mean(2, keepdim=True) + 1e-5).rsqrt()
        yr = xc * nrm * w64[deg].unsqueeze(0)
        yr = torch.cat([yr[:, :1] + b64.view(1, 1, -1), yr[:, 1:]], 1)
        yr.backward(g.double())
        for name, ref in (("y", yr.detach()), ("gx", x64.grad), ("gw", w64.grad), ("gb", b64.grad)):
            got = out[(L, N, name)].double()
            print(f"L {L} N {N:5d} {name:3s} rel err vs float64 {float((got - ref).norm() / ref.norm()):.2e}")
        print(f"L {L} N {N:5d} gx_skip rel err {float((out[(L, N, 'gx_skip')].double() - x64.grad - g2.double()).norm() / (x64.grad + g2.double()).norm()):.2e}")
torch.save(out, sys.argv[1])
