"""Lab: backward passes of the two norms at config-3 sizes (forward + backward through ops, eager), us per call.
    [SINGA_PROBE_LIB=path] python tools/lab/norm_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import _lib
if os.environ.get("SINGA_PROBE_LIB"):
    _lib.LIB_PATH = os.environ["SINGA_PROBE_LIB"]
from singa_amd import ops


def t_us(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for N in (49267, 6500):
    x = torch.randn(N, 25, 16, device="cuda", requires_grad=True)
    w, b = torch.randn(5, 16, device="cuda", requires_grad=True), torch.randn(16, device="cuda", requires_grad=True)
    g = torch.randn(N, 25, 16, device="cuda")
    with torch.no_grad():
        print(f"so3_rmsnorm forward N={N}: {t_us(lambda: ops.so3_rmsnorm(x, w, b, 4)):7.1f} us")
    y = ops.so3_rmsnorm(x, w, b, 4)
    f = lambda: torch.autograd.grad(y, (x, w, b), g, retain_graph=True)
    print(f"so3_rmsnorm backward N={N}: {t_us(f):7.1f} us (incl. the column sums of the partials)")
    a = torch.randn(N, 256, device="cuda", requires_grad=True)
    r = torch.randn(N, 256, device="cuda", requires_grad=True)
    ln = torch.nn.LayerNorm(256).cuda()
    g2 = torch.randn(N, 256, device="cuda")
    y2 = ops.layer_norm_residual(a, r, ln)
    f2 = lambda: torch.autograd.grad(y2, (a, r, ln.weight, ln.bias), g2, retain_graph=True)
    print(f"layer_norm_residual backward N={N}: {t_us(f2):7.1f} us (incl. the column sums of the partials)")

for M in (186_000, 26_000):                                 # (node, head) rows of config 3 / of the 17-graph shard
    x = torch.randn(M, 32, device="cuda", requires_grad=True)
    b = torch.randn(32, device="cuda", requires_grad=True)
    g = torch.randn(M, device="cuda")
    y = ops.rowdot_bias(x, b, 0.25)
    f = lambda: torch.autograd.grad(y, (x, b), g, retain_graph=True)
    print(f"rowdot_bias backward M={M}: {t_us(f):7.1f} us (incl. the column sum of the partials)")
