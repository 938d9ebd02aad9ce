"""Lab: k11s expand / reduce, VALU (lane broadcast) vs MFMA (16x16x4) form, at the config-3 launch size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import ops, _lib

dev = "cuda"


def timeit(fn, n=10):
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


for N, L in ((49267, 4), (6499, 4), (53760, 6)):
    K = (L + 1) ** 2
    x = torch.randn(N, K, 16, device=dev, requires_grad=True)
    w1 = (torch.randn(L + 1, 512, 16, device=dev) * 0.2).requires_grad_(True)
    b1 = torch.randn(512, device=dev, requires_grad=True)
    w2 = (torch.randn(L + 1, 16, 512, device=dev) * 0.05).requires_grad_(True)
    b2 = torch.randn(16, device=dev, requires_grad=True)
    gb = N * K * 512 * 4 / 1e9
    for valu in (1, 0):
        _lib.lib().singa_so3_skinny_variant(valu)
        y = ops.so3_linear(x, w1, b1, L)
        g = torch.randn_like(y)
        t_f = timeit(lambda: ops.so3_linear(x, w1, b1, L))
        t_b = timeit(lambda: torch.autograd.grad(y, [x, w1, b1], g, retain_graph=True))
        h = y.detach().requires_grad_(True)
        z = ops.so3_linear(h, w2, b2, L)
        gz = torch.randn_like(z)
        t_b2 = timeit(lambda: torch.autograd.grad(z, [h, w2, b2], gz, retain_graph=True))
        print(f"N={N} L={L} {'VALU' if valu else 'MFMA'}: 16->512 fwd (expand) {t_f:7.1f} us = {gb / t_f * 1e3:5.2f} TB/s | bwd (MFMA dx + reduce) "
              f"{t_b:7.1f} us | 512->16 bwd (expand + reduce) {t_b2:7.1f} us")
    _lib.lib().singa_so3_skinny_variant(0)
