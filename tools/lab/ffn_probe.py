"""SO3_LinearV2 16->512 and 512->16 on the node tensors of the config-3 step (N = 50432, L = 4): forward / backward of the
own grouped-row GEMM, GB/s of the [N, 25, 512] tensor they stream (lab probe, not a test)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import ops

dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50432
L = 4
K = (L + 1) ** 2


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


big = N * K * 512 * 4 / 1e9
for cin, cout in ((16, 512), (512, 16)):
    x = torch.randn(N, K, cin, device=dev, requires_grad=True)
    w = (torch.randn(L + 1, cout, cin, device=dev) * 0.05).requires_grad_(True)
    b = torch.zeros(cout, device=dev, requires_grad=True)
    t_f = timeit(lambda: ops.so3_linear(x, w, b, L))
    y = ops.so3_linear(x, w, b, L)
    g = torch.randn_like(y)
    t_x = timeit(lambda: torch.autograd.grad(y, [x], g, retain_graph=True))
    t_w = timeit(lambda: torch.autograd.grad(y, [w, b], g, retain_graph=True))
    t_all = timeit(lambda: torch.autograd.grad(y, [x, w, b], g, retain_graph=True))
    print(f"so3_linear {cin}->{cout} N={N}: fwd {t_f:7.1f} us ({big / t_f * 1e6:5.0f} GB/s of the big tensor)  "
          f"dX-only {t_x:7.1f}  dW-only {t_w:7.1f}  bwd all {t_all:7.1f} us")

# Tried and dropped (round 2): gate Linear + SO3 linear 16->512 + S2 activation as ONE kernel (thread = (node, hidden channel),
# its (L+2)*16 weights in registers, the node's input rows through LDS, hidden tensor never stored).  At N = 50432, L = 4:
# forward 1876 us vs 2364 us for the three ops, but the backward twin needs 274 VGPRs (one wavefront per SIMD) and took
# 4681 us vs 3834 us - the S2 activation is VALU-bound and loses more to the lower occupancy than it gains from the saved
# 2 x 2.6 GB of traffic.
