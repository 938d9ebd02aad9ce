"""FFN-grid / attention-grid S2 activation timing at the config-3 sizes (lab probe; SINGA_LAB_LIB=<.so> times a lab build).

Floors of the FFN-grid forward at L = 4, N = 49,267 (this probe: 1574 us incl. the autograd wrapper; 1388 us in the step):
a build that only copies the rows (no arithmetic) 852 us = 5.9 TB/s; a build without global loads / stores 1180 us - the
kernel is VALU-bound (2,082 VALU instructions per thread, 222 of them quarter-rate v_exp / v_rcp), the memory time adds
~0.3 ms on top.  A persistent variant that fetches the next item's rows before the arithmetic of the current one (25 more
VGPRs, one wave less per SIMD) was slower: 1770 us."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import _lib
if os.environ.get("SINGA_LAB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SINGA_LAB_LIB"])
from singa_amd import ops
dev = "cuda"
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for L, N in ((4, 49267), (6, 6720), (2, 7360)):
    K = (L + 1) ** 2
    x = torch.randn(N, K, 512, device=dev, requires_grad=True); gate = torch.randn(N, 512, device=dev, requires_grad=True)
    g = torch.randn(N, K, 512, device=dev)
    t_f = timeit(lambda: ops.s2act_node(x, gate, L))
    y = ops.s2act_node(x, gate, L)
    t_b = timeit(lambda: torch.autograd.grad(y, (x, gate), g, retain_graph=True))
    by = N * K * 512 * 4
    print(f"FFN grid L={L} N={N}: fwd {t_f:8.1f} us = {2 * by / t_f / 1e3:6.0f} GB/s   bwd {t_b:8.1f} us = {3 * by / t_b / 1e3:6.0f} GB/s")
