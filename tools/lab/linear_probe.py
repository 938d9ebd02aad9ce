"""nn.Linear shapes of the CProMG transformer at config 3 (rows = atoms / tokens of the whole batch): the own MFMA GEMM
against the BLAS-library path (ops._Linear), forward and backward (lab probe, not a test).

Result (round 2, MI355X): no gain - [45440 x 256] -> 1024: own 270 / 575 us (fwd / bwd) vs library 283 / 503 us;
[45440 x 1024] -> 256: 286 / 558 vs 198 / 489; [45440 x 256] -> 256: 77 / 239 vs 91 / 191.  With reductions of 256 (eight K
steps) the 128 x 128 tile's prologue / epilogue weigh too much; the transformer's Linear layers stay on hipBLASLt / rocBLAS.
To rerun, define an autograd Function around ops.gemm_nt / ops._gemm as ops._SO2Linear3 does and pass it below."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import ops

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


for M, K, N in ((45440, 256, 1024), (45440, 1024, 256), (45440, 256, 256), (45440, 256, 128), (25600, 256, 1024),
                (25600, 1024, 256), (25600, 256, 256), (25600, 256, 116), (3840, 256, 1024), (8192, 256, 256)):
    x = torch.randn(M, K, device=dev, requires_grad=True)
    w = (torch.randn(N, K, device=dev) * 0.05).requires_grad_(True)
    b = torch.zeros(N, device=dev, requires_grad=True)
    fl = 2 * M * K * N
    line = f"[{M:6d} x {K:4d}] -> {N:4d}: "
    for name, fn in (("lib", ops._Linear),):
        if name == "own" and (N % 4 or K % 4):
            continue
        t_f = timeit(lambda: fn.apply(x, w, b))
        y = fn.apply(x, w, b)
        g = torch.randn_like(y)
        t_b = timeit(lambda: torch.autograd.grad(y, [x, w, b], g, retain_graph=True))
        line += f"{name} fwd {t_f:6.1f} us ({fl / t_f / 1e6:5.1f} TF/s) bwd {t_b:6.1f} us ({2 * fl / t_b / 1e6:5.1f} TF/s)   "
    print(line)
