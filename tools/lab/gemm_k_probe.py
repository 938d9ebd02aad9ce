"""Own NT GEMM vs torch at several K (steady-state rate vs per-tile overhead) - lab probe."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import ops
dev = "cuda"

def timeit(fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for M, N, K in ((99302, 1024, 4096), (99302, 1024, 1024), (99302, 1024, 256), (99302, 1024, 64), (16384, 1024, 1024), (32768, 4096, 4096)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.03
    out = torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * K
    t1 = timeit(lambda: ops.gemm_nt(x, w, None, out))
    t2 = timeit(lambda: torch.mm(x, w.t(), out=out))
    print(f"M={M} N={N} K={K}: own {t1:9.1f} us = {fl / t1 / 1e6:6.1f} TF/s   torch {t2:9.1f} us = {fl / t2 / 1e6:6.1f} TF/s")
