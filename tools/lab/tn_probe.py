"""Lab: a^T b for tall-skinny operands - library GEMM vs batched split-K (+ colsum of the partials) at several chunk sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__; __graft_entry__.build()
from singa_amd import ops

def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

shapes = [(256, 256, 6400), (128, 256, 6432), (64, 64, 5938), (32, 64, 5938), (1024, 256, 6400), (256, 1024, 6400),
          (64, 64, 4224), (32, 32, 25600), (448, 512, 7296), (736, 96, 7296), (64, 64, 374578), (16, 48, 56448), (256, 64, 7296)]
for p, q, K in shapes:
    a, b = torch.randn(K, p, device="cuda"), torch.randn(K, q, device="cuda")
    row = [f"{p}x{K}x{q}".ljust(18), f"lt {t(lambda: a.t() @ b):7.1f}"]
    torch.backends.cuda.preferred_blas_library("cublas")
    row.append(f"rocblas {t(lambda: a.t() @ b):7.1f}")
    torch.backends.cuda.preferred_blas_library("cublaslt")
    for chunk in (256, 512, 1024, 2048, 8192):
        S = K // chunk
        if S < 2: continue
        Mc = S * chunk
        def f():
            out = ops.colsum(torch.bmm(a[:Mc].view(S, chunk, p).transpose(1, 2), b[:Mc].view(S, chunk, q)))
            if Mc < K: out = out + a[Mc:].t() @ b[Mc:]
            return out
        row.append(f"c{chunk} {t(f):7.1f}")
    print("  ".join(row), flush=True)
