"""Lab: weight-gradient products dW[N, K] = g^T x of the transformer's Linears: own GEMM (0, 0) form + the column sum of the
partial slabs, as a function of the split count S (ops._tn_splits picks one per shape)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import ops

dev = "cuda"


def t_us(fn, n=8):
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


SHAPES = [(M, N, K) for M in (49267, 6499) for N, K in ((256, 256), (1024, 256), (256, 1024), (128, 256), (256, 400), (256, 8))]
SHAPES += [(2_900_000, 64, 20), (2_900_000, 64, 64), (186_000, 128, 32), (374_000, 64, 20)]
for M, N, K in SHAPES:
    if True:
        g, x = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)
        line = f"M={M:6d} [{N:4d} x {K:4d}]  chosen S={ops._tn_splits(M, N, K):2d}:"
        for S in (8, 16, 32, 64, 128, 256, 512, 1024):
            if S > max(1, M // 256):
                continue
            part = torch.empty(S, N * K, device=dev)

            def run():
                ops._gemm([dict(a=g.data_ptr(), lda=N, b=x.data_ptr(), ldb=K, c=part.data_ptr(), ldc=K, I=N, J=K, R=M,
                                c_split_stride=N * K)], False, False, S)
                ops.colsum(part)
            line += f"  S{S}: {t_us(run):6.1f}"
        print(line, flush=True)
