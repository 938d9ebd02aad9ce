"""Lab: fixed cost vs per-K-step cost of the own GEMM at the 17-graph shard's row counts.  For every form (NT forward, NN dX,
TN dW with the split count the product would choose) and tile shape (forced 128 x 128 / 64 x 64): us per launch at K = 256,
512, 1024 - the slope is the cost of a K step, the intercept the launch's fixed cost."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import ops, _lib

dev = "cuda"


def t_us(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


cfgs = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,3".split(","))]
for M in (6499, 3417, 13000):
    for N in (256, 1024):
        for K in (256, 512, 1024):
            x = torch.randn(M, K, device=dev)
            w = torch.randn(N, K, device=dev) * 0.05
            g = torch.randn(M, N, device=dev)
            y = torch.empty(M, N, device=dev)
            gx = torch.empty(M, K, device=dev)
            row = []
            for cfg in cfgs:
                _lib.lib().singa_gemm_force_cfg(cfg)
                nt = t_us(lambda: ops._gemm([dict(a=x.data_ptr(), lda=K, b=w.data_ptr(), ldb=K, c=y.data_ptr(), ldc=N, I=M, J=N, R=K)], True, True))
                nn = t_us(lambda: ops._gemm([dict(a=g.data_ptr(), lda=N, b=w.data_ptr(), ldb=K, c=gx.data_ptr(), ldc=K, I=M, J=K, R=N)], True, False))
                S = ops._tn_splits(M, N, K)
                part = torch.empty(S, N * K, device=dev)
                tn = t_us(lambda: ops._gemm([dict(a=g.data_ptr(), lda=N, b=x.data_ptr(), ldb=K, c=part.data_ptr(), ldc=K, I=N, J=K, R=M,
                                                   c_split_stride=N * K)], False, False, S))
                row.append(f"cfg{cfg}: NT {nt:6.1f} NN {nn:6.1f} TN(S={S}) {tn:6.1f}")
            _lib.lib().singa_gemm_force_cfg(-1)
            fl = 2 * M * K * N / 1e6
            print(f"M {M:5d} N {N:4d} K {K:4d} ({fl / 1e3:5.2f} GFLOP; ideal {fl / 157.3:5.1f} us)  " + "  |  ".join(row), flush=True)
