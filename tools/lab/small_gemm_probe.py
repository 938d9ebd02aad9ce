"""Lab: the transformer's Linear shapes at the 17-graph shard's row counts: own GEMM (ops.linear forward / dX / dW) next to
torch.nn.functional.linear (hipBLASLt), us per call averaged over back-to-back launches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import ops

dev = "cuda"


def t_us(fn, n=50):
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


for M in (6499, 512, 3400):
    for K, N in ((256, 256), (256, 1024), (1024, 256), (256, 768)):
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.05
        b = torch.zeros(N, device=dev)
        g = torch.randn(M, N, device=dev)
        with torch.no_grad():
            t_own = t_us(lambda: ops.linear(x, w, b))
            t_lib = t_us(lambda: torch.nn.functional.linear(x, w, b))
            gx = torch.empty(M, K, device=dev)
            t_dx = t_us(lambda: ops._gemm([dict(a=g.data_ptr(), lda=N, b=w.data_ptr(), ldb=K, c=gx.data_ptr(), ldc=K, I=M, J=K, R=N)], True, False))
            t_dx_lib = t_us(lambda: g @ w)
            t_dw_lib = t_us(lambda: g.t() @ x)
        fl = 2 * M * K * N
        print(f"[{M:5d} x {K:4d}] -> {N:4d}: fwd own {t_own:6.1f} us ({fl / t_own / 1e6:5.1f} TF/s)  lib {t_lib:6.1f} us   "
              f"dX own {t_dx:6.1f}  lib {t_dx_lib:6.1f}   dW lib {t_dw_lib:6.1f}", flush=True)
