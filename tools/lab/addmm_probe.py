import torch
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for M, K, N in [(6400, 256, 1024), (6432, 256, 1024), (6400, 1024, 256), (6400, 256, 256), (6432, 256, 128), (960, 256, 1024), (960, 256, 256),
                (374578, 64, 64), (374578, 64, 32), (374578, 32, 32), (56448, 384, 336), (56448, 96, 736), (56448, 16, 192), (7360, 16, 512)]:
    x, w, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda"), torch.randn(N, device="cuda")
    row = [f"{M}x{K}x{N}".ljust(16)]
    for lib in ("cublaslt", "cublas"):
        torch.backends.cuda.preferred_blas_library(lib)
        row.append(f"{lib}: addmm {t(lambda: torch.addmm(b, x, w.t())):6.1f}  mm {t(lambda: x @ w.t()):6.1f}")
    torch.backends.cuda.preferred_blas_library("cublaslt")
    print("   ".join(row), flush=True)
