import torch
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
B = 128
cases = {  # name: (a shape, a transposed?, b shape, b transposed?)
 "QK^T  q[201,32] k[230,32]^T": ((B, 201, 32), False, (B, 230, 32), True),
 "QK^T  q[201,32] k[201,32]^T": ((B, 201, 32), False, (B, 201, 32), True),
 "PV    p[201,230] v[230,64]": ((B, 201, 230), False, (B, 230, 64), False),
 "PV    p[201,201] v[201,64]": ((B, 201, 201), False, (B, 201, 64), False),
 "dP    g[201,64] v[230,64]^T": ((B, 201, 64), False, (B, 230, 64), True),
 "dV    p[201,230]^T g[201,64]": ((B, 201, 230), True, (B, 201, 64), False),
 "dQ    ds[201,230] k[230,32]": ((B, 201, 230), False, (B, 230, 32), False),
 "dK    ds[201,230]^T q[201,32]": ((B, 201, 230), True, (B, 201, 32), False),
 "grouped q [4][6400,64]x[64,32]": ((4, 6400, 64), False, (4, 64, 32), False),
 "grouped dW [4][64,6400]x[6400,32]": ((4, 6400, 64), True, (4, 6400, 32), False),
 "so3lin [9][7360,512]x[512,16]": ((9, 7360, 512), False, (9, 512, 16), False),
 "so3lin dW [9][16,7360]x[7360,512]": ((9, 7360, 16), True, (9, 7360, 512), False),
}
for name, (sa, ta, sb, tb) in cases.items():
    a, b = torch.randn(*sa, device="cuda"), torch.randn(*sb, device="cuda")
    A = a.transpose(1, 2) if ta else a
    Bm = b.transpose(1, 2) if tb else b
    row = [name.ljust(36)]
    for lib in ("cublaslt", "cublas"):
        torch.backends.cuda.preferred_blas_library(lib)
        row.append(f"{lib} {t(lambda: torch.bmm(A, Bm)):6.1f}")
    torch.backends.cuda.preferred_blas_library("cublaslt")
    print("  ".join(row), flush=True)
