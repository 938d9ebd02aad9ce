"""Mid-sized library GEMMs of the config-3 step that looked slow in tools/lab/mm_sites.py (e.g. 149 us for a 256 x 3807 x 256
accumulate): both libraries, in isolation.  Result: 7-30 us each in isolation - the profiler's per-op times were inflated
by the ligand encoder running concurrently on the second stream.  A measured per-shape library choice (both libraries timed
once per shape at first sight) was tried on top of the static rule of ops._blas and changed nothing (182.0 vs 181.4 ms)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = "cuda"


def timeit(fn, n=20):
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


R = lambda *s: torch.randn(*s, device=dev)
cases = []
x, y, o = R(3807, 256), R(3807, 256), R(256, 256)
cases.append(("addmm [256,256] += [256,3807]@[3807,256]", lambda: torch.addmm(o, x.t(), y)))
cases.append(("mm    [256,3807]@[3807,256]", lambda: x.t() @ y))
a, w = R(4480, 128), R(256, 128)
cases.append(("mm    [4480,128]@[128,256] (w.t())", lambda: a @ w.t()))
w2 = R(128, 256)
cases.append(("mm    [4480,128]@[128,256] (plain)", lambda: a @ w2))
q, k = R(4, 3807, 32), R(4, 32, 64)
cases.append(("bmm   [4,3807,32]@[4,32,64]", lambda: torch.bmm(q, k)))
qs = R(3807, 4, 32).transpose(0, 1)
cases.append(("bmm   [4,3807,32](strided)@[4,32,64]", lambda: torch.bmm(qs, k)))
e, v = R(15228, 32), R(32, 32)
cases.append(("mm    [15228,32]@[32,32]", lambda: e @ v))
g4, h4, o4 = R(4, 3807, 64), R(4, 3807, 64), R(4, 64, 64)
cases.append(("baddbmm [4,64,64] += [4,64,3807]@[4,3807,64]", lambda: torch.baddbmm(o4, g4.transpose(1, 2), h4)))
f, wf = R(3807, 256), R(256, 256)
cases.append(("mm    [3807,256]@[256,256]", lambda: f @ wf.t()))
b16, c16 = R(16, 3752, 128), R(16, 3752, 256)
cases.append(("bmm   [16,128,3752]@[16,3752,256]", lambda: torch.bmm(b16.transpose(1, 2), c16)))
for name, fn in cases:
    t = {}
    for lib in ("cublas", "cublaslt"):
        torch.backends.cuda.preferred_blas_library(lib)
        t[lib] = timeit(fn)
    torch.backends.cuda.preferred_blas_library("cublaslt")
    print(f"{name:52s} rocBLAS {t['cublas']:7.1f} us   hipBLASLt {t['cublaslt']:7.1f} us")
