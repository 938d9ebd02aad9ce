"""Time the conv2-shaped NT / TN launches with a lab build of the library (SINGA_LAB_LIB=path) - lab probe."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import _lib
if os.environ.get("SINGA_LAB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SINGA_LAB_LIB"])
from singa_amd import ops
dev = "cuda"
E = 99302
blocks = [(640, 560), (1024, 896), (768, 672)]
nin, nout = sum(b[0] for b in blocks), sum(b[1] for b in blocks)
X = torch.randn(E, nin, device=dev); G = torch.randn(E, nout, device=dev)
ws = [torch.randn(o, i, device=dev) * 0.05 for i, o in blocks]
H = torch.empty(E, nout, device=dev)
flops = 2 * E * sum(i * o for i, o in blocks)

def nt():
    items, ai, ci = [], 0, 0
    for w, (k, o) in zip(ws, blocks):
        items.append(dict(a=X.data_ptr() + 4 * ai, lda=nin, b=w.data_ptr(), ldb=k, c=H.data_ptr() + 4 * ci, ldc=nout, I=E, J=o, R=k))
        ai, ci = ai + k, ci + o
    ops._gemm(items, True, True)

for _ in range(3):
    nt()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(8):
    nt()
b.record(); torch.cuda.synchronize()
t = a.elapsed_time(b) / 8 * 1e3
print(f"{os.environ.get('SINGA_LAB_LIB', 'product')}: NT {t:8.1f} us = {flops / t / 1e6:6.1f} TF/s")
