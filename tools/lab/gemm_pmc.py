"""One SO(2)-conv-2 sized product per direction, a few launches each, for `rocprofv3 --pmc` (lab probe)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import ops
dev = "cuda"
E = 99302
blocks = [(640, 560), (1024, 896), (768, 672)]
nin, nout = sum(b[0] for b in blocks), sum(b[1] for b in blocks)
X = torch.randn(E, nin, device=dev); G = torch.randn(E, nout, device=dev)
ws = [torch.randn(o, i, device=dev) * 0.05 for i, o in blocks]
H = torch.empty(E, nout, device=dev); gX = torch.empty_like(X)
tot = sum(i * o for i, o in blocks); S = 49
part = torch.empty(S, tot, device=dev)
for _ in range(4):
    items, ai, ci = [], 0, 0
    for w, (k, o) in zip(ws, blocks):
        items.append(dict(a=X.data_ptr() + 4 * ai, lda=nin, b=w.data_ptr(), ldb=k, c=H.data_ptr() + 4 * ci, ldc=nout, I=E, J=o, R=k))
        ai, ci = ai + k, ci + o
    ops._gemm(items, True, True)
    items, ai, ci, off = [], 0, 0, 0
    for (k, o) in blocks:
        items.append(dict(a=G.data_ptr() + 4 * ci, lda=nout, b=X.data_ptr() + 4 * ai, ldb=nin, c=part.data_ptr() + 4 * off, ldc=k,
                          I=o, J=k, R=E, c_split_stride=tot))
        ai, ci, off = ai + k, ci + o, off + k * o
    ops._gemm(items, False, False, S)
    for w, (k, o) in zip(ws, blocks):
        Y = X[:, :k].contiguous() @ w.t()
torch.cuda.synchronize()
