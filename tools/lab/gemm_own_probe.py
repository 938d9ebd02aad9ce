"""Own f32 MFMA GEMM (singa_gemm_f32) vs the BLAS libraries on the shapes of the config-3 step (lab probe, not a test)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import ops

dev = "cuda"


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3       # us


E = int(sys.argv[1]) if len(sys.argv) > 1 else 99302
print(f"E = {E}")
for name, blocks in (("conv1 L4", [(160, 992), (256, 1024), (192, 768)]), ("conv2 L4", [(640, 560), (1024, 896), (768, 672)])):
    nin, nout = sum(b[0] for b in blocks), sum(b[1] for b in blocks)
    X = torch.randn(E, nin, device=dev)
    ws = [torch.randn(o, i, device=dev) * 0.05 for i, o in blocks]
    b0 = torch.randn(blocks[0][1], device=dev)
    G = torch.randn(E, nout, device=dev)
    flops = 2 * E * sum(i * o for i, o in blocks)
    for own in (True, False):
        ops.USE_OWN_GEMM = own
        Xr = X.clone().requires_grad_(True)
        wr = [w.clone().requires_grad_(True) for w in ws]
        br = b0.clone().requires_grad_(True)

        def fwd():
            return ops.so2_linear3(Xr, wr[0], br, wr[1], wr[2], blocks[0][0], blocks[1][0])
        t_f = timeit(lambda: fwd())
        hs = fwd()
        gs = list(G.split([b[1] for b in blocks], 1))

        def bwd():
            torch.autograd.grad(hs, [Xr] + wr + [br], gs, retain_graph=True)
        t_b = timeit(bwd)
        print(f"{name} {'own' if own else 'lib'}: fwd {t_f:8.1f} us = {flops / t_f / 1e6:6.1f} TF/s   bwd (dX+dW+db) {t_b:8.1f} us = {2 * flops / t_b / 1e6:6.1f} TF/s")
    # the three directions separately on the own kernel
    ops.USE_OWN_GEMM = True
    H = torch.empty(E, nout, device=dev)

    def nt():
        items, ai, ci = [], 0, 0
        for w, (k, o) in zip(ws, blocks):
            items.append(dict(a=X.data_ptr() + 4 * ai, lda=nin, b=w.data_ptr(), ldb=k, c=H.data_ptr() + 4 * ci, ldc=nout, I=E, J=o, R=k))
            ai, ci = ai + k, ci + o
        ops._gemm(items, True, True)
    gX = torch.empty_like(X)

    def nn():
        items, ai, ci = [], 0, 0
        for w, (k, o) in zip(ws, blocks):
            items.append(dict(a=G.data_ptr() + 4 * ci, lda=nout, b=w.data_ptr(), ldb=k, c=gX.data_ptr() + 4 * ai, ldc=nin, I=E, J=k, R=o))
            ai, ci = ai + k, ci + o
        ops._gemm(items, True, False)
    tot = sum(i * o for i, o in blocks)
    for rows in (1024, 2048, 4096, 8192):
        S = max(1, min(64, -(-E // rows)))
        part = torch.empty(S, tot, device=dev)

        def tn():
            items, ai, ci, off = [], 0, 0, 0
            for (k, o) in blocks:
                items.append(dict(a=G.data_ptr() + 4 * ci, lda=nout, b=X.data_ptr() + 4 * ai, ldb=nin, c=part.data_ptr() + 4 * off, ldc=k,
                                  I=o, J=k, R=E, c_split_stride=tot))
                ai, ci, off = ai + k, ci + o, off + k * o
            ops._gemm(items, False, False, S)
        t = timeit(tn)
        print(f"   TN  split rows {rows:5d} (S={S:2d}): {t:8.1f} us = {flops / t / 1e6:6.1f} TF/s (+ colsum {timeit(lambda: ops.colsum(part)):6.1f} us)")
    t = timeit(nt)
    print(f"   NT: {t:8.1f} us = {flops / t / 1e6:6.1f} TF/s")
    t = timeit(nn)
    print(f"   NN: {t:8.1f} us = {flops / t / 1e6:6.1f} TF/s")

N = 49267
for name, L, cin, cout in (("ffn lin1 L4", 4, 16, 512), ("ffn lin2 L4", 4, 512, 16), ("proj L4", 4, 112, 16)):
    K = (L + 1) ** 2
    x = torch.randn(N, K, cin, device=dev)
    w = torch.randn(L + 1, cout, cin, device=dev)
    b = torch.randn(cout, device=dev)
    g = torch.randn(N, K, cout, device=dev)
    for own in (True, False):
        ops.USE_OWN_GEMM = own
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        t_f = timeit(lambda: ops.so3_linear(xr, wr, br, L))
        y = ops.so3_linear(xr, wr, br, L)
        t_b = timeit(lambda: torch.autograd.grad(y, [xr, wr, br], g, retain_graph=True))
        print(f"{name} {'own' if own else 'lib'}: fwd {t_f:8.1f} us   bwd {t_b:8.1f} us")
