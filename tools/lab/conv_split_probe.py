"""Lab: the SO(2)-convolution GEMMs (L = 4 block shapes) at a given edge count: forward (NT), dX (NN) and the weight
gradient (TN) as a function of its split count S, with the workgroup count each S gives (256 CUs, two 128 x 128 workgroups
per CU).     python tools/lab/conv_split_probe.py [E ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from singa_amd import ops

dev = "cuda"


def t_us(fn, n=6):
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


for E in [int(a) for a in sys.argv[1:]] or [15000, 99302]:
    for name, blocks in (("conv2", [(640, 560), (1024, 896), (768, 672)]), ("conv1", [(160, 992), (256, 1024), (192, 768)])):
        nin, nout = sum(b[0] for b in blocks), sum(b[1] for b in blocks)
        X = torch.randn(E, nin, device=dev)
        G = torch.randn(E, nout, device=dev)
        ws = [torch.randn(o, i, device=dev) * 0.05 for i, o in blocks]
        H, GX = torch.empty(E, nout, device=dev), torch.empty(E, nin, device=dev)
        flops = 2 * E * sum(i * o for i, o in blocks)
        tiles = sum(-(-o // 128) * -(-k // 128) for k, o in blocks)

        def nt():
            items, ai, ci = [], 0, 0
            for w, (k, o) in zip(ws, blocks):
                items.append(dict(a=X.data_ptr() + 4 * ai, lda=nin, b=w.data_ptr(), ldb=k, c=H.data_ptr() + 4 * ci, ldc=nout, I=E, J=o, R=k))
                ai, ci = ai + k, ci + o
            ops._gemm(items, True, True)

        def nn():
            items, ai, ci = [], 0, 0
            for w, (k, o) in zip(ws, blocks):
                items.append(dict(a=G.data_ptr() + 4 * ci, lda=nout, b=w.data_ptr(), ldb=k, c=GX.data_ptr() + 4 * ai, ldc=nin, I=E, J=k, R=o))
                ai, ci = ai + k, ci + o
            ops._gemm(items, True, False)

        def tn(S):
            part = torch.empty(S, sum(i * o for i, o in blocks), device=dev)
            tot = part.shape[1]

            def run():
                items, ai, ci, off = [], 0, 0, 0
                for w, (k, o) in zip(ws, blocks):
                    items.append(dict(a=G.data_ptr() + 4 * ci, lda=nout, b=X.data_ptr() + 4 * ai, ldb=nin, c=part.data_ptr() + 4 * off, ldc=k,
                                      I=o, J=k, R=E, c_split_stride=tot))
                    ai, ci, off = ai + k, ci + o, off + o * k
                ops._gemm(items, False, False, S)
                ops.colsum(part)
            return run

        t1, t2 = t_us(nt), t_us(nn)
        print(f"E={E} {name}: NT {t1:7.1f} us {flops / t1 / 1e6:6.1f} TF/s   NN {t2:7.1f} us {flops / t2 / 1e6:6.1f} TF/s   "
              f"TN tiles {tiles}, product S={ops._splits_for(E)}", flush=True)
        line = "    TN + column sum:"
        for S in (range(1, 33) if E < 40000 else (8, 12, 16, 20, 22, 24, 26, 28, 30, 32, 35, 38, 40, 44, 48, 52, 56, 59, 60, 64)):
            if S > max(1, E // 256):
                break
            t = t_us(tn(S))
            line += f"  S{S}({tiles * S / 256:.2f}): {t:6.1f}"
        print(line, flush=True)
