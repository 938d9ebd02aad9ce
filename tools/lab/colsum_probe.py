"""Lab: in-graph cost of ops.colsum for the shapes of the step (100 dependent calls per graph)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__; __graft_entry__.build()
from singa_amd import ops
for M, n in [(6400, 256), (6432, 1024), (960, 256), (56448, 16), (56448, 736), (374578, 64), (64, 4096), (2048, 512), (7360, 144)]:
    x = torch.randn(M, n, device="cuda")
    def body():
        acc = None
        for _ in range(100):
            c = ops.colsum(x)
            acc = c if acc is None else acc + c * 0
        return acc
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): body()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): out = body()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5 / 100 * 1e6
    ref = x.double().sum(0)
    print(f"[{M:6d},{n:5d}]  {dt:7.2f} us per colsum (incl. one tiny add)   {M * n * 4 / dt / 1e3:7.1f} GB/s   err {float((ops.colsum(x).double() - ref).abs().max()):.2e}", flush=True)
