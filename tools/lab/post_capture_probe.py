"""Lab probe: does <op> still work after a HIP-graph capture + replay in the same process? (one op per process)"""
import sys
import torch

op = sys.argv[1]
big = int(sys.argv[2]) if len(sys.argv) > 2 else 1_300_000
pool = sys.argv[3] if len(sys.argv) > 3 else "small"
dev = "cuda"
x = torch.randn(1024, 1024, device=dev)
if pool == "big":   # capture a graph whose private pool holds GBs, like the training step
    w = [torch.randn(4096, 4096, device=dev) for _ in range(4)]
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    y = x @ x
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y = x @ x
    if pool == "big":
        keep = [torch.relu(a @ a) for a in w for _ in range(8)]
g.replay()
torch.cuda.synchronize()
k = torch.randint(0, big // 3, (big,), device=dev, dtype=torch.int64)
if op == "unique":
    r = torch.unique(k, sorted=True, return_inverse=True, return_counts=True)[0]
elif op == "sort":
    r = torch.sort(k)[0]
elif op == "argsort":
    r = torch.argsort(k, stable=True)
elif op == "nonzero":
    r = torch.nonzero(k % 3 == 0)
elif op == "mask":
    r = k[k % 3 == 0]
elif op == "bincount":
    r = torch.bincount(k, minlength=big)
torch.cuda.synchronize()
print(op, pool, "OK", tuple(r.shape))
