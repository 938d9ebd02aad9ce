import torch, time
from torch.profiler import profile, ProfilerActivity
g = torch.randn(56448, 448, device="cuda"); w = torch.randn(448, 512, device="cuda")
buf = torch.empty(56448, 1152, device="cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("mm -> new tensor      ", t(lambda: g @ w))
print("mm out=strided view   ", t(lambda: torch.mm(g, w, out=buf[:, 128:640])))
with profile(activities=[ProfilerActivity.CUDA]) as p:
    torch.mm(g, w, out=buf[:, 128:640]); torch.cuda.synchronize()
print([ (e.key[:60], e.count) for e in p.key_averages()])
ref = g @ w
print("equal", torch.equal(ref, buf[:, 128:640]))
