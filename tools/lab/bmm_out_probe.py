import torch
from torch.profiler import profile, ProfilerActivity
N, K, C, O = 7360, 9, 512, 16
x = torch.randn(N, K, C, device="cuda"); w = torch.randn(K, O, C, device="cuda")
out = torch.empty(N, K, O, device="cuda")
ref = torch.bmm(x.transpose(0, 1), w.transpose(1, 2)).transpose(0, 1).contiguous()
def kernels(fn):
    fn(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as p:
        fn(); torch.cuda.synchronize()
    return [(e.key[:50], e.count, round(e.device_time_total)) for e in p.key_averages() if e.device_time_total > 0]
print("bmm + transpose copy:", kernels(lambda: torch.bmm(x.transpose(0, 1), w.transpose(1, 2)).transpose(0, 1).contiguous()))
print("bmm out=view        :", kernels(lambda: torch.bmm(x.transpose(0, 1), w.transpose(1, 2), out=out.transpose(0, 1))))
print("equal", torch.allclose(out, ref, atol=1e-4))
g = torch.randn(N, K, O, device="cuda")
gx = torch.empty(N, K, C, device="cuda")
print("dX out=view         :", kernels(lambda: torch.bmm(g.transpose(0, 1), w, out=gx.transpose(0, 1))))
print("dW                  :", kernels(lambda: torch.bmm(g.transpose(0, 1).transpose(1, 2), x.transpose(0, 1))))
print("equal dX", torch.allclose(gx, torch.bmm(g.transpose(0, 1).contiguous(), w).transpose(0, 1), atol=1e-3))
