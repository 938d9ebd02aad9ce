"""Lab: do external event-record nodes (torch.cuda.Event(external=True) -> hipEventRecordWithFlags(hipEventRecordExternal)) give
per-kernel times inside a replayed HIP graph on this ROCm build?  A known ~0.1 ms kernel between two of them."""
import torch
x = torch.zeros(64 << 20, device="cuda")
y = torch.zeros(1 << 16, device="cuda")
e = [torch.cuda.Event(enable_timing=True, external=True) for _ in range(3)]
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        x.add_(1.0); y.add_(1.0)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y.add_(1.0)
    e[0].record()
    x.add_(1.0)
    e[1].record()
    y.add_(1.0)
    e[2].record()
for i in range(3):
    g.replay()
    torch.cuda.synchronize()
    print(f"replay {i}: big add {e[0].elapsed_time(e[1]) * 1e3:.1f} us, small add {e[1].elapsed_time(e[2]) * 1e3:.1f} us")
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); x.add_(1.0); b.record(); torch.cuda.synchronize()
print(f"eager big add between plain events: {a.elapsed_time(b) * 1e3:.1f} us")
