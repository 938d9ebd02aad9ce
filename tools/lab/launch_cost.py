"""Lab: cost of one tiny dependent kernel inside a replayed HIP graph (and eager), for several payload sizes."""
import time, torch
dev = "cuda"
for n in (256, 65536, 1 << 20, 1 << 22):
    x = torch.zeros(n, device=dev)
    def body(k=1000):
        y = x
        for _ in range(k):
            y = y + 1.0
        return y
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body(10)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = body()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize(); t_graph = (time.perf_counter() - t0) / 10 / 1000 * 1e6
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): body()
    torch.cuda.synchronize(); t_eager = (time.perf_counter() - t0) / 3 / 1000 * 1e6
    print(f"n={n:8d} floats: graph replay {t_graph:6.2f} us per kernel, eager {t_eager:6.2f} us per kernel", flush=True)
