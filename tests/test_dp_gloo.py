"""N > 1 path on CPU: two gloo ranks average gradients through singa_amd.dp.GradAllReducer exactly as the RCCL path
does on GPUs (one process per rank, same-seed init, shard by unit, bucketed all-reduce, unused params excluded)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(8, 16)
        self.unused = torch.nn.Linear(4, 4)          # never receives a gradient (like SURVEY Q10 tensors)
        self.b = torch.nn.Linear(16, 3)

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    from singa_amd import dp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(7)
    model = Toy()
    red = dp.GradAllReducer(model, bucket_mb=0.0001)        # tiny buckets -> several of them
    red.check_same_init()
    data = torch.randn(10, 8, generator=torch.Generator().manual_seed(1))
    tgt = torch.randn(10, 3, generator=torch.Generator().manual_seed(2))
    lo, hi = dp.shard_range(10, rank, world)
    for _ in range(2):                                      # second step reuses the static buckets
        model.zero_grad(set_to_none=True)
        loss = ((model(data[lo:hi]) - tgt[lo:hi]) ** 2).mean()
        loss.backward()
        red.reduce()
    if rank == 0:
        torch.save({"grads": {n: p.grad for n, p in model.named_parameters()}, "nb": len(red.buckets),
                    "payload": red.payload_bytes}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_average(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    sys.path.insert(0, ROOT)
    from singa_amd import dp
    torch.manual_seed(7)
    model = Toy()
    data = torch.randn(10, 8, generator=torch.Generator().manual_seed(1))
    tgt = torch.randn(10, 3, generator=torch.Generator().manual_seed(2))
    # equal shards -> the average of the per-rank mean losses is the global mean loss
    loss = sum(((model(data[slice(*dp.shard_range(10, r, 2))]) - tgt[slice(*dp.shard_range(10, r, 2))]) ** 2).mean()
               for r in range(2)) / 2
    loss.backward()
    assert got["nb"] > 1 and got["payload"] == sum(p.numel() * 4 for n, p in model.named_parameters() if "unused" not in n)
    for n, p in model.named_parameters():
        if "unused" in n:
            assert got["grads"][n] is None
        else:
            assert torch.allclose(got["grads"][n], p.grad, atol=1e-6), n


def test_shard_range_covers_everything():
    from singa_amd import dp
    for n in (1, 7, 32, 129):
        for w in (1, 2, 3, 8):
            parts = [dp.shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def test_shard_ranges_by_cost():
    from singa_amd.dp import shard_range, shard_ranges_by_cost
    costs = [10, 10, 10, 10, 40, 40, 10, 10, 10, 10]                  # two heavy graphs in the middle
    r = shard_ranges_by_cost(costs, 4)
    assert r[0][0] == 0 and r[-1][1] == len(costs) and all(a[1] == b[0] for a, b in zip(r, r[1:]))
    loads = [sum(costs[lo:hi]) for lo, hi in r]
    by_count = [sum(costs[slice(*shard_range(len(costs), k, 4))]) for k in range(4)]
    assert max(loads) <= max(by_count) and all(hi > lo for lo, hi in r)
    assert shard_ranges_by_cost([5, 5, 5], 3) == [(0, 1), (1, 2), (2, 3)]
    assert shard_ranges_by_cost([1, 1, 1, 1], 1) == [(0, 4)]
    one_heavy = shard_ranges_by_cost([100, 1, 1, 1], 4)
    assert one_heavy == [(0, 1), (1, 2), (2, 3), (3, 4)]
