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
    for it in range(2):                                     # second step reuses the static buckets
        model.zero_grad(set_to_none=True)
        loss = ((model(data[lo:hi]) - tgt[lo:hi]) ** 2).mean()
        loss.backward()
        if it == 0:
            red.reduce()
        else:                                               # the three phases the step engine issues separately
            red.flatten(fresh=True)
            red.allreduce()
            red.unflatten()
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


def _worker_unequal(rank, world, port, out):
    """Unequal shards (7 + 3 rows): token-weighted combination = the global-batch mean gradient; and the collective
    host-side decision helper."""
    sys.path.insert(0, ROOT)
    from singa_amd import dp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(7)
    model = Toy()
    red = dp.GradAllReducer(model)
    data = torch.randn(10, 8, generator=torch.Generator().manual_seed(1))
    tgt = torch.randn(10, 3, generator=torch.Generator().manual_seed(2))
    lo, hi = (0, 7) if rank == 0 else (7, 10)
    red.set_shard_weight(hi - lo, 10)
    loss = ((model(data[lo:hi]) - tgt[lo:hi]) ** 2).mean()
    loss.backward()
    red.reduce()
    flags = [red.any_rank(rank == 1), red.any_rank(False), red.any_rank(True)]
    if rank == 0:
        torch.save({"grads": {n: p.grad for n, p in model.named_parameters()}, "flags": flags}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_unequal_shards_are_token_weighted(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker_unequal, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    torch.manual_seed(7)
    model = Toy()
    data = torch.randn(10, 8, generator=torch.Generator().manual_seed(1))
    tgt = torch.randn(10, 3, generator=torch.Generator().manual_seed(2))
    ((model(data) - tgt) ** 2).mean().backward()            # the global-batch mean loss
    for n, p in model.named_parameters():
        if "unused" not in n:
            assert torch.allclose(got["grads"][n], p.grad, atol=1e-6), n
    assert got["flags"] == [True, False, True]


def singa_grad_free_names(model):
    """the 90 parameter tensors of SINGA that never receive a gradient."""
    out = set()
    for n, _ in model.named_parameters():
        parts = n.split(".")
        if n.startswith("embedding.sphere_embedding"):
            out.add(n)
        elif parts[:2] in (["model", "encoder"], ["model", "encoder2"]) and parts[2] in ("out", "layer_norm"):
            out.add(n)
        elif "pos_ffn.batch_norm" in n:
            out.add(n)
        elif parts[:3] == ["model", "encoder2", "layers"] and parts[3] not in ("2", "5") and parts[4] in (
                "proj", "cross_attn", "layer_norm"):
            out.add(n)
    return out


def _worker_singa(rank, world, port, out):
    sys.path.insert(0, ROOT)
    from singa_amd import dp
    from singa_amd.config import load_config
    from singa_amd.model.GAN import SINGA
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(2022)
    model = SINGA(load_config(lmax=6), device="cpu")         # parameter container: the forward is not run here
    red = dp.GradAllReducer(model)
    red.check_same_init()
    free = singa_grad_free_names(model)
    gen = torch.Generator().manual_seed(100 + rank)
    for n, p in model.named_parameters():
        p.grad = None if n in free else torch.randn(p.shape, generator=gen)
    mine = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    red.reduce()
    if rank == 0:
        order = [id(p) for b in red.buckets for p in b]
        names = {id(p): n for n, p in model.named_parameters()}
        torch.save({"n_tensors": len(order), "payload": red.payload_bytes, "nb": len(red.buckets),
                    "first": names[order[0]], "last": names[order[-1]], "bucket_order": [names[i] for i in order],
                    "sample": {n: (model.get_parameter(n).grad.clone(), mine[n]) for n in
                               ("model.projection.weight", "embedding.blocks.0.ga.alpha_dot")}}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_singa_parameter_set_buckets(tmp_path):
    """The real SINGA parameter list at L = 6 through the reducer: 634 gradient tensors = 25,164,960 floats =
    100.66 MB per step, the 90 gradient-free tensors excluded statically, buckets in reverse
    execution order (vocabulary projection / decoder first, equivariant blocks last), averaged values exact."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker_singa, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    assert got["n_tensors"] == 634 and got["payload"] == 25_164_960 * 4
    assert got["nb"] >= 3
    assert got["first"] == "model.projection.weight" and got["last"].startswith("embedding.")
    sys.path.insert(0, ROOT)
    from singa_amd.config import load_config
    from singa_amd.model.GAN import SINGA
    model = SINGA(load_config(lmax=6), device="cpu")
    decl = [n for n, _ in model.named_parameters()]
    assert len(decl) == 724 and len(singa_grad_free_names(model)) == 90
    assert got["bucket_order"] == [n for n in reversed(decl) if n not in singa_grad_free_names(model)]
    # rank 1's gradients are reproducible here: the reduced value is the mean of the two ranks'
    gen = torch.Generator().manual_seed(101)
    other = {}
    free = singa_grad_free_names(model)
    for n, p in model.named_parameters():
        if n not in free:
            other[n] = torch.randn(p.shape, generator=gen)
    for n, (red, mine) in got["sample"].items():
        assert torch.allclose(red, 0.5 * (mine + other[n]), atol=1e-6), n


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


def _worker_agree(rank, world, port, out):
    sys.path.insert(0, ROOT)
    from singa_amd import dp
    from singa_amd.engine import TrainStep
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    red = dp.GradAllReducer(Toy())
    # ragged shards: every rank proposes its own sizes; all derive ONE size class / capacity set from the element-wise MAX
    sizes = [(5000, 400, 10000, 900, 1300), (5300, 380, 10600, 860, 1400)][rank]
    agreed = red.max_ints(sizes)
    eng = TrainStep.__new__(TrainStep)
    eng._base, eng.growth = None, 1.04
    c0, caps0 = eng._class_caps(tuple(agreed))
    later = red.max_ints([(5100, 410, 10100, 905, 1290), (4900, 395, 10900, 880, 1310)][rank])
    c1, caps1 = eng._class_caps(tuple(later))
    torch.save({"agreed": agreed, "class": (c0, caps0, c1, caps1), "flag": red.any_rank(rank == 1)}, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_agree_on_padded_sizes(tmp_path):
    """TrainStep._stage under data parallelism: the size class and capacities a batch is padded to come from the MAX of
    the ranks' sizes, so ragged shards replay the same capture signature on every rank (no lone re-captures)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "a")
    mp.spawn(_worker_agree, args=(2, port, out), nprocs=2, join=True)
    a, b = torch.load(out + ".0"), torch.load(out + ".1")
    assert a["agreed"] == b["agreed"] == [5300, 400, 10600, 900, 1400]
    assert a["class"] == b["class"] and a["flag"] is True and b["flag"] is True
    c0, caps0, c1, caps1 = a["class"]
    assert c0 == 0 and all(cap >= v for cap, v in zip(caps0, a["agreed"]))
    assert all(cap >= v for cap, v in zip(caps1, (5100, 410, 10900, 905, 1310)))


class ToyTwoPart(torch.nn.Module):
    """embedding -> head, with SINGA's backward_phases() contract (head first, then embedding)."""

    def __init__(self):
        super().__init__()
        self.embedding = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 12))
        self.model = torch.nn.Sequential(torch.nn.Linear(12, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))
        self.unused = torch.nn.Parameter(torch.zeros(3))

    def backward_phases(self):
        return [list(self.model.parameters()) + [self.unused], list(self.embedding.parameters())]

    def forward(self, x, boundary=None):
        e = self.embedding(x)
        if boundary is not None:
            d = e.detach().requires_grad_(True)
            boundary.append((e, d))
            e = d
        return self.model(e)


def _worker_phases(rank, world, port, out):
    sys.path.insert(0, ROOT)
    from singa_amd import dp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(7)
    model = ToyTwoPart()
    red = dp.GradAllReducer(model, bucket_mb=0.0001, phases=True)
    assert red.phases is not None and len(red.phases) == 2
    data = torch.randn(10, 8, generator=torch.Generator().manual_seed(1))
    tgt = torch.randn(10, 3, generator=torch.Generator().manual_seed(2))
    lo, hi = dp.shard_range(10, rank, world)
    for it in range(2):
        model.zero_grad(set_to_none=True)
        if it == 0:                                         # first pass: one backward, the buckets are laid out from it
            ((model(data[lo:hi]) - tgt[lo:hi]) ** 2).mean().backward()
            red.reduce()
        else:                                               # the engine's two-phase order: head, launch, embedding, launch, wait
            boundary = []
            ((model(data[lo:hi], boundary=boundary) - tgt[lo:hi]) ** 2).mean().backward()
            assert all(p.grad is None for p in model.embedding.parameters())
            red.flatten(fresh=True, phase=0)
            red.launch(phase=0)
            torch.autograd.backward([x for x, _ in boundary], [d.grad for _, d in boundary])
            red.reduce(skip_flatten_of=(0,))
    if rank == 0:
        torch.save({"grads": {n: p.grad for n, p in model.named_parameters()}, "phase": red.bucket_phase}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_phase_backward_reduces_the_same_gradients(tmp_path):
    """GradAllReducer with phases: no bucket mixes the two parameter groups, and flatten / launch of phase 0 between the two
    parts of the backward pass gives the gradients of one backward() + reduce()."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "p0.pt")
    mp.spawn(_worker_phases, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    sys.path.insert(0, ROOT)
    from singa_amd import dp
    torch.manual_seed(7)
    model = ToyTwoPart()
    data = torch.randn(10, 8, generator=torch.Generator().manual_seed(1))
    tgt = torch.randn(10, 3, generator=torch.Generator().manual_seed(2))
    loss = sum(((model(data[slice(*dp.shard_range(10, r, 2))]) - tgt[slice(*dp.shard_range(10, r, 2))]) ** 2).mean()
               for r in range(2)) / 2
    loss.backward()
    assert sorted(set(got["phase"])) == [0, 1] and got["phase"] == sorted(got["phase"])
    for n, p in model.named_parameters():
        if n == "unused":
            assert got["grads"][n] is None
        else:
            assert torch.allclose(got["grads"][n], p.grad, atol=1e-6), n
