"""Data-parallel gradient averaging: one process per GPU, graphs sharded across ranks, one bucketed all-reduce of
the generator's gradients per step over RCCL/xGMI (torch.distributed backend "nccl" on ROCm; "gloo" in CPU tests).

The reference has no distributed code; the path shards by graph with no other exchange
(SURVEY.md §8e).  Parameters that never receive a gradient (Q10: 90 tensors) are excluded statically from the
buckets after the first backward, so no unused-parameter scan runs per step.  Buckets follow reverse execution
order (decoder -> encoders -> equivariant blocks) and are launched asynchronously so that the first reductions
overlap the flattening of the later ones.
"""
import torch
import torch.distributed as dist


class GradAllReducer:
    def __init__(self, module, bucket_mb=32.0, group=None, always=False):
        self.module, self.group = module, group
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self.buckets = None
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # always: run the collectives on a one-rank group too (self-test of the RCCL path on a one-GPU box)
        self.active = self.world > 1 or (always and dist.is_initialized())

    def _build(self):
        params = [p for p in reversed(list(self.module.parameters())) if p.grad is not None]
        self.buckets, cur, size = [], [], 0
        for p in params:
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= self.bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self.flat = [torch.empty(sum(p.numel() for p in b), dtype=b[0].dtype, device=b[0].device) for b in self.buckets]

    def check_same_init(self):
        """Same-seed initialisation replaces a parameter broadcast; verify it with one checksum exchange."""
        if not self.active:
            return
        s = torch.stack([p.detach().double().sum() for p in self.module.parameters()]).sum().reshape(1)
        lo, hi = s.clone(), s.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        assert float(hi - lo) == 0.0, "ranks were not initialised with identical parameters"

    def _side_group(self):
        """A gloo group for the small host-side agreements (no device synchronisation involved)."""
        if getattr(self, "_cpu_group", None) is None:
            self._cpu_group = (self.group if dist.get_backend(self.group) == "gloo" else dist.new_group(backend="gloo"))
        return self._cpu_group

    def any_rank(self, flag):
        """True on every rank if `flag` is true on any (host-side decision that all ranks must take together, e.g.
        re-capturing the step: its warm-up issues collectives)."""
        if not self.active:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self._side_group())
        return bool(int(t))

    def max_ints(self, values):
        """Element-wise MAX of a short list of integers over the ranks (host-side agreement: the sizes a padded batch is
        staged with, see TrainStep._stage).  Every rank must call it the same number of times."""
        if not self.active:
            return [int(v) for v in values]
        t = torch.tensor([int(v) for v in values], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self._side_group())
        return t.tolist()

    def set_shard_weight(self, n_local, n_global):
        """Shards of different sizes (batch % world != 0, or `shard_ranges_by_cost`): every rank's CrossEntropy is a mean
        over ITS tokens, so the global-batch gradient is sum_r (tokens_r / tokens) grad_r, not the plain average.  All
        graphs carry the same number of target tokens (tgt_len), hence the weight n_local / n_global."""
        self.weight = float(n_local) / float(n_global)

    weight = None

    def reduce(self):
        """Combine .grad over ranks in place: the mean over ranks, or - after `set_shard_weight` - the token-weighted sum.
        Call after backward(), before the optimizer step."""
        if not self.active:
            return
        self.flatten()
        self.allreduce()
        self.unflatten()

    # The three phases of reduce().  The step engine captures flatten() at the end of its forward+backward HIP graph and
    # unflatten() at the start of its optimizer graph, so that only the RCCL calls themselves run between the two replays.
    def flatten(self, fresh=False):
        """gradients -> flat bucket buffers, scaled by the combination weight.  fresh: look the .grad tensors up again."""
        if not self.active:
            return
        if self.buckets is None:
            self._build()
        pre = self.weight if self.weight is not None else 1.0 / self.world
        # the per-tensor views of the flat buffers never change; the list of .grad tensors is rebuilt unless the caller
        # vouches (grads_token) that they are the same objects as last time - building two 634-element lists per bucket
        # cost 2 ms of host time per step
        if self._views is None:
            self._views = [list(flat.split([p.numel() for p in bucket])) for flat, bucket in zip(self.flat, self.buckets)]
        if fresh or self.grads_token is None or self.grads_token != self._grads_for:
            self._grads = [[p.grad.reshape(-1) for p in bucket] for bucket in self.buckets]
            self._grads_for = None if fresh else self.grads_token
        for flat, views, grads in zip(self.flat, self._views, self._grads):
            torch._foreach_copy_(views, grads)
            if pre != 1.0:
                flat.mul_(pre)

    def allreduce(self):
        """SUM all-reduce of the flat buffers (asynchronous launches, then the current stream waits for all of them)."""
        if not self.active:
            return
        works = [dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for flat in self.flat]
        for w in works:
            w.wait()

    def unflatten(self):
        """flat bucket buffers -> the .grad tensors flatten() read."""
        if not self.active:
            return
        for views, grads in zip(self._views, self._grads):
            torch._foreach_copy_(grads, views)

    # set by the caller when the .grad tensors are known to be the same objects on every call with the same token (the
    # step engine: one token per captured graph); None = look them up every time
    grads_token = None
    _views = _grads = _grads_for = None

    @property
    def payload_bytes(self):
        return 0 if self.buckets is None else sum(f.numel() * f.element_size() for f in self.flat)


def shard_range(n_items, rank, world):
    """Contiguous balanced shard [lo, hi) of n_items units for `rank`."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_ranges_by_cost(costs, world):
    """Contiguous shards [lo, hi) per rank balanced by a per-graph cost (the number of edges
    E_pp + E_ll + 2 E_x drives the step time, not the graph count).  Greedy prefix split: rank r ends where the running
    cost passes (r+1)/world of the total; every rank gets at least one graph when there are enough."""
    n = len(costs)
    total = float(sum(costs))
    bounds, acc, r = [0], 0.0, 1
    for i, c in enumerate(costs):
        acc += float(c)
        if r < world and acc >= total * r / world and n - (i + 1) >= world - r:
            bounds.append(i + 1)
            r += 1
    while len(bounds) < world:
        bounds.append(max(bounds[-1], n - (world - len(bounds))))
    bounds.append(n)
    return [(bounds[k], bounds[k + 1]) for k in range(world)]
