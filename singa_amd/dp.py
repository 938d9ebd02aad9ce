"""Data-parallel gradient averaging: one process per GPU, graphs sharded across ranks, one bucketed all-reduce of
the generator's gradients per step over RCCL/xGMI (torch.distributed backend "nccl" on ROCm; "gloo" in CPU tests).

The reference has no distributed code; the path shards by graph with no other exchange
(SURVEY.md §8e).  Parameters that never receive a gradient (Q10: 90 tensors) are excluded statically from the
buckets after the first backward, so no unused-parameter scan runs per step.  Buckets follow reverse execution
order (decoder -> encoders -> equivariant blocks) and are launched asynchronously.  With `phases` (the step engine's
two-phase backward: transformer first, equivariant embedding second) no bucket mixes parameters of two phases, and the
buckets of phase 0 are reduced while phase 1 of the backward pass still computes.
"""
import torch
import torch.distributed as dist


class GradAllReducer:
    def __init__(self, module, bucket_mb=32.0, group=None, always=False, phases=None):
        """phases: lists of parameters in the order their gradients become final during the backward pass; flatten / launch
        take a phase index.  True: module.backward_phases() (SINGA: [transformer, embedding] - the step engine's two-phase
        backward with the first all-reduce in flight during the second part).  Default (None / False): ONE phase, all
        buckets reduced after the backward pass - the overlapped order is opt-in (`bench.py / train.py
        --allreduce-overlap`) until a run on several RCCL ranks has shown it to give the same parameters
        (bench.py --gpus N measures and checks exactly that, `allreduce_overlap` in its line)."""
        self.module, self.group = module, group
        if phases is True:
            phases = module.backward_phases()
        self.phases = [list(ps) for ps in phases] if phases else None
        if self.phases is not None:
            ids = [id(p) for ps in self.phases for p in ps]
            assert len(ids) == len(set(ids)) and set(ids) == {id(p) for p in module.parameters()}, \
                "phases must partition the module's parameters"
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self.buckets = None
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # always: run the collectives on a one-rank group too (self-test of the RCCL path on a one-GPU box)
        self.active = self.world > 1 or (always and dist.is_initialized())

    def _build(self):
        """Buckets of every phase, from the parameters that own a gradient NOW (call it after a complete backward pass)."""
        groups = self.phases if self.phases is not None else [list(self.module.parameters())]
        self.buckets, self.bucket_phase = [], []
        for k, group in enumerate(groups):
            cur, size = [], 0
            for p in reversed(group):
                if p.grad is None:
                    continue
                cur.append(p)
                size += p.numel() * p.element_size()
                if size >= self.bucket_bytes:
                    self.buckets.append(cur)
                    self.bucket_phase.append(k)
                    cur, size = [], 0
            if cur:
                self.buckets.append(cur)
                self.bucket_phase.append(k)
        self.flat = [torch.empty(sum(p.numel() for p in b), dtype=b[0].dtype, device=b[0].device) for b in self.buckets]
        self._views = [list(flat.split([p.numel() for p in bucket])) for flat, bucket in zip(self.flat, self.buckets)]
        self._grads = [None] * len(self.buckets)
        self._grads_for = [None] * len(self.buckets)
        # the combination weight lives in a one-element device tensor the (possibly captured) scaling reads: changing it
        # after a capture - shards of a different size per batch - takes effect in every existing capture
        self._w = torch.empty((), dtype=self.flat[0].dtype, device=self.flat[0].device) if self.flat else None
        self._push_weight()

    def _push_weight(self):
        if getattr(self, "_w", None) is not None:
            self._w.fill_(self.weight if self.weight is not None else 1.0 / self.world)

    def _of(self, phase):
        return [i for i, k in enumerate(self.bucket_phase) if phase is None or k == phase]

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
            # short timeout: a rank that died while staging its batch must not leave the others blocked for the default 30 min
            import datetime
            self._cpu_group = (self.group if dist.get_backend(self.group) == "gloo" else
                               dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=300)))
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
        self._push_weight()

    def clear_shard_weight(self):
        """Back to the plain mean over the ranks (equal shards)."""
        self.weight = None
        self._push_weight()

    weight = None
    _w = None

    def reduce(self, skip_flatten_of=()):
        """Combine .grad over ranks in place: the mean over ranks, or - after `set_shard_weight` - the token-weighted sum.
        Call after backward(), before the optimizer step.  skip_flatten_of: phases whose buckets were flattened and
        launched already (the engine's two-phase backward)."""
        if not self.active:
            return
        if self.buckets is None:
            self._build()                                        # reduce() follows a complete backward pass
        nph = len(self.phases) if self.phases is not None else 1
        for k in range(nph):
            if k not in skip_flatten_of:
                self.flatten(phase=k)
                self.launch(phase=k)
        self.wait()
        self.unflatten()

    # The phases of reduce().  The step engine captures flatten() at the end of its backward HIP graph(s) and unflatten() at
    # the start of its optimizer graph, so that only the RCCL calls themselves run between the replays.
    def flatten(self, fresh=False, phase=None):
        """gradients -> flat bucket buffers (of one phase, or all), scaled by the combination weight.  fresh: look the .grad
        tensors up again."""
        if not self.active:
            return
        if self.buckets is None:
            assert phase is None, "the buckets are built from the gradients of a COMPLETE backward pass: flatten everything first"
            self._build()
        # the per-tensor views of the flat buffers never change; the list of .grad tensors is rebuilt unless the caller
        # vouches (grads_token) that they are the same objects as last time - building two 634-element lists per bucket
        # cost 2 ms of host time per step
        for i in self._of(phase):
            if fresh or self.grads_token is None or self.grads_token != self._grads_for[i] or self._grads[i] is None:
                self._grads[i] = [p.grad.reshape(-1) for p in self.buckets[i]]
                self._grads_for[i] = None if fresh else self.grads_token
            torch._foreach_copy_(self._views[i], self._grads[i])
            self.flat[i].mul_(self._w)

    def launch(self, phase=None):
        """SUM all-reduce of the flat buffers of one phase (or all): asynchronous - on RCCL the collectives run on the
        process group's own stream behind the work already queued on the current stream, and whatever is queued on the
        current stream next (the rest of the backward pass) runs beside them.  `wait` joins them."""
        if not self.active:
            return
        self._works = self._works + [dist.all_reduce(self.flat[i], op=dist.ReduceOp.SUM, group=self.group, async_op=True) for i in self._of(phase)]

    def wait(self):
        """The current stream waits for every launched collective."""
        works, self._works = self._works, []
        for w in works:
            w.wait()

    def allreduce(self):
        """launch + wait of all buckets."""
        if not self.active:
            return
        self.launch()
        self.wait()

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
    _works = []

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
