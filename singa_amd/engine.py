"""Training-step engine: the reference's step (train.py:113-133: zero_grad -> forward -> CrossEntropy -> backward ->
clip_grad_norm_(inf) -> Adam) as `prepare` (per-batch graph structure, eager, host syncs allowed) + `compute`
(forward/backward/optimizer: enqueue-only).  With `use_graph=True` the compute part is captured once into HIP graphs
(torch.cuda.CUDAGraph = hipGraph on ROCm) and replayed: a step is then ~5,000 kernel launches issued by the GPU's
command processor instead of by Python, which is what bounds the eager step (host ~110 ms vs ~78 ms of GPU work at
the bench workload).  Shapes are static per capture; a batch with different sizes triggers a re-capture.

Multi-GPU: forward+backward replay -> bucketed RCCL all-reduce (singa_amd.dp, eager) -> optimizer replay.
"""
import math

import torch

from . import graph as G, ops
from .graph import E_LL, E_LP, E_PL, E_PP, LA, PA
from .model import EF_layers


def _copy_tree(dst, src, pairs=None):
    """Copy every tensor of a nested dict/list structure into its counterpart; False on a shape mismatch.  With `pairs`
    (a list) the (dst, src) tensors are only collected, for one multi-tensor copy afterwards (`_flush_copies`)."""
    if torch.is_tensor(dst):
        if dst.shape != src.shape:
            return False
        if pairs is None:
            dst.copy_(src)
        else:
            pairs.append((dst, src))
        return True
    if isinstance(dst, dict):
        return all(_copy_tree(dst[k], src[k], pairs) for k in dst)
    if isinstance(dst, (list, tuple)):
        return len(dst) == len(src) and all(_copy_tree(a, b, pairs) for a, b in zip(dst, src))
    return True


def _flush_copies(pairs):
    """All static-buffer updates of a step as a few multi-tensor copies (one per dtype) instead of ~100 launches."""
    by_dtype = {}
    for d, s in pairs:
        if d.data_ptr() == s.data_ptr():
            continue
        if d.dtype != s.dtype or not (d.is_contiguous() and s.is_contiguous()):
            d.copy_(s)
        else:
            by_dtype.setdefault(d.dtype, ([], []))
            by_dtype[d.dtype][0].append(d)
            by_dtype[d.dtype][1].append(s)
    for ds, ss in by_dtype.values():
        torch._foreach_copy_(ds, ss)


def _clone_tree(x):
    if torch.is_tensor(x):
        return x.clone()
    if isinstance(x, dict):
        return type(x)((k, _clone_tree(v)) for k, v in x.items())
    if isinstance(x, (list, tuple)):
        return type(x)(_clone_tree(v) for v in x)
    return x


def _record_tree(x, stream):
    """Tell the caching allocator that `stream` uses these tensors too (they were allocated on another stream)."""
    if torch.is_tensor(x):
        x.record_stream(stream)
    elif isinstance(x, dict):
        for v in x.values():
            _record_tree(v, stream)
    elif isinstance(x, (list, tuple)):
        for v in x:
            _record_tree(v, stream)


def _prep_tensors(prep):
    out = {"p": prep["p"]["dense"].tensors() + prep["p"]["edges"].tensors(),
           "l": prep["l"]["dense"].tensors() + prep["l"]["edges"].tensors(),
           "es": {k: v.tensors() for k, v in prep["es"].items()}}
    if "homo" in prep:
        out["homo"] = [prep["homo"]["ei"]] + prep["homo"]["es"].tensors()
    return out


def concurrent_stream(priority=0, tries=8, group=None):
    """A stream whose work really runs BESIDE the current stream's.  HIP maps streams onto a few hardware queues (4 per
    priority by default) and the work of two streams on one queue runs in order: which queue a new stream gets depends on
    how many streams the process has created (graph captures, an RCCL communicator, ...).  A prefetch stream that shared
    the compute stream's queue - or the queue of the process group's RCCL stream, whose all-reduce sits there waiting for the
    backward pass - ran the batch preparation behind the step instead of beside it: 174-177 ms instead of 165-168 ms per
    config-3 step in about half of the runs (profiles/r03z/dp_timeline.txt).  A high-priority stream is not the answer: the
    step then took 178 ms and the 17-graph shard 47 ms instead of 32 (profiles/r03z/prefetch_priority_ab.txt).  So: probe.
    A few ms of streaming work on the current stream, (with an RCCL `group`) a tiny all-reduce behind it, a tiny kernel on the
    candidate: concurrent if the tiny kernel is done long before the long work.  Every rank tries all `tries` candidates
    (the same number of collectives everywhere) and takes its first concurrent one, the last one if there is none."""
    import torch.distributed as dist
    rccl = group is not False and dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl"
    cur = torch.cuda.current_stream()
    big = torch.empty(64 << 20, device="cuda", dtype=torch.float32)           # 256 MB: ~0.1 ms per pass
    small = torch.empty(1024, device="cuda", dtype=torch.float32)
    tiny = torch.zeros(8, device="cuda", dtype=torch.float32)
    big.zero_()
    best = cand = None
    for _ in range(tries):
        cand = torch.cuda.Stream(priority=priority)
        e0, e_long, e_small = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        torch.cuda.synchronize()
        e0.record(cur)
        for _ in range(40):
            big.add_(1.0)
        e_long.record(cur)
        work = dist.all_reduce(tiny, group=group, async_op=True) if rccl else None    # waits on the RCCL stream for `big`
        with torch.cuda.stream(cand):
            small.fill_(1.0)
            e_small.record(cand)
        if work is not None:
            work.wait()
        torch.cuda.synchronize()
        if best is None and e0.elapsed_time(e_small) < 0.5 * e0.elapsed_time(e_long):
            best = cand
            if not rccl:
                break
    del big, small, tiny
    return best if best is not None else cand


class TrainStep:
    """bucket=True (graph mode): ragged batches - every real CrossDocked batch has its own atom and edge counts - are
    padded with inert atoms / edges (graph.pad_batch) to the capacities of a geometric size class (x `growth` per class),
    so that all batches of a class replay ONE capture; a few captures (`max_cached`, least recently used evicted) are
    kept side by side.  Without it a batch whose shapes differ from the capture triggers a re-capture."""

    def __init__(self, model, optimizer, reducer=None, use_graph=True, max_grad_norm=float("inf"), bucket=False,
                 growth=1.08, max_cached=3, direct_grads=True):
        self.model, self.opt, self.reducer = model, optimizer, reducer
        self.use_graph, self.max_grad_norm = use_graph, max_grad_norm
        self.crit = torch.nn.CrossEntropyLoss()
        self.static = None
        self.g_fb = self.g_opt = None
        self.captures = 0
        self._aux = None
        self.bucket, self.growth, self.max_cached = bucket and use_graph, growth, max_cached
        self._base = None                  # sizes of the first batch: the size classes are multiples of these
        self._mx = {PA: 0, LA: 0}          # widest graph seen so far per node type (dense-layout width, only grows)
        self._knn_cap = {}                 # (class, node type) -> kNN edge capacity
        self._slots = {}                   # signature -> captured state (insertion order = recency)
        self._active = None
        self.direct_grads = direct_grads   # see _fwd_bwd
        self._sink_params = None           # parameters whose gradients are accumulated in place (found in the first step)
        self._opt_gen = None               # optimizer.generation the current captures were made with
        self._pinned_table = None
        self.g_b = None                    # two-phase backward: the second graph (embedding backward)
        self.pre_capture_hook = None       # called right before the forward+backward graph is captured (bench.py: tagging on)
        self.post_capture_hook = None      # ... right after the optimizer graph was captured
        self.comm_events = None            # a list: every replayed step appends (backward done, all-reduce joined) events
        self.prefetch_priority = 0         # priority of the prefetch stream (see `concurrent_stream`)

    # ------------------------------------------------------------------------------------------------ size classes
    def _class_caps(self, sizes):
        if self._base is None:      # the first batch sits in the middle of class 0 (batches a few % larger share it)
            self._base = tuple(max(1, v) * math.sqrt(self.growth) for v in sizes)
        r = max(v / b for v, b in zip(sizes, self._base))
        c = int(math.ceil(math.log(r) / math.log(self.growth) - 1e-9))
        scale = self.growth ** c
        up = lambda v, g: -(-int(math.ceil(v)) // g) * g
        n_p, n_l, e_pp, e_ll, e_x = (b * scale for b in self._base)
        # at least 64 padding atoms per type: the padding edges are spread over them (no heavy segment)
        return c, (up(n_p + 64, 64), up(n_l + 64, 64), up(e_pp, 256), up(e_ll, 256), up(e_x, 256))

    def _stage(self, batch):
        """Pad `batch` to its size class and build the graph structure of the padded batch (bucket mode).  Returns the
        padded batch; its extras['pad']['sig'] identifies the capture it replays."""
        if "pad" in batch.extras and "sig" in batch.extras["pad"]:
            return batch
        B = batch.num_graphs
        # Several ranks: sizes, layout widths and kNN edge counts are agreed with a MAX over the ranks (a few integers over
        # the gloo side group), so every rank derives the SAME size class, capacities and signature from them and the
        # ranks capture / replay in lockstep (ragged shards otherwise land in different classes on different ranks, and
        # every new signature anywhere forces a collective re-capture everywhere).  Every exchange carries an error code
        # in front: a rank whose batch fails to stage (any exception) makes ALL ranks raise together instead of leaving
        # the others blocked in the next exchange.
        agree = self.reducer.max_ints if (self.reducer is not None and self.reducer.active) else (lambda v: list(v))

        def together(err, values):
            out = agree([0 if err is None else 1] + list(values))
            if out[0]:
                if err is not None:
                    raise err
                raise RuntimeError("TrainStep._stage: another rank failed while staging its batch")
            return out[1:]

        err, vals = None, [0] * 7
        try:
            # the Laplacian positional encoding (reference: dgl.lap_pe inside forward, GAN.py:71,77): computed here, i.e.
            # inside the step (on the prefetch stream when the batch was prefetched), for batches that do not carry one -
            # and again on every arrival for batches marked `lap_pe_in_step` (bench.py: resident batches cycled through the
            # timed steps must not keep the encoding of their previous visit)
            for nt, et in ((PA, E_PP), (LA, E_LL)):
                if "lap_pe" not in batch[nt] or batch.extras.get("lap_pe_in_step"):
                    batch[nt]["lap_pe"] = G.laplacian_pe_batched(batch[et]["edge_index"], batch[nt]["batch"], B,
                                                                 self.model.config.model.encoder.lap_dim)
            widths = torch.stack([(batch[nt]["ptr"][1:] - batch[nt]["ptr"][:-1]).max() for nt in (PA, LA)]).tolist()
            vals = list(G.batch_sizes(batch)) + [int(w) for w in widths]
        except Exception as e:                          # noqa: BLE001 - re-raised on every rank by `together`
            err = e
        agreed = together(err, vals)
        c, caps = self._class_caps(tuple(agreed[:5]))
        for nt, w in zip((PA, LA), agreed[5:]):
            self._mx[nt] = max(self._mx[nt], -(-int(w) // 16) * 16)
        try:
            pb = G.pad_batch(batch, *caps)
        except Exception as e:                          # noqa: BLE001
            pb, err = None, e
        EF_layers._edge_cache.clear()
        for attempt in range(3):
            kp, kl = self._knn_cap.get((c, PA)), self._knn_cap.get((c, LA))
            prep, over = None, 0
            if err is None:
                pb.extras["pad"].update(mx_p=self._mx[PA], mx_l=self._mx[LA], knn_p=kp, knn_l=kl)
                pb.extras.pop("prepared", None)
                try:
                    prep = self.model.prepare(pb)
                except OverflowError:                   # a denser batch than the class has seen: forget, measure again
                    over = 1
                except Exception as e:                  # noqa: BLE001
                    err = e
            over, e_p, e_l = together(err, [over] + [prep[k]["edges"].n_edges if prep is not None else 0 for k in ("p", "l")])
            if over:
                self._knn_cap.pop((c, PA), None)
                self._knn_cap.pop((c, LA), None)
                continue
            if kp is not None and kl is not None:
                break
            # first batch of this class: its edge counts (+4 %) become the class capacities, then prepare with them
            for nt, e in ((PA, e_p), (LA, e_l)):
                self._knn_cap[(c, nt)] = -(-int(e * 1.04) // 1024) * 1024
        else:
            raise RuntimeError("TrainStep._stage: the kNN edge capacities of a size class did not settle in 3 attempts")
        pb.extras["pad"]["sig"] = (c, self._mx[PA], self._mx[LA], self._knn_cap[(c, PA)], self._knn_cap[(c, LA)])
        return pb

    # ------------------------------------------------------------------------------------------------ eager pieces
    @property
    def two_phase(self):
        """Several ranks with a reducer built for it (GradAllReducer(phases=[transformer, embedding])): the backward pass
        runs in two parts - loss -> transformer parameters and the embedding's OUTPUTS, then embedding outputs -> embedding
        parameters - and the all-reduce of the transformer's gradients (70 % of the bytes) is in flight while the second
        part computes.  Same arithmetic as one backward() call.  The first pass ever is single-phase: the
        reducer lays out its buckets from the gradients a complete backward pass leaves."""
        r = self.reducer
        return r is not None and r.active and r.phases is not None and r.buckets is not None

    def _sink_begin(self):
        """Parameter gradients that are column sums (biases, affine parameters, split-reduction weight gradients) or small
        transposed products are accumulated straight into `.grad` buffers (ops._GradSink): the first call only records
        which parameters are produced that way; afterwards their buffers are zeroed in one multi-tensor launch, the
        backward functions queue / accumulate into them and one launch pair per backward phase reduces the whole queue
        (instead of two reductions and an add per parameter)."""
        sink = ops._GradSink
        if not self.direct_grads:
            sink.on, sink.found = False, None
        elif self._sink_params is None:
            sink.on, sink.found = False, {}
        else:
            for p in self._sink_params:
                if p.grad is None:
                    p.grad = torch.empty_like(p, memory_format=torch.contiguous_format)
            if self._sink_params:
                torch._foreach_zero_([p.grad for p in self._sink_params])
            sink.on, sink.found = True, None

    @staticmethod
    def _sink_end(ok=True):
        sink = ops._GradSink
        found = sink.found if ok and not sink.on else None
        sink.on, sink.found, sink.jobs = False, None, []
        return found

    def _phase_a(self, batch, split):
        """Forward, CrossEntropy and the backward pass down to the parameters of the transformer and - when `split` - to the
        embedding's outputs (returned as (output, detached twin holding the gradient) pairs)."""
        boundary = [] if split else None
        logits = self.model(batch, boundary=boundary) if split else self.model(batch)
        loss = self.crit(logits, batch["ligand_data"]["smiIndices_tgt"].reshape(-1))
        loss.backward()
        if ops._GradSink.on:
            ops._GradSink.flush()
        return loss, boundary

    @staticmethod
    def _phase_b(boundary):
        """The rest of the backward pass: embedding outputs -> embedding parameters."""
        torch.autograd.backward([x for x, _ in boundary], [d.grad for _, d in boundary])
        if ops._GradSink.on:
            ops._GradSink.flush()

    def _fwd_bwd(self, batch):
        """Forward, CrossEntropy, backward (eager).  Two-phase mode: the transformer's buckets are flattened and their
        all-reduce launched between the phases; the caller's reducer.reduce(skip_flatten_of=(0,)) does the rest."""
        self._sink_begin()
        ok = False
        try:
            split = self.two_phase
            loss, boundary = self._phase_a(batch, split)
            if split:
                self.reducer.flatten(fresh=True, phase=0)
                self.reducer.launch(phase=0)
                self._phase_b(boundary)
            self._phases_done = (0,) if split else ()
            ok = True
        finally:
            found = self._sink_end(ok)
            if found is not None:
                self._sink_params = list(found.values())
        return loss

    _phases_done = ()

    def _update(self):
        # clip_grad_norm_(parameters, max_norm) of train.py:126.  The norm comes from the library's own chunked
        # sum-of-squares (singa_grad_norm: fixed summation order, no cross-launch state).  torch's norm goes through its
        # multi-block reduction kernels, which return garbage for a few outputs under HIP-graph replay on this build (the
        # same defect that singa_colsum replaces for bias gradients): that garbage, not the gradients, was what made
        # clamp(inf / (norm + 1e-6), max=1) NaN in replayed steps (tests/test_engine_gpu.py checks the replayed norm
        # against the eager one).  The reference passes max_norm = inf - the coefficient is then exactly 1, so the
        # multiplication is skipped; a finite max_norm scales the gradients as torch does.
        if hasattr(self.opt, "grad_norm"):
            self.grad_norm = self.opt.grad_norm()
        else:
            grads = [p.grad for p in self.model.parameters() if p.grad is not None]
            self.grad_norm = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads)))
        if self.max_grad_norm != float("inf"):
            grads = [p.grad for p in self.model.parameters() if p.grad is not None]
            coef = (self.max_grad_norm / (self.grad_norm + 1e-6)).clamp(max=1.0)
            torch._foreach_mul_(grads, coef)
        if hasattr(self.opt, "sync_hyper") and not torch.cuda.is_current_stream_capturing():
            self.opt.sync_hyper()
        self.opt.step()

    def prefetch(self, batch):
        """Build the graph structure of a COMING batch (edge sorting, kNN graphs, dense maps: `SINGA.prepare`, a few ms
        with two host read-backs) on a second stream, so that it overlaps the step that is computing now.  `batch` is
        a HeteroGraph whose tensors are complete, or a callable that makes one (run under the second stream, so a
        host->device copy of the next batch does not queue behind the running step either).  `step` then finds the
        batch prepared and only waits for the event."""
        with torch.cuda.stream(self.aux_stream()):
            if callable(batch):
                batch = batch()
            if self.bucket:
                batch = self._stage(batch)     # padded to its size class, structure built on the padded batch
            else:
                EF_layers._edge_cache.clear()  # a new batch: sort its edges again
                batch.extras.pop("prepared", None)
                self.model.prepare(batch)
            batch.extras["prepared_by_prefetch"] = True
            batch.extras["prefetched"] = torch.cuda.Event()
            batch.extras["prefetched"].record(self._aux)
        return batch

    def aux_stream(self):
        """The prefetch stream (found by the concurrency probe of `concurrent_stream`: 256 MB of scratch, a few device
        synchronisations, with a reducer a few tiny all-reduces - every rank must call it at the same point).  `step`
        creates it at the end of every capture (eager mode: before its first step), so that it never falls into a timed
        steady-state region."""
        if self._aux is None:
            r = self.reducer
            self._aux = concurrent_stream(self.prefetch_priority, group=r.group if (r is not None and r.active) else False)
        return self._aux

    @staticmethod
    def _join_prefetch(batch):
        """If `prefetch` prepared this batch on the second stream: make the current stream wait for it and tell the
        allocator that the batch's and the preparation's tensors are used here too.  Called before ANYTHING reads the
        batch (clones for the static buffers, prepare, warm-up)."""
        ev = batch.extras.pop("prefetched", None)
        if ev is None:
            return False
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        if "prepared" in batch.extras:
            _record_tree(_prep_tensors(batch.extras["prepared"]), cur)
        _record_tree([batch.nodes, batch.edges, batch.globals], cur)
        return True

    def _prepared(self, batch):
        """The batch's graph structure: taken from `prefetch` if it ran, else built now."""
        self._join_prefetch(batch)
        if "prepared" in batch.extras and (batch.extras.pop("prepared_by_prefetch", False) or
                                           "sig" in batch.extras.get("pad", {})):
            return batch.extras["prepared"]      # built by `prefetch`, or together with the padding (`_stage`)
        EF_layers._edge_cache.clear()          # a new batch: sort its edges again
        batch.extras.pop("prepared", None)
        return self.model.prepare(batch)

    def eager_step(self, batch):
        self._prepared(batch)
        self.opt.zero_grad(set_to_none=True)
        loss = self._fwd_bwd(batch)
        if self.reducer is not None:
            self.reducer.grads_token = None                          # fresh gradient tensors every eager step
            self.reducer.reduce(skip_flatten_of=self._phases_done)
        self._update()
        return loss

    # ------------------------------------------------------------------------------------------------ graph mode
    def _make_static(self, batch):
        import copy
        st = copy.copy(batch)
        st.nodes = _clone_tree(batch.nodes)
        st.edges = _clone_tree(batch.edges)
        st.globals = _clone_tree(batch.globals)
        st.extras = type(batch.extras)((k, (dict(v) if k == "pad" else _clone_tree(v))) for k, v in batch.extras.items()
                                       if k not in ("prepared", "prefetched", "prepared_by_prefetch"))
        EF_layers._edge_cache.clear()
        prep = self.model.prepare(st)
        n_p, n_l = st[PA]["x"].shape[0], st[LA]["x"].shape[0]
        for key, et in (("pp", E_PP), ("ll", E_LL), ("lp", E_LP), ("pl", E_PL)):
            EF_layers._edge_pinned[st[et]["edge_index"].data_ptr()] = prep["es"][key]
        if "homo" in prep:
            EF_layers._edge_pinned[prep["homo"]["ei"].data_ptr()] = prep["homo"]["es"]
        self.static, self.static_prep = st, prep
        self._sig = (n_p, n_l, prep["p"]["dense"].mx, prep["l"]["dense"].mx)

    def _load(self, batch):
        """Per-step work for an arriving batch: rebuild its graph structure (eager) and copy everything into the static
        buffers the captured graphs read.  False if any shape differs from the capture."""
        st = self.static
        prep = self._prepared(batch)
        sig = (batch[PA]["x"].shape[0], batch[LA]["x"].shape[0], prep["p"]["dense"].mx, prep["l"]["dense"].mx)
        if sig != self._sig:
            return False
        pairs = []
        ok = (_copy_tree(st.nodes, batch.nodes, pairs) and _copy_tree(st.edges, batch.edges, pairs)
              and _copy_tree(st.globals, batch.globals, pairs))
        keys = [k for k in st.extras if k != "pad"]
        ok = ok and all(k in batch.extras for k in keys)
        ok = ok and _copy_tree({k: st.extras[k] for k in keys}, {k: batch.extras[k] for k in keys}, pairs)
        ok = ok and _copy_tree(_prep_tensors(self.static_prep), _prep_tensors(prep), pairs)
        if ok:
            _flush_copies(pairs)
        return ok

    def _capture(self, batch):
        """Capture the compute part for batches shaped like `batch`.  Side-effect free: the two warm-up steps graph
        capture needs (allocator, library handles, autograd buffers) run on real data, but parameters, both Adam moments
        and the step count are put back afterwards, so a (re-)capture never changes the training trajectory - the
        batch gets exactly one update, from the replay that follows."""
        self._join_prefetch(batch)
        EF_layers._edge_pinned.clear()
        self._make_static(batch)
        st = self.static
        EF_layers._frame_flag_tensor(st[PA]["pos"].device)     # the guards' device-side statistics must exist before capture
        snap = self.opt.snapshot() if hasattr(self.opt, "snapshot") else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up on a side stream, as graph capture requires
            for _ in range(2):
                self.opt.zero_grad(set_to_none=True)
                self._fwd_bwd(st)
                if self.reducer is not None:
                    self.reducer.grads_token = None
                    self.reducer.reduce(skip_flatten_of=self._phases_done)
                self._update()
            if snap is not None:
                self.opt.restore(snap)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.opt.zero_grad(set_to_none=True)
        torch.cuda.empty_cache()      # hand the warm-up's cached blocks back: the graph pool needs the same amount again
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)   # warm-up ran on the side stream
        self.g_fb, self.g_b = torch.cuda.CUDAGraph(), None
        split = self.two_phase
        if self.pre_capture_hook is not None:
            self.pre_capture_hook()
        try:
            # thread_local: the RCCL watchdog thread may touch the HIP runtime while this thread captures
            with torch.cuda.graph(self.g_fb, capture_error_mode="thread_local"):
                self._sink_begin()                                   # (allocates / zeroes the in-place gradient buffers: part of the graph)
                self.static_loss, boundary = self._phase_a(st, split)
                if self.reducer is not None:
                    # the copy of this capture's gradients into the all-reduce buckets rides in the graph; between the
                    # replays only the RCCL calls themselves are issued (GradAllReducer.flatten / launch / wait / unflatten)
                    self.reducer.flatten(fresh=True, phase=0 if split else None)
            if split:
                # second graph, same memory pool (the embedding's saved activations live in it): the embedding's backward
                # pass and its buckets; replayed while the transformer's all-reduce is in flight
                self.g_b = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g_b, pool=self.g_fb.pool(), capture_error_mode="thread_local"):
                    self._phase_b(boundary)
                    self.reducer.flatten(fresh=True, phase=1)
                del boundary
        finally:
            self._sink_end(False)
        if hasattr(self.opt, "_grad_table"):
            # the .grad tensors now live in the graph pool: publish their addresses in a table of this capture's own (a
            # host->device copy, so it has to happen outside the capture; pinned until the capture is dropped)
            if not self.bucket and getattr(self, "_pinned_table", None) is not None:
                self.opt.unpin(self._pinned_table)
            self._pinned_table = self.opt._grad_table(pin=True)
        self.g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_opt, capture_error_mode="thread_local"):
            if self.reducer is not None:
                self.reducer.unflatten()
            self._update()
        if self.post_capture_hook is not None:
            self.post_capture_hook()
        # the prefetch stream is (re-)chosen AFTER every capture: a capture creates streams of its own, and a stream probed
        # before it ended up sharing a hardware queue with the replayed step (round 4, first bench: the 17-graph shard at
        # 39.9 ms instead of 31.2 - exactly the preparation running behind the step instead of beside it).  Captures happen in
        # warm-up and are collective under data parallelism, so the probe never lands in a timed steady-state region and every
        # rank runs its collectives at the same point.
        self._aux = None
        self.aux_stream()
        self.captures += 1
        self._opt_gen = getattr(self.opt, "generation", 0)     # (the warm-up may have built the optimizer)

    # ------------------------------------------------------------------------------------------------ capture slots
    _SLOT_FIELDS = ("static", "static_prep", "_sig", "g_fb", "g_b", "g_opt", "static_loss")

    def _activate(self, sig):
        """Make the capture of signature `sig` the current one (bucket mode): its static buffers, graphs and - because the
        gradients live in each capture's private pool - its .grad tensors."""
        if self._active == sig:
            return
        slot = self._slots[sig]
        for f in self._SLOT_FIELDS:
            setattr(self, f, slot[f])
        for p, g in zip(self.model.parameters(), slot["grads"]):
            p.grad = g
        self._slots[sig] = self._slots.pop(sig)            # most recently used last
        self._active = sig

    def _store(self, sig):
        slot = {f: getattr(self, f) for f in self._SLOT_FIELDS}
        slot["grads"] = [p.grad for p in self.model.parameters()]
        slot["g_ptr"] = getattr(self.opt, "g_ptr", None)   # the address table the optimizer graph reads (pinned in the optimizer)
        old_slot = self._slots.pop(sig, None)
        if old_slot is not None and hasattr(self.opt, "unpin"):
            self.opt.unpin(old_slot["g_ptr"])
        self._slots[sig] = slot
        self._active = sig
        while len(self._slots) > self.max_cached:
            old = next(iter(self._slots))
            gone = self._slots.pop(old)                    # its graphs and pool are freed with the last reference
            if hasattr(self.opt, "unpin"):
                self.opt.unpin(gone["g_ptr"])
            torch.cuda.synchronize()
            torch.cuda.empty_cache()

    def release(self):
        """Drop the captured graphs and their private memory pool (e.g. before running large eager steps)."""
        self._slots.clear()
        self._active = None
        if hasattr(self.opt, "unpin_all") and getattr(self.opt, "_built", False):
            self.opt.unpin_all()
        self._pinned_table = None
        self.g_fb = self.g_b = self.g_opt = self.static = self.static_prep = self.static_loss = None
        EF_layers._edge_pinned.clear()
        self.opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    def check(self):
        """The reference's edge-frame guards for frames built inside replayed graphs (one read-back; call it where the
        loss is read back anyway)."""
        EF_layers.check_edge_frames()

    def step(self, batch):
        if not self.use_graph:
            if self._aux is None and batch[PA]["x"].is_cuda:
                self.aux_stream()           # first step: the probe runs here, not inside somebody's timed region
            return self.eager_step(batch)
        gen = getattr(self.opt, "generation", 0)
        if gen != self._opt_gen:
            # the optimizer re-created its moment buffers (load_state_dict / first build): captures made before still
            # point at the old ones
            if self._opt_gen is not None and (self.static is not None or self._slots):
                self.release()
            self._opt_gen = gen
        if self.bucket:
            return self._bucket_step(batch)
        need = self.static is None or not self._load(batch)
        if self.reducer is not None:
            # a re-capture runs two warm-up steps with their own all-reduces: every rank has to take that path together
            need = self.reducer.any_rank(need)
        if need:
            self._capture(batch)
            ok = self._load(batch)
            assert ok, "the batch does not fit the buffers captured from it"
        return self._replay()

    def _replay(self):
        self.g_fb.replay()
        ev = None
        if self.comm_events is not None and self.reducer is not None and self.reducer.active:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        if self.g_b is not None:
            self.reducer.launch(phase=0)                            # the transformer's buckets travel ...
            self.g_b.replay()                                       # ... while the embedding's backward pass computes
            if ev:
                ev[0].record()                                      # the backward pass is done here ...
            self.reducer.launch(phase=1)
            self.reducer.wait()
        elif self.reducer is not None:
            if ev:
                ev[0].record()
            self.reducer.allreduce()                                # flatten / unflatten are inside the graphs
        if ev:
            ev[1].record()                                          # ... and every bucket has arrived here: the time between the
            self.comm_events.append(ev)                             # two is what the all-reduce costs the step (exposed)
        if hasattr(self.opt, "sync_hyper"):
            self.opt.sync_hyper()                            # scheduler-updated learning rate -> device scalar
        self.g_opt.replay()
        return self.static_loss

    def _bucket_step(self, batch):
        self._join_prefetch(batch)
        pb = self._stage(batch)
        sig = pb.extras["pad"]["sig"]
        need = sig not in self._slots
        if self.reducer is not None:
            need = self.reducer.any_rank(need)               # NB: the ranks' batches must then fall into the same class
        if need:
            self.static = None
            self._capture(pb)
            self._store(sig)
        else:
            self._activate(sig)
        ok = self._load(pb)
        assert ok, "a padded batch does not fit the buffers of its own size class"
        return self._replay()
