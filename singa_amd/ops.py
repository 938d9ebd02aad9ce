"""torch.autograd.Function wrappers around the C ABI of libsinga_hip.so.

Every op takes CUDA (ROCm) fp32 tensors, hands raw device pointers plus the current HIP stream to the library and
returns tensors allocated by PyTorch.  There is NO CPU path: CPU tensors raise.  Rotations are non-differentiable
inputs exactly as in the reference (`.detach()` at EF:490-491, 2351); backward kernels recompute instead of saving
rotated intermediates.
"""
import ctypes
import os
import math
import warnings

import torch

warnings.filterwarnings("ignore", message=".*preferred_blas_library is an experimental feature.*")

from . import _capi, _lib, so3


# Optional exact timing of the scatter-TP forward dispatches (bench.py): while PROFILE_ON, the library attaches a
# start and a stop event to each such dispatch on the stream it is launched on (singa_prof_enable / _collect).
PROFILE_ON = False


def profile_start(in_graph=False):
    """in_graph: a device-timestamp kernel in front of and behind every tagged dispatch instead of events attached to it -
    plain kernel nodes that are captured with the step and re-run by every replay (read with profile_read after a replay)."""
    global PROFILE_ON, _STAMPS
    PROFILE_ON = True
    if in_graph:
        _STAMPS = torch.zeros(2 * 8192 + 2, dtype=torch.int64, device="cuda")
        _chk(_lib.lib().singa_prof_stamps(_p(_STAMPS), _STAMPS.numel()), "singa_prof_stamps")
    else:
        _chk(_lib.lib().singa_prof_enable(1), "singa_prof_enable")


_STAMPS = None


def profile_pause():
    """Stop tagging new dispatches but keep the records and the stamp buffer (graph mode: after the capture, before the
    replays)."""
    global PROFILE_ON
    PROFILE_ON = False
    _chk(_lib.lib().singa_prof_stamps(None, 0), "singa_prof_stamps")
    _chk(_lib.lib().singa_prof_enable(0), "singa_prof_enable")


def profile_read():
    """Graph mode, after a replay + synchronize: [(tag, ms, n_edges, n_nodes)] of the captured dispatches (kept)."""
    cap = 8192
    host = _STAMPS.cpu().numpy()
    ms, tg, ne, nn = (ctypes.c_float * cap)(), (ctypes.c_int * cap)(), (ctypes.c_int * cap)(), (ctypes.c_int * cap)()
    n = _lib.lib().singa_prof_read_stamps(host.ctypes.data_as(ctypes.c_void_p), ms, tg, ne, nn, cap)
    return [(PROF_TAGS.get(tg[i], str(tg[i])), ms[i], ne[i], nn[i]) for i in range(n)]


def profile_end():
    """Graph mode: forget the records and release the stamp buffer (the capture that writes into it must be gone)."""
    global _STAMPS
    _chk(_lib.lib().singa_prof_stamps(None, 0), "singa_prof_stamps")
    _chk(_lib.lib().singa_prof_reset(), "singa_prof_reset")
    _STAMPS = None


PROF_TAGS = {1: "k10_fwd", 2: "k10_bwd", 3: "k4_fwd", 4: "k4_bwd_rad", 5: "k4_bwd_dst", 6: "k4_bwd_src",           # include/singa_hip.h
             7: "gemm_nt", 8: "gemm_nn", 9: "gemm_tn", 10: "s2_edge_fwd", 11: "s2_edge_bwd", 12: "s2_node_fwd", 13: "s2_node_bwd",
             14: "cgemm_nt", 15: "cgemm_nn", 16: "cgemm_tn"}


def profile_collect():
    """Call after torch.cuda.synchronize().  Returns [(kernel tag name, ms, n_edges, n_nodes)] for every profiled
    dispatch (tags: SINGA_PROF_* of include/singa_hip.h)."""
    global PROFILE_ON
    PROFILE_ON = False
    lib = _lib.lib()
    _chk(lib.singa_prof_enable(0), "singa_prof_enable")
    cap = 8192
    ms, tg, ne, nn = (ctypes.c_float * cap)(), (ctypes.c_int * cap)(), (ctypes.c_int * cap)(), (ctypes.c_int * cap)()
    n = lib.singa_prof_collect_tagged(ms, tg, ne, nn, cap)
    return [(PROF_TAGS.get(tg[i], str(tg[i])), ms[i], ne[i], nn[i]) for i in range(n)]


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_branch_streams = {}


def branch_stream(device=None):
    """THE second stream of the model's forward / backward passes on `device` (one per device, created on first use): the
    ligand encoder beside the protein encoder (CProMG.Transformer), the ligand -> protein pass's live layer beside the protein
    -> ligand pass (EquivariantEmbedding).  One shared stream on purpose: a captured step then never has more than two
    concurrent branches - with a stream per module the embedding's backward branch could run beside the ligand encoder's
    (three branches), and an instrumented capture of such a step crashed inside hipGraphLaunch on this ROCm build."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    dev = torch.cuda.current_device() if dev is None else dev
    if dev not in _branch_streams:
        _branch_streams[dev] = torch.cuda.Stream(device=dev)
    return _branch_streams[dev]


DW_SIDE_STREAM = os.environ.get("SINGA_DW_SIDE", "0") == "1"      # OFF: measured slower, see _dw_side


class _dw_side:
    """Context for the launches that produce a WEIGHT gradient while the parameter-gradient queue (_GradSink) is on: nothing
    in the backward pass waits for them (only the queue's flush does), so they run on the model's second stream beside the
    chain of input-gradient launches - at shard size (3-7 k rows) those are latency-bound 64 x 64-tile GEMMs that leave most
    of every CU idle.  Fork: the second stream waits for everything issued so far on the current one (the incoming gradient);
    join: _GradSink.flush.  Tensors the side launches read are handed to the allocator with record_stream.  No fork when
    the queue is off (eager callers get their gradients returned), when already on the second stream (the ligand encoder's
    and the hetero pass's backward run there).  OFF by default (SINGA_DW_SIDE=1 turns it on): correct (engine and GEMM suites
    pass) but SLOWER in the replayed step - ~200 fork edges per step cost more than the overlap returns on this ROCm build:
    config 3 150.8 -> 162.0 ms, the 17-graph shard 29.6 -> 40.7 ms (same box, back to back)."""

    def __init__(self, *reads, params=()):
        """reads: tensors the launches read; params: the parameters whose gradients they produce - the fork only happens
        when ALL of them take their gradient through the queue (otherwise the caller reduces the partials right away)."""
        self.reads, self.side, self.ctx = reads, None, None
        self.direct = all(_GradSink.takes(p) for p in params if p is not None)

    def __enter__(self):
        if DW_SIDE_STREAM and _GradSink.on and self.direct:
            cur = torch.cuda.current_stream()
            side = branch_stream(cur.device)
            if side != cur:
                side.wait_stream(cur)
                self.side, self.cur = side, cur
                self.ctx = torch.cuda.stream(side)
                self.ctx.__enter__()
                _GradSink.forked = True
        return self

    def keep(self, *outs):
        """outs: tensors allocated inside the context that the current stream reads later (at the flush)."""
        if self.side is not None:
            for t in outs:
                t.record_stream(self.cur)

    def __exit__(self, *exc):
        if self.side is not None:
            self.ctx.__exit__(*exc)
            for t in self.reads:
                if t is not None:
                    t.record_stream(self.side)
        return False


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _chk(code, what):
    _capi.check(_lib.lib(), code, what)


def _dev(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("singa_amd ops run on the GPU only (no CPU fallback); got a CPU tensor")
        if t.dtype not in (torch.float32, torch.int32):
            raise RuntimeError(f"singa_amd ops take float32 / int32 tensors, got {t.dtype}")
    _lib.ensure_init(ts[0].device.index if ts[0].device.index is not None else torch.cuda.current_device())


class EdgeSet:
    """Edges of one edge type sorted by destination (CSR) plus the source-sorted permutation used by backward.

    order: the permutation that was applied to the caller's edge list (edge i here = original edge order[i])."""

    def __init__(self, edge_index, n_src, n_dst):
        ei = edge_index.to(torch.int64)
        dev = ei.device
        self.n_src, self.n_dst, self.E = int(n_src), int(n_dst), int(ei.shape[1])
        order = torch.argsort(ei[1], stable=True)
        self.order = order
        self.src64, self.dst64 = ei[0][order].contiguous(), ei[1][order].contiguous()
        self.src, self.dst = self.src64.to(torch.int32), self.dst64.to(torch.int32)
        # CSR / CSC pointers by binary search in the sorted index lists (no atomics)
        self.row_ptr = torch.searchsorted(self.dst64, torch.arange(self.n_dst + 1, device=dev)).to(torch.int32)
        eperm = torch.argsort(self.src64, stable=True)
        self.eperm = eperm.to(torch.int32)
        self.col_ptr = torch.searchsorted(self.src64[eperm], torch.arange(self.n_src + 1, device=dev)).to(torch.int32)

    def tensors(self):
        return [self.order, self.src64, self.dst64, self.src, self.dst, self.row_ptr, self.eperm, self.col_ptr]


def edge_frames(vec, rand, stats):
    """k1: edge vectors [E,3] + the uniform draws [E,3] of the reference's torch.rand_like (Q6) -> edge frames [E,3,3]
    (no gradient, EF:2286-2351, `.detach()` at EF:2351).  stats: float32 [2] on the same device, updated in place to
    (min(stats[0], shortest edge), max(stats[1], largest |cos(edge, helper)|)) - the reference's two guards."""
    vec, rand = vec.detach().contiguous().float(), rand.detach().contiguous().float()
    _dev(vec, rand, stats)
    E = vec.shape[0]
    assert vec.shape == (E, 3) and rand.shape == (E, 3) and stats.shape == (2,) and stats.is_contiguous()
    rot = torch.empty(E, 3, 3, device=vec.device, dtype=torch.float32)
    _chk(_lib.lib().singa_edge_frames(_p(vec), _p(rand), _p(rot), _p(stats), E, _stream()), "singa_edge_frames")
    return rot


def lap_pe(src, dst, eptr, num, start, n_total, mx, k=8):
    """n2: Laplacian positional encoding of every graph of a batch (reference model/CProMG.py:562-571; no gradient).
    src / dst: int32 LOCAL atom indices of the edges, grouped by graph; eptr [B + 1] the graphs' edge ranges; num [B]
    atoms per graph; start [B] first row of each graph in the result; mx >= max(num).  -> fp32 [n_total, k]."""
    for t in (src, dst, eptr):
        if not t.is_cuda or t.dtype != torch.int32:
            raise RuntimeError("lap_pe: CUDA int32 edge arrays (no CPU path)")
    B = num.numel()
    dev = src.device
    _lib.ensure_init(dev.index if dev.index is not None else torch.cuda.current_device())
    lib = _lib.lib()
    src, dst, eptr = src.contiguous(), dst.contiguous(), eptr.contiguous()
    nn, st = num.to(torch.int32).contiguous(), start.to(torch.int32).contiguous()
    scratch = torch.empty(B * mx * mx, device=dev, dtype=torch.float64)           # only the components' diagonal blocks are touched
    work = torch.empty(max(lib.singa_lap_pe_work(B, mx), 1), device=dev, dtype=torch.float64)
    out = torch.zeros(n_total, k, device=dev, dtype=torch.float32)
    _chk(lib.singa_lap_pe(_p(scratch), _p(src), _p(dst), _p(eptr), _p(nn), _p(st), _p(work), _p(out), B, mx, k, _stream()),
         "singa_lap_pe")
    return out


def wigner_rows(rot, L, M=2):
    """k2: rot [E,3,3] -> reduced Wigner rows [E, WSZ] (no gradient, EF:485-528)."""
    rot = rot.detach().contiguous().float()
    _dev(rot)
    lay = so3.layout(L, M)
    E = rot.shape[0]
    wr = torch.zeros(E, lay.WSZ, device=rot.device, dtype=torch.float32)     # WSZ is padded to 16-byte records; pad floats = 0
    _chk(_lib.lib().singa_wigner_rows(_p(rot), _p(wr), E, L, M, _stream()), "singa_wigner_rows")
    return wr


class _GatherRotate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_src, x_dst, rad, wr, es, L, M):
        x_src, x_dst, rad = x_src.contiguous(), x_dst.contiguous(), rad.contiguous()
        _dev(x_src, x_dst, rad, wr)
        lay = so3.layout(L, M)
        C = x_src.shape[2]
        assert x_src.shape[1] == lay.K and x_dst.shape[1] == lay.K and wr.shape == (es.E, lay.WSZ)
        assert x_src.shape[0] == es.n_src and x_dst.shape[0] == es.n_dst and rad.shape == (es.E, lay.rad_rows * 2 * C)
        out = torch.empty(es.E, lay.KR * 2 * C, device=x_src.device, dtype=torch.float32)
        _chk(_lib.lib().singa_gather_rotate_fwd(_p(x_src), _p(x_dst), _p(es.src), _p(es.dst), _p(wr), _p(rad), _p(out),
                                                es.E, C, L, M, _stream()), "singa_gather_rotate_fwd")
        ctx.save_for_backward(x_src, x_dst, rad, wr)
        ctx.es, ctx.L, ctx.M = es, L, M
        return out

    @staticmethod
    def backward(ctx, g):
        x_src, x_dst, rad, wr = ctx.saved_tensors
        es, L, M = ctx.es, ctx.L, ctx.M
        g = g.contiguous()
        C = x_src.shape[2]
        g_rad = torch.empty_like(rad)
        gx_src, gx_dst = torch.empty_like(x_src), torch.empty_like(x_dst)
        _chk(_lib.lib().singa_gather_rotate_bwd(_p(g), _p(x_src), _p(x_dst), _p(es.src), _p(es.dst), _p(wr), _p(rad),
                                                _p(es.row_ptr), _p(es.col_ptr), _p(es.eperm), _p(g_rad), _p(gx_src),
                                                _p(gx_dst), es.E, es.n_src, es.n_dst, C, L, M, _stream()),
             "singa_gather_rotate_bwd")
        return gx_src, gx_dst, g_rad, None, None, None, None


def gather_rotate(x_src, x_dst, rad, wr, es, L, M=2):
    """k3-k6: [N,K,C] node tensors -> m-primary edge tensor [E, KR*2C], multiplied by rad [E, RAD_ROWS*2C]."""
    return _GatherRotate.apply(x_src, x_dst, rad, wr, es, L, M)


def _segs3(ts, rows, CH):
    return _capi.segs([(t.data_ptr(), t.stride(0), r) for t, r in zip(ts, rows)])


class _RotateBackScatter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y0, y1, y2, alpha, wr, es, heads, L, M):
        y0, y1, y2, alpha = _rows(y0), _rows(y1), _rows(y2), alpha.contiguous()      # column-block views are taken as they are
        _dev(y0, y1, y2, alpha, wr)
        lay = so3.layout(L, M)
        CH = y0.shape[1] // lay.seg_rows[0]
        assert y1.shape[1] == lay.seg_rows[1] * CH and y2.shape[1] == lay.seg_rows[2] * CH
        assert alpha.shape == (es.E, heads) and wr.shape == (es.E, lay.WSZ)
        out = torch.empty(es.n_dst, lay.K, CH, device=y0.device, dtype=torch.float32)
        seg, n = _segs3((y0, y1, y2), lay.seg_rows, CH)
        if PROFILE_ON:
            _lib.lib().singa_prof_hint_edges(es.E)
        _chk(_lib.lib().singa_rotate_back_scatter_fwd(seg, n, _p(alpha), _p(wr), _p(es.row_ptr), _p(out), es.n_dst, CH,
                                                      heads, L, M, 0, 1.0, _stream()), "singa_rotate_back_scatter_fwd")
        ctx.save_for_backward(y0, y1, y2, alpha, wr)
        ctx.es, ctx.heads, ctx.L, ctx.M, ctx.CH = es, heads, L, M, CH
        return out

    @staticmethod
    def backward(ctx, g):
        y0, y1, y2, alpha, wr = ctx.saved_tensors
        es, heads, L, M, CH = ctx.es, ctx.heads, ctx.L, ctx.M, ctx.CH
        lay = so3.layout(L, M)
        g = g.contiguous()
        widths = [y0.shape[1], y1.shape[1], y2.shape[1]]
        gY = torch.empty(es.E, sum(widths), device=g.device, dtype=torch.float32)   # the three gradients as column blocks
        gy = list(gY.split(widths, dim=1))
        gap = torch.empty(es.E, CH, device=g.device, dtype=torch.float32)
        seg, n = _segs3((y0, y1, y2), lay.seg_rows, CH)
        gseg, _ = _segs3(gy, lay.seg_rows, CH)
        if PROFILE_ON:
            _lib.lib().singa_prof_hint_edges(es.E)
        _chk(_lib.lib().singa_rotate_back_scatter_bwd(_p(g), seg, gseg, n, _p(alpha), _p(wr), _p(es.row_ptr), _p(gap),
                                                      es.n_dst, CH, heads, L, M, 0, 1.0, _stream()),
             "singa_rotate_back_scatter_bwd")
        g_alpha = gap.view(es.E, heads, CH // heads).sum(-1)
        return gy[0], gy[1], gy[2], g_alpha, None, None, None, None, None


def rotate_back_scatter(y0, y1, y2, alpha, wr, es, heads, L, M=2):
    """k10: per-m SO(2)-conv outputs (m-primary) * alpha -> rotate back -> sum into destination nodes [Nd,K,CH]."""
    return _RotateBackScatter.apply(y0, y1, y2, alpha, wr, es, heads, L, M)


class _EdgeDegreeScatter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, r, wr, es, L, M, scale):
        r = r.contiguous()
        _dev(r, wr)
        lay = so3.layout(L, M)
        C = r.shape[1] // lay.m_size[0]
        out = torch.empty(es.n_dst, lay.K, C, device=r.device, dtype=torch.float32)
        seg, n = _capi.segs([(r.data_ptr(), r.stride(0), lay.m_size[0])])
        _chk(_lib.lib().singa_rotate_back_scatter_fwd(seg, n, None, _p(wr), _p(es.row_ptr), _p(out), es.n_dst, C, 1, L, M,
                                                      1, scale, _stream()), "singa_rotate_back_scatter_fwd(m0)")
        ctx.save_for_backward(wr)
        ctx.es, ctx.L, ctx.M, ctx.scale, ctx.shape, ctx.C = es, L, M, scale, r.shape, C
        return out

    @staticmethod
    def backward(ctx, g):
        (wr,) = ctx.saved_tensors
        es, L, M = ctx.es, ctx.L, ctx.M
        lay = so3.layout(L, M)
        g = g.contiguous()
        gr = torch.empty(ctx.shape, device=g.device, dtype=torch.float32)
        gseg, n = _capi.segs([(gr.data_ptr(), gr.stride(0), lay.m_size[0])])
        _chk(_lib.lib().singa_rotate_back_scatter_bwd(_p(g), None, gseg, n, None, _p(wr), _p(es.row_ptr), None, es.n_dst,
                                                      ctx.C, 1, L, M, 1, ctx.scale, _stream()),
             "singa_rotate_back_scatter_bwd(m0)")
        return gr, None, None, None, None, None


def edge_degree_scatter(r, wr, es, L, M=2, scale=1.0):
    """k13: m=0 radial output [E,(L+1)*C] -> rotate back (m=0 Wigner columns only) -> node sum * scale."""
    return _EdgeDegreeScatter.apply(r, wr, es, L, M, scale)


class _SegmentSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, row_ptr, eps):
        x = x.contiguous()
        _dev(x, row_ptr)
        y = torch.empty_like(x)
        N = row_ptr.numel() - 1
        ctx.dense = int(x.shape[1] == 4 and x.shape[0] >= 16 * N)      # >= 16 edges per segment on average: wave per segment
        _chk(_lib.lib().singa_segment_softmax_fwd(_p(x), _p(row_ptr), _p(y), N, x.shape[1], eps, ctx.dense, _stream()),
             "singa_segment_softmax_fwd")
        ctx.save_for_backward(y, row_ptr)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, row_ptr = ctx.saved_tensors
        gy = gy.contiguous()
        gx = torch.empty_like(y)
        N = row_ptr.numel() - 1
        _chk(_lib.lib().singa_segment_softmax_bwd(_p(y), _p(gy), _p(row_ptr), _p(gx), N, y.shape[1], ctx.dense, _stream()),
             "singa_segment_softmax_bwd")
        return gx, None, None


def segment_softmax(x, row_ptr, eps):
    """k9: softmax of x[E,H] over the edge segments of row_ptr (edges sorted by segment)."""
    return _SegmentSoftmax.apply(x, row_ptr, eps)


class _SegmentWSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, v, row_ptr):
        w, v = w.contiguous(), v.contiguous()
        _dev(w, v, row_ptr)
        E, H, F = v.shape
        N = row_ptr.numel() - 1
        out = torch.empty(N, H, F, device=v.device, dtype=torch.float32)
        _chk(_lib.lib().singa_segment_wsum_fwd(_p(w), _p(v), _p(row_ptr), _p(out), N, H, F, _stream()),
             "singa_segment_wsum_fwd")
        ctx.save_for_backward(w, v, row_ptr)
        return out

    @staticmethod
    def backward(ctx, g):
        w, v, row_ptr = ctx.saved_tensors
        g = g.contiguous()
        E, H, F = v.shape
        N = row_ptr.numel() - 1
        gw, gv = torch.empty_like(w), torch.empty_like(v)
        _chk(_lib.lib().singa_segment_wsum_bwd(_p(g), _p(w), _p(v), _p(row_ptr), _p(gw), _p(gv), N, H, F, _stream()),
             "singa_segment_wsum_bwd")
        return gw, gv, None


def segment_sum_rows(v, row_ptr):
    """out[n, :] = sum of the rows v[row_ptr[n] : row_ptr[n+1], :] (no gradient; rows sorted by segment): the segmented
    sum kernel with unit weights, no atomics."""
    v = v.detach()
    _dev(v, row_ptr)
    E, F = v.shape
    N = row_ptr.numel() - 1
    out = torch.empty(N, 1, F, device=v.device, dtype=torch.float32)
    w = torch.ones(max(E, 1), 1, device=v.device, dtype=torch.float32)
    _chk(_lib.lib().singa_segment_wsum_fwd(_p(w), _p(v), _p(row_ptr), _p(out), N, 1, F, _stream()), "singa_segment_wsum_fwd")
    return out.view(N, F)


def knn_edge_attr(ln, seg_ptr, E0, n_real, offset, coeff):
    """n1: the kNN graph's edge features in their final layout (singa_knn_edge_attr): ln [n_real] lengths of the first n_real
    of the E0 row-sorted undirected edges (the rest: inert padding edges, zeros), seg_ptr [N + 1] int32 -> [E0 + N, 64]:
    -smear(len) at row e + i for edge e of node i, the row sums (the degree) at the self loops' rows."""
    ln, offset = ln.detach().contiguous(), offset.detach().contiguous()
    _dev(ln, seg_ptr, offset)
    N = seg_ptr.numel() - 1
    out = torch.empty(E0 + N, offset.numel(), device=ln.device, dtype=torch.float32)
    _chk(_lib.lib().singa_knn_edge_attr(_p(ln), _p(seg_ptr), int(n_real), _p(offset), float(coeff), _p(out), N, offset.numel(),
                                        _stream()), "singa_knn_edge_attr")
    return out


def knn_graph(pos, k, batch, ptr, max_nodes):
    """n1: torch_cluster.knn_graph(pos, k, batch, flow='target_to_source') (CP:293,330) -> [2, N * k] int64, row = centre,
    -1 where a slot does not exist (singa_knn_graph).  batch ids outside [0, len(ptr) - 1) = atoms of no molecule."""
    pos = pos.detach().to(torch.float32).contiguous()
    b32 = batch.to(torch.int32).contiguous()
    _dev(pos, b32)
    p64 = ptr.to(device=pos.device, dtype=torch.int64).contiguous()
    N = pos.shape[0]
    out = torch.empty(2, N * k, dtype=torch.int64, device=pos.device)
    _chk(_lib.lib().singa_knn_graph(_p(pos), _p(b32), _p(p64), p64.numel() - 1, N, int(k), int(max_nodes), _p(out[0]), _p(out[1]),
                                    _stream()), "singa_knn_graph")
    return out


def segment_wsum(w, v, row_ptr):
    """k15: out[n,H,F] = sum over the segment of w[e,H] * v[e,H,F]."""
    return _SegmentWSum.apply(w, v, row_ptr)


_grid_cache = {}


def _grid_factors(L, M, m_primary, device):
    """Separable S2-grid tables (P [RB,KIN], Q [RB,KIN], A [RA,2M+1]) on `device`, cached (so3.s2_grid_factors)."""
    key = (L, M, m_primary, str(device))
    if key not in _grid_cache:
        _grid_cache[key] = tuple(torch.tensor(t, dtype=torch.float32, device=device).contiguous()
                                 for t in so3.s2_grid_factors(L, M, m_primary))
    return _grid_cache[key]


class _S2ActEdge(torch.autograd.Function):
    """SeparableS2Activation on the attention grid [L][M] applied directly to the three SO(2)-conv GEMM outputs
    (m-primary).  h0 = [alpha inputs | gate | m=0 rows]; returns the activated m-primary tensor [E, KR*C]."""

    @staticmethod
    def forward(ctx, h0, h1, h2, gate_off, x_off, C, L, M):
        h0, h1, h2 = _rows(h0), _rows(h1), _rows(h2)
        _dev(h0, h1, h2)
        lay = so3.layout(L, M)
        P, Q, A = _grid_factors(L, M, True, h0.device)
        E = h0.shape[0]
        out = torch.empty(E, lay.KR * C, device=h0.device, dtype=torch.float32)
        seg, n = _capi.segs([(h0.data_ptr() + 4 * x_off, h0.stride(0), lay.seg_rows[0]),
                             (h1.data_ptr(), h1.stride(0), lay.seg_rows[1]),
                             (h2.data_ptr(), h2.stride(0), lay.seg_rows[2])])
        _chk(_lib.lib().singa_s2act_sep_fwd(seg, n, ctypes.c_void_p(h0.data_ptr() + 4 * gate_off), h0.stride(0), _p(P),
                                            _p(Q), _p(A), _p(out), E, C, L, _stream()), "singa_s2act_sep_fwd")
        ctx.save_for_backward(h0, h1, h2)
        ctx.cfg = (gate_off, x_off, C, L, M)
        return out

    @staticmethod
    def backward(ctx, g):
        h0, h1, h2 = ctx.saved_tensors
        gate_off, x_off, C, L, M = ctx.cfg
        lay = so3.layout(L, M)
        P, Q, A = _grid_factors(L, M, True, h0.device)
        g = g.contiguous()
        E = h0.shape[0]
        gx = torch.empty(E, lay.KR * C, device=g.device, dtype=torch.float32)
        gg = torch.empty(E, C, device=g.device, dtype=torch.float32)
        seg, n = _capi.segs([(h0.data_ptr() + 4 * x_off, h0.stride(0), lay.seg_rows[0]),
                             (h1.data_ptr(), h1.stride(0), lay.seg_rows[1]),
                             (h2.data_ptr(), h2.stride(0), lay.seg_rows[2])])
        _chk(_lib.lib().singa_s2act_sep_bwd(seg, n, ctypes.c_void_p(h0.data_ptr() + 4 * gate_off), h0.stride(0), _p(P),
                                            _p(Q), _p(A), _p(g), _p(gx), _p(gg), E, C, L, _stream()),
             "singa_s2act_sep_bwd")
        n0, n1 = lay.seg_rows[0] * C, lay.seg_rows[1] * C
        g0 = torch.zeros_like(h0)
        g0[:, gate_off:gate_off + C] = gg
        g0[:, x_off:] = gx[:, :n0]
        return g0, gx[:, n0:n0 + n1], gx[:, n0 + n1:], None, None, None, None, None


def s2act_edge(h0, h1, h2, gate_off, x_off, C, L, M=2):
    return _S2ActEdge.apply(h0, h1, h2, gate_off, x_off, C, L, M)


class _EdgeHead(torch.autograd.Function):
    """Everything that consumes the first SO(2) convolution (EF:1148-1178): attention logits (LayerNorm(A) ->
    SmoothLeakyReLU -> dot with alpha_dot, k9a) from the first heads*A columns of h0, and the separable S2 activation
    (k8) of [gate | m=0 rows] + h1 + h2.  One autograd node, so the gradient of h0 is assembled once (no zero-filled
    full-width buffers, no add of two partial gradients)."""

    @staticmethod
    def forward(ctx, h0, h1, h2, ln_w, ln_b, dot, heads, A, C, L, M, eps):
        ctx.params = (ln_w, ln_b, dot)
        h0, h1, h2 = _rows(h0), _rows(h1), _rows(h2)
        ln_w, ln_b, dot = ln_w.contiguous(), ln_b.contiguous(), dot.contiguous()
        _dev(h0, h1, h2, ln_w, ln_b, dot)
        lay = so3.layout(L, M)
        lib = _lib.lib()
        E = h0.shape[0]
        gate_off, x_off = heads * A, heads * A + C
        logits = torch.empty(E, heads, device=h0.device, dtype=torch.float32)
        _chk(lib.singa_alpha_logits_fwd(_p(h0), h0.stride(0), _p(ln_w), _p(ln_b), _p(dot), _p(logits), E, heads, A, eps,
                                        _stream()), "singa_alpha_logits_fwd")
        P, Q, Az = _grid_factors(L, M, True, h0.device)
        act = torch.empty(E, lay.KR * C, device=h0.device, dtype=torch.float32)
        seg, n = _capi.segs([(h0.data_ptr() + 4 * x_off, h0.stride(0), lay.seg_rows[0]),
                             (h1.data_ptr(), h1.stride(0), lay.seg_rows[1]),
                             (h2.data_ptr(), h2.stride(0), lay.seg_rows[2])])
        _chk(lib.singa_s2act_sep_fwd(seg, n, ctypes.c_void_p(h0.data_ptr() + 4 * gate_off), h0.stride(0), _p(P), _p(Q),
                                     _p(Az), _p(act), E, C, L, _stream()), "singa_s2act_sep_fwd")
        ctx.save_for_backward(h0, h1, h2, ln_w, ln_b, dot)
        ctx.cfg = (heads, A, C, L, M, eps)
        return logits, act

    @staticmethod
    def backward(ctx, g_logits, g_act):
        h0, h1, h2, ln_w, ln_b, dot = ctx.saved_tensors
        heads, A, C, L, M, eps = ctx.cfg
        lay = so3.layout(L, M)
        lib = _lib.lib()
        E = h0.shape[0]
        gate_off, x_off = heads * A, heads * A + C
        g_logits, g_act = g_logits.contiguous(), g_act.contiguous()
        # the gradients of the convolution's three outputs are assembled in place: h0's is [alpha inputs | gate | m = 0 rows],
        # written by two kernels into column blocks of one buffer (the concatenation they replaced cost 0.6 ms per step)
        n0, n1, n2 = (r * C for r in lay.seg_rows)
        g_h0 = torch.empty(E, x_off + n0, device=h0.device, dtype=torch.float32)
        g_h1 = torch.empty(E, n1, device=h0.device, dtype=torch.float32)
        g_h2 = torch.empty(E, n2, device=h0.device, dtype=torch.float32)
        nslots = lib.singa_alpha_logits_nslots(E)
        part = torch.empty(nslots, (2 + heads) * A, device=h0.device, dtype=torch.float32)
        _chk(lib.singa_alpha_logits_bwd_ld(_p(h0), h0.stride(0), _p(ln_w), _p(ln_b), _p(dot), _p(g_logits), _p(g_h0),
                                           g_h0.stride(0), _p(part), E, heads, A, eps, _stream()), "singa_alpha_logits_bwd_ld")
        pw, pb, pd = ctx.params
        g_lnw, g_lnb, g_dot = param_colsum(part, [(0, A, pw), (A, A, pb), (2 * A, heads * A, pd)])
        if g_dot is not None:
            g_dot = g_dot.view(heads, A)
        P, Q, Az = _grid_factors(L, M, True, h0.device)
        seg, n = _capi.segs([(h0.data_ptr() + 4 * x_off, h0.stride(0), lay.seg_rows[0]),
                             (h1.data_ptr(), h1.stride(0), lay.seg_rows[1]),
                             (h2.data_ptr(), h2.stride(0), lay.seg_rows[2])])
        gseg, _ = _capi.segs([(g_h0.data_ptr() + 4 * x_off, g_h0.stride(0), lay.seg_rows[0]),
                              (g_h1.data_ptr(), g_h1.stride(0), lay.seg_rows[1]),
                              (g_h2.data_ptr(), g_h2.stride(0), lay.seg_rows[2])])
        if E > 0:
            _chk(lib.singa_s2act_sep_bwd_seg(seg, n, ctypes.c_void_p(h0.data_ptr() + 4 * gate_off), h0.stride(0), _p(P), _p(Q),
                                             _p(Az), _p(g_act), gseg, ctypes.c_void_p(g_h0.data_ptr() + 4 * gate_off),
                                             g_h0.stride(0), E, C, L, _stream()), "singa_s2act_sep_bwd_seg")
        return (g_h0, g_h1, g_h2, g_lnw, g_lnb, g_dot, None, None, None, None, None, None)


def edge_head(h0, h1, h2, ln_w, ln_b, alpha_dot, heads, A, C, L, M=2, eps=1e-5):
    """-> (attention logits [E, heads], activated m-primary tensor [E, KR*C])"""
    return _EdgeHead.apply(h0, h1, h2, ln_w, ln_b, alpha_dot, heads, A, C, L, M, eps)


class _S2ActNode(torch.autograd.Function):
    """SeparableS2Activation on the FFN grid [L][L] for an l-primary node tensor [N,K,C] and gate [N,C]."""

    @staticmethod
    def forward(ctx, x, gate, L):
        x, gate = x.contiguous(), gate.contiguous()
        _dev(x, gate)
        P, Q, A = _grid_factors(L, L, False, x.device)
        N, K, C = x.shape
        out = torch.empty_like(x)
        seg, n = _capi.segs([(x.data_ptr(), K * C, K)])
        _chk(_lib.lib().singa_s2act_sep_fwd(seg, n, _p(gate), gate.stride(0), _p(P), _p(Q), _p(A), _p(out), N, C, L,
                                            _stream()), "singa_s2act_sep_fwd(node)")
        ctx.save_for_backward(x, gate)
        ctx.L = L
        return out

    @staticmethod
    def backward(ctx, g):
        x, gate = ctx.saved_tensors
        L = ctx.L
        P, Q, A = _grid_factors(L, L, False, x.device)
        g = g.contiguous()
        N, K, C = x.shape
        gx, gg = torch.empty_like(x), torch.empty_like(gate)
        seg, n = _capi.segs([(x.data_ptr(), K * C, K)])
        _chk(_lib.lib().singa_s2act_sep_bwd(seg, n, _p(gate), gate.stride(0), _p(P), _p(Q), _p(A), _p(g), _p(gx), _p(gg),
                                            N, C, L, _stream()), "singa_s2act_sep_bwd(node)")
        return gx, gg, None


def s2act_node(x, gate, L):
    return _S2ActNode.apply(x, gate, L)


_deg_cache = {}


def _degree_onehot(L, device):
    """[L+1, K] 0/1 matrix that sums coefficient rows by degree: a small GEMM instead of index_add_ (atomics, ~20 us for
    nine indices)."""
    key = ("onehot", L, str(device))
    if key not in _deg_cache:
        deg = torch.as_tensor(so3.layout(L, L).degree, dtype=torch.int64)
        _deg_cache[key] = torch.nn.functional.one_hot(deg, L + 1).t().to(torch.float32).contiguous().to(device)
    return _deg_cache[key]


def _degree_index(L, device):
    """degree l of every coefficient row, as a device tensor (cached: no host->device copy inside a graph capture)."""
    key = (L, str(device))
    if key not in _deg_cache:
        _deg_cache[key] = torch.as_tensor(so3.layout(L, L).degree, device=device, dtype=torch.int64)
    return _deg_cache[key]


class _SO3RMSNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, L, eps):
        ctx.params = (weight, bias)
        x, weight, bias = x.contiguous(), weight.contiguous(), bias.contiguous()
        _dev(x, weight, bias)
        N, K, C = x.shape
        y = torch.empty_like(x)
        _chk(_lib.lib().singa_so3_rmsnorm_fwd(_p(x), _p(weight), _p(bias), _p(y), N, C, L, eps, _stream()),
             "singa_so3_rmsnorm_fwd")
        ctx.save_for_backward(x, weight)
        ctx.L, ctx.eps = L, eps
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        L, eps = ctx.L, ctx.eps
        gy = gy.contiguous()
        N, K, C = x.shape
        nparts = _lib.lib().singa_so3_rmsnorm_nparts(N)
        gx = torch.empty_like(x)
        gwp = torch.empty(nparts, (L + 1) * C, device=x.device, dtype=torch.float32)       # already summed over each degree's rows
        gbp = torch.empty(nparts, C, device=x.device, dtype=torch.float32)
        _chk(_lib.lib().singa_so3_rmsnorm_bwd(_p(x), _p(weight), _p(gy), _p(gx), _p(gwp), _p(gbp), N, C, L, eps,
                                              _stream()), "singa_so3_rmsnorm_bwd")
        gw = param_colsum(gwp, [(0, (L + 1) * C, ctx.params[0])])[0]
        if gw is not None:
            gw = gw.view(L + 1, C)
        return gx, gw, param_colsum(gbp, [(0, C, ctx.params[1])])[0], None, None


def so3_rmsnorm(x, weight, bias, L, eps=1e-5):
    """k12: centred, degree-balanced RMS norm with per-degree affine weight and l=0 bias (EF:2155-2192, Q3)."""
    return _SO3RMSNorm.apply(x, weight, bias, L, eps)


class _SO3RMSNormSkip(torch.autograd.Function):
    """(norm(x), x): the norm together with the residual branch that leaves its input (x + f(norm(x)) of a TransBlockV2,
    EF:1383-1384, 1405-1406).  One autograd node for both uses of x, so that the backward kernel adds the residual's gradient
    to the norm's input gradient itself (singa_so3_rmsnorm_bwd_add) instead of autograd adding two [N, K, C] tensors afterwards."""

    @staticmethod
    def forward(ctx, x, weight, bias, L, eps):
        ctx.params = (weight, bias)
        xc, weight, bias = x.contiguous(), weight.contiguous(), bias.contiguous()
        _dev(xc, weight, bias)
        N, K, C = xc.shape
        y = torch.empty_like(xc)
        _chk(_lib.lib().singa_so3_rmsnorm_fwd(_p(xc), _p(weight), _p(bias), _p(y), N, C, L, eps, _stream()),
             "singa_so3_rmsnorm_fwd")
        ctx.save_for_backward(xc, weight)
        ctx.L, ctx.eps = L, eps
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, gy, gskip):
        x, weight = ctx.saved_tensors
        L, eps = ctx.L, ctx.eps
        N, K, C = x.shape
        lib = _lib.lib()
        if gy is None:                                  # (the norm's output unused: only the skip carries a gradient)
            return gskip, None, None, None, None
        gy = gy.contiguous()
        nparts = lib.singa_so3_rmsnorm_nparts(N)
        gx = torch.empty_like(x)
        gwp = torch.empty(nparts, (L + 1) * C, device=x.device, dtype=torch.float32)
        gbp = torch.empty(nparts, C, device=x.device, dtype=torch.float32)
        if gskip is not None:
            gskip = gskip.contiguous()
            _chk(lib.singa_so3_rmsnorm_bwd_add(_p(x), _p(weight), _p(gy), _p(gskip), _p(gx), _p(gwp), _p(gbp), N, C, L, eps,
                                               _stream()), "singa_so3_rmsnorm_bwd_add")
        else:
            _chk(lib.singa_so3_rmsnorm_bwd(_p(x), _p(weight), _p(gy), _p(gx), _p(gwp), _p(gbp), N, C, L, eps, _stream()),
                 "singa_so3_rmsnorm_bwd")
        gw = param_colsum(gwp, [(0, (L + 1) * C, ctx.params[0])])[0]
        if gw is not None:
            gw = gw.view(L + 1, C)
        return gx, gw, param_colsum(gbp, [(0, C, ctx.params[1])])[0], None, None


def so3_rmsnorm_skip(x, weight, bias, L, eps=1e-5):
    """-> (norm(x), x) as ONE autograd node: see _SO3RMSNormSkip."""
    return _SO3RMSNormSkip.apply(x, weight, bias, L, eps)


# ----------------------------------------------------------------------------------------------- fused graph attention
class _EdgeLogits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qp, wk, hk, cterm, edges, scale):
        qp, wk, hk, cterm = qp.contiguous(), wk.contiguous(), hk.contiguous(), cterm.contiguous()
        _dev(qp, wk, hk, cterm)
        N, H, D = qp.shape
        qk = torch.empty(wk.shape[0], H, device=qp.device, dtype=torch.float32)
        _chk(_lib.lib().singa_edge_logits_fwd(_p(qp), _p(wk), _p(hk), _p(cterm), _p(edges.row_ptr), _p(edges.col32),
                                              _p(qk), N, H, D, scale, _stream()), "singa_edge_logits_fwd")
        ctx.save_for_backward(qp, wk, hk)
        ctx.edges, ctx.scale = edges, scale
        return qk

    @staticmethod
    def backward(ctx, g):
        qp, wk, hk = ctx.saved_tensors
        e, N, H, D = ctx.edges, qp.shape[0], qp.shape[1], qp.shape[2]
        g = g.contiguous()
        g_qp, g_wk, g_hk = torch.empty_like(qp), torch.empty_like(wk), torch.empty_like(hk)
        g_c = torch.empty(N, H, device=g.device, dtype=torch.float32)
        _chk(_lib.lib().singa_edge_logits_bwd(_p(g), _p(qp), _p(wk), _p(hk), _p(e.row_ptr), _p(e.col32), _p(e.col_ptr),
                                              _p(e.eperm), _p(e.row32), _p(g_qp), _p(g_wk), _p(g_hk), _p(g_c), N, H, D,
                                              ctx.scale, _stream()), "singa_edge_logits_bwd")
        return g_qp, g_wk, g_hk, g_c, None, None


def edge_logits(qp, wk, hk, cterm, edges, scale):
    """qk[e,h] = scale * sum_d qp[row,h,d] wk[e,d] hk[col,h,d] + cterm[row,h]  (CP:61-65 with weight_k_lin hoisted)."""
    return _EdgeLogits.apply(qp, wk, hk, cterm, edges, scale)


class _GatherWSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alpha, wv, hv, edges):
        alpha, wv, hv = alpha.contiguous(), wv.contiguous(), hv.contiguous()
        _dev(alpha, wv, hv)
        N, H, F = hv.shape
        out = torch.empty(N, H, F, device=hv.device, dtype=torch.float32)
        _chk(_lib.lib().singa_gather_wsum_fwd(_p(alpha), _p(wv), _p(hv), _p(edges.row_ptr), _p(edges.col32), _p(out), N,
                                              H, F, _stream()), "singa_gather_wsum_fwd")
        ctx.save_for_backward(alpha, wv, hv)
        ctx.edges = edges
        return out

    @staticmethod
    def backward(ctx, g):
        alpha, wv, hv = ctx.saved_tensors
        e = ctx.edges
        N, H, F = hv.shape
        g = g.contiguous()
        g_a, g_wv, g_hv = torch.empty_like(alpha), torch.empty_like(wv), torch.empty_like(hv)
        _chk(_lib.lib().singa_gather_wsum_bwd(_p(g), _p(alpha), _p(wv), _p(hv), _p(e.row_ptr), _p(e.col32), _p(e.col_ptr),
                                              _p(e.eperm), _p(e.row32), _p(g_a), _p(g_wv), _p(g_hv), N, H, F, _stream()),
             "singa_gather_wsum_bwd")
        return g_a, g_wv, g_hv, None


def gather_wsum(alpha, wv, hv, edges):
    """out[n,h,f] = sum_e alpha[e,h] wv[e,f] hv[col_e,h,f]  (CP:70-74 with weight_v_lin hoisted)."""
    return _GatherWSum.apply(alpha, wv, hv, edges)


class _LnSilu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        ctx.params = (gamma, beta)
        x, gamma, beta = x.contiguous(), gamma.contiguous(), beta.contiguous()
        _dev(x, gamma, beta)
        C = x.shape[-1]
        M = x.numel() // C
        out = torch.empty_like(x)
        _chk(_lib.lib().singa_ln_silu_fwd(_p(x), _p(gamma), _p(beta), _p(out), M, C, eps, _stream()), "singa_ln_silu_fwd")
        ctx.save_for_backward(x, gamma, beta)
        ctx.eps = eps
        return out

    @staticmethod
    def backward(ctx, g):
        x, gamma, beta = ctx.saved_tensors
        g = g.contiguous()
        C = x.shape[-1]
        M = x.numel() // C
        lib = _lib.lib()
        gx = torch.empty_like(x)
        part = torch.empty(lib.singa_ln_silu_nparts(M), 2 * C, device=x.device, dtype=torch.float32)
        _chk(lib.singa_ln_silu_bwd(_p(x), _p(gamma), _p(beta), _p(g), _p(gx), _p(part), M, C, ctx.eps, _stream()),
             "singa_ln_silu_bwd")
        gg, gb = param_colsum(part, [(0, C, ctx.params[0]), (C, C, ctx.params[1])])
        return gx, gg, gb, None


def ln_silu(x, gamma, beta, eps=1e-5):
    """SiLU(LayerNorm(x)) over the last axis of 16 channels (k6a): one pass forward, one pass backward."""
    return _LnSilu.apply(x, gamma, beta, eps)


def dec_layer_step(x, w, k_cache, v_cache, pos, cross_k, cross_v, pad, beams):
    """One decoder layer for one new position of every beam-search row (k17; inference only, no autograd): three launches.
    `w`: dict of the layer's transposed weights as built by BeamSearch.KVDecoder; k_cache [R,4,P,32], v_cache [R,4,P,64];
    pos: int64 device scalar; cross_k [B,4,32,S], cross_v [B,4,S,64], pad [B,S] uint8."""
    _dev(x, k_cache, v_cache, cross_k, cross_v)
    lib, st = _lib.lib(), _stream()
    R, P, S = x.shape[0], k_cache.shape[2], cross_v.shape[2]
    y, z, out = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    a, c, f = w["self"], w["cross"], w["ffn"]
    _chk(lib.singa_dec_self_attn(_p(x), _p(a["wqkv_t"]), _p(a["bqkv"]), _p(a["wo_t"]), _p(a["bo"]), _p(a["gamma"]), _p(a["beta"]),
                                 _p(k_cache), _p(v_cache), _p(pos), R, P, _p(y), a["eps"], st), "singa_dec_self_attn")
    _chk(lib.singa_dec_cross_attn(_p(y), _p(c["wq_t"]), _p(c["bq"]), _p(cross_k), _p(cross_v), _p(pad), _p(c["wo_t"]), _p(c["bo"]),
                                  _p(c["gamma"]), _p(c["beta"]), R, beams, S, _p(z), c["eps"], st), "singa_dec_cross_attn")
    _chk(lib.singa_dec_ffn(_p(z), _p(f["w1_t"]), _p(f["b1"]), _p(f["w2_t"]), _p(f["b2"]), _p(f["gamma"]), _p(f["beta"]), R, _p(out),
                           f["eps"], st), "singa_dec_ffn")
    return out


class _MaskedSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s, mask, scale, heads):
        s = s.contiguous()
        _dev(s)
        assert mask.dtype == torch.bool and mask.dim() == 3 and mask.stride(2) == 1 and mask.is_cuda
        BH, T, S = s.shape
        p = torch.empty_like(s)
        assert mask.shape[0] * heads == BH and mask.shape[2] == S and mask.shape[1] in (1, T)
        ctx.mst = 0 if mask.shape[1] == 1 else mask.stride(1)          # a [B, 1, S] padding mask serves every query row
        _chk(_lib.lib().singa_masked_softmax_fwd(_p(s), _p(mask), mask.stride(0), ctx.mst, _p(p), BH, T, S, heads, scale,
                                                 _stream()), "singa_masked_softmax_fwd")
        ctx.save_for_backward(p, mask)
        ctx.scale, ctx.heads = scale, heads
        return p

    @staticmethod
    def backward(ctx, gp):
        p, mask = ctx.saved_tensors
        gp = gp.contiguous()
        BH, T, S = p.shape
        gs = torch.empty_like(p)
        _chk(_lib.lib().singa_masked_softmax_bwd(_p(p), _p(gp), _p(mask), mask.stride(0), ctx.mst, _p(gs), BH, T, S, ctx.heads,
                                                 ctx.scale, _stream()), "singa_masked_softmax_bwd")
        return gs, None, None, None


def masked_softmax(s, mask, scale, heads):
    """softmax(masked_fill(s * scale, mask, -1e9), -1) for attention scores s[B*heads, T, S] and a boolean mask [B, T, S]
    (k18): one pass instead of divide, masked_fill and softmax, forward and backward."""
    return _MaskedSoftmax.apply(s, mask, scale, heads)


def _tok_view(t, heads, D):
    """A token-major attention operand [B, T, heads, D]: dense, or a column block of a wider [B, T, P] buffer (the fused
    W_Q | W_K | W_V projection output).  Returns (tensor usable as it is, token pitch in floats) - a copy if the view
    cannot be addressed as (base, pitch)."""
    B, T = t.shape[0], t.shape[1]
    st = t.stride()
    ok = (st[3] == 1 and st[2] == D and st[1] % 4 == 0 and st[1] >= heads * D and (T == 1 or st[0] == T * st[1] or B == 1)
          and t.data_ptr() % 16 == 0)
    if B > 1 and T == 1:
        ok = ok and st[0] == st[1]
    if not ok:
        t = t.contiguous()
        return t, heads * D
    return t, st[1]


def _fused_blocks(ts):
    """True when the views ts (same leading shape) are consecutive column blocks of one row-pitched buffer."""
    p0, pitch = ts[0].data_ptr(), ts[0].stride(1)
    off = 0
    for t in ts:
        if t.stride(1) != pitch or t.stride(0) != ts[0].stride(0) or t.data_ptr() != p0 + 4 * off:
            return False
        off += t.shape[2] * t.shape[3]
    return pitch >= off


class _Attention(torch.autograd.Function):
    """softmax(masked_fill(q k^T * scale, mask, -1e9)) v on the MFMA (k19): scores never leave registers; the backward
    recomputes them from q, k and the stored log-sum-exp.  token_major: q / k / v / context are [B, T|S, heads, D] (as the
    projections produce them) instead of [B*heads, T|S, D] - no head transposes; q / k / v may then be column blocks of one
    fused projection output, and their gradients are written as the same column blocks of one buffer, so that the fused
    projection's backward takes them without a copy."""

    @staticmethod
    def forward(ctx, q, k, v, mask, scale, heads, token_major):
        assert mask.dtype == torch.bool and mask.dim() == 3 and mask.stride(2) == 1 and mask.is_cuda
        if token_major:
            B, T, _, DK = q.shape
            S, DV = v.shape[1], v.shape[3]
            BH = B * heads
            assert q.shape[2] == heads and k.shape == (B, S, heads, DK) and v.shape[:3] == (B, S, heads)
            (q, lq), (k, lk), (v, lv) = _tok_view(q, heads, DK), _tok_view(k, heads, DK), _tok_view(v, heads, DV)
            out = torch.empty(B, T, heads, DV, device=q.device, dtype=torch.float32)
        else:
            q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
            lq = lk = lv = 0
            BH, T, DK = q.shape
            S, DV = v.shape[1], v.shape[2]
            out = torch.empty(BH, T, DV, device=q.device, dtype=torch.float32)
        _dev(q, k, v)
        assert mask.shape[0] * heads == BH and mask.shape[2] == S and mask.shape[1] in (1, T)
        ctx.mst = 0 if mask.shape[1] == 1 else mask.stride(1)
        lse = torch.empty(BH, T, 2, device=q.device, dtype=torch.float32)       # (row maximum, 1 / row sum)
        _chk(_lib.lib().singa_attn_fwd(_p(q), _p(k), _p(v), _p(mask), mask.stride(0), ctx.mst, _p(out), _p(lse), BH, T, S, heads,
                                       DK, DV, int(token_major), lq, lk, lv, scale, _stream()), "singa_attn_fwd")
        ctx.save_for_backward(q, k, v, mask, out, lse)
        ctx.scale, ctx.heads, ctx.tm, ctx.dims, ctx.ld = scale, heads, bool(token_major), (BH, T, S, DK, DV), (lq, lk, lv)
        return out

    @staticmethod
    def backward(ctx, g):
        q, k, v, mask, out, lse = ctx.saved_tensors
        g = g.contiguous()
        BH, T, S, DK, DV = ctx.dims
        lq, lk, lv = ctx.ld

        def like(ts):
            """Gradients laid out like the operands: column blocks of ONE buffer where the operands are."""
            if ctx.tm and len(ts) > 1 and _fused_blocks(ts):
                B, rows, pitch = ts[0].shape[0], ts[0].shape[1], ts[0].stride(1)
                buf = torch.empty(B, rows, pitch, device=g.device, dtype=torch.float32)
                outs, off = [], 0
                for t in ts:
                    w = t.shape[2] * t.shape[3]
                    outs.append(buf[:, :, off:off + w].view(B, rows, t.shape[2], t.shape[3]))
                    off += w
                if off < pitch:                       # columns of the buffer no operand covers (none in the model's layouts)
                    buf[:, :, off:].zero_()
                return outs
            return [torch.empty(t.shape, device=g.device, dtype=torch.float32) for t in ts]

        if ctx.tm and T == S and _fused_blocks([q, k, v]):
            gq, gk, gv = like([q, k, v])
        elif ctx.tm and _fused_blocks([k, v]):
            (gq,), (gk, gv) = like([q]), like([k, v])
        else:
            (gq,), (gk,), (gv,) = like([q]), like([k]), like([v])
        if ctx.tm:
            lq, lk, lv = gq.stride(1), gk.stride(1), gv.stride(1)
            # the kernels address the gradients with the operands' pitches: they agree by construction
            assert (lq, lk, lv) == (q.stride(1), k.stride(1), v.stride(1))
        dsum = torch.empty(BH, T, device=q.device, dtype=torch.float32)
        _chk(_lib.lib().singa_attn_bwd(_p(q), _p(k), _p(v), _p(mask), mask.stride(0), ctx.mst, _p(out), _p(lse), _p(g), _p(gq),
                                       _p(gk), _p(gv), _p(dsum), BH, T, S, ctx.heads, DK, DV, int(ctx.tm), lq, lk, lv, ctx.scale,
                                       _stream()), "singa_attn_bwd")
        return gq, gk, gv, None, None, None, None


def attention(q, k, v, mask, scale, heads, token_major=False):
    """Dense attention core (k19) for q[B*heads,T,32], k[B*heads,S,32], v[B*heads,S,64] - or, token_major, q[B,T,heads,32],
    k[B,S,heads,32], v[B,S,heads,64] (dense, or column blocks of a fused projection output) -> context in the same layout
    - and a boolean mask [B, T|1, S]."""
    return _Attention.apply(q, k, v, mask, scale, heads, token_major)


class _LayerNorm256(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, r, gamma, beta, eps):
        ctx.params = (gamma, beta)
        a = a.contiguous()
        r = r.contiguous() if r is not None else None
        gamma, beta = gamma.contiguous(), beta.contiguous()
        _dev(a, gamma, beta)
        M = a.numel() // 256
        y = torch.empty_like(a)
        _chk(_lib.lib().singa_ln256_fwd(_p(a), _p(r) if r is not None else None, _p(gamma), _p(beta), _p(y), M, 256, eps,
                                        _stream()), "singa_ln256_fwd")
        ctx.save_for_backward(a, r, gamma)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, g):
        a, r, gamma = ctx.saved_tensors
        g = g.contiguous()
        M = a.numel() // 256
        lib = _lib.lib()
        gs = torch.empty_like(a)
        part = torch.empty(lib.singa_ln256_nparts(M), 512, device=a.device, dtype=torch.float32)
        _chk(lib.singa_ln256_bwd(_p(a), _p(r) if r is not None else None, _p(gamma), _p(g), _p(gs), _p(part), M, 256, ctx.eps,
                                 _stream()), "singa_ln256_bwd")
        gg, gb = param_colsum(part, [(0, 256, ctx.params[0]), (256, 256, ctx.params[1])])
        return gs, (gs if r is not None else None), gg, gb, None


def layer_norm_residual(a, r, ln):
    """ln(a + r) for an nn.LayerNorm `ln` (r may be None).  256-channel rows on the GPU take the fused kernel (k16): the sum
    is never materialised and the backward is one pass + a column sum."""
    if a.shape[-1] == 256 and tuple(ln.normalized_shape) == (256,) and (r is None or r.shape == a.shape):
        return _LayerNorm256.apply(a, r, ln.weight, ln.bias, ln.eps)
    return ln(a if r is None else a + r)


class _BiasSsp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, b):
        ctx.param = b
        u, b = u.contiguous(), b.contiguous()
        _dev(u, b)
        n = u.shape[-1]
        y = torch.empty_like(u)
        _chk(_lib.lib().singa_bias_ssp_fwd(_p(u), _p(b), _p(y), u.numel() // n, n, _stream()), "singa_bias_ssp_fwd")
        ctx.save_for_backward(u, b)
        return y

    @staticmethod
    def backward(ctx, g):
        u, b = ctx.saved_tensors
        g = g.contiguous()
        n = u.shape[-1]
        gu = torch.empty_like(u)
        _chk(_lib.lib().singa_bias_ssp_bwd(_p(u), _p(b), _p(g), _p(gu), u.numel() // n, n, _stream()), "singa_bias_ssp_bwd")
        return gu, param_colsum(gu.reshape(-1, n), [(0, n, ctx.param)])[0]


def bias_ssp(u, b):
    """softplus(u + b) - ln 2 (k15d): ShiftedSoftplus with the preceding Linear's bias folded in."""
    return _BiasSsp.apply(u, b)


class _EdgeMLPPair(torch.autograd.Function):
    """(W_k, W_v) = (L2k(ssp(L1k(attr))), L2v(ssp(L1v(attr)))) for all kNN edges (CP:41-48, 58, 68) on the f32 MFMA
    (k15c): one kernel forward for both nets, one kernel per net backward; only the attribute rows are kept for the
    backward pass, which recomputes the hidden units and reduces all four parameter gradients itself."""

    @staticmethod
    def forward(ctx, attr, w1k, b1k, w2k, b2k, w1v, b1v, w2v, b2v):
        ctx.params = (w1k, b1k, w2k, b2k, w1v, b1v, w2v, b2v)
        attr = attr.contiguous()
        fw = [t.contiguous() for t in (w1k, b1k, w2k, b2k, w1v, b1v, w2v, b2v)]      # nn.Linear's own layouts, no copies
        _dev(attr, *fw)
        E, CIN = attr.shape
        HK, HV = w1k.shape[0], w1v.shape[0]
        wk = torch.empty(E, HK, device=attr.device, dtype=torch.float32)
        wv = torch.empty(E, HV, device=attr.device, dtype=torch.float32)
        _chk(_lib.lib().singa_edge_mlp_fwd(_p(attr), *[_p(t) for t in fw], _p(wk), _p(wv), E, CIN, HK, HV, _stream()),
             "singa_edge_mlp_fwd")
        ctx.save_for_backward(attr, fw[0], fw[1], fw[2], fw[4], fw[5], fw[6])
        return wk, wv

    @staticmethod
    def backward(ctx, gk, gv):
        attr, w1tk, b1k, w2k, w1tv, b1v, w2v = ctx.saved_tensors
        E, CIN = attr.shape
        lib = _lib.lib()
        grads = []
        for g, w1t, b1, w2, params in ((gk, w1tk, b1k, w2k, ctx.params[:4]), (gv, w1tv, b1v, w2v, ctx.params[4:])):
            g = g.contiguous()
            H = w2.shape[0]
            row = H * CIN + H + H * H + H                 # [dW1 | db1 | dW2 | db2] in the parameters' layouts (include/singa_hip.h)
            part = torch.empty(lib.singa_edge_mlp_bwd_nparts(E, H), row, device=attr.device, dtype=torch.float32)
            _chk(lib.singa_edge_mlp_bwd(_p(attr), _p(g), _p(w1t), _p(b1), _p(w2), _p(part), E, CIN, H, _stream()),
                 "singa_edge_mlp_bwd")
            o1, o2, o3 = H * CIN, H * CIN + H, H * CIN + H + H * H
            gs = param_colsum(part, [(0, o1, params[0]), (o1, H, params[1]), (o2, H * H, params[2]), (o3, H, params[3])])
            grads += [gs[0].view(H, CIN) if gs[0] is not None else None, gs[1], gs[2].view(H, H) if gs[2] is not None else None, gs[3]]
        return (None, *grads)


def edge_mlp_pair(attr, k_net, v_net):
    """k_net / v_net: (first Linear, second Linear) of `weight_k_net` / `weight_v_net`."""
    return _EdgeMLPPair.apply(attr, k_net[0].weight, k_net[0].bias, k_net[1].weight, k_net[1].bias,
                              v_net[0].weight, v_net[0].bias, v_net[1].weight, v_net[1].bias)


def colsum(t):
    """Column sums of a [M, ...] tensor over dim 0 with the library's two-pass kernel.  torch's own long-column
    reductions (bias gradients of Linear layers, broadcast gradients, `t.sum(0)`) go through a multi-block kernel with
    global semaphores that returned garbage (1e12..1e36) for a few outputs per step under HIP-graph replay on this ROCm
    build; this kernel has no cross-launch state and a fixed summation order."""
    if t.shape[0] == 0:                       # nothing to add up (an empty edge / node set)
        return torch.zeros(t.shape[1:], device=t.device, dtype=torch.float32)
    t2 = t.reshape(t.shape[0], -1)
    _dev(t2)
    if t2.stride(1) != 1:
        t2 = t2.contiguous()
    M, n = t2.shape
    lib = _lib.lib()
    work = torch.empty(lib.singa_colsum_work(M, n), device=t.device, dtype=torch.float32)
    out = torch.empty(n, device=t.device, dtype=torch.float32)
    _chk(lib.singa_colsum(_p(t2), t2.stride(0), M, n, _p(work), _p(out), _stream()), "singa_colsum")
    return out.view(t.shape[1:])


class _GradSink:
    """Direct accumulation of parameter gradients (TrainStep turns it on around its backward pass).  A backward function
    whose output is the gradient of a leaf parameter that already owns a `.grad` buffer does not return it to autograd
    (which would launch one AccumulateGrad add per parameter on top of the reduction that produced it): column-sum
    reductions are queued and run as ONE multi-job launch pair that adds into the `.grad` buffers
    (singa_colsum_multi), GEMM-shaped weight gradients accumulate in the GEMM itself (beta = 1).  `flush()` must run after
    the backward pass and before anything reads the gradients.  Off (the default): every backward returns its gradients."""
    on = False
    found = None       # a dict while the engine records which parameters are produced by sink-aware backward functions
    jobs = []          # (src [M, n] kept alive until the flush, [(col0, flat .grad view)])
    forked = False     # weight-gradient launches of this pass are running on the second stream (_dw_side): flush joins
    @staticmethod
    def takes(*params):
        """True when every given parameter can receive its gradient directly."""
        if _GradSink.found is not None:
            for p in params:
                if p is not None and p.is_leaf and p.requires_grad:
                    _GradSink.found[id(p)] = p
        if not _GradSink.on:
            return False
        for p in params:
            if p is None or not (p.is_leaf and p.requires_grad) or p.grad is None or not p.grad.is_contiguous():
                return False
        return True

    @staticmethod
    def flush():
        if _GradSink.forked:
            cur = torch.cuda.current_stream()
            cur.wait_stream(branch_stream(cur.device))
            _GradSink.forked = False
        jobs, _GradSink.jobs = _GradSink.jobs, []
        jobs = [j for j in jobs if j[0].shape[0] > 0 and j[0].shape[1] > 0]
        if not jobs:
            return
        # jobs of one call run concurrently, so two jobs that add into the same buffer (a parameter used several times
        # per forward pass) go to different rounds, in queue order: a fixed summation order, no atomics
        seen, rounds = {}, []
        for job in jobs:
            r = max(seen.get(g.data_ptr(), 0) for _, g in job[1])
            for _, g in job[1]:
                seen[g.data_ptr()] = r + 1
            while len(rounds) <= r:
                rounds.append([])
            rounds[r].append(job)
        lib = _lib.lib()
        for jobs in rounds:
            nj = len(jobs)
            ns = sum(len(j[1]) for j in jobs)
            xs, lds, Ms, ns_ = ((ctypes.c_void_p * nj)(), (ctypes.c_longlong * nj)(), (ctypes.c_longlong * nj)(),
                                (ctypes.c_int * nj)())
            seg0, col0, dst = (ctypes.c_int * nj)(), (ctypes.c_int * ns)(), (ctypes.c_void_p * ns)()
            q = total = 0
            for k, (src, segs) in enumerate(jobs):
                xs[k], lds[k], Ms[k], ns_[k], seg0[k] = src.data_ptr(), src.stride(0), src.shape[0], src.shape[1], q
                total += lib.singa_colsum_multi_work(src.shape[0], src.shape[1])
                for c0, g in segs:
                    col0[q], dst[q] = c0, g.data_ptr()
                    q += 1
            work = torch.empty(max(total, 1), device=jobs[0][0].device, dtype=torch.float32)
            _chk(lib.singa_colsum_multi(nj, xs, lds, Ms, ns_, seg0, ns, col0, dst, _p(work), total, _stream()),
                 "singa_colsum_multi")


def param_colsum(src, targets):
    """Column sums of src [M, n] split over parameters: targets = [(col0, ncols, param)] covering 0..n in order.  Returns
    one gradient per target - None where the sum was queued to be added straight into param.grad (_GradSink).  Runs of
    neighbouring targets of the same kind share one job / one immediate reduction."""
    src = src.reshape(src.shape[0], math.prod(src.shape[1:]))
    if src.stride(1) != 1:
        src = src.contiguous()
    direct = [src.shape[0] > 0 and _GradSink.takes(t[2]) for t in targets]
    out = [None] * len(targets)
    i = 0
    while i < len(targets):
        j = i
        while j + 1 < len(targets) and direct[j + 1] == direct[i]:
            j += 1
        c0, c1 = targets[i][0], targets[j][0] + targets[j][1]
        blk = src[:, c0:c1]
        if direct[i]:
            _GradSink.jobs.append((blk, [(t[0] - c0, t[2].grad) for t in targets[i:j + 1]]))
        else:
            tot = colsum(blk)
            for k in range(i, j + 1):
                out[k] = tot[targets[k][0] - c0: targets[k][0] - c0 + targets[k][1]]
        i = j + 1
    return out


class _BiasAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, b):
        ctx.param = b
        return x + b

    @staticmethod
    def backward(ctx, g):
        n = g.shape[-1]
        return g, param_colsum(g.reshape(-1, n), [(0, n, ctx.param)])[0]


def bias_add(x, b):
    """x[..., n] + b[n] whose bias gradient is a replay-safe column sum."""
    return _BiasAdd.apply(x, b)


import os as _os


class _rocblas:
    """Inside: torch's GEMMs go to rocBLAS (the `cublas` backend) instead of hipBLASLt.  Only the generic dense-attention
    branch (head sizes other than the shipped 32 / 64) still multiplies through the library, see _BmmSmall."""

    def __enter__(self):
        torch.backends.cuda.preferred_blas_library("cublas")

    def __exit__(self, *a):
        torch.backends.cuda.preferred_blas_library("cublaslt")
        return False


class _BmmSmall(torch.autograd.Function):
    """torch.bmm for batches of small matrices (the dense attention of the decoder: 128 batches of 201 x 230 scores with
    32 / 64 channels), forward and both backward products through rocBLAS: hipBLASLt runs one 256x256 tile per batch
    there (QK^T 32 us vs 13 us, dP 40 vs 20, dV 21 vs 12, dK 18 vs 11: tools/lab/bmm_probe.py).  A plain torch.bmm would
    take whatever library is preferred at the time autograd runs its backward."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        with _rocblas():
            return torch.bmm(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        with _rocblas():
            ga = torch.bmm(g, b.transpose(1, 2)) if ctx.needs_input_grad[0] else None
            gb = torch.bmm(a.transpose(1, 2), g) if ctx.needs_input_grad[1] else None
        return ga, gb


def bmm_small(a, b):
    return _BmmSmall.apply(a, b)


# ---------------------------------------------------------------------------------------- k7 / k11: own f32 MFMA GEMM
# The BLAS-library evaluation of the same contractions lives in tests/lib_gemm.py (the tests' second opinion); the product has
# one path.  USE_SKINNY_SO3: the VALU kernels k11s for the 16 <-> 512 / 112 channel SO3 linears (the tests switch it off to
# reach the MFMA kernel's grouped-row problems on those shapes too).
USE_SKINNY_SO3 = True
_GEMM_SPLIT_ROWS = 2048


def _rows(t):
    """A 2-D operand with unit column stride and 16-byte aligned rows (column-block views qualify); a copy otherwise."""
    if t.stride(-1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16:
        t = t.contiguous()
    return t


def _gemm(items, a_rc, b_rc, splits=1):
    arr, n = _capi.gemm_probs(items)
    _chk(_lib.lib().singa_gemm_f32(arr, n, int(a_rc), int(b_rc), int(splits), _stream()), "singa_gemm_f32")


def _splits_for(rows, tiles=None):
    """Split count of a weight-gradient launch over `rows` rows: one split per _GEMM_SPLIT_ROWS rows.  With `tiles` (128 x 128
    output tiles of all problems) given and only 2-8 rounds of workgroups over the 256 CUs, the count is moved (up to 2x,
    at least 512 rows per split) to the one whose last round is fullest: 117 tiles x 9 splits = 4.11 rounds run as 5
    (656 us for the 16.5 k-edge shard of config 4, 566 us with 13 splits = 5.94 rounds; tools/lab/conv_split_probe.py)."""
    base = max(1, min(64, -(-rows // _GEMM_SPLIT_ROWS)))
    if not tiles or not 512 <= tiles * base < 2048:
        return base
    best, best_fill = base, 0.0
    for S in range(base, max(base, min(64, rows // 512, 2 * base)) + 1):
        w = tiles * S
        fill = w / (-(-w // 256) * 256)
        if fill > best_fill + 0.02:
            best, best_fill = S, fill
    return best


def _splits_few(rows, tiles):
    """Split count of a weight-gradient launch with FEW output tiles (an SO(2) convolution's m = 0 block alone, or its two
    complex blocks: 10-50 tiles).  Model: the launch runs in rounds of 512 workgroups (two per CU), a workgroup costs its
    rows / S reduction rows plus ~64 rows' worth of prologue and epilogue; the cheapest S of 1..64 with at least 128 rows per
    split (50 tiles x 11 splits = 1.07 rounds ran as two: 393 us for the 13 k-edge shard's conv2, tools/lab/cgemm_probe.py)."""
    best, best_cost = 1, None
    for S in range(1, max(1, min(64, rows // 128)) + 1):
        cost = -(-tiles * S // 512) * (-(-rows // S) + 64)
        if best_cost is None or cost < best_cost:
            best, best_cost = S, cost
    return best


def gemm_nt(x, w, bias=None, out=None):
    """y = x w^T (+ bias) on the library's own MFMA kernel; x [M, K] (row-strided views allowed), w [N, K]."""
    x, w = _rows(x), _rows(w)
    _dev(x, w)
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, device=x.device, dtype=torch.float32)
    _gemm([dict(a=x.data_ptr(), lda=x.stride(0), b=w.data_ptr(), ldb=w.stride(0), c=out.data_ptr(), ldc=out.stride(0),
                bias=bias.data_ptr() if bias is not None else None, I=M, J=N, R=K)], True, True)
    return out


class _SO2Linear3(torch.autograd.Function):
    """The three contractions of one SO(2) convolution (m = 0 with bias, m = 1, m = 2; EF:807-875 with the +-m
    recombination folded into the block weights) on column blocks of ONE m-primary edge matrix X [E, n0+n1+n2]: one launch
    of the library's f32 MFMA GEMM (k7) forward, one for dX (written straight into the column blocks of one buffer), one
    split-reduction launch + one column sum for the three weight gradients.  The results are column blocks of one
    [E, o0+o1+o2] buffer; their consumers take (pointer, row pitch) pairs."""

    @staticmethod
    def forward(ctx, X, w0, b0, w1, w2, n0, n1):
        ctx.params = (w0, w1, w2, b0)
        X = _rows(X)
        ws = [_rows(w0), _rows(w1), _rows(w2)]
        b0 = b0.contiguous()
        _dev(X, *ws, b0)
        E, nin = X.shape
        ins = (n0, n1, nin - n0 - n1)
        outs = tuple(w.shape[0] for w in ws)
        assert all(w.shape[1] == k for w, k in zip(ws, ins))
        H = torch.empty(E, sum(outs), device=X.device, dtype=torch.float32)
        items, ai, ci = [], 0, 0
        for w, k, o, b in zip(ws, ins, outs, (b0, None, None)):
            items.append(dict(a=X.data_ptr() + 4 * ai, lda=X.stride(0), b=w.data_ptr(), ldb=w.stride(0),
                              c=H.data_ptr() + 4 * ci, ldc=H.stride(0), bias=b.data_ptr() if b is not None else None,
                              I=E, J=o, R=k))
            ai, ci = ai + k, ci + o
        if E > 0:
            _gemm(items, True, True)
        ctx.save_for_backward(X, *ws)
        ctx.ins, ctx.outs = ins, outs
        o0, o1 = outs[0], outs[0] + outs[1]
        return H[:, :o0], H[:, o0:o1], H[:, o1:]

    @staticmethod
    def backward(ctx, g0, g1, g2):
        X, w0, w1, w2 = ctx.saved_tensors
        ins, outs = ctx.ins, ctx.outs
        E = X.shape[0]
        gs = [_rows(g) for g in (g0, g1, g2)]
        ws = (w0, w1, w2)
        gX = None
        if ctx.needs_input_grad[0]:
            gX = torch.empty_like(X)
            items, ai = [], 0
            for g, w, k, o in zip(gs, ws, ins, outs):       # dX_blk = g_blk @ w_blk: A = g [E, o] (RC), B = w [o, k] ([R][J])
                items.append(dict(a=g.data_ptr(), lda=g.stride(0), b=w.data_ptr(), ldb=w.stride(0),
                                  c=gX.data_ptr() + 4 * ai, ldc=gX.stride(0), I=E, J=k, R=o))
                ai += k
            if E > 0:
                _gemm(items, True, False)
        # dW_blk = g_blk^T X_blk: reduction over the edges, split over workgroups into dense partial slabs
        sizes = [o * k for o, k in zip(outs, ins)]
        tot = sum(sizes)
        row = tot + outs[0]                      # + the m = 0 bias gradient: column sums of g0, taken inside the same launch
        S = _splits_for(E, sum(-(-o // 128) * -(-k // 128) for o, k in zip(outs, ins)))
        part = torch.empty(S, row, device=X.device, dtype=torch.float32)
        items, ai, off = [], 0, 0
        for n, (g, k, o, sz) in enumerate(zip(gs, ins, outs, sizes)):
            items.append(dict(a=g.data_ptr(), lda=g.stride(0), b=X.data_ptr() + 4 * ai, ldb=X.stride(0),
                              c=part.data_ptr() + 4 * off, ldc=k, I=o, J=k, R=E, c_split_stride=row,
                              asum=(part.data_ptr() + 4 * tot) if n == 0 else None, asum_stride=row))
            ai, off = ai + k, off + sz
        pw = ctx.params[:3]
        if E > 0:
            _gemm(items, False, False, S)
            offs = [0, sizes[0], sizes[0] + sizes[1]]
            res = param_colsum(part, [(off, sz, p) for off, sz, p in zip(offs, sizes, pw)] + [(tot, outs[0], ctx.params[3])])
            gws, gb = res[:3], res[3]
        else:
            gws = list(torch.zeros(tot, device=X.device, dtype=torch.float32).split(sizes))
            gb = torch.zeros(outs[0], device=X.device, dtype=torch.float32)
        gws = [g.view(o, k) if g is not None else None for g, o, k in zip(gws, outs, ins)]
        return gX, gws[0], gb, gws[1], gws[2], None, None


def _cgemm(items, a_rc, b_rc, splits=1):
    arr, n = _capi.cgemm_probs(items)
    _chk(_lib.lib().singa_cgemm3m_f32(arr, n, int(a_rc), int(b_rc), int(splits), _stream()), "singa_cgemm3m_f32")


class _SO2Conv3M(torch.autograd.Function):
    """One SO(2) convolution (EF:807-875) on the m-primary edge matrix X [E, n0 + n1 + n2], the m = 1, 2 blocks as COMPLEX
    products in three real multiplications (k7c, `singa_cgemm3m_f32`) on the modules' own fc weights [2N, K] (Wr = w[:N],
    Wi = w[N:]; no block weight is built), the m = 0 block on the real GEMM (k7): two launches forward, two for dX, two
    split reductions + one column sum for the weight gradients, which arrive in the parameters' own layout."""

    @staticmethod
    def forward(ctx, X, w0, b0, w1, w2, n0, n1):
        ctx.params = (w0, w1, w2, b0)
        X = _rows(X)
        ws = [_rows(w0), w1.contiguous(), w2.contiguous()]
        b0 = b0.contiguous()
        _dev(X, *ws, b0)
        E, nin = X.shape
        ins = (n0, n1, nin - n0 - n1)
        outs = tuple(w.shape[0] for w in ws)
        assert ws[0].shape[1] == n0 and all(2 * w.shape[1] == k and w.shape[0] % 2 == 0 for w, k in zip(ws[1:], ins[1:]))
        H = torch.empty(E, sum(outs), device=X.device, dtype=torch.float32)
        if E > 0:
            _gemm([dict(a=X.data_ptr(), lda=X.stride(0), b=ws[0].data_ptr(), ldb=ws[0].stride(0), c=H.data_ptr(),
                        ldc=H.stride(0), bias=b0.data_ptr(), I=E, J=outs[0], R=n0)], True, True)
            items, ai, ci = [], n0, outs[0]
            for w, k, o in zip(ws[1:], ins[1:], outs[1:]):
                K, N = k // 2, o // 2
                items.append(dict(a=X.data_ptr() + 4 * ai, lda=X.stride(0), a_im=K, b=w.data_ptr(), ldb=K, b_im=N * K,
                                  c=H.data_ptr() + 4 * ci, ldc=H.stride(0), c_im=N, I=E, J=N, R=K, sigma=1.0))
                ai, ci = ai + k, ci + o
            _cgemm(items, True, True)
        ctx.save_for_backward(X, *ws)
        ctx.ins, ctx.outs = ins, outs
        o0, o1 = outs[0], outs[0] + outs[1]
        return H[:, :o0], H[:, o0:o1], H[:, o1:]

    @staticmethod
    def backward(ctx, g0, g1, g2):
        X, w0, w1, w2 = ctx.saved_tensors
        ins, outs = ctx.ins, ctx.outs
        E = X.shape[0]
        gs = [_rows(g) for g in (g0, g1, g2)]
        ws = (w0, w1, w2)
        # weight gradients: reductions over the edges, split over workgroups into dense partial slabs (the fc weights' layouts)
        sizes = [outs[0] * ins[0], outs[1] * ins[1] // 2, outs[2] * ins[2] // 2]
        tot = sum(sizes)
        # the two launches have few tiles each (conv1: 16 and 14): each gets the split count that fills the CUs
        S0 = _splits_few(E, -(-outs[0] // 128) * -(-ins[0] // 128))
        Sc = _splits_few(E, sum(-(-(o // 2) // 128) * -(-(k // 2) // 64) for o, k in zip(outs[1:], ins[1:])))
        pw = ctx.params[:3]
        if E > 0:
            row0, rowc = sizes[0] + outs[0], sizes[1] + sizes[2]
            with _dw_side(X, *gs, params=ctx.params) as side:
                part0 = torch.empty(S0, row0, device=X.device, dtype=torch.float32)
                partc = torch.empty(Sc, rowc, device=X.device, dtype=torch.float32)
                _gemm([dict(a=gs[0].data_ptr(), lda=gs[0].stride(0), b=X.data_ptr(), ldb=X.stride(0), c=part0.data_ptr(),
                            ldc=ins[0], I=outs[0], J=ins[0], R=E, c_split_stride=row0, asum=part0.data_ptr() + 4 * sizes[0],
                            asum_stride=row0)], False, False, S0)
                items, ai, off = [], ins[0], 0
                for g, k, o, sz in zip(gs[1:], ins[1:], outs[1:], sizes[1:]):
                    K, N = k // 2, o // 2
                    items.append(dict(a=g.data_ptr(), lda=g.stride(0), a_im=N, b=X.data_ptr() + 4 * ai, ldb=X.stride(0), b_im=K,
                                      c=partc.data_ptr() + 4 * off, ldc=K, c_im=N * K, I=N, J=K, R=E, sigma=-1.0,
                                      c_split_stride=rowc))
                    ai, off = ai + k, off + sz
                _cgemm(items, False, False, Sc)
                side.keep(part0, partc)
            r0 = param_colsum(part0, [(0, sizes[0], pw[0]), (sizes[0], outs[0], ctx.params[3])])
            rc = param_colsum(partc, [(0, sizes[1], pw[1]), (sizes[1], sizes[2], pw[2])])
            gws, gb = [r0[0], rc[0], rc[1]], r0[1]
        else:
            gws = list(torch.zeros(tot, device=X.device, dtype=torch.float32).split(sizes))
            gb = torch.zeros(outs[0], device=X.device, dtype=torch.float32)
        gX = None
        if ctx.needs_input_grad[0]:
            gX = torch.empty_like(X)
            if E > 0:
                _gemm([dict(a=gs[0].data_ptr(), lda=gs[0].stride(0), b=w0.data_ptr(), ldb=w0.stride(0), c=gX.data_ptr(),
                            ldc=gX.stride(0), I=E, J=ins[0], R=outs[0])], True, False)
                items, ai = [], ins[0]
                for g, w, k, o in zip(gs[1:], ws[1:], ins[1:], outs[1:]):
                    K, N = k // 2, o // 2
                    items.append(dict(a=g.data_ptr(), lda=g.stride(0), a_im=N, b=w.data_ptr(), ldb=K, b_im=N * K,
                                      c=gX.data_ptr() + 4 * ai, ldc=gX.stride(0), c_im=K, I=E, J=K, R=N, sigma=-1.0))
                    ai += k
                _cgemm(items, True, False)
        shapes = [(outs[0], ins[0]), (outs[1], ins[1] // 2), (outs[2], ins[2] // 2)]
        gws = [g.view(*sh) if g is not None else None for g, sh in zip(gws, shapes)]
        return gX, gws[0], gb, gws[1], gws[2], None, None


def so2_conv3m(X, w0, b0, w1, w2, n0, n1):
    """The SO(2) convolution on the fc weights themselves (w1, w2: [2N, K] = [Wr; Wi]) - see _SO2Conv3M."""
    return _SO2Conv3M.apply(X, w0, b0, w1, w2, n0, n1)


def so2_linear3(X, w0, b0, w1, w2, n0, n1):
    """(X[:, :n0] w0^T + b0, X[:, n0:n0+n1] w1^T, X[:, n0+n1:] w2^T) - see _SO2Linear3."""
    return _SO2Linear3.apply(X, w0, b0, w1, w2, n0, n1)


class _SO3Linear(torch.autograd.Function):
    """SO3_LinearV2 (EF:624-674) on [N, K, C] coefficient rows with the library's f32 MFMA GEMM (k11): one launch with one
    problem per degree l (its 2l+1 rows of every node form a grouped row set, so the [N, K, out] result is written in
    place, bias on the l = 0 row), one launch for dX, one split-reduction launch + column sum for the per-degree weight
    gradients.  No expanded [K, out, in] weight, no transposed copies."""

    @staticmethod
    def forward(ctx, x, weight, bias, L, addend=None):
        ctx.params = (weight, bias)
        x, weight, bias = x.contiguous(), weight.contiguous(), bias.contiguous()
        _dev(x, weight, bias)
        N, K, cin = x.shape
        cout = weight.shape[1]
        out = torch.empty(N, K, cout, device=x.device, dtype=torch.float32)
        ctx.has_addend = addend is not None
        if addend is not None:                    # a residual [N, K, cout] added in the GEMM's epilogue (MFMA path only)
            addend = addend.contiguous()
            _dev(addend)
            assert addend.shape == out.shape and not (USE_SKINNY_SO3 and cin == 16 and cout == 512)
        if USE_SKINNY_SO3 and cin == 16 and cout == 512:
            # k11s: a 16-long contraction - VALU kernel, thread = output channel, whole 2 KB rows per store
            _chk(_lib.lib().singa_so3_skinny_expand(_p(x), _p(weight), cout * cin, cin, 1, _p(bias), _p(out), N, cout, L,
                                                    _stream()), "singa_so3_skinny_expand")
        else:
            items = []
            for l in range(L + 1):
                n = 2 * l + 1
                items.append(dict(a=x.data_ptr() + 4 * l * l * cin, lda=cin, a_group=n, a_group_ld=K * cin,
                                  b=weight.data_ptr() + 4 * l * cout * cin, ldb=cin,
                                  c=out.data_ptr() + 4 * l * l * cout, ldc=cout, c_group=n, c_group_ld=K * cout,
                                  bias=bias.data_ptr() if l == 0 else None,
                                  addend=(addend.data_ptr() + 4 * l * l * cout) if addend is not None else None,
                                  I=N * n, J=cout, R=cin))
            if N > 0:
                _gemm(items, True, True)
        ctx.save_for_backward(x, weight)
        ctx.L = L
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        L = ctx.L
        g = g.contiguous()
        N, K, cin = x.shape
        cout = weight.shape[1]
        lib = _lib.lib()
        # (wide, 16) pairs the VALU kernels are built for: the feed-forward block's 512 and the output projection's 112
        skinny = USE_SKINNY_SO3 and N > 0 and ((cin == 16 and cout == 512) or (cin in (512, 112) and cout == 16))
        wide = max(cin, cout)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            if skinny and cout == 16:
                # k11s: d x = g expanded through weight[l][u][c] - again a 16-long contraction
                _chk(lib.singa_so3_skinny_expand(_p(g), _p(weight), cout * cin, 1, cin, None, _p(gx), N, wide, L, _stream()),
                     "singa_so3_skinny_expand(dx)")
            else:
                items = []
                for l in range(L + 1):
                    n = 2 * l + 1
                    items.append(dict(a=g.data_ptr() + 4 * l * l * cout, lda=cout, a_group=n, a_group_ld=K * cout,
                                      b=weight.data_ptr() + 4 * l * cout * cin, ldb=cin,
                                      c=gx.data_ptr() + 4 * l * l * cin, ldc=cin, c_group=n, c_group_ld=K * cin,
                                      I=N * n, J=cin, R=cout))
                if N > 0:
                    _gemm(items, True, False)
        if skinny:
            # k11s: the weight gradient as a VALU reduction over whole 2 KB rows of the 512-channel tensor (16 -> 512:
            # small = x, big = g, rows [l][c][u] + the bias gradient; 512 -> 16: small = g, big = x, rows [l][u][c])
            wide_out = cout == wide
            wsz = (L + 1) * 16 * wide
            part = torch.empty(lib.singa_so3_skinny_nparts(N, L, wide), wsz + (wide if wide_out else 0), device=x.device,
                               dtype=torch.float32)
            small, big = (x, g) if wide_out else (g, x)
            _chk(lib.singa_so3_skinny_reduce(_p(small), _p(big), _p(part), N, wide, L, int(wide_out), int(wide_out), _stream()),
                 "singa_so3_skinny_reduce")
            if wide_out:
                gw, gb = param_colsum(part, [(0, wsz, ctx.params[0]), (wsz, wide, ctx.params[1])])
            else:
                gw = param_colsum(part, [(0, wsz, ctx.params[0])])[0]
                gb = param_colsum(g[:, 0, :], [(0, cout, ctx.params[1])])[0]
            return gx, (gw.view(L + 1, cout, cin) if gw is not None else None), gb, None, (g if ctx.has_addend else None)
        # dW_l = sum over the (node, row) pairs of degree l of g_row^T x_row
        sz = cout * cin
        S = _splits_for(N * (2 * L + 1))
        part = torch.empty(S, (L + 1) * sz, device=x.device, dtype=torch.float32)
        items = []
        for l in range(L + 1):
            n = 2 * l + 1
            items.append(dict(a=g.data_ptr() + 4 * l * l * cout, lda=cout, a_group=n, a_group_ld=K * cout,
                              b=x.data_ptr() + 4 * l * l * cin, ldb=cin, b_group=n, b_group_ld=K * cin,
                              c=part.data_ptr() + 4 * l * sz, ldc=cin, I=cout, J=cin, R=N * n, c_split_stride=(L + 1) * sz))
        if N > 0:
            _gemm(items, False, False, S)
            gw = param_colsum(part, [(0, (L + 1) * sz, ctx.params[0])])[0]
            if gw is not None:
                gw = gw.view(L + 1, cout, cin)
        else:
            gw = torch.zeros(L + 1, cout, cin, device=x.device, dtype=torch.float32)
        return gx, gw, param_colsum(g[:, 0, :], [(0, cout, ctx.params[1])])[0], None, (g if ctx.has_addend else None)


def so3_linear(x, weight, bias, L, addend=None):
    """SO3_LinearV2 (EF:624-674); `addend` [N, K, out]: a residual added in the same launch (EF:1383-1384, 1405-1406)."""
    return _SO3Linear.apply(x, weight, bias, L, addend)


class _FFNTail(torch.autograd.Function):
    """The back half of the feed-forward block, a = SeparableS2Activation(h, gate) -> SO3_LinearV2(512 -> 16)(a) (+ residual)
    (EF:256-262, 1405-1406), as ONE autograd node, so that the backward pass touches the [N, K, 512] tensors once less: the
    gradient of `a` (the linear's input gradient, a 16-long contraction per row) is formed inside the activation's backward kernel
    (singa_s2act_ffn_bwd) instead of being written by an expand launch and read back.  Forward: the two existing launches."""

    @staticmethod
    def forward(ctx, h, gate, weight, bias, L, addend=None):
        ctx.params = (weight, bias)
        h, gate, weight, bias = h.contiguous(), gate.contiguous(), weight.contiguous(), bias.contiguous()
        _dev(h, gate, weight, bias)
        N, K, cin = h.shape
        cout = weight.shape[1]
        assert cin == 512 and cout == 16 and weight.shape[2] == cin
        lib = _lib.lib()
        P, Q, A = _grid_factors(L, L, False, h.device)
        a = torch.empty_like(h)
        seg, n = _capi.segs([(h.data_ptr(), K * cin, K)])
        _chk(lib.singa_s2act_sep_fwd(seg, n, _p(gate), gate.stride(0), _p(P), _p(Q), _p(A), _p(a), N, cin, L, _stream()),
             "singa_s2act_sep_fwd(node)")
        out = torch.empty(N, K, cout, device=h.device, dtype=torch.float32)
        ctx.has_addend = addend is not None
        if addend is not None:
            addend = addend.contiguous()
            _dev(addend)
            assert addend.shape == out.shape
        items = []
        for l in range(L + 1):
            nl = 2 * l + 1
            items.append(dict(a=a.data_ptr() + 4 * l * l * cin, lda=cin, a_group=nl, a_group_ld=K * cin,
                              b=weight.data_ptr() + 4 * l * cout * cin, ldb=cin,
                              c=out.data_ptr() + 4 * l * l * cout, ldc=cout, c_group=nl, c_group_ld=K * cout,
                              bias=bias.data_ptr() if l == 0 else None,
                              addend=(addend.data_ptr() + 4 * l * l * cout) if addend is not None else None,
                              I=N * nl, J=cout, R=cin))
        if N > 0:
            _gemm(items, True, True)
        ctx.save_for_backward(h, gate, a, weight)
        ctx.L = L
        return out

    @staticmethod
    def backward(ctx, g):
        h, gate, a, weight = ctx.saved_tensors
        L = ctx.L
        g = g.contiguous()
        N, K, cin = h.shape
        cout = weight.shape[1]
        lib = _lib.lib()
        P, Q, _ = _grid_factors(L, L, False, h.device)
        gh, gg = torch.empty_like(h), torch.empty_like(gate)
        if N > 0:
            _chk(lib.singa_s2act_ffn_bwd(_p(h), _p(gate), gate.stride(0), _p(P), _p(Q), _p(g), _p(weight), _p(gh), _p(gg), N, cin, L,
                                         _stream()), "singa_s2act_ffn_bwd")
        # the linear's weight gradient: rows [l][u][c] summed over the nodes (k11s reduce: small = g, big = a)
        wsz = (L + 1) * 16 * cin
        gw = None
        if N > 0:
            part = torch.empty(lib.singa_so3_skinny_nparts(N, L, cin), wsz, device=h.device, dtype=torch.float32)
            _chk(lib.singa_so3_skinny_reduce(_p(g), _p(a), _p(part), N, cin, L, 0, 0, _stream()), "singa_so3_skinny_reduce")
            gw = param_colsum(part, [(0, wsz, ctx.params[0])])[0]
        else:
            gw = torch.zeros(wsz, device=h.device, dtype=torch.float32)
        gb = param_colsum(g[:, 0, :], [(0, cout, ctx.params[1])])[0]
        return gh, gg, (gw.view(L + 1, cout, cin) if gw is not None else None), gb, None, (g if ctx.has_addend else None)


def ffn_tail(h, gate, weight, bias, L, addend=None):
    """SO3_LinearV2(512 -> 16)(SeparableS2Activation(h, gate)) (+ addend) - see _FFNTail."""
    return _FFNTail.apply(h, gate, weight, bias, L, addend)


class _GroupedLinear3(torch.autograd.Function):
    """The three grouped 1x1 Conv1d layers of the graph attention (k_lin, q_lin, v_lin: CP:27-29, 55-57) on the SAME node
    rows h [N, heads * ig] as ONE launch of the own MFMA GEMM (k7) with one problem per (layer, head) - h's column block
    of the head times that head's [og, ig] weight block into the head's column block of the layer's output - instead of
    three batched library GEMMs; backward: the three input-gradient contributions as three chained launches (each adds the
    previous one's result in its epilogue: no separate additions) and one split-reduction launch for all weight gradients."""

    @staticmethod
    def forward(ctx, h, wk, wq, wv, heads):
        ctx.params = (wk, wq, wv)
        h = _rows(h)
        ws = [_rows(w.view(w.shape[0], w.shape[1])) for w in (wk, wq, wv)]       # [heads * og, ig] (the Conv1d weight as it is)
        _dev(h, *ws)
        N, ig = h.shape[0], ws[0].shape[1]
        ogs = [w.shape[0] // heads for w in ws]
        outs = [torch.empty(N, heads, og, device=h.device, dtype=torch.float32) for og in ogs]
        items = []
        for w, o, og in zip(ws, outs, ogs):
            for g in range(heads):
                items.append(dict(a=h.data_ptr() + 4 * g * ig, lda=h.stride(0), b=w.data_ptr() + 4 * g * og * ig, ldb=ig,
                                  c=o.data_ptr() + 4 * g * og, ldc=heads * og, I=N, J=og, R=ig))
        if N > 0:
            _gemm(items, True, True)
        ctx.save_for_backward(h, *ws)
        ctx.heads, ctx.ogs = heads, ogs
        return tuple(outs)

    @staticmethod
    def backward(ctx, gk, gq, gv):
        h, *ws = ctx.saved_tensors
        heads, ogs = ctx.heads, ctx.ogs
        N, ig = h.shape[0], ws[0].shape[1]
        gs = [_rows(g.reshape(N, heads * og)) for g, og in zip((gk, gq, gv), ogs)]
        gh = None
        if ctx.needs_input_grad[0] and N > 0:
            prev = None
            for g, w, og in zip(gs, ws, ogs):                        # dh += g_layer W_layer, head by head; chained through `addend`
                cur = torch.empty(N, heads * ig, device=h.device, dtype=torch.float32)
                items = []
                for hd in range(heads):
                    items.append(dict(a=g.data_ptr() + 4 * hd * og, lda=g.stride(0), b=w.data_ptr() + 4 * hd * og * ig, ldb=ig,
                                      c=cur.data_ptr() + 4 * hd * ig, ldc=heads * ig, I=N, J=ig, R=og,
                                      addend=(prev.data_ptr() + 4 * hd * ig) if prev is not None else None))
                _gemm(items, True, False)
                prev = cur
            gh = prev
        elif ctx.needs_input_grad[0]:
            gh = torch.zeros_like(h)
        sizes = [heads * og * ig for og in ogs]
        tot = sum(sizes)
        if N > 0:
            S = _tn_splits(N, max(ogs), ig)
            part = torch.empty(S, tot, device=h.device, dtype=torch.float32)
            items, off = [], 0
            for g, og in zip(gs, ogs):
                for hd in range(heads):                              # dW[layer][head] [og, ig] = g_head^T h_head
                    items.append(dict(a=g.data_ptr() + 4 * hd * og, lda=g.stride(0), b=h.data_ptr() + 4 * hd * ig, ldb=h.stride(0),
                                      c=part.data_ptr() + 4 * off, ldc=ig, I=og, J=ig, R=N, c_split_stride=tot))
                    off += og * ig
            _gemm(items, False, False, S)
            offs = [0, sizes[0], sizes[0] + sizes[1]]
            gws = param_colsum(part, [(o, sz, p) for o, sz, p in zip(offs, sizes, ctx.params)])
            gws = [gw.view(p.shape) if gw is not None else None for gw, p in zip(gws, ctx.params)]
        else:
            gws = [torch.zeros_like(p) for p in ctx.params]
        return gh, gws[0], gws[1], gws[2], None


def grouped_linear3(h, wk, wq, wv, heads):
    """(k_lin(h), q_lin(h), v_lin(h)) for grouped 1x1 Conv1d weights [heads * og, ig, 1] -> three [N, heads, og] tensors."""
    ig = wk.shape[1]
    ok = (h.is_cuda and ig % 4 == 0 and all((w.shape[0] // heads) % 4 == 0 and w.shape[1] == ig for w in (wk, wq, wv))
          and 3 * heads <= 12)
    if not ok:
        raise ValueError("grouped_linear3: channel counts per head must be multiples of 4 and heads <= 4 (the own GEMM's "
                         "float4 accesses, 12 problems per launch)")
    return _GroupedLinear3.apply(h, wk, wq, wv, heads)


def _tn_splits(rows, out, cin):
    """Split count of a weight-gradient GEMM dW[out, cin] = g^T x over `rows` rows (tools/lab/tn_split_probe.py): at most
    4,096 rows per split; else enough workgroups to fill the chip (~512 tiles in all) but no more than 64 splits for outputs
    of 128 x 128 and up - every split writes a partial slab - and proportionally more for smaller ones; at least 256 rows
    per split.  (A flat cap of 64 used to sit on top: the gradient of a 20 -> 64 Linear over the 2.9 M kNN edges ran as 64
    workgroups of 45 k rows, 1.6 ms instead of 0.3; a 32 -> 128 one over 186 k rows 101 us instead of 34.)"""
    tiles = -(-out // 128) * -(-cin // 128)
    fill = min(-(-512 // tiles), 64 * max(1, 16384 // max(1, out * cin)))
    return max(1, min(1024, max(-(-rows // 4096), fill), rows // 256))


def _own_linear_ok(x, w, b):
    K = x.shape[-1]
    N = w.shape[0]
    return x.is_cuda and x.dtype == torch.float32 and K % 4 == 0 and N % 4 == 0 and x.numel() > 0


class _LinearOwn(torch.autograd.Function):
    """nn.Linear / 1x1 Conv1d on rows (CP:55-61, 96-117, 161-191) on the library's own f32 MFMA GEMM (k7): y = x W^T + b
    (+ addend, a tensor shaped like y: the sum of two Linears' outputs costs no extra pass) in ONE launch, dX one launch,
    dW one split-reduction launch whose partial slabs - like the bias gradient - are added up by the step's shared
    column-sum launch (ops._GradSink)."""

    @staticmethod
    def forward(ctx, x, w, b, addend):
        ctx.params = (w, b)
        K, N = x.shape[-1], w.shape[0]
        x2 = _rows(x.reshape(-1, K))
        w2 = _rows(w.view(N, K))
        _dev(x2, w2, b)
        M = x2.shape[0]
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        ad = None
        if addend is not None:
            ad = addend.reshape(M, N).contiguous()
            _dev(ad)
        _gemm([dict(a=x2.data_ptr(), lda=x2.stride(0), b=w2.data_ptr(), ldb=w2.stride(0), c=y.data_ptr(), ldc=N,
                    bias=b.data_ptr() if b is not None else None, addend=ad.data_ptr() if ad is not None else None,
                    I=M, J=N, R=K)], True, True)
        ctx.save_for_backward(x2, w2)
        ctx.xshape, ctx.has_bias, ctx.ashape = x.shape, b is not None, (addend.shape if addend is not None else None)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, g):
        x2, w2 = ctx.saved_tensors
        M, K = x2.shape
        N = w2.shape[0]
        g2 = _rows(g.reshape(-1, N))
        wp, bp = ctx.params
        gx = gw = gb = None
        # the weight gradient first: it may fork onto the second stream (_dw_side) and then only waits for what produced g
        if ctx.needs_input_grad[1] and ctx.has_bias and ctx.needs_input_grad[2] and M > 0:
            gw, gb = _tn_grad(g2, x2, wp, bp)                         # the bias gradient rides in the weight gradient's launch
        else:
            if ctx.needs_input_grad[1]:
                gw = _tn_grad(g2, x2, wp)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                gb = param_colsum(g2, [(0, N, bp)])[0]
        if ctx.needs_input_grad[0]:
            gx = torch.empty(M, K, device=g.device, dtype=torch.float32)
            _gemm([dict(a=g2.data_ptr(), lda=g2.stride(0), b=w2.data_ptr(), ldb=w2.stride(0), c=gx.data_ptr(), ldc=K,
                        I=M, J=K, R=N)], True, False)
            gx = gx.view(ctx.xshape)
        ga = g.reshape(ctx.ashape) if ctx.ashape is not None and ctx.needs_input_grad[3] else None
        return gx, gw, gb, ga


class _LinearNN(torch.autograd.Function):
    """y = x W for a parameter W [K, N] used untransposed (the hoisted `weight_k_lin` of the graph attention: (q W), CP:61
    with W(w*k) moved to the query side) - the (1, 0) form of the own GEMM, no transposed copy of W."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.param = w
        K, N = w.shape
        x2 = _rows(x.reshape(-1, K))
        w2 = _rows(w)
        _dev(x2, w2)
        M = x2.shape[0]
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        _gemm([dict(a=x2.data_ptr(), lda=x2.stride(0), b=w2.data_ptr(), ldb=w2.stride(0), c=y.data_ptr(), ldc=N, I=M, J=N, R=K)],
              True, False)
        ctx.save_for_backward(x2, w2)
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, g):
        x2, w2 = ctx.saved_tensors
        M, K = x2.shape
        N = w2.shape[1]
        g2 = _rows(g.reshape(-1, N))
        gx = gw = None
        if ctx.needs_input_grad[0]:                  # dx = g W^T: W [K, N] is the [J][R] operand of the (1, 1) form
            gx = torch.empty(M, K, device=g.device, dtype=torch.float32)
            _gemm([dict(a=g2.data_ptr(), lda=g2.stride(0), b=w2.data_ptr(), ldb=w2.stride(0), c=gx.data_ptr(), ldc=K,
                        I=M, J=K, R=N)], True, True)
            gx = gx.view(ctx.xshape)
        if ctx.needs_input_grad[1]:                  # dW [K, N] = x^T g
            gw = _tn_grad(x2, g2, ctx.param)
        return gx, gw


def linear_nn(x, w):
    if x.is_cuda and w.shape[0] % 4 == 0 and w.shape[1] % 4 == 0 and x.numel() > 0:
        return _LinearNN.apply(x, w)
    return linear(x, w.t())


def _tn_grad(g2, x2, param, bias=None):
    """dW[N, K] = g2^T x2 (reduction over the rows, split over workgroups) as partial slabs -> param_colsum: returns the
    gradient shaped like `param`, or None when it was queued into param.grad.  (Tried and dropped in round 3: queueing the
    products themselves and launching them 12 at a time - one split count and one tile shape for a mixed batch cost 12 ms
    per step at config 3 and gained nothing on the 17-graph shard, tools/lab/ab_bench.py.)"""
    M, N = g2.shape
    K = x2.shape[1]
    S = _tn_splits(M, N, K)
    row = N * K + (N if bias is not None else 0)         # the bias gradient's per-split column sums of g ride behind the slab
    with _dw_side(g2, x2, params=(param, bias)) as side:
        part = torch.empty(S, row, device=g2.device, dtype=torch.float32)
        _gemm([dict(a=g2.data_ptr(), lda=g2.stride(0), b=x2.data_ptr(), ldb=x2.stride(0), c=part.data_ptr(), ldc=K,
                    I=N, J=K, R=M, c_split_stride=row, asum=(part.data_ptr() + 4 * N * K) if bias is not None else None,
                    asum_stride=row)], False, False, S)
        side.keep(part)
    if bias is not None:
        gw, gb = param_colsum(part, [(0, N * K, param), (N * K, N, bias)])
        return (gw.view(param.shape) if gw is not None else None), gb
    gw = param_colsum(part, [(0, N * K, param)])[0]
    return gw.view(param.shape) if gw is not None else None


def _pad4(x, w):
    """Zero-pad the reduction axis of (x [.., K], w [N, K]) to a multiple of 4 floats - the own GEMM reads float4s (the
    3-column property embedding prop_nn, CP:379, is the one such Linear of the model).  Plain differentiable torch ops."""
    K = x.shape[-1]
    if w.dim() == 3:
        w = w.view(w.shape[0], w.shape[1])
    pad = -K % 4
    if pad:
        x = torch.nn.functional.pad(x, (0, pad))
        w = torch.nn.functional.pad(w, (0, pad))
    return x, w


def linear(x, w, b=None):
    """y = x W^T + b (w: [out, in] or a 1x1 Conv1d weight [out, in, 1]) on the own MFMA GEMM (k7).  A reduction that is not a
    multiple of 4 floats is zero-padded, an output width that is not is computed 4-aligned and cut; there is no
    BLAS-library path."""
    if not x.is_cuda:
        raise RuntimeError("singa_amd ops run on the GPU only (no CPU fallback); got a CPU tensor")
    N = w.shape[0]
    if x.numel() == 0:                              # nothing to multiply: zeros that still hang in the autograd graph
        return x.new_zeros(*x.shape[:-1], N) + 0.0 * w.sum() + (0.0 * b.sum() if b is not None else 0.0)
    if _own_linear_ok(x, w, b):
        return _LinearOwn.apply(x, w, b, None)
    x, w = _pad4(x, w)
    padn = -N % 4
    if padn:
        w = torch.nn.functional.pad(w, (0, 0, 0, padn))
        b = torch.nn.functional.pad(b, (0, padn)) if b is not None else None
    y = _LinearOwn.apply(x.contiguous(), w.contiguous(), b, None)
    return y[..., :N] if padn else y


class _LinearMulti(torch.autograd.Function):
    """Several nn.Linear layers applied to the SAME input (W_Q | W_K | W_V of a self-attention, W_K | W_V of a cross
    attention; CP:96-101, 140-143): one launch forward - one problem per layer, each writing its column block of ONE
    [M, sum N] buffer - one launch for dX (a single product over the concatenated output columns: no per-layer input
    gradients to add up) and one split-reduction launch for all weight gradients.  The outputs are views of the fused
    buffer; the attention kernels read them in place and hand back their gradients in the same arrangement."""

    @staticmethod
    def forward(ctx, x, *wb):
        ws, bs = wb[0::2], wb[1::2]
        ctx.params = (ws, bs)
        K = x.shape[-1]
        x2 = _rows(x.reshape(-1, K))
        w2 = [_rows(w.view(w.shape[0], K)) for w in ws]
        _dev(x2, *w2, *bs)
        M = x2.shape[0]
        ns = [w.shape[0] for w in w2]
        tot = sum(ns)
        y = torch.empty(M, tot, device=x.device, dtype=torch.float32)
        items, off = [], 0
        for w, b, n in zip(w2, bs, ns):
            items.append(dict(a=x2.data_ptr(), lda=x2.stride(0), b=w.data_ptr(), ldb=w.stride(0), c=y.data_ptr() + 4 * off,
                              ldc=tot, bias=b.data_ptr() if b is not None else None, I=M, J=n, R=K))
            off += n
        _gemm(items, True, True)
        ctx.save_for_backward(x2, *w2)
        ctx.xshape, ctx.ns = x.shape, ns
        lead = x.shape[:-1]
        outs, off = [], 0
        for n in ns:
            outs.append(y[:, off:off + n].view(*lead, n))
            off += n
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        x2, *w2 = ctx.saved_tensors
        ns = ctx.ns
        tot = sum(ns)
        M, K = x2.shape
        ws, bs = ctx.params
        g2 = [g.reshape(M, n) for g, n in zip(gs, ns)]
        # the gradients normally arrive as the column blocks of one [M, tot] buffer (ops._Attention writes them so)
        fused = all(g.stride(1) == 1 and g.stride(0) == g2[0].stride(0) for g in g2) and g2[0].stride(0) >= tot
        off = 0
        for g, n in zip(g2, ns):
            fused = fused and g.data_ptr() == g2[0].data_ptr() + 4 * off
            off += n
        if fused and g2[0].data_ptr() % 16 == 0 and g2[0].stride(0) % 4 == 0:
            G, ldg = g2[0], g2[0].stride(0)
        else:
            G = torch.cat(g2, 1)
            ldg = tot
        gp = G.data_ptr()
        gx = None
        if ctx.needs_input_grad[0]:
            wcat = torch.cat(w2, 0)                                  # [tot, K]: the reduction runs over all output columns
            gx = torch.empty(M, K, device=x2.device, dtype=torch.float32)
            _gemm([dict(a=gp, lda=ldg, b=wcat.data_ptr(), ldb=K, c=gx.data_ptr(), ldc=K, I=M, J=K, R=tot)], True, False)
            gx = gx.view(ctx.xshape)
        sizes = [n * K for n in ns]
        S = _tn_splits(M, max(ns), K)
        row = tot * K + tot                       # weight-gradient slabs + the bias gradients (column sums of G, same launch)
        part = torch.empty(S, row, device=x2.device, dtype=torch.float32)
        items, off, poff = [], 0, 0
        for n, sz in zip(ns, sizes):
            items.append(dict(a=gp + 4 * off, lda=ldg, b=x2.data_ptr(), ldb=x2.stride(0), c=part.data_ptr() + 4 * poff, ldc=K,
                              I=n, J=K, R=M, c_split_stride=row, asum=part.data_ptr() + 4 * (tot * K + off), asum_stride=row))
            off, poff = off + n, poff + sz
        _gemm(items, False, False, S)
        offs = [sum(sizes[:i]) for i in range(len(sizes))]
        coff = [tot * K + sum(ns[:i]) for i in range(len(ns))]
        res = param_colsum(part, [(o, sz, w) for o, sz, w in zip(offs, sizes, ws)] + [(o, n, b) for o, n, b in zip(coff, ns, bs)])
        gws = [gw.view(w.shape) if gw is not None else None for gw, w in zip(res[:len(ns)], ws)]
        gbs = res[len(ns):]
        out = [gx]
        for gw, gb in zip(gws, gbs):
            out += [gw, gb]
        return tuple(out)


def linear_multi(x, layers):
    """[(weight, bias), ...] applied to the same x -> tuple of outputs (views of one fused buffer).  Falls back to
    separate ops.linear calls for shapes the own GEMM cannot take."""
    if all(_own_linear_ok(x, w, b) and b is not None for w, b in layers) and len(layers) <= 8:
        flat = []
        for w, b in layers:
            flat += [w, b]
        return _LinearMulti.apply(x, *flat)
    return tuple(linear(x, w, b) for w, b in layers)


def linear_add(x, w, b, addend):
    """x W^T + b + addend (addend shaped like the result) in one launch."""
    if _own_linear_ok(x, w, b):
        return _LinearOwn.apply(x, w, b, addend)
    return linear(x, w, b) + addend


class _PosFFN(torch.autograd.Function):
    """PoswiseFeedForward(De)Net up to its residual LayerNorm (CP:161-191): Linear(256 -> 1024) + ReLU + Linear(1024 -> 256)
    as two launches forward (ReLU in the first GEMM's epilogue) and four backward: dh = (g W2) masked by h > 0 in the
    epilogue of its own GEMM, dx = dh W1, and the two split-reduction weight gradients; bias gradients ride in the step's
    shared column-sum launch."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        ctx.params = (w1, b1, w2, b2)
        K, H = x.shape[-1], w1.shape[0]
        x2 = _rows(x.reshape(-1, K))
        a1, a2 = _rows(w1.view(H, K)), _rows(w2.view(w2.shape[0], H))
        _dev(x2, a1, a2, b1, b2)
        M, N = x2.shape[0], a2.shape[0]
        h = torch.empty(M, H, device=x.device, dtype=torch.float32)
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        _gemm([dict(a=x2.data_ptr(), lda=x2.stride(0), b=a1.data_ptr(), ldb=a1.stride(0), c=h.data_ptr(), ldc=H,
                    bias=b1.data_ptr(), I=M, J=H, R=K, relu=1)], True, True)
        _gemm([dict(a=h.data_ptr(), lda=H, b=a2.data_ptr(), ldb=a2.stride(0), c=y.data_ptr(), ldc=N, bias=b2.data_ptr(),
                    I=M, J=N, R=H)], True, True)
        ctx.save_for_backward(x2, a1, a2, h)
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, g):
        x2, a1, a2, h = ctx.saved_tensors
        M, K = x2.shape
        H, N = a1.shape[0], a2.shape[0]
        g2 = _rows(g.reshape(-1, N))
        w1, b1, w2, b2 = ctx.params
        gw2, gb2 = _tn_grad(g2, h, w2, b2)                 # (weight gradients first: they may fork onto the second stream)
        dh = torch.empty(M, H, device=g.device, dtype=torch.float32)
        _gemm([dict(a=g2.data_ptr(), lda=g2.stride(0), b=a2.data_ptr(), ldb=a2.stride(0), c=dh.data_ptr(), ldc=H,
                    I=M, J=H, R=N, mask=h.data_ptr())], True, False)
        gw1, gb1 = _tn_grad(dh, x2, w1, b1)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(M, K, device=g.device, dtype=torch.float32)
            _gemm([dict(a=dh.data_ptr(), lda=H, b=a1.data_ptr(), ldb=a1.stride(0), c=gx.data_ptr(), ldc=K, I=M, J=K, R=H)],
                  True, False)
            gx = gx.view(ctx.xshape)
        return gx, gw1, gb1, gw2, gb2


def pos_ffn(x, w1, b1, w2, b2):
    """relu(x W1^T + b1) W2^T + b2 (the position-wise feed-forward of CP:161-191 before its residual LayerNorm)."""
    return _PosFFN.apply(x, w1, b1, w2, b2)


class _SmallVocabEmbedding(torch.autograd.Function):
    """weight[idx] for small tables (atomic numbers, SMILES tokens).  The backward of nn.Embedding on this ROCm build goes
    through thrust::unique_by_key with its own hipMalloc'ed scratch and a host read-back of the segment count - neither
    survives HIP-graph replay (dangling scratch pointers -> memory-aperture faults).  Here the weight gradient is a
    one-hot GEMM: deterministic, allocation-free apart from PyTorch's own pool, capturable."""

    @staticmethod
    def forward(ctx, weight, idx, padding_idx):
        ctx.save_for_backward(idx)
        ctx.V, ctx.padding_idx = weight.shape[0], padding_idx
        return weight.index_select(0, idx.reshape(-1)).view(*idx.shape, weight.shape[1])

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        onehot = torch.nn.functional.one_hot(idx.reshape(-1), ctx.V).to(g.dtype)
        gw = onehot.t() @ g.reshape(-1, g.shape[-1])
        if ctx.padding_idx is not None:                     # nn.Embedding(padding_idx=i): row i receives no gradient
            gw[ctx.padding_idx] = 0
        return gw, None, None


def embedding(weight, idx, padding_idx=None):
    return _SmallVocabEmbedding.apply(weight, idx, padding_idx)


class _RowDotBias(torch.autograd.Function):
    """scale * (x[..., d] * b[d]).sum(-1) (the hoisted bias term of the graph attention's logits, CP:61-65) as one launch
    each way for 32-channel rows (k15e); the bias gradient's per-workgroup partials ride in the step's shared column-sum
    launch.  Other widths: plain torch ops with the same replay-safe bias gradient."""

    @staticmethod
    def forward(ctx, x, b, scale):
        ctx.param, ctx.scale = b, scale
        D = x.shape[-1]
        ctx.own = bool(x.is_cuda and D == 32 and x.numel() > 0)
        if ctx.own:
            x2, b2 = x.reshape(-1, D).contiguous(), b.contiguous()
            _dev(x2, b2)
            out = torch.empty(x2.shape[0], device=x.device, dtype=torch.float32)
            _chk(_lib.lib().singa_rowdot_fwd(_p(x2), _p(b2), _p(out), x2.shape[0], D, scale, _stream()), "singa_rowdot_fwd")
            ctx.save_for_backward(x2, b2)
            ctx.xshape = x.shape
            return out.view(x.shape[:-1])
        ctx.save_for_backward(x, b)
        return (x * (b * scale)).sum(-1)

    @staticmethod
    def backward(ctx, g):
        x, b = ctx.saved_tensors
        if ctx.own:
            M, D = x.shape
            g = g.reshape(-1).contiguous()
            lib = _lib.lib()
            gx = torch.empty_like(x)
            part = torch.empty(lib.singa_rowdot_nparts(M), D, device=x.device, dtype=torch.float32)
            _chk(lib.singa_rowdot_bwd(_p(g), _p(x), _p(b), _p(gx), _p(part), M, D, ctx.scale, _stream()), "singa_rowdot_bwd")
            return gx.view(ctx.xshape), param_colsum(part, [(0, D, ctx.param)])[0], None
        ge = g.unsqueeze(-1) * ctx.scale
        return ge * b, param_colsum((ge * x).reshape(-1, x.shape[-1]), [(0, x.shape[-1], ctx.param)])[0], None


def rowdot_bias(x, b, scale=1.0):
    return _RowDotBias.apply(x, b, scale)


class _BlockWeight(torch.autograd.Function):
    """[[Wr, -Wi], [Wi, Wr]] of an SO2_m_Convolution's Linear weight w = [Wr; Wi] (EF:677-729 with the recombination of
    EF:721-729 folded in): one launch instead of two slices, a negation and three concatenations, and one launch for its
    gradient - added straight into w.grad when the step engine's gradient sink is on."""

    @staticmethod
    def forward(ctx, w):
        ctx.param = w
        w2 = w.contiguous()
        _dev(w2)
        h, k = w2.shape[0] // 2, w2.shape[1]
        out = torch.empty(2 * h, 2 * k, device=w.device, dtype=torch.float32)
        _chk(_lib.lib().singa_block_weight_fwd(_p(w2), _p(out), h, k, _stream()), "singa_block_weight_fwd")
        ctx.hk = (h, k)
        return out

    @staticmethod
    def backward(ctx, G):
        h, k = ctx.hk
        G = G.contiguous()
        wp = ctx.param
        if _GradSink.takes(wp):
            _chk(_lib.lib().singa_block_weight_bwd(_p(G), _p(wp.grad), h, k, 1, _stream()), "singa_block_weight_bwd")
            return None
        gw = torch.empty(2 * h, k, device=G.device, dtype=torch.float32)
        _chk(_lib.lib().singa_block_weight_bwd(_p(G), _p(gw), h, k, 0, _stream()), "singa_block_weight_bwd")
        return gw


def block_weight(w):
    return _BlockWeight.apply(w)
