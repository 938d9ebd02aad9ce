"""-m gpu: every HIP kernel, called through the C ABI (singa_amd.ops) on the real gfx950 build, against the CPU
oracle on the same seeded inputs.  fp32 tolerance: 1e-4 relative (north_star), tighter where the op is exact."""
import numpy as np
import pytest
import torch

from oracle import singa_oracle as O
from singa_amd import so3

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from singa_amd import ops
    return ops


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rand_edges(rs, n_src, n_dst, E):
    return torch.tensor(np.stack([rs.randint(0, n_src, E), rs.randint(0, n_dst, E)]), dtype=torch.int64)


def rand_rot(rs, E):
    v = torch.tensor(rs.randn(E, 3), dtype=torch.float32)
    return O.edge_rot_mat(v, torch.tensor(rs.rand(E, 3), dtype=torch.float32))


def reduced_rows(w, L, M=2):
    out = []
    for l in range(L + 1):
        mm = min(l, M)
        out.append(w[:, l * l + l - mm: l * l + l + mm + 1, l * l:(l + 1) ** 2].reshape(w.shape[0], -1))
    out = torch.cat(out, 1)
    return torch.nn.functional.pad(out, (0, -out.shape[1] % 4))       # records are padded to 16 bytes (so3_index.h), pad = 0


def rad_row_index(lay):
    idx, off = [np.arange(lay.m_size[0])], lay.m_size[0]
    for s in lay.m_size[1:]:
        idx += [off + np.arange(s), off + np.arange(s)]
        off += s
    return torch.as_tensor(np.concatenate(idx))


def test_library_loaded_and_no_cpu_path():
    ops = _ops()
    from singa_amd import _lib
    assert _lib.lib().singa_version() >= 100
    with pytest.raises(RuntimeError):
        ops.so3_rmsnorm(torch.zeros(2, 9, 16), torch.ones(3, 16), torch.zeros(16), 2)


@pytest.mark.parametrize("name", ["3wi2_4tpp", "4agq_5a7b", "5cp5_4nue"])
def test_edge_frames_match_reference_draws(name):
    """k1 on the GPU through the product's init_edge_rot_mat: the frames the REFERENCE built from its own torch.rand_like
    draws (EF:2286-2351; tests/golden/rot_rand_*.npz + rot_* of embed_L2_*.npz), 1e-6 absolute."""
    from singa_amd.model.EF_layers import init_edge_rot_mat
    from tests.test_oracle_conventions import reference_frame_cases
    for vec, rand, want in reference_frame_cases(name):
        got = init_edge_rot_mat(vec.float().to(DEV), rand=rand.float().to(DEV))
        assert got.shape == want.shape and float((got.cpu() - want).abs().max()) < 1e-6


def test_edge_frame_guards():
    """EF:2292-2297 (short edge: report only) and EF:2329 (aligned helper / NaN: abort) in the product, on the GPU."""
    from singa_amd.model.EF_layers import init_edge_rot_mat
    rs = np.random.RandomState(2)
    vec = torch.tensor(rs.randn(16, 3), dtype=torch.float32, device=DEV)
    rand = torch.tensor(rs.rand(16, 3), dtype=torch.float32, device=DEV)
    v = vec.clone()
    v[3] = torch.tensor([5e-5, 0.0, 0.0], device=DEV)
    with pytest.warns(RuntimeWarning, match="edge_vec_0_distance"):
        init_edge_rot_mat(v, rand=rand)
    v[3] = 0.0
    with pytest.warns(RuntimeWarning), pytest.raises(RuntimeError, match="aligned"):
        init_edge_rot_mat(v, rand=rand)
    assert init_edge_rot_mat(vec[:0], rand=rand[:0]).shape == (0, 3, 3)


def test_edge_frame_statistics_replay_at_bench_size():
    """the guards' statistics must come out of a replayed HIP graph exactly as out of an eager launch at the
    edge counts of the bench (120 k - 500 k edges: many workgroups).  They are integer min / max atomics on float bit
    patterns, so the values are order-independent and exact."""
    ops = _ops()
    for E in (120_001, 500_000):
        g = torch.Generator(device="cpu").manual_seed(E)
        vec = (torch.randn(E, 3, generator=g) * 2.0).to(DEV)
        rand = torch.rand(E, 3, generator=g).to(DEV)
        eager = torch.tensor([float("inf"), 0.0], device=DEV)
        rot_e = ops.edge_frames(vec, rand, eager)
        want_min = float(vec.norm(dim=1).min())
        assert abs(float(eager[0]) - want_min) <= 1e-6 * want_min and 0.0 < float(eager[1]) < 0.99
        stat = torch.tensor([float("inf"), 0.0], device=DEV)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.edge_frames(vec, rand, stat)
        torch.cuda.current_stream().wait_stream(side)
        stat.copy_(torch.tensor([float("inf"), 0.0]))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            rot_g = ops.edge_frames(vec, rand, stat)
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(stat, eager) and torch.equal(rot_g, rot_e)


@pytest.mark.parametrize("L", [2, 4, 6])
def test_wigner_rows(L):
    ops = _ops()
    rs = np.random.RandomState(L)
    rot = rand_rot(rs, 1000)
    wr = ops.wigner_rows(rot.to(DEV), L)
    ref = reduced_rows(O.wigner_dense(rot, L), L)
    assert wr.shape == ref.shape and wr.shape[1] % 4 == 0     # records padded to a multiple of 4 floats, pad = 0
    assert float((wr.cpu() - ref).abs().max()) < 3e-5
    # orthogonality of full blocks l <= 2: rows of D_l are orthonormal
    blk = wr[:, 1:10].view(-1, 3, 3)
    assert float((blk @ blk.transpose(1, 2) - torch.eye(3, device=DEV)).abs().max()) < 1e-5


@pytest.mark.parametrize("L", [2, 4, 6])
@pytest.mark.parametrize("homo", [True, False])
def test_gather_rotate_fwd_bwd(L, homo):
    ops = _ops()
    rs = np.random.RandomState(100 + L)
    C, Ns, Nd, E = 16, (60 if not homo else 80), 80, 700
    lay = so3.layout(L, 2)
    ei = rand_edges(rs, Ns, Nd, E)
    rot = rand_rot(rs, E)
    xs = torch.tensor(rs.randn(Ns, lay.K, C), dtype=torch.float32, requires_grad=True)
    xd = xs if homo else torch.tensor(rs.randn(Nd, lay.K, C), dtype=torch.float32, requires_grad=True)
    rad = torch.tensor(rs.randn(E, lay.rad_rows * 2 * C), dtype=torch.float32, requires_grad=True)
    g = torch.tensor(rs.randn(E, lay.KR * 2 * C), dtype=torch.float32)
    # oracle (original edge order)
    fr = O.Frame(rot, L, 2)
    ref = (torch.bmm(fr.fwd, torch.cat([xs[ei[0]], xd[ei[1]]], 2))[:, fr.to_m]
           * rad.view(E, lay.rad_rows, 2 * C)[:, rad_row_index(lay)]).reshape(E, -1)
    ref.backward(g)
    # HIP path (destination-sorted order)
    es = ops.EdgeSet(ei.to(DEV), Ns, Nd)
    order = es.order.cpu()
    wr = ops.wigner_rows(rot[order].to(DEV), L)
    xs_g = xs.detach().to(DEV).requires_grad_(True)
    xd_g = xs_g if homo else xd.detach().to(DEV).requires_grad_(True)
    rad_g = rad.detach()[order].to(DEV).requires_grad_(True)
    out = ops.gather_rotate(xs_g, xd_g, rad_g, wr, es, L)
    assert rel(out, ref[order]) < 2e-5
    out.backward(g[order].to(DEV))
    assert rel(xs_g.grad, xs.grad) < 2e-5
    if not homo:
        assert rel(xd_g.grad, xd.grad) < 2e-5
    assert rel(rad_g.grad, rad.grad[order]) < 2e-5


@pytest.mark.parametrize("L", [2, 4, 6])
def test_rotate_back_scatter_fwd_bwd(L):
    ops = _ops()
    rs = np.random.RandomState(200 + L)
    CH, heads, Nd, E = 112, 7, 90, 800
    lay = so3.layout(L, 2)
    ei = rand_edges(rs, 50, Nd, E)
    ei[1, :40] = 3                                    # one heavy destination; some nodes stay empty
    rot = rand_rot(rs, E)
    fr = O.Frame(rot, L, 2)
    parts = [torch.tensor(rs.randn(E, r * CH), dtype=torch.float32, requires_grad=True) for r in lay.seg_rows]
    alpha = torch.tensor(rs.rand(E, heads), dtype=torch.float32, requires_grad=True)
    msg_l = torch.cat([p.view(E, -1, CH) for p in parts], 1)[:, fr.to_l]
    msg_l = (msg_l.view(E, lay.KR, heads, CH // heads) * alpha.view(E, 1, heads, 1)).reshape(E, lay.KR, CH)
    ref = O.seg_sum(torch.bmm(fr.inv, msg_l), ei[1], Nd)
    g = torch.tensor(rs.randn(Nd, lay.K, CH), dtype=torch.float32)
    ref.backward(g)
    es = ops.EdgeSet(ei.to(DEV), 50, Nd)
    order = es.order.cpu()
    wr = ops.wigner_rows(rot[order].to(DEV), L)
    ys = [p.detach()[order].to(DEV).requires_grad_(True) for p in parts]
    al = alpha.detach()[order].to(DEV).requires_grad_(True)
    out = ops.rotate_back_scatter(ys[0], ys[1], ys[2], al, wr, es, heads, L)
    assert rel(out, ref) < 2e-5
    out.backward(g.to(DEV))
    for y, p in zip(ys, parts):
        assert rel(y.grad, p.grad[order]) < 2e-5
    assert rel(al.grad, alpha.grad[order]) < 5e-5


@pytest.mark.parametrize("L", [2, 6])
def test_edge_degree_scatter(L):
    ops = _ops()
    rs = np.random.RandomState(300 + L)
    C, Nd, E = 16, 70, 500
    lay = so3.layout(L, 2)
    ei = rand_edges(rs, 40, Nd, E)
    rot = rand_rot(rs, E)
    fr = O.Frame(rot, L, 2)
    r = torch.tensor(rs.randn(E, lay.m_size[0] * C), dtype=torch.float32, requires_grad=True)
    full = torch.cat([r.view(E, -1, C), torch.zeros(E, lay.KR - lay.m_size[0], C)], 1)[:, fr.to_l]
    ref = O.seg_sum(torch.bmm(fr.inv, full), ei[1], Nd) / O.AVG_DEGREE
    g = torch.tensor(rs.randn(Nd, lay.K, C), dtype=torch.float32)
    ref.backward(g)
    es = ops.EdgeSet(ei.to(DEV), 40, Nd)
    order = es.order.cpu()
    wr = ops.wigner_rows(rot[order].to(DEV), L)
    rg = r.detach()[order].to(DEV).requires_grad_(True)
    out = ops.edge_degree_scatter(rg, wr, es, L, 2, 1.0 / O.AVG_DEGREE)
    assert rel(out, ref) < 2e-5
    out.backward(g.to(DEV))
    assert rel(rg.grad, r.grad[order]) < 2e-5


@pytest.mark.parametrize("H,eps,N,E", [(7, 1e-16, 300, 5000), (4, 0.0, 300, 5000), (4, 0.0, 20, 5000), (4, 1e-16, 300, 900)],
                         ids=["7heads", "4heads-dense", "4heads-long-segments", "4heads-sparse-with-empty-segments"])
def test_segment_softmax(H, eps, N, E):
    ops = _ops()
    rs = np.random.RandomState(5)
    dst = torch.tensor(np.sort(rs.randint(0, N, E)), dtype=torch.int64)
    rp = torch.zeros(N + 1, dtype=torch.int64)
    rp[1:] = torch.bincount(dst, minlength=N).cumsum(0)
    x = torch.tensor(rs.randn(E, H) * 4, dtype=torch.float32, requires_grad=True)
    ref = O.seg_softmax(x, dst, N, eps)
    g = torch.tensor(rs.randn(E, H), dtype=torch.float32)
    ref.backward(g)
    xg = x.detach().to(DEV).requires_grad_(True)
    y = ops.segment_softmax(xg, rp.to(torch.int32).to(DEV), eps)
    assert float((y.cpu() - ref.detach()).abs().max()) < 1e-6
    y.backward(g.to(DEV))
    assert rel(xg.grad, x.grad) < 1e-5
    sums = torch.zeros(N, H).index_add_(0, dst, y.detach().cpu())
    assert float((sums[torch.bincount(dst, minlength=N) > 0] - 1).abs().max()) < 1e-5


def test_segment_wsum():
    ops = _ops()
    rs = np.random.RandomState(6)
    N, E, H, F = 200, 6000, 4, 64
    dst = torch.tensor(np.sort(rs.randint(0, N, E)), dtype=torch.int64)
    rp = torch.zeros(N + 1, dtype=torch.int64)
    rp[1:] = torch.bincount(dst, minlength=N).cumsum(0)
    w = torch.tensor(rs.rand(E, H), dtype=torch.float32, requires_grad=True)
    v = torch.tensor(rs.randn(E, H, F), dtype=torch.float32, requires_grad=True)
    ref = O.seg_sum(w.unsqueeze(-1) * v, dst, N)
    g = torch.tensor(rs.randn(N, H, F), dtype=torch.float32)
    ref.backward(g)
    wg, vg = w.detach().to(DEV).requires_grad_(True), v.detach().to(DEV).requires_grad_(True)
    out = ops.segment_wsum(wg, vg, rp.to(torch.int32).to(DEV))
    assert rel(out, ref) < 1e-5
    out.backward(g.to(DEV))
    assert rel(wg.grad, w.grad) < 1e-5 and rel(vg.grad, v.grad) < 1e-6


@pytest.mark.parametrize("L", [2, 4, 6])
def test_s2act_edge_and_node(L):
    ops = _ops()
    rs = np.random.RandomState(400 + L)
    lay = so3.layout(L, 2)
    E, C, extra = 300, 128, 224
    h0 = torch.tensor(rs.randn(E, extra + C + lay.seg_rows[0] * C), dtype=torch.float32, requires_grad=True)
    h1 = torch.tensor(rs.randn(E, lay.seg_rows[1] * C), dtype=torch.float32, requires_grad=True)
    h2 = torch.tensor(rs.randn(E, lay.seg_rows[2] * C), dtype=torch.float32, requires_grad=True)
    xm = torch.cat([h0[:, extra + C:].view(E, -1, C), h1.view(E, -1, C), h2.view(E, -1, C)], 1)
    to_m = torch.as_tensor(lay.to_m)
    ref = O.sep_s2_act(h0[:, extra:extra + C], xm[:, torch.argsort(to_m)], L, 2)[:, to_m].reshape(E, -1)
    g = torch.tensor(rs.randn(*ref.shape), dtype=torch.float32)
    ref.backward(g)
    hs = [h.detach().to(DEV).requires_grad_(True) for h in (h0, h1, h2)]
    out = ops.s2act_edge(hs[0], hs[1], hs[2], extra, extra + C, C, L)
    assert rel(out, ref) < 2e-5
    out.backward(g.to(DEV))
    for a, b in zip(hs, (h0, h1, h2)):
        assert rel(a.grad, b.grad) < 5e-5
    # node / FFN flavour on the [L][L] grid
    N, Cn = 64, 512
    K = (L + 1) ** 2
    x = torch.tensor(rs.randn(N, K, Cn), dtype=torch.float32, requires_grad=True)
    gt = torch.tensor(rs.randn(N, Cn), dtype=torch.float32, requires_grad=True)
    ref = O.sep_s2_act(gt, x, L, L)
    g = torch.tensor(rs.randn(N, K, Cn), dtype=torch.float32)
    ref.backward(g)
    xg, gg = x.detach().to(DEV).requires_grad_(True), gt.detach().to(DEV).requires_grad_(True)
    out = ops.s2act_node(xg, gg, L)
    assert rel(out, ref) < 2e-5
    out.backward(g.to(DEV))
    assert rel(xg.grad, x.grad) < 5e-5 and rel(gg.grad, gt.grad) < 1e-5


@pytest.mark.parametrize("L", [2, 4, 6])
def test_ffn_tail_matches_oracle(L):
    """ops.ffn_tail = SeparableS2Activation on the [L][L] grid followed by SO3_LinearV2(512 -> 16) (+ residual), the back half
    of the feed-forward block (reference model/EF_layers.py:256-262, 655-671, 1405-1406), against the oracle: output and ALL
    gradients (hidden tensor, gate, weight, bias, residual).  Its backward forms the activation's output gradient inside the
    activation's backward kernel (singa_s2act_ffn_bwd) - also checked against the two-launch form of the same product."""
    ops = _ops()
    rs = np.random.RandomState(450 + L)
    N, C, K = 67, 512, (L + 1) ** 2
    h = torch.tensor(rs.randn(N, K, C), dtype=torch.float32, requires_grad=True)
    gt = torch.tensor(rs.randn(N, C), dtype=torch.float32, requires_grad=True)
    w = torch.tensor(rs.randn(L + 1, 16, C) / np.sqrt(C), dtype=torch.float32, requires_grad=True)
    b = torch.tensor(0.1 * rs.randn(16), dtype=torch.float32, requires_grad=True)
    res = torch.tensor(rs.randn(N, K, 16), dtype=torch.float32, requires_grad=True)
    sd = {"l.weight": w, "l.bias": b}
    ref = O.so3_linear(sd, "l", O.sep_s2_act(gt, h, L, L), L) + res
    g = torch.tensor(rs.randn(N, K, 16), dtype=torch.float32)
    ref.backward(g)
    dev = [t.detach().to(DEV).requires_grad_(True) for t in (h, gt, w, b, res)]
    out = ops.ffn_tail(dev[0], dev[1], dev[2], dev[3], L, dev[4])
    assert rel(out, ref) < 2e-5
    out.backward(g.to(DEV))
    for a, r, tol in zip(dev, (h, gt, w, b, res), (5e-5, 2e-5, 2e-5, 1e-5, 1e-6)):
        assert rel(a.grad, r.grad) < tol
    # the two-launch form (activation node + linear node) of the same product
    dev2 = [t.detach().to(DEV).requires_grad_(True) for t in (h, gt, w, b, res)]
    out2 = ops.so3_linear(ops.s2act_node(dev2[0], dev2[1], L), dev2[2], dev2[3], L, dev2[4])
    out2.backward(g.to(DEV))
    assert rel(out, out2) < 1e-6
    for a, c in zip(dev, dev2):
        assert rel(a.grad, c.grad) < 2e-6


@pytest.mark.parametrize("L", [2, 4, 6])
def test_so3_rmsnorm(L):
    ops = _ops()
    rs = np.random.RandomState(500 + L)
    N, C, K = 777, 16, (L + 1) ** 2
    x = torch.tensor(rs.randn(N, K, C) * 2 + 0.3, dtype=torch.float32, requires_grad=True)
    w = torch.tensor(1 + 0.1 * rs.randn(L + 1, C), dtype=torch.float32, requires_grad=True)
    b = torch.tensor(0.1 * rs.randn(C), dtype=torch.float32, requires_grad=True)
    sd = {"n.affine_weight": w, "n.affine_bias": b}
    ref = O.rms_norm(sd, "n", x, L)
    g = torch.tensor(rs.randn(N, K, C), dtype=torch.float32)
    ref.backward(g)
    xg, wg, bg = (t.detach().to(DEV).requires_grad_(True) for t in (x, w, b))
    y = ops.so3_rmsnorm(xg, wg, bg, L)
    assert rel(y, ref) < 1e-5
    y.backward(g.to(DEV))
    assert rel(xg.grad, x.grad) < 2e-5
    # the norm with the residual branch that leaves its input as ONE autograd node (x + f(norm(x)) of a TransBlockV2): the
    # backward kernel adds the skip's gradient itself
    g2 = torch.tensor(rs.randn(N, K, C), dtype=torch.float32)
    xs, ws, bs = (t.detach().to(DEV).requires_grad_(True) for t in (x, w, b))
    y2, skip = ops.so3_rmsnorm_skip(xs, ws, bs, L)
    assert torch.equal(y2, y) and torch.equal(skip, xs)
    torch.autograd.backward([y2, skip], [g.to(DEV), g2.to(DEV)])
    assert rel(xs.grad, x.grad + g2) < 2e-5 and rel(ws.grad, wg.grad) < 1e-6 and rel(bs.grad, bg.grad) < 1e-6
    xs.grad = None
    y3, skip3 = ops.so3_rmsnorm_skip(xs, ws, bs, L)
    skip3.backward(g2.to(DEV))                       # only the skip used
    assert torch.equal(xs.grad, g2.to(DEV))
    assert rel(wg.grad, w.grad) < 2e-5 and rel(bg.grad, b.grad) < 2e-5
    # identity property: affine = identity -> balanced RMS of the output is 1
    y1 = ops.so3_rmsnorm(xg.detach(), torch.ones_like(wg), torch.zeros_like(bg), L).cpu()
    lk = torch.as_tensor(so3.layout(L, L).degree)
    bal = (1.0 / ((2 * lk + 1).float() * (L + 1))).view(1, -1, 1)
    assert float((((y1 ** 2) * bal).sum(1).mean(1) - 1).abs().max()) < 1e-3


def test_fused_graph_attention_ops():
    """edge_logits / gather_wsum (fwd + bwd) against the un-fused reference formulation of CP:59-74."""
    ops = _ops()
    rs = np.random.RandomState(9)
    N, E, H, D, F_ = 150, 4000, 4, 32, 64
    row = torch.tensor(np.sort(rs.randint(0, N, E)), dtype=torch.int64)
    col = torch.tensor(rs.randint(0, N, E), dtype=torch.int64)

    class Edges:
        pass
    e = Edges()
    rp = torch.zeros(N + 1, dtype=torch.int64)
    rp[1:] = torch.bincount(row, minlength=N).cumsum(0)
    cp = torch.zeros(N + 1, dtype=torch.int64)
    cp[1:] = torch.bincount(col, minlength=N).cumsum(0)
    e.row_ptr, e.col_ptr = rp.to(torch.int32).to(DEV), cp.to(torch.int32).to(DEV)
    e.row32, e.col32 = row.to(torch.int32).to(DEV), col.to(torch.int32).to(DEV)
    e.eperm = torch.argsort(col, stable=True).to(torch.int32).to(DEV)
    t = lambda *s: torch.tensor(rs.randn(*s), dtype=torch.float32, requires_grad=True)
    q, hk, wk, wkl, bkl = t(N, H, D), t(N, H, D), t(E, D), t(D, D), t(D)
    keys = F_lin(wk.unsqueeze(1) * hk[col], wkl, bkl)
    ref = (q[row] * keys).sum(-1) / np.sqrt(D)
    g = torch.tensor(rs.randn(E, H), dtype=torch.float32)
    ref.backward(g)
    qg, hkg, wkg, wklg, bklg = (x.detach().to(DEV).requires_grad_(True) for x in (q, hk, wk, wkl, bkl))
    scale = 1.0 / np.sqrt(D)
    out = ops.edge_logits(torch.matmul(qg, wklg), wkg, hkg, (qg * bklg).sum(-1) * scale, e, scale)
    assert rel(out, ref) < 1e-5
    out.backward(g.to(DEV))
    for a, b in ((qg, q), (hkg, hk), (wkg, wk), (wklg, wkl), (bklg, bkl)):
        assert rel(a.grad, b.grad) < 2e-5
    # value path
    alpha, hv, wv, wvl, bvl = torch.tensor(rs.rand(E, H), dtype=torch.float32, requires_grad=True), t(N, H, F_), t(E, F_), t(F_, F_), t(F_)
    msg = alpha.unsqueeze(-1) * F_lin(wv.unsqueeze(1) * hv[col], wvl, torch.zeros(F_))
    ref = torch.zeros(N, H, F_).index_add_(0, row, msg)
    g = torch.tensor(rs.randn(N, H, F_), dtype=torch.float32)
    ref.backward(g)
    ag, hvg, wvg, wvlg = (x.detach().to(DEV).requires_grad_(True) for x in (alpha, hv, wv, wvl))
    out = F_lin(ops.gather_wsum(ag, wvg, hvg, e), wvlg, None)
    assert rel(out, ref) < 1e-5
    out.backward(g.to(DEV))
    for a, b in ((ag, alpha), (hvg, hv), (wvg, wv), (wvlg, wvl)):
        assert rel(a.grad, b.grad) < 2e-5


def F_lin(x, w, b):
    return torch.nn.functional.linear(x, w, b)


def test_linear_pads_odd_shapes():
    """ops.linear on shapes the own GEMM's float4 accesses cannot take directly - a 3-long reduction (the property embedding
    prop_nn, CP:379), an output width of 6, an empty input - against torch: zero-padded / cut, no BLAS-library path."""
    ops = _ops()
    rs = np.random.RandomState(10)
    for M, K, N in ((40000, 64, 32), (128, 3, 256), (77, 10, 6), (0, 3, 8)):
        x = torch.tensor(rs.randn(M, K), dtype=torch.float32, device=DEV, requires_grad=True)
        w = torch.tensor(rs.randn(N, K), dtype=torch.float32, device=DEV, requires_grad=True)
        b = torch.tensor(rs.randn(N), dtype=torch.float32, device=DEV, requires_grad=True)
        g = torch.tensor(rs.randn(M, N), dtype=torch.float32, device=DEV)
        ref = F_lin(x, w, b)
        ref.backward(g)
        gx, gw, gb = x.grad.clone(), w.grad.clone(), b.grad.clone()
        x.grad = w.grad = b.grad = None
        out = ops.linear(x, w, b)
        out.backward(g)
        assert out.shape == ref.shape
        if M == 0:
            assert float(w.grad.abs().max()) == 0.0 and float(b.grad.abs().max()) == 0.0
            continue
        assert rel(out, ref) < 1e-6 and rel(x.grad, gx) < 1e-6 and rel(w.grad, gw) < 1e-5 and rel(b.grad, gb) < 1e-5


def test_edge_head_logits_and_activation():
    """k9a (LayerNorm -> SmoothLeakyReLU -> dot) + k8 in one node vs the torch formulation, incl. parameter grads."""
    ops = _ops()
    rs = np.random.RandomState(11)
    L, heads, A, C = 2, 7, 32, 128
    lay = so3.layout(L, 2)
    E = 3000
    h0 = torch.tensor(rs.randn(E, heads * A + C + lay.seg_rows[0] * C), dtype=torch.float32, requires_grad=True)
    h1 = torch.tensor(rs.randn(E, lay.seg_rows[1] * C), dtype=torch.float32, requires_grad=True)
    h2 = torch.tensor(rs.randn(E, lay.seg_rows[2] * C), dtype=torch.float32, requires_grad=True)
    w = torch.tensor(1 + 0.2 * rs.randn(A), dtype=torch.float32, requires_grad=True)
    b = torch.tensor(0.2 * rs.randn(A), dtype=torch.float32, requires_grad=True)
    dot = torch.tensor(rs.randn(heads, A) * 0.2, dtype=torch.float32, requires_grad=True)
    a = torch.nn.functional.layer_norm(h0[:, :heads * A].reshape(-1, heads, A), (A,), w, b, 1e-5)
    a = 0.6 * a + 0.4 * a * (2 * torch.sigmoid(a) - 1)
    ref_logits = (a * dot).sum(-1)
    to_m = torch.as_tensor(lay.to_m)
    xm = torch.cat([h0[:, heads * A + C:].view(E, -1, C), h1.view(E, -1, C), h2.view(E, -1, C)], 1)
    ref_act = O.sep_s2_act(h0[:, heads * A:heads * A + C], xm[:, torch.argsort(to_m)], L, 2)[:, to_m].reshape(E, -1)
    g1 = torch.tensor(rs.randn(E, heads), dtype=torch.float32)
    g2 = torch.tensor(rs.randn(*ref_act.shape), dtype=torch.float32)
    ((ref_logits * g1).sum() + (ref_act * g2).sum()).backward()
    dev = [t.detach().to(DEV).requires_grad_(True) for t in (h0, h1, h2, w, b, dot)]
    logits, act = ops.edge_head(dev[0], dev[1], dev[2], dev[3], dev[4], dev[5], heads, A, C, L)
    assert rel(logits, ref_logits) < 2e-5 and rel(act, ref_act) < 2e-5
    ((logits * g1.to(DEV)).sum() + (act * g2.to(DEV)).sum()).backward()
    for d, r in zip(dev, (h0, h1, h2, w, b, dot)):
        assert rel(d.grad, r.grad) < 1e-4



def cpu_f64(fn, inputs, grad_outs, wrt=None):
    """Evaluate `fn` on float64 CPU copies of `inputs` (the CPU reference of a kernel: same formula as the reference's
    torch code, computed away from the GPU) and return (outputs, gradients w.r.t. inputs[wrt]) as float32 CUDA tensors."""
    ins = [t.detach().cpu().double().requires_grad_(True) if (torch.is_tensor(t) and t.is_floating_point()) else
           (t.cpu() if torch.is_tensor(t) else t) for t in inputs]
    outs = fn(*ins)
    single = torch.is_tensor(outs)
    outs_l = [outs] if single else list(outs)
    gos = [g.detach().cpu().double() for g in ([grad_outs] if torch.is_tensor(grad_outs) else grad_outs)]
    idx = range(len(ins)) if wrt is None else wrt
    grads = torch.autograd.grad(outs_l, [ins[i] for i in idx], gos)
    to = lambda t: t.detach().float().to(DEV)
    return (to(outs) if single else [to(o) for o in outs_l]), [to(g) for g in grads]


@pytest.mark.parametrize("M", [1, 63, 56448])
def test_ln_silu_matches_torch(M):
    """k6a against LayerNorm + SiLU evaluated in float64 on the CPU (outputs, input gradient, d gamma, d beta)."""
    from singa_amd import ops
    torch.manual_seed(M)
    x = (torch.randn(M, 16, device="cuda") * 2 + 0.3).requires_grad_(True)
    gamma = torch.randn(16, device="cuda", requires_grad=True)
    beta = torch.randn(16, device="cuda", requires_grad=True)
    g = torch.randn(M, 16, device="cuda")
    want, want_g = cpu_f64(lambda x_, ga, be: torch.nn.functional.silu(torch.nn.functional.layer_norm(x_, (16,), ga, be, 1e-5)),
                           (x, gamma, beta), g)
    got = ops.ln_silu(x, gamma, beta, 1e-5)
    got_g = torch.autograd.grad(got, (x, gamma, beta), g)
    assert float((got - want).abs().max()) < 1e-5 * max(1.0, float(want.abs().max()))
    for a, b in zip(got_g, want_g):
        assert float((a - b).abs().max()) < 1e-4 * max(1e-3, float(b.abs().max())), a.shape


def test_bias_ssp_matches_torch():
    """k15d: softplus(u + b) - ln 2 and its gradients against torch (including the x > 20 branch)."""
    import math
    from singa_amd import ops
    torch.manual_seed(2)
    u = (torch.randn(5003, 64, device="cuda") * 6).requires_grad_(True)
    b = torch.randn(64, device="cuda", requires_grad=True)
    g = torch.randn(5003, 64, device="cuda")
    want, want_g = cpu_f64(lambda u_, b_: torch.nn.functional.softplus(u_ + b_) - math.log(2.0), (u, b), g)
    got = ops.bias_ssp(u, b)
    got_g = torch.autograd.grad(got, (u, b), g)
    assert float((got - want).abs().max()) < 1e-5 * float(want.abs().max())
    for a, c in zip(got_g, want_g):
        assert float((a - c).abs().max()) < 1e-4 * float(c.abs().max())


@pytest.mark.parametrize("M,res", [(1, True), (6432, True), (6400, False), (5, False)])
def test_layer_norm_256_residual_matches_torch(M, res):
    """k16 against torch.nn.LayerNorm on a + r: output and gradients w.r.t. a, r, gamma, beta."""
    from singa_amd import ops
    torch.manual_seed(M)
    ln = torch.nn.LayerNorm(256, device="cuda")
    with torch.no_grad():
        ln.weight.copy_(torch.randn(256)), ln.bias.copy_(torch.randn(256))
    a = (torch.randn(M, 256, device="cuda") * 3).requires_grad_(True)
    r = torch.randn(M, 256, device="cuda", requires_grad=True) if res else None
    g = torch.randn(M, 256, device="cuda")
    ins = (a, r, ln.weight, ln.bias) if res else (a, ln.weight, ln.bias)
    if res:
        want, want_g = cpu_f64(lambda a_, r_, w_, b_: torch.nn.functional.layer_norm(a_ + r_, (256,), w_, b_, ln.eps), ins, g)
    else:
        want, want_g = cpu_f64(lambda a_, w_, b_: torch.nn.functional.layer_norm(a_, (256,), w_, b_, ln.eps), ins, g)
    got = ops.layer_norm_residual(a, r, ln)
    got_g = torch.autograd.grad(got, ins, g)
    assert float((got - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))
    for x, y in zip(got_g, want_g):
        assert float((x - y).abs().max()) < 1e-4 * max(1e-3, float(y.abs().max())), x.shape


@pytest.mark.parametrize("E", [1, 31, 1000, 40_007])
def test_edge_mlp_pair_matches_module_path(E):
    """k15c (both per-edge MLPs on the MFMA, recomputing backward) against Linear -> softplus - ln2 -> Linear evaluated
    with torch autograd: outputs and all eight parameter gradients, including partial 32-edge tiles."""
    import math
    from singa_amd import ops
    torch.manual_seed(E)
    attr = torch.randn(E, 64, device="cuda")
    nets = [(torch.nn.Linear(64, H, device="cuda"), torch.nn.Linear(H, H, device="cuda")) for H in (32, 64)]
    gk, gv = torch.randn(E, 32, device="cuda"), torch.randn(E, 64, device="cuda")
    params = [p for l1, l2 in nets for p in (l1.weight, l1.bias, l2.weight, l2.bias)]
    F = torch.nn.functional

    def ref(a, w1k, b1k, w2k, b2k, w1v, b1v, w2v, b2v):       # CP:41-48: Linear -> softplus - ln 2 -> Linear, both nets
        return [F.linear(F.softplus(F.linear(a, w1k, b1k)) - math.log(2.0), w2k, b2k),
                F.linear(F.softplus(F.linear(a, w1v, b1v)) - math.log(2.0), w2v, b2v)]
    want, want_g = cpu_f64(ref, [attr] + params, [gk, gv], wrt=range(1, 9))
    got = ops.edge_mlp_pair(attr, nets[0], nets[1])
    got_g = torch.autograd.grad(list(got), params, [gk, gv])
    for a, b in zip(got, want):
        assert float((a - b).abs().max()) < 2e-6 * max(1.0, float(b.abs().max()))
    for a, b in zip(got_g, want_g):
        assert float((a - b).norm()) < 2e-5 * max(1e-6, float(b.norm())), a.shape


@pytest.mark.parametrize("S,expanded", [(201, False), (230, True), (700, True)])
def test_masked_softmax_matches_torch(S, expanded):
    """k18 against divide + masked_fill(-1e9) + softmax with torch autograd, for a full [B,T,S] mask, an expanded padding
    mask (stride 0 over the query axis), and rows longer than four wavefront passes."""
    from singa_amd import ops
    torch.manual_seed(S)
    B, heads, T = 3, 4, 37
    s = (torch.randn(B * heads, T, S, device="cuda") * 5).requires_grad_(True)
    if expanded:
        mask = (torch.rand(B, 1, S, device="cuda") < 0.3).expand(B, T, S)
    else:
        mask = (torch.rand(B, T, S, device="cuda") < 0.3) | torch.triu(torch.ones(T, S, dtype=torch.bool, device="cuda"), 1)
        mask[:, :, 0] = False
    g = torch.randn(B * heads, T, S, device="cuda")
    scale = 1.0 / 32 ** 0.5
    mask_c = mask.cpu()
    ref, (ref_g,) = cpu_f64(lambda s_: torch.softmax((s_.view(B, heads, T, S) * scale).masked_fill(mask_c.unsqueeze(1), -1e9), -1)
                            .view(B * heads, T, S), (s,), g)
    got = ops.masked_softmax(s, mask, scale, heads)
    got_g, = torch.autograd.grad(got, s, g)
    assert float((got - ref).abs().max()) < 1e-6
    assert float((got_g - ref_g).abs().max()) < 1e-5 * max(1e-3, float(ref_g.abs().max()))


@pytest.mark.parametrize("T,S,kind", [(201, 201, "causal"), (201, 230, "padding"), (37, 700, "padding"), (5, 3, "none"), (64, 33, "row-masked")])
def test_attention_matches_torch(T, S, kind):
    """k19 (flash-style attention on the MFMA, recomputing backward) against matmul / masked_fill(-1e9) / softmax / matmul
    with torch autograd: context and the gradients of q, k, v; partial tiles, an expanded padding mask, a fully masked row."""
    import math
    from singa_amd import ops
    torch.manual_seed(T * 1000 + S)
    B, heads = 3, 4
    q, k, v = (torch.randn(B * heads, n, d, device="cuda", requires_grad=True) for n, d in ((T, 32), (S, 32), (S, 64)))
    if kind == "causal":
        mask = torch.triu(torch.ones(T, S, dtype=torch.bool, device="cuda"), 1).unsqueeze(0) | (torch.rand(B, 1, S, device="cuda") < 0.1)
        mask[:, :, 0] = False
    elif kind == "padding":
        mask = (torch.rand(B, 1, S, device="cuda") < 0.3).expand(B, T, S)
    elif kind == "row-masked":
        mask = torch.rand(B, T, S, device="cuda") < 0.2
        mask[1, 7] = True                                  # every key masked: the reference gives a uniform row
    else:
        mask = torch.zeros(B, 1, S, dtype=torch.bool, device="cuda")
    g = torch.randn(B * heads, T, 64, device="cuda")
    scale = 1.0 / math.sqrt(32)
    mask_c = mask.cpu()

    def ref(q_, k_, v_):                                   # CP:107-117 / 136-148
        sc = (torch.bmm(q_, k_.transpose(1, 2)) * scale).view(B, heads, T, S).masked_fill(mask_c.unsqueeze(1), -1e9)
        return torch.bmm(torch.softmax(sc, -1).view(B * heads, T, S), v_)
    want, want_g = cpu_f64(ref, (q, k, v), g)
    got = ops.attention(q, k, v, mask, scale, heads)
    got_g = torch.autograd.grad(got, (q, k, v), g)
    assert float((got - want).abs().max()) < 1e-5 * max(1.0, float(want.abs().max()))
    for a, b in zip(got_g, want_g):
        assert float((a - b).norm()) < 2e-5 * float(b.norm())
    # the same through the token-major layout [B, T|S, heads, D] the model uses (no head transposes): identical numbers
    tm = lambda t: t.detach().view(B, heads, t.shape[1], t.shape[2]).transpose(1, 2).contiguous().requires_grad_(True)
    q2, k2, v2 = tm(q), tm(k), tm(v)
    got2 = ops.attention(q2, k2, v2, mask, scale, heads, token_major=True)
    assert got2.shape == (B, T, heads, 64)
    got2_g = torch.autograd.grad(got2, (q2, k2, v2), g.view(B, heads, T, 64).transpose(1, 2).contiguous())
    back = lambda t: t.transpose(1, 2).reshape(B * heads, t.shape[1], t.shape[3])
    assert torch.equal(back(got2), got)
    for a, b in zip(got2_g, got_g):
        assert torch.equal(back(a), b)


def _lap_np(ei, n):
    a = np.zeros((n, n))
    a[ei[0], ei[1]] = 1.0
    dinv = np.clip(a.sum(0), 1, None) ** -0.5
    lap = np.eye(n) - dinv[:, None] * a * dinv[None, :]
    return 0.5 * (lap + lap.T)


def _sym_edges(pairs):
    a, b = np.asarray(pairs, dtype=np.int64).reshape(-1, 2).T
    return np.stack([np.concatenate([a, b]), np.concatenate([b, a])])


def test_lap_eig_batched_against_numpy():
    """n2 (singa_lap_eig through graph.laplacian_pe_batched): per graph the 8 eigenvectors after the smallest of the
    normalised Laplacian, checked basis-free against numpy's eigh of the same matrix (reference model/CProMG.py:562-571;
    dgl's own output carries random signs, so the subspace is what can be compared): orthonormal columns, invariant
    subspace, Ritz values = eigenvalues 1..8, sign convention.  Cases: a path (distinct, clustered small eigenvalues), a
    ring with chords, many small components (a 30-fold zero eigenvalue), graphs with fewer than 9 nodes (zero-padded
    columns), a single node, an edgeless graph, 800-atom graphs (the two-pass register layout for n > 512) and a 1,000-atom
    graph (above the kernel's limit: the dense torch.linalg.eigh route)."""
    from singa_amd import graph as G
    rs = np.random.RandomState(0)
    graphs = []
    n = 300
    graphs.append((n, _sym_edges([(i, i + 1) for i in range(n - 1)])))                                   # path
    n = 470
    graphs.append((n, _sym_edges([(i, (i + 1) % n) for i in range(n)] + [(i, (i * 7 + 3) % n) for i in range(0, n, 5)])))
    n = 240
    graphs.append((n, _sym_edges([(8 * c + i, 8 * c + i + 1) for c in range(30) for i in range(7)])))    # 30 components
    graphs.append((5, _sym_edges([(0, 1), (1, 2), (3, 4)])))
    graphs.append((1, np.zeros((2, 0), np.int64)))
    graphs.append((12, np.zeros((2, 0), np.int64)))
    graphs.append((9, _sym_edges([(i, i + 1) for i in range(8)])))
    # repeated edges (count once) and one-directional edges (in-degree normalisation, then symmetrised)
    one_way = np.array([[0, 1, 2, 2, 5, 6, 7, 3, 3], [1, 2, 3, 3, 6, 7, 5, 0, 0]], np.int64)
    graphs.append((10, np.concatenate([one_way, _sym_edges([(8, 9), (8, 9), (4, 8)])], 1)))
    # components interleaved in index order (positions != atom indices inside the kernel)
    n = 60
    graphs.append((n, _sym_edges([(i, i + 3) for i in range(n - 3)])))                                   # three interleaved paths
    # big = 800: the kernel's two-pass register layout; big = 1000: above the kernel's 896-atom limit, i.e. the dense fp64
    # route of graph.laplacian_pe_batched (one batched torch.linalg.eigh on the device, padding rows sorted last) - a host-side
    # graph utility for molecules larger than anything BASELINE.json's configurations hold (config 5: 800 + 40 atoms)
    from singa_amd import _lib
    for big in (0, 800, 1000, -800, -1):
        # big < 0: the same cases with the SPARSE route (Chebyshev-filtered subspace iteration, taken by default for components of
        # >= 384 atoms) forced on every graph of more than 32 atoms: the clustered spectrum of the 300-atom path, the 30-fold zero
        # eigenvalue (more than the block holds: the route gives up and the dense one answers), interleaved components
        _lib.lib().singa_lap_pe_fsi_min(1 if big < 0 else 384)
        big = 0 if big == -1 else abs(big)
        gs = list(graphs)
        if big:
            n = big
            pairs = [(i, i + 1) for i in range(n - 1)] + [(int(a), int(b)) for a, b in rs.randint(0, n, (400, 2)) if a != b]
            gs = [(n, _sym_edges(pairs)), gs[0], gs[3]]
        off, eis, batch = 0, [], []
        for gi, (n, ei) in enumerate(gs):
            eis.append(ei + off)
            batch += [gi] * n
            off += n
        ei = torch.tensor(np.concatenate(eis, 1), device=DEV)
        bt = torch.tensor(batch, device=DEV)
        pe = G.laplacian_pe_batched(ei, bt, len(gs)).cpu().double().numpy()
        assert pe.shape == (off, 8) and np.isfinite(pe).all()
        off = 0
        for n, e in gs:
            v = pe[off:off + n]
            off += n
            lap = _lap_np(e, n)
            w = np.linalg.eigvalsh(lap)
            kk = min(8, n - 1)
            assert np.abs(v[:, kk:]).max(initial=0.0) == 0.0                      # columns the graph is too small for
            if kk == 0:
                continue
            v = v[:, :kk]
            assert np.abs(v.T @ v - np.eye(kk)).max() < 2e-6, n
            ritz = v.T @ lap @ v
            assert np.abs(lap @ v - v @ ritz).max() < 2e-6, n
            assert np.abs(np.linalg.eigvalsh(ritz) - w[1:kk + 1]).max() < 2e-6, n
            assert (v.max(0) >= (-v).max(0) - 1e-6).all()
    _lib.lib().singa_lap_pe_fsi_min(384)


@pytest.mark.parametrize("M", [1, 37, 26000])
def test_rowdot_bias_matches_torch(M):
    """k15e: scale * (x . b) per 32-channel row and its gradients (the hoisted bias term of CP:61-65) vs float64 torch."""
    ops = _ops()
    g = torch.Generator().manual_seed(M)
    x, b, gy = torch.randn(M // 4 + 1, 4, 32, generator=g), torch.randn(32, generator=g), torch.randn(M // 4 + 1, 4, generator=g)
    xr, br = x.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = 0.25 * (xr * br).sum(-1)
    (ref * gy.double()).sum().backward()
    xd, bd = x.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    out = ops.rowdot_bias(xd, bd, 0.25)
    assert out.shape == ref.shape and rel(out, ref) < 2e-6
    (out * gy.to(DEV)).sum().backward()
    assert rel(xd.grad, xr.grad) < 2e-6 and rel(bd.grad, br.grad) < 1e-5


def test_block_weight_autograd():
    """ops.block_weight through autograd on the GPU (and accumulated into an existing .grad by the gradient sink)."""
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    w = torch.randn(2 * 48, 40, generator=g)
    G = torch.randn(2 * 48, 2 * 40, generator=g)
    wr_ = w.double().requires_grad_(True)
    h = 48
    ref = torch.cat([torch.cat([wr_[:h], -wr_[h:]], 1), torch.cat([wr_[h:], wr_[:h]], 1)], 0)
    (ref * G.double()).sum().backward()
    wd = w.to(DEV).requires_grad_(True)
    out = ops.block_weight(wd)
    assert torch.equal(out.detach().cpu().double(), ref.detach())
    (out * G.to(DEV)).sum().backward()
    assert rel(wd.grad, wr_.grad) < 1e-6
    # gradient-sink mode: the backward adds into the existing buffer and returns nothing
    wd2 = w.to(DEV).requires_grad_(True)
    wd2.grad = torch.ones_like(wd2)
    ops._GradSink.on = True
    try:
        (ops.block_weight(wd2) * G.to(DEV)).sum().backward()
    finally:
        ops._GradSink.on = False
    assert rel(wd2.grad - 1.0, wr_.grad) < 1e-5


def test_knn_graph_kernel_matches_oracle():
    """n1: singa_knn_graph (reference model/CProMG.py:293,330 -> torch_cluster.knn_graph, flow='target_to_source') against the
    oracle's restatement on ragged batches: molecules smaller than k + 1 atoms (absent slots = -1), a single-atom molecule, an
    empty molecule, atoms of no molecule (the inert padding of graph.pad_batch), the three register layouts (<= 512 / 1024 / 2048
    atoms per molecule), exact ties (lower index first), neighbours in order of increasing distance."""
    from oracle import singa_oracle as O
    from singa_amd import ops
    g = torch.Generator().manual_seed(5)
    for sizes, k, mx in (([70, 5, 1, 0, 33, 49], 48, 80), ([600, 31], 30, 608), ([1100], 48, 1104)):
        B = len(sizes)
        n_real = sum(sizes)
        pos = torch.rand(n_real + 7, 3, generator=g) * 30.0
        if sizes[0] >= 70:                                   # exact ties: a 2 x 2 x 2 lattice corner inside molecule 0
            pos[:8] = torch.tensor([[x, y, z] for x in (1.0, 2.0) for y in (1.0, 2.0) for z in (1.0, 2.0)])
        batch = torch.cat([torch.repeat_interleave(torch.arange(B), torch.tensor(sizes)), torch.full((7,), B)])
        ptr = torch.tensor([0] + sizes).cumsum(0)
        out = ops.knn_graph(pos.to(DEV), k, batch.to(DEV), ptr.to(DEV), mx).cpu()
        N = pos.shape[0]
        assert out.shape == (2, N * k)
        row, col = out[0].view(N, k), out[1].view(N, k)
        assert (row[n_real:] == -1).all() and (col[n_real:] == -1).all()          # atoms of no molecule
        for b in range(B):
            ids = torch.arange(int(ptr[b]), int(ptr[b + 1]))
            if ids.numel() == 0:
                continue
            kk = min(k, ids.numel() - 1)
            d = torch.cdist(pos[ids].double(), pos[ids].double())
            df = pos[ids][:, None, :] - pos[ids][None, :, :]
            d32 = (df[..., 0] * df[..., 0] + df[..., 1] * df[..., 1]) + df[..., 2] * df[..., 2]     # the kernel's fp32 arithmetic
            for li, i in enumerate(ids.tolist()):
                assert (row[i, :kk] == i).all() and (row[i, kk:] == -1).all() and (col[i, kk:] == -1).all()
                nb = col[i, :kk]
                if kk == 0:
                    continue
                assert len(set(nb.tolist())) == kk and i not in nb.tolist()
                assert int(nb.min()) >= int(ids[0]) and int(nb.max()) <= int(ids[-1])        # same molecule
                dn = d32[li, nb - int(ids[0])]
                assert (dn[1:] >= dn[:-1]).all()                                              # increasing distance
                tie = dn[1:] == dn[:-1]
                assert (nb[1:][tie] > nb[:-1][tie]).all()                                     # exact ties: lower index first
                # the selected set is the oracle's (fp64 distances) wherever the k-th / (k+1)-th gap is above fp32 rounding
                want = d[li].clone()
                want[li] = float("inf")
                srt = torch.sort(want).values
                if kk < ids.numel() - 1 and float(srt[kk] - srt[kk - 1]) < 1e-5:
                    continue
                assert set(nb.tolist()) == set(ids[torch.topk(want, kk, largest=False).indices].tolist()), (b, i)
    # and the whole oracle list on a generated batch
    from singa_amd import graph as G
    L, kw, _, _ = G.resolve_workload("cfg3_b128_l4")
    graphs = [G.synthetic_graph(i, with_lap=False, **G.graph_sizes(i, **kw)) for i in (5, 6, 7)]
    bt = G.collate(graphs)
    for nt, k in ((G.PA, 48), (G.LA, 30)):
        pos, batch, ptr = bt[nt]["pos"], bt[nt]["batch"], bt[nt]["ptr"]
        own = ops.knn_graph(pos.to(DEV), k, batch.to(DEV), ptr.to(DEV), int((ptr[1:] - ptr[:-1]).max())).cpu()
        ref = O.knn_graph(pos, k, batch)
        N = pos.shape[0]
        ok = own[0] >= 0
        assert set((own[0][ok] * N + own[1][ok]).tolist()) == set((ref[0] * N + ref[1]).tolist())


@pytest.mark.parametrize("hidden,key,heads", [(256, 128, 4), (128, 64, 4), (256, 256, 8)])
def test_dense_attention_module_all_head_geometries(hidden, key, heads):
    """MultiHeadDeAttention (reference model/CProMG.py:134-158) against the oracle's dense_mha: the shipped head geometry
    (32 / 64 channels per head: the MFMA attention kernel k19 on fused projections) AND other geometries, which take the
    generic branch of CProMG._dense_attention (separate projections, masked softmax kernel, batched library products) -
    self attention with a causal + padding mask, cross attention with a padding mask, all parameter gradients."""
    from singa_amd.model.CProMG import MultiHeadDeAttention
    torch.manual_seed(hidden + key + heads)
    B, T, S = 3, 37, 53
    m = MultiHeadDeAttention(hidden, key, heads, device=DEV)
    sd = {"a." + k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    x = torch.randn(B, T, hidden)
    enc = torch.randn(B, S, hidden)
    causal = torch.triu(torch.ones(T, T, dtype=torch.bool), 1).unsqueeze(0).expand(B, T, T) | (torch.rand(B, 1, T) < 0.15)
    causal[:, :, 0] = False
    pad = (torch.rand(B, 1, S) < 0.3).expand(B, T, S)
    for Qc, Kc, mask in ((x, x, causal), (x, enc, pad)):
        for v in sd.values():
            v.grad = None
        m.zero_grad(set_to_none=True)
        q_o = Qc.clone().requires_grad_(True)
        k_o = q_o if Kc is Qc else Kc.clone().requires_grad_(True)
        ref = O.dense_mha(sd, "a", q_o, k_o, k_o, mask, heads=heads)
        g = torch.randn_like(ref)
        ref.backward(g)
        q_d = Qc.to(DEV).requires_grad_(True)
        k_d = q_d if Kc is Qc else Kc.to(DEV).requires_grad_(True)
        out = m(q_d, k_d, k_d, mask.to(DEV))
        out.backward(g.to(DEV))
        assert rel(out, ref) < 2e-5
        assert rel(q_d.grad, q_o.grad) < 5e-5
        if k_d is not q_d:
            assert rel(k_d.grad, k_o.grad) < 5e-5
        for name, p in m.named_parameters():
            want = sd["a." + name].grad
            if name == "W_K.bias":          # softmax is shift-invariant along the keys: this gradient is zero up to rounding
                assert float(p.grad.abs().max()) < 1e-5 and float(want.abs().max()) < 1e-5
                continue
            assert rel(p.grad, want) < 1e-4, name


def test_knn_edge_attr_kernel_matches_torch_form():
    """singa_knn_edge_attr (n1): Gaussian smearing of the edge lengths, get_laplacian's -w rows and degree rows, placed behind
    each other per centre node - against the torch form (GaussianSmearing, segment sums, stable argsort of [rows ; loops]) on a
    ragged row-sorted edge list with empty rows, a heavy row and inert padding edges behind the real ones."""
    from singa_amd import ops
    from singa_amd.model.CProMG import GaussianSmearing
    g = torch.Generator().manual_seed(3)
    N, n_real, extra = 37, 500, 23
    row = torch.sort(torch.randint(0, N - 4, (n_real,), generator=g)).values
    row[row == 5] = 6                                              # node 5: no edges
    row[100:260] = 11                                              # a heavy row
    row = torch.sort(row).values
    row = torch.cat([row, torch.full((extra,), N - 2)])            # padding edges among the last (padding) atoms
    ln = torch.rand(n_real, generator=g) * 14.0
    smear = GaussianSmearing(stop=15, num_gaussians=64, device=DEV)
    seg = torch.searchsorted(row, torch.arange(N + 1)).to(torch.int32)
    out = ops.knn_edge_attr(ln.to(DEV), seg.to(DEV), row.numel(), n_real, smear.offset, smear.coeff).cpu()
    ea = torch.cat([smear(ln.to(DEV)).cpu(), torch.zeros(extra, 64)])
    deg = torch.zeros(N, 64).index_add_(0, row, ea)
    order = torch.argsort(torch.cat([row, torch.arange(N)]), stable=True)
    want = torch.cat([-ea, deg])[order]
    assert out.shape == want.shape
    assert float((out - want).abs().max()) < 2e-6 * max(1.0, float(want.abs().max()))
    with pytest.raises(RuntimeError):
        ops.knn_edge_attr(ln.to(DEV), seg.to(DEV), row.numel(), n_real, smear.offset[:50].contiguous(), smear.coeff)
