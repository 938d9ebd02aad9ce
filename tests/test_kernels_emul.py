"""CPU pre-flight of the HIP kernels: singa_amd/csrc/singa_hip.hip is compiled with g++ against tests/emul (a
sequential stand-in for the HIP runtime) and every kernel without cross-lane intrinsics is run thread by thread and
compared with the oracle.  This checks index algebra, segment handling and bounds in the GPU-less container; the
same comparisons run against the real gfx950 build in tests/test_kernels_gpu.py (-m gpu)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import singa_oracle as O
from oracle import so3_tables as T
from singa_amd import _capi, so3

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("emul") / "libsinga_emul.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-x", "c++", "-I",
                           os.path.join(ROOT, "tests", "emul"), "-o", out,
                           os.path.join(ROOT, "singa_amd", "csrc", "singa_hip.hip")])
    lib = _capi.bind(out)
    jd = np.ascontiguousarray(so3.jd_flat(6))
    assert lib.singa_init(jd.ctypes.data_as(ctypes.c_void_p), 6) == 0
    assert lib.singa_so3_skinny_variant(1) == 0     # the lane-broadcast k11s kernels (the matrix-core ones run on the GPU only)
    return lib


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def iptr(a):
    return a.ctypes.data


def rand_graph(rs, n_src, n_dst, E):
    dst = np.sort(rs.randint(0, n_dst, E)).astype(np.int32)
    src = rs.randint(0, n_src, E).astype(np.int32)
    row_ptr = np.zeros(n_dst + 1, np.int32)
    np.add.at(row_ptr, dst + 1, 1)
    row_ptr = np.cumsum(row_ptr).astype(np.int32)
    eperm = np.argsort(src, kind="stable").astype(np.int32)
    col_ptr = np.zeros(n_src + 1, np.int32)
    np.add.at(col_ptr, src + 1, 1)
    col_ptr = np.cumsum(col_ptr).astype(np.int32)
    return src, dst, row_ptr, col_ptr, eperm


def rand_rot(rs, E):
    v = torch.tensor(rs.randn(E, 3), dtype=torch.float32)
    return O.edge_rot_mat(v, torch.tensor(rs.rand(E, 3), dtype=torch.float32))


def reduced_rows(w, L, M=2):
    """dense [E,K,K] Wigner -> [E,WSZ] reduced-row records."""
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
    return np.concatenate(idx)


@pytest.mark.parametrize("name", ["3wi2_4tpp", "4agq_5a7b", "5cp5_4nue"])
def test_edge_frames_match_reference_draws(emul, name):
    """k1 (the kernel source, run sequentially): on the reference's own torch.rand_like draws the frames equal the ones the
    reference built (EF:2286-2351; goldens recorded by oracle/make_golden.py), and the two guard statistics are the
    shortest edge and the largest |cos(edge, helper)| of the oracle's restatement."""
    from tests.test_oracle_conventions import reference_frame_cases
    for vec, rand, want in reference_frame_cases(name):
        vec, rand = vec.float().numpy().copy(), rand.float().numpy().copy()
        E = vec.shape[0]
        rot = np.full((E, 3, 3), np.nan, np.float32)
        stats = np.array([np.inf, 0.0], np.float32)
        assert emul.singa_edge_frames(ptr(vec), ptr(rand), ptr(rot), ptr(stats), E, None) == 0
        assert np.abs(rot - want.numpy()).max() < 1e-6
        assert abs(stats[0] - np.linalg.norm(vec, axis=1).min()) < 1e-6
        assert 0.0 <= stats[1] < 0.99


def test_edge_frame_guard_statistics(emul):
    """A short edge lowers stats[0] below the reference's 1e-4 threshold (EF:2292-2297); a zero-length edge makes the
    |cos| statistic NaN, which the product turns into the RuntimeError that stands for the reference's assert (EF:2329)."""
    rs = np.random.RandomState(3)
    vec, rand = rs.randn(16, 3).astype(np.float32), rs.rand(16, 3).astype(np.float32)
    vec[3] = [5e-5, 0.0, 0.0]
    rot = np.zeros((16, 3, 3), np.float32)
    stats = np.array([np.inf, 0.0], np.float32)
    assert emul.singa_edge_frames(ptr(vec), ptr(rand), ptr(rot), ptr(stats), 16, None) == 0
    assert abs(stats[0] - 5e-5) < 1e-9 and stats[1] < 0.99
    vec[3] = 0.0
    stats = np.array([np.inf, 0.0], np.float32)
    assert emul.singa_edge_frames(ptr(vec), ptr(rand), ptr(rot), ptr(stats), 16, None) == 0
    assert stats[0] == 0.0 and np.isnan(stats[1])
    # statistics accumulate across launches (what the captured graphs rely on)
    stats = np.array([0.25, 0.5], np.float32)
    v2 = np.tile(np.array([[1.0, 0.0, 0.0]], np.float32), (4, 1))
    r2 = np.tile(np.array([[0.5, 0.9, 0.5]], np.float32), (4, 1))
    assert emul.singa_edge_frames(ptr(v2), ptr(r2), ptr(rot), ptr(stats), 4, None) == 0
    assert stats[0] == 0.25 and stats[1] == 0.5


@pytest.mark.parametrize("L", [2, 4, 6])
def test_wigner_rows(emul, L):
    rs = np.random.RandomState(L)
    E = 23
    rot = rand_rot(rs, E)
    lay = so3.layout(L, 2)
    wr = np.zeros((E, lay.WSZ), np.float32)
    assert emul.singa_wigner_rows(ptr(rot.numpy()), ptr(wr), E, L, 2, None) == 0
    ref = reduced_rows(O.wigner_dense(rot, L), L).numpy()
    assert np.abs(wr - ref).max() < 2e-5


@pytest.mark.parametrize("L", [2, 4, 6])
def test_gather_rotate(emul, L):
    rs = np.random.RandomState(10 + L)
    C, Ns, Nd, E = 16, 9, 7, 31
    lay = so3.layout(L, 2)
    src, dst, row_ptr, col_ptr, eperm = rand_graph(rs, Ns, Nd, E)
    rot = rand_rot(rs, E)
    fr = O.Frame(rot, L, 2)
    wr = reduced_rows(O.wigner_dense(rot, L), L).numpy().copy()
    xs = torch.tensor(rs.randn(Ns, lay.K, C), dtype=torch.float32, requires_grad=True)
    xd = torch.tensor(rs.randn(Nd, lay.K, C), dtype=torch.float32, requires_grad=True)
    rad = torch.tensor(rs.randn(E, lay.rad_rows, 2 * C), dtype=torch.float32, requires_grad=True)
    ridx = torch.as_tensor(rad_row_index(lay))
    ref = torch.bmm(fr.fwd, torch.cat([xs[src.astype(np.int64)], xd[dst.astype(np.int64)]], 2))[:, fr.to_m] * rad[:, ridx]
    out = np.zeros((E, lay.KR, 2 * C), np.float32)
    assert emul.singa_gather_rotate_fwd(ptr(xs.detach().numpy()), ptr(xd.detach().numpy()), ptr(src), ptr(dst), ptr(wr),
                                        ptr(rad.detach().numpy()), ptr(out), E, C, L, 2, None) == 0
    assert np.abs(out - ref.detach().numpy()).max() < 1e-4
    g = torch.tensor(rs.randn(*out.shape), dtype=torch.float32)
    ref.backward(g)
    g_rad = np.zeros((E, lay.rad_rows, 2 * C), np.float32)
    gxs = np.full((Ns, lay.K, C), np.nan, np.float32)
    gxd = np.full((Nd, lay.K, C), np.nan, np.float32)
    code = emul.singa_gather_rotate_bwd(ptr(g.numpy()), ptr(xs.detach().numpy()), ptr(xd.detach().numpy()), ptr(src),
                                        ptr(dst), ptr(wr), ptr(rad.detach().numpy()), ptr(row_ptr), ptr(col_ptr),
                                        ptr(eperm), ptr(g_rad), ptr(gxs), ptr(gxd), E, Ns, Nd, C, L, 2, None)
    assert code == 0
    assert np.abs(g_rad - rad.grad.numpy()).max() < 2e-4
    assert np.abs(gxs - xs.grad.numpy()).max() < 2e-4
    assert np.abs(gxd - xd.grad.numpy()).max() < 2e-4


@pytest.mark.parametrize("L,CH,heads,m0", [(2, 112, 7, 0), (4, 112, 7, 0), (6, 112, 7, 0), (6, 16, 1, 1), (2, 16, 1, 1)])
def test_rotate_back_scatter(emul, L, CH, heads, m0):
    rs = np.random.RandomState(20 + L + m0)
    Nd, E = 8, 29
    lay = so3.layout(L, 2)
    _, dst, row_ptr, _, _ = rand_graph(rs, 5, Nd, E)
    rot = rand_rot(rs, E)
    fr = O.Frame(rot, L, 2)
    wr = reduced_rows(O.wigner_dense(rot, L), L).numpy().copy()
    scale = 0.37 if m0 else 1.0
    rows = lay.seg_rows if not m0 else [lay.m_size[0]]
    parts = [torch.tensor(rs.randn(E, r * CH), dtype=torch.float32, requires_grad=True) for r in rows]
    alpha = None if m0 else torch.tensor(rs.rand(E, heads), dtype=torch.float32, requires_grad=True)
    msg_m = torch.cat([p.view(E, -1, CH) for p in parts], 1)
    if m0:
        msg_m = torch.cat([msg_m, msg_m.new_zeros(E, lay.KR - rows[0], CH)], 1)
    msg_l = msg_m[:, fr.to_l]
    if alpha is not None:
        msg_l = (msg_l.view(E, lay.KR, heads, CH // heads) * alpha.view(E, 1, heads, 1)).reshape(E, lay.KR, CH)
    ref = O.seg_sum(torch.bmm(fr.inv, msg_l), torch.as_tensor(dst.astype(np.int64)), Nd) * scale
    seg_in, n = _capi.segs([(iptr(p.detach().numpy()), r * CH, r) for p, r in zip(parts, rows)])
    out = np.full((Nd, lay.K, CH), np.nan, np.float32)
    code = emul.singa_rotate_back_scatter_fwd(seg_in, n, ptr(alpha.detach().numpy()) if alpha is not None else None,
                                              ptr(wr), ptr(row_ptr), ptr(out), Nd, CH, heads, L, 2, m0, scale, None)
    assert code == 0, emul.singa_last_error_string()
    assert np.abs(out - ref.detach().numpy()).max() < 2e-4
    g = torch.tensor(rs.randn(*out.shape), dtype=torch.float32)
    ref.backward(g)
    gparts = [np.full((E, r * CH), np.nan, np.float32) for r in rows]
    seg_out, _ = _capi.segs([(iptr(p), r * CH, r) for p, r in zip(gparts, rows)])
    gap = np.zeros((E, CH), np.float32)
    code = emul.singa_rotate_back_scatter_bwd(ptr(g.numpy()), seg_in, seg_out, n,
                                              ptr(alpha.detach().numpy()) if alpha is not None else None, ptr(wr),
                                              ptr(row_ptr), ptr(gap) if alpha is not None else None, Nd, CH, heads, L, 2,
                                              m0, scale, None)
    assert code == 0, emul.singa_last_error_string()
    for gp, p in zip(gparts, parts):
        assert np.abs(gp - p.grad.numpy()).max() < 2e-4
    if alpha is not None:
        assert np.abs(gap.reshape(E, heads, -1).sum(-1) - alpha.grad.numpy()).max() < 5e-4


@pytest.mark.parametrize("H,eps", [(7, 1e-16), (4, 0.0)])
def test_segment_softmax(emul, H, eps):
    rs = np.random.RandomState(3)
    N, E = 9, 40
    _, dst, row_ptr, _, _ = rand_graph(rs, 3, N, E)
    x = torch.tensor(rs.randn(E, H) * 3, dtype=torch.float32, requires_grad=True)
    ref = O.seg_softmax(x, torch.as_tensor(dst.astype(np.int64)), N, eps)
    y = np.zeros((E, H), np.float32)
    assert emul.singa_segment_softmax_fwd(ptr(x.detach().numpy()), ptr(row_ptr), ptr(y), N, H, eps, 0, None) == 0
    assert np.abs(y - ref.detach().numpy()).max() < 1e-6
    g = torch.tensor(rs.randn(E, H), dtype=torch.float32)
    ref.backward(g)
    gx = np.zeros((E, H), np.float32)
    assert emul.singa_segment_softmax_bwd(ptr(y), ptr(g.numpy()), ptr(row_ptr), ptr(gx), N, H, 0, None) == 0
    assert np.abs(gx - x.grad.numpy()).max() < 1e-5


@pytest.mark.parametrize("L,M,C,edge", [(2, 2, 128, True), (4, 2, 128, True), (6, 2, 128, True), (4, 4, 512, False),
                                        (6, 6, 512, False), (2, 2, 512, False)])
def test_s2act(emul, L, M, C, edge):
    """attention flavour: m-primary rows in three strided segments (the SO(2)-conv GEMM outputs, gate inside the
    m=0 buffer); FFN flavour: one contiguous l-primary [N,K,C] tensor."""
    rs = np.random.RandomState(L * 10 + M)
    E = 5
    lay = so3.layout(L, M)
    to, fr_ = so3.s2_grid(L, M)
    G, KIN = to.shape
    if edge:
        to_k, fr_k = to[:, lay.to_m], fr_[:, lay.to_m]
        extra = 96
        h0 = torch.tensor(rs.randn(E, extra + C + lay.seg_rows[0] * C), dtype=torch.float32, requires_grad=True)
        h1 = torch.tensor(rs.randn(E, lay.seg_rows[1] * C), dtype=torch.float32, requires_grad=True)
        h2 = torch.tensor(rs.randn(E, lay.seg_rows[2] * C), dtype=torch.float32, requires_grad=True)
        gate = h0[:, extra:extra + C]
        xm = torch.cat([h0[:, extra + C:].view(E, -1, C), h1.view(E, -1, C), h2.view(E, -1, C)], 1)
        to_l = torch.as_tensor(np.argsort(lay.to_m))
        ref = O.sep_s2_act(gate, xm[:, to_l], L, M)[:, torch.as_tensor(lay.to_m)]
        base = h0.detach().numpy()
        items = [(iptr(base) + 4 * (extra + C), base.shape[1], lay.seg_rows[0]),
                 (iptr(h1.detach().numpy()), h1.shape[1], lay.seg_rows[1]),
                 (iptr(h2.detach().numpy()), h2.shape[1], lay.seg_rows[2])]
        gate_ptr, ldg = iptr(base) + 4 * extra, base.shape[1]
    else:
        to_k, fr_k = to, fr_
        x = torch.tensor(rs.randn(E, KIN, C), dtype=torch.float32, requires_grad=True)
        gt = torch.tensor(rs.randn(E, C), dtype=torch.float32, requires_grad=True)
        ref = O.sep_s2_act(gt, x, L, M)
        items = [(iptr(x.detach().numpy()), KIN * C, KIN)]
        gate_ptr, ldg = iptr(gt.detach().numpy()), C
    to32 = np.ascontiguousarray(to_k, np.float32)
    fr32 = np.ascontiguousarray(fr_k, np.float32)
    seg, n = _capi.segs(items)
    out = np.full((E, KIN, C), np.nan, np.float32)
    code = emul.singa_s2act_fwd(seg, n, gate_ptr, ldg, ptr(to32), ptr(fr32), ptr(out), E, C, KIN, G, None)
    assert code == 0, emul.singa_last_error_string()
    assert np.abs(out - ref.detach().numpy()).max() < 3e-5 * max(1.0, float(ref.abs().max()))
    g = torch.tensor(rs.randn(E, KIN, C), dtype=torch.float32)
    ref.backward(g)
    gx = np.full((E, KIN, C), np.nan, np.float32)
    gg = np.full((E, C), np.nan, np.float32)
    code = emul.singa_s2act_bwd(seg, n, gate_ptr, ldg, ptr(to32), ptr(fr32), ptr(g.numpy()), ptr(gx), ptr(gg), E, C, KIN,
                                G, None)
    assert code == 0
    if edge:
        want = torch.cat([h0.grad[:, extra + C:].view(E, -1, C), h1.grad.view(E, -1, C), h2.grad.view(E, -1, C)], 1)
        want_g = h0.grad[:, extra:extra + C]
    else:
        want, want_g = x.grad, gt.grad
    tol = 1e-4 * max(1.0, float(want.abs().max()))
    assert np.abs(gx - want.numpy()).max() < tol
    assert np.abs(gg - want_g.numpy()).max() < 1e-5


def test_argument_errors(emul):
    assert emul.singa_wigner_rows(None, None, 4, 6, 2, None) == -1
    a = np.zeros(16, np.float32)
    assert emul.singa_wigner_rows(ptr(a), ptr(a), 1, 5, 2, None) == -2
    assert emul.singa_wigner_rows(ptr(a), ptr(a), 1, 6, 3, None) == -2
    assert b"lmax" in emul.singa_last_error_string() or b"mmax" in emul.singa_last_error_string()
    kr, wsz, rr = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    for L in (2, 4, 6):
        assert emul.singa_dims(L, 2, ctypes.byref(kr), ctypes.byref(wsz), ctypes.byref(rr)) == 0
        lay = so3.layout(L, 2)
        assert (kr.value, wsz.value, rr.value) == (lay.KR, lay.WSZ, lay.rad_rows)


def test_colsum(emul):
    rs = np.random.RandomState(1)
    for M, n, ld in ((1, 5, 5), (128, 3, 3), (700, 33, 40), (5000, 300, 300), (40000, 7, 9)):
        x = rs.randn(M, ld).astype(np.float32)
        part = np.zeros(emul.singa_colsum_work(M, n), np.float32)
        out = np.zeros(n, np.float32)
        assert emul.singa_colsum(ptr(x), ld, M, n, ptr(part), ptr(out), None) == 0
        assert np.abs(out - x[:, :n].astype(np.float64).sum(0)).max() < 1e-3


def test_colsum_multi(emul):
    """Many column sums in one call, each ADDED into its destination segments (the parameter-gradient queue of one
    backward pass): single-slab and multi-slab jobs, strided sources, several segments per job, more jobs and segments
    than one launch carries."""
    rs = np.random.RandomState(3)
    shapes = [(1, 5, 5), (17, 3, 3), (700, 33, 40), (5000, 300, 300), (40000, 7, 9), (0, 4, 4)] + \
             [(int(rs.randint(1, 3000)), int(rs.randint(1, 70)), 80) for _ in range(90)]
    # wide jobs whose segments start at multiples of 4: the float4 path (the partial slabs of the weight-gradient GEMMs)
    vec_cuts = {(49, 4096, 4096): [0, 1024, 3072], (64, 2048, 2048): [0], (300, 1024, 1028): [0, 512], (16, 1536, 1536): [0, 4]}
    shapes += list(vec_cuts)
    xs, dsts, refs, segs = [], [], [], []
    for M, n, ld in shapes:
        x = rs.randn(M, ld).astype(np.float32)
        cuts = sorted(set([0] + ([int(c) for c in rs.randint(1, n, size=rs.randint(0, 3))] if n > 1 else [])))
        cuts = vec_cuts.get((M, n, ld), cuts)
        d = [rs.randn(b - a).astype(np.float32) for a, b in zip(cuts, cuts[1:] + [n])]
        tot = x[:, :n].astype(np.float64).sum(0)
        refs.append([di.astype(np.float64) + tot[a:a + di.size] for di, a in zip(d, cuts)])
        xs.append(x)
        dsts.append(d)
        segs.append(cuts)
    nj = len(shapes)
    ns = sum(len(c) for c in segs)
    X, LD, MM, NN = (ctypes.c_void_p * nj)(), (ctypes.c_longlong * nj)(), (ctypes.c_longlong * nj)(), (ctypes.c_int * nj)()
    S0, C0, D = (ctypes.c_int * nj)(), (ctypes.c_int * ns)(), (ctypes.c_void_p * ns)()
    q = work = 0
    for k, ((M, n, ld), x) in enumerate(zip(shapes, xs)):
        X[k], LD[k], MM[k], NN[k], S0[k] = x.ctypes.data, ld, M, n, q
        work += emul.singa_colsum_multi_work(M, n)
        for c, d in zip(segs[k], dsts[k]):
            C0[q], D[q] = c, d.ctypes.data
            q += 1
    w = np.zeros(work + 1, np.float32)
    assert emul.singa_colsum_multi(nj, X, LD, MM, NN, S0, ns, C0, D, ptr(w), work, None) == 0
    for d, r in zip(dsts, refs):
        for di, ri in zip(d, r):
            assert np.abs(di - ri).max() < 2e-3
    assert emul.singa_colsum_multi(nj, X, LD, MM, NN, S0, ns, C0, D, ptr(w), work // 2, None) == -3    # workspace too small


@pytest.mark.parametrize("L,edge", [(2, True), (4, True), (6, True), (2, False), (4, False), (6, False)])
def test_s2act_separable(emul, L, edge):
    """Separable (Legendre x Fourier) S2 activation == the dense-grid oracle, forward and backward."""
    rs = np.random.RandomState(77 + L)
    M = 2 if edge else L
    C = 128 if edge else 512
    E = 3
    lay = so3.layout(L, M)
    P, Q, A = (np.ascontiguousarray(t, np.float32) for t in so3.s2_grid_factors(L, M, edge))
    KIN = P.shape[1]
    if edge:
        extra = 64
        h0 = torch.tensor(rs.randn(E, extra + C + lay.seg_rows[0] * C), dtype=torch.float32, requires_grad=True)
        h1 = torch.tensor(rs.randn(E, lay.seg_rows[1] * C), dtype=torch.float32, requires_grad=True)
        h2 = torch.tensor(rs.randn(E, lay.seg_rows[2] * C), dtype=torch.float32, requires_grad=True)
        xm = torch.cat([h0[:, extra + C:].view(E, -1, C), h1.view(E, -1, C), h2.view(E, -1, C)], 1)
        to_l = torch.as_tensor(np.argsort(lay.to_m))
        ref = O.sep_s2_act(h0[:, extra:extra + C], xm[:, to_l], L, M)[:, torch.as_tensor(lay.to_m)]
        base = h0.detach().numpy()
        items = [(iptr(base) + 4 * (extra + C), base.shape[1], lay.seg_rows[0]),
                 (iptr(h1.detach().numpy()), h1.shape[1], lay.seg_rows[1]),
                 (iptr(h2.detach().numpy()), h2.shape[1], lay.seg_rows[2])]
        gate_ptr, ldg = iptr(base) + 4 * extra, base.shape[1]
    else:
        x = torch.tensor(rs.randn(E, KIN, C), dtype=torch.float32, requires_grad=True)
        gt = torch.tensor(rs.randn(E, C), dtype=torch.float32, requires_grad=True)
        ref = O.sep_s2_act(gt, x, L, M)
        items = [(iptr(x.detach().numpy()), KIN * C, KIN)]
        gate_ptr, ldg = iptr(gt.detach().numpy()), C
    seg, n = _capi.segs(items)
    out = np.full((E, KIN, C), np.nan, np.float32)
    code = emul.singa_s2act_sep_fwd(seg, n, gate_ptr, ldg, ptr(P), ptr(Q), ptr(A), ptr(out), E, C, L, None)
    assert code == 0, emul.singa_last_error_string()
    assert np.abs(out - ref.detach().numpy()).max() < 3e-5 * max(1.0, float(ref.detach().abs().max()))
    g = torch.tensor(rs.randn(E, KIN, C), dtype=torch.float32)
    ref.backward(g)
    gx = np.full((E, KIN, C), np.nan, np.float32)
    gg = np.full((E, C), np.nan, np.float32)
    code = emul.singa_s2act_sep_bwd(seg, n, gate_ptr, ldg, ptr(P), ptr(Q), ptr(A), ptr(g.numpy()), ptr(gx), ptr(gg), E, C,
                                    L, None)
    assert code == 0
    if edge:
        want = torch.cat([h0.grad[:, extra + C:].view(E, -1, C), h1.grad.view(E, -1, C), h2.grad.view(E, -1, C)], 1)
        want_g = h0.grad[:, extra:extra + C]
    else:
        want, want_g = x.grad, gt.grad
    assert np.abs(gx - want.numpy()).max() < 1e-4 * max(1.0, float(want.abs().max()))
    assert np.abs(gg - want_g.numpy()).max() < 1e-5
    # the segmented form: the same gradient written into column blocks of wider tensors (what ops._EdgeHead uses to assemble
    # the gradients of an SO(2) convolution's outputs in place) - bit-identical to the contiguous form
    rows = [it[2] for it in items]
    pad = 12
    bufs = [np.full((E, pad + (C if k == 0 else 0) + r * C + 4), np.nan, np.float32) for k, r in enumerate(rows)]
    gsegs, _ = _capi.segs([(iptr(b) + 4 * (pad + (C if k == 0 else 0)), b.shape[1], r) for k, (b, r) in enumerate(zip(bufs, rows))])
    code = emul.singa_s2act_sep_bwd_seg(seg, n, gate_ptr, ldg, ptr(P), ptr(Q), ptr(A), ptr(g.numpy()), gsegs,
                                        iptr(bufs[0]) + 4 * pad, bufs[0].shape[1], E, C, L, None)
    assert code == 0, emul.singa_last_error_string()
    at = 0
    for k, (b, r) in enumerate(zip(bufs, rows)):
        off = pad + (C if k == 0 else 0)
        assert np.array_equal(b[:, off:off + r * C].reshape(E, r, C), gx[:, at:at + r])
        assert np.isnan(b[:, :pad]).all() and np.isnan(b[:, off + r * C:]).all()
        at += r
    assert np.array_equal(bufs[0][:, pad:pad + C], gg)
    short, _ = _capi.segs([(iptr(b), r * C - 1, r) for b, r in zip(bufs, rows)])
    assert emul.singa_s2act_sep_bwd_seg(seg, n, gate_ptr, ldg, ptr(P), ptr(Q), ptr(A), ptr(g.numpy()), short,
                                        iptr(bufs[0]), C, E, C, L, None) == -3


@pytest.mark.parametrize("L,C", [(2, 512), (4, 512), (6, 512), (4, 112), (6, 112)])
def test_so3_skinny_kernels(emul, L, C):
    """k11s: the 16 <-> 512 channel SO3 linears as VALU kernels - expand (forward 16 -> 512 with bias, and d x of 512 -> 16
    through weight[l][u][c]) and reduce (both weight gradients + the bias row) against einsum in float64."""
    rs = np.random.RandomState(40 + L)
    N, K = 13, (L + 1) ** 2
    deg = np.asarray(so3.layout(L, L).degree)
    small = rs.randn(N, K, 16).astype(np.float32)
    w_cu = (rs.randn(L + 1, C, 16) * 0.25).astype(np.float32)           # weight[l][c][u] of a 16 -> 512 map
    w_uc = (rs.randn(L + 1, 16, C) * 0.25).astype(np.float32)           # weight[l][u][c] of a 512 -> 16 map
    bias = rs.randn(C).astype(np.float32)
    big = np.full((N, K, C), np.nan, np.float32)
    assert emul.singa_so3_skinny_expand(ptr(small), ptr(w_cu), C * 16, 16, 1, ptr(bias), ptr(big), N, C, L, None) == 0
    want = np.einsum("nku,kcu->nkc", small.astype(np.float64), w_cu[deg].astype(np.float64))
    want[:, 0] += bias
    assert np.abs(big - want).max() < 1e-4
    big2 = np.full((N, K, C), np.nan, np.float32)
    assert emul.singa_so3_skinny_expand(ptr(small), ptr(w_uc), 16 * C, 1, C, None, ptr(big2), N, C, L, None) == 0
    assert np.abs(big2 - np.einsum("nku,kuc->nkc", small.astype(np.float64), w_uc[deg].astype(np.float64))).max() < 1e-4
    # reductions
    g = rs.randn(N, K, C).astype(np.float32)
    onehot = np.eye(L + 1)[deg]                                           # [K, L+1]
    ref_uc = np.einsum("kl,nku,nkc->luc", onehot, small.astype(np.float64), g.astype(np.float64))
    nparts = emul.singa_so3_skinny_nparts(N, L, C)
    wsz = (L + 1) * 16 * C
    for out_cu, bias_row in ((1, 1), (0, 0)):
        part = np.full((nparts, wsz + (C if bias_row else 0)), np.nan, np.float32)
        assert emul.singa_so3_skinny_reduce(ptr(small), ptr(g), ptr(part), N, C, L, out_cu, bias_row, None) == 0
        tot = part.astype(np.float64).sum(0)
        got = tot[:wsz].reshape(L + 1, C, 16).transpose(0, 2, 1) if out_cu else tot[:wsz].reshape(L + 1, 16, C)
        assert np.abs(got - ref_uc).max() < 1e-3 * max(1.0, np.abs(ref_uc).max())
        if bias_row:
            assert np.abs(tot[wsz:] - g[:, 0].astype(np.float64).sum(0)).max() < 1e-4
    assert emul.singa_so3_skinny_expand(ptr(small), ptr(w_cu), C * 16, 16, 1, None, ptr(big), N, C, 5, None) == -2
    assert emul.singa_so3_skinny_expand(ptr(small), ptr(w_cu), C * 16, 16, 1, None, ptr(big), N, 100, L, None) == -3


def test_ln_silu(emul):
    """k6a: SiLU(LayerNorm(x)) over 16 channels and its backward, against torch on the CPU."""
    rs = np.random.RandomState(5)
    M, C = 333, 16
    x = (rs.randn(M, C) * 2 + 0.5).astype(np.float32)
    gamma, beta = rs.randn(C).astype(np.float32), rs.randn(C).astype(np.float32)
    g = rs.randn(M, C).astype(np.float32)
    out = np.zeros_like(x)
    assert emul.singa_ln_silu_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(out), M, C, 1e-5, None) == 0
    xt = torch.tensor(x, requires_grad=True)
    gt, bt = torch.tensor(gamma, requires_grad=True), torch.tensor(beta, requires_grad=True)
    ref = torch.nn.functional.silu(torch.nn.functional.layer_norm(xt, (C,), gt, bt, 1e-5))
    ref.backward(torch.tensor(g))
    assert np.abs(out - ref.detach().numpy()).max() < 2e-5
    n = emul.singa_ln_silu_nparts(M)
    gx, part = np.zeros_like(x), np.zeros((n, 2 * C), np.float32)
    assert emul.singa_ln_silu_bwd(ptr(x), ptr(gamma), ptr(beta), ptr(g), ptr(gx), ptr(part), M, C, 1e-5, None) == 0
    assert np.abs(gx - xt.grad.numpy()).max() < 5e-5 * np.abs(xt.grad.numpy()).max()
    s = part.sum(0)
    assert np.abs(s[:C] - gt.grad.numpy()).max() < 1e-4 * np.abs(gt.grad.numpy()).max()
    assert np.abs(s[C:] - bt.grad.numpy()).max() < 1e-4 * np.abs(bt.grad.numpy()).max()
    assert emul.singa_ln_silu_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(out), M, 32, 1e-5, None) == -3


def test_block_weight(emul):
    """[[Wr, -Wi], [Wi, Wr]] of an SO2_m_Convolution weight [Wr; Wi] (EF:677-729, 721-729) and the map of its gradient back
    to the weight, against the torch expression the reference's recombination implies."""
    rs = np.random.RandomState(4)
    h, k = 6, 10
    w = torch.tensor(rs.randn(2 * h, k), dtype=torch.float32, requires_grad=True)
    wr, wi = w[:h], w[h:]
    ref = torch.cat([torch.cat([wr, -wi], 1), torch.cat([wi, wr], 1)], 0)
    out = np.full((2 * h, 2 * k), np.nan, np.float32)
    assert emul.singa_block_weight_fwd(ptr(w.detach().numpy()), ptr(out), h, k, None) == 0
    assert np.array_equal(out, ref.detach().numpy())
    G = torch.tensor(rs.randn(2 * h, 2 * k), dtype=torch.float32)
    ref.backward(G)
    gw = np.full((2 * h, k), np.nan, np.float32)
    assert emul.singa_block_weight_bwd(ptr(G.numpy()), ptr(gw), h, k, 0, None) == 0
    assert np.abs(gw - w.grad.numpy()).max() < 1e-6
    base = rs.randn(2 * h, k).astype(np.float32)
    acc = base.copy()
    assert emul.singa_block_weight_bwd(ptr(G.numpy()), ptr(acc), h, k, 1, None) == 0
    assert np.abs(acc - (base + w.grad.numpy())).max() < 1e-6
