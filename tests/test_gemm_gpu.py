"""-m gpu: the library's f32 MFMA GEMM (k7 / k11, singa_gemm_f32) - raw products against float64 on the CPU, the SO(2)
convolution against the oracle's restatement of the reference (EF:807-875, 715-729: per-m Linear with the (r0 - i1,
r1 + i0) recombination) and SO3_LinearV2 against EF:655-671, forward and every gradient, plus the BLAS-library path as a
second opinion."""
import numpy as np
import pytest
import torch

from oracle import singa_oracle as O
from tests.helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _f64(t):
    return t.detach().cpu().double()


@pytest.fixture
def tile_shape(request):
    """Force the 128 x 128 (0) or the 64 x 64 (3) macro tile wherever the library would choose between the two."""
    from singa_amd import _lib
    assert _lib.lib().singa_gemm_force_cfg(request.param) == 0
    yield request.param
    assert _lib.lib().singa_gemm_force_cfg(-1) == 0


@pytest.mark.parametrize("tile_shape", [0, 3], indirect=True)
@pytest.mark.parametrize("M,N,K", [(1, 16, 16), (130, 40, 36), (129, 992, 160), (1000, 560, 640), (257, 16, 112), (300, 32, 512),
                                   (64, 1024, 256), (5000, 128, 8)])
def test_gemm_nt_nn_tn_against_float64(M, N, K, tile_shape):
    """y = x W^T + b, dx = dy W, dW = dy^T x (split reduction) for sizes with ragged tiles in every dimension, on both
    macro-tile shapes; exact integers first (any indexing slip shows as an O(1) error), then random floats."""
    from singa_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    for kind in ("int", "float"):
        mk = (lambda *s: torch.randint(-3, 4, s, generator=g).float()) if kind == "int" else (lambda *s: torch.randn(*s, generator=g))
        x, w, b, dy = mk(M, K), mk(N, K), mk(N), mk(M, N)
        xd, wd, bd, dyd = x.to(DEV), w.to(DEV), b.to(DEV), dy.to(DEV)
        tol = 0 if kind == "int" else 2e-6
        y = ops.gemm_nt(xd, wd, bd)
        ref = _f64(x) @ _f64(w).t() + _f64(b)
        assert float((_f64(y) - ref).abs().max()) <= tol * float(ref.abs().max()) + (0 if kind == "int" else 1e-6)
        # dx through the (1, 0) form, dW through the split (0, 0) form
        dx = torch.empty(M, K, device=DEV)
        ops._gemm([dict(a=dyd.data_ptr(), lda=N, b=wd.data_ptr(), ldb=K, c=dx.data_ptr(), ldc=K, I=M, J=K, R=N)], True, False)
        ref = _f64(dy) @ _f64(w)
        assert float((_f64(dx) - ref).abs().max()) <= tol * float(ref.abs().max()) + (0 if kind == "int" else 1e-6)
        for S in (1, 3):
            part = torch.full((S, N * K), float("nan"), device=DEV)
            asum = torch.full((S, N), float("nan"), device=DEV)        # column sums of dy per split (the bias gradient)
            ops._gemm([dict(a=dyd.data_ptr(), lda=N, b=xd.data_ptr(), ldb=K, c=part.data_ptr(), ldc=K, I=N, J=K, R=M,
                            c_split_stride=N * K, asum=asum.data_ptr(), asum_stride=N)], False, False, S)
            ref = _f64(dy).t() @ _f64(x)
            got = _f64(part).sum(0).view(N, K)
            assert float((got - ref).abs().max()) <= tol * float(ref.abs().max()) + (0 if kind == "int" else 2e-6 * M ** 0.5)
            refb = _f64(dy).sum(0)
            assert float((_f64(asum).sum(0) - refb).abs().max()) <= (0 if kind == "int" else 2e-6 * M ** 0.5 + 1e-6 * float(refb.abs().max()))


@pytest.mark.parametrize("M,N,K", [(1, 4, 4), (130, 36, 20), (129, 96, 128), (1000, 512, 96), (257, 64, 16), (3000, 384, 512)])
def test_complex_gemm_3m_all_forms_against_float64(M, N, K):
    """singa_cgemm3m_f32 (k7c): forward (1, 1), d input (1, 0), d weight (0, 0, split) of out = (x_re + i x_im)(Wr + i Wi)^T
    with ragged tiles in every dimension; exact on small integers (any indexing slip or sign error is an O(1) error), then
    random floats against float64.  M rows, N complex outputs, K complex inputs."""
    from singa_amd import ops
    g = torch.Generator().manual_seed(M * 5 + N)
    for kind in ("int", "float"):
        mk = (lambda *s: torch.randint(-3, 4, s, generator=g).float()) if kind == "int" else (lambda *s: torch.randn(*s, generator=g))
        x, w, dy = mk(M, 2 * K), mk(2 * N, K), mk(M, 2 * N)
        xd, wd, dyd = x.to(DEV), w.to(DEV), dy.to(DEV)
        xr, xi, wr, wi, gr, gi = _f64(x[:, :K]), _f64(x[:, K:]), _f64(w[:N]), _f64(w[N:]), _f64(dy[:, :N]), _f64(dy[:, N:])
        tol = 0 if kind == "int" else 3e-6
        slack = 0 if kind == "int" else 1e-6

        def check(got, ref, extra=0.0):
            assert float((_f64(got) - ref).abs().max()) <= tol * float(ref.abs().max()) + slack + extra

        y = torch.full((M, 2 * N), float("nan"), device=DEV)
        ops._cgemm([dict(a=xd.data_ptr(), lda=2 * K, a_im=K, b=wd.data_ptr(), ldb=K, b_im=N * K, c=y.data_ptr(), ldc=2 * N, c_im=N,
                         I=M, J=N, R=K, sigma=1.0)], True, True)
        check(y, torch.cat([xr @ wr.t() - xi @ wi.t(), xr @ wi.t() + xi @ wr.t()], 1))
        dx = torch.full((M, 2 * K), float("nan"), device=DEV)
        ops._cgemm([dict(a=dyd.data_ptr(), lda=2 * N, a_im=N, b=wd.data_ptr(), ldb=K, b_im=N * K, c=dx.data_ptr(), ldc=2 * K, c_im=K,
                         I=M, J=K, R=N, sigma=-1.0)], True, False)
        check(dx, torch.cat([gr @ wr + gi @ wi, gi @ wr - gr @ wi], 1))
        ref = torch.cat([gr.t() @ xr + gi.t() @ xi, gi.t() @ xr - gr.t() @ xi], 0)
        for S in (1, 3):
            part = torch.full((S, 2 * N * K), float("nan"), device=DEV)
            ops._cgemm([dict(a=dyd.data_ptr(), lda=2 * N, a_im=N, b=xd.data_ptr(), ldb=2 * K, b_im=K, c=part.data_ptr(), ldc=K,
                             c_im=N * K, I=N, J=K, R=M, sigma=-1.0, c_split_stride=2 * N * K)], False, False, S)
            check(part.sum(0).view(2 * N, K), ref, 0 if kind == "int" else 3e-6 * M ** 0.5)


def test_complex_gemm_argument_errors():
    from singa_amd import ops
    z = torch.zeros(64, 64, device=DEV)
    ok = dict(a=z.data_ptr(), lda=64, a_im=32, b=z.data_ptr(), ldb=32, b_im=32 * 32, c=z.data_ptr(), ldc=64, c_im=32, I=64, J=32, R=32,
              sigma=1.0)
    ops._cgemm([ok], True, True)
    for bad in (dict(sigma=0.5), dict(a_im=30), dict(R=30), dict(a=0), dict(ldc=62)):
        with pytest.raises(RuntimeError):
            ops._cgemm([{**ok, **bad}], True, True)
    with pytest.raises(RuntimeError):
        ops._cgemm([ok], False, True)                      # (0, 1) is not built
    with pytest.raises(RuntimeError):
        ops._cgemm([ok] * 5, True, True)                   # SINGA_CGEMM_MAX = 4 problems per launch


def test_gemm_argument_errors():
    from singa_amd import ops
    x, w = torch.zeros(8, 6, device=DEV), torch.zeros(4, 6, device=DEV)
    with pytest.raises(RuntimeError, match="multiples of 4"):
        ops.gemm_nt(x, w)                                   # K = 6 is not a multiple of 4 floats
    with pytest.raises(RuntimeError):
        ops._gemm([dict(a=0, b=0, c=0, I=1, J=1, R=4)], True, True)


def _so2_case(L, E, cin, cout, extra, seed):
    """Random m-primary edge rows + weights in the reference's parameter layout."""
    from singa_amd import so3
    lay = so3.layout(L, 2)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(E, lay.KR, cin, generator=g)                      # reduced l-primary, as the oracle takes it
    sd = {"c.fc_m0.weight": torch.randn((L + 1) * cout + extra, (L + 1) * cin, generator=g) / ((L + 1) * cin) ** 0.5,
          "c.fc_m0.bias": torch.randn((L + 1) * cout + extra, generator=g)}
    for m in (1, 2):
        n = L - m + 1
        sd[f"c.so2_m_conv.{m - 1}.fc.weight"] = torch.randn(2 * cout * n, cin * n, generator=g) / (cin * n) ** 0.5
    return lay, x, sd


@pytest.mark.parametrize("L,cin,cout,extra", [(2, 32, 128, 352), (4, 32, 128, 352), (4, 128, 112, 0), (6, 128, 112, 0)])
def test_so2_convolution_matches_oracle(L, cin, cout, extra):
    """ops.so2_linear3 on the own MFMA kernel (and on the BLAS libraries) vs oracle.so2_conv: outputs per m, d input,
    d weights, d bias.  E = 777 edges: ragged row tiles; the weight-gradient reduction is forced into several splits."""
    from singa_amd import ops
    from singa_amd.model.EF_layers import SO2_m_Convolution
    E = 777
    lay, x, sd = _so2_case(L, E, cin, cout, extra, seed=L * 10 + cin)
    fr = O.Frame(torch.eye(3).repeat(E, 1, 1), L, 2)
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xo = x.clone().requires_grad_(True)
    want, ex = O.so2_conv(sdo, "c", xo, None, fr, cout, extra)       # [E, KR, cout] reduced l-primary (+ extra)
    gy = torch.randn(want.shape, generator=torch.Generator().manual_seed(1))
    gex = torch.randn(E, extra, generator=torch.Generator().manual_seed(2)) if extra else None
    ((want * gy).sum() + ((ex * gex).sum() if extra else 0.0)).backward()
    to_m = torch.as_tensor(lay.to_m)
    st = lay.seg_start
    from tests.lib_gemm import _SO2Linear3Lib
    for own in ("3m", True, False):      # the product's path (k7c on the fc weights), the block-weight form on k7, on the BLAS library
        ops._GEMM_SPLIT_ROWS = 200
        so2 = ops.so2_linear3 if own is True else _SO2Linear3Lib.apply
        try:
            X = x[:, to_m].reshape(E, -1).to(DEV).requires_grad_(True)       # m-primary rows, what k4 hands over
            w0 = sd["c.fc_m0.weight"].to(DEV).requires_grad_(True)
            b0 = sd["c.fc_m0.bias"].to(DEV).requires_grad_(True)
            fcs = []
            for m in (1, 2):
                mod = SO2_m_Convolution(m, cin, cout, [L], [2], device=DEV)
                with torch.no_grad():
                    mod.fc.weight.copy_(sd[f"c.so2_m_conv.{m - 1}.fc.weight"])
                fcs.append(mod)
            if own == "3m":
                h0, h1, h2 = ops.so2_conv3m(X, w0, b0, fcs[0].fc.weight, fcs[1].fc.weight, st[1] * cin, (st[2] - st[1]) * cin)
            else:
                h0, h1, h2 = so2(X, w0, b0, fcs[0].block_weight(), fcs[1].block_weight(), st[1] * cin, (st[2] - st[1]) * cin)
            got = torch.cat([h0[:, extra:], h1, h2], 1).reshape(E, lay.KR, cout)        # m-primary rows
            assert rel_err(got.detach().cpu(), want[:, to_m]) < 2e-6
            if extra:
                assert rel_err(h0[:, :extra].detach().cpu(), ex) < 2e-6
            loss = (got * gy[:, to_m].to(DEV)).sum() + ((h0[:, :extra] * gex.to(DEV)).sum() if extra else 0.0)
            loss.backward()
            assert rel_err(X.grad.cpu().reshape(E, lay.KR, cin), xo.grad[:, to_m]) < 5e-6
            assert rel_err(w0.grad.cpu(), sdo["c.fc_m0.weight"].grad) < 1e-5
            assert rel_err(b0.grad.cpu(), sdo["c.fc_m0.bias"].grad) < 1e-5
            for m in (1, 2):
                assert rel_err(fcs[m - 1].fc.weight.grad.cpu(), sdo[f"c.so2_m_conv.{m - 1}.fc.weight"].grad) < 1e-5, (own, m)
        finally:
            ops._GEMM_SPLIT_ROWS = 2048


@pytest.mark.parametrize("L,cin,cout", [(2, 16, 512), (2, 512, 16), (4, 16, 512), (4, 512, 16), (4, 112, 16), (6, 16, 512),
                                        (6, 512, 16)])
def test_so3_linear_matches_oracle(L, cin, cout):
    """ops.so3_linear vs oracle.so3_linear (EF:655-671): output, d input, d weight per degree, d bias; N = 333 nodes (ragged
    tiles, several reduction splits / partial rows).  Four evaluations: the k11s kernels for the 16 <-> 512 shapes (matrix-core and lane-broadcast variants),
    the MFMA kernel k11 (one grouped-row problem per degree), the BLAS libraries."""
    from singa_amd import ops
    N, K = 333, (L + 1) ** 2
    g = torch.Generator().manual_seed(L + cin)
    x = torch.randn(N, K, cin, generator=g)
    sd = {"p.weight": torch.randn(L + 1, cout, cin, generator=g) / cin ** 0.5, "p.bias": torch.randn(cout, generator=g)}
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xo = x.clone().requires_grad_(True)
    want = O.so3_linear(sdo, "p", xo, L)
    gy = torch.randn(want.shape, generator=g)
    (want * gy).sum().backward()
    from tests.lib_gemm import _SO3LinearLib
    from singa_amd import _lib
    for own, skinny, valu in ((True, True, 0), (True, True, 1), (True, False, 0), (False, False, 0)):
        ops._GEMM_SPLIT_ROWS, ops.USE_SKINNY_SO3 = 500, skinny
        _lib.lib().singa_so3_skinny_variant(valu)
        so3 = ops.so3_linear if own else _SO3LinearLib.apply
        try:
            xd = x.to(DEV).requires_grad_(True)
            w, b = sd["p.weight"].to(DEV).requires_grad_(True), sd["p.bias"].to(DEV).requires_grad_(True)
            got = so3(xd, w, b, L)
            assert rel_err(got.detach().cpu(), want) < 2e-6
            (got * gy.to(DEV)).sum().backward()
            assert rel_err(xd.grad.cpu(), xo.grad) < 5e-6
            assert rel_err(w.grad.cpu(), sdo["p.weight"].grad) < 1e-5
            assert rel_err(b.grad.cpu(), sdo["p.bias"].grad) < 1e-5
        finally:
            ops._GEMM_SPLIT_ROWS, ops.USE_SKINNY_SO3 = 2048, True
            _lib.lib().singa_so3_skinny_variant(0)


def test_so2_and_so3_linear_on_empty_inputs():
    """An edge type without edges / a node type without nodes (possible in real batches): empty outputs, zero weight
    gradients, no launch."""
    from singa_amd import ops
    X = torch.zeros(0, 96, device=DEV, requires_grad=True)
    w0, b0 = torch.randn(40, 48, device=DEV, requires_grad=True), torch.randn(40, device=DEV, requires_grad=True)
    w1, w2 = torch.randn(32, 32, device=DEV, requires_grad=True), torch.randn(16, 16, device=DEV, requires_grad=True)
    h0, h1, h2 = ops.so2_linear3(X, w0, b0, w1, w2, 48, 32)
    assert h0.shape == (0, 40) and h1.shape == (0, 32) and h2.shape == (0, 16)
    (h0.sum() + h1.sum() + h2.sum()).backward()
    assert float(w0.grad.abs().max()) == 0.0 and float(w2.grad.abs().max()) == 0.0 and float(b0.grad.abs().max()) == 0.0
    for t in (w0, b0):
        t.grad = None
    f1, f2 = torch.randn(32, 16, device=DEV, requires_grad=True), torch.randn(16, 8, device=DEV, requires_grad=True)
    h0, h1, h2 = ops.so2_conv3m(X, w0, b0, f1, f2, 48, 32)
    assert h0.shape == (0, 40) and h1.shape == (0, 32) and h2.shape == (0, 16)
    (h0.sum() + h1.sum() + h2.sum()).backward()
    assert float(w0.grad.abs().max()) == 0.0 and float(f1.grad.abs().max()) == 0.0 and float(f2.grad.abs().max()) == 0.0
    x = torch.zeros(0, 9, 16, device=DEV, requires_grad=True)
    w, b = torch.randn(3, 32, 16, device=DEV, requires_grad=True), torch.randn(32, device=DEV, requires_grad=True)
    y = ops.so3_linear(x, w, b, 2)
    assert y.shape == (0, 9, 32)
    y.sum().backward()
    assert float(w.grad.abs().max()) == 0.0


@pytest.mark.parametrize("tile_shape", [0, 3], indirect=True)
def test_gemm_epilogue_options(tile_shape):
    """bias + addend + ReLU, and the positive-mask (ReLU gradient) epilogue, on ragged tiles; exact on integers."""
    from singa_amd import ops
    g = torch.Generator().manual_seed(5)
    M, N, K = 333, 200, 72
    ri = lambda *s: torch.randint(-3, 4, s, generator=g).float()
    x, w, b, ad, mk = ri(M, K), ri(N, K), ri(N), ri(M, N), ri(M, N)
    xd, wd, bd, add, mkd = (t.to(DEV) for t in (x, w, b, ad, mk))
    y = torch.empty(M, N, device=DEV)
    ops._gemm([dict(a=xd.data_ptr(), lda=K, b=wd.data_ptr(), ldb=K, c=y.data_ptr(), ldc=N, bias=bd.data_ptr(),
                    addend=add.data_ptr(), relu=1, I=M, J=N, R=K)], True, True)
    assert torch.equal(y.cpu(), torch.relu(x @ w.t() + b + ad))
    dy = ri(M, N).to(DEV)
    dx = torch.empty(M, K, device=DEV)
    hm = ri(M, K).to(DEV)                                     # "forward output" whose sign gates the gradient
    ops._gemm([dict(a=dy.data_ptr(), lda=N, b=wd.data_ptr(), ldb=K, c=dx.data_ptr(), ldc=K, I=M, J=K, R=N, mask=hm.data_ptr())],
              True, False)
    assert torch.equal(dx.cpu(), (dy.cpu() @ w) * (hm.cpu() > 0))


@pytest.mark.parametrize("M,K,N", [(3417, 256, 256), (700, 256, 1024), (129, 64, 32), (45, 400, 256), (2000, 8, 256), (640, 256, 116)])
def test_linear_and_pos_ffn_match_torch(M, K, N):
    """ops.linear / linear_add (own GEMM with the bias and the addend in the epilogue) and ops.pos_ffn (CP:161-191: Linear,
    ReLU, Linear) against plain torch in float64, outputs and all gradients - the CProMG transformer's projections."""
    from singa_amd import ops
    g = torch.Generator().manual_seed(M + N)
    x, w, b, ad = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    gy = torch.randn(M, N, generator=g)
    ref_in = [t.double().requires_grad_(True) for t in (x, w, b, ad)]
    ref = ref_in[0] @ ref_in[1].t() + ref_in[2] + ref_in[3]
    (ref * gy.double()).sum().backward()
    got_in = [t.to(DEV).requires_grad_(True) for t in (x, w, b, ad)]
    got = ops.linear_add(got_in[0], got_in[1], got_in[2], got_in[3])
    assert rel_err(got.detach().cpu(), ref.detach()) < 2e-6
    (got * gy.to(DEV)).sum().backward()
    for a, r in zip(got_in, ref_in):
        assert rel_err(a.grad.cpu(), r.grad) < 1e-5
    if K == 256 and N % 4 == 0:
        H = 1024 if N == 256 else 64
        w1, b1 = torch.randn(H, K, 1, generator=g) / K ** 0.5, torch.randn(H, generator=g)         # 1x1 Conv1d weights as they are
        w2, b2 = torch.randn(N, H, 1, generator=g) / H ** 0.5, torch.randn(N, generator=g)
        rin = [t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
        ref = torch.relu(rin[0] @ rin[1][:, :, 0].t() + rin[2]) @ rin[3][:, :, 0].t() + rin[4]
        (ref * gy.double()).sum().backward()
        gin = [t.to(DEV).requires_grad_(True) for t in (x, w1, b1, w2, b2)]
        got = ops.pos_ffn(*gin)
        assert rel_err(got.detach().cpu(), ref.detach()) < 3e-6
        (got * gy.to(DEV)).sum().backward()
        for a, r in zip(gin, rin):
            assert rel_err(a.grad.cpu(), r.grad) < 2e-5


@pytest.mark.parametrize("N", [1, 333, 6499])
def test_grouped_linear3_matches_torch(N):
    """The graph attention's three grouped 1x1 Conv1d layers (k_lin, q_lin, v_lin: CP:27-29, 55-57) as one 12-problem launch
    against torch's grouped conv1d in float64: outputs, the (chained) input gradient, all weight gradients."""
    from singa_amd import ops
    heads, ig = 4, 64
    g = torch.Generator().manual_seed(N)
    h = torch.randn(N, heads * ig, generator=g)
    ws = [torch.randn(heads * og, ig, 1, generator=g) / ig ** 0.5 for og in (32, 32, 64)]
    gys = [torch.randn(N, heads, og, generator=g) for og in (32, 32, 64)]
    hr = h.double().requires_grad_(True)
    wr = [w.double().requires_grad_(True) for w in ws]
    loss = 0.0
    refs = []
    for w, gy in zip(wr, gys):
        y = torch.nn.functional.conv1d(hr.t().unsqueeze(0), w, groups=heads)[0].t().reshape(N, heads, -1)
        refs.append(y)
        loss = loss + (y * gy.double()).sum()
    loss.backward()
    hd = h.to(DEV).requires_grad_(True)
    wd = [w.to(DEV).requires_grad_(True) for w in ws]
    outs = ops.grouped_linear3(hd, *wd, heads)
    for o, r in zip(outs, refs):
        assert o.shape == r.shape and rel_err(o.detach().cpu(), r.detach()) < 2e-6
    sum((o * gy.to(DEV)).sum() for o, gy in zip(outs, gys)).backward()
    assert rel_err(hd.grad.cpu(), hr.grad) < 1e-5
    for a, r in zip(wd, wr):
        assert rel_err(a.grad.cpu(), r.grad) < 1e-5


@pytest.mark.parametrize("L,cin,cout", [(4, 512, 16), (4, 112, 16), (6, 512, 16)])
def test_so3_linear_with_residual(L, cin, cout):
    """The residual of TransBlockV2 (EF:1383-1384, 1405-1406) added in the SO3 linear's own launch = the plain sum."""
    from singa_amd import ops
    N, K = 257, (L + 1) ** 2
    g = torch.Generator().manual_seed(L * 100 + cin)
    x = torch.randn(N, K, cin, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(L + 1, cout, cin, generator=g) / cin ** 0.5).to(DEV).requires_grad_(True)
    b = torch.randn(cout, generator=g).to(DEV).requires_grad_(True)
    r = torch.randn(N, K, cout, generator=g).to(DEV).requires_grad_(True)
    gy = torch.randn(N, K, cout, generator=g).to(DEV)
    want = ops.so3_linear(x, w, b, L) + r
    grads_want = torch.autograd.grad(want, [x, w, b, r], gy)
    got = ops.so3_linear(x, w, b, L, r)
    assert rel_err(got.detach().cpu(), want.detach().cpu()) < 1e-6
    grads_got = torch.autograd.grad(got, [x, w, b, r], gy)
    for a, c in zip(grads_got, grads_want):
        assert rel_err(a.cpu(), c.cpu()) < 1e-6
