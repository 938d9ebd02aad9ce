"""TEST-ONLY second opinion for the product's own MFMA GEMM (k7 / k11): the same contractions evaluated by the BLAS
libraries through torch (hipBLASLt / rocBLAS).  Moved out of singa_amd/ops.py: the product has one path."""
import torch

from singa_amd.ops import _degree_index, _degree_onehot, colsum


class _blas:                       # (kept as a no-op context: the library choice no longer matters for a cross-check)
    def __init__(self, *a):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _splitk_tn(a, b):
    """a^T @ b through the BLAS library."""
    return a.t() @ b


class _SO2Linear3Lib(torch.autograd.Function):
    """The same three GEMMs through the BLAS libraries (hipBLASLt / rocBLAS via torch): the cross-check of k7."""

    @staticmethod
    def forward(ctx, X, w0, b0, w1, w2, n0, n1):
        blocks = (X[:, :n0], X[:, n0:n0 + n1], X[:, n0 + n1:])
        ctx.save_for_backward(X, w0, w1, w2)
        ctx.n0, ctx.n1 = n0, n1
        outs = []
        for xb, w, b in zip(blocks, (w0, w1, w2), (b0, None, None)):
            with _blas(xb.shape[0], w.shape[0], xb.shape[1]):
                outs.append(torch.addmm(b, xb, w.t()) if b is not None else xb @ w.t())
        return tuple(outs)

    @staticmethod
    def backward(ctx, g0, g1, g2):
        X, w0, w1, w2 = ctx.saved_tensors
        n0, n1 = ctx.n0, ctx.n1
        bounds = ((0, n0), (n0, n0 + n1), (n0 + n1, X.shape[1]))
        gX = torch.empty_like(X) if ctx.needs_input_grad[0] else None
        gws = []
        for g, w, (a, b) in zip((g0, g1, g2), (w0, w1, w2), bounds):
            g = g.contiguous()
            if gX is not None:
                with _blas(g.shape[0], w.shape[1], g.shape[1]):
                    torch.mm(g, w, out=gX[:, a:b])
            gws.append(_splitk_tn(g, X[:, a:b]))
        return gX, gws[0], colsum(g0), gws[1], gws[2], None, None


class _SO3LinearLib(torch.autograd.Function):
    """SO3_LinearV2 as one batched library GEMM over the K rows (the cross-check of k11)."""

    @staticmethod
    def forward(ctx, x, weight, bias, L):
        x = x.contiguous()
        N, K, _ = x.shape
        w = weight.index_select(0, _degree_index(L, x.device))                    # [K, out, in]
        out = torch.empty(N, K, weight.shape[1], device=x.device, dtype=x.dtype)
        torch.bmm(x.transpose(0, 1), w.transpose(1, 2), out=out.transpose(0, 1))
        out[:, 0, :] += bias
        ctx.save_for_backward(x, w)
        ctx.L = L
        return out

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        K = x.shape[1]
        gT = g.transpose(0, 1)                                                    # [K, N, out] view
        gx = torch.empty_like(x)
        torch.bmm(gT, w, out=gx.transpose(0, 1))
        gw_rows = torch.bmm(gT.transpose(1, 2), x.transpose(0, 1))               # [K, out, in]
        gw = (_degree_onehot(ctx.L, x.device) @ gw_rows.view(K, -1)).view(ctx.L + 1, *gw_rows.shape[1:])
        return gx, gw, colsum(g[:, 0, :]), None


