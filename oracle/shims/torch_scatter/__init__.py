"""Stand-in for torch_scatter 2.1.1: `scatter_sum`, `scatter_softmax` (reference model/CProMG.py:15,66,74)."""
import torch


def scatter_sum(src, index, dim=0, dim_size=None):
    assert dim == 0
    n = int(index.max()) + 1 if dim_size is None else dim_size
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    return out.index_add_(0, index, src)


def scatter_softmax(src, index, dim=0):
    assert dim == 0
    n = int(index.max()) + 1
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    mx = torch.full((n,) + tuple(src.shape[1:]), float("-inf"), dtype=src.dtype, device=src.device)
    mx = mx.scatter_reduce(0, idx, src.detach(), reduce="amax", include_self=True)
    ex = (src - mx[index]).exp()
    den = torch.zeros_like(mx).index_add_(0, index, ex)
    return ex / den[index]
