import torch


def softmax(src, index, ptr=None, num_nodes=None, dim=0):
    """Segment softmax: exp(x - max_seg) / (sum_seg + 1e-16)."""
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    mx = torch.full((n,) + tuple(src.shape[1:]), float("-inf"), dtype=src.dtype, device=src.device)
    mx = mx.scatter_reduce(0, idx, src.detach(), reduce="amax", include_self=True)
    out = (src - mx.index_select(0, index)).exp()
    den = torch.zeros_like(mx).index_add_(0, index, out) + 1e-16
    return out / den.index_select(0, index)


def coalesce(edge_index, edge_attr, num_nodes, reduce="mean"):
    key = edge_index[0] * num_nodes + edge_index[1]
    uniq, inv = torch.unique(key, sorted=True, return_inverse=True)
    ei = torch.stack([uniq // num_nodes, uniq % num_nodes], 0)
    if edge_attr is None:
        return ei, None
    out = torch.zeros((uniq.numel(),) + tuple(edge_attr.shape[1:]), dtype=edge_attr.dtype, device=edge_attr.device)
    out.index_add_(0, inv, edge_attr)
    if reduce == "mean":
        cnt = torch.zeros(uniq.numel(), dtype=edge_attr.dtype, device=edge_attr.device).index_add_(
            0, inv, torch.ones_like(inv, dtype=edge_attr.dtype))
        out = out / cnt.view(-1, *([1] * (edge_attr.dim() - 1)))
    return ei, out


def to_undirected(edge_index, edge_attr=None, num_nodes=None, reduce="add"):
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    row, col = edge_index[0], edge_index[1]
    ei = torch.stack([torch.cat([row, col]), torch.cat([col, row])], 0)
    ea = None if edge_attr is None else torch.cat([edge_attr, edge_attr], 0)
    ei, ea = coalesce(ei, ea, n, reduce)
    return (ei, ea) if edge_attr is not None else ei


def get_laplacian(edge_index, edge_weight=None, normalization=None, dtype=None, num_nodes=None):
    assert normalization is None
    keep = edge_index[0] != edge_index[1]
    edge_index = edge_index[:, keep]
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1), dtype=dtype, device=edge_index.device)
    else:
        edge_weight = edge_weight[keep]
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    deg = torch.zeros((n,) + tuple(edge_weight.shape[1:]), dtype=edge_weight.dtype,
                      device=edge_weight.device).index_add_(0, edge_index[0], edge_weight)
    loop = torch.arange(n, device=edge_index.device)
    edge_index = torch.cat([edge_index, torch.stack([loop, loop], 0)], 1)
    return edge_index, torch.cat([-edge_weight, deg], 0)


def to_dense_batch(x, batch=None, fill_value=0.0, max_num_nodes=None, batch_size=None):
    b = int(batch.max()) + 1 if batch_size is None else batch_size
    num = torch.zeros(b, dtype=torch.long, device=x.device).index_add_(0, batch, torch.ones_like(batch))
    cum = torch.cat([num.new_zeros(1), num.cumsum(0)])
    mx = int(num.max()) if max_num_nodes is None else max_num_nodes
    idx = torch.arange(batch.size(0), device=x.device) - cum[batch] + batch * mx
    out = x.new_full((b * mx,) + tuple(x.shape[1:]), fill_value)
    out[idx] = x
    mask = torch.zeros(b * mx, dtype=torch.bool, device=x.device)
    mask[idx] = True
    return out.view(b, mx, *x.shape[1:]), mask.view(b, mx)
