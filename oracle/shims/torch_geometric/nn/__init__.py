import torch


def knn_graph(x, k, batch=None, loop=False, flow="source_to_target", **kw):
    """For each node its k nearest other nodes of the same graph; 'target_to_source' -> row = centre."""
    n = x.size(0)
    if batch is None:
        batch = torch.zeros(n, dtype=torch.long, device=x.device)
    rows, cols = [], []
    for b in range(int(batch.max()) + 1):
        ids = (batch == b).nonzero().view(-1)
        p = x[ids]
        d = torch.cdist(p.double(), p.double())
        kk = min(k if loop else k + 1, ids.numel())
        nb = d.topk(kk, dim=1, largest=False, sorted=True).indices
        rows.append(ids.view(-1, 1).expand(-1, kk).reshape(-1))
        cols.append(ids[nb].reshape(-1))
    centre, neigh = torch.cat(rows), torch.cat(cols)
    row, col = (neigh, centre) if flow == "source_to_target" else (centre, neigh)
    if not loop:
        keep = row != col
        row, col = row[keep], col[keep]
    return torch.stack([row, col], 0)
