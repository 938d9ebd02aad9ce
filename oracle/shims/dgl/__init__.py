"""Stand-in for dgl 1.1.2 (environment.yml:88): `dgl.graph`, `dgl.lap_pe` (reference model/CProMG.py:524,569).

dgl.lap_pe(g, k): L = I - D^-1/2 A D^-1/2 with in-degrees clipped at 1, the k eigenvectors after the smallest,
each multiplied by a RANDOM sign, from a dense non-symmetric eigensolver. That output is not reproducible by
construction (random signs, arbitrary basis inside degenerate eigenspaces), so at our boundary the Laplacian PE
is an INPUT. This stand-in uses a symmetric eigensolver and a deterministic sign (first entry
of largest magnitude made positive) so golden files are reproducible; the tensors it returned are stored in the
golden files and fed to both the oracle and the HIP path.
"""
import numpy as np
import torch


class _Graph:
    def __init__(self, row, col):
        self.row, self.col = row, col
        self.ndata, self.edata = {}, {}
        self._n = int(max(row.max(), col.max())) + 1 if row.numel() else 0

    def num_nodes(self):
        n = self._n
        for v in self.ndata.values():
            n = max(n, v.shape[0])
        return n


def graph(rc):
    return _Graph(rc[0], rc[1])


def laplacian_pe_dense(row, col, n, k):
    a = np.zeros((n, n))
    a[row, col] = 1.0
    indeg = np.clip(a.sum(0), 1, None) ** -0.5
    lap = np.eye(n) - indeg[:, None] * a * indeg[None, :]
    lap = 0.5 * (lap + lap.T)
    w, v = np.linalg.eigh(lap)
    v = v[:, 1:k + 1]
    for j in range(v.shape[1]):
        i = np.argmax(np.abs(v[:, j]))
        if v[i, j] < 0:
            v[:, j] = -v[:, j]
    return v


def lap_pe(g, k, padding=False, return_eigval=False):
    n = g.num_nodes()
    pe = laplacian_pe_dense(g.row.cpu().numpy(), g.col.cpu().numpy(), n, k)
    return torch.tensor(pe, dtype=torch.float32)
