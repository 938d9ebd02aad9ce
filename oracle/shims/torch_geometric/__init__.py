"""Stand-in for torch_geometric 2.3.1 (environment.yml:173): only what the reference's hot path touches.

Call sites: `pyg.utils.softmax` (model/EF_layers.py:1180); `knn_graph`, `to_undirected`, `get_laplacian`,
`to_dense_batch` (model/CProMG.py:293-298,264,304-306,330-340); `Data`, `HeteroData`, `Batch`, `DataLoader`
type names (model/Embedding.py:7-10, model/CProMG.py:12). Un-pickling of example/*.pt needs the
`torch_geometric.data.{hetero_data,storage,graph_store,feature_store}` class names.
Restated from the packages' documentation; unpinned by any reference test.
"""
from . import data, loader, nn, utils  # noqa: F401
