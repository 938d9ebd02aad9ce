from .hetero_data import HeteroData, Data, Batch  # noqa: F401
