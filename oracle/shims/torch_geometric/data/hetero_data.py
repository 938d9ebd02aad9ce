import torch

from .storage import BaseStorage, EdgeStorage, NodeStorage


def _to(v, device):
    if torch.is_tensor(v):
        return v.to(device)
    if isinstance(v, dict):
        return {k: _to(x, device) for k, x in v.items()}
    return v


class HeteroData:
    def __init__(self):
        self._global_store = BaseStorage()
        self._node_store_dict = {}
        self._edge_store_dict = {}

    def __getitem__(self, key):
        if isinstance(key, tuple):
            return self._edge_store_dict[key]
        if key in self._node_store_dict:
            return self._node_store_dict[key]
        return self._global_store[key]

    def to(self, device):
        for st in [self._global_store, *self._node_store_dict.values(), *self._edge_store_dict.values()]:
            st._mapping = {k: _to(v, device) for k, v in st._mapping.items()}
        return self


class Data:
    def __init__(self, **kw):
        self._store = dict(kw)

    def __getattr__(self, k):
        try:
            return self.__dict__["_store"][k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __getitem__(self, k):
        return self._store[k]

    def node_attrs(self):
        n = self._store["x"].size(0)
        return [k for k, v in self._store.items() if torch.is_tensor(v) and k != "edge_index" and "edge" not in k and v.size(0) == n]

    def edge_attrs(self):
        return [k for k in self._store if "edge" in k]


def _collate_value(vals):
    v0 = vals[0]
    if torch.is_tensor(v0):
        return torch.cat(vals, 0)
    if isinstance(v0, (float, int)):
        return torch.tensor(vals)
    if isinstance(v0, dict):
        return {k: _collate_value([v[k] for v in vals]) for k in v0}
    return list(vals)


class Batch(HeteroData):
    """PyG collate of HeteroData: concat node tensors, offset edge_index, add `ptr`/`batch`."""

    @classmethod
    def from_data_list(cls, data_list):
        out = cls()
        offs = {}
        for nt in data_list[0]._node_store_dict:
            sizes = [d._node_store_dict[nt]["x"].size(0) for d in data_list]
            ptr = torch.tensor([0] + sizes).cumsum(0)
            offs[nt] = ptr
            st = NodeStorage({k: _collate_value([d._node_store_dict[nt][k] for d in data_list])
                              for k in data_list[0]._node_store_dict[nt].keys()})
            st["ptr"] = ptr
            st["batch"] = torch.repeat_interleave(torch.arange(len(data_list)), torch.tensor(sizes))
            out._node_store_dict[nt] = st
        for et in data_list[0]._edge_store_dict:
            m = {}
            for k in data_list[0]._edge_store_dict[et].keys():
                vals = [d._edge_store_dict[et][k] for d in data_list]
                if k == "edge_index":
                    vals = [v + torch.stack([offs[et[0]][i], offs[et[2]][i]]).view(2, 1) for i, v in enumerate(vals)]
                    m[k] = torch.cat(vals, 1)
                else:
                    m[k] = _collate_value(vals)
            out._edge_store_dict[et] = EdgeStorage(m)
        out._global_store = BaseStorage({k: _collate_value([d._global_store[k] for d in data_list])
                                         for k in data_list[0]._global_store.keys()})
        return out
