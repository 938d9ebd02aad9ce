class BaseStorage:
    def __init__(self, mapping=None):
        self._mapping = dict(mapping or {})

    def __getitem__(self, k):
        return self._mapping[k]

    def __setitem__(self, k, v):
        self._mapping[k] = v

    def __getattr__(self, k):
        if k.startswith("_"):
            raise AttributeError(k)
        try:
            return self.__dict__["_mapping"][k]
        except KeyError as e:
            raise AttributeError(k) from e

    def get(self, k, d=None):
        return self._mapping.get(k, d)

    def items(self):
        return self._mapping.items()

    def keys(self):
        return self._mapping.keys()


class NodeStorage(BaseStorage):
    pass


class EdgeStorage(BaseStorage):
    pass
