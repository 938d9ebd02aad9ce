"""YAML -> attribute-dict configuration (what the reference gets from EasyDict, utils/misc.py:137-146)."""
import copy
import os

import yaml

DEFAULT_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config", "train.yml")


class Config(dict):
    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, Config(v) if isinstance(v, dict) and not isinstance(v, Config) else v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def to_dict(self):
        """Plain nested dicts/lists (what a checkpoint stores, so that `torch.load(weights_only=True)` accepts it)."""
        def plain(v):
            if isinstance(v, dict):
                return {k: plain(x) for k, x in v.items()}
            return [plain(x) for x in v] if isinstance(v, (list, tuple)) else v
        return plain(self)

    def __deepcopy__(self, memo):
        return Config({k: copy.deepcopy(v, memo) for k, v in self.items()})


def load_config(path=None, lmax=None):
    with open(path or DEFAULT_PATH) as f:
        cfg = Config(yaml.safe_load(f))
    if lmax is not None:
        cfg.embedding.lmax_list = [int(lmax)]
    L = int(cfg.embedding.lmax_list[0])
    cfg.model.featurizer_feat_dim = (L + 1) ** 2 * int(cfg.embedding.sphere_channels)
    return cfg
