"""Stand-in for e3nn 0.5.1 (environment.yml:90). Only `o3` is used by the reference (model/EF_layers.py:19-20)."""
from . import o3  # noqa: F401
