"""Stand-in for torch_cluster (imported, never called directly: reference model/CProMG.py:16)."""
