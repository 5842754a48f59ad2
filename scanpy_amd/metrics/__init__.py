"""`sc.metrics.modularity` on MI355X (src/scanpy/metrics/_metrics.py:125-214)."""
from ._metrics import modularity, modularity_adata

__all__ = ["modularity", "modularity_adata"]
