from ._leiden import leiden
from ._umap import umap

__all__ = ["leiden", "umap"]
