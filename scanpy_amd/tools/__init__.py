from ._leiden import leiden
from ._leiden_multires import leiden_multires
from ._umap import umap

__all__ = ["leiden", "leiden_multires", "umap"]
