from ..neighbors import neighbors
from ._pca import pca

__all__ = ["pca", "neighbors"]
