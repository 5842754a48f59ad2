from ..neighbors import neighbors
from ._filter import filter_cells, filter_genes
from ._highly_variable_genes import highly_variable_genes
from ._normalization import normalize_total
from ._pca import pca
from ._scale import scale
from ._simple import log1p

__all__ = ["pca", "neighbors", "filter_cells", "filter_genes", "normalize_total", "log1p", "highly_variable_genes", "scale"]
