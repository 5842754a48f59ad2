"""scanpy_amd -- MI355X-native `sc.pp.pca -> sc.pp.neighbors -> sc.tl.leiden`.

    import scanpy_amd as sc
    sc.pp.pca(adata); sc.pp.neighbors(adata); sc.tl.leiden(adata)

Same function signatures and AnnData slots as scverse/scanpy for that path; the arithmetic runs in
hand-written HIP kernels for gfx950 behind the C ABI of include/scanpy_amd.h.  There is no CPU
fallback: without the built library or without a GPU the calls raise.
"""
from . import metrics
from . import preprocessing as pp
from . import tools as tl
from ._anndata import AnnData
from ._settings import settings
from .neighbors import MI355XKNNTransformer, Neighbors
from .readwrite import read, read_10x_h5, read_10x_mtx, read_h5ad, read_zarr, write, write_h5ad, write_zarr

__all__ = ["pp", "tl", "metrics", "AnnData", "settings", "Neighbors", "MI355XKNNTransformer", "read_zarr",
           "write_zarr", "read_h5ad", "read_10x_h5", "write_h5ad", "read", "write", "read_10x_mtx"]
__version__ = "0.1.0"
