"""`sc.pp.log1p` on MI355X (reference: src/scanpy/preprocessing/_simple.py:310-423); the transform itself is
`scamd_pp_log1p_f32` over the stored values."""
from __future__ import annotations

import logging

from .._anndata import is_anndata
from .._utils import view_to_actual
from . import _csr_device
from ._normalization import _set_obs_rep
from ._pca import _get_arr


def _log1p_matrix(x, *, base=None):
    be = _csr_device.default_backend()
    m = be.upload(_csr_device.in_memory(x), want_csr_rows=False)  # element-wise: the storage format is kept (`log1p_sparse`, `:359-365`)
    be.log1p_(m, base)
    return be.download(m)


def log1p(data, *, base=None, copy: bool = False, chunked: bool | None = None, chunk_size: int | None = None,
          layer: str | None = None, obsm: str | None = None):
    """Logarithmize the data matrix, X = log(X + 1) (drop-in for `scanpy.pp.log1p`, `_simple.py:310`).

    AnnData: updates X / layer / obsm in place (or a copy), records `uns['log1p'] = {'base': base}` and warns when
    the data look log-transformed already (`:393-394`).  Array / sparse matrix: returns the transformed copy."""
    if base is not None and (base <= 0 or base == 1):
        raise ValueError("`base` must be positive and different from 1")
    if not is_anndata(data):
        if chunked or chunk_size is not None or layer is not None or obsm is not None:
            raise TypeError("`chunked`, `chunk_size`, `layer` and `obsm` only apply to AnnData input")  # `:354-357`
        return _log1p_matrix(data, base=base)
    adata = data
    if "log1p" in adata.uns:
        logging.getLogger("scanpy_amd").warning("adata.X seems to be already log-transformed.")  # `logg.warning`, `:393-394`
    adata = adata.copy() if copy else adata
    view_to_actual(adata)  # `_simple.py:397`
    if chunked:
        msg = "chunked log1p is not implemented on the MI355X path: the whole matrix is transformed in one device pass"
        raise NotImplementedError(msg)
    x = _get_arr(adata, layer=layer, obsm=obsm)
    if getattr(x, "is_backed", False):  # an on-disk matrix: a pending transform, applied per uploaded row chunk
        _set_obs_rep(adata, x.with_op("log1p", base), layer=layer, obsm=obsm)
    else:
        _set_obs_rep(adata, _log1p_matrix(x, base=base), layer=layer, obsm=obsm)
    adata.uns["log1p"] = {"base": base}
    return adata if copy else None
