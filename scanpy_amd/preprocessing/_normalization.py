"""`sc.pp.normalize_total` on MI355X: signature, errors, warnings and write-back of the reference
(src/scanpy/preprocessing/_normalization.py:127-327); the sweeps over the matrix are HIP kernels
(`scamd_pp_row_sums_f32`, `scamd_pp_count_high_f32`, `scamd_pp_row_divide_f32`)."""
from __future__ import annotations

import logging
import warnings

import numpy as np
from scipy import sparse

from .._anndata import is_anndata
from .._utils import view_to_actual
from . import _csr_device
from ._pca import _get_arr


_log = logging.getLogger("scanpy_amd")  # the reference logs through scanpy.logging (`logg.info`)


def _set_obs_rep(adata, val, *, layer=None, obsm=None) -> None:
    """src/scanpy/get/get.py:573-604"""
    if layer is not None:
        adata.layers[layer] = val
    elif obsm is not None:
        adata.obsm[obsm] = val
    else:
        adata.X = val


def _compute_nnz_median(counts: np.ndarray):
    """`_normalization.py:20-26`"""
    return np.median(counts[counts > 0])


def _normalize_total_backed(adata, x, be, *, target_sum, exclude_highly_expressed, key_added, layer, obsm, inplace, copy):
    """An on-disk matrix: the counts per cell are summed on the device chunk by chunk (one streamed pass; per-row sums
    do not depend on the chunking), and the division becomes a PENDING transform of the matrix, applied to every row
    chunk after its upload -- nothing is rewritten and nothing is materialised."""
    from .._backed import apply_ops_pp

    if exclude_highly_expressed:
        raise NotImplementedError("exclude_highly_expressed needs two more passes over the matrix and is not offered "
                                  "for a backed matrix: load it with `adata.X = adata.X.to_memory()`")
    parts = []
    for c in x.row_chunks(1_000_000):
        rows = c.load()
        m = be.upload(rows.to_scipy())
        apply_ops_pp(be, m, rows.ops)
        parts.append(be.row_sums(m))
    counts = np.concatenate(parts) if parts else np.zeros(0, dtype=np.float32)
    if target_sum is None:
        target_sum = _compute_nnz_median(counts)
    factor = counts / target_sum
    if not np.all(factor > 0):
        warnings.warn("Some cells have zero counts", UserWarning, stacklevel=3)
    out = x.with_op("row_divide", np.ascontiguousarray(factor, dtype=np.float32))
    dat = dict(X=out, norm_factor=factor)
    if inplace:
        if key_added is not None:
            adata.obs[key_added] = factor
        _set_obs_rep(adata, out, layer=layer, obsm=obsm)
    if copy:
        return adata
    return None if inplace else dat


def normalize_total(  # noqa: PLR0912
    adata,
    *,
    target_sum: float | None = None,
    exclude_highly_expressed: bool = False,
    max_fraction: float = 0.05,
    key_added: str | None = None,
    layer: str | None = None,
    obsm: str | None = None,
    inplace: bool = True,
    copy: bool = False,
):
    """Normalize counts per cell (drop-in for `scanpy.pp.normalize_total`, `_normalization.py:127`).

    Returns `None` (in place), the copied AnnData (`copy=True`) or `dict(X=..., norm_factor=...)`
    (`inplace=False`), exactly as the reference."""
    if not is_anndata(adata):
        raise TypeError("normalize_total expects an AnnData object")
    if copy:
        if not inplace:
            raise ValueError("`copy=True` cannot be used with `inplace=False`.")
        adata = adata.copy()
    if max_fraction < 0 or max_fraction > 1:
        raise ValueError("Choose max_fraction between 0 and 1.")
    view_to_actual(adata)  # `_normalization.py:264`
    x = _get_arr(adata, layer=layer, obsm=obsm)
    be = _csr_device.default_backend()
    if getattr(x, "is_backed", False):
        return _normalize_total_backed(adata, x, be, target_sum=target_sum,
                                       exclude_highly_expressed=exclude_highly_expressed, key_added=key_added,
                                       layer=layer, obsm=obsm, inplace=inplace, copy=copy)
    m = be.upload(_csr_device.in_memory(x))  # CSC -> CSR like the reference (`:266-267`); integers -> float32 (`:271-272`)
    counts = be.row_sums(m)
    gene_subset = None
    if exclude_highly_expressed:
        per_col = be.count_high(m, counts, max_fraction)
        gene_subset = per_col == 0
        counts = be.row_sums(m, col_skip=per_col)
    if target_sum is None:
        target_sum = _compute_nnz_median(counts)
    factor = counts / target_sum  # float32 / python float -> float32, as in the reference
    if exclude_highly_expressed:
        _log.info("The following highly-expressed genes are not considered during normalization factor computation:\n%s",
                  list(adata.var_names[~gene_subset]) if hasattr(adata, "var_names") else list(np.flatnonzero(~gene_subset)))
    be.row_divide_(m, factor)
    if not np.all(factor > 0):
        warnings.warn("Some cells have zero counts", UserWarning, stacklevel=2)
    out = be.download(m)
    if sparse.issparse(x) and x.format != "csr":
        out = out.tocsr()
    dat = dict(X=out, norm_factor=factor)
    if inplace:
        if key_added is not None:
            adata.obs[key_added] = dat["norm_factor"]
        _set_obs_rep(adata, dat["X"], layer=layer, obsm=obsm)
    if copy:
        return adata
    if not inplace:
        return dat
    return None
