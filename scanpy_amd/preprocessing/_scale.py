"""`sc.pp.scale` on MI355X (reference: src/scanpy/preprocessing/_scale.py:72-330).

Per-gene mean / variance: one device sweep (`scamd_pp_col_stats_f32`, masked rows); the scaling itself is
`scamd_pp_scale_dense_f32` (zero_center=True: dense output, as the reference densifies) or `scamd_pp_scale_csr_f32`
(zero_center=False: the sparsity is kept, values are clipped from above only)."""
from __future__ import annotations

import warnings

import numpy as np
from scipy import sparse

from .._anndata import is_anndata
from .._utils import view_to_actual
from . import _csr_device
from ._normalization import _set_obs_rep
from ._pca import _check_mask, _get_arr


def _scale_matrix(x, *, zero_center: bool, max_value, mask_obs):
    """`scale_array` / `scale_array_masked` (`_scale.py:153-277`) -> (X_scaled, mean, std)."""
    be = _csr_device.default_backend()
    issp = sparse.issparse(x)
    if mask_obs is not None:
        mask_obs = np.asarray(mask_obs)
        if mask_obs.dtype != bool or mask_obs.shape != (x.shape[0],):
            raise ValueError("`mask_obs` must be a boolean vector with one entry per observation")
    m = be.upload(_csr_device.in_memory(x))
    n_rows = x.shape[0] if mask_obs is None else int(mask_obs.sum())
    s, sq, _ = be.col_stats(m, row_mask=mask_obs)
    mean, var = _csr_device.mean_var_from_sums(s, sq, n_rows, correction=1)
    std = np.sqrt(np.maximum(var, 0.0))
    std[std == 0] = 1
    if zero_center:
        if issp:
            warnings.warn("zero-centering a sparse array/matrix densifies it.", UserWarning, stacklevel=3)
        # numpy semantics of the reference: a float32 ndarray stays float32, everything else becomes float64
        out_f64 = issp or np.asarray(x).dtype != np.float32
        out = be.scale_dense(m, mean, std, max_value=max_value, row_mask=mask_obs, out_f64=out_f64)
        return out, mean, std
    be.scale_csr_(m, std, max_value=max_value, row_mask=mask_obs)
    return be.download(m), mean, std


def scale(data, *, zero_center: bool = True, max_value: float | None = None, copy: bool = False,
          layer: str | None = None, obsm: str | None = None, mask_obs=None):
    """Scale data to unit variance and zero mean (drop-in for `scanpy.pp.scale`, `_scale.py:72`).

    AnnData: X / layer / obsm replaced, `var['mean']`, `var['std']` written (`:298-330`); array or sparse matrix:
    the scaled matrix is returned (`copy` is implied: device results are new host arrays)."""
    if not is_anndata(data):
        if layer is not None:
            raise ValueError(f"`layer` argument inappropriate for value of type {type(data)}")
        if obsm is not None:
            raise ValueError(f"`obsm` argument inappropriate for value of type {type(data)}")
        if isinstance(mask_obs, str):
            raise ValueError("Cannot refer to mask with string without providing anndata object as argument")
        return _scale_matrix(data, zero_center=zero_center, max_value=max_value, mask_obs=mask_obs)[0]
    adata = data.copy() if copy else data
    str_mean_std = ("mean", "std")
    if mask_obs is not None:
        if isinstance(mask_obs, str):
            str_mean_std = (f"mean of {mask_obs}", f"std of {mask_obs}")
        else:
            str_mean_std = ("mean with mask", "std with mask")
        mask_obs = _check_mask(adata, mask_obs, "obs")
    view_to_actual(adata)  # `_scale.py:315`
    x = _get_arr(adata, layer=layer, obsm=obsm)
    out, mean, std = _scale_matrix(x, zero_center=zero_center, max_value=max_value, mask_obs=mask_obs)
    adata.var[str_mean_std[0]] = mean
    adata.var[str_mean_std[1]] = std
    _set_obs_rep(adata, out, layer=layer, obsm=obsm)
    return adata if copy else None
