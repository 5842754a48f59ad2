"""`sc.pp.pca` on MI355X: same signature, defaults, errors and AnnData write-back as the reference
(src/scanpy/preprocessing/_pca/__init__.py:53-384); the arithmetic runs in `_pca_solver.pca_fit`."""
from __future__ import annotations

import warnings

import numpy as np
from scipy import sparse

from .._anndata import AnnData, is_anndata
from .._backed import is_backed
from .._settings import settings
from .._utils import _UNSET, as_csr_f32, resolve_seed

def torch_free_bytes() -> int:
    """free HBM on the current device (0 when there is none: the CPU stand-in of the tests streams every pass)"""
    import torch

    if not torch.cuda.is_available():
        return 0
    return int(torch.cuda.mem_get_info()[0])


_DEFAULT_MASK = object()  # Default("adata.var.get('highly_variable')"), _pca/__init__.py:65-67


def _check_mask(adata, mask, dim: str):
    """src/scanpy/get/get.py:607-665 (the slice the path uses)."""
    if mask is None:
        return None
    if isinstance(mask, str):
        annot = adata.var if dim == "var" else adata.obs
        if mask not in annot.columns:
            msg = f"Did not find `adata.{dim}[{mask!r}]`. Either add the mask first to `adata.{dim}`or consider using the mask argument with an array."
            raise ValueError(msg)
        mask = annot[mask].to_numpy()
    else:
        mask = np.asarray(mask)
        n = adata.n_vars if dim == "var" else adata.n_obs
        if len(mask) != n:
            raise ValueError("The shape of the mask do not match the data.")
    if mask.dtype != bool:
        raise ValueError("Mask array must be boolean.")
    return mask


def _get_arr(adata, *, layer=None, obsm=None):
    """src/scanpy/get/get.py:505-570 (X / layers / obsm)."""
    if layer is not None and obsm is not None:
        raise ValueError("Only one of `layer` or `obsm` can be specified.")
    if layer is not None:
        return adata.layers[layer]
    if obsm is not None:
        return adata.obsm[obsm]
    return adata.X


def pca(  # noqa: PLR0912, PLR0913, PLR0915
    data,
    n_comps: int | None = None,
    *,
    layer: str | None = None,
    obsm: str | None = None,
    zero_center: bool = True,
    svd_solver: str | None = None,
    chunked: bool = False,
    chunk_size: int | None = None,
    rng=None,
    random_state=_UNSET,
    return_info: bool = False,
    mask_var=_DEFAULT_MASK,
    dtype="float32",
    key_added: str | None = None,
    copy: bool = False,
):
    """Principal component analysis (drop-in for `scanpy.pp.pca`, src/scanpy/preprocessing/_pca/__init__.py:53).

    Differences from the reference are confined to HOW the decomposition is computed:
    `svd_solver` None/'arpack' -> block-Krylov solver converged to ARPACK-level accuracy,
    'randomized' -> randomized subspace iteration, 'covariance_eigh' -> exact covariance + eigh.
    `chunked=True` streams row chunks of `chunk_size` cells through the device; the result is exact (not
    IncrementalPCA's approximation) and identical to the one-shot fit.
    """
    from ._pca_solver import GpuBackend, pca_fit

    seed, _meta = resolve_seed(rng, random_state)
    if chunked and not zero_center:  # the reference's chunked path is IncrementalPCA: centred only (`:262-266`)
        raise ValueError("chunked PCA centres the data: pass zero_center=True (the default) with chunked=True")
    if return_anndata := is_anndata(data):
        adata = data.copy() if copy else data
    else:
        adata = AnnData(data)

    # mask handling: _pca/__init__.py:221-232
    if mask_var is _DEFAULT_MASK:
        mask_var = "highly_variable" if "highly_variable" in adata.var.columns else None
    elif mask_var is not None and obsm is not None:
        msg = "Argument `mask_var` is incompatible with `obsm`."
        raise ValueError(msg)
    mask_var_param, mask = mask_var, _check_mask(adata, mask_var, "var")

    x = _get_arr(adata, layer=layer, obsm=obsm)
    if obsm is not None:
        # the reference subsets the AnnData by var and then reads `adata_comp.obsm[obsm]` unmasked
        # (_pca/__init__.py:228-232): the var mask is recorded in `params` but never slices an obsm matrix
        mask = None
    if mask is not None:
        x = x[:, mask]
    n_obs, n_vars = x.shape
    backed = is_backed(x)  # an on-disk CSR matrix (`read_zarr(..., backed='r')`): streamed through the device by rows
    if backed and not zero_center:
        raise ValueError("a backed matrix is streamed through the chunked PCA, which centres the data: pass "
                         "zero_center=True (the default)")
    chunked = chunked or backed

    if n_comps is None:  # _pca/__init__.py:234-236
        min_dim = min(n_vars, n_obs)
        n_comps = min_dim - 1 if min_dim <= settings.N_PCS else settings.N_PCS

    is_sparse = sparse.issparse(x) or backed
    if svd_solver in {"auto", "randomized"} and not is_sparse:
        pass  # reference only logs a reproducibility note here (_pca/__init__.py:207-212)
    if svd_solver is None:
        svd_solver = "arpack"  # _handle_sklearn_args default for PCA / TruncatedSVD (_pca/__init__.py:439-442)
    elif svd_solver in {"auto", "full", "tsqr"}:
        warnings.warn(f"Ignoring svd_solver={svd_solver!r} and using arpack-accuracy block Krylov on the GPU.",
                      UserWarning, stacklevel=2)
        svd_solver = "arpack"
    elif svd_solver == "randomized" and is_sparse and zero_center:
        # the reference rejects 'randomized' for sparse input with a warning and falls back to arpack
        # (tests/test_pca.py:236-261); mirror that.
        warnings.warn("Ignoring svd_solver='randomized' and using arpack, sparse PCA with sklearn < 1.4 only supports ['lobpcg', 'arpack'].",
                      UserWarning, stacklevel=2)
        svd_solver = "arpack"
    elif svd_solver == "lobpcg":
        warnings.warn("svd_solver='lobpcg' for sparse relies on legacy code and will not be supported in the future. "
                      "Also the lobpcg solver has been observed to be inaccurate. Please use 'arpack' instead.",
                      FutureWarning, stacklevel=2)
        svd_solver = "arpack"
    elif svd_solver not in {"arpack", "randomized", "covariance_eigh"}:
        raise ValueError(f"svd_solver={svd_solver!r} is not supported")

    backend = GpuBackend()
    if chunked:
        # Rows are streamed through the device `chunk_size` at a time (the reference runs sklearn's IncrementalPCA over
        # `adata.chunked_X(chunk_size)`, `_pca/__init__.py:262-285`, an approximation).  Here the fixed-point Gram
        # matrix is additive over row chunks, so the chunked fit is EXACT and bitwise equal to the one-shot fit; HBM
        # holds two chunks at a time unless everything fits, in which case the chunks stay resident between passes.
        from ._pca_solver import CsrRowsView, _ChunkedRows

        step = int(chunk_size) if chunk_size is not None else 1_000_000
        if step < 1:
            raise ValueError("chunk_size must be a positive number of observations")
        import os

        # SCAMD_PCA_CHUNK_RESIDENT=0 forces the streaming mode (every pass uploads again) whatever the free memory
        budget = 0 if os.environ.get("SCAMD_PCA_CHUNK_RESIDENT") == "0" else int(0.4 * torch_free_bytes())
        if backed:  # lazy row chunks: read + decompressed by a reader thread one chunk ahead of the device
            host_chunks = x.row_chunks(step)
        else:
            xc = as_csr_f32(x)
            host_chunks = [CsrRowsView(xc, i, min(i + step, n_obs)) for i in range(0, n_obs, step)]
        rows = _ChunkedRows(host_chunks, n_vars, resident_budget_bytes=budget)
        res = pca_fit(rows, n_comps, backend=backend, zero_center=zero_center, svd_solver=svd_solver, seed=seed)
    else:
        xc = as_csr_f32(x)
        import os

        # (SCAMD_PCA_OVERLAP_MIN_NNZ: stored entries from which the overlapped upload is used -- default 16M, below that the
        # whole matrix is on the device in a millisecond; the tests set 0; SCAMD_PCA_OVERLAP_UPLOAD=0 switches it off)
        if xc.nnz >= int(os.environ.get("SCAMD_PCA_OVERLAP_MIN_NNZ", 16 << 20)) and os.environ.get("SCAMD_PCA_OVERLAP_UPLOAD") != "0":
            # a large host matrix: the value array goes up first (max|x| fixes the fixed-point scale), the column indices follow
            # in row chunks UNDER the Gram kernel of the chunk before (same bits as the one-shot fit: integer sums)
            from ._pca_solver import _HostCsrOverlapped

            handle = _HostCsrOverlapped(xc, int(os.environ.get("SCAMD_PCA_OVERLAP_CHUNKS", 6)))
        else:
            handle = backend.upload(xc)
        res = pca_fit(handle, n_comps, backend=backend, zero_center=zero_center, svd_solver=svd_solver, seed=seed)
    from .._device import to_host

    x_pca = to_host(res.scores)
    in_dtype = x.dtype if np.issubdtype(x.dtype, np.floating) else np.dtype("float64")
    components = res.components.astype(in_dtype, copy=False)
    variance = res.explained_variance.astype(in_dtype, copy=False)
    variance_ratio = res.explained_variance_ratio.astype(in_dtype, copy=False)
    if x_pca.dtype.descr != np.dtype(dtype).descr:
        x_pca = x_pca.astype(dtype)

    if return_anndata:
        k_obsm, k_varm, k_uns = ("X_pca", "PCs", "pca") if key_added is None else (key_added,) * 3
        adata.obsm[k_obsm] = x_pca
        if obsm:
            pass
        elif mask is not None:
            pcs = np.zeros(shape=(adata.n_vars, n_comps))
            pcs[mask] = components.T
            adata.varm[k_varm] = pcs
        else:
            adata.varm[k_varm] = components.T
        adata.uns[k_uns] = dict(
            params=dict(
                zero_center=zero_center,
                mask_var=mask_var_param,
                **(dict(layer=layer) if layer is not None else {}),
                **(dict(obsm=obsm) if obsm is not None else {}),
            ),
            variance=variance,
            variance_ratio=variance_ratio,
            **(dict(components=components.T) if obsm is not None else {}),
        )
        return adata if copy else None
    if return_info:
        return x_pca, components, variance_ratio, variance
    return x_pca
