"""TEST INFRASTRUCTURE — CPU restatement (numpy/scipy) of the reference's upstream normalisation chain
`normalize_total -> log1p -> highly_variable_genes -> scale`.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this module; nothing under
`scanpy_amd/` does.  Every function cites the reference lines it follows (paths relative to /root/reference).

Pinning (tests/test_oracle_preprocess.py):
  * normalize_total: the doctest values of `_normalization.py:218-253`, `tests/test_normalization.py:29-30, 62-74,
    336-353` (X_total / X_frac / zero-count cells);
  * normalize_total + log1p + highly_variable_genes (seurat, cell_ranger): the Seurat / Cell Ranger goldens
    `tests/_scripts/seurat_hvg.csv`, `tests/_scripts/cell_ranger_hvg.csv` on bundled pbmc68k_reduced
    (`tests/test_highly_variable_genes.py:367-422`, tolerance 2e-5) -> `tests/golden/hvg_golden.npz`;
  * scale: the literal arrays of `tests/test_scaling.py:13-72`.
`statsmodels` (used by the reference for the MAD, `_highly_variable_genes.py:506-513`) is not installed here: its
`robust.mad` default (median absolute deviation about the median / Phi^-1(3/4)) is restated.
"""
from __future__ import annotations

import warnings

import numpy as np
import pandas as pd
from scipy import sparse

_MAD_C = 0.6744897501960817  # scipy.stats.norm.ppf(0.75): statsmodels.robust.mad's normalisation constant


# --------------------------------------------------------------------------------------------------
# normalize_total  (src/scanpy/preprocessing/_normalization.py)
# --------------------------------------------------------------------------------------------------
def nnz_median(counts: np.ndarray) -> float:
    """`_compute_nnz_median` (`_normalization.py:20-26`): median of the non-zero counts."""
    return float(np.median(counts[counts > 0]))


def normalize_total(x, *, target_sum=None, exclude_highly_expressed=False, max_fraction=0.05):
    """`_normalize_total_helper` (`_normalization.py:69-124`) + the CSR kernel `_normalize_csr` (`:29-66`).

    Returns (X_normalised, norm_factor = counts_per_cell / target_sum, counts_per_cols or None).  Integer input is
    promoted to float32 (`normalize_total`, `:271-272`).  A cell with zero counts is divided by 1
    (`axis_mul_or_truediv(..., allow_divide_by_zero=False)`, `_utils/__init__.py:638-639`)."""
    if not 0 <= max_fraction <= 1:
        raise ValueError("Choose max_fraction between 0 and 1.")  # `:260-262`
    issp = sparse.issparse(x)
    x = x.tocsr().copy() if issp else np.array(x, copy=True)
    if np.issubdtype(x.dtype, np.integer):
        x = x.astype(np.float32)
    counts_per_cols = None
    if issp:
        # row sums accumulate in float64 and are stored in the matrix dtype (`:40-46`: `count = 0.0`)
        counts = np.asarray(x.astype(np.float64).sum(axis=1)).ravel().astype(x.dtype)
        if exclude_highly_expressed:
            rows = np.repeat(np.arange(x.shape[0]), np.diff(x.indptr))
            hi = x.data > max_fraction * counts[rows]  # `:53`
            counts_per_cols = np.bincount(x.indices[hi], minlength=x.shape[1]).astype(np.int32)
            keep = counts_per_cols[x.indices] == 0  # `:60-65`
            counts = np.bincount(rows[keep], weights=x.data[keep].astype(np.float64), minlength=x.shape[0]).astype(x.dtype)
    else:
        counts = x.sum(axis=1)
        if exclude_highly_expressed:  # `:108-113`
            hi = x > counts[:, None] * max_fraction
            subset = hi.sum(axis=0) == 0
            counts_per_cols = (~subset).astype(np.int32)
            counts = x[:, subset].sum(axis=1)
    if target_sum is None:
        target_sum = nnz_median(counts)
    factor = counts / target_sum
    div = factor + (factor == 0)
    if issp:
        x.data = x.data / np.repeat(div, np.diff(x.indptr))
    else:
        x = x / div[:, None]
    return x, factor, counts_per_cols


# --------------------------------------------------------------------------------------------------
# log1p  (src/scanpy/preprocessing/_simple.py:310-423)
# --------------------------------------------------------------------------------------------------
def log1p(x, *, base=None):
    """`log1p_array` / `log1p_sparse` (`_simple.py:359-379`): natural log of 1 + x, divided by log(base)."""
    issp = sparse.issparse(x)
    x = x.copy()
    data = x.data if issp else x
    if not np.issubdtype(data.dtype, np.floating):
        data = data.astype(float)
    data = np.log1p(data)
    if base is not None:
        data = data / np.log(base)
    if issp:
        x.data = data
        return x
    return data


# --------------------------------------------------------------------------------------------------
# highly_variable_genes, flavors 'seurat' and 'cell_ranger'
# (src/scanpy/preprocessing/_highly_variable_genes.py:367-553)
# --------------------------------------------------------------------------------------------------
def mean_var(x, *, correction=1):
    """`fast_array_utils.stats.mean_var(x, axis=0, correction=1)` as called at `_highly_variable_genes.py:413` and
    `_scale.py:189`: float64 mean and `(E[x^2] - E[x]^2) * n / (n - correction)`."""
    n = x.shape[0]
    if sparse.issparse(x):
        x64 = x.astype(np.float64)
        mean = np.asarray(x64.sum(axis=0)).ravel() / n
        mean_sq = np.asarray(x64.multiply(x64).sum(axis=0)).ravel() / n
    else:
        x64 = np.asarray(x, dtype=np.float64)
        mean = x64.sum(axis=0) / n
        mean_sq = (x64 * x64).sum(axis=0) / n
    var = mean_sq - mean * mean
    if correction and n > correction:
        var = var * (n / (n - correction))
    return mean, var


def mad(a) -> float:
    """statsmodels.robust.mad(a) (default c, center=median), used at `_highly_variable_genes.py:506-513`."""
    a = np.asarray(a, dtype=np.float64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)
        return float(np.median(np.abs(a - np.median(a))) / _MAD_C)


def highly_variable_genes(x, *, flavor="seurat", n_top_genes=None, min_disp=0.5, max_disp=np.inf, min_mean=0.0125,
                          max_mean=3.0, n_bins=20, log1p_base=None) -> pd.DataFrame:
    """`_highly_variable_genes_single_batch` (`:367-450`) with `filter_unexpressed_genes=False`.

    x: log1p-transformed expression (cells x genes).  Returns a DataFrame with `means`, `dispersions`,
    `dispersions_norm` (float64) and `highly_variable`."""
    if flavor not in ("seurat", "cell_ranger"):
        raise ValueError('`flavor` needs to be "seurat" or "cell_ranger"')
    if flavor == "seurat":  # `:402-410`: back to counts space
        x = x.copy()
        if log1p_base is not None:
            x = x * np.log(log1p_base)
        if sparse.issparse(x):
            x = x.expm1()
        else:
            x = np.expm1(x)
    mean, var = mean_var(x, correction=1)
    mean[mean == 0] = 1e-12  # `:415`
    dispersion = var / mean
    if flavor == "seurat":  # `:417-420`
        dispersion[dispersion == 0] = np.nan
        dispersion = np.log(dispersion)
        mean = np.log1p(mean)
    df = pd.DataFrame({"means": mean, "dispersions": dispersion})
    # `_get_mean_bins` (`:453-467`)
    if flavor == "seurat":
        bins = n_bins
    else:
        bins = np.r_[-np.inf, np.percentile(df["means"], np.arange(10, 105, 5)), np.inf]
    df["mean_bin"] = pd.cut(df["means"], bins=bins)
    # `_get_disp_stats` (`:470-482`)
    grouped = df.groupby("mean_bin", observed=True)["dispersions"]
    if flavor == "seurat":
        stats = grouped.agg(avg="mean", dev="std")
        one_gene = stats["dev"].isna()  # `_postprocess_dispersions_seurat` (`:485-503`)
        stats.loc[one_gene, "dev"] = stats.loc[one_gene, "avg"]
        stats.loc[one_gene, "avg"] = 0
    else:
        stats = grouped.agg(avg="median", dev=mad)
    per_gene = stats.loc[df["mean_bin"]].set_index(df.index)
    df["dispersions_norm"] = (df["dispersions"] - per_gene["avg"]) / per_gene["dev"]
    dn = df["dispersions_norm"].to_numpy()
    # `_subset_genes` (`:515-537`)
    if n_top_genes is None:
        dn0 = np.nan_to_num(dn)
        hv = (mean > min_mean) & (mean < max_mean) & (dn0 > min_disp) & (dn0 < max_disp)
    else:
        n_top = min(int(n_top_genes), x.shape[1])
        finite = dn[~np.isnan(dn)]
        n_top = min(n_top, finite.size)
        cut = np.sort(finite)[::-1][n_top - 1]
        hv = np.nan_to_num(dn, nan=-np.inf) >= cut
    df["highly_variable"] = hv
    return df.drop(columns="mean_bin")


# --------------------------------------------------------------------------------------------------
# scale  (src/scanpy/preprocessing/_scale.py:137-282)
# --------------------------------------------------------------------------------------------------
def highly_variable_genes_seurat_v3(x, *, n_top_genes=2000, batch=None, span=0.3, flavor="seurat_v3"):
    """`_highly_variable_genes_seurat_v3` (src/scanpy/preprocessing/_highly_variable_genes.py:118-316) on a dense or
    sparse count matrix -> DataFrame (means, variances, variances_norm, highly_variable_rank,
    highly_variable_nbatches, highly_variable); the LOESS is oracle/loess.py."""
    from .loess import loess

    xd = np.asarray(x.toarray() if sparse.issparse(x) else x, dtype=np.float64)
    n, g = xd.shape
    means, variances = xd.mean(axis=0), xd.var(axis=0, ddof=1)
    batch = np.zeros(n, dtype=int) if batch is None else np.asarray(batch)
    norm = []
    for b in np.unique(batch):
        xb = xd[batch == b]
        nb = xb.shape[0]
        mean, var = xb.mean(axis=0), xb.var(axis=0, ddof=1)
        est = np.zeros(g)
        ok = var > 0
        if ok.any():
            est[ok] = loess(np.log10(mean[ok]), np.log10(var[ok]), span=span, degree=2)
        reg_std = np.sqrt(10 ** est)
        clip = reg_std * np.sqrt(nb) + mean
        stored = xb != 0  # the reference sums over the STORED values of a sparse batch (`:75-115`)
        clipped = np.where(stored, np.minimum(xb, clip[None, :]), 0.0)
        with np.errstate(divide="ignore", invalid="ignore"):
            norm.append((1 / ((nb - 1) * reg_std ** 2)) * (nb * mean ** 2 + (clipped ** 2).sum(axis=0)
                                                             - 2 * clipped.sum(axis=0) * mean))
    norm = np.stack(norm)
    ranked = np.argsort(np.argsort(-norm, axis=1), axis=1).astype(np.float32)
    nb_high = (ranked < n_top_genes).sum(axis=0)
    ranked[ranked >= n_top_genes] = np.nan
    med = np.ma.median(np.ma.masked_invalid(ranked), axis=0).filled(np.nan)
    df = pd.DataFrame({"means": means, "variances": variances, "variances_norm": norm.mean(axis=0),
                       "highly_variable_rank": med, "highly_variable_nbatches": nb_high})
    keys, asc = (["highly_variable_rank", "highly_variable_nbatches"], [True, False]) if flavor == "seurat_v3" \
        else (["highly_variable_nbatches", "highly_variable_rank"], [False, True])
    top = df[keys].sort_values(keys, ascending=asc, na_position="last").index[:n_top_genes]
    df["highly_variable"] = False
    df.loc[top, "highly_variable"] = True
    return df


def scale(x, *, zero_center=True, max_value=None, mask_obs=None):
    """`scale_array` (`:153-229`) / `scale_array_masked` (`:232-277`).  Returns (X_scaled, mean, std).

    zero_center=True densifies a sparse input (`:194-201`; float64 result because the float64 mean is subtracted);
    zero_center=False keeps the sparsity and divides the stored values (`scale_and_clip_csr`, `:280-295`, which clips
    only from above).  `std == 0` is replaced by 1 (`:191`)."""
    issp = sparse.issparse(x)
    x = x.tocsr().copy() if issp else np.array(x, copy=True)
    if np.issubdtype(x.dtype, np.integer):
        x = x.astype(np.float64)  # `:176-181`
    rows = np.arange(x.shape[0]) if mask_obs is None else np.flatnonzero(np.asarray(mask_obs, dtype=bool))
    sub = x[rows]
    mean, var = mean_var(sub, correction=1)
    std = np.sqrt(var)
    std[std == 0] = 1
    if zero_center:
        dense = np.asarray(sub.todense()) if issp else sub
        dense = (dense - mean) / std
        if max_value is not None:
            dense = np.clip(dense, -max_value, max_value)  # `clip_array` (`:53-69`)
        if mask_obs is None:
            return dense, mean, std
        out = np.asarray(x.todense(), dtype=dense.dtype) if issp else x.astype(dense.dtype)
        out[rows] = dense
        return out, mean, std
    if issp:
        sel = np.zeros(x.shape[0], dtype=bool)
        sel[rows] = True
        erow = np.repeat(sel, np.diff(x.indptr))
        vals = x.data[erow] / std[x.indices[erow]]
        if max_value is not None:
            vals = np.minimum(max_value, vals)
        x.data[erow] = vals.astype(x.dtype)
        return x, mean, std
    scaled = sub / std
    if max_value is not None:
        scaled = np.minimum(scaled, max_value)  # zero_center=False: upper clip only (`:58-60`)
    x[rows] = scaled
    return x, mean, std
