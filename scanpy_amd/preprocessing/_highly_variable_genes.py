"""`sc.pp.highly_variable_genes` (flavors 'seurat' and 'cell_ranger') on MI355X.

Reference: src/scanpy/preprocessing/_highly_variable_genes.py:630-880 (dispatch / write-back), :367-553 (single
batch), :566-627 (batches).  The passes over the matrix -- per-gene sum / sum of squares / number of expressing cells
of expm1(X) ('seurat') or X ('cell_ranger') -- are ONE device sweep (`scamd_pp_col_stats_f32`, row-masked per batch);
what follows works on g-sized vectors on the host exactly as the reference does (pandas `cut`, per-bin statistics)."""
from __future__ import annotations

import warnings

import numpy as np
import pandas as pd

from .._anndata import is_anndata
from . import _csr_device
from ._pca import _get_arr

_MAD_C = 0.6744897501960817  # Phi^-1(3/4): statsmodels.robust.mad's constant (`_highly_variable_genes.py:506-513`)
_DEFAULTS = dict(min_disp=0.5, max_disp=np.inf, min_mean=0.0125, max_mean=3)


class _Cutoffs:
    """`_highly_variable_genes.py:317-358`"""

    def __init__(self, min_disp, max_disp, min_mean, max_mean):
        self.min_disp, self.max_disp, self.min_mean, self.max_mean = min_disp, max_disp, min_mean, max_mean

    @classmethod
    def validate(cls, *, n_top_genes, min_disp, max_disp, min_mean, max_mean):
        if n_top_genes is None:
            return cls(min_disp, max_disp, min_mean, max_mean)
        if dict(min_disp=min_disp, max_disp=max_disp, min_mean=min_mean, max_mean=max_mean) != _DEFAULTS:
            warnings.warn("If you pass `n_top_genes`, all cutoffs are ignored.", UserWarning, stacklevel=3)
        return n_top_genes

    def in_bounds(self, mean, dispersion_norm):
        return ((mean > self.min_mean) & (mean < self.max_mean) & (dispersion_norm > self.min_disp)
                & (dispersion_norm < self.max_disp))


def _mad(a) -> float:
    a = np.asarray(a, dtype=np.float64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)
        return float(np.median(np.abs(a - np.median(a))) / _MAD_C)


def _nth_highest(x: np.ndarray, n: int) -> float:
    """`:540-553`"""
    x = x[~np.isnan(x)]
    if n > x.size:
        warnings.warn(f"`n_top_genes` (={n}) > number of normalized dispersions (={x.size}), "
                      "returning all genes with normalized dispersions.", UserWarning, stacklevel=4)
        n = x.size
    return np.sort(x)[::-1][n - 1]


class _StreamedColStats:
    """`col_stats` of a backed CSR matrix (`_backed.BackedCsr`): row chunks are read, uploaded and swept one after the
    other by the real backend; sums, sums of squares and positive counts add up."""

    def __init__(self, be, x, step: int = 1_000_000):
        self.be, self.x, self.step = be, x, step

    def _chunk(self, c):
        """upload one row chunk and run the matrix' pending transforms (normalize_total / log1p) on it"""
        from .._backed import apply_ops_pp

        rows = c.load()
        m = self.be.upload(rows.to_scipy())
        apply_ops_pp(self.be, m, rows.ops)
        return m

    def col_stats(self, m, *, row_mask=None, expm1_scale=None, count_positive: bool = False):
        tot = None
        for c in self.x.row_chunks(self.step):
            if row_mask is not None and not row_mask[c.i0:c.i1].any():
                continue
            part = self.be.col_stats(self._chunk(c),
                                     row_mask=None if row_mask is None else row_mask[c.i0:c.i1],
                                     expm1_scale=expm1_scale, count_positive=count_positive)
            tot = part if tot is None else tuple(None if a is None else a + b for a, b in zip(tot, part))
        if tot is None:
            g = self.x.shape[1]
            tot = (np.zeros(g), np.zeros(g), np.zeros(g, dtype=np.int64) if count_positive else None)
        return tot

    def clip_col_sums(self, m, clip_val, *, row_mask=None):
        g = self.x.shape[1]
        tot = (np.zeros(g), np.zeros(g))
        for c in self.x.row_chunks(self.step):
            if row_mask is not None and not row_mask[c.i0:c.i1].any():
                continue
            part = self.be.clip_col_sums(self._chunk(c), clip_val,
                                         row_mask=None if row_mask is None else row_mask[c.i0:c.i1])
            tot = (tot[0] + part[0], tot[1] + part[1])
        return tot

    def nonnegative_integers(self, m) -> bool:
        return all(self.be.nonnegative_integers(self._chunk(c))
                   for c in self.x.row_chunks(self.step))


def _single_batch(be, m, var_names, *, n_rows: int, row_mask, filter_unexpressed_genes: bool, cutoff, n_bins: int,
                  flavor: str, log1p_base) -> pd.DataFrame:
    """`_highly_variable_genes_single_batch` (`:367-450`) on the rows selected by `row_mask`."""
    expm1_scale = None
    if flavor == "seurat":  # counts space: expm1(x * log(base)) (`:402-410`)
        expm1_scale = 1.0 if log1p_base is None else float(np.log(log1p_base))
    # expm1(x) > 0 <=> x > 0: the count of expressing cells rides on the same sweep
    s, sq, npos = be.col_stats(m, row_mask=row_mask, expm1_scale=expm1_scale, count_positive=filter_unexpressed_genes)
    if filter_unexpressed_genes:  # filter_genes(min_cells=1) (`:387-395`)
        filt = npos >= 1
    else:
        filt = np.ones(m.shape[1], dtype=bool)
    mean, var = _csr_device.mean_var_from_sums(s[filt], sq[filt], n_rows, correction=1)
    mean[mean == 0] = 1e-12
    dispersion = var / mean
    if flavor == "seurat":
        dispersion[dispersion == 0] = np.nan
        with np.errstate(invalid="ignore", divide="ignore"):
            dispersion = np.log(dispersion)
        mean = np.log1p(mean)
    df = pd.DataFrame({"means": mean, "dispersions": dispersion})
    if flavor == "seurat":  # `_get_mean_bins` (`:453-467`)
        bins = n_bins
    else:
        bins = np.r_[-np.inf, np.percentile(df["means"], np.arange(10, 105, 5)), np.inf]
    rv = pd.cut(df["means"], bins=bins)
    # like the reference (`:464-467`): interval categories as strings; the column is part of the `inplace=False` result
    df["mean_bin"] = rv.cat.set_categories(rv.cat.categories.astype("string"), rename=True)
    # per-bin statistics of the dispersions through pandas, like the reference (`_get_disp_stats`, `:470-482`): ties at
    # the `n_top_genes` cut-off (two-gene bins give +-1/sqrt(2) exactly) then break the same way
    grouped = df.groupby("mean_bin", observed=True)["dispersions"]
    if flavor == "seurat":
        stats = grouped.agg(avg="mean", dev="std")
        one_gene = stats["dev"].isna()  # a single gene in the bin: normalised dispersion 1 (`:485-503`)
        stats.loc[one_gene, "dev"] = stats.loc[one_gene, "avg"]
        stats.loc[one_gene, "avg"] = 0
    else:
        stats = grouped.agg(avg="median", dev=_mad)
    per_gene = stats.loc[df["mean_bin"]].set_index(df.index)
    df["dispersions_norm"] = (df["dispersions"] - per_gene["avg"]) / per_gene["dev"]
    dn = df["dispersions_norm"].to_numpy()
    if isinstance(cutoff, _Cutoffs):  # `_subset_genes` (`:515-537`)
        hv = cutoff.in_bounds(mean, np.nan_to_num(dn))
    else:
        n_top = min(int(cutoff), int(filt.sum()))
        hv = np.nan_to_num(dn, nan=-np.inf) >= _nth_highest(dn.copy(), n_top)
    df["highly_variable"] = hv
    df.index = var_names[filt]
    n_removed = int((~filt).sum())
    if n_removed:
        missing = pd.DataFrame(np.zeros((n_removed, len(df.columns))), columns=df.columns)
        missing["highly_variable"] = missing["highly_variable"].astype(bool)
        missing.index = var_names[~filt]
        df = pd.concat([df, missing]).loc[var_names]
    return df


def _seurat_v3(adata, *, flavor: str, layer, n_top_genes: int, batch_key, check_values: bool, span: float,
               subset: bool, inplace: bool):
    """`_highly_variable_genes_seurat_v3` (`:118-316`): variance of each gene after standardising with a LOESS trend of
    log10(variance) on log10(mean) (per batch) and clipping at sqrt(n) standard deviations.  Three device sweeps per
    batch -- per-gene sums (mean, variance), the clipped sums -- and a LOESS on the host (`_loess.py`, R's dloess)."""
    from ._loess import loess_fit

    x = _get_arr(adata, layer=layer)
    be = _csr_device.default_backend()
    if getattr(x, "is_backed", False):  # every sweep of this flavor is a sum over cells: streamed like `col_stats`
        be, m = _StreamedColStats(be, x), x
    else:
        m = be.upload(x)
    n_obs, n_vars = adata.n_obs, adata.n_vars
    if check_values and not be.nonnegative_integers(m):
        warnings.warn(f"`flavor={flavor!r}` expects raw count data, but non-integers were found.", UserWarning,
                      stacklevel=3)
    s, sq, _ = be.col_stats(m)
    means, variances = _csr_device.mean_var_from_sums(s, sq, n_obs, correction=1)
    if batch_key is None:
        batch_codes, batches = np.zeros(n_obs, dtype=np.int64), [0]
    else:
        col = adata.obs[batch_key]
        batch_codes, uniq = pd.factorize(col.to_numpy(), sort=True)  # `np.unique(batch_info)` order (`:196`)
        batches = list(range(len(uniq)))
    norm_gene_vars = []
    for b in batches:
        mask = None if batch_key is None else batch_codes == b
        n_b = n_obs if mask is None else int(mask.sum())
        if mask is None:
            mean, var = means, variances
        else:
            sb, sqb, _ = be.col_stats(m, row_mask=mask)
            mean, var = _csr_device.mean_var_from_sums(sb, sqb, n_b, correction=1)
        estimat_var = np.zeros(n_vars, dtype=np.float64)
        not_const = var > 0
        if not_const.any():
            estimat_var[not_const] = loess_fit(np.log10(mean[not_const]), np.log10(var[not_const]), span=span, degree=2)
        reg_std = np.sqrt(10 ** estimat_var)
        clip_val = reg_std * np.sqrt(n_b) + mean  # clip large values as in Seurat (`:231-233`)
        sq_sum, c_sum = be.clip_col_sums(m, clip_val, row_mask=mask)
        with np.errstate(divide="ignore", invalid="ignore"):
            norm_gene_vars.append((1 / ((n_b - 1) * np.square(reg_std)))
                                  * ((n_b * np.square(mean)) + sq_sum - 2 * c_sum * mean))
    norm_gene_vars = np.stack(norm_gene_vars, axis=0)
    ranked = np.argsort(np.argsort(-norm_gene_vars, axis=1), axis=1).astype(np.float32)  # small rank = most variable
    num_batches_high_var = np.sum((ranked < n_top_genes).astype(int), axis=0)
    ranked[ranked >= n_top_genes] = np.nan
    median_ranked = np.ma.median(np.ma.masked_invalid(ranked), axis=0).filled(np.nan)
    df = pd.DataFrame(index=adata.var_names)
    df["means"], df["variances"] = means, variances
    df = df.assign(gene_name=df.index, highly_variable_nbatches=num_batches_high_var,
                   highly_variable_rank=median_ranked, variances_norm=np.mean(norm_gene_vars, axis=0))
    if flavor == "seurat_v3":
        sort_cols, ascending = ["highly_variable_rank", "highly_variable_nbatches"], [True, False]
    else:
        sort_cols, ascending = ["highly_variable_nbatches", "highly_variable_rank"], [False, True]
    sorted_index = df[sort_cols].sort_values(sort_cols, ascending=ascending, na_position="last").index
    df["highly_variable"] = False
    df.loc[sorted_index[:int(n_top_genes)], "highly_variable"] = True
    if not inplace:
        if batch_key is None:
            df = df.drop(["highly_variable_nbatches"], axis=1)
        if subset:
            df = df.iloc[df["highly_variable"].to_numpy(), :]
        return df
    adata.uns["hvg"] = {"flavor": flavor}
    for key in ("highly_variable", "highly_variable_rank", "means", "variances"):
        adata.var[key] = df[key].to_numpy()
    adata.var["variances_norm"] = df["variances_norm"].to_numpy().astype("float64", copy=False)
    if batch_key is not None:
        adata.var["highly_variable_nbatches"] = df["highly_variable_nbatches"].to_numpy()
    if subset:
        adata._inplace_subset_var(df["highly_variable"].to_numpy())
    return None


def highly_variable_genes(  # noqa: PLR0913
    adata,
    *,
    layer: str | None = None,
    n_top_genes: int | None = None,
    min_disp: float = 0.5,
    max_disp: float = np.inf,
    min_mean: float = 0.0125,
    max_mean: float = 3,
    span: float = 0.3,
    n_bins: int = 20,
    flavor: str = "seurat",
    subset: bool = False,
    inplace: bool = True,
    batch_key: str | None = None,
    filter_unexpressed_genes: bool | None = None,
    check_values: bool = True,
):
    """Annotate highly variable genes (drop-in for `scanpy.pp.highly_variable_genes`, `:630`).

    Expects logarithmised data, except `flavor='seurat_v3'` / `'seurat_v3_paper'`, which expect counts; their LOESS
    (scikit-misc in the reference) is `_loess.py`, pinned to Seurat's own numbers."""
    if not is_anndata(adata):
        msg = ("`pp.highly_variable_genes` expects an `AnnData` argument, "
               "pass `inplace=False` if you want to return a `pd.DataFrame`.")
        raise ValueError(msg)
    if flavor in {"seurat_v3", "seurat_v3_paper"}:
        return _seurat_v3(adata, flavor=flavor, layer=layer, n_top_genes=2000 if n_top_genes is None else n_top_genes,
                          batch_key=batch_key, check_values=check_values, span=span, subset=subset, inplace=inplace)
    if flavor not in {"seurat", "cell_ranger"}:
        raise ValueError('`flavor` needs to be "seurat" or "cell_ranger"')
    cutoff = _Cutoffs.validate(n_top_genes=n_top_genes, min_disp=min_disp, max_disp=max_disp, min_mean=min_mean,
                               max_mean=max_mean)
    x = _get_arr(adata, layer=layer)
    be = _csr_device.default_backend()
    if getattr(x, "is_backed", False):
        # an on-disk matrix: the per-gene sums are additive over row chunks, so the one sweep this function needs is
        # streamed (float64 partial sums added on the host) and the matrix never has to fit anywhere
        be, m = _StreamedColStats(be, x), x
    else:
        m = be.upload(x)
    var_names = adata.var_names
    base = adata.uns.get("log1p", {}).get("base")
    kw = dict(cutoff=cutoff, n_bins=n_bins, flavor=flavor, log1p_base=base)
    if not batch_key:
        df = _single_batch(be, m, var_names, n_rows=adata.n_obs, row_mask=None,
                           filter_unexpressed_genes=bool(filter_unexpressed_genes), **kw)
    else:
        if filter_unexpressed_genes is False:
            warnings.warn(f"filter_unexpressed_genes is set to False, but will ignored for batch-aware {flavor=!r} HVG "
                          "computation", UserWarning, stacklevel=2)
        col = adata.obs[batch_key]
        if not isinstance(col.dtype, pd.CategoricalDtype):
            col = col.astype("category")
        batches = col.cat.categories
        dfs = []
        for b in batches:  # `_highly_variable_genes_batched` (`:566-627`)
            mask = (col == b).to_numpy()
            dfs.append(_single_batch(be, m, var_names, n_rows=int(mask.sum()), row_mask=mask,
                                     filter_unexpressed_genes=True, **kw))
        df = pd.concat(dfs, axis=0)
        df["highly_variable"] = df["highly_variable"].astype(int)
        df = df.groupby(df.index, observed=True).agg(dict(means="mean", dispersions="mean", dispersions_norm="mean",
                                                          highly_variable="sum"))
        df["highly_variable_nbatches"] = df["highly_variable"]
        df["highly_variable_intersection"] = df["highly_variable_nbatches"] == len(batches)
        if isinstance(cutoff, int):
            df = df.sort_values(["highly_variable_nbatches", "dispersions_norm"], ascending=False, na_position="last")
            df["highly_variable"] = np.arange(df.shape[0]) < cutoff
            df = df.loc[var_names]
        else:
            df = df.loc[var_names]
            df["dispersions_norm"] = df["dispersions_norm"].fillna(0)
            df["highly_variable"] = cutoff.in_bounds(df["means"], df["dispersions_norm"])
    if not inplace:
        if subset:
            df = df.loc[df["highly_variable"]]
        return df
    adata.uns["hvg"] = {"flavor": flavor}
    adata.var["highly_variable"] = df["highly_variable"]
    adata.var["means"] = df["means"]
    adata.var["dispersions"] = df["dispersions"]
    adata.var["dispersions_norm"] = df["dispersions_norm"].astype(np.float32)
    if batch_key is not None:
        adata.var["highly_variable_nbatches"] = df["highly_variable_nbatches"]
        adata.var["highly_variable_intersection"] = df["highly_variable_intersection"]
    if subset:
        adata._inplace_subset_var(df["highly_variable"].to_numpy())
    return None
