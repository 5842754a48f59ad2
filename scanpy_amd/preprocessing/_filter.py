"""`sc.pp.filter_cells` / `sc.pp.filter_genes` on MI355X (reference: src/scanpy/preprocessing/_simple.py:53-307).

The per-cell / per-gene numbers (`n_counts`, `n_genes`, `n_cells`) are one device sweep each (`scamd_pp_row_sums_f32`,
`scamd_pp_row_count_positive_f32`, `scamd_pp_col_stats_f32`); thresholding, logging and the in-place subsetting of the
AnnData are host work exactly as in the reference."""
from __future__ import annotations

import logging

import numpy as np

from .._anndata import is_anndata
from . import _csr_device

_log = logging.getLogger("scanpy_amd")


def _one_option(**options) -> None:
    if sum(v is not None for v in options.values()) != 1:
        names = ", ".join(f"`{k}`" for k in options)
        raise ValueError(f"Provide exactly one of the optional parameters {names} per call.")


def _subset(number, lo, hi):
    keep = np.ones(number.shape[0], dtype=bool)
    if lo is not None:
        keep = number >= lo
    if hi is not None:
        keep = number <= hi
    return keep


def filter_cells(data, *, min_counts=None, min_genes=None, max_counts=None, max_genes=None, inplace: bool = True,
                 copy: bool = False):
    """Filter cell outliers based on counts and numbers of genes expressed (drop-in for `scanpy.pp.filter_cells`,
    `_simple.py:53-196`).  AnnData: annotates `obs['n_counts']` / `obs['n_genes']` and subsets in place; matrix: returns
    `(cell_subset, number_per_cell)`."""
    if copy:
        _log.warning("`copy` is deprecated, use `inplace` instead.")
    _one_option(min_counts=min_counts, min_genes=min_genes, max_counts=max_counts, max_genes=max_genes)
    if is_anndata(data):
        adata = data.copy() if copy else data
        cell_subset, number = filter_cells(adata.X, min_counts=min_counts, min_genes=min_genes, max_counts=max_counts,
                                           max_genes=max_genes)
        if not inplace:
            return cell_subset, number
        adata.obs["n_counts" if min_genes is None and max_genes is None else "n_genes"] = number
        adata._inplace_subset_obs(cell_subset)
        return adata if copy else None
    be = _csr_device.default_backend()
    m = be.upload(_csr_device.in_memory(data))
    by_counts = min_genes is None and max_genes is None
    number = be.row_sums(m) if by_counts else be.row_count_positive(m)
    cell_subset = _subset(number, min_counts if by_counts else min_genes, max_counts if by_counts else max_genes)
    s = int((~cell_subset).sum())
    if s > 0:
        what = (f"less than {min_counts} counts" if min_counts is not None else
                f"less than {min_genes} genes expressed" if min_genes is not None else
                f"more than {max_counts} counts" if max_counts is not None else f"more than {max_genes} genes expressed")
        _log.info("filtered out %d cells that have %s", s, what)
    return cell_subset, number


def filter_genes(data, *, min_counts=None, min_cells=None, max_counts=None, max_cells=None, inplace: bool = True,
                 copy: bool = False):
    """Filter genes based on number of cells or counts (drop-in for `scanpy.pp.filter_genes`, `_simple.py:199-307`).
    AnnData: annotates `var['n_counts']` / `var['n_cells']` and subsets in place; matrix: returns
    `(gene_subset, number_per_gene)`."""
    if copy:
        _log.warning("`copy` is deprecated, use `inplace` instead.")
    _one_option(min_counts=min_counts, min_cells=min_cells, max_counts=max_counts, max_cells=max_cells)
    if is_anndata(data):
        adata = data.copy() if copy else data
        gene_subset, number = filter_genes(adata.X, min_counts=min_counts, min_cells=min_cells, max_counts=max_counts,
                                           max_cells=max_cells)
        if not inplace:
            return gene_subset, number
        adata.var["n_counts" if min_cells is None and max_cells is None else "n_cells"] = number
        adata._inplace_subset_var(gene_subset)
        return adata if copy else None
    be = _csr_device.default_backend()
    m = be.upload(_csr_device.in_memory(data))
    by_counts = min_cells is None and max_cells is None
    s_, _, npos = be.col_stats(m, count_positive=not by_counts)
    number = s_.astype(np.float32) if by_counts else npos
    gene_subset = _subset(number, min_counts if by_counts else min_cells, max_counts if by_counts else max_cells)
    s = int((~gene_subset).sum())
    if s > 0:
        what = (f"less than {min_counts} counts" if min_counts is not None else
                f"less than {min_cells} cells" if min_cells is not None else
                f"more than {max_counts} counts" if max_counts is not None else f"more than {max_cells} cells")
        _log.info("filtered out %d genes that are detected in %s", s, what)
    return gene_subset, number
