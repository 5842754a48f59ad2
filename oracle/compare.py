"""Parity comparisons between a product result and the CPU chain (north_star gates).  Test infrastructure:
imported by `tests/`, `__graft_entry__.smoke()` and the `parity` / `cpu_baseline` legs of `bench.py` only.

Gates (BASELINE.json north_star): kNN index sets equal (ties at the k-th distance excepted), PCA loadings within
1e-4 up to sign, Leiden ARI >= 0.99.  The connectivities bar (1e-5 absolute, identical pattern) is the reference's own
tolerance for the same comparison (tests/test_neighbors.py:43-48, :275-296).
"""
from __future__ import annotations

import numpy as np

GATES = {"pca_loading_err": 1e-4, "knn_rows_differing_beyond_ties": 0, "conn_max_abs": 1e-5, "conn_max_rel": 1e-5,
         "conn_e2e_max_rel": 1e-5, "conn_max_abs_from_gpu_distances": 1e-4, "leiden_ari_vs_cpu_chain": 0.99}


def pca_loading_err(components_a, components_b) -> float:
    """max |  |V_a| - |V_b|  | over all loadings: sign-free, the form of tests/test_pca.py:225-274"""
    a, b = np.abs(np.asarray(components_a, dtype=np.float64)), np.abs(np.asarray(components_b, dtype=np.float64))
    if a.shape != b.shape:
        a = a.T
    return float(np.abs(a - b).max())


def knn_rows_differing_beyond_ties(idx_a, dist_a, idx_b, dist_b, *, rtol=1e-6, atol=1e-9):
    """-> (rows whose index SETS differ for a reason other than a tie at the k-th distance, rows that differ at all).
    Rows of both results are sorted by distance; the self column is part of both."""
    idx_a, idx_b = np.asarray(idx_a), np.asarray(idx_b)
    dist_a, dist_b = np.asarray(dist_a, dtype=np.float64), np.asarray(dist_b, dtype=np.float64)
    assert idx_a.shape == idx_b.shape
    sa, sb = np.sort(idx_a, axis=1), np.sort(idx_b, axis=1)
    differ = np.flatnonzero((sa != sb).any(axis=1))
    bad = 0
    for r in differ:
        only_a = ~np.isin(idx_a[r], idx_b[r])
        only_b = ~np.isin(idx_b[r], idx_a[r])
        kth = max(dist_a[r].max(), dist_b[r].max())
        tol = atol + rtol * kth
        # every member of the symmetric difference must sit at the k-th distance (a genuine tie)
        if not (np.all(np.abs(dist_a[r][only_a] - kth) <= tol) and np.all(np.abs(dist_b[r][only_b] - kth) <= tol)):
            bad += 1
    return bad, int(differ.size)


def conn_max_abs(conn_a, conn_b):
    """-> (max |a - b|, identical sparsity pattern?) of two CSR connectivity matrices"""
    a, b = conn_a.tocsr(), conn_b.tocsr()
    a.sort_indices()
    b.sort_indices()
    same = a.nnz == b.nnz and np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices)
    if same:
        return float(np.abs(a.data.astype(np.float64) - b.data.astype(np.float64)).max(initial=0.0)), True
    d = (a - b).tocsr()
    return float(np.abs(d.data).max(initial=0.0)), False


def ari(labels_a, labels_b) -> float:
    from sklearn.metrics import adjusted_rand_score

    return float(adjusted_rand_score(np.asarray(labels_a), np.asarray(labels_b)))


def conn_max_rel(conn_a, conn_b, row_ok=None):
    """-> (max |a - b| / |b| over the stored entries, number of entries compared); both matrices must have the same
    sparsity pattern.  `row_ok` (bool per row): only entries (i, j) with row_ok[i] and row_ok[j] are compared -- the value
    of an entry of the fuzzy union depends on the neighbour lists of both of its end points.  This is the form of the
    reference's own check of connectivities recomputed from given distances (tests/test_neighbors.py:275-296, rtol 1e-5)."""
    a, b = conn_a.tocsr(), conn_b.tocsr()
    a.sort_indices()
    b.sort_indices()
    if not (a.nnz == b.nnz and np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices)):
        return float("inf"), 0
    da, db = a.data.astype(np.float64), b.data.astype(np.float64)
    keep = np.ones(a.nnz, dtype=bool)
    if row_ok is not None:
        rows = np.repeat(np.arange(a.shape[0]), np.diff(a.indptr))
        keep = row_ok[rows] & row_ok[a.indices]
    if not keep.any():
        return 0.0, 0
    return float((np.abs(da - db)[keep] / np.abs(db[keep])).max()), int(keep.sum())
