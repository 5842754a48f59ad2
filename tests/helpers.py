"""Shared comparison helpers for parity tests."""
from __future__ import annotations

import numpy as np


def knn_sets_equal_mod_ties(idx_a, dist_a, idx_b, dist_b, *, rtol=1e-6, atol=1e-9):
    """Compare two kNN results (rows sorted by distance).  Returns the number of rows whose index
    SETS differ for a reason other than a tie at the k-th distance."""
    idx_a, idx_b = np.asarray(idx_a), np.asarray(idx_b)
    dist_a, dist_b = np.asarray(dist_a, dtype=np.float64), np.asarray(dist_b, dtype=np.float64)
    assert idx_a.shape == idx_b.shape
    bad = 0
    sa, sb = np.sort(idx_a, axis=1), np.sort(idx_b, axis=1)
    differ = np.flatnonzero((sa != sb).any(axis=1))
    for r in differ:
        only_a = np.setdiff1d(idx_a[r], idx_b[r])
        only_b = np.setdiff1d(idx_b[r], idx_a[r])
        da = dist_a[r][np.isin(idx_a[r], only_a)]
        db = dist_b[r][np.isin(idx_b[r], only_b)]
        kth = max(dist_a[r].max(), dist_b[r].max())
        # all symmetric-difference members must sit at the k-th distance (a genuine tie)
        tol = atol + rtol * kth
        if not (np.all(np.abs(da - kth) <= tol) and np.all(np.abs(db - kth) <= tol)):
            bad += 1
    return bad, len(differ)


def csr_from_parts(indptr, indices, data, n):
    from scipy import sparse

    return sparse.csr_matrix((np.asarray(data), np.asarray(indices), np.asarray(indptr)), shape=(n, n))
