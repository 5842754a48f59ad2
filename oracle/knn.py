"""Oracle: exact kNN as the reference's `transformer='sklearn'` path.  Test infrastructure.

Reference: src/scanpy/neighbors/__init__.py:754-768 (KNeighborsTransformer(algorithm='brute')),
:638-650 (truncate to n_neighbors incl. self, zero diagonal, rebuild CSR without self) and
src/scanpy/neighbors/_common.py:35-61, 74-98, 126-143 (index/distance <-> CSR conventions).
scikit-learn is installed here, so the search itself is the reference's own call.
"""
from __future__ import annotations

import numpy as np
from scipy import sparse


def has_self_column(indices: np.ndarray) -> bool:
    """_common.py:17-22 (`.any()` because duplicates may displace the self entry)."""
    return bool((indices[:, 0] == np.arange(indices.shape[0])).any())


def indices_distances_from_sparse(d: sparse.csr_matrix, n_neighbors: int):
    """_common.py:74-98 restricted to constant-nnz rows (the shortcut path, :126-143)."""
    nnzs = np.diff(d.indptr)
    assert (nnzs == nnzs[0]).all(), "oracle only covers the constant-nnz shortcut"
    n, k = d.shape[0], int(nnzs[0])
    indices = d.indices.reshape(n, k)
    distances = d.data.reshape(n, k)
    if not has_self_column(indices):  # RAPIDS style -> add self column (:88-91)
        indices = np.hstack([np.arange(n)[:, None], indices])
        distances = np.hstack([np.zeros(n)[:, None], distances])
    if indices.shape[1] > n_neighbors:  # (:95-96)
        indices, distances = indices[:, :n_neighbors], distances[:, :n_neighbors]
    return indices, distances


def sparse_from_indices_distances(indices, distances, *, keep_self: bool) -> sparse.csr_matrix:
    """_common.py:35-61."""
    if not keep_self:
        assert has_self_column(indices), "The first neighbor should be the cell itself."
        indices, distances = indices[:, 1:], distances[:, 1:]
    n, k = indices.shape
    indptr = np.arange(0, n * k + 1, k)
    return sparse.csr_matrix((distances.copy().ravel(), indices.copy().ravel(), indptr), shape=(n, n))


def knn_sklearn(x: np.ndarray, n_neighbors: int, *, n_jobs: int | None = None, metric: str = "euclidean"):
    """Reference shortcut path.  Returns (knn_indices (n,k), knn_distances (n,k), distances CSR k-1/row).

    Column 0 is the cell itself with distance exactly 0 (diagonal zeroed in place at
    neighbors/__init__.py:644, aliasing the knn_distances view -- SURVEY.md discrepancy 6).
    """
    from sklearn.neighbors import KNeighborsTransformer

    n = x.shape[0]
    k = min(n - 1, n_neighbors)
    tr = KNeighborsTransformer(algorithm="brute", n_neighbors=k, metric=metric, n_jobs=n_jobs)
    d = tr.fit_transform(x).tocsr()
    knn_indices, knn_distances = indices_distances_from_sparse(d, n_neighbors)
    knn_distances = knn_distances.copy()
    knn_indices = knn_indices.copy()
    # zero the diagonal (only touches entries whose column index equals the row)
    self_mask = knn_indices == np.arange(n)[:, None]
    knn_distances[self_mask] = 0.0
    dist_csr = sparse_from_indices_distances(knn_indices, knn_distances, keep_self=False)
    return knn_indices, knn_distances, dist_csr


def knn_exact_f64(x: np.ndarray, queries: np.ndarray, k: int, block: int = 2048):
    """Direct float64 (q-c)^2 brute force for index `queries` of x: ground truth for tie analysis."""
    x64 = np.asarray(x, dtype=np.float64)
    idx = np.empty((len(queries), k), dtype=np.int64)
    dist = np.empty((len(queries), k), dtype=np.float64)
    for s in range(0, len(queries), block):
        q = x64[queries[s : s + block]]
        d2 = ((q[:, None, :] - x64[None, :, :]) ** 2).sum(-1) if x64.shape[0] * len(q) * x64.shape[1] < 5e7 else (
            (q * q).sum(1)[:, None] + (x64 * x64).sum(1)[None, :] - 2 * q @ x64.T
        )
        part = np.argpartition(d2, k - 1, axis=1)[:, :k]
        pd = np.take_along_axis(d2, part, axis=1)
        # exact re-evaluation of the selected candidates, then order by (distance, index)
        pd = ((q[:, None, :] - x64[part]) ** 2).sum(-1)
        order = np.lexsort((part, pd), axis=1)
        idx[s : s + block] = np.take_along_axis(part, order, axis=1)
        dist[s : s + block] = np.sqrt(np.take_along_axis(pd, order, axis=1))
    return idx, dist


def knn_exact_f64_sample(x: np.ndarray, queries: np.ndarray, k: int, chunk: int = 131072):
    """Float64 brute force for a SAMPLE of query rows of a large x (the full-size check of bench.py): the candidates are
    streamed in chunks (|q|^2 + |c|^2 - 2 q.c in float64 to pick k + 8 per chunk), the survivors re-evaluated as direct
    (q - c)^2 sums and ordered by (distance, index).  -> (idx int64 [m, k], dist float64 [m, k])"""
    queries = np.asarray(queries, dtype=np.int64)
    n, m = x.shape[0], len(queries)
    q = np.asarray(x[queries], dtype=np.float64)
    qq = (q * q).sum(1)
    keep = min(n, k + 8)
    best_i = np.empty((m, 0), dtype=np.int64)
    best_d = np.empty((m, 0), dtype=np.float64)
    for s0 in range(0, n, chunk):
        c = np.asarray(x[s0:s0 + chunk], dtype=np.float64)
        d2 = qq[:, None] + (c * c).sum(1)[None, :] - 2.0 * (q @ c.T)
        kk = min(keep, c.shape[0])
        part = np.argpartition(d2, kk - 1, axis=1)[:, :kk] if kk < c.shape[0] else np.broadcast_to(np.arange(c.shape[0]), (m, c.shape[0]))
        best_i = np.hstack([best_i, part + s0])
        best_d = np.hstack([best_d, np.take_along_axis(d2, part, axis=1)])
        if best_i.shape[1] > 4 * keep:
            sel = np.argpartition(best_d, keep - 1, axis=1)[:, :keep]
            best_i, best_d = np.take_along_axis(best_i, sel, 1), np.take_along_axis(best_d, sel, 1)
    x64 = np.asarray(x[best_i.ravel()], dtype=np.float64).reshape(m, best_i.shape[1], -1)
    exact = ((q[:, None, :] - x64) ** 2).sum(-1)
    order = np.lexsort((best_i, exact), axis=1)[:, :k]
    return np.take_along_axis(best_i, order, 1), np.sqrt(np.take_along_axis(exact, order, 1))
