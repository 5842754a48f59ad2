"""Oracle: umap-learn `fuzzy_simplicial_set` as the reference calls it.  Test infrastructure.

Reference call site: src/scanpy/neighbors/_connectivity.py:103-138
  fuzzy_simplicial_set(coo((n,1)), n_neighbors, None, None, knn_indices=..., knn_dists=...,
                       set_op_mix_ratio=1.0, local_connectivity=1.0)
umap-learn (>=0.5.12, pyproject.toml:76) is NOT in this container; the arithmetic below
restates its published `smooth_knn_dist` / `compute_membership_strengths` / fuzzy-union
 and is pinned by tests/test_neighbors.py:43-48 and by the bundled
pbmc68k_reduced fixture (tests/golden/pbmc68k_reduced.npz).
"""
from __future__ import annotations

import numpy as np
from scipy import sparse

SMOOTH_K_TOLERANCE = 1e-5
MIN_K_DIST_SCALE = 1e-3
N_ITER = 64


def smooth_knn_dist(distances: np.ndarray, k: float, local_connectivity: float = 1.0):
    """Per-row (sigma, rho).  `distances` is (n, k) float32 with the self column first.

    umap semantics: bisection variables are float64, inputs/outputs float32.
    """
    distances = np.asarray(distances, dtype=np.float32)
    n = distances.shape[0]
    target = np.log2(k)
    rho = np.zeros(n, dtype=np.float32)
    result = np.zeros(n, dtype=np.float32)
    mean_distances = float(np.mean(distances))
    for i in range(n):
        lo, hi, mid = 0.0, np.inf, 1.0
        ith = distances[i]
        non_zero = ith[ith > 0.0]
        if non_zero.shape[0] >= local_connectivity:
            index = int(np.floor(local_connectivity))
            interpolation = local_connectivity - index
            if index > 0:
                rho[i] = non_zero[index - 1]
                if interpolation > SMOOTH_K_TOLERANCE:
                    rho[i] += interpolation * (non_zero[index] - non_zero[index - 1])
            else:
                rho[i] = interpolation * non_zero[0]
        elif non_zero.shape[0] > 0:
            rho[i] = np.max(non_zero)
        for _ in range(N_ITER):
            psum = 0.0
            for j in range(1, distances.shape[1]):
                d = float(np.float32(distances[i, j] - rho[i]))
                psum += np.exp(-(d / mid)) if d > 0 else 1.0
            if abs(psum - target) < SMOOTH_K_TOLERANCE:
                break
            if psum > target:
                hi = mid
                mid = (lo + hi) / 2.0
            else:
                lo = mid
                if hi == np.inf:
                    mid *= 2
                else:
                    mid = (lo + hi) / 2.0
        result[i] = mid
        if rho[i] > 0.0:
            mean_ith = float(np.mean(ith))
            if result[i] < MIN_K_DIST_SCALE * mean_ith:
                result[i] = MIN_K_DIST_SCALE * mean_ith
        elif result[i] < MIN_K_DIST_SCALE * mean_distances:
            result[i] = MIN_K_DIST_SCALE * mean_distances
    return result, rho


def smooth_knn_dist_vec(distances: np.ndarray, k: float, mean_all: float | None = None):
    """Vectorised (all rows at once) version of `smooth_knn_dist` for local_connectivity=1.

    Same arithmetic (float64 bisection on float32 inputs); used for n up to ~1e6 on the host.  `mean_all`: the mean of
    ALL distances of the problem when `distances` holds a subset of its rows (bench.py's sampled full-size check).
    """
    d32 = np.asarray(distances, dtype=np.float32)
    n, kk = d32.shape
    target = np.log2(k)
    pos = d32 > 0
    has_pos = pos.any(axis=1)
    first_pos = np.argmax(pos, axis=1)
    rho = np.where(has_pos, d32[np.arange(n), first_pos], np.float32(0)).astype(np.float32)
    dd = (d32[:, 1:] - rho[:, None]).astype(np.float32).astype(np.float64)  # f32 subtraction, then f64
    lo = np.zeros(n)
    hi = np.full(n, np.inf)
    mid = np.ones(n)
    active = np.ones(n, dtype=bool)
    for _ in range(N_ITER):
        if not active.any():
            break
        with np.errstate(over="ignore", divide="ignore", invalid="ignore"):
            e = np.where(dd[active] > 0, np.exp(-(dd[active] / mid[active, None])), 1.0)
        psum = e.sum(axis=1)
        done = np.abs(psum - target) < SMOOTH_K_TOLERANCE
        idx = np.flatnonzero(active)
        gt = psum > target
        # psum > target: hi = mid; mid = (lo+hi)/2
        i_gt = idx[gt & ~done]
        hi[i_gt] = mid[i_gt]
        mid[i_gt] = (lo[i_gt] + hi[i_gt]) / 2.0
        i_le = idx[~gt & ~done]
        lo[i_le] = mid[i_le]
        inf_hi = np.isinf(hi[i_le])
        mid[i_le[inf_hi]] *= 2
        fin = i_le[~inf_hi]
        mid[fin] = (lo[fin] + hi[fin]) / 2.0
        active[idx[done]] = False
    result = mid.astype(np.float32)
    mean_all = float(np.mean(d32)) if mean_all is None else float(mean_all)
    mean_ith = d32.mean(axis=1, dtype=np.float64)
    floor = np.where(rho > 0, MIN_K_DIST_SCALE * mean_ith, MIN_K_DIST_SCALE * mean_all)
    result = np.where(result < floor, floor, result).astype(np.float32)
    return result, rho


def compute_membership_strengths(knn_indices, knn_dists, sigmas, rhos):
    """COO triplets (rows, cols, vals float32); self / -1 entries get weight 0."""
    knn_indices = np.asarray(knn_indices)
    d = np.asarray(knn_dists, dtype=np.float32)
    n, k = knn_indices.shape
    rows = np.repeat(np.arange(n, dtype=np.int64), k)
    cols = knn_indices.astype(np.int64).ravel()
    sig = np.asarray(sigmas, dtype=np.float32)[:, None]
    rho = np.asarray(rhos, dtype=np.float32)[:, None]
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        diff = (d - rho).astype(np.float32)
        val = np.exp(-(diff / sig).astype(np.float32)).astype(np.float32)
    val = np.where((diff <= 0) | (sig == 0), np.float32(1), val)
    val = np.where(knn_indices == np.arange(n)[:, None], np.float32(0), val)
    val = np.where(knn_indices == -1, np.float32(0), val)
    cols = np.where(cols == -1, 0, cols)
    return rows, cols, val.astype(np.float32).ravel()


def fuzzy_simplicial_set(knn_indices, knn_dists, n_obs: int, n_neighbors: int, *, vectorised: bool = True):
    """-> CSR float32 connectivities = W + W^T - W o W^T (set_op_mix_ratio = 1).

    knn_* have the self column first (src/scanpy/neighbors/__init__.py:639-665).
    """
    knn_dists = np.asarray(knn_dists).astype(np.float32)
    if vectorised:
        sigmas, rhos = smooth_knn_dist_vec(knn_dists, float(n_neighbors))
    else:
        sigmas, rhos = smooth_knn_dist(knn_dists, float(n_neighbors))
    rows, cols, vals = compute_membership_strengths(knn_indices, knn_dists, sigmas, rhos)
    w = sparse.coo_matrix((vals, (rows, cols)), shape=(n_obs, n_obs)).tocsr()
    w.eliminate_zeros()
    wt = w.T.tocsr()
    prod = w.multiply(wt)
    res = (w + wt - prod).tocsr()
    res.eliminate_zeros()
    res.sort_indices()
    return res.astype(np.float32), sigmas, rhos


def gauss_knn(knn_indices, knn_dists, n_obs: int):
    """`gauss(distances, n_neighbors, knn=True)` for a sparse distance matrix (src/scanpy/neighbors/_connectivity.py:
    21-100, the CSR branch): sigma_i^2 = median of the squared distances to the stored neighbours,
    w_ij = sqrt(2 sigma_i sigma_j / (sigma_i^2 + sigma_j^2)) exp(-d_ij^2 / (sigma_i^2 + sigma_j^2)) on the kNN pattern,
    then w_ji := w_ij wherever i is not among j's neighbours.  knn_* include the self column (index 0).  float64."""
    idx = np.asarray(knn_indices)[:, 1:]
    d_sq = np.asarray(knn_dists, dtype=np.float64)[:, 1:] ** 2
    sig_sq = np.median(d_sq, axis=1)
    sig = np.sqrt(sig_sq)
    num = 2 * sig[:, None] * sig[idx]
    den = sig_sq[:, None] + sig_sq[idx]
    w = np.sqrt(num / den) * np.exp(-d_sq / den)
    rows = np.repeat(np.arange(n_obs), idx.shape[1])
    m = sparse.csr_matrix((w.ravel(), (rows, idx.ravel())), shape=(n_obs, n_obs))
    pattern = m.copy()
    pattern.data[:] = 1.0
    missing = pattern.T - pattern.T.multiply(pattern)  # (j, i) stored only as (i, j)
    fill = m.T.multiply(missing)
    return (m + fill).tocsr()


def jaccard_knn(knn_indices, n_obs: int, n_neighbors: int):
    """`jaccard(knn_indices, n_obs=, n_neighbors=)` (src/scanpy/neighbors/_connectivity.py:141-186): PhenoGraph's
    |N(i) & N(j)| / (2 (k - 1) - |N(i) & N(j)|) on the kNN pattern (self excluded), symmetrised by averaging."""
    idx = np.asarray(knn_indices)[:, 1:]
    rows = np.repeat(np.arange(n_obs), idx.shape[1])
    adj = sparse.csr_matrix((np.ones(idx.size), (rows, idx.ravel())), shape=(n_obs, n_obs))
    shared = np.asarray(adj[rows].multiply(adj[idx.ravel()]).sum(axis=1)).ravel()
    jac = shared / (2 * (n_neighbors - 1) - shared)
    mask = jac != 0
    c = sparse.csr_matrix((jac[mask], (rows[mask], idx.ravel()[mask])), shape=(n_obs, n_obs))
    return ((c + c.T) / 2).tocsr()
