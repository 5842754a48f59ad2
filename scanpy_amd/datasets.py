"""Synthetic inputs for the hot path: seeded log-normal CSR with planted cell types.

The reference's pbmc3k needs a download (src/scanpy/datasets/_datasets.py:433-482), so benchmarks and
parity tests use this generator: `n_types` cell types of unequal size, each with its own gene
programme, so that (a) the leading n_types-1 singular values are separated (PCA loadings are well
conditioned) and (b) the kNN graph has unambiguous communities (Leiden ARI is meaningful).
"""
from __future__ import annotations

import numpy as np
from scipy import sparse


def synthetic_planted(
    n_obs: int,
    n_vars: int = 2000,
    *,
    n_types: int = 64,
    density: float = 0.05,
    p_programme: float = 0.7,
    boost: float = 2.0,
    size_ratio: float = 8.0,
    seed: int = 0,
    chunk: int = 100_000,
    row_range: tuple[int, int] | None = None,
):
    """-> (X csr float32 [n_obs, n_vars] with sorted int32 indices, labels int32 [n_obs]).

    Every cell expresses exactly r = round(density * n_vars) genes, one per stratum of
    n_vars // r consecutive genes (so columns are unique and sorted by construction).  In each
    stratum a cell of type t expresses the type's programme gene with probability `p_programme`
    (value boosted by `boost`), otherwise a uniformly random gene of the stratum.  Values are
    log1p(lognormal(0, 1)).  Type sizes follow a geometric progression spanning `size_ratio`.

    Rows are generated in independent chunks of `chunk` rows (chunk c uses the stream seeded with
    (seed, c)), so `row_range=(start, stop)` yields exactly rows start..stop-1 of the full matrix
    without generating the rest: every rank of a row-sharded run builds only its own block.
    """
    rng = np.random.default_rng(seed)
    r = max(1, int(round(density * n_vars)))
    width = n_vars // r
    if width < 2:
        raise ValueError("density too high for the stratified generator")
    weights = size_ratio ** (-np.arange(n_types) / max(1, n_types - 1))
    weights /= weights.sum()
    pref = rng.integers(0, width, size=(n_types, r), dtype=np.int32)
    strata = (np.arange(r, dtype=np.int32) * width)[None, :]
    start, stop = (0, n_obs) if row_range is None else row_range
    if not 0 <= start <= stop <= n_obs:
        raise ValueError("row_range outside [0, n_obs]")
    n_out = stop - start
    labels = np.empty(n_out, dtype=np.int32)
    indices = np.empty((n_out, r), dtype=np.int32)
    data = np.empty((n_out, r), dtype=np.float32)
    for c in range(start // chunk, (max(stop, 1) - 1) // chunk + 1):
        cs, ce = c * chunk, min(n_obs, (c + 1) * chunk)
        m = ce - cs
        crng = np.random.default_rng([seed, c])
        lab = crng.choice(n_types, size=m, p=weights).astype(np.int32)
        is_prog = crng.random((m, r), dtype=np.float32) < p_programme
        rand_gene = crng.integers(0, width, size=(m, r), dtype=np.int32)
        gene = np.where(is_prog, pref[lab], rand_gene)
        v = np.exp(crng.standard_normal((m, r), dtype=np.float32))
        v *= np.where(is_prog, np.float32(boost), np.float32(1.0))
        lo, hi = max(cs, start), min(ce, stop)
        if lo >= hi:
            continue
        src = slice(lo - cs, hi - cs)
        dst = slice(lo - start, hi - start)
        labels[dst] = lab[src]
        indices[dst] = (strata + gene)[src]
        data[dst] = np.log1p(v)[src]
    n_obs = n_out
    indptr = np.arange(0, n_obs * r + 1, r, dtype=np.int64)
    if indptr[-1] < 2**31:
        indptr = indptr.astype(np.int32)
    x = sparse.csr_matrix((data.ravel(), indices.ravel(), indptr), shape=(n_obs, n_vars))
    x.has_sorted_indices = True
    return x, labels


def blobs_embedding(n_obs: int, n_dims: int = 50, *, n_types: int = 64, spread: float = 1.0, seed: int = 0):
    """Dense float32 [n_obs, n_dims] Gaussian blobs (a stand-in for an X_pca embedding) + labels."""
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((n_types, n_dims)).astype(np.float32) * 4.0
    labels = rng.integers(0, n_types, size=n_obs).astype(np.int32)
    x = centers[labels] + spread * rng.standard_normal((n_obs, n_dims), dtype=np.float32)
    return np.ascontiguousarray(x, dtype=np.float32), labels
