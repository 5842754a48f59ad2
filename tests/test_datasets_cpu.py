"""Synthetic workload generators (scanpy_amd/datasets.py): what bench.py and the sharded tests rely on."""
from __future__ import annotations

import numpy as np
from scipy import sparse

from scanpy_amd._pipeline import shard_bounds
from scanpy_amd.datasets import blobs_embedding, synthetic_planted


def test_row_range_shards_are_slices_of_the_full_matrix():
    """every rank generates ONLY its rows; the shards must be exactly the row blocks of the single-rank matrix"""
    n, g = 10001, 300
    full, lab = synthetic_planted(n, g, n_types=8, seed=3)
    for world in (2, 3, 8):
        parts = [synthetic_planted(n, g, n_types=8, seed=3, row_range=shard_bounds(n, world, r)) for r in range(world)]
        assert (sparse.vstack([p[0] for p in parts]).tocsr() != full).nnz == 0
        assert np.array_equal(np.concatenate([p[1] for p in parts]), lab)


def test_synthetic_planted_shape_density_and_determinism():
    x, lab = synthetic_planted(4000, 500, n_types=12, seed=0)
    assert x.shape == (4000, 500) and x.dtype == np.float32 and x.indices.dtype == np.int32 and x.has_sorted_indices
    assert abs(x.nnz / (4000 * 500) - 0.05) < 0.005 and (x.data > 0).all()
    assert lab.shape == (4000,) and len(np.unique(lab)) == 12
    sizes = np.bincount(lab)
    assert sizes.max() > 1.5 * sizes.min()  # unequal cell-type sizes, as SURVEY 8(d) asks
    y, lab2 = synthetic_planted(4000, 500, n_types=12, seed=0)
    assert (x != y).nnz == 0 and np.array_equal(lab, lab2)
    z, _ = synthetic_planted(4000, 500, n_types=12, seed=1)
    assert (x != z).nnz > 0


def test_blobs_embedding():
    x, lab = blobs_embedding(3000, 50, n_types=7, seed=2)
    assert x.shape == (3000, 50) and x.dtype == np.float32 and len(np.unique(lab)) == 7
    x2, _ = blobs_embedding(3000, 50, n_types=7, seed=2)
    assert np.array_equal(x, x2)
