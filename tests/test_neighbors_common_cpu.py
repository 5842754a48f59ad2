"""Host-side kNN <-> CSR conventions (scanpy_amd/neighbors/_common.py) against the cases of the reference's
tests/test_neighbors_common.py:25-103 (self column styles, duplicates, shortcut path)."""
from __future__ import annotations

import numpy as np
import pytest
from sklearn.neighbors import KNeighborsTransformer

from scanpy_amd.neighbors._common import (
    _ind_dist_shortcut,
    get_indices_distances_from_sparse_matrix,
    get_sparse_matrix_from_indices_distances,
    has_self_column,
)


def mk_knn_matrix(n_obs, n_neighbors, *, style, duplicates=False):
    """tests/test_neighbors_common.py:25-57"""
    rng = np.random.default_rng(0)
    n_col = n_neighbors + (1 if style == "sklearn" else 0)
    dists = np.abs(rng.standard_normal((n_obs, n_col))) + 1e-8
    idxs = np.arange(n_obs * n_col).reshape((n_col, n_obs)).T % n_obs
    idxs[:, 0] = np.arange(n_obs)
    if style == "rapids":
        idxs[:, 0] = (idxs[:, 0] + 1) % n_obs  # does not include the cell itself
    else:
        dists[:, 0] = 0.0
    if duplicates:
        dists[n_obs // 4:n_obs, 2] = 0.0
    mat = get_sparse_matrix_from_indices_distances(idxs, dists, keep_self=True)
    assert has_self_column(idxs) == (style != "rapids")
    if duplicates:  # explicit zeros keep the sparsity pattern regular
        nnz = np.diff(mat.indptr)
        assert (nnz == nnz[0]).all()
        sp = mat.copy()
        sp.eliminate_zeros()
        nnz2 = np.diff(sp.indptr)
        assert not (nnz2 == nnz2[0]).all()
    return mat, idxs, dists


@pytest.mark.parametrize("n_neighbors", [3, None], ids=["3", "all"])
@pytest.mark.parametrize("style", ["basic", "rapids", "sklearn"])
@pytest.mark.parametrize("duplicates", [True, False], ids=["duplicates", "unique"])
def test_ind_dist_shortcut_manual(n_neighbors, style, duplicates):
    n_obs = 10
    k = n_obs if n_neighbors is None else n_neighbors
    mat, idxs, dists = mk_knn_matrix(n_obs, k, style=style, duplicates=duplicates)
    assert (mat.nnz / n_obs) == k + (1 if style == "sklearn" else 0)
    assert _ind_dist_shortcut(mat) is not None
    ind, dist = get_indices_distances_from_sparse_matrix(mat, k)
    assert ind.shape == dist.shape == (n_obs, k)
    assert (ind[:, 0] == np.arange(n_obs)).all() and (dist[:, 0] == 0).all()  # self first, whatever the style
    if style == "rapids":  # the self column was inserted in front, the rest kept in order
        np.testing.assert_array_equal(ind[:, 1:], idxs[:, :k - 1])
        np.testing.assert_array_equal(dist[:, 1:], dists[:, :k - 1])
    else:
        np.testing.assert_array_equal(ind, idxs[:, :k])


@pytest.mark.parametrize("n_neighbors", [3, None], ids=["3", "all"])
def test_ind_dist_shortcut_premade(n_neighbors):
    """tests/test_neighbors_common.py:81-103: sklearn's own transformer output takes the shortcut"""
    n_obs = 10
    k = n_obs - 1 if n_neighbors is None else n_neighbors
    mat = KNeighborsTransformer(n_neighbors=k).fit_transform(np.random.default_rng(0).standard_normal((n_obs, n_obs // 4)))
    assert (mat.nnz / n_obs) == k + 1
    assert _ind_dist_shortcut(mat) is not None


def test_ragged_matrix_takes_slow_path_with_warning():
    mat, _, _ = mk_knn_matrix(10, 4, style="basic")
    mat = mat.tolil()
    mat[3, mat[3].nonzero()[1][-1]] = 0
    mat = mat.tocsr()
    mat.eliminate_zeros()
    with pytest.warns(RuntimeWarning, match="no constant number of neighbors"):
        ind, dist = get_indices_distances_from_sparse_matrix(mat, 4)
    assert ind.shape == (10, 4)


def test_device_list_route_equals_the_reference_conventions():
    """`sparse_distances_from_device` / `graph_from_device` (the built-in search keeps its lists as torch tensors and
    wraps the CSR slots without scipy's validation passes) == the checked constructions of `_common.py:35-61`"""
    import torch
    from scipy import sparse

    from scanpy_amd.neighbors import _common as c

    rng = np.random.default_rng(0)
    n, k = 300, 7
    idx = np.stack([rng.permutation(n)[:k] for _ in range(n)]).astype(np.int32)
    idx[:, 0] = np.arange(n)
    dist = np.sort(rng.random((n, k)), axis=1)
    dist[:, 0] = 0.0
    fast = c.sparse_distances_from_device(torch.from_numpy(idx), torch.from_numpy(dist))
    ref = c.get_sparse_matrix_from_indices_distances(idx.astype(np.int64), dist, keep_self=False)
    assert fast.shape == ref.shape and fast.dtype == ref.dtype == np.float64
    assert fast.indices.dtype == fast.indptr.dtype == ref.indices.dtype == np.int32
    np.testing.assert_array_equal(fast.indptr, ref.indptr)
    np.testing.assert_array_equal(fast.indices, ref.indices)
    np.testing.assert_array_equal(fast.data, ref.data)
    assert (fast @ np.ones(n)).shape == (n,) and (fast.T.tocsr() != ref.T.tocsr()).nnz == 0  # behaves as a CSR
    bad = idx.copy()
    bad[:, 0] = (np.arange(n) + 1) % n
    with pytest.raises(AssertionError, match="first neighbor should be the cell itself"):
        c.sparse_distances_from_device(torch.from_numpy(bad), torch.from_numpy(dist))
    g = sparse.random(n, n, density=0.05, format="csr", dtype=np.float32, random_state=1)
    g.sort_indices()
    m = c.graph_from_device(torch.from_numpy(g.indptr.astype(np.int64)), torch.from_numpy(g.indices.astype(np.int32)),
                            torch.from_numpy(g.data), n)
    assert (m != g).nnz == 0 and m.indices.dtype == m.indptr.dtype == np.int32
    assert m.has_sorted_indices and m.has_canonical_format  # recorded, and true of what the kernels return
    with pytest.raises(AssertionError):
        c.csr_from_trusted_arrays(g.data, g.indices.astype(np.int32), g.indptr.astype(np.int64), (n, n))
