"""Pin the CPU oracle against every golden vector the reference ships for the path."""
from __future__ import annotations

import numpy as np
import pytest
from scipy import sparse

from oracle import connectivities as oc
from oracle import knn as oknn
from oracle import pca as opca


# ---- PCA: tests/test_pca.py:34-59, tolerance from :225-233 (2e-5 on absolute values) -------------
@pytest.mark.parametrize("fmt", ["dense", "csr"])
def test_pca_golden_A(pca_toy, fmt):
    a = pca_toy["A_list"].astype("float32")
    x = sparse.csr_matrix(a) if fmt == "csr" else a
    res = opca.pca_reference(x, 4)
    assert np.linalg.norm(np.abs(pca_toy["A_pca"][:, :4]) - np.abs(res["X_pca"])) < 2e-05


def test_pca_golden_A_svd(pca_toy):
    """tests/test_pca.py:264-274: zero_center=False -> TruncatedSVD; golden A_svd."""
    a = sparse.csr_matrix(pca_toy["A_list"].astype("float32"))
    res = opca.pca_reference(a, 4, zero_center=False, svd_solver="arpack")
    assert np.linalg.norm(np.abs(pca_toy["A_svd"][:, :4]) - np.abs(res["X_pca"])) < 2e-05


def test_pca_dense_truth_matches_reference(pca_toy):
    a = pca_toy["A_list"].astype("float64")
    ref = opca.pca_reference(a, 4)
    tru = opca.pca_dense_f64(a, 4)
    np.testing.assert_allclose(np.abs(ref["X_pca"]), np.abs(tru["X_pca"]), atol=1e-10)
    np.testing.assert_allclose(ref["variance"], tru["variance"], rtol=1e-10)
    np.testing.assert_allclose(ref["variance_ratio"], tru["variance_ratio"], rtol=1e-10)
    np.testing.assert_allclose(ref["components"], tru["components"], atol=1e-10)  # same sign rule


# ---- kNN: tests/test_neighbors.py:23-39, 151-165 ---------------------------------------------------
def test_knn_golden_toy(neighbors_toy):
    x = neighbors_toy["X"]
    k = int(neighbors_toy["n_neighbors"])
    idx, dist, dcsr = oknn.knn_sklearn(x, k)
    np.testing.assert_allclose(dcsr.toarray(), neighbors_toy["distances_euclidean"], rtol=1e-6)
    assert idx.shape == (4, 3) and (idx[:, 0] == np.arange(4)).all() and (dist[:, 0] == 0).all()


def test_knn_conventions_rapids_style():
    """tests/test_neighbors_common.py:25-74: a k-column CSR without self gets the self column prepended."""
    n, k = 10, 3
    rng = np.random.default_rng(0)
    d = np.abs(rng.standard_normal((n, k))) + 1e-8
    idx = (np.arange(n)[:, None] + 1 + np.arange(k)[None, :]) % n
    m = oknn.sparse_from_indices_distances(idx, d, keep_self=True)
    i2, d2 = oknn.indices_distances_from_sparse(m, k)
    assert i2.shape == (n, k) and (i2[:, 0] == np.arange(n)).all() and (d2[:, 0] == 0).all()
    np.testing.assert_array_equal(i2[:, 1:], idx[:, : k - 1])


# ---- connectivities: tests/test_neighbors.py:43-48 + bundled fixture -------------------------------
@pytest.mark.parametrize("vectorised", [True, False])
def test_connectivities_golden_toy(neighbors_toy, vectorised):
    x = neighbors_toy["X"]
    k = int(neighbors_toy["n_neighbors"])
    idx, dist, _ = oknn.knn_sklearn(x, k)
    c, _, _ = oc.fuzzy_simplicial_set(idx, dist, 4, k, vectorised=vectorised)
    np.testing.assert_allclose(c.toarray(), neighbors_toy["connectivities_umap"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("vectorised", [True, False])
def test_connectivities_fixture(pbmc68k, vectorised):
    """Stored obsp['distances'] (9/row, k=10) -> stored obsp['connectivities'] (9992 nnz)."""
    d = pbmc68k["distances"]
    k = pbmc68k["n_neighbors"]
    idx, dist = oknn.indices_distances_from_sparse(d, k)
    # the stored CSR is column-sorted; the graph was built from the kNN output, which is
    # distance-sorted (rho = first positive distance depends on that order) -> restore it
    order = np.argsort(dist, axis=1, kind="stable")
    idx, dist = np.take_along_axis(idx, order, 1), np.take_along_axis(dist, order, 1)
    c, _, _ = oc.fuzzy_simplicial_set(idx, dist, d.shape[0], k, vectorised=vectorised)
    ref = pbmc68k["connectivities"].copy()
    ref.sort_indices()
    assert c.nnz == ref.nnz == 9992
    np.testing.assert_array_equal(c.indptr, ref.indptr)
    np.testing.assert_array_equal(c.indices, ref.indices)
    np.testing.assert_allclose(c.data, ref.data, rtol=0, atol=1e-5)


def test_gauss_and_jaccard_golden_toy(neighbors_toy):
    """tests/test_neighbors.py:65-71, 120-125, 196-227: method='gauss' (knn) and 'jaccard' on the 4-point toy"""
    x = neighbors_toy["X"]
    k = int(neighbors_toy["n_neighbors"])
    idx, dist, _ = oknn.knn_sklearn(x, k)
    g = oc.gauss_knn(idx, dist, x.shape[0])
    np.testing.assert_allclose(g.toarray(), neighbors_toy["connectivities_gauss_knn"], rtol=1e-6)
    j = oc.jaccard_knn(idx, x.shape[0], k)
    np.testing.assert_allclose(j.toarray(), neighbors_toy["connectivities_jaccard"], rtol=1e-12)
