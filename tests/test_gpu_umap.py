"""-m gpu: scanpy_amd.tl.umap / scamd_umap_optimize_f32 against the oracle (oracle/umap.py, PARITY UNPINNED: umap-learn is
not installed and the reference ships no golden embedding -- see the oracle's header).
  * kernel == the same synchronous scheme on the CPU, element for element, after a few epochs;
  * full runs: objective (fuzzy cross entropy), trustworthiness and separation of planted clusters on a par with the
    reference's sequential sweep; bitwise determinism per seed."""
from __future__ import annotations

import numpy as np
import pytest
from scipy import sparse
from sklearn.manifold import trustworthiness

import scanpy_amd as sc
from oracle import umap as ou

pytestmark = pytest.mark.gpu


def _adata(pbmc68k):
    adata = sc.AnnData(pbmc68k["X"].copy())
    adata.obsm["X_pca"] = pbmc68k["X_pca"]
    adata.obsp["connectivities"] = pbmc68k["connectivities"].astype(np.float32)
    adata.obsp["distances"] = pbmc68k["distances"]
    adata.uns["neighbors"] = dict(connectivities_key="connectivities", distances_key="distances",
                                  params=dict(n_neighbors=10, method="umap"))
    return adata


@pytest.mark.parametrize("dim", [2, 3, 5])
@pytest.mark.parametrize(("n_epochs", "atol"), [(2, 1e-5), (4, 5e-3)])
def test_kernel_matches_synchronous_oracle(pbmc68k, dim, n_epochs, atol):
    """Same scheme on the CPU (oracle_umap_synchronous), element for element.  The first epochs (alpha ~ 1, random start)
    are violently expansive: a 1-ulp difference of powf grows ~20x per epoch (measured 1e-6, 1e-4, 5e-4, 2e-2, 0.6 after
    2, 3, 4, 6, 8 epochs), so the comparison is made after 2 epochs (ulp level) and after 4."""
    import torch

    from scanpy_amd import _kernels as K

    g = ou.prune_graph(pbmc68k["connectivities"], n_epochs).tocsr()
    g.sort_indices()
    n = g.shape[0]
    eps = ou.make_epochs_per_sample(g.data, n_epochs).astype(np.float32)
    a, b = ou.find_ab_params()
    y0 = np.random.default_rng(dim).uniform(0, 10, size=(n, dim)).astype(np.float32)
    ref = ou.optimize_layout(g, y0, n_epochs=n_epochs, a=a, b=b, seed=11, scheme="synchronous")
    dev = torch.device("cuda")
    y = torch.from_numpy(y0.copy()).to(dev)
    K.umap_optimize_(torch.from_numpy(g.indptr.astype(np.int64)).to(dev), torch.from_numpy(g.indices.astype(np.int32)).to(dev),
                     torch.from_numpy(eps).to(dev), n, y, n_epochs=n_epochs, a=a, b=b, seed=11)
    got = y.cpu().numpy()
    assert np.abs(got - y0).max() > 0.5  # it moved
    np.testing.assert_allclose(got, ref, rtol=0, atol=atol)


def test_umap_fixture_quality_and_determinism(pbmc68k):
    adata = _adata(pbmc68k)
    sc.tl.umap(adata)
    y = adata.obsm["X_umap"]
    assert y.shape == (700, 2) and y.dtype == np.float32 and np.isfinite(y).all()
    a, b = ou.find_ab_params()
    assert abs(adata.uns["umap"]["params"]["a"] - a) < 1e-9 and adata.uns["umap"]["params"]["random_state"] == 0
    g = pbmc68k["connectivities"]
    y_ref = ou.simplicial_set_embedding(g, seed=0, scheme="sequential")
    ce, ce_ref = ou.fuzzy_cross_entropy(g, y, a, b), ou.fuzzy_cross_entropy(g, y_ref, a, b)
    assert ce < 1.06 * ce_ref
    x = pbmc68k["X_pca"]
    assert trustworthiness(x, y, n_neighbors=15) > trustworthiness(x, y_ref, n_neighbors=15) - 0.02
    other = _adata(pbmc68k)
    sc.tl.umap(other)
    assert np.array_equal(other.obsm["X_umap"], y)  # bitwise reproducible
    sc.tl.umap(other, random_state=3, key_added="X_alt")
    assert not np.array_equal(other.obsm["X_alt"], y) and "X_alt" in other.uns


def test_umap_init_variants_and_components(pbmc68k):
    adata = _adata(pbmc68k)
    sc.tl.umap(adata, init_pos="random", n_components=3, maxiter=50)
    assert adata.obsm["X_umap"].shape == (700, 3)
    for dtype in (np.float32, np.float64):  # tests/test_embedding.py:55-66: the init dtype does not matter
        sc.tl.umap(adata, init_pos=adata.obsm["X_pca"][:, :2].astype(dtype), key_added=f"u_{np.dtype(dtype).name}", maxiter=30)
    np.testing.assert_array_equal(adata.obsm["u_float32"], adata.obsm["u_float64"])
    sc.tl.umap(adata, init_pos="u_float32", maxiter=5)
    conn = adata.obsp["connectivities"].copy()
    sc.tl.umap(adata, maxiter=5)
    assert (adata.obsp["connectivities"] != conn).nnz == 0  # tests/test_embedding.py:83-95
    # init_pos='paga' (src/scanpy/tools/_umap.py:175-179): a PAGA layout brought along in uns['paga']; zero epochs return
    # the initial coordinates themselves (rescaled to [0, 10] like every initial embedding), cells of a group next to its node
    import pandas as pd

    grp = pd.Categorical((np.arange(700) % 3).astype(str))
    adata.obs["grp"] = grp
    adata.uns["paga"] = dict(pos=np.array([[0.0, 0.0], [10.0, 0.0], [0.0, 10.0]]), groups="grp",
                             connectivities=np.array([[0, 1.0, 0.2], [1.0, 0, 0], [0.2, 0, 0]]))
    sc.tl.umap(adata, init_pos="paga", key_added="u_paga", maxiter=1)
    y = adata.obsm["u_paga"]
    assert y.shape == (700, 2) and np.isfinite(y).all()
    cen = np.array([y[np.asarray(grp == c)].mean(0) for c in grp.categories])
    assert np.linalg.norm(cen[0] - cen[1]) > 1.0 and np.linalg.norm(cen[0] - cen[2]) > 1.0  # the groups start apart
    with pytest.raises(ValueError, match="n_components must be 2"):
        sc.tl.umap(adata, init_pos="paga", n_components=3)


def test_umap_separates_planted_clusters():
    from oracle import connectivities as oc
    from oracle import knn as oknn
    from scanpy_amd.datasets import blobs_embedding
    from sklearn.neighbors import NearestNeighbors

    n = 6000
    x, lab = blobs_embedding(n, 50, n_types=10, seed=5)
    idx, dist, _ = oknn.knn_sklearn(x, 15, n_jobs=-1)
    c, _, _ = oc.fuzzy_simplicial_set(idx, dist, n, 15)
    adata = sc.AnnData(x)
    adata.obsp["connectivities"] = sparse.csr_matrix(c).astype(np.float32)
    adata.uns["neighbors"] = dict(connectivities_key="connectivities", distances_key="distances",
                                  params=dict(n_neighbors=15, method="umap"))
    sc.tl.umap(adata)
    y = adata.obsm["X_umap"]
    nb = NearestNeighbors(n_neighbors=11).fit(y).kneighbors(y, return_distance=False)[:, 1:]
    purity = (lab[nb] == lab[:, None]).mean()
    assert purity > 0.97


def test_layout_preserves_the_graph_as_well_as_the_sequential_oracle_at_30k():
    """UMAP's parity is unpinned (no umap-learn here, no golden embedding in the reference: tests/test_embedding.py only
    smoke-tests it) -- this is the quality floor instead.  30 000 cells on a curved 2-D sheet in 50 dimensions (one
    connected graph): the GPU layout keeps as many of a cell's GRAPH neighbours among its 30 nearest points of the plane,
    and follows the sheet's own geometry as closely (correlation of plane distances with latent distances), as the CPU
    restatement of the reference's sequential sweep does (oracle/umap.py, scheme 'sequential': 0.97 / 0.96), within 0.05."""
    from oracle import connectivities as oc
    from oracle import knn as oknn
    from oracle import umap as ou
    from sklearn.neighbors import NearestNeighbors

    n, k = 30_000, 15
    rng = np.random.default_rng(9)
    uv = rng.random((n, 2))
    feat = np.stack([uv[:, 0], uv[:, 1], np.sin(3 * uv[:, 0]), np.cos(3 * uv[:, 1]), uv[:, 0] * uv[:, 1],
                     np.sin(2 * (uv[:, 0] + uv[:, 1]))], axis=1)
    x = (feat @ rng.standard_normal((6, 50)) + 0.01 * rng.standard_normal((n, 50))).astype(np.float32)
    idx, dist, _ = oknn.knn_sklearn(x, k, n_jobs=-1)
    c, _, _ = oc.fuzzy_simplicial_set(idx, dist, n, k)
    g = sparse.csr_matrix(c).astype(np.float32)
    adata = sc.AnnData(x)
    adata.obsp["connectivities"] = g
    adata.uns["neighbors"] = dict(connectivities_key="connectivities", distances_key="distances",
                                  params=dict(n_neighbors=k, method="umap"))
    sc.tl.umap(adata)
    y_gpu = adata.obsm["X_umap"]
    y_cpu = ou.simplicial_set_embedding(g, seed=0, scheme="sequential")
    sample = rng.choice(n, 2000, replace=False)
    d_lat = np.linalg.norm(uv[sample][:, None] - uv[sample][None], axis=2).ravel()

    def scores(y):
        nb = NearestNeighbors(n_neighbors=31).fit(y).kneighbors(y, return_distance=False)[:, 1:]
        kept = np.mean([np.isin(g.indices[g.indptr[i]:g.indptr[i + 1]], nb[i]).mean() for i in range(0, n, 3)])
        d_y = np.linalg.norm(y[sample][:, None] - y[sample][None], axis=2).ravel()
        return float(kept), float(np.corrcoef(d_lat, d_y)[0, 1])

    (kept_g, corr_g), (kept_c, corr_c) = scores(y_gpu), scores(y_cpu)
    print(f"graph neighbours kept among the 30 nearest in the plane: gpu {kept_g:.3f}, sequential oracle {kept_c:.3f}; "
          f"distance correlation with the latent sheet: gpu {corr_g:.3f}, oracle {corr_c:.3f}")
    assert np.isfinite(y_gpu).all() and kept_c > 0.9 and kept_g >= kept_c - 0.05 and corr_g >= corr_c - 0.05


@pytest.mark.parametrize("n_epochs", [5, 200, 500])
def test_device_pruning_equals_host_pruning(pbmc68k, n_epochs):
    import torch

    from scanpy_amd.tools import _umap

    g = pbmc68k["connectivities"].astype(np.float32).tocsr()
    csr, eps = _umap._prune_and_schedule(g, n_epochs)
    dev = torch.device("cuda")
    ip, ix, w, eps_d = _umap.prune_and_schedule_device(torch.from_numpy(g.indptr.astype(np.int64)).to(dev),
                                                       torch.from_numpy(g.indices.astype(np.int32)).to(dev),
                                                       torch.from_numpy(g.data).to(dev), g.shape[0], n_epochs)
    assert np.array_equal(ip.cpu().numpy(), csr.indptr) and np.array_equal(ix.cpu().numpy(), csr.indices)
    np.testing.assert_allclose(eps_d.cpu().numpy(), eps, rtol=1e-6)
    np.testing.assert_array_equal(w.cpu().numpy(), csr.data)


@pytest.mark.parametrize("dim", [2, 3])
def test_spectral_init_on_the_device_equals_arpack(dim, monkeypatch):
    """`init_pos='spectral'` (src/scanpy/tools/_umap.py:165-215 -> umap-learn's spectral_layout, ARPACK on the normalised
    Laplacian): `scamd_spectral_embedding_f32` -- Chebyshev-filtered subspace iteration on the kernels of csrc/dense.hip --
    returns the same eigen-SPACE as scipy's ARPACK on a graph WITHOUT clusters (a 30k-vertex sheet: eigenvalues 1 - O(1e-4), the
    case a block power iteration cannot do), orthonormal and orthogonal to the trivial eigenvector; and `tl.umap` reaches no
    torch.linalg routine on the way (round 5: QR / Cholesky / eigh of rocSOLVER)."""
    import torch
    from scipy.sparse.linalg import eigsh
    from sklearn.neighbors import kneighbors_graph

    from scanpy_amd import _kernels as K

    rng = np.random.default_rng(0)
    n = 30000
    pts = rng.uniform(size=(n, 2)) * [3.0, 1.0]
    g = kneighbors_graph(pts, 12, mode="distance")
    g.data = np.exp(-g.data / g.data.mean())
    a = (g + g.T).tocsr().astype(np.float32)
    a.sort_indices()
    dev = torch.device("cuda")
    out, info = K.spectral_embedding(torch.from_numpy(a.indptr.astype(np.int64)).to(dev), torch.from_numpy(a.indices.astype(np.int32)).to(dev),
                                     torch.from_numpy(a.data).to(dev), n, dim, seed=0)
    v = out.cpu().numpy()
    print(info)
    assert info["converged"] and info["residual"] < 2e-6
    deg = np.asarray(a.sum(1)).ravel().astype(np.float64)
    dis = 1.0 / np.sqrt(deg)
    s_mat = sparse.diags(dis) @ a.astype(np.float64) @ sparse.diags(dis)
    lam, vec = eigsh(s_mat, k=dim + 1, which="LA", tol=1e-10)
    order = np.argsort(-lam)
    ref = vec[:, order][:, 1:dim + 1]
    cosines = np.linalg.svd(ref.T @ v, compute_uv=False)
    print("principal cosines", cosines, "eigenvalues", lam[order], info["ritz_values"])
    # (an angle of residual / eigenvalue gap: 2e-6 / 1e-4 on this sheet -- 1 - cos ~ 1e-5 measured; a power iteration gives 0.87)
    assert cosines.min() > 1 - 1e-4
    assert np.abs(v.T @ v - np.eye(dim)).max() < 1e-10
    assert np.abs(v.T @ (np.sqrt(deg) / np.linalg.norm(np.sqrt(deg)))).max() < 1e-6
    np.testing.assert_allclose(info["ritz_values"][:dim], lam[order][1:dim + 1], atol=2e-6)

    def refuse(*args, **kwargs):
        raise AssertionError("a torch.linalg routine was reached from tl.umap")

    for name in ("eigh", "qr", "cholesky_ex", "cholesky", "solve_triangular", "svd"):
        monkeypatch.setattr(torch.linalg, name, refuse)
    adata = sc.AnnData(sparse.csr_matrix((n, 1), dtype=np.float32))
    adata.obsp["connectivities"] = a
    adata.uns["neighbors"] = dict(connectivities_key="connectivities", distances_key="distances", params=dict(n_neighbors=12, method="umap"))
    sc.tl.umap(adata, n_components=dim, maxiter=20)
    assert adata.obsm["X_umap"].shape == (n, dim) and np.isfinite(adata.obsm["X_umap"]).all()
