"""-m gpu: the routes added after this round's GPU minutes were spent, on the real kernels for the first time -- the
out-of-core PCA (zarr / h5ad, resident and streamed), `highly_variable_genes` on a backed matrix and with
`flavor='seurat_v3'`, and the device-list route of `pp.neighbors`.  (Named to sort last: these have only ever run
against the CPU stand-in kernels, tests/test_dropin_host_cpu.py, tests/test_loess_cpu.py.)"""
from __future__ import annotations

import numpy as np
import pytest
from scipy import sparse

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sc():
    import scanpy_amd

    return scanpy_amd


def _matrix(n=6000, g=400, seed=3):
    from scanpy_amd.datasets import synthetic_planted

    return synthetic_planted(n, g, n_types=8, seed=seed)[0]


@pytest.mark.parametrize("container", ["zarr", "h5ad"])
@pytest.mark.parametrize("resident", ["1", "0"])
def test_backed_pca_equals_in_memory(sc, tmp_path, monkeypatch, container, resident):
    from scanpy_amd import readwrite as rw

    monkeypatch.setenv("SCAMD_PCA_CHUNK_RESIDENT", resident)
    monkeypatch.setattr(rw, "CHUNK_ELEMS", 50_021)
    x = _matrix()
    a = sc.AnnData(x)
    a.var["highly_variable"] = np.arange(x.shape[1]) % 4 != 0
    path = tmp_path / f"a.{container}"
    sc.write(path, a)
    b = sc.read(path, backed="r")
    assert b.X.is_backed
    sc.pp.pca(a, n_comps=30)
    sc.pp.pca(b, n_comps=30, chunk_size=1700)
    np.testing.assert_array_equal(b.obsm["X_pca"], a.obsm["X_pca"])
    np.testing.assert_array_equal(b.varm["PCs"], a.varm["PCs"])
    np.testing.assert_array_equal(b.uns["pca"]["variance_ratio"], a.uns["pca"]["variance_ratio"])
    sc.pp.neighbors(b)
    sc.tl.leiden(b, flavor="igraph", n_iterations=2)
    assert b.obs["leiden"].nunique() >= 4 and b.X.is_backed


def test_hvg_on_a_backed_matrix_and_seurat_v3(sc, tmp_path, pbmc68k):
    from oracle import preprocess as op

    counts = pbmc68k["counts"].astype(np.float32)
    a = sc.AnnData(counts)
    sc.pp.highly_variable_genes(a, flavor="seurat_v3", n_top_genes=200)
    ref = op.highly_variable_genes_seurat_v3(counts, n_top_genes=200)
    np.testing.assert_allclose(a.var["means"], ref["means"], rtol=1e-6)
    np.testing.assert_allclose(a.var["variances"], ref["variances"], rtol=1e-5)
    np.testing.assert_allclose(a.var["variances_norm"], ref["variances_norm"], rtol=1e-5, equal_nan=True)
    assert (a.var["highly_variable"].to_numpy() != ref["highly_variable"].to_numpy()).sum() <= 2
    batch = np.arange(counts.shape[0]) % 2
    a.obs["batch"] = batch
    df = sc.pp.highly_variable_genes(a, flavor="seurat_v3_paper", n_top_genes=150, batch_key="batch", inplace=False)
    refb = op.highly_variable_genes_seurat_v3(counts, n_top_genes=150, batch=batch, flavor="seurat_v3_paper")
    np.testing.assert_allclose(df["variances_norm"], refb["variances_norm"], rtol=1e-5, equal_nan=True)
    # the classic flavor, streamed from disk
    logn = sparse.csr_matrix(pbmc68k["raw_X"]).astype(np.float32)
    mem = sc.AnnData(logn)
    sc.write_zarr(tmp_path / "l.zarr", mem)
    disk = sc.read_zarr(tmp_path / "l.zarr", backed="r")
    sc.pp.highly_variable_genes(mem, n_top_genes=100)
    sc.pp.highly_variable_genes(disk, n_top_genes=100)
    np.testing.assert_allclose(disk.var["dispersions_norm"], mem.var["dispersions_norm"], rtol=1e-5, atol=1e-7,
                               equal_nan=True)


def test_neighbors_device_list_route(sc, pbmc68k):
    """the lists stay on the device between the search and the connectivity kernel; the slots must be what the
    host-side conventions (`_common.py:35-61`) build from the same search"""
    from scanpy_amd.neighbors._common import get_sparse_matrix_from_indices_distances
    from scanpy_amd.neighbors._transformer import knn_search

    a = sc.AnnData(pbmc68k["X"])
    a.obsm["X_pca"] = pbmc68k["X_pca"].astype(np.float32)
    sc.pp.neighbors(a, n_neighbors=12, use_rep="X_pca")
    idx, dist = knn_search(a.obsm["X_pca"], 12)
    want = get_sparse_matrix_from_indices_distances(idx, dist, keep_self=False)
    got = a.obsp["distances"]
    assert got.shape == want.shape and got.dtype == np.float64 and got.indices.dtype == got.indptr.dtype
    np.testing.assert_array_equal(got.indptr, want.indptr)
    np.testing.assert_array_equal(got.indices, want.indices)
    np.testing.assert_array_equal(got.data, want.data)
    conn = a.obsp["connectivities"]
    chk = conn.copy()
    chk.has_sorted_indices = False  # force scipy to look: the recorded flag must be true
    chk.sort_indices()
    assert (chk != conn).nnz == 0 and np.array_equal(chk.indices, conn.indices)
    assert (abs(conn - conn.T) > 1e-7).nnz == 0
    sc.tl.leiden(a, flavor="igraph", n_iterations=2)
    assert a.obs["leiden"].cat.categories.tolist() == [str(i) for i in range(a.obs["leiden"].nunique())]


def test_counts_on_disk_to_clusters(sc, tmp_path, pbmc68k):
    """normalize_total / log1p as pending transforms of an on-disk count matrix, streamed HVG and PCA: the PCA equals
    the in-memory chain bit for bit (tests/test_dropin_host_cpu.py runs the same body on the CPU stand-ins)"""
    counts = sparse.csr_matrix(pbmc68k["counts"]).astype(np.float32)
    a = sc.AnnData(counts.copy())
    sc.write_h5ad(tmp_path / "counts.h5ad", a)
    b = sc.read_h5ad(tmp_path / "counts.h5ad", backed="r")
    for ad in (a, b):
        sc.pp.normalize_total(ad, target_sum=1e4, key_added="nf")
        sc.pp.log1p(ad)
        sc.pp.highly_variable_genes(ad, n_top_genes=300)
    np.testing.assert_array_equal(b.obs["nf"].to_numpy(), a.obs["nf"].to_numpy())
    np.testing.assert_allclose(b.var["dispersions_norm"], a.var["dispersions_norm"], rtol=1e-5, atol=1e-7, equal_nan=True)
    b.var["highly_variable"] = a.var["highly_variable"].to_numpy()
    sc.pp.pca(a, n_comps=15)
    sc.pp.pca(b, n_comps=15, chunk_size=130)
    assert b.X.is_backed
    np.testing.assert_array_equal(b.obsm["X_pca"], a.obsm["X_pca"])
    np.testing.assert_array_equal(b.varm["PCs"], a.varm["PCs"])
    assert (b.X.to_memory() != a.X).nnz == 0


def test_pca_overlapped_upload_equals_plain_upload(monkeypatch):
    """`pp.pca` of a large HOST matrix uploads the value array first (max|x| fixes the fixed-point scale) and the column indices in
    row chunks under the Gram kernel of the chunk before (`_pca_solver._HostCsrOverlapped`: the host-to-host metric of SURVEY
    8(d)).  Integer sums: every output equals the plain upload's bit for bit -- also with empty rows, one chunk, many chunks."""
    import scanpy_amd as sc
    from scanpy_amd.datasets import synthetic_planted
    from scanpy_amd.preprocessing import _pca_solver

    x, _ = synthetic_planted(30000, 900, n_types=20, seed=11)
    x = x.tolil()
    x[5, :] = 0  # an empty row, and an empty stretch at the end
    x[29990:, :] = 0
    x = x.tocsr().astype(np.float32)
    x.eliminate_zeros()
    monkeypatch.setenv("SCAMD_PCA_OVERLAP_UPLOAD", "0")
    ref = sc.AnnData(x.copy())
    sc.pp.pca(ref, n_comps=30)
    monkeypatch.delenv("SCAMD_PCA_OVERLAP_UPLOAD")
    monkeypatch.setenv("SCAMD_PCA_OVERLAP_MIN_NNZ", "0")
    seen = []
    orig = _pca_solver._HostCsrOverlapped.__init__

    for n_chunks in (1, 6, 37):
        def init(self, x_csr, n_chunks_=6, _n=n_chunks):
            orig(self, x_csr, _n)
            seen.append(self.n_chunks)

        monkeypatch.setattr(_pca_solver._HostCsrOverlapped, "__init__", init)
        a = sc.AnnData(x.copy())
        sc.pp.pca(a, n_comps=30)
        np.testing.assert_array_equal(a.obsm["X_pca"], ref.obsm["X_pca"])
        np.testing.assert_array_equal(a.varm["PCs"], ref.varm["PCs"])
        np.testing.assert_array_equal(a.uns["pca"]["variance"], ref.uns["pca"]["variance"])
    assert seen == [1, 6, 37]


def test_deferred_distances_download_equals_direct():
    """`pp.neighbors` downloads `distances` on a side stream while the connectivity kernels run (neighbors/_common.py:
    sparse_distances_from_device(deferred=True)); at a size that takes the page-locked route (>= 8 MB per array) the matrix
    equals the one the blocking download builds, and the connectivities are those of the same lists"""
    import torch

    import scanpy_amd as sc
    from scanpy_amd.neighbors._common import sparse_distances_from_device
    from scanpy_amd.neighbors._transformer import knn_search_device

    rng = np.random.default_rng(3)
    n = 120_000
    emb = rng.normal(size=(n, 12)).astype(np.float32)
    adata = sc.AnnData(sparse.csr_matrix((n, 1), dtype=np.float32))
    adata.obsm["X_pca"] = emb
    sc.pp.neighbors(adata, n_neighbors=15, use_rep="X_pca")  # (one variable in X: the default representation would be X itself)
    idx, dist = knn_search_device(emb, 15)
    direct = sparse_distances_from_device(idx, dist)
    d = adata.obsp["distances"]
    assert d.nnz == direct.nnz == n * 14
    np.testing.assert_array_equal(d.indices, direct.indices)
    np.testing.assert_array_equal(d.data, direct.data)
    np.testing.assert_array_equal(d.indptr, direct.indptr)
    c = adata.obsp["connectivities"]
    assert c.shape == (n, n) and abs(c - c.T).max() < 1e-7 and c.nnz >= d.nnz
    torch.cuda.synchronize()
