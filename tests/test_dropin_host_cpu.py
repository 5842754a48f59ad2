"""The drop-in front-ends (sc.pp.pca / sc.pp.neighbors / sc.tl.leiden: signatures, slots, params, errors, key_added,
restrict_to, transformer routes) on a machine WITHOUT a GPU: the kernel layer is replaced by the CPU stand-ins the
gloo tests use (tests/dist_worker.py: sklearn brute kNN, oracle fuzzy set, oracle Leiden; tests/stub_backend.py for the
PCA data passes), so what runs here is the product's host logic, including the real PCA solver.  The same test bodies
run against the HIP kernels in tests/test_gpu_pipeline.py."""
from __future__ import annotations

import numpy as np
import pytest
import torch

import test_gpu_pipeline as gp
from dist_worker import _patch_kernels
from stub_backend import CpuStubBackend


@pytest.fixture(autouse=True)
def _cpu_kernel_layer(monkeypatch):
    from scanpy_amd import _device, _kernels
    from scanpy_amd.preprocessing import _pca_solver

    saved = {name: getattr(_kernels, name) for name in ("knn", "fuzzy_simplicial_set", "leiden", "modularity")}
    monkeypatch.setattr(_device, "require_gpu", lambda: torch.device("cpu"))
    monkeypatch.setattr(_pca_solver, "GpuBackend", CpuStubBackend)
    _patch_kernels()

    def modularity(indptr, indices, weights, n, membership, *, resolution=1.0):
        from scipy import sparse

        from oracle import leiden as ol

        adj = sparse.csr_matrix((weights.numpy(), indices.numpy(), indptr.numpy()), shape=(n, n))
        return ol.modularity(adj, membership.numpy(), resolution=resolution)

    _kernels.modularity = modularity

    def _graph_tensors(m):
        m = m.tocsr()
        m.sort_indices()
        return (torch.from_numpy(m.indptr.astype(np.int64)), torch.from_numpy(m.indices.astype(np.int32)),
                torch.from_numpy(m.data.astype(np.float32)))

    def gauss_connectivities(knn_idx, knn_dist):
        from oracle import connectivities as oc

        return _graph_tensors(oc.gauss_knn(knn_idx.numpy(), knn_dist.numpy(), knn_idx.shape[0]))

    def jaccard_connectivities(knn_idx):
        from oracle import connectivities as oc

        return _graph_tensors(oc.jaccard_knn(knn_idx.numpy(), knn_idx.shape[0], knn_idx.shape[1]))

    saved.update({name: getattr(_kernels, name) for name in ("gauss_connectivities", "jaccard_connectivities")})
    _kernels.gauss_connectivities = gauss_connectivities
    _kernels.jaccard_connectivities = jaccard_connectivities
    yield
    for name, fn in saved.items():
        setattr(_kernels, name, fn)


@pytest.fixture(scope="module")
def sc():
    import scanpy_amd

    return scanpy_amd


@pytest.mark.parametrize("fmt", ["csr", "dense"])
def test_pca_transform_golden(sc, pca_toy, fmt):
    gp.test_pca_transform_golden(sc, pca_toy, fmt)


def test_pca_no_zero_center_golden(sc, pca_toy):
    gp.test_pca_no_zero_center_golden(sc, pca_toy)


def test_pca_randomized_sparse_warns(sc, pca_toy):
    gp.test_pca_randomized_sparse_warns(sc, pca_toy)


def test_pca_shapes_and_errors(sc, pca_toy):
    gp.test_pca_shapes_and_errors(sc, pca_toy)


def test_pca_real_counts_layer_and_mask(sc, pbmc68k):
    gp.test_pca_real_counts_layer_and_mask(sc, pbmc68k)


def test_pca_key_added_and_copy(sc, pca_toy):
    gp.test_pca_key_added_and_copy(sc, pca_toy)


def test_neighbors_toy_golden(sc, neighbors_toy):
    gp.test_neighbors_toy_golden(sc, neighbors_toy)


def test_neighbors_key_added_use_rep_n_pcs(sc, pbmc68k):
    gp.test_neighbors_key_added_use_rep_n_pcs(sc, pbmc68k)


def test_neighbors_transformer_plugin_route(sc, pbmc68k):
    gp.test_neighbors_transformer_plugin_route(sc, pbmc68k)


def test_neighbors_precomputed_distances(sc, pbmc68k):
    gp.test_neighbors_precomputed_distances(sc, pbmc68k)


def test_neighbors_auto_pca_fallback(sc, pbmc68k):
    gp.test_neighbors_auto_pca_fallback(sc, pbmc68k)


def test_leiden_basic_and_params(sc, pbmc68k):
    gp.test_leiden_basic_and_params(sc, pbmc68k)


def test_leiden_errors(sc, pbmc68k):
    gp.test_leiden_errors(sc, pbmc68k)


def test_leiden_restrict_to_and_keys(sc, pbmc68k):
    gp.test_leiden_restrict_to_and_keys(sc, pbmc68k)


@pytest.mark.parametrize("chunk_size", [333, 2000])
def test_pca_chunked_equals_one_shot(sc, pbmc68k, chunk_size, monkeypatch):
    gp.test_pca_chunked_equals_one_shot(sc, pbmc68k, chunk_size, "0", monkeypatch)


# ---- more of the reference's tests/test_pca.py semantics, on the host logic (kernel layer stubbed) ------------------
from scipy import sparse  # noqa: E402

A_LIST = np.array([[0, 0, 7, 0, 0], [8, 5, 0, 2, 0], [6, 0, 0, 2, 5], [0, 0, 0, 1, 0], [8, 8, 2, 1, 0], [0, 0, 0, 4, 5]],
                  dtype=np.float32)  # tests/test_pca.py:34-41
ARRAY_TYPES = [pytest.param(lambda a: np.array(a), id="dense"), pytest.param(sparse.csr_matrix, id="csr")]


@pytest.mark.parametrize("typ", ARRAY_TYPES)
def test_mask_var_error(sc, typ):
    """tests/test_pca.py:406-413"""
    with pytest.raises(ValueError, match=r"Did not find `adata\.var\['highly_variable'\]`\."):
        sc.pp.pca(sc.AnnData(typ(A_LIST)), mask_var="highly_variable")


def test_mask_length_and_obsm_errors(sc):
    """tests/test_pca.py:416-436"""
    adata = sc.AnnData(A_LIST.copy())
    with pytest.raises(ValueError, match=r"The shape of the mask do not match the data\."):
        sc.pp.pca(adata, mask_var=np.ones(adata.shape[1] + 1, dtype=bool), copy=True)
    adata.obsm["X_alt"] = A_LIST.copy()
    for mask in ("highly_variable", np.array([True, False, True, True, False])):
        with pytest.raises(ValueError, match=r"Argument `mask_var` is incompatible with `obsm`."):
            sc.pp.pca(adata, mask_var=mask, obsm="X_alt", copy=True)


@pytest.mark.parametrize("typ", ARRAY_TYPES)
def test_mask_var_argument_equivalence_and_masked_loadings(sc, typ):
    """tests/test_pca.py:439-481: mask as array == mask as column name; masked genes get zero loadings; the embedding equals
    the PCA of the subset matrix"""
    rng = np.random.default_rng(0)
    x = rng.random((100, 10)).astype(np.float32)
    mask = np.array([1, 1, 0, 1, 1, 0, 1, 1, 1, 0], dtype=bool)
    a1, a2, a3 = sc.AnnData(typ(x)), sc.AnnData(typ(x)), sc.AnnData(typ(x[:, mask]))
    sc.pp.pca(a1, n_comps=4, mask_var=mask)
    a2.var["mask"] = mask
    sc.pp.pca(a2, n_comps=4, mask_var="mask")
    sc.pp.pca(a3, n_comps=4)
    np.testing.assert_array_equal(a1.obsm["X_pca"], a2.obsm["X_pca"])
    assert a2.uns["pca"]["params"]["mask_var"] == "mask"
    assert not a1.varm["PCs"][~mask].any()
    np.testing.assert_array_equal(a1.obsm["X_pca"], a3.obsm["X_pca"])
    np.testing.assert_allclose(a1.varm["PCs"][mask], a3.varm["PCs"], rtol=1e-10)


@pytest.mark.parametrize("typ", ARRAY_TYPES)
def test_mask_defaults(sc, typ):
    """tests/test_pca.py:484-506: `var['highly_variable']` is the default mask; `mask_var=None` switches it off"""
    adata = sc.AnnData(typ(A_LIST))
    without_var = sc.pp.pca(adata, n_comps=3, copy=True)
    adata.var["highly_variable"] = np.array([True, True, False, True, True])
    with_var = sc.pp.pca(adata, n_comps=3, copy=True)
    assert without_var.uns["pca"]["params"]["mask_var"] is None
    assert with_var.uns["pca"]["params"]["mask_var"] == "highly_variable"
    assert not np.array_equal(without_var.obsm["X_pca"], with_var.obsm["X_pca"])
    with_no_mask = sc.pp.pca(adata, n_comps=3, mask_var=None, copy=True)
    np.testing.assert_array_equal(without_var.obsm["X_pca"], with_no_mask.obsm["X_pca"])


@pytest.mark.parametrize("rep", ["layer", "obsm"])
def test_pca_rep(sc, pbmc68k, rep):
    """tests/test_pca.py:509-540: `layer=` / `obsm=` work like `X`; `obsm` stores the components in `uns`"""
    x = pbmc68k["counts"].astype(np.float32)[:200]
    adata = sc.AnnData(x.copy())
    rep_adata = sc.AnnData(x.copy())
    if rep == "layer":
        rep_adata.layers["counts"] = x.copy()
    else:
        rep_adata.obsm["counts"] = x.copy()[:, :100]
        adata = sc.AnnData(x.copy()[:, :100])
    rep_adata.X = sparse.csr_matrix(x.shape, dtype=np.float32)  # X must not be what gets decomposed
    sc.pp.pca(adata, n_comps=10, mask_var=None)
    sc.pp.pca(rep_adata, n_comps=10, **{rep: "counts"}, mask_var=None)
    assert rep_adata.uns["pca"]["params"][rep] == "counts" and rep not in adata.uns["pca"]["params"]
    np.testing.assert_array_equal(adata.uns["pca"]["variance"], rep_adata.uns["pca"]["variance"])
    np.testing.assert_array_equal(adata.uns["pca"]["variance_ratio"], rep_adata.uns["pca"]["variance_ratio"])
    np.testing.assert_array_equal(adata.obsm["X_pca"], rep_adata.obsm["X_pca"])
    pcs = rep_adata.varm["PCs"] if rep == "layer" else rep_adata.uns["pca"]["components"]
    np.testing.assert_array_equal(adata.varm["PCs"], pcs)


def test_pca_n_pcs_with_renamed_representation(sc, pbmc68k):
    """tests/test_pca.py:389-401: `n_pcs` applies to any `use_rep`"""
    adata = sc.AnnData(pbmc68k["counts"].astype(np.float32))
    sc.pp.pca(adata, n_comps=20, dtype=np.float64)
    adata.obsm["X_pca_test"] = adata.obsm["X_pca"]
    original = sc.pp.neighbors(adata, n_pcs=5, use_rep="X_pca", copy=True)
    renamed = sc.pp.neighbors(adata, n_pcs=5, use_rep="X_pca_test", copy=True)
    assert np.allclose(original.obsp["distances"].toarray(), renamed.obsp["distances"].toarray())
    assert original.uns["neighbors"]["params"]["n_pcs"] == 5


# ---- more of the reference's tests/test_clustering.py semantics -------------------------------------------------------
@pytest.mark.parametrize("rng_arg", ["rng", "random_state"])
@pytest.mark.parametrize("flavor", ["igraph", "leidenalg"])
def test_leiden_random_state(sc, pbmc68k, flavor, rng_arg):
    """tests/test_clustering.py:67-102: same seed -> same labels and modularity, another seed -> another clustering"""
    import pandas as pd

    base = gp._graph_adata(sc, pbmc68k)
    runs = [sc.tl.leiden(base, flavor=flavor, copy=True, directed=(flavor == "leidenalg"),
                         n_iterations=2 if flavor == "leidenalg" else -1, **{rng_arg: seed}) for seed in (1, 1, 42, 7, 99)]
    pd.testing.assert_series_equal(runs[0].obs["leiden"], runs[1].obs["leiden"])
    assert runs[0].uns["leiden"]["modularity"] == runs[1].uns["leiden"]["modularity"]
    # the seed reaches the optimiser: some other seed gives another clustering (on a 700-cell graph two seeds can agree)
    assert any(not r.obs["leiden"].equals(runs[1].obs["leiden"]) for r in runs[2:])
    assert ("random_state" in runs[0].uns["leiden"]["params"]) == (rng_arg == "random_state")


def test_clustering_subset(sc, pbmc68k):
    """tests/test_clustering.py:177-213: `restrict_to` re-clusters only the chosen cluster's cells"""
    adata = gp._graph_adata(sc, pbmc68k)
    sc.tl.leiden(adata, flavor="igraph", key_added="leiden")
    for c in adata.obs["leiden"].unique()[:4]:
        cells_in_c = adata.obs["leiden"] == c
        n_in_c = int(cells_in_c.sum())
        sc.tl.leiden(adata, flavor="igraph", restrict_to=("leiden", [c]), key_added="leiden_sub")
        new = adata.obs["leiden_sub"]
        counts = new[cells_in_c].value_counts()
        assert counts.sum() == n_in_c
        nonzero = counts[counts > 0].index
        assert len(nonzero.intersection(adata.obs["leiden"].cat.categories)) == 0  # only new category names inside c
        assert (new[~cells_in_c].astype(str) == adata.obs["leiden"][~cells_in_c].astype(str)).all()


def test_clustering_custom_key_and_objective(sc, pbmc68k):
    """tests/test_clustering.py:166-242: parameters are stored under the user's key and never overwritten"""
    adata = gp._graph_adata(sc, pbmc68k)
    sc.tl.leiden(adata, flavor="igraph", resolution=0.8)
    for res in (0.9, 1.1):
        sc.tl.leiden(adata, flavor="igraph", resolution=res, key_added=f"leiden_{res}")
    assert adata.uns["leiden"]["params"]["resolution"] == 0.8
    for res in (0.9, 1.1):
        assert adata.uns[f"leiden_{res}"]["params"]["resolution"] == res
    sc.tl.leiden(adata, objective_function="modularity", flavor="igraph", directed=False)
    with pytest.raises(ValueError, match='must be "CPM" or "modularity"'):  # (igraph's message)
        sc.tl.leiden(adata, objective_function="surprise", flavor="igraph")
    with pytest.raises(TypeError, match="objective_function is igraph's argument"):
        with pytest.warns(FutureWarning):
            sc.tl.leiden(adata, objective_function="CPM")  # default flavor leidenalg: takes a partition class instead


def test_partition_type_resolution():
    """`partition_type` of the leidenalg flavor (src/scanpy/tools/_leiden.py:107-110, 174-186) -> (objective, resolution):
    classes matched by name (leidenalg is absent here; a user's `leidenalg.CPMVertexPartition` arrives as the class),
    `resolution=None` = the class's default, a class without a resolution parameter refuses one as its constructor would"""
    from scanpy_amd.tools._leiden import _resolve_partition_type as r

    cls = lambda name: type(name, (), {})  # noqa: E731
    assert r(None, 0.7) == ("modularity", 0.7) and r(None, None) == ("modularity", 1.0)
    assert r(cls("RBConfigurationVertexPartition"), 2.0) == ("modularity", 2.0)
    assert r("CPMVertexPartition", 0.01) == ("cpm", 0.01) and r(cls("CPMVertexPartition"), None) == ("cpm", 1.0)
    assert r(cls("ModularityVertexPartition"), None) == ("modularity", 1.0)
    with pytest.raises(TypeError, match="resolution_parameter"):
        r(cls("ModularityVertexPartition"), 1.0)
    assert r(cls("RBERVertexPartition"), 0.5) == ("cpm_density", 0.5)  # (CPM at resolution x the graph's density: tl.leiden)
    for name in ("SignificanceVertexPartition", "SurpriseVertexPartition"):
        with pytest.raises(NotImplementedError, match=name):
            r(cls(name), 1.0)


# ---- tests/test_neighbors_key_added.py semantics ------------------------------------------------------------------------
@pytest.mark.parametrize("rng_arg", ["rng", "random_state"])
def test_neighbors_key_added_and_downstream_keys(sc, pbmc68k, rng_arg):
    """tests/test_neighbors_key_added.py:35-52, 66-100: `key_added` renames the three slots; `neighbors_key=` / `obsp=` of
    tl.leiden find them"""
    key = "test"
    adata = sc.AnnData(pbmc68k["X"], obsm={"X_pca": pbmc68k["X_pca"]})
    sc.pp.neighbors(adata, n_neighbors=5, **{rng_arg: 0})
    sc.pp.neighbors(adata, n_neighbors=5, **{rng_arg: 0}, key_added=key)
    conns_key, dists_key = adata.uns[key]["connectivities_key"], adata.uns[key]["distances_key"]
    assert (conns_key, dists_key) == (f"{key}_connectivities", f"{key}_distances")
    assert adata.uns["neighbors"]["params"] == adata.uns[key]["params"]
    assert ("random_state" in adata.uns[key]["params"]) == (rng_arg == "random_state")
    assert np.allclose(adata.obsp["connectivities"].toarray(), adata.obsp[conns_key].toarray())
    assert np.allclose(adata.obsp["distances"].toarray(), adata.obsp[dists_key].toarray())
    sc.tl.leiden(adata, flavor="igraph", **{rng_arg: 0})
    for arg in ({"neighbors_key": key}, {"obsp": conns_key}):
        other = adata.copy()
        sc.tl.leiden(other, flavor="igraph", **{rng_arg: 0}, **arg)
        assert adata.uns["leiden"]["params"] == other.uns["leiden"]["params"]
        assert np.all(adata.obs["leiden"] == other.obs["leiden"])


def test_neighbors_without_previous_pca_run(sc, pbmc68k):
    """tests/test_neighbors_key_added.py:55-63"""
    adata = sc.AnnData(pbmc68k["X"].copy())
    with pytest.warns(UserWarning, match=r".*Falling back to preprocessing with `sc.pp.pca` and default params"):
        sc.pp.neighbors(adata, n_neighbors=5, random_state=0)
    assert "pca" in adata.uns and adata.obsm["X_pca"].shape == (700, 50)


# ---- tests/test_neighbors.py semantics ------------------------------------------------------------------------------
@pytest.mark.parametrize("method", ["umap", "gauss", "jaccard"])
def test_distances_and_connectivities_by_method(sc, neighbors_toy, method):
    """tests/test_neighbors.py:151-164, 196-227: the three connectivity kernels share the distances; each reproduces its
    golden matrix"""
    k = int(neighbors_toy["n_neighbors"])
    adata = sc.AnnData(neighbors_toy["X"])
    sc.pp.neighbors(adata, n_neighbors=k, method=method)
    np.testing.assert_allclose(adata.obsp["distances"].toarray(), neighbors_toy["distances_euclidean"])
    golden = {"umap": "connectivities_umap", "gauss": "connectivities_gauss_knn", "jaccard": "connectivities_jaccard"}[method]
    np.testing.assert_allclose(adata.obsp["connectivities"].toarray(), neighbors_toy[golden], rtol=1e-6)
    with pytest.raises(ValueError, match="`method` needs to be one of"):
        sc.pp.neighbors(adata, n_neighbors=k, method="rbf")


def test_use_rep_argument(sc):
    """tests/test_neighbors.py:251-261"""
    adata = sc.AnnData(np.random.default_rng(0).standard_normal((30, 300)).astype(np.float32))
    sc.pp.pca(adata, n_comps=20)
    a = sc.Neighbors(adata)
    a.compute_neighbors(n_pcs=5, use_rep="X_pca")
    b = sc.Neighbors(adata)
    b.compute_neighbors(n_pcs=5, use_rep=None)
    np.testing.assert_allclose(a.distances.toarray(), b.distances.toarray())


def test_neighbors_distance_equivalence(sc, pbmc68k):
    """tests/test_neighbors.py:275-296: `distances=` reuses a distance matrix; only `metric` differs in the params"""
    adata = sc.AnnData(pbmc68k["X"], obsm={"X_pca": pbmc68k["X_pca"]})
    adata_d = adata.copy()
    sc.pp.neighbors(adata)
    sc.pp.neighbors(adata_d, distances=adata.obsp["distances"])
    np.testing.assert_allclose(adata.obsp["connectivities"].toarray(), adata_d.obsp["connectivities"].toarray(), rtol=1e-5)
    np.testing.assert_allclose(adata.obsp["distances"].toarray(), adata_d.obsp["distances"].toarray(), rtol=1e-5)
    p, p_d = (ad.uns["neighbors"]["params"].copy() for ad in (adata, adata_d))
    assert p.pop("metric") == "euclidean" and p_d.pop("metric") is None and p == p_d


@pytest.mark.parametrize("chunk_size", [None, 256, 97])
@pytest.mark.parametrize("masked", [False, True])
def test_pca_streams_a_backed_zarr_matrix(sc, pbmc68k, tmp_path, monkeypatch, chunk_size, masked):
    """`read_zarr(backed='r')` leaves X on disk; `pp.pca` streams its row chunks (reader thread ->
    upload -> Gram / projection passes) and reproduces the in-memory fit bit for bit, `highly_variable` mask included"""
    from scipy import sparse

    from scanpy_amd import readwrite as rw
    from scanpy_amd._backed import BackedCsr

    monkeypatch.setattr(rw, "CHUNK_ELEMS", 5003)  # many inner chunks and several shard objects at this size
    monkeypatch.setattr(rw, "CHUNKS_PER_SHARD", 4)
    x = sparse.csr_matrix(np.maximum(pbmc68k["X"], 0).astype(np.float32))
    a = sc.AnnData(x)
    if masked:
        a.var["highly_variable"] = np.arange(x.shape[1]) % 3 != 0
    sc.write_zarr(tmp_path / "a.zarr", a)
    b = sc.read_zarr(tmp_path / "a.zarr", backed="r")
    assert isinstance(b.X, BackedCsr)
    sc.pp.pca(a, n_comps=20)
    sc.pp.pca(b, n_comps=20, chunk_size=chunk_size)
    assert isinstance(b.X, BackedCsr)  # still on disk
    np.testing.assert_array_equal(b.obsm["X_pca"], a.obsm["X_pca"])
    np.testing.assert_array_equal(b.varm["PCs"], a.varm["PCs"])
    np.testing.assert_array_equal(b.uns["pca"]["variance"], a.uns["pca"]["variance"])
    np.testing.assert_array_equal(b.uns["pca"]["variance_ratio"], a.uns["pca"]["variance_ratio"])
    with pytest.raises(ValueError, match="zero_center"):
        sc.pp.pca(b, n_comps=20, zero_center=False)


def test_pca_streams_a_backed_h5ad_matrix(sc):
    """the same through an `.h5ad` written by h5py (tests/golden/make_h5_golden.py): gzip + shuffle chunks inflated
    into recycled buffers, `highly_variable` mask applied per chunk; bit-identical to the in-memory fit"""
    from pathlib import Path

    from scanpy_amd._backed import BackedCsr

    path = Path(__file__).parent / "golden" / "h5" / "adata_layout.h5ad"
    a = sc.read_h5ad(path)
    b = sc.read_h5ad(path, backed="r")
    assert isinstance(b.X, BackedCsr) and "highly_variable" in b.var.columns
    sc.pp.pca(a, n_comps=10)
    sc.pp.pca(b, n_comps=10, chunk_size=128)
    np.testing.assert_array_equal(b.obsm["X_pca"], a.obsm["X_pca"])
    np.testing.assert_array_equal(b.varm["PCs"], a.varm["PCs"])
    assert (b.varm["PCs"][~a.var["highly_variable"].to_numpy()] == 0).all()
    np.testing.assert_array_equal(b.uns["pca"]["variance_ratio"], a.uns["pca"]["variance_ratio"])


@pytest.mark.parametrize("chunk_size", [None, 130])
def test_counts_on_disk_to_clusters_without_materialising(sc, pbmc68k, tmp_path, chunk_size, monkeypatch):
    """the whole chain out of core: counts in an .h5ad -> normalize_total -> log1p (pending transforms) ->
    highly_variable_genes (streamed) -> pca (streamed, masked) -> neighbors -> leiden; the PCA equals the in-memory chain
    bit for bit (same float32 element-wise kernels per row chunk, additive fixed-point Gram matrix)"""
    from scipy import sparse

    from scanpy_amd.preprocessing import _csr_device
    from stub_backend import CpuStubPPBackend

    monkeypatch.setattr(_csr_device, "default_backend", lambda: CpuStubPPBackend())
    counts = sparse.csr_matrix(pbmc68k["counts"]).astype(np.float32)
    a = sc.AnnData(counts.copy())
    sc.write_h5ad(tmp_path / "counts.h5ad", a)
    b = sc.read_h5ad(tmp_path / "counts.h5ad", backed="r")
    for ad in (a, b):
        sc.pp.normalize_total(ad, target_sum=1e4)
        sc.pp.log1p(ad)
        sc.pp.highly_variable_genes(ad, n_top_genes=300)
    b.var["highly_variable"] = a.var["highly_variable"].to_numpy()  # (ties at the cut may differ by a gene or two)
    sc.pp.pca(a, n_comps=15)
    sc.pp.pca(b, n_comps=15, chunk_size=chunk_size)
    assert b.X.is_backed
    np.testing.assert_array_equal(b.obsm["X_pca"], a.obsm["X_pca"])
    np.testing.assert_array_equal(b.varm["PCs"], a.varm["PCs"])
    sc.pp.neighbors(b)
    sc.tl.leiden(b, flavor="igraph", n_iterations=2)
    sc.write_h5ad(tmp_path / "out.h5ad", b)  # X with pending transforms is materialised through the device backend
    back = sc.read_h5ad(tmp_path / "out.h5ad")
    assert (back.X != a.X).nnz == 0 and list(back.obs["leiden"]) == list(b.obs["leiden"])


def test_pca_obsm_ignores_the_highly_variable_mask(monkeypatch):
    """`pp.pca(adata, obsm='rep')` with a `highly_variable` column in `.var` must not slice the obsm
    matrix with the var mask (the reference subsets the AnnData and then reads `obsm` unmasked,
    src/scanpy/preprocessing/_pca/__init__.py:228-232); `params` still records the mask."""
    import numpy as np
    import pandas as pd
    from scipy import sparse

    import scanpy_amd as sc
    from scanpy_amd.preprocessing import _pca as P
    from scanpy_amd.preprocessing import _pca_solver as S
    from stub_backend import CpuStubBackend

    monkeypatch.setattr(S, "GpuBackend", CpuStubBackend)
    rng = np.random.default_rng(0)
    x = sparse.random(60, 30, density=0.3, random_state=1, format="csr", dtype=np.float32)
    a = sc.AnnData(x)
    a.var["highly_variable"] = rng.random(30) < 0.5
    a.obsm["rep"] = rng.standard_normal((60, 12)).astype(np.float32)
    P.pca(a, n_comps=5, obsm="rep")
    assert a.obsm["X_pca"].shape == (60, 5)
    assert a.uns["pca"]["params"]["mask_var"] == "highly_variable" and a.uns["pca"]["params"]["obsm"] == "rep"
    assert a.uns["pca"]["components"].shape == (12, 5)
    assert "PCs" not in a.varm


def test_write_to_the_backing_file_keeps_the_data(tmp_path):
    """`sc.write(p, sc.read(p, backed='r'))` used to truncate the file it was streaming from"""
    import numpy as np
    from scipy import sparse

    import scanpy_amd as sc

    x = sparse.random(200, 40, density=0.2, random_state=3, format="csr", dtype=np.float32)
    for ext in ("h5ad", "zarr"):
        p = tmp_path / f"same.{ext}"
        sc.write(p, sc.AnnData(x))
        b = sc.read(p, backed="r")
        sc.write(p, b)
        again = sc.read(p)
        assert (again.X != x).nnz == 0
        assert not [f for f in tmp_path.iterdir() if ".tmp" in f.name or ".old" in f.name]


def test_anndata_copy_and_subset_carry_raw_and_varp():
    """the stand-in AnnData dropped `raw` / `varp` on copy and never subset `raw` along obs"""
    import numpy as np
    from scipy import sparse

    import scanpy_amd as sc

    x = sparse.random(30, 8, density=0.5, random_state=0, format="csr", dtype=np.float32)
    a = sc.AnnData(x)
    a.raw = sc.AnnData(sparse.random(30, 20, density=0.5, random_state=1, format="csr", dtype=np.float32), a.obs)
    a.varp["corr"] = np.eye(8)
    a.obsp["g"] = sparse.identity(30, format="csr")
    c = a.copy()
    assert c.raw is not None and c.raw.X.shape == (30, 20) and "corr" in c.varp
    keep = np.arange(30) % 3 == 0
    a._inplace_subset_obs(keep)
    assert a.raw.X.shape == (10, 20) and a.X.shape == (10, 8) and a.obsp["g"].shape == (10, 10)
    sub = c[:, np.arange(8) < 4]
    assert sub.obsp["g"].shape == (30, 30) and sub.varp["corr"].shape == (4, 4)
