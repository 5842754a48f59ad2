"""End-to-end parity of the drop-in API (sc.pp.pca -> sc.pp.neighbors -> sc.tl.leiden) on the GPU against the
CPU oracle and the reference's golden vectors.  Mirrors tests/test_pca.py, tests/test_neighbors.py,
tests/test_neighbors_key_added.py and tests/test_clustering.py of the reference where they touch the path."""
from __future__ import annotations

import warnings

import numpy as np
import pandas as pd
import pytest
from scipy import sparse
from sklearn.metrics import adjusted_rand_score

from helpers import knn_sets_equal_mod_ties
from oracle import connectivities as oc
from oracle import knn as oknn
from oracle import leiden as ol
from oracle import pca as opca

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sc():
    import scanpy_amd

    return scanpy_amd


# ---------------------------------------------------------------------------------------- pca ------
@pytest.mark.parametrize("fmt", ["csr", "dense"])
def test_pca_transform_golden(sc, pca_toy, fmt):
    """tests/test_pca.py:225-233."""
    a = pca_toy["A_list"].astype("float32")
    adata = sc.AnnData(sparse.csr_matrix(a) if fmt == "csr" else a)
    sc.pp.pca(adata, n_comps=4, zero_center=True, dtype="float64")
    assert np.linalg.norm(np.abs(pca_toy["A_pca"][:, :4]) - np.abs(adata.obsm["X_pca"])) < 2e-05
    assert adata.obsm["X_pca"].dtype == np.float64
    assert adata.varm["PCs"].shape == (5, 4)
    assert set(adata.uns["pca"]) == {"params", "variance", "variance_ratio"}
    assert adata.uns["pca"]["params"] == {"zero_center": True, "mask_var": None}


def test_pca_no_zero_center_golden(sc, pca_toy):
    """tests/test_pca.py:264-274."""
    adata = sc.AnnData(sparse.csr_matrix(pca_toy["A_list"].astype("float32")))
    sc.pp.pca(adata, n_comps=4, zero_center=False, dtype="float64", random_state=14)
    assert np.linalg.norm(np.abs(pca_toy["A_svd"][:, :4]) - np.abs(adata.obsm["X_pca"])) < 2e-05


def test_pca_randomized_sparse_warns(sc, pca_toy):
    """tests/test_pca.py:236-261: sparse + 'randomized' warns and still gives the exact answer."""
    adata = sc.AnnData(sparse.csr_matrix(pca_toy["A_list"].astype("float32")))
    with pytest.warns(UserWarning, match=r"Ignoring.*'randomized"):
        sc.pp.pca(adata, n_comps=4, svd_solver="randomized", dtype="float64", random_state=14)
    assert np.linalg.norm(np.abs(pca_toy["A_pca"][:, :4]) - np.abs(adata.obsm["X_pca"])) < 2e-05


def test_pca_shapes_and_errors(sc, pca_toy):
    """tests/test_pca.py:277-296."""
    rng = np.random.default_rng(0)
    adata = sc.AnnData(rng.standard_normal((30, 20)).astype(np.float32))
    sc.pp.pca(adata)
    assert adata.obsm["X_pca"].shape == (30, 19)  # min_dim - 1
    adata = sc.AnnData(rng.standard_normal((20, 30)).astype(np.float32))
    sc.pp.pca(adata)
    assert adata.obsm["X_pca"].shape == (20, 19)
    with pytest.raises(ValueError, match=r"n_components=100 must be between 1 and.*20 with svd_solver='arpack'"):
        sc.pp.pca(adata, n_comps=100)
    x = sc.pp.pca(pca_toy["A_list"].astype("float32"), n_comps=3)
    assert isinstance(x, np.ndarray) and x.shape == (6, 3) and x.dtype == np.float32
    out = sc.pp.pca(pca_toy["A_list"].astype("float32"), n_comps=3, return_info=True)
    assert len(out) == 4 and out[1].shape == (3, 5)


def test_pca_synthetic_vs_reference(sc):
    """north_star gate: loadings within 1e-4 up to sign vs reference Scanpy (= sklearn arpack)."""
    from scanpy_amd.datasets import synthetic_planted

    x, _ = synthetic_planted(20000, 2000, seed=0)
    adata = sc.AnnData(x)
    sc.pp.pca(adata)
    ref = opca.pca_reference(x, 50)
    pcs = adata.varm["PCs"]
    assert pcs.shape == (2000, 50) and adata.obsm["X_pca"].shape == (20000, 50)
    err = np.abs(np.abs(pcs.T) - np.abs(ref["components"])).max()
    print("max |loading| diff vs sklearn arpack:", err)
    assert err < 1e-4
    np.testing.assert_allclose(adata.uns["pca"]["variance"], ref["variance"], rtol=1e-4)
    np.testing.assert_allclose(adata.uns["pca"]["variance_ratio"], ref["variance_ratio"], rtol=1e-4)
    sgn = np.sign((pcs.T * ref["components"]).sum(1))
    assert np.abs(adata.obsm["X_pca"] * sgn[None, :] - ref["X_pca"]).max() < 2e-3
    assert (sgn > 0).all(), "same svd_flip sign convention as sklearn"


def test_pca_real_counts_layer_and_mask(sc, pbmc68k):
    """Real CSR input (bundled counts layer) + mask_var semantics (tests/test_pca.py:461-506)."""
    counts = pbmc68k["counts"].astype(np.float32)
    x = counts.copy()
    x.data = np.log1p(x.data)
    var = pd.DataFrame({"highly_variable": pbmc68k["highly_variable"]}, index=[f"g{i}" for i in range(x.shape[1])])
    adata = sc.AnnData(x, var=var)
    sc.pp.pca(adata, n_comps=30)
    mask = pbmc68k["highly_variable"]
    assert adata.uns["pca"]["params"]["mask_var"] == "highly_variable"
    assert (adata.varm["PCs"][~mask] == 0).all() and adata.varm["PCs"].shape == (x.shape[1], 30)
    ref = opca.pca_reference(x[:, mask], 30)
    err = np.abs(np.abs(adata.varm["PCs"][mask].T) - np.abs(ref["components"]))
    # trailing components of real data can be nearly degenerate: compare the well separated ones strictly
    gaps = -np.diff(ref["variance"]) / ref["variance"][:-1]
    sep = gaps > 1e-2  # sep[i]: component i is separated from component i+1
    ok = np.concatenate([[True], sep[:-1]]) & sep  # both neighbours separated (components 0..28)
    err = err[:-1]
    print("components compared strictly:", ok.sum(), "max err", err[ok].max())
    assert err[ok].max() < 1e-4
    np.testing.assert_allclose(adata.uns["pca"]["variance"], ref["variance"], rtol=1e-4)
    # explicit mask argument equals the subset run
    ad2 = sc.AnnData(x)
    sc.pp.pca(ad2, n_comps=30, mask_var=mask)
    np.testing.assert_array_equal(ad2.obsm["X_pca"], adata.obsm["X_pca"])
    with pytest.raises(ValueError, match="incompatible with `obsm`"):
        sc.pp.pca(ad2, mask_var=mask, obsm="X_pca")


def test_pca_reproducible(sc, pbmc68k):
    """tests/test_pca.py:333-354."""
    x = pbmc68k["counts"].astype(np.float32)
    a = sc.pp.pca(x, n_comps=10, random_state=42)
    b = sc.pp.pca(x, n_comps=10, random_state=42)
    np.testing.assert_array_equal(a, b)


def test_pca_key_added_and_copy(sc, pca_toy):
    adata = sc.AnnData(sparse.csr_matrix(pca_toy["A_list"].astype("float32")))
    out = sc.pp.pca(adata, n_comps=3, key_added="mypca", copy=True)
    assert "mypca" in out.obsm and "mypca" in out.varm and "mypca" in out.uns
    assert not adata.obsm and not adata.uns


# ------------------------------------------------------------------------------------ neighbors ----
def test_neighbors_toy_golden(sc, neighbors_toy):
    """tests/test_neighbors.py:151-226 (distances_euclidean, connectivities_umap)."""
    adata = sc.AnnData(neighbors_toy["X"].astype(np.float32))
    sc.pp.neighbors(adata, n_neighbors=int(neighbors_toy["n_neighbors"]))
    np.testing.assert_allclose(adata.obsp["distances"].toarray(), neighbors_toy["distances_euclidean"], rtol=1e-6)
    np.testing.assert_allclose(adata.obsp["connectivities"].toarray(), neighbors_toy["connectivities_umap"], rtol=1e-5, atol=1e-6)
    u = adata.uns["neighbors"]
    assert u["connectivities_key"] == "connectivities" and u["distances_key"] == "distances"
    assert u["params"] == {"n_neighbors": 3, "method": "umap", "random_state": 0, "metric": "euclidean"}


def test_neighbors_fixture_vs_oracle(sc, pbmc68k):
    x = pbmc68k["X_pca"]
    adata = sc.AnnData(pbmc68k["X"], obsm={"X_pca": x})
    sc.pp.neighbors(adata, n_neighbors=10)
    oi, od, odist = oknn.knn_sklearn(x, 10)
    d = adata.obsp["distances"]
    assert d.nnz == 700 * 9 and (np.diff(d.indptr) == 9).all()
    gi, gd = d.indices.reshape(700, 9), d.data.reshape(700, 9)
    bad, _ = knn_sets_equal_mod_ties(gi, gd, oi[:, 1:], od[:, 1:])
    assert bad == 0
    np.testing.assert_allclose(gd, od[:, 1:], rtol=2e-6, atol=1e-5)
    oc_ref, _, _ = oc.fuzzy_simplicial_set(oi, od, 700, 10)
    c = adata.obsp["connectivities"]
    assert abs(c - oc_ref).max() < 5e-6 and c.nnz == oc_ref.nnz


def test_neighbors_key_added_use_rep_n_pcs(sc, pbmc68k):
    """tests/test_neighbors_key_added.py:35-50 + params recording."""
    adata = sc.AnnData(pbmc68k["X"], obsm={"X_pca": pbmc68k["X_pca"]})
    sc.pp.neighbors(adata, n_neighbors=5, n_pcs=20, key_added="nb", random_state=3)
    assert "nb_distances" in adata.obsp and "nb_connectivities" in adata.obsp
    assert adata.uns["nb"]["params"] == {"n_neighbors": 5, "method": "umap", "random_state": 3, "metric": "euclidean", "n_pcs": 20}
    oi, od, _ = oknn.knn_sklearn(pbmc68k["X_pca"][:, :20], 5)
    gi = adata.obsp["nb_distances"].indices.reshape(700, 4)
    bad, _ = knn_sets_equal_mod_ties(gi, adata.obsp["nb_distances"].data.reshape(700, 4), oi[:, 1:], od[:, 1:])
    assert bad == 0
    with pytest.raises(ValueError, match="Did not find"):
        sc.pp.neighbors(adata, use_rep="nope")


def test_neighbors_transformer_plugin_route(sc, pbmc68k):
    """The estimator route of neighbors/_types.py:53-64 gives the same slots as the built-in search."""
    x = pbmc68k["X_pca"]
    a = sc.AnnData(pbmc68k["X"], obsm={"X_pca": x})
    b = sc.AnnData(pbmc68k["X"], obsm={"X_pca": x})
    sc.pp.neighbors(a, n_neighbors=12)
    sc.pp.neighbors(b, transformer=sc.MI355XKNNTransformer(n_neighbors=12))
    assert abs(a.obsp["distances"] - b.obsp["distances"]).max() == 0
    assert abs(a.obsp["connectivities"] - b.obsp["connectivities"]).max() == 0
    t = sc.MI355XKNNTransformer(n_neighbors=12, include_self=True).fit_transform(x)
    assert t.nnz == 700 * 13 and (t.indices.reshape(700, 13)[:, 0] == np.arange(700)).all()


def test_neighbors_precomputed_distances(sc, pbmc68k):
    """tests/test_neighbors.py:275-296: recomputing connectivities from stored distances."""
    adata = sc.AnnData(pbmc68k["X"], obsm={"X_pca": pbmc68k["X_pca"]})
    sc.pp.neighbors(adata, n_neighbors=10)
    ad2 = sc.AnnData(pbmc68k["X"])
    sc.pp.neighbors(ad2, n_neighbors=10, distances=adata.obsp["distances"])
    assert abs(ad2.obsp["connectivities"] - adata.obsp["connectivities"]).max() < 1e-6
    with pytest.warns(UserWarning, match="ignored if `distances` is given"):
        sc.pp.neighbors(ad2, n_neighbors=10, distances=adata.obsp["distances"], n_pcs=5)


def test_neighbors_auto_pca_fallback(sc, pbmc68k):
    """tests/test_neighbors_key_added.py:53-61: missing X_pca -> warning + pca."""
    adata = sc.AnnData(pbmc68k["X"])
    with pytest.warns(UserWarning, match="Falling back to preprocessing with `sc.pp.pca`"):
        sc.pp.neighbors(adata, n_neighbors=5)
    assert adata.obsm["X_pca"].shape == (700, 50)


# --------------------------------------------------------------------------------------- leiden ----
def _graph_adata(sc, pbmc68k):
    adata = sc.AnnData(pbmc68k["X"], obsm={"X_pca": pbmc68k["X_pca"]})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sc.pp.neighbors(adata, n_neighbors=10)
    return adata


def test_leiden_basic_and_params(sc, pbmc68k):
    """tests/test_clustering.py:36-102."""
    adata = _graph_adata(sc, pbmc68k)
    with pytest.warns(FutureWarning, match="default backend for leiden will be igraph"):
        sc.tl.leiden(adata)
    lab = adata.obs["leiden"]
    assert isinstance(lab.dtype, pd.CategoricalDtype) and list(lab.cat.categories) == [str(i) for i in range(len(lab.cat.categories))]
    assert adata.uns["leiden"]["params"] == {"resolution": 1, "n_iterations": -1, "random_state": 0}
    codes = lab.cat.codes.to_numpy()
    # `part.modularity` (src/scanpy/tools/_leiden.py:219) of the leidenalg flavor is igraph's VertexClustering.modularity with
    # NO modularity parameters (leidenalg's MutableVertexPartition passes none): the UNWEIGHTED resolution-1 modularity
    conn = adata.obsp["connectivities"]
    pattern = conn.copy()
    pattern.data[:] = 1.0
    assert abs(adata.uns["leiden"]["modularity"] - ol.modularity(pattern, codes)) < 1e-6
    q = ol.modularity(conn, codes)
    ad_ig = _graph_adata(sc, pbmc68k)
    sc.tl.leiden(ad_ig, flavor="igraph", resolution=0.6)  # igraph flavor: weighted, at resolution 1 whatever was optimised
    assert abs(ad_ig.uns["leiden"]["modularity"] - ol.modularity(conn, ad_ig.obs["leiden"].cat.codes.to_numpy())) < 1e-6
    mo, qo = ol.leiden(adata.obsp["connectivities"])
    print("Q gpu/oracle", q, qo, "ARI", adjusted_rand_score(mo, codes))
    assert q > qo - 0.01
    # same seed => identical, flavor igraph accepted, rng style drops random_state from params
    ad2 = _graph_adata(sc, pbmc68k)
    sc.tl.leiden(ad2, flavor="igraph", n_iterations=2, random_state=0, key_added="l2")
    sc.tl.leiden(ad2, flavor="igraph", n_iterations=2, random_state=0, key_added="l3")
    assert (ad2.obs["l2"] == ad2.obs["l3"]).all() and ad2.uns["l2"]["modularity"] == ad2.uns["l3"]["modularity"]
    sc.tl.leiden(ad2, flavor="igraph", rng=5, key_added="l4")
    assert "random_state" not in ad2.uns["l4"]["params"]


def test_leiden_errors(sc, pbmc68k):
    """tests/test_clustering.py:105-127."""
    adata = _graph_adata(sc, pbmc68k)
    with pytest.raises(ValueError, match=r"flavor must be either"):
        sc.tl.leiden(adata, flavor="foo")
    with pytest.raises(ValueError, match=r"Cannot use igraph"):
        sc.tl.leiden(adata, flavor="igraph", directed=True)
    with pytest.raises(ValueError, match=r"Do not pass in partition_type"):
        sc.tl.leiden(adata, flavor="igraph", partition_type=object)
    with pytest.raises(ValueError, match="run `pp.neighbors` first"):
        sc.tl.leiden(sc.AnnData(pbmc68k["X"]), flavor="igraph")


def test_leiden_initial_membership(sc, pbmc68k):
    """`initial_membership` through `**clustering_args` (src/scanpy/tools/_leiden.py:66, 174-196 hand it to leidenalg /
    igraph): zero iterations return the given partition (renumbered by size) with its modularity; a stable partition stays
    what it is; a coarse start is refined to a partition of the quality of a run from singletons; bad input raises"""
    from sklearn.metrics import adjusted_rand_score

    adata = _graph_adata(sc, pbmc68k)
    sc.tl.leiden(adata, flavor="igraph", n_iterations=-1)
    base, q_base = adata.obs["leiden"].cat.codes.to_numpy(), adata.uns["leiden"]["modularity"]
    given = pbmc68k["bulk_labels_codes"].astype(np.int64) * 7 + 3  # (any non-negative integer labels)
    sc.tl.leiden(adata, flavor="igraph", n_iterations=0, key_added="given", initial_membership=given)
    assert adjusted_rand_score(adata.obs["given"].cat.codes, given) == 1.0
    assert abs(adata.uns["given"]["modularity"] - sc.metrics.modularity(adata.obsp["connectivities"], given, is_directed=False)) < 1e-9
    sizes = adata.obs["given"].cat.codes.value_counts().sort_index().to_numpy()
    assert (np.diff(sizes) <= 0).all()  # ids by decreasing community size, as every result
    sc.tl.leiden(adata, flavor="igraph", n_iterations=-1, key_added="again", initial_membership=base)
    assert adjusted_rand_score(adata.obs["again"].cat.codes, base) == 1.0 and adata.uns["again"]["modularity"] == q_base
    sc.tl.leiden(adata, flavor="leidenalg", n_iterations=-1, key_added="coarse", initial_membership=np.arange(700) % 2)
    assert sc.metrics.modularity(adata.obsp["connectivities"], adata.obs["coarse"].cat.codes.to_numpy(), is_directed=False) > q_base - 5e-3
    with pytest.raises(ValueError, match="one non-negative integer per vertex"):
        sc.tl.leiden(adata, flavor="igraph", initial_membership=np.arange(10))
    with pytest.raises(ValueError, match="one non-negative integer per vertex"):
        sc.tl.leiden(adata, flavor="igraph", initial_membership=-np.ones(700, dtype=np.int64))


def test_leiden_cpm_objective(sc, pbmc68k):
    """`sc.tl.leiden(flavor='igraph', objective_function='CPM')` (src/scanpy/tools/_leiden.py:188-196 hands it to igraph's
    community_leiden): node optimal and separated under the CPM objective, quality above the trivial partitions', the
    stored `modularity` is the partition's modularity; the leidenalg flavor has no such argument.  Under the default
    (V1) preset the igraph flavor's graph holds every symmetric pair TWICE (src/scanpy/_utils/__init__.py:292-298), so the
    objective on the matrix itself is CPM at resolution / 2 -- the guarantees are checked THERE, and on 2 x the matrix at the
    resolution as given (the graph igraph sees)"""
    from oracle import leiden_guarantees as lg

    adata = _graph_adata(sc, pbmc68k)
    conn = adata.obsp["connectivities"]
    for gamma in (0.01, 0.1):
        sc.tl.leiden(adata, flavor="igraph", objective_function="CPM", resolution=gamma, key_added=f"cpm_{gamma}")
        lab = adata.obs[f"cpm_{gamma}"].cat.codes.to_numpy()
        q_cpm = lg.quality(conn, lab, resolution=gamma / 2, objective="cpm")
        assert q_cpm > max(lg.quality(conn, np.arange(700), resolution=gamma / 2, objective="cpm"),
                           lg.quality(conn, np.zeros(700, dtype=int), resolution=gamma / 2, objective="cpm"))
        assert lg.improving_moves(conn, lab, resolution=gamma / 2, objective="cpm")["count"] == 0
        assert lg.mergeable_pairs(conn, lab, resolution=gamma / 2, objective="cpm")["count"] == 0
        assert lg.improving_moves(2 * conn, lab, resolution=gamma, objective="cpm")["count"] == 0  # igraph's V1 graph
        assert abs(adata.uns[f"cpm_{gamma}"]["modularity"] - sc.metrics.modularity(conn, lab, is_directed=False)) < 1e-9
        assert adata.uns[f"cpm_{gamma}"]["params"]["resolution"] == gamma
    assert adata.obs["cpm_0.1"].nunique() > adata.obs["cpm_0.01"].nunique()  # a higher resolution: smaller communities
    with pytest.raises(ValueError, match='must be "CPM" or "modularity"'):
        sc.tl.leiden(adata, flavor="igraph", objective_function="surprise")
    with pytest.raises(TypeError, match="objective_function is igraph's argument"):
        sc.tl.leiden(adata, flavor="leidenalg", objective_function="CPM")


def test_leiden_cpm_node_weights(sc, pbmc68k):
    """`node_weights` (igraph) / `node_sizes` (leidenalg's CPMVertexPartition) through `**clustering_args`
    (src/scanpy/tools/_leiden.py:66, 174-196): vertex weights of the CPM quality.  Ones = the unweighted call; weighted: node
    optimal and separated under the WEIGHTED objective; each flavor refuses the other's name; modularity refuses both"""
    from oracle import leiden_guarantees as lg

    adata = _graph_adata(sc, pbmc68k)
    conn = adata.obsp["connectivities"]
    gamma = 0.02
    sc.tl.leiden(adata, flavor="igraph", objective_function="CPM", resolution=gamma, key_added="plain")
    sc.tl.leiden(adata, flavor="igraph", objective_function="CPM", resolution=gamma, node_weights=np.ones(700), key_added="ones")
    assert (adata.obs["plain"] == adata.obs["ones"]).all()
    nw = np.random.default_rng(2).integers(4, 49, 700) / 16.0
    sc.tl.leiden(adata, flavor="igraph", objective_function="CPM", resolution=gamma, node_weights=list(nw), key_added="w")
    lab = adata.obs["w"].cat.codes.to_numpy()
    # (igraph flavor, V1 preset: every pair twice = resolution / 2 on the matrix; leidenalg's directed graph: as given)
    assert lg.improving_moves(conn, lab, resolution=gamma / 2, objective="cpm", node_weights=nw)["count"] == 0
    assert lg.mergeable_pairs(conn, lab, resolution=gamma / 2, objective="cpm", node_weights=nw)["count"] == 0
    sc.tl.leiden(adata, flavor="leidenalg", partition_type=type("CPMVertexPartition", (), {}), resolution=gamma / 2, node_sizes=nw,
                 key_added="w_leidenalg")
    assert (adata.obs["w"] == adata.obs["w_leidenalg"]).all()
    with pytest.raises(TypeError, match="node_sizes is not an argument of the igraph flavor"):
        sc.tl.leiden(adata, flavor="igraph", objective_function="CPM", node_sizes=nw)
    with pytest.raises(NotImplementedError, match="CPM objective only"):
        sc.tl.leiden(adata, flavor="igraph", node_weights=nw)
    with pytest.raises(ValueError, match=r"one number in \[0, 1e6\] per vertex"):
        sc.tl.leiden(adata, flavor="igraph", objective_function="CPM", node_weights=nw[:10])


def test_leiden_partition_type_of_the_leidenalg_flavor(sc, pbmc68k):
    """`partition_type=` (src/scanpy/tools/_leiden.py:107-110, 174-186: the class `leidenalg.find_partition` optimises, with
    `resolution_parameter=resolution` unless `resolution=None`): the classes are matched by name -- RBConfiguration is the
    default, Modularity is RBConfiguration at 1 and takes no resolution, CPM is the igraph flavor's objective_function='CPM'"""
    adata = _graph_adata(sc, pbmc68k)
    rb = type("RBConfigurationVertexPartition", (), {})
    mod = type("ModularityVertexPartition", (), {})
    cpm = type("CPMVertexPartition", (), {})
    sc.tl.leiden(adata, flavor="leidenalg", resolution=0.7, key_added="default")
    sc.tl.leiden(adata, flavor="leidenalg", resolution=0.7, partition_type=rb, key_added="rb")
    assert (adata.obs["default"] == adata.obs["rb"]).all()
    sc.tl.leiden(adata, flavor="leidenalg", resolution=None, partition_type=mod, key_added="mod")
    sc.tl.leiden(adata, flavor="leidenalg", resolution=1.0, key_added="one")
    assert (adata.obs["mod"] == adata.obs["one"]).all() and adata.uns["mod"]["params"]["resolution"] is None
    # CPM is not scale invariant and the two flavors see different graphs (src/scanpy/_utils/__init__.py:292-298: the igraph
    # flavor's V1 graph holds every symmetric pair twice, leidenalg's directed graph every direction once): the igraph flavor
    # at resolution g optimises what the leidenalg flavor optimises at g / 2 -- and NOT what it optimises at g
    from scanpy_amd import settings

    sc.tl.leiden(adata, flavor="leidenalg", resolution=0.05, partition_type=cpm, key_added="cpm")
    sc.tl.leiden(adata, flavor="igraph", resolution=0.1, objective_function="CPM", key_added="cpm_igraph")
    assert (adata.obs["cpm"] == adata.obs["cpm_igraph"]).all()
    sc.tl.leiden(adata, flavor="igraph", resolution=0.05, objective_function="CPM", key_added="cpm_igraph_same_resolution")
    assert adata.obs["cpm_igraph_same_resolution"].nunique() < adata.obs["cpm"].nunique()
    settings.preset = "ScanpyV2Preview"  # Weighted_Adjacency(mode=undirected): every pair once (:285-290)
    try:
        sc.tl.leiden(adata, flavor="igraph", resolution=0.05, objective_function="CPM", key_added="cpm_igraph_v2")
    finally:
        settings.preset = "ScanpyV1"
    assert (adata.obs["cpm"] == adata.obs["cpm_igraph_v2"]).all()
    # RBERVertexPartition: sum_ij (A_ij - resolution p) delta with p = the density of the (directed) graph = CPM at resolution x p
    conn = adata.obsp["connectivities"]
    dens = conn.sum() / (700 * 699)
    sc.tl.leiden(adata, flavor="leidenalg", resolution=1.5, partition_type=type("RBERVertexPartition", (), {}), key_added="rber")
    sc.tl.leiden(adata, flavor="leidenalg", resolution=1.5 * dens, partition_type=cpm, key_added="cpm_at_density")
    assert (adata.obs["rber"] == adata.obs["cpm_at_density"]).all() and 1 < adata.obs["rber"].nunique() < 700
    with pytest.raises(TypeError, match="unexpected keyword argument 'resolution_parameter'"):
        sc.tl.leiden(adata, flavor="leidenalg", resolution=1.0, partition_type=mod)
    with pytest.raises(NotImplementedError, match="SurpriseVertexPartition"):
        sc.tl.leiden(adata, flavor="leidenalg", partition_type=type("SurpriseVertexPartition", (), {}))


def test_leiden_restrict_to_and_keys(sc, pbmc68k):
    """tests/test_clustering.py:177-242."""
    adata = _graph_adata(sc, pbmc68k)
    adata.obs["bulk"] = pd.Categorical(pbmc68k["bulk_labels_codes"].astype(str))
    cats = list(adata.obs["bulk"].cat.categories[:2])
    sc.tl.leiden(adata, flavor="igraph", restrict_to=("bulk", cats), resolution=0.5)
    assert "leiden_R" in adata.obs
    sub = adata.obs["bulk"].isin(cats).to_numpy()
    assert (adata.obs["leiden_R"][~sub].astype(str) == adata.obs["bulk"][~sub].astype(str)).all()
    assert all(v.startswith("-".join(cats) + ",") for v in adata.obs["leiden_R"][sub].astype(str))
    with pytest.raises(ValueError, match="not a valid category"):
        sc.tl.leiden(adata, flavor="igraph", restrict_to=("bulk", ["nope"]))
    sc.tl.leiden(adata, flavor="igraph", obsp="connectivities", key_added="viaobsp")
    sc.tl.leiden(adata, flavor="igraph", neighbors_key="neighbors", key_added="viakey")
    assert (adata.obs["viaobsp"] == adata.obs["viakey"]).all()
    with pytest.raises(ValueError, match="can't specify both"):
        sc.tl.leiden(adata, flavor="igraph", obsp="connectivities", neighbors_key="neighbors")


# ------------------------------------------------------------------------------------- pipeline ----
def test_full_pipeline_planted_ari(sc):
    """north_star gates on one AnnData: PCA loadings 1e-4, kNN sets equal, Leiden ARI >= 0.99 vs the CPU chain."""
    from scanpy_amd.datasets import synthetic_planted

    x, truth = synthetic_planted(30000, 2000, seed=1)
    adata = sc.AnnData(x)
    sc.pp.pca(adata)
    sc.pp.neighbors(adata)
    sc.tl.leiden(adata, flavor="igraph")
    gpu = adata.obs["leiden"].cat.codes.to_numpy()
    # CPU reference chain on the same input
    ref = opca.pca_reference(x, 50)
    assert np.abs(np.abs(adata.varm["PCs"].T) - np.abs(ref["components"])).max() < 1e-4
    # kNN parity is defined on the SAME embedding: feed ours to the oracle
    oi, od, _ = oknn.knn_sklearn(adata.obsm["X_pca"], 15, n_jobs=-1)
    d = adata.obsp["distances"]
    bad, differ = knn_sets_equal_mod_ties(d.indices.reshape(-1, 14), d.data.reshape(-1, 14), oi[:, 1:], od[:, 1:])
    assert bad == 0, (bad, differ)
    # CPU chain end to end (its own PCA -> kNN -> graph -> Leiden)
    ci, cd, _ = oknn.knn_sklearn(ref["X_pca"].astype(np.float32), 15, n_jobs=-1)
    cg, _, _ = oc.fuzzy_simplicial_set(ci, cd, x.shape[0], 15)
    cpu, _ = ol.leiden(cg)
    ari = adjusted_rand_score(cpu, gpu)
    print("ARI gpu-vs-cpu-chain", ari, "ARI vs truth", adjusted_rand_score(truth, gpu), "n clusters", gpu.max() + 1)
    assert ari >= 0.99


def test_pca_sparse_equals_dense_and_integer_input(sc, pbmc68k):
    """tests/test_pca.py:306-330 (implicit centring of sparse input == explicit centring of dense input, atol 1e-6)
    and integer counts as input (promoted like the reference: float64 components)."""
    counts = pbmc68k["counts"][:300]
    xs = counts.astype(np.float32)
    xs.data = np.log1p(xs.data)
    a_sparse = sc.AnnData(xs.copy())
    a_dense = sc.AnnData(xs.toarray())
    sc.pp.pca(a_sparse, n_comps=20)
    sc.pp.pca(a_dense, n_comps=20)
    np.testing.assert_allclose(a_sparse.obsm["X_pca"], a_dense.obsm["X_pca"], atol=1e-6)
    np.testing.assert_allclose(a_sparse.varm["PCs"], a_dense.varm["PCs"], atol=1e-6)
    np.testing.assert_allclose(a_sparse.uns["pca"]["variance"], a_dense.uns["pca"]["variance"], rtol=1e-6)
    a_int = sc.AnnData(counts.astype(np.int32))
    sc.pp.pca(a_int, n_comps=10)
    assert a_int.obsm["X_pca"].dtype == np.float32 and a_int.varm["PCs"].dtype == np.float64
    ref = opca.pca_reference(counts.astype(np.float32), 10)
    assert np.abs(np.abs(a_int.varm["PCs"].T) - np.abs(ref["components"])).max() < 1e-4


def test_counts_to_clusters_to_umap_with_the_dropin_calls():
    """The whole widened chain through the drop-in functions, as a scanpy script would call them:
    normalize_total -> log1p -> highly_variable_genes -> scale -> pca -> neighbors -> leiden -> umap."""
    from sklearn.metrics import adjusted_rand_score
    from sklearn.neighbors import NearestNeighbors

    import scanpy_amd as sc
    from scanpy_amd.datasets import synthetic_planted

    x, truth = synthetic_planted(5000, 800, n_types=12, seed=7)
    x.data = np.expm1(x.data).astype(np.float32)  # back to a counts-like scale
    adata = sc.AnnData(x)
    sc.pp.normalize_total(adata, target_sum=1e4)
    sc.pp.log1p(adata)
    sc.pp.highly_variable_genes(adata, n_top_genes=400)
    assert int(adata.var["highly_variable"].sum()) == 400
    with pytest.warns(UserWarning, match="densifies"):
        sc.pp.scale(adata, max_value=10)
    assert adata.X.shape == (5000, 800) and abs(float(adata.X[:, adata.var["highly_variable"].to_numpy()].mean())) < 1e-2  # clipped tails shift it slightly
    sc.pp.pca(adata, n_comps=30)  # uses var['highly_variable'] as the gene mask, like the reference
    assert (np.abs(adata.varm["PCs"][~adata.var["highly_variable"].to_numpy()]).sum() == 0)
    sc.pp.neighbors(adata)
    sc.tl.leiden(adata)
    sc.tl.umap(adata)
    assert adjusted_rand_score(truth, adata.obs["leiden"].cat.codes.to_numpy()) > 0.95
    y = adata.obsm["X_umap"]
    nb = NearestNeighbors(n_neighbors=11).fit(y).kneighbors(y, return_distance=False)[:, 1:]
    assert (truth[nb] == truth[:, None]).mean() > 0.95


@pytest.mark.parametrize("metric", ["cosine", "correlation", "sqeuclidean", "l2"])
def test_neighbors_metrics_served_by_the_euclidean_kernel(sc, pbmc68k, metric):
    """metric= 'cosine' / 'correlation' / 'sqeuclidean' / 'l2' (the reference hands the name to sklearn,
    neighbors/__init__.py:761-768; the accepted names: neighbors/_types.py:23-50): same neighbour sets and distances as
    sklearn's brute-force search with that metric -- cosine / correlation through the Euclidean kernel on unit-length
    (row-centred) rows, sqeuclidean = squared distances of the same lists"""
    adata = sc.AnnData(pbmc68k["X"].copy())
    adata.obsm["X_pca"] = pbmc68k["X_pca"]
    sc.pp.neighbors(adata, n_neighbors=12, metric=metric)
    assert adata.uns["neighbors"]["params"]["metric"] == metric
    oi, od, _ = oknn.knn_sklearn(pbmc68k["X_pca"], 12, metric=metric)
    d = adata.obsp["distances"]
    assert (np.diff(d.indptr) == 11).all()
    got_i, got_d = d.indices.reshape(-1, 11), d.data.reshape(-1, 11)
    order = np.argsort(got_d, axis=1, kind="stable")
    got_i, got_d = np.take_along_axis(got_i, order, 1), np.take_along_axis(got_d, order, 1)
    bad, _ = knn_sets_equal_mod_ties(got_i, got_d, oi[:, 1:], od[:, 1:], rtol=1e-5, atol=1e-7)
    assert bad == 0
    np.testing.assert_allclose(got_d, od[:, 1:], rtol=2e-5, atol=2e-7)
    tr = sc.MI355XKNNTransformer(n_neighbors=12, metric=metric)
    g = tr.fit_transform(pbmc68k["X_pca"])
    assert abs(g - d).max() < 1e-12
    if metric == "cosine":
        with pytest.raises(ValueError, match="all-zero"):
            sc.MI355XKNNTransformer(metric="cosine").fit_transform(np.zeros((50, 4), dtype=np.float32))
    if metric == "correlation":
        with pytest.raises(ValueError, match="constant rows"):
            sc.MI355XKNNTransformer(metric="correlation").fit_transform(np.ones((50, 4), dtype=np.float32))
    with pytest.raises(NotImplementedError, match="manhattan"):
        sc.pp.neighbors(adata, n_neighbors=12, metric="manhattan")


@pytest.mark.parametrize("resident", ["1", "0"], ids=["resident", "streamed"])
@pytest.mark.parametrize("chunk_size", [333, 2000])
def test_pca_chunked_equals_one_shot(sc, pbmc68k, chunk_size, resident, monkeypatch):
    """`chunked=True` streams row chunks through the device; the fixed-point Gram matrix is additive over chunks, so the
    result is the one-shot result bit for bit (the reference's chunked path, IncrementalPCA, is an approximation)"""
    monkeypatch.setenv("SCAMD_PCA_CHUNK_RESIDENT", resident)  # "0": every pass re-uploads, copies overlap the kernels
    x = pbmc68k["counts"].astype(np.float32)
    a1, a2 = sc.AnnData(x.copy()), sc.AnnData(x.copy())
    sc.pp.pca(a1, n_comps=20)
    sc.pp.pca(a2, n_comps=20, chunked=True, chunk_size=chunk_size)
    np.testing.assert_array_equal(a1.varm["PCs"], a2.varm["PCs"])
    np.testing.assert_array_equal(a1.obsm["X_pca"], a2.obsm["X_pca"])
    np.testing.assert_array_equal(a1.uns["pca"]["variance"], a2.uns["pca"]["variance"])
    with pytest.raises(ValueError, match="zero_center"):
        sc.pp.pca(a2, n_comps=20, chunked=True, zero_center=False)


@pytest.mark.parametrize("method", ["gauss", "jaccard"])
def test_neighbors_gauss_and_jaccard(sc, neighbors_toy, pbmc68k, method):
    """tests/test_neighbors.py:196-227: the reference's golden connectivities of the 4-point toy, then the bundled fixture
    against the oracle restatement (sparsity pattern identical, values to float32 accuracy)"""
    x = neighbors_toy["X"]
    k = int(neighbors_toy["n_neighbors"])
    adata = sc.AnnData(x)
    sc.pp.neighbors(adata, n_neighbors=k, method=method)
    golden = neighbors_toy["connectivities_gauss_knn" if method == "gauss" else "connectivities_jaccard"]
    np.testing.assert_allclose(adata.obsp["connectivities"].toarray(), golden, rtol=1e-6)
    assert adata.uns["neighbors"]["params"]["method"] == method
    big = sc.AnnData(pbmc68k["X"].copy())
    big.obsm["X_pca"] = pbmc68k["X_pca"]
    sc.pp.neighbors(big, n_neighbors=12, method=method)
    oi, od, _ = oknn.knn_sklearn(pbmc68k["X_pca"], 12)
    ref = oc.gauss_knn(oi, od, 700) if method == "gauss" else oc.jaccard_knn(oi, 700, 12)
    got = big.obsp["connectivities"]
    assert (abs(got - got.T) > 1e-7).nnz == 0 and got.has_sorted_indices
    ref.sort_indices()
    assert np.array_equal(got.indptr, ref.indptr) and np.array_equal(got.indices, ref.indices)
    np.testing.assert_allclose(got.data, ref.data, rtol=2e-6, atol=1e-9)
    with pytest.raises(NotImplementedError, match="knn=False"):
        sc.pp.neighbors(big, method="gauss", knn=False)
