"""Host logic of scanpy_amd.pp.normalize_total / log1p / highly_variable_genes / scale on a CPU stand-in for the
device passes (tests/stub_backend.CpuStubPPBackend), checked against the oracle and the reference's goldens.
The same assertions run against the HIP kernels in tests/test_gpu_preprocess.py."""
from __future__ import annotations

import logging

import numpy as np
import pandas as pd
import pytest
from scipy import sparse

import scanpy_amd as sc
from oracle import preprocess as op
from scanpy_amd.preprocessing import _csr_device
from tests.stub_backend import CpuStubPPBackend


@pytest.fixture(autouse=True)
def _stub(monkeypatch):
    monkeypatch.setattr(_csr_device, "default_backend", lambda: CpuStubPPBackend())


def _dense(x):
    return x.toarray() if sparse.issparse(x) else np.asarray(x)


def check_chain_against_goldens(pbmc68k, hvg_golden, typ):
    """tests/test_highly_variable_genes.py:367-422 through the drop-in functions"""
    for flavor, params in (("seurat", dict(min_mean=0.0125, max_mean=3, min_disp=0.5)), ("cell_ranger", dict(n_top_genes=100))):
        adata = sc.AnnData(typ(pbmc68k["raw_X"].copy()))
        sc.pp.normalize_total(adata, target_sum=1e4)
        sc.pp.log1p(adata)
        assert adata.uns["log1p"] == {"base": None}
        sc.pp.highly_variable_genes(adata, flavor=flavor, **params)
        assert np.array_equal(adata.var["highly_variable"].to_numpy(), hvg_golden[f"{flavor}_highly_variable"])
        for col in ("means", "dispersions", "dispersions_norm"):
            np.testing.assert_allclose(adata.var[col].to_numpy(), hvg_golden[f"{flavor}_{col}"], rtol=2e-5, atol=2e-5)
        assert adata.uns["hvg"] == {"flavor": flavor} and adata.var["dispersions_norm"].dtype == np.float32


def check_normalize_total(typ):
    for dtype in ("float32", "int64"):
        x_total = np.array([[1, 0], [3, 0], [5, 6]]).astype(dtype)
        adata = sc.AnnData(typ(x_total))
        sc.pp.normalize_total(adata, key_added="n_counts")
        assert np.allclose(_dense(adata.X).sum(axis=1), 3.0)
        assert np.allclose(adata.obs["n_counts"], [1 / 3, 1.0, 11 / 3])
        sc.pp.normalize_total(adata, target_sum=1, key_added="n_counts2")
        assert np.allclose(_dense(adata.X).sum(axis=1), 1.0)
        adata = sc.AnnData(typ(np.array([[1, 0, 1], [3, 0, 1], [5, 6, 1]]).astype(dtype)))
        sc.pp.normalize_total(adata, exclude_highly_expressed=True, max_fraction=0.7)
        assert np.allclose(_dense(adata.X)[:, 1:3].sum(axis=1), 1.0)
    a = np.array([[3, 3, 3, 6, 6], [1, 1, 1, 2, 2], [1, 22, 1, 2, 2]], dtype="float32")
    out = sc.pp.normalize_total(sc.AnnData(typ(a)), target_sum=1, exclude_highly_expressed=True, max_fraction=0.2, inplace=False)
    assert np.allclose(_dense(out["X"]), [[0.5, 0.5, 0.5, 1, 1], [0.5, 0.5, 0.5, 1, 1], [0.5, 11, 0.5, 1, 1]])
    # zero-count cells: warning, median over the non-zero sums (tests/test_normalization.py:336-353)
    adata = sc.AnnData(typ(np.array([[0.0, 0.0], [4.0, 6.0], [8.0, 12.0], [12.0, 18.0]], dtype=np.float32)))
    with pytest.warns(UserWarning, match="Some cells have zero counts"):
        sc.pp.normalize_total(adata)
    assert np.allclose(_dense(adata.X).sum(axis=1)[1:], 20.0)
    with pytest.raises(ValueError, match="max_fraction"):
        sc.pp.normalize_total(adata, max_fraction=1.5)
    with pytest.raises(ValueError, match="copy=True"):
        sc.pp.normalize_total(adata, copy=True, inplace=False)


def check_scale(scale_toy, typ):
    """tests/test_scaling.py:75-130"""
    t = scale_toy
    mask = np.array((0, 0, 1, 1, 1, 0, 0), dtype=bool)
    for dtype in (np.float32, np.int64):
        for zero_center in (True, False):
            for mk, xk, ck, sk in ((None, "X_original", "X_centered_original", "X_scaled_original"),
                                   (mask, "X_for_mask", "X_centered_for_mask", "X_scaled_for_mask")):
                adata = sc.AnnData(typ(t[xk].astype(dtype)))
                if zero_center and typ.__name__ != "array":
                    with pytest.warns(UserWarning, match=r"zero-center.*densifies"):
                        sc.pp.scale(adata, zero_center=zero_center, mask_obs=mk)
                else:
                    sc.pp.scale(adata, zero_center=zero_center, mask_obs=mk)
                assert np.allclose(_dense(adata.X), t[ck] if zero_center else t[sk])
                arr = sc.pp.scale(typ(t[xk].astype(dtype)), zero_center=False, mask_obs=mk, max_value=1)
                assert np.allclose(_dense(arr), t["X_scaled_original_clipped" if mk is None else "X_scaled_for_mask_clipped"])
    adata = sc.AnnData(np.array(t["X_for_mask"], dtype="float32"))
    adata.obs["some cells"] = mask
    sc.pp.scale(adata, mask_obs="some cells")
    assert np.array_equal(adata.X, t["X_centered_for_mask"]) and "mean of some cells" in adata.var.columns
    assert adata.X.dtype == np.float32
    with pytest.raises(ValueError, match=r"Cannot.*refer.*mask.*without.*anndata"):
        sc.pp.scale(np.array(t["X_original"], dtype=np.float32), mask_obs="mask")


def check_random_against_oracle(typ, seed=0):
    """normalize -> log1p(base 2) -> HVG (both flavors, top-n and cut-offs, batches) -> scale(max_value) vs the oracle"""
    rng = np.random.default_rng(seed)
    n, g = 400, 120
    lam = rng.gamma(0.3, 4.0, size=g)
    counts = rng.poisson(lam[None, :] * rng.uniform(0.3, 2.0, size=(n, 1))).astype(np.float32)
    counts[5] = 0  # an empty cell
    counts[:, 7] = 0  # an unexpressed gene
    adata = sc.AnnData(typ(counts))
    with pytest.warns(UserWarning, match="zero counts"):
        sc.pp.normalize_total(adata, exclude_highly_expressed=True, max_fraction=0.2)
    xo, fo, _ = op.normalize_total(sparse.csr_matrix(counts), exclude_highly_expressed=True, max_fraction=0.2)
    np.testing.assert_allclose(_dense(adata.X), xo.toarray(), rtol=1e-6, atol=1e-7)
    sc.pp.log1p(adata, base=2)
    xo = op.log1p(xo, base=2)
    np.testing.assert_allclose(_dense(adata.X), xo.toarray(), rtol=2e-6, atol=1e-7)
    for flavor in ("seurat", "cell_ranger"):
        for kw in (dict(n_top_genes=30), dict(min_mean=0.01, max_mean=5, min_disp=0.2)):
            with pytest.warns(UserWarning) if False else _nullcontext():
                df = sc.pp.highly_variable_genes(adata, flavor=flavor, inplace=False, **kw)
            do = op.highly_variable_genes(xo, flavor=flavor, log1p_base=2, **kw)
            assert np.array_equal(df["highly_variable"].to_numpy(), do["highly_variable"].to_numpy())
            for col in ("means", "dispersions", "dispersions_norm"):
                np.testing.assert_allclose(df[col].to_numpy(), do[col].to_numpy(), rtol=2e-5, atol=2e-5, equal_nan=True)
    # batches: equals the per-batch oracle combined as `_highly_variable_genes_batched` does
    adata.obs["batch"] = pd.Categorical(np.where(np.arange(n) % 3 == 0, "a", "b"))
    df = sc.pp.highly_variable_genes(adata, batch_key="batch", n_top_genes=25, inplace=False)
    assert int(df["highly_variable"].sum()) == 25 and set(df["highly_variable_nbatches"].unique()) <= {0, 1, 2}
    per, ambiguous = [], np.zeros(g, dtype=bool)
    for b in ("a", "b"):
        rows = (adata.obs["batch"] == b).to_numpy()
        sub = xo[rows]
        expressed = np.flatnonzero(np.asarray((sub > 0).sum(axis=0)).ravel() >= 1)
        d = op.highly_variable_genes(sub[:, expressed], n_top_genes=25, log1p_base=2)
        hv, mean = np.zeros(g, dtype=bool), np.zeros(g)
        hv[expressed], mean[expressed] = d["highly_variable"].to_numpy(), d["means"].to_numpy()
        per.append((hv, mean))
        # two-gene bins give normalised dispersions of exactly +-1/sqrt(2): several genes tie at the cut-off to the last
        # ulp, and which side of `>=` they fall on depends on the summation order of the per-gene sums
        dn = d["dispersions_norm"].to_numpy()
        cut = np.sort(dn[~np.isnan(dn)])[::-1][24]
        ambiguous[expressed[np.abs(dn - cut) <= 1e-9 * max(1.0, abs(cut))]] = True
    nb = per[0][0].astype(int) + per[1][0].astype(int)
    got = df["highly_variable_nbatches"].to_numpy()
    assert np.array_equal(got[~ambiguous], nb[~ambiguous]) and np.all(np.abs(got - nb)[ambiguous] <= 1)
    np.testing.assert_allclose(df["means"].to_numpy(), (per[0][1] + per[1][1]) / 2, rtol=2e-5, atol=2e-6)
    # scale with clipping, both centring modes
    for zero_center in (True, False):
        a2 = sc.AnnData(typ(_dense(adata.X).astype(np.float32)))
        if zero_center and typ.__name__ != "array":
            with pytest.warns(UserWarning, match="densifies"):
                sc.pp.scale(a2, zero_center=zero_center, max_value=3)
        else:
            sc.pp.scale(a2, zero_center=zero_center, max_value=3)
        so, mo, sdo = op.scale(typ(_dense(adata.X).astype(np.float32)), zero_center=zero_center, max_value=3)
        np.testing.assert_allclose(_dense(a2.X), _dense(so), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(a2.var["mean"], mo, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(a2.var["std"], sdo, rtol=1e-6, atol=1e-9)
        if zero_center and typ.__name__ != "array":
            assert a2.X.dtype == np.float64


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def array(x, dtype=None):
    """dense ndarray from anything (np.array(csr) would give an object array)"""
    return np.array(x.toarray() if sparse.issparse(x) else x, dtype=dtype)


TYPES = [array, sparse.csr_matrix, sparse.csc_matrix]


@pytest.mark.parametrize("typ", TYPES, ids=lambda t: t.__name__)
def test_chain_goldens(pbmc68k, hvg_golden, typ):
    check_chain_against_goldens(pbmc68k, hvg_golden, typ)


@pytest.mark.parametrize("typ", TYPES, ids=lambda t: t.__name__)
def test_normalize_total(typ):
    check_normalize_total(typ)


@pytest.mark.parametrize("typ", TYPES, ids=lambda t: t.__name__)
def test_scale(scale_toy, typ):
    check_scale(scale_toy, typ)


@pytest.mark.parametrize("typ", TYPES, ids=lambda t: t.__name__)
def test_random_against_oracle(typ):
    check_random_against_oracle(typ)


def test_log1p_warns_when_already_logged_and_keeps_format(caplog):
    x = sparse.random(20, 9, density=0.4, format="csc", dtype=np.float32, random_state=1)
    adata = sc.AnnData(x.copy())
    sc.pp.log1p(adata)
    assert adata.X.format == "csc" and np.allclose(adata.X.toarray(), np.log1p(x.toarray()), rtol=1e-6)
    with caplog.at_level(logging.WARNING, logger="scanpy_amd"):
        sc.pp.log1p(adata)
    assert "already log-transformed" in caplog.text
    out = sc.pp.log1p(x.toarray(), base=10)
    assert np.allclose(out, np.log10(1 + x.toarray()), rtol=2e-6, atol=1e-7)
    with pytest.raises(NotImplementedError):
        sc.pp.log1p(adata, chunked=True)


def test_hvg_errors_and_subset(pbmc68k):
    adata = sc.AnnData(pbmc68k["raw_X"].copy())
    with pytest.raises(ValueError, match='`flavor` needs to be "seurat" or "cell_ranger"'):
        sc.pp.highly_variable_genes(adata, flavor="svr")
    with pytest.raises(ValueError, match="expects an `AnnData`"):
        sc.pp.highly_variable_genes(adata.X)
    with pytest.warns(UserWarning, match="all cutoffs are ignored"):
        sc.pp.highly_variable_genes(adata, n_top_genes=50, min_mean=0.1)
    sc.pp.highly_variable_genes(adata, n_top_genes=50, subset=True)
    assert adata.shape == (700, 50) and adata.var["highly_variable"].all()
    with pytest.raises(NotImplementedError, match="float32"):
        sc.pp.log1p(np.ones((3, 3), dtype=np.float64))


def check_filters(typ):
    """src/scanpy/preprocessing/_simple.py:53-307 semantics against plain numpy"""
    rng = np.random.default_rng(3)
    dense = rng.poisson(0.4, size=(300, 40)).astype(np.float32)
    dense[7] = 0
    dense[:, 5] = 0
    for kw, number, keep in (
        (dict(min_counts=10), dense.sum(axis=1), dense.sum(axis=1) >= 10),
        (dict(max_counts=20), dense.sum(axis=1), dense.sum(axis=1) <= 20),
        (dict(min_genes=8), (dense > 0).sum(axis=1), (dense > 0).sum(axis=1) >= 8),
        (dict(max_genes=15), (dense > 0).sum(axis=1), (dense > 0).sum(axis=1) <= 15),
    ):
        subset, num = sc.pp.filter_cells(typ(dense), **kw)
        assert np.array_equal(subset, keep) and np.allclose(num, number)
        adata = sc.AnnData(typ(dense))
        sc.pp.filter_cells(adata, **kw)
        key = "n_counts" if "counts" in next(iter(kw)) else "n_genes"
        assert adata.n_obs == int(keep.sum()) and np.allclose(adata.obs[key], number[keep])
        assert np.allclose(_dense(adata.X), dense[keep])
    for kw, number, keep in (
        (dict(min_counts=100), dense.sum(axis=0), dense.sum(axis=0) >= 100),
        (dict(min_cells=90), (dense > 0).sum(axis=0), (dense > 0).sum(axis=0) >= 90),
        (dict(max_cells=110), (dense > 0).sum(axis=0), (dense > 0).sum(axis=0) <= 110),
    ):
        subset, num = sc.pp.filter_genes(typ(dense), **kw)
        assert np.array_equal(subset, keep) and np.allclose(num, number)
        adata = sc.AnnData(typ(dense))
        sc.pp.filter_genes(adata, **kw)
        key = "n_counts" if "counts" in next(iter(kw)) else "n_cells"
        assert adata.n_vars == int(keep.sum()) and np.allclose(adata.var[key], number[keep])
    with pytest.raises(ValueError, match="exactly one"):
        sc.pp.filter_cells(typ(dense), min_counts=1, min_genes=1)
    with pytest.raises(ValueError, match="exactly one"):
        sc.pp.filter_genes(typ(dense))


@pytest.mark.parametrize("typ", TYPES, ids=lambda t: t.__name__)
def test_filters(typ):
    check_filters(typ)


def check_rep_mutation(typ):
    """`layer=` / `obsm=` transform only the named array and give the same numbers as transforming X
    (src/testing/scanpy/_helpers/__init__.py:35-63, used by tests/test_normalization.py:77-83 and the log1p / scale tests)"""
    x = typ(sparse.random(100, 50, format="csr", density=0.2, dtype=np.float32, random_state=0))
    funcs = (
        (sc.pp.normalize_total, {}),
        (sc.pp.log1p, {}),
        (sc.pp.scale, dict(zero_center=False)),
    )
    for func, kw in funcs:
        def make():
            a = sc.AnnData(x.copy())
            a.layers["layer"] = x.copy()
            a.obsm["obsm"] = x.copy()
            return a

        with warnings_ignored():
            out_x = make()
            func(out_x, **kw)
            per_field = {}
            for field in ("layer", "obsm"):
                a = make()
                func(a, **{field: field}, **kw)
                per_field[field] = a
        for field, a in per_field.items():
            got = a.layers["layer"] if field == "layer" else a.obsm["obsm"]
            np.testing.assert_array_equal(_dense(out_x.X), _dense(got))  # same result as on X
            np.testing.assert_array_equal(_dense(x), _dense(a.X))  # X untouched
            other = a.obsm["obsm"] if field == "layer" else a.layers["layer"]
            np.testing.assert_array_equal(_dense(x), _dense(other))  # the other representation untouched
        np.testing.assert_array_equal(_dense(x), _dense(out_x.layers["layer"]))
        np.testing.assert_array_equal(_dense(x), _dense(out_x.obsm["obsm"]))


class warnings_ignored:
    def __enter__(self):
        import warnings

        self._cm = warnings.catch_warnings()
        self._cm.__enter__()
        warnings.simplefilter("ignore")

    def __exit__(self, *a):
        return self._cm.__exit__(*a)


@pytest.mark.parametrize("typ", TYPES, ids=lambda t: t.__name__)
def test_rep_mutation(typ):
    check_rep_mutation(typ)


# ---- tests/test_highly_variable_genes.py:47-118 semantics (blobs-like data, as sc.datasets.blobs()) -----------------------
def _blobs_adata():
    from sklearn.datasets import make_blobs

    x, _ = make_blobs(n_samples=640, n_features=11, centers=5, cluster_std=1.0, random_state=0)
    return sc.AnnData(x.astype(np.float32))


def check_hvg_reference_semantics():
    with warnings_ignored():
        adata = _blobs_adata()
        sc.pp.highly_variable_genes(adata)  # test_runs
        no_batch = adata.var["highly_variable"].copy()
        gen = np.random.default_rng(0)
        adata.obs["batch"] = pd.Categorical(gen.binomial(3, 0.5, size=adata.n_obs))
        sc.pp.highly_variable_genes(adata, batch_key="batch")  # test_supports_batch
        assert {"highly_variable_nbatches", "highly_variable_intersection"} <= set(adata.var.columns)
        # test_no_batch_matches_batch: one batch == no batch
        one = _blobs_adata()
        one.obs["batch"] = pd.Categorical(["batch"] * one.n_obs)
        sc.pp.highly_variable_genes(one, batch_key="batch")
        assert np.all(no_batch.to_numpy() == one.var["highly_variable"].to_numpy())
        assert np.all(one.var["highly_variable_intersection"].to_numpy() == one.var["highly_variable"].to_numpy())

        # test_supports_layers: the layer, not X, is analysed
        def execute(layer):
            g2 = np.random.default_rng(0)
            a = _blobs_adata()
            if layer:
                shuffled = a.X.copy()
                g2.shuffle(shuffled)
                a.layers[layer] = shuffled
                a.X = np.zeros_like(a.X)
            a.obs["batch"] = pd.Categorical(g2.binomial(4, 0.5, size=a.n_obs))
            sc.pp.highly_variable_genes(a, batch_key="batch", n_top_genes=3, layer=layer)
            assert "highly_variable_nbatches" in a.var.columns and int(a.var["highly_variable"].sum()) == 3
            return a

        a1, a2 = execute(None), execute("test_layer")
        assert (a1.var["highly_variable"].to_numpy() != a2.var["highly_variable"].to_numpy()).any()
        # test_no_inplace: columns of the returned frame
        for batch_key in (None, "batch"):
            a = _blobs_adata()
            if batch_key:
                a.obs[batch_key] = np.tile(["a", "b"], a.n_obs // 2)
            df = sc.pp.highly_variable_genes(a, batch_key=batch_key, n_bins=3, inplace=False)
            cols = {"means", "dispersions", "dispersions_norm", "highly_variable"} | (
                {"mean_bin"} if batch_key is None else {"highly_variable_nbatches", "highly_variable_intersection"})
            assert isinstance(df, pd.DataFrame) and set(df.columns) == cols


def check_hvg_keeps_the_matrix(pbmc68k):
    """tests/test_highly_variable_genes.py:118-137 (`test_keep_layer`): annotating does not touch X"""
    for base in (None, 10):
        for flavor in ("seurat", "cell_ranger"):
            adata = sc.AnnData(pbmc68k["counts"].astype(np.float32))
            sc.pp.filter_genes(adata, min_counts=1)
            sc.pp.log1p(adata, base=base)
            assert sparse.issparse(adata.X)
            x_orig = adata.X.copy()
            if flavor == "seurat":
                sc.pp.highly_variable_genes(adata, n_top_genes=50, flavor=flavor)
            else:
                sc.pp.highly_variable_genes(adata, flavor=flavor)
            assert np.allclose(x_orig.toarray(), adata.X.toarray())


def test_hvg_reference_semantics():
    check_hvg_reference_semantics()


def test_hvg_keeps_the_matrix(pbmc68k):
    check_hvg_keeps_the_matrix(pbmc68k)


def check_hvg_batches(pbmc68k):
    """tests/test_highly_variable_genes.py:513-569 (`test_batches`): the batched result is the per-batch result averaged,
    genes unexpressed in a batch count as zeros there; :572-580 degenerate batches run; :598-610 n_top_genes warning"""
    x = pbmc68k["X"].copy()  # the scaled matrix of the fixture, like the reference test
    x[:100, :100] = 0.0
    adata = sc.AnnData(x)
    adata.obs["batch"] = ["0" if i < 100 else "1" for i in range(adata.n_obs)]
    a1 = sc.AnnData(x[:100].copy())
    a2 = sc.AnnData(x[100:].copy())
    with warnings_ignored():
        sc.pp.highly_variable_genes(adata, batch_key="batch", flavor="cell_ranger", n_top_genes=200)
        sc.pp.filter_genes(a1, min_cells=1)
        sc.pp.filter_genes(a2, min_cells=1)
        hvg1 = sc.pp.highly_variable_genes(a1, flavor="cell_ranger", n_top_genes=200, inplace=False)
        hvg2 = sc.pp.highly_variable_genes(a2, flavor="cell_ranger", n_top_genes=200, inplace=False)
    dn = adata.var["dispersions_norm"]
    # gene 100 is the first gene expressed in batch 0 (genes 0..99 were zeroed there and filtered out of hvg1)
    first_kept = int(np.flatnonzero((x[:100] > 0).sum(axis=0) >= 1)[0])
    assert first_kept >= 100
    pos1 = {g: i for i, g in enumerate(a1.var_names)}
    pos2 = {g: i for i, g in enumerate(a2.var_names)}
    for g in (first_kept, first_kept + 1):
        name = adata.var_names[g]
        np.testing.assert_allclose(dn.iloc[g], 0.5 * hvg1["dispersions_norm"].iloc[pos1[name]]
                                   + 0.5 * hvg2["dispersions_norm"].iloc[pos2[name]], rtol=1e-6, atol=1e-6)
    name0 = adata.var_names[0]
    if name0 in pos2:
        np.testing.assert_allclose(dn.iloc[0], 0.5 * hvg2["dispersions_norm"].iloc[pos2[name0]], rtol=1e-6, atol=1e-6)
    assert {"means", "dispersions", "dispersions_norm", "highly_variable"} <= set(hvg1.columns)
    rng = np.random.default_rng(0)
    deg = sc.AnnData(rng.standard_normal((10, 100)).astype(np.float32))
    deg.obs["batch"] = pd.Categorical([*([1] * 4), *([2] * 5), 3])
    with warnings_ignored():
        sc.pp.highly_variable_genes(deg, batch_key="batch")
    small = sc.AnnData(rng.poisson(2, (100, 30)).astype(np.float32))
    sc.pp.normalize_total(small)
    sc.pp.log1p(small)
    with pytest.warns(UserWarning, match="`n_top_genes`.*> number of normalized dispersions.*returning all genes with normalized dispersions."):
        sc.pp.highly_variable_genes(small, n_top_genes=1000, flavor="cell_ranger")


def test_hvg_batches(pbmc68k):
    check_hvg_batches(pbmc68k)


def test_views_are_actualised_with_a_warning():
    """tests/test_normalization.py:85-95 (`test_normalize_total_view`) and the same rule for log1p / scale"""
    x = np.array([[1, 0], [3, 0], [5, 6]], dtype=np.float32)
    for func, kw in ((sc.pp.normalize_total, {}), (sc.pp.log1p, {}), (sc.pp.scale, dict(zero_center=False))):
        adata = sc.AnnData(x.copy())
        v = adata[:, :]
        assert v.is_view
        with pytest.warns(UserWarning, match=r"Received a view"):
            func(v, **kw)
        func(adata, **kw)
        assert not v.is_view
        np.testing.assert_array_equal(_dense(adata.X), _dense(v.X))


def test_backed_matrices_passes_that_need_the_whole_matrix_are_refused():
    """an on-disk X (`read_h5ad(..., backed='r')`): `scale` and the filters rewrite / subset the matrix and say so
    instead of failing somewhere inside numpy (normalize_total, log1p, highly_variable_genes and pca stream)"""
    from pathlib import Path

    b = sc.read_h5ad(Path(__file__).parent / "golden" / "h5" / "adata_layout.h5ad", backed="r")
    for fn in (sc.pp.scale, lambda a: sc.pp.filter_cells(a, min_counts=1), lambda a: sc.pp.filter_genes(a, min_cells=1)):
        with pytest.raises(NotImplementedError, match="to_memory"):
            fn(b)
    with pytest.raises(NotImplementedError, match="exclude_highly_expressed"):
        sc.pp.normalize_total(b, exclude_highly_expressed=True)


@pytest.mark.parametrize("container", ["h5ad", "zarr"])
def test_counts_on_disk_to_log_normalised_statistics_without_materialising(pbmc68k, tmp_path, monkeypatch, container):
    """normalize_total / log1p on a backed count matrix become PENDING transforms (applied on the device to every row
    chunk after its upload): factors, statistics and the materialised matrix equal the in-memory chain"""
    from scipy import sparse

    from scanpy_amd._backed import BackedCsr
    from scanpy_amd.preprocessing import _highly_variable_genes as hvg

    counts = sparse.csr_matrix(pbmc68k["counts"]).astype(np.float32)
    a = sc.AnnData(counts.copy())
    sc.write(tmp_path / f"c.{container}", a, **({"compression": None} if container == "h5ad" else {}))
    b = sc.read(tmp_path / f"c.{container}", backed="r")
    orig = hvg._StreamedColStats.__init__
    monkeypatch.setattr(hvg._StreamedColStats, "__init__", lambda self, be, x, step=150: orig(self, be, x, step))
    sc.pp.normalize_total(a, target_sum=1e4, key_added="nf")
    sc.pp.normalize_total(b, target_sum=1e4, key_added="nf")
    np.testing.assert_array_equal(b.obs["nf"].to_numpy(), a.obs["nf"].to_numpy())
    sc.pp.log1p(a)
    sc.pp.log1p(b)
    assert isinstance(b.X, BackedCsr) and [k for k, _ in b.X._ops] == ["row_divide", "log1p"] and b.uns["log1p"] == {"base": None}
    assert "pending ['row_divide', 'log1p']" in repr(b.X) and b.X.absmax() is None
    sc.pp.highly_variable_genes(a, n_top_genes=150)
    sc.pp.highly_variable_genes(b, n_top_genes=150)
    np.testing.assert_allclose(b.var["means"], a.var["means"], rtol=1e-9)
    np.testing.assert_allclose(b.var["dispersions_norm"], a.var["dispersions_norm"], rtol=1e-6, atol=1e-9, equal_nan=True)
    assert (b.var["highly_variable"].to_numpy() != a.var["highly_variable"].to_numpy()).sum() <= 2
    got = b.X.to_memory()  # (through the device backend: no arithmetic on the host)
    assert (got != a.X).nnz == 0 and got.dtype == np.float32
    # median target, base-2 logarithm, a second normalisation on top of a pending one
    a2, b2 = sc.AnnData(counts.copy()), sc.read(tmp_path / f"c.{container}", backed="r")
    for ad in (a2, b2):
        sc.pp.normalize_total(ad)
        sc.pp.log1p(ad, base=2)
        sc.pp.normalize_total(ad, target_sum=50.0)
    assert (b2.X.to_memory() != a2.X).nnz == 0
    res = sc.pp.normalize_total(sc.read(tmp_path / f"c.{container}", backed="r"), inplace=False)
    assert set(res) == {"X", "norm_factor"} and res["X"].is_backed


@pytest.mark.parametrize("flavor", ["seurat", "cell_ranger"])
@pytest.mark.parametrize("batched", [False, True])
def test_hvg_streams_a_backed_matrix(pbmc68k, tmp_path, monkeypatch, flavor, batched):
    """`highly_variable_genes` needs one sweep of per-gene sums: on a backed matrix it is streamed by row chunks and
    agrees with the in-memory result (float64 partial sums in another order: 1e-12)"""
    from scipy import sparse

    from scanpy_amd.preprocessing import _highly_variable_genes as hvg

    raw = pbmc68k["raw_X"]
    x = sparse.csr_matrix(np.log1p(raw.toarray() if sparse.issparse(raw) else raw).astype(np.float32))
    a = sc.AnnData(x)
    a.obs["batch"] = pd.Categorical(np.arange(x.shape[0]) % 3)
    sc.write_zarr(tmp_path / "a.zarr", a)
    b = sc.read_zarr(tmp_path / "a.zarr", backed="r")
    orig = hvg._StreamedColStats.__init__
    monkeypatch.setattr(hvg._StreamedColStats, "__init__", lambda self, be, x, step=97: orig(self, be, x, step))
    kw = dict(flavor=flavor, n_top_genes=100, batch_key="batch" if batched else None)
    sc.pp.highly_variable_genes(a, **kw)
    sc.pp.highly_variable_genes(b, **kw)
    assert b.X.is_backed
    for col in ("means", "dispersions", "dispersions_norm"):
        np.testing.assert_allclose(b.var[col].to_numpy(), a.var[col].to_numpy(), rtol=1e-6, atol=1e-9, equal_nan=True)
    assert (b.var["highly_variable"].to_numpy() != a.var["highly_variable"].to_numpy()).sum() <= 2  # ties at the cut
    sub = sc.read_zarr(tmp_path / "a.zarr", backed="r")
    sc.pp.highly_variable_genes(sub, flavor=flavor, n_top_genes=100, subset=True)
    assert sub.shape == (x.shape[0], 100) and sub.X.shape == sub.shape and sub.X.is_backed
