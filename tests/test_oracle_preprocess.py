"""Pins oracle/preprocess.py (CPU restatement of normalize_total / log1p / highly_variable_genes / scale) to the
reference's own golden vectors and known-answer tests."""
from __future__ import annotations

import numpy as np
import pytest
from scipy import sparse

from oracle import preprocess as op

X_TOTAL = np.array([[1, 0], [3, 0], [5, 6]])  # tests/test_normalization.py:29
X_FRAC = np.array([[1, 0, 1], [3, 0, 1], [5, 6, 1]])  # :30


@pytest.mark.parametrize("typ", [np.array, sparse.csr_matrix], ids=["dense", "csr"])
@pytest.mark.parametrize("dtype", ["float32", "int64"])
def test_normalize_total_kats(typ, dtype):
    """tests/test_normalization.py:62-74"""
    x, f, _ = op.normalize_total(typ(X_TOTAL.astype(dtype)))
    assert np.allclose(np.asarray(x.sum(axis=1)).ravel(), 3.0)
    assert np.allclose(f, [1 / 3, 1.0, 11 / 3])
    x, _, _ = op.normalize_total(typ(X_TOTAL.astype(dtype)), target_sum=1)
    assert np.allclose(np.asarray(x.sum(axis=1)).ravel(), 1.0)
    x, _, _ = op.normalize_total(typ(X_FRAC.astype(dtype)), exclude_highly_expressed=True, max_fraction=0.7)
    x = x.toarray() if sparse.issparse(x) else x
    assert np.allclose(x[:, 1:3].sum(axis=1), 1.0)


@pytest.mark.parametrize("typ", [np.array, sparse.csr_matrix], ids=["dense", "csr"])
def test_normalize_total_doctest(typ):
    """src/scanpy/preprocessing/_normalization.py:218-253"""
    a = np.array([[3, 3, 3, 6, 6], [1, 1, 1, 2, 2], [1, 22, 1, 2, 2]], dtype="float32")
    x, _, _ = op.normalize_total(typ(a), target_sum=1)
    x = x.toarray() if sparse.issparse(x) else x
    assert np.allclose(x, [[0.14, 0.14, 0.14, 0.29, 0.29], [0.14, 0.14, 0.14, 0.29, 0.29], [0.04, 0.79, 0.04, 0.07, 0.07]], atol=5e-3)
    x, _, cols = op.normalize_total(typ(a), target_sum=1, exclude_highly_expressed=True, max_fraction=0.2)
    x = x.toarray() if sparse.issparse(x) else x
    assert np.allclose(x, [[0.5, 0.5, 0.5, 1, 1], [0.5, 0.5, 0.5, 1, 1], [0.5, 11, 0.5, 1, 1]])
    assert list(np.flatnonzero(cols)) == [1, 3, 4]


@pytest.mark.parametrize("typ", [np.array, sparse.csr_matrix], ids=["dense", "csr"])
def test_normalize_total_ignores_zero_count_cells(typ):
    """tests/test_normalization.py:336-353: the median is over the non-zero row sums"""
    a = np.array([[0.0, 0.0], [4.0, 6.0], [8.0, 12.0], [12.0, 18.0]])
    x, f, _ = op.normalize_total(typ(a))
    x = x.toarray() if sparse.issparse(x) else x
    assert np.allclose(x.sum(axis=1)[1:], 20.0) and np.all(x[0] == 0) and f[0] == 0


def test_nnz_median():
    """tests/test_normalization.py:328-332"""
    assert op.nnz_median(np.array([0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9])) == 5


@pytest.mark.parametrize(
    ("flavor", "params"),
    [("seurat", dict(min_mean=0.0125, max_mean=3, min_disp=0.5)), ("cell_ranger", dict(n_top_genes=100))],
)
@pytest.mark.parametrize("dense", [False, True], ids=["csr", "dense"])
def test_chain_matches_seurat_and_cell_ranger_goldens(pbmc68k, hvg_golden, flavor, params, dense):
    """tests/test_highly_variable_genes.py:367-422: raw.X -> normalize_total(1e4) -> log1p -> HVG == the R outputs"""
    x = pbmc68k["raw_X"]
    x = x.toarray() if dense else x
    x, _, _ = op.normalize_total(x, target_sum=1e4)
    x = op.log1p(x)
    df = op.highly_variable_genes(x, flavor=flavor, **params)
    assert np.array_equal(df["highly_variable"].to_numpy(), hvg_golden[f"{flavor}_highly_variable"])
    for col in ("means", "dispersions", "dispersions_norm"):
        np.testing.assert_allclose(df[col].to_numpy(), hvg_golden[f"{flavor}_{col}"], rtol=2e-5, atol=2e-5)


def test_log1p_base():
    a = np.array([[0.0, 1.0], [3.0, 7.0]])
    assert np.allclose(op.log1p(a, base=2), np.log2(1 + a))
    assert np.allclose(op.log1p(sparse.csr_matrix(a)).toarray(), np.log1p(a))


@pytest.mark.parametrize("typ", [np.array, sparse.csr_matrix], ids=["dense", "csr"])
@pytest.mark.parametrize("dtype", [np.float32, np.int64])
def test_scale_kats(scale_toy, typ, dtype):
    """tests/test_scaling.py:13-117"""
    t = scale_toy
    for zero_center, key in ((True, "X_centered_original"), (False, "X_scaled_original")):
        out, mean, std = op.scale(typ(t["X_original"].astype(dtype)), zero_center=zero_center)
        out = out.toarray() if sparse.issparse(out) else out
        assert np.allclose(out, t[key])
        assert np.allclose(mean, [0, 2, 2, 0]) and np.allclose(std, [1, 1, 2, 1])
    mask = np.array((0, 0, 1, 1, 1, 0, 0), dtype=bool)
    for zero_center, key in ((True, "X_centered_for_mask"), (False, "X_scaled_for_mask")):
        out, _, _ = op.scale(typ(t["X_for_mask"].astype(dtype)), zero_center=zero_center, mask_obs=mask)
        out = out.toarray() if sparse.issparse(out) else out
        assert np.allclose(out, t[key])
    out, _, _ = op.scale(typ(t["X_original"].astype(dtype)), zero_center=False, max_value=1)
    out = out.toarray() if sparse.issparse(out) else out
    assert np.allclose(out, t["X_scaled_original_clipped"])
    out, _, _ = op.scale(typ(t["X_for_mask"].astype(dtype)), zero_center=False, max_value=1, mask_obs=mask)
    out = out.toarray() if sparse.issparse(out) else out
    assert np.allclose(out, t["X_scaled_for_mask_clipped"])


def test_scale_sparse_zero_center_is_float64_dense():
    out, _, _ = op.scale(sparse.random(30, 7, density=0.3, format="csr", dtype=np.float32, random_state=0))
    assert isinstance(out, np.ndarray) and out.dtype == np.float64
