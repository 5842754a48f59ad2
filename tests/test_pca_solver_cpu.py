"""Host-side PCA solver logic (block Krylov / randomized / covariance_eigh) against the oracle, on CPU,
with the kernel layer replaced by tests/stub_backend.py."""
from __future__ import annotations

import numpy as np
import pytest
from scipy import sparse

from oracle import pca as opca
from scanpy_amd.datasets import synthetic_planted
from scanpy_amd.preprocessing._pca_solver import pca_fit
from stub_backend import CpuStubBackend


def _fit(x, k, **kw):
    be = CpuStubBackend()
    return pca_fit(be.upload(sparse.csr_matrix(x)), k, backend=be, **kw)


@pytest.mark.parametrize("solver", ["arpack", "covariance_eigh"])
def test_golden_A(pca_toy, solver):
    """tests/test_pca.py:225-233: ||abs(A_pca[:, :4]) - abs(X_pca)|| < 2e-5."""
    res = _fit(pca_toy["A_list"].astype(np.float32), 4, svd_solver=solver)
    assert np.linalg.norm(np.abs(pca_toy["A_pca"][:, :4]) - np.abs(res.scores.numpy())) < 2e-5


def test_golden_A_no_center(pca_toy):
    """tests/test_pca.py:264-274 (TruncatedSVD path, golden A_svd)."""
    res = _fit(pca_toy["A_list"].astype(np.float32), 4, zero_center=False)
    assert np.linalg.norm(np.abs(pca_toy["A_svd"][:, :4]) - np.abs(res.scores.numpy())) < 2e-5
    ref = opca.pca_reference(sparse.csr_matrix(pca_toy["A_list"].astype(np.float32)), 4, zero_center=False)
    np.testing.assert_allclose(res.explained_variance, ref["variance"], rtol=1e-4)
    np.testing.assert_allclose(res.explained_variance_ratio, ref["variance_ratio"], rtol=1e-4)


@pytest.mark.parametrize("solver", ["arpack", "covariance_eigh"])
def test_synthetic_vs_reference(solver):
    x, _ = synthetic_planted(4000, 600, n_types=24, seed=1)
    res = _fit(x, 20, svd_solver=solver)
    ref = opca.pca_reference(x, 20)
    tru = opca.pca_dense_f64(x, 20)
    # loadings within 1e-4 up to sign (sign rule is the same, so compare directly, then up to sign)
    err = np.abs(np.abs(res.components) - np.abs(ref["components"])).max()
    err_truth = np.abs(res.components - tru["components"]).max()
    print(solver, "loadings err vs sklearn", err, "vs f64 truth", err_truth, res.info)
    assert err < 1e-4 and err_truth < 1e-5
    np.testing.assert_allclose(res.explained_variance, tru["variance"], rtol=1e-6)
    np.testing.assert_allclose(res.explained_variance_ratio, tru["variance_ratio"], rtol=1e-6)
    np.testing.assert_allclose(res.explained_variance, ref["variance"], rtol=2e-5)
    assert np.abs(np.abs(res.scores.numpy()) - np.abs(ref["X_pca"])).max() < 5e-4
    np.testing.assert_allclose(res.mean, ref["mean"], rtol=1e-6, atol=1e-8)


def test_randomized_close():
    x, _ = synthetic_planted(3000, 400, n_types=12, seed=2)
    res = _fit(x, 10, svd_solver="randomized")
    tru = opca.pca_dense_f64(x, 10)
    np.testing.assert_allclose(res.explained_variance, tru["variance"], rtol=1e-3)


def test_rank_deficient_and_wide():
    rng = np.random.default_rng(0)
    x = sparse.random(30, 200, density=0.2, random_state=rng, format="csr", dtype=np.float32)
    res = _fit(x, 20)
    tru = opca.pca_dense_f64(x, 20)
    np.testing.assert_allclose(res.explained_variance, tru["variance"], rtol=1e-6, atol=1e-9)
    # rank <= 29 after centring: asking for the maximum arpack allows
    res = _fit(x, 29)
    assert res.components.shape == (29, 200)
    np.testing.assert_allclose(res.components @ res.components.T, np.eye(29), atol=1e-8)


def test_errors():
    x = sparse.random(10, 8, density=0.5, format="csr", dtype=np.float32, random_state=1)
    with pytest.raises(ValueError, match="must be between 1 and"):
        _fit(x, 9)
    with pytest.raises(ValueError, match="strictly less"):
        _fit(x, 8)
    with pytest.raises(ValueError, match="not supported"):
        _fit(x, 2, svd_solver="full")


def test_reproducible_and_seed():
    x, _ = synthetic_planted(2000, 300, n_types=10, seed=3)
    a = _fit(x, 10, seed=0)
    b = _fit(x, 10, seed=0)
    np.testing.assert_array_equal(a.components, b.components)
    np.testing.assert_array_equal(a.scores.numpy(), b.scores.numpy())


def test_gram_route_matches_krylov_route():
    """Default (covariance/Gram + Chebyshev subspace) and block Krylov on the CSR operator agree to 1e-7."""
    x, _ = synthetic_planted(3000, 700, n_types=30, seed=4)
    a = _fit(x, 25)
    b = _fit(x, 25, block_size=64)
    assert a.info["solver"] == "gram" and a.info["dense_solver"] == "chebyshev_subspace", a.info
    assert b.info["solver"] == "arpack"
    assert np.abs(a.components - b.components).max() < 1e-7
    np.testing.assert_allclose(a.explained_variance, b.explained_variance, rtol=1e-9)
    assert a.info["residual"] < 2e-8


def test_dense_topk_flat_tail():
    """Chebyshev-filtered subspace iteration on a spectrum with a nearly flat unwanted tail and on a rank-deficient
    matrix (smallest Ritz value 0)."""
    import torch

    from scanpy_amd.preprocessing._pca_solver import _dense_topk_eigh

    rng = np.random.default_rng(0)
    g, k = 640, 20
    q, _ = np.linalg.qr(rng.standard_normal((g, g)))
    for lam in (np.concatenate([np.linspace(5, 1.2, k), np.linspace(1.0, 0.9, g - k)]),
                np.concatenate([np.linspace(5, 1.2, 60), np.zeros(g - 60)])):
        amat = torch.from_numpy((q * lam) @ q.T)
        info = {}
        got_l, got_v = _dense_topk_eigh(amat, k, np.random.default_rng(1), 2e-8, info)
        np.testing.assert_allclose(got_l.numpy(), lam[:k], rtol=1e-10)
        err = np.abs(np.abs(got_v.numpy().T @ q[:, :k]) - np.eye(k)).max()
        assert err < 1e-6, (err, info)


@pytest.mark.parametrize("chunk", [1, 97, 400, 5000])
def test_chunked_rows_equal_one_shot(chunk):
    """the fixed-point Gram matrix is additive over row chunks: the streamed fit IS the one-shot fit, bit for bit"""
    from scanpy_amd.datasets import synthetic_planted
    from scanpy_amd.preprocessing._pca_solver import _ChunkedRows

    x, _ = synthetic_planted(1200, 150, n_types=6, seed=3)
    be = CpuStubBackend()
    one = pca_fit(be.upload(x), 10, backend=be)
    rows = _ChunkedRows([x[i:i + chunk] for i in range(0, x.shape[0], chunk)], x.shape[1])
    assert not rows.resident
    many = pca_fit(rows, 10, backend=be)
    np.testing.assert_array_equal(many.components, one.components)
    np.testing.assert_array_equal(many.explained_variance, one.explained_variance)
    np.testing.assert_array_equal(many.scores.numpy(), one.scores.numpy())
    if chunk < 1200:
        assert many.info["row_chunks"] == -(-1200 // chunk)


def test_chunked_rows_prefetch_protocol():
    """the next block's upload is issued right after the current block is handed to the caller (so it overlaps that
    block's kernels), `wait_prefetch` precedes every use, and resident blocks are uploaded once for all passes"""
    from scanpy_amd.preprocessing._pca_solver import _ChunkedRows

    class Recorder:
        def __init__(self):
            self.log = []

        def upload_prefetch(self, c):
            self.log.append(("upload", c))
            return (c,)

        def wait_prefetch(self):
            self.log.append(("wait",))

    class Chunk:
        def __init__(self, tag, rows):
            self.tag, self.shape = tag, (rows, 5)
            self.data = self.indices = np.zeros(4, dtype=np.float32)

        def __repr__(self):
            return self.tag

    chunks = [Chunk("a", 3), Chunk("b", 3), Chunk("c", 2)]
    rows = _ChunkedRows(chunks, 5)
    assert (rows.n_rows, rows.n_cols, rows.n_chunks, rows.resident) == (8, 5, 3, False)
    be = Recorder()
    for h in rows.handles(be):
        be.log.append(("use", h[0]))
    a, b, c = chunks
    assert be.log == [("upload", a), ("wait",), ("use", a), ("upload", b), ("wait",), ("use", b), ("upload", c), ("wait",),
                      ("use", c)]
    n_first = len(be.log)
    list(rows.handles(be))  # streamed mode: a second pass uploads again
    assert len(be.log) == n_first + 6
    res = _ChunkedRows(chunks, 5, resident_budget_bytes=1 << 30)
    be2 = Recorder()
    list(res.handles(be2))
    n_up = sum(1 for e in be2.log if e[0] == "upload")
    list(res.handles(be2))
    assert res.resident and n_up == 3 and sum(1 for e in be2.log if e[0] == "upload") == 3
