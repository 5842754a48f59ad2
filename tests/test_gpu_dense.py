"""The hand-written float64 side of the PCA (csrc/dense.hip): MFMA GEMM, CholeskyQR factor, one-workgroup Jacobi,
the top-k eigensolver and the one-call `scamd_pca_csr_f32` -- each against numpy / the reference's sklearn call."""
from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    from scanpy_amd import _kernels

    return _kernels


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize(("kdim", "m", "n"), [(2000, 2000, 128), (2000, 128, 128), (37, 50, 70), (4, 16, 16), (1001, 33, 129), (3, 1, 1)])
def test_dgemm_tn(K, kdim, m, n):
    rng = np.random.default_rng(kdim + m + n)
    p, q = rng.standard_normal((kdim, m)), rng.standard_normal((kdim, n))
    got = K.dense_debug(1, _dev(p), _dev(q)).cpu().numpy()
    ref = p.T @ q
    assert np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()) * np.sqrt(kdim)
    # fixed reduction order: bitwise reproducible
    assert np.array_equal(got, K.dense_debug(1, _dev(p), _dev(q)).cpu().numpy())


@pytest.mark.parametrize("b", [128, 96, 50, 7])
def test_cholqr_factor(K, b):
    rng = np.random.default_rng(b)
    z = rng.standard_normal((900, b)) * np.exp(rng.uniform(-3, 3, size=b))[None, :]  # badly scaled columns
    s, bad = K.dense_debug(2, _dev(z.T @ z))
    assert bad == 0
    q = z @ s.cpu().numpy()
    assert np.abs(q.T @ q - np.eye(b)).max() < 1e-10
    # a singular Gram matrix is reported, not factorised
    z[:, -1] = z[:, 0]
    s2, bad = K.dense_debug(2, _dev(z.T @ z))
    s2 = s2.cpu().numpy()
    # (a pivot of +1e-17 instead of -1e-17 "succeeds" with a factor of ~1e8: either way the caller sees it)
    assert bad == 1 or not np.isfinite(s2).all() or np.abs(s2).max() > 1e5


@pytest.mark.parametrize("b", [128, 127, 64, 9, 2])
def test_jacobi_eigh(K, b):
    rng = np.random.default_rng(b + 5)
    m = rng.standard_normal((3 * b, b)) * np.exp(rng.uniform(-2, 2, size=b))[None, :]
    t = m.T @ m
    theta, y, sweeps = K.dense_debug(3, _dev(t))
    theta, y = theta.cpu().numpy(), y.cpu().numpy()
    ref = np.linalg.eigvalsh(t)[::-1]
    print(f"b={b}: {sweeps} sweeps, max rel eigenvalue error {np.abs(theta - ref).max() / ref[0]:.2e}")
    assert np.abs(theta - ref).max() < 1e-12 * ref[0]
    assert np.abs(y.T @ y - np.eye(b)).max() < 1e-11
    assert np.abs(t @ y - y * theta[None, :]).max() < 1e-11 * ref[0]
    assert (np.diff(theta) <= 0).all()


def _spectrum_matrix(g, rng, decay):
    q, _ = np.linalg.qr(rng.standard_normal((g, g)))
    lam = decay(np.arange(g))
    return (q * lam[None, :]) @ q.T, lam


@pytest.mark.parametrize(("g", "k"), [(2000, 50), (700, 30), (300, 50), (100, 20), (128, 50)])
def test_eigh_topk(K, g, k):
    rng = np.random.default_rng(g + k)
    a, lam = _spectrum_matrix(g, rng, lambda i: 100.0 * 0.93 ** np.minimum(i, 80) * (1.0 + 0.0 * i) * np.where(i < 80, 1.0, 0.5))
    a = 0.5 * (a + a.T)
    ad = _dev(a)
    lv, v, info = K.eigh_topk(ad, k)
    lv, v = lv.cpu().numpy(), v.cpu().numpy()
    ref_l, ref_v = np.linalg.eigh(a)
    ref_l, ref_v = ref_l[::-1][:k], ref_v[:, ::-1][:, :k]
    print(g, k, info)
    assert np.abs(lv - ref_l).max() < 1e-9 * ref_l[0]
    assert np.abs(a @ v - v * lv[None, :]).max() < 1e-6 * ref_l[0]
    assert np.abs(np.abs(np.sum(v * ref_v, axis=0)) - 1.0).max() < 1e-8  # same vectors up to sign
    assert np.abs(v.T @ v - np.eye(k)).max() < 1e-10


def test_eigh_topk_unsupported_shapes(K):
    from scanpy_amd._lib import ScamdError

    a = _dev(np.eye(150))
    with pytest.raises(ScamdError):
        K.eigh_topk(a, 50)  # 128 < g < 2 * block (block = k + 32 rounded up to 16 = 96)
    with pytest.raises(ScamdError):
        K.eigh_topk(_dev(np.eye(1000)), 110)  # k beyond the block


@pytest.mark.parametrize("zero_center", [True, False])
def test_pca_csr_entry_vs_sklearn(K, zero_center):
    """`scamd_pca_csr_f32` (one C call) against the reference's own sklearn call on the planted synthetic matrix"""
    from oracle import pca as opca
    from scanpy_amd.datasets import synthetic_planted

    n, g, k = 30000, 2000, 50
    x, _ = synthetic_planted(n, g, seed=3)
    import torch

    ip = torch.from_numpy(x.indptr.astype(np.int64)).cuda()
    ix = torch.from_numpy(x.indices.astype(np.int32)).cuda()
    dv = torch.from_numpy(x.data.astype(np.float32)).cuda()
    scores, comps, var, ratio, mean, info = K.pca_csr(ip, ix, dv, n, g, k, zero_center=zero_center)
    print(info)
    ref = opca.pca_reference(x, k, zero_center=zero_center, svd_solver="arpack")
    comps, var, ratio = comps.cpu().numpy(), var.cpu().numpy(), ratio.cpu().numpy()
    err = np.abs(np.abs(comps) - np.abs(ref["components"])).max()
    print("loading err", err, "variance rel err", np.abs(var / ref["variance"] - 1).max())
    assert err < 1e-4
    assert np.abs(var / ref["variance"] - 1).max() < 2e-5
    assert np.abs(ratio / ref["variance_ratio"] - 1).max() < 2e-5
    # sign convention: largest-|.| loading of every component positive
    assert (comps[np.arange(k), np.abs(comps).argmax(axis=1)] > 0).all()
    s = scores.cpu().numpy()
    sign = np.sign(np.sum(comps * ref["components"], axis=1))
    assert np.abs(s * sign[None, :] - ref["X_pca"]).max() < 5e-4 * np.abs(ref["X_pca"]).max()
    if zero_center:
        assert np.abs(mean.cpu().numpy() - np.asarray(x.mean(axis=0)).ravel()).max() < 1e-6


def test_pca_csr_entry_equals_python_route(K):
    """same model as the building-block route of `pca_fit` (multi-rank capable), to the solver tolerance"""
    from scanpy_amd.datasets import synthetic_planted
    from scanpy_amd.preprocessing._pca_solver import GpuBackend, pca_fit

    n, g, k = 20000, 1500, 40
    x, _ = synthetic_planted(n, g, seed=5, n_types=30)
    backend = GpuBackend()
    h = backend.upload(x)
    res = pca_fit(h, k, backend=backend)
    scores, comps, var, ratio, mean, info = K.pca_csr(h[0], h[1], h[2], n, g, k)
    assert np.abs(np.abs(comps.cpu().numpy()) - np.abs(res.components)).max() < 1e-6
    assert np.abs(var.cpu().numpy() / res.explained_variance - 1).max() < 1e-9


@pytest.mark.parametrize("k", [120, 150])
def test_pca_more_components_than_one_block(K, k, monkeypatch):
    """n_comps > 96 (`sc.pp.pca` takes any n_comps < min(n, g): src/scanpy/preprocessing/_pca/__init__.py:234-236): the device
    eigensolver delivers them in batches of 96 on the DEFLATED matrix (csrc/dense.hip: dense_topk_batched) -- no torch.linalg
    call is reachable (SCAMD_ALLOW_TORCH_FALLBACK unset).  Against the reference's sklearn ARPACK call: variances to 2e-5; the
    loadings of the separated part of the spectrum to 1e-4 up to sign; for the components inside the noise bulk (eigenvalue
    gaps of 1e-4 relative: single vectors are ill conditioned for ANY solver) the spanned subspace is compared instead."""
    import scanpy_amd as sc
    from oracle import pca as opca
    from scanpy_amd.datasets import synthetic_planted

    monkeypatch.delenv("SCAMD_ALLOW_TORCH_FALLBACK", raising=False)
    n, g = 20000, 1200
    x, _ = synthetic_planted(n, g, seed=7, n_types=40)
    adata = sc.AnnData(x)
    sc.pp.pca(adata, n_comps=k)
    comps = adata.varm["PCs"].T.astype(np.float64)
    ref = opca.pca_reference(x, k, zero_center=True, svd_solver="arpack")
    rc = ref["components"]
    var = adata.uns["pca"]["variance"]
    assert np.abs(var / ref["variance"] - 1).max() < 2e-5
    assert np.abs(comps @ comps.T - np.eye(k)).max() < 1e-5  # orthonormal ACROSS the batches too
    lead = 30  # the planted programmes: well separated eigenvalues
    err = np.abs(np.abs(comps[:lead]) - np.abs(rc[:lead])).max()
    print("k", k, "leading loading err", err, "all", np.abs(np.abs(comps) - np.abs(rc)).max())
    assert err < 1e-4
    # subspaces: the cosines of the principal angles between the two k-dimensional spaces (all but the very last directions,
    # whose neighbours outside the space are as close as their neighbours inside)
    cosines = np.linalg.svd(comps @ rc.T, compute_uv=False)
    print("smallest principal cosines", cosines[-4:])
    assert cosines[: k - 3].min() > 1 - 1e-6
    s = adata.obsm["X_pca"]
    sign = np.sign(np.sum(comps[:lead] * rc[:lead], axis=1))
    assert np.abs(s[:, :lead] * sign[None, :] - ref["X_pca"][:, :lead]).max() < 5e-4 * np.abs(ref["X_pca"]).max()
