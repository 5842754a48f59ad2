"""Oracle: PCA exactly as the reference computes it on CSR input.  Test infrastructure.

Reference: src/scanpy/preprocessing/_pca/__init__.py:284-308 -- for sparse, zero_center=True,
svd_solver=None the reference builds sklearn.decomposition.PCA(n_components, svd_solver='arpack',
random_state=0) and calls fit_transform; scikit-learn 1.7.2 is installed here, so this IS the
reference arithmetic (sklearn/decomposition/_pca.py:704-793).  `pca_dense_f64` is an independent
float64 ground truth (explicit centering + LAPACK eigh) used to show which side is closer to the
truth when float32 ARPACK and the device solver disagree at the 1e-5 level.
"""
from __future__ import annotations

import numpy as np
from scipy import sparse


def pca_reference(x, n_comps: int, *, random_state: int = 0, zero_center: bool = True, svd_solver: str = "arpack"):
    """-> dict(X_pca, components (k,g), variance, variance_ratio, mean)."""
    if zero_center:
        from sklearn.decomposition import PCA

        p = PCA(n_components=n_comps, svd_solver=svd_solver, random_state=random_state)
        x_pca = p.fit_transform(x)
        mean = p.mean_
    else:  # _pca/__init__.py:309-336 -> TruncatedSVD(algorithm=svd_solver)
        from sklearn.decomposition import TruncatedSVD

        p = TruncatedSVD(n_components=n_comps, random_state=random_state, algorithm=svd_solver)
        x_pca = p.fit_transform(x)
        mean = None
    return dict(
        X_pca=x_pca,
        components=p.components_,
        variance=p.explained_variance_,
        variance_ratio=p.explained_variance_ratio_,
        mean=mean,
    )


def svd_flip_v(u: np.ndarray, vt: np.ndarray):
    """sklearn.utils.extmath.svd_flip(u_based_decision=False): largest-|.| entry of each row of Vt positive."""
    max_abs = np.argmax(np.abs(vt), axis=1)
    signs = np.sign(vt[np.arange(vt.shape[0]), max_abs])
    signs[signs == 0] = 1
    return u * signs[None, :], vt * signs[:, None]


def pca_dense_f64(x, n_comps: int):
    """Float64 ground truth: covariance eigendecomposition of explicitly centred data."""
    xd = np.asarray(x.todense() if sparse.issparse(x) else x, dtype=np.float64)
    n = xd.shape[0]
    mean = xd.mean(axis=0)
    xc = xd - mean
    cov = xc.T @ xc / (n - 1)
    w, v = np.linalg.eigh(cov)
    order = np.argsort(w)[::-1][:n_comps]
    w, v = w[order], v[:, order]
    scores = xc @ v
    scores, vt = svd_flip_v(scores, v.T)
    total_var = xc.var(axis=0, ddof=1).sum()
    return dict(X_pca=scores, components=vt, variance=w, variance_ratio=w / total_var, mean=mean)
