"""TEST INFRASTRUCTURE -- an independent restatement of R's `loess()` for one predictor (netlib dloess, surface =
"interpolate"), recursive and unvectorised on purpose; the product's version is scanpy_amd/preprocessing/_loess.py.

Both are checked against the same known answers: the `variance.expected` column Seurat wrote for pbmc3k
(reference `tests/_scripts/seurat_hvg_v3.csv.gz`, committed as tests/golden/loess_seurat_v3.npz).  The reference
itself calls scikit-misc's wrapper of the same Fortran (`src/scanpy/preprocessing/_highly_variable_genes.py:222-225`).
"""
from __future__ import annotations

import numpy as np


def local_quadratic(xs, ys, z, q, degree=2):
    """(value, slope) at z of the tricube-weighted polynomial fit over the q points of xs nearest to z"""
    dist = np.abs(xs - z)
    near = np.argsort(dist, kind="stable")[:q]
    rho = dist[near].max()
    w = np.where(dist[near] < rho, (1 - (dist[near] / rho) ** 3) ** 3, 0.0)
    a = np.vander(xs[near] - z, degree + 1, increasing=True)
    wa = a * w[:, None]
    coef = np.linalg.solve(a.T @ wa, wa.T @ ys[near])  # normal equations (the product uses a QR-based lstsq)
    return coef[0], coef[1]


def loess(x, y, span=0.3, degree=2, cell=0.2):
    x, y = np.asarray(x, float), np.asarray(y, float)
    n = len(x)
    order = np.argsort(x, kind="stable")
    xs, ys = x[order], y[order]
    q = min(n, int(np.floor(n * span + 1e-5)))
    fc = int(np.floor(n * span * cell))
    margin = 0.005 * (xs[-1] - xs[0])
    verts = [xs[0] - margin, xs[-1] + margin]

    def build(l, u, vlo, vhi):  # noqa: E741 - 1-based inclusive indices, as in ehg124
        if u - l + 1 <= fc:
            return
        m = (l + u) // 2
        offset = 0
        while l <= m + offset < u:
            if xs[m + offset - 1] == xs[m + offset]:  # x(pi(m+offset)) == x(pi(m+offset+1)), 1-based
                offset = -offset
                if offset >= 0:
                    offset += 1
            else:
                m += offset
                break
        t = xs[m - 1]
        if t in (vlo, vhi):
            return
        verts.append(t)
        build(l, m, vlo, t)
        build(m + 1, u, t, vhi)

    build(1, n, verts[0], verts[1])
    v = np.array(sorted(set(verts)))
    gs = [local_quadratic(xs, ys, z, q, degree) for z in v]
    out = np.empty(n)
    for i, z in enumerate(x):
        j = min(max(int(np.searchsorted(v, z, side="left")) - 1, 0), len(v) - 2)
        h = v[j + 1] - v[j]
        u = (z - v[j]) / h
        (g0, s0), (g1, s1) = gs[j], gs[j + 1]
        out[i] = ((1 - u) ** 2 * (1 + 2 * u) * g0 + u ** 2 * (3 - 2 * u) * g1
                  + (u * (1 - u) ** 2 * s0 - u ** 2 * (1 - u) * s1) * h)
    return out
