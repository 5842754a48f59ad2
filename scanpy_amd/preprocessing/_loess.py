"""LOESS of one predictor as R's `loess()` / scikit-misc compute it (the netlib `dloess` code with its default
`surface="interpolate"`), for `highly_variable_genes(flavor='seurat_v3')`.

The reference calls `skmisc.loess.loess(x, y, span=span, degree=2).fit()` (`_highly_variable_genes.py:222-225`);
scikit-misc is not installed here and the fit is NOT a plain local regression at every point: dloess builds a k-d tree
over x, fits the local quadratic (tricube weights over the q = floor(n * span) nearest points) only at the tree's
vertices, keeps value and slope there, and blends cubic Hermite pieces in between.  Restated from the algorithm
(Cleveland & Grosse 1991; routines ehg126 / ehg124 / ehg127 / ehg128 of loessf.f) for d = 1:

  * bounding box = data range widened by 0.5 % on each side; its ends are the first two vertices;
  * a cell holding more than fc = floor(n * span * cell) points (cell = 0.2) is cut at its median point m = (l + u) / 2,
    moved to the nearest index (trying m, m+1, m-1, m+2, ...) where x[m] != x[m+1] so that ties stay together; the new
    vertex sits AT x[m] (the largest value of the low son), and a cell whose cut value equals one of its own vertices
    stays a leaf;
  * prediction inside a cell [v0, v1]: phi0 g0 + phi1 g1 + (psi0 s0 + psi1 s1)(v1 - v0) with the cubic Hermite basis.

Pinned to Seurat's own output: `tests/_scripts/seurat_hvg_v3.csv.gz` of the reference holds, for 13714 genes, the
`variance.expected` that R's loess produced from (`mean`, `variance`); this module reproduces it to 2.5e-13 relative
(`tests/test_loess_cpu.py`; a plain pointwise LOESS is off by 2.4e-2).  Host code on purpose: ~30 weighted 3-column
least-squares fits over ~0.3 n points each -- microseconds of work next to the device sweeps that produce x and y.
"""
from __future__ import annotations

import numpy as np


def _vertex_fit(xs: np.ndarray, ys: np.ndarray, z: float, q: int, degree: int) -> tuple[float, float]:
    """value and slope at z of the tricube-weighted local polynomial over the q nearest of the sorted xs (ehg127)"""
    n = xs.shape[0]
    lo = int(np.clip(np.searchsorted(xs, z) - q // 2, 0, n - q))
    while lo + q < n and xs[lo + q] - z < z - xs[lo]:
        lo += 1
    while lo > 0 and z - xs[lo - 1] < xs[lo + q - 1] - z:
        lo -= 1
    dx = xs[lo:lo + q] - z
    dist = np.abs(dx)
    rho = dist.max()
    if rho <= 0:
        raise ValueError("loess: the span covers identical x values only")
    w = np.clip(1.0 - (dist / rho) ** 3, 0.0, None) ** 3
    sw = np.sqrt(w)
    design = np.vander(dx, degree + 1, increasing=True) * sw[:, None]
    coef, *_ = np.linalg.lstsq(design, ys[lo:lo + q] * sw, rcond=None)
    return float(coef[0]), float(coef[1]) if degree >= 1 else 0.0


def _kd_vertices(xs: np.ndarray, fc: int) -> np.ndarray:
    """vertices of dloess' k-d tree over the sorted xs (ehg126 + ehg124, one dimension)"""
    n = xs.shape[0]
    a, b = float(xs[0]), float(xs[-1])
    mu = 0.005 * max(b - a, 1e-10 * max(abs(a), abs(b)) + 1e-30)
    verts = [a - mu, b + mu]
    cells = [(0, n - 1, verts[0], verts[1])]  # inclusive point range, low / high vertex
    while cells:
        l, u, vlo, vhi = cells.pop()
        if u - l + 1 <= fc:
            continue
        m = (l + u + 2) // 2 - 1  # Fortran's (l + u) / 2 on 1-based indices
        offset = 0
        while l <= m + offset < u:  # keep ties together: nearest index where the value changes
            if xs[m + offset] == xs[m + offset + 1]:
                offset = -offset
                if offset >= 0:
                    offset += 1
                continue
            m += offset
            break
        t = float(xs[m])
        if t == vlo or t == vhi:
            continue
        verts.append(t)
        cells.append((l, m, vlo, t))
        cells.append((m + 1, u, t, vhi))
    return np.unique(np.asarray(verts))


def loess_fit(x, y, *, span: float = 0.3, degree: int = 2, cell: float = 0.2) -> np.ndarray:
    """fitted values of `loess(y ~ x, span, degree)` at the x themselves (R / scikit-misc defaults otherwise)"""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    if x.ndim != 1 or x.shape != y.shape:
        raise ValueError("loess_fit expects two vectors of one length")
    if degree not in (1, 2):
        raise ValueError("degree must be 1 or 2")
    n = x.shape[0]
    q = min(n, int(np.floor(n * span + 1e-5)))
    if q < degree + 1:
        raise ValueError(f"loess: span {span} covers {q} of {n} points, too few for a degree-{degree} fit")
    order = np.argsort(x, kind="stable")
    xs, ys = x[order], y[order]
    v = _kd_vertices(xs, int(np.floor(n * span * cell)))
    fits = np.array([_vertex_fit(xs, ys, float(z), q, degree) for z in v])
    g, s = fits[:, 0], fits[:, 1]
    j = np.clip(np.searchsorted(v, x, side="left") - 1, 0, v.shape[0] - 2)
    h = v[j + 1] - v[j]
    u = (x - v[j]) / h
    return ((1 - u) ** 2 * (1 + 2 * u) * g[j] + u ** 2 * (3 - 2 * u) * g[j + 1]
            + (u * (1 - u) ** 2 * s[j] - u ** 2 * (1 - u) * s[j + 1]) * h)
