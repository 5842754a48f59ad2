"""kNN estimator for the reference's plug-in protocol (`KnnTransformerLike`,
src/scanpy/neighbors/_types.py:53-64; template src/scanpy/neighbors/_backends/rapids.py:39-101).

`MI355XKNNTransformer(n_neighbors=15)` can be passed as `transformer=` to UNMODIFIED upstream
`scanpy.pp.neighbors`: `fit_transform(X)` returns a CSR with a constant number of stored entries per
row.  With `include_self=False` (default) rows hold the n_neighbors-1 nearest OTHER cells (RAPIDS style,
_common.py:88-91 re-inserts the self column), so `obsp['distances']` equals the reference's default
result; `include_self=True` gives sklearn-style rows (self first with an explicit 0, n_neighbors+1
entries).
"""
from __future__ import annotations

import numpy as np
from scipy import sparse


# metrics the Euclidean kernel serves exactly (the reference takes any sklearn / scipy metric name, neighbors/_types.py:23-50):
# 'sqeuclidean' = the same lists with squared distances; 'cosine' / 'correlation' = the Euclidean search on unit-length
# (/ row-centred unit-length) rows, 1 - cos = |x^ - y^|^2 / 2
METRICS = ("euclidean", "l2", "sqeuclidean", "cosine", "correlation")


def knn_search_device(x, k: int, *, q_begin: int = 0, n_query: int | None = None, metric: str = "euclidean",
                      nprobe: int | None = None):
    """kNN on the GPU, results left on the device: (indices int32 [nq, k], distances float64 [nq, k]); column 0
    is the row itself with distance exactly 0.  Exact unless `nprobe` > 0 (the approximate IVF mode: every query sees
    the rows of the `nprobe` quantiser cells nearest to its own; exact among those).

    metric 'cosine' (sklearn: 1 - x.y / (|x||y|)): on unit-length rows the Euclidean order IS the cosine order and
    1 - cos = |x^ - y^|^2 / 2, so the rows are normalised on the device and the Euclidean kernel does the search."""
    import torch

    from .. import _kernels
    from .._device import require_gpu

    if metric not in METRICS:
        raise ValueError(f"metric={metric!r}: the MI355X kNN kernel offers {METRICS}")
    dev = require_gpu()
    if sparse.issparse(x):
        x = x.toarray()
    xd = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
    if metric in ("cosine", "correlation"):
        x64 = xd.to(torch.float64)
        if metric == "correlation":  # scipy: 1 - (x - mean x) . (y - mean y) / (|x - mean x| |y - mean y|)
            x64 = x64 - x64.mean(dim=1, keepdim=True)
        norm = torch.linalg.norm(x64, dim=1, keepdim=True)
        if bool((norm == 0).any()):
            raise ValueError(f"metric={metric!r} is undefined for " + ("all-zero rows" if metric == "cosine" else "constant rows"))
        xd = (x64 / norm).to(torch.float32).contiguous()
    idx, dist, _ = _kernels.knn(xd, k, q_begin=q_begin, n_query=n_query, nprobe=nprobe)
    if metric in ("cosine", "correlation"):
        dist = 0.5 * dist * dist
    elif metric == "sqeuclidean":
        dist = dist * dist
    return idx, dist


def knn_search(x, k: int, *, q_begin: int = 0, n_query: int | None = None, metric: str = "euclidean",
               nprobe: int | None = None):
    """`knn_search_device` with the results on the host: (indices int64 [nq, k], distances float64 [nq, k])."""
    idx, dist = knn_search_device(x, k, q_begin=q_begin, n_query=n_query, metric=metric, nprobe=nprobe)
    return idx.cpu().numpy().astype(np.int64), dist.cpu().numpy()


class MI355XKNNTransformer:
    """sklearn-estimator-shaped kNN on MI355X (`fit`, `transform`, `fit_transform`, `get_params`, `set_params`): exact
    by default; `nprobe=p` selects the approximate IVF mode (every query sees the `p` quantiser cells nearest to its
    own -- what the reference's `transformer='pynndescent'` default above 8192 cells trades, recall for time)."""

    def __init__(self, n_neighbors: int = 15, *, metric: str = "euclidean", include_self: bool = False,
                 nprobe: int | None = None):
        if metric not in METRICS:
            msg = f"metric={metric!r}: the MI355X kNN kernel offers {METRICS}"
            raise ValueError(msg)
        self.n_neighbors = n_neighbors
        self.metric = metric
        self.include_self = include_self
        self.nprobe = self._check_nprobe(nprobe)
        self._fit_x = None

    @staticmethod
    def _check_nprobe(nprobe):
        """None / 0 = exact; a positive integer = cells probed.  Checked wherever the value can arrive (`__init__`,
        `set_params` -- sklearn's `clone` / grid searches go through that --, and again at `transform`)."""
        if nprobe is None:
            return None
        if isinstance(nprobe, bool) or int(nprobe) != nprobe or int(nprobe) < 0:
            raise ValueError(f"nprobe={nprobe!r}: expected None (exact) or a positive number of cells")
        return int(nprobe)

    def get_params(self, deep: bool = True) -> dict:
        return dict(n_neighbors=self.n_neighbors, metric=self.metric, include_self=self.include_self, nprobe=self.nprobe)

    def set_params(self, **params):
        for k, v in params.items():
            if k not in ("n_neighbors", "metric", "include_self", "nprobe"):
                raise ValueError(f"Invalid parameter {k!r}")
            if k == "nprobe":
                v = self._check_nprobe(v)
            if k == "metric" and v not in METRICS:
                raise ValueError(f"metric={v!r}: the MI355X kNN kernel offers {METRICS}")
            setattr(self, k, v)
        return self

    def fit(self, x, y=None):
        self._fit_x = x
        return self

    def transform(self, x) -> sparse.csr_matrix:
        if self._fit_x is None:
            raise RuntimeError("call fit first")
        if x is not self._fit_x and (x.shape != self._fit_x.shape or not _same(x, self._fit_x)):
            msg = "MI355XKNNTransformer only answers queries for the fitted data (fit_transform semantics)"
            raise NotImplementedError(msg)
        n = x.shape[0]
        self.nprobe = self._check_nprobe(self.nprobe)
        if self.include_self:  # sklearn style: self + n_neighbors others
            k = min(self.n_neighbors + 1, n)
            idx, dist = knn_search(x, k, metric=self.metric, nprobe=self.nprobe)
        else:  # RAPIDS style: n_neighbors - 1 others, no self
            k = min(self.n_neighbors, n)
            idx, dist = knn_search(x, k, metric=self.metric, nprobe=self.nprobe)
            idx, dist = idx[:, 1:], dist[:, 1:]
        kk = idx.shape[1]
        indptr = np.arange(0, n * kk + 1, kk)
        return sparse.csr_matrix((dist.ravel(), idx.ravel(), indptr), shape=(n, n))

    def fit_transform(self, x, y=None) -> sparse.csr_matrix:
        return self.fit(x).transform(x)


def _same(a, b) -> bool:
    if sparse.issparse(a) or sparse.issparse(b):
        return (abs(sparse.csr_matrix(a) - sparse.csr_matrix(b))).nnz == 0
    return bool(np.array_equal(np.asarray(a), np.asarray(b)))
