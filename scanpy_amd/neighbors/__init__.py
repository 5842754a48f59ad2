"""`sc.pp.neighbors` on MI355X (src/scanpy/neighbors/__init__.py:88-299, 399-710).

kNN search and the umap connectivities run in HIP kernels (`scamd_knn_l2_f32`,
`scamd_fuzzy_simplicial_set_f32`); slot names, `uns[...]['params']`, the self-column conventions of
src/scanpy/neighbors/_common.py and the error behaviour follow the reference.  The search is EXACT
(the reference's `transformer='sklearn'` semantics) for every n -- where the reference would switch to
approximate NN-descent at n >= 8192 (neighbors/__init__.py:734-739) this path stays exact, also for
`transformer='pynndescent'`.  The approximate search is opt-in: `transformer='ivf'` (probes
`settings.knn_nprobe` cells of the k-means quantiser, `scamd_knn_l2_ivf_f32`) or an
`MI355XKNNTransformer(nprobe=...)` instance.
"""
from __future__ import annotations

import warnings
from types import MappingProxyType

import numpy as np
from scipy import sparse

from .._anndata import is_anndata
from .._utils import _UNSET, choose_representation, resolve_seed
from ._common import (
    get_indices_distances_from_dense_matrix,
    get_indices_distances_from_sparse_matrix,
    graph_from_device,
    sparse_distances_from_device,
)
from ._transformer import METRICS, MI355XKNNTransformer, knn_search_device

__all__ = ["neighbors", "Neighbors", "MI355XKNNTransformer"]

_METHODS = ("umap", "gauss", "jaccard")


def _to_device(a, dtype):
    """host array or torch tensor (already on the device: the built-in search hands its lists over without a round
    trip through the host) -> contiguous device tensor of `dtype`"""
    import torch

    from .._device import require_gpu

    if isinstance(a, torch.Tensor):
        return a.to(dtype).contiguous()
    np_dtype = {torch.int32: np.int32, torch.float32: np.float32}[dtype]
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np_dtype)).to(require_gpu())


def _connectivities_umap(knn_indices, knn_dists, n_obs: int) -> sparse.csr_matrix:
    """neighbors/_connectivity.py:103-138 -> scamd_fuzzy_simplicial_set_f32."""
    import torch

    from .. import _kernels

    indptr, indices, data, _, _ = _kernels.fuzzy_simplicial_set(_to_device(knn_indices, torch.int32),
                                                                _to_device(knn_dists, torch.float32))
    return graph_from_device(indptr, indices, data, n_obs)


def _connectivities_kernel(method: str, knn_indices, knn_dists, n_obs: int) -> sparse.csr_matrix:
    """neighbors/_connectivity.py:21-100 ('gauss', sparse branch) / :141-186 ('jaccard') ->
    scamd_gauss_connectivities_f32 / scamd_jaccard_connectivities_f32."""
    import torch

    from .. import _kernels

    idx = _to_device(knn_indices, torch.int32)
    if method == "gauss":
        indptr, indices, data = _kernels.gauss_connectivities(idx, _to_device(knn_dists, torch.float32))
    else:
        indptr, indices, data = _kernels.jaccard_connectivities(idx)
    return graph_from_device(indptr, indices, data, n_obs)


class Neighbors:
    """Data represented as graph of nearest neighbors (the slice of the reference class the path needs)."""

    def __init__(self, adata, *, n_dcs=None, neighbors_key=None):
        self._adata = adata
        self._distances = None
        self._connectivities = None
        self.n_neighbors = None
        self.knn = None
        self.rp_forest = None
        self._cc = None
        key = "neighbors" if neighbors_key is None else neighbors_key
        if key in adata.uns:  # restore an existing graph (neighbors/__init__.py:437-474)
            info = adata.uns[key]
            ck, dk = info.get("connectivities_key", "connectivities"), info.get("distances_key", "distances")
            if ck in adata.obsp:
                self._connectivities = adata.obsp[ck]
            if dk in adata.obsp:
                self._distances = adata.obsp[dk]
            self.knn = sparse.issparse(self._distances) or sparse.issparse(self._connectivities)
            if "params" in info:
                self.n_neighbors = info["params"]["n_neighbors"]
            else:  # estimate from the stored graph (neighbors/__init__.py:446-464)

                def count_nonzero(a) -> int:
                    return a.count_nonzero() if sparse.issparse(a) else int(np.count_nonzero(a))

                if self._connectivities is None:
                    self.n_neighbors = int(count_nonzero(self._distances) / self._distances.shape[0])
                else:
                    self.n_neighbors = int(count_nonzero(self._connectivities) / self._connectivities.shape[0] / 2)
            self._cc = None  # `:466-472`, lazily (see `_connected_components`)

    @property
    def _connected_components(self):
        """(n_components, labels) of the connectivity graph.  The reference computes it eagerly at the end of
        `compute_neighbors` / on restore (scipy `connected_components`: ~0.3-3 s at 1M cells, more than the GPU spends
        on the whole path); nothing on this path reads it, so it is computed when first asked for."""
        if self._cc is None and sparse.issparse(self._connectivities):
            from scipy.sparse.csgraph import connected_components

            self._cc = connected_components(self._connectivities)
        return self._cc

    @property
    def _number_connected_components(self):
        if self._connectivities is None:
            return None
        cc = self._connected_components
        return 1 if cc is None else cc[0]

    @property
    def distances(self):
        return self._distances

    @property
    def connectivities(self):
        return self._connectivities

    def compute_neighbors(self, n_neighbors: int = 30, n_pcs: int | None = None, *, use_rep: str | None = None,
                          knn: bool = True, method: str | None = "umap", transformer=None,
                          metric="euclidean", metric_kwds=MappingProxyType({}), rng=None, random_state=_UNSET):
        """neighbors/__init__.py:578-673."""
        if transformer is not None and not isinstance(transformer, str):
            n_neighbors = transformer.get_params()["n_neighbors"]
        elif n_neighbors > self._adata.shape[0]:  # very small datasets
            n_neighbors = 1 + int(0.5 * self._adata.shape[0])
            warnings.warn(f"n_obs too small: adjusting to `n_neighbors = {n_neighbors}`", UserWarning, stacklevel=2)
        if method not in _METHODS and method is not None:
            msg = f"`method` needs to be one of {set(_METHODS)}."
            raise ValueError(msg)
        conn_method = method if method in {"gauss", "jaccard", None} else "umap"
        if not knn and not (conn_method == "gauss" and transformer is None):
            msg = f"`method = {method!r} only with `knn = True`."
            raise ValueError(msg)
        if conn_method == "gauss" and not knn:
            msg = "method='gauss' with knn=False builds a dense n x n kernel matrix: not offered on the MI355X path."
            raise NotImplementedError(msg)
        if isinstance(transformer, str) and transformer not in {"sklearn", "pynndescent", "mi355x", "ivf"}:
            msg = f"Unknown transformer: {transformer}. Try passing a class or one of {{'pynndescent', 'sklearn'}}"
            raise ValueError(msg)
        self.n_neighbors = n_neighbors
        self.knn = knn
        x = choose_representation(self._adata, use_rep=use_rep, n_pcs=n_pcs)
        if transformer is None or isinstance(transformer, str):
            if callable(metric) or metric not in METRICS:
                msg = (f"metric={metric!r}: the MI355X kNN kernel offers {METRICS}; pass a `transformer` for other "
                       "metrics.")
                raise NotImplementedError(msg)
            # built-in exact search: (n, k) arrays straight from the device, self column first with an
            # exact 0 (what the reference gets after zeroing the diagonal, neighbors/__init__.py:639-648)
            k = min(n_neighbors, self._adata.n_obs)
            # (the lists stay on the device: they feed the connectivity kernel as they are)
            nprobe = None
            if transformer == "ivf":  # approximate: the cells of the quantiser nearest to the query's own
                from .._settings import settings

                nprobe = int(settings.knn_nprobe)
            knn_indices, knn_distances = knn_search_device(x, k, metric=metric, nprobe=nprobe)
            # (the download of the distances runs on a side stream under the connectivity kernels below)
            pending_distances = sparse_distances_from_device(knn_indices, knn_distances, deferred=True)
        else:  # user-supplied estimator instance: the reference's plug-in route, used as-is (:788, :638)
            self._distances = transformer.fit_transform(x)
            knn_indices, knn_distances = get_indices_distances_from_sparse_matrix(self._distances, n_neighbors)
            pending_distances = None
        self._connectivities = None
        try:
            if conn_method == "umap":
                self._connectivities = _connectivities_umap(knn_indices, knn_distances, self._adata.n_obs)
            elif conn_method in {"gauss", "jaccard"}:  # neighbors/__init__.py:694-708
                self._connectivities = _connectivities_kernel(conn_method, knn_indices, knn_distances, self._adata.n_obs)
        finally:
            if pending_distances is not None:
                self._distances = pending_distances()
        self._cc = None  # connected components (neighbors/__init__.py:666-671) are computed on first use


def _get_metadata(key_added, **params):
    """neighbors/__init__.py:302-316."""
    if key_added is None:
        return "neighbors", dict(connectivities_key="connectivities", distances_key="distances", params=params)
    return key_added, dict(connectivities_key=f"{key_added}_connectivities", distances_key=f"{key_added}_distances",
                           params=params)


def neighbors(  # noqa: PLR0913
    adata,
    n_neighbors: int = 15,
    n_pcs: int | None = None,
    *,
    distances=None,
    use_rep: str | None = None,
    knn: bool = True,
    method: str = "umap",
    transformer=None,
    metric=None,
    metric_kwds=MappingProxyType({}),
    rng=None,
    random_state=_UNSET,
    key_added: str | None = None,
    copy: bool = False,
):
    """Nearest-neighbour distance matrix and neighbourhood graph (drop-in for `scanpy.pp.neighbors`,
    src/scanpy/neighbors/__init__.py:88-299).  Writes `.obsp['distances']` (n_neighbors-1 stored entries
    per row), `.obsp['connectivities']` and `.uns['neighbors']`."""
    if not is_anndata(adata):
        raise TypeError("neighbors() expects an AnnData-like object")
    _, meta_random_state = resolve_seed(rng, random_state)
    if distances is None:
        if metric is None:
            metric = "euclidean"
        adata = adata.copy() if copy else adata
        neighbors_ = Neighbors(adata)
        neighbors_.compute_neighbors(n_neighbors, n_pcs=n_pcs, use_rep=use_rep, knn=knn, method=method,
                                     transformer=transformer, metric=metric, metric_kwds=metric_kwds)
    else:  # neighbors/__init__.py:232-270: precomputed distances, connectivities only
        ignored = set()
        if use_rep is not None:
            ignored.add("use_rep")
        if knn is not True:
            ignored.add("knn")
        if n_pcs is not None:
            ignored.add("n_pcs")
        if metric_kwds:
            ignored.add("metric_kwds")
        if meta_random_state.get("random_state", None) != 0 or rng is not None:
            ignored.add("rng/random_state")
            meta_random_state.pop("random_state", None)
        if ignored:
            warnings.warn(f"Parameter(s) ignored if `distances` is given: {ignored}", UserWarning, stacklevel=2)
        if callable(metric):
            msg = "`metric` must be a string if `distances` is given."
            raise TypeError(msg)
        adata = adata.copy() if copy else adata
        if sparse.issparse(distances):
            distances = sparse.csr_matrix(distances, copy=True)
            distances.setdiag(0)
            distances.eliminate_zeros()
        else:
            distances = np.asarray(distances).copy()
            np.fill_diagonal(distances, 0)
        if method != "umap":
            raise NotImplementedError(f"method={method!r} is outside the MI355X hot path")
        neighbors_ = Neighbors(adata)
        neighbors_.n_neighbors = n_neighbors
        neighbors_.knn = True
        neighbors_._distances = distances
        if sparse.issparse(distances):
            knn_indices, knn_distances = get_indices_distances_from_sparse_matrix(distances, n_neighbors)
        else:
            knn_indices, knn_distances = get_indices_distances_from_dense_matrix(distances, n_neighbors)
        neighbors_._connectivities = _connectivities_umap(knn_indices, knn_distances, adata.n_obs)

    key_added, neighbors_dict = _get_metadata(
        key_added,
        n_neighbors=neighbors_.n_neighbors,
        method=method,
        metric=metric,
        **meta_random_state,
        **({} if not metric_kwds else dict(metric_kwds=metric_kwds)),
        **({} if use_rep is None else dict(use_rep=use_rep)),
        **({} if n_pcs is None else dict(n_pcs=n_pcs)),
    )
    adata.uns[key_added] = neighbors_dict
    adata.obsp[neighbors_dict["distances_key"]] = neighbors_.distances
    adata.obsp[neighbors_dict["connectivities_key"]] = neighbors_.connectivities
    return adata if copy else None
