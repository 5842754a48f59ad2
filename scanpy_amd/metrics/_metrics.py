"""Graph modularity of a clustering (drop-in for `scanpy.metrics.modularity`,
src/scanpy/metrics/_metrics.py:125-214).  The reference builds an igraph graph from the adjacency
(`_utils/__init__.py:278-304`) and calls `Graph.modularity(codes, 'weight')`; here the same quantity
    Q = 1/(2m) sum_ij (A_ij - k_i k_j / (2m)) delta(c_i, c_j)
is computed by `scamd_modularity_csr_f32` (fixed-point sums on the device, deterministic)."""
from __future__ import annotations

import numpy as np
import pandas as pd
from scipy import sparse

from .._anndata import is_anndata
from .._utils import choose_graph


def _codes(labels) -> np.ndarray:
    """src/scanpy/metrics/_metrics.py:216-223."""
    if isinstance(labels, pd.Series):
        labels = labels.astype("category").array
    if not isinstance(labels, pd.Categorical):
        labels = pd.Categorical(labels)
    return np.asarray(labels.codes)


def modularity_array(connectivities, *, labels, is_directed: bool) -> float:
    """`graph.modularity(codes, 'weight')` for the graph `get_igraph_from_adjacency` would build."""
    import torch

    from .. import _kernels
    from .._device import require_gpu

    adj = sparse.csr_matrix(connectivities) if not sparse.issparse(connectivities) else connectivities.tocsr()
    n = adj.shape[0]
    codes = _codes(labels)
    if len(codes) != n:
        msg = f"Membership vector size differs from number of vertices ({len(codes)} != {n})."
        raise ValueError(msg)
    adj = adj.astype(np.float64)
    if not is_directed:
        # every stored entry is an undirected edge: (i, j) and (j, i) add up; modularity is scale invariant
        adj = (adj + adj.T).tocsr()
    elif (abs(adj - adj.T)).nnz != 0:
        msg = "directed modularity of a non-symmetric adjacency is outside the MI355X hot path"
        raise NotImplementedError(msg)
    adj.sum_duplicates()
    adj.eliminate_zeros()
    adj.sort_indices()
    if adj.nnz == 0 or adj.data.sum() <= 0:
        return float("nan")  # igraph: undefined for a graph without edges
    dev = require_gpu()
    indptr = torch.from_numpy(np.ascontiguousarray(adj.indptr, dtype=np.int64)).to(dev)
    indices = torch.from_numpy(np.ascontiguousarray(adj.indices, dtype=np.int32)).to(dev)
    weights = torch.from_numpy(np.ascontiguousarray(adj.data, dtype=np.float32)).to(dev)
    memb = torch.from_numpy(np.ascontiguousarray(codes, dtype=np.int32)).to(dev)
    return _kernels.modularity(indptr, indices, weights, n, memb)


def modularity_adata(adata, *, labels="leiden", neighbors_key=None, mode="calculate") -> float:
    """src/scanpy/metrics/_metrics.py:177-198."""
    if mode in {"retrieve", "update"} and not isinstance(labels, str):
        msg = "`labels` must be a string when `mode` is `'retrieve'` or `'update'`"
        raise ValueError(msg)
    if mode == "retrieve":
        return adata.uns[labels]["modularity"]
    labels_vec = adata.obs[labels] if isinstance(labels, str) else labels
    connectivities = choose_graph(adata, None, neighbors_key)
    m = modularity(connectivities, labels_vec, is_directed=False)
    if mode == "update":
        adata.uns[labels]["modularity"] = m
    return m


def modularity(adata_or_connectivities, /, labels="leiden", *, neighbors_key=None, is_directed=None,
               mode="calculate") -> float:
    """Modularity of a graph given its connectivities and labels (same signature as the reference)."""
    if is_anndata(adata_or_connectivities):
        if is_directed:
            msg = f"Connectivities stored in `AnnData` are undirected, can’t specify `{is_directed=!r}`"
            raise ValueError(msg)
        return modularity_adata(adata_or_connectivities, labels=labels, neighbors_key=neighbors_key, mode=mode)
    if isinstance(labels, str):
        msg = "`labels` must be provided as array when passing a connectivities array"
        raise TypeError(msg)
    if is_directed is None:
        msg = "`is_directed` must be provided when passing a connectivities array"
        raise TypeError(msg)
    return modularity_array(adata_or_connectivities, labels=labels, is_directed=is_directed)
