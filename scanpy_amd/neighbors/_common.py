"""kNN index/distance <-> CSR conventions of the reference (src/scanpy/neighbors/_common.py), host side."""
from __future__ import annotations

import warnings

import numpy as np
from scipy import sparse


def has_self_column(indices: np.ndarray) -> bool:
    """_common.py:17-22: some row lists itself first (`.any()`, duplicates may displace self)."""
    return bool((indices[:, 0] == np.arange(indices.shape[0])).any())


def remove_self_column(indices, distances):
    """_common.py:25-32."""
    if not has_self_column(indices):
        msg = "The first neighbor should be the cell itself."
        raise AssertionError(msg)
    return indices[:, 1:], distances[:, 1:]


def get_sparse_matrix_from_indices_distances(indices, distances, *, keep_self: bool) -> sparse.csr_matrix:
    """_common.py:35-61: constant-nnz CSR; duplicates stay as explicitly stored zeros."""
    if not keep_self:
        indices, distances = remove_self_column(indices, distances)
    n, k = indices.shape
    indptr = np.arange(0, n * k + 1, k)
    return sparse.csr_matrix((distances.copy().ravel(), indices.copy().ravel(), indptr), shape=(n, n))


def csr_from_trusted_arrays(data, indices, indptr, shape) -> sparse.csr_matrix:
    """scipy CSR around arrays that come straight out of our kernels (consistent by construction), skipping the
    constructor's O(nnz) passes (index-dtype scan of min / max, format checks, copies): ~60 ms per matrix at 1M cells,
    more than the kernels that produced it.  `indices` and `indptr` must share one integer dtype."""
    assert indices.dtype == indptr.dtype and indptr.shape[0] == shape[0] + 1
    m = sparse.csr_matrix(shape, dtype=data.dtype)
    m.data, m.indices, m.indptr = data, indices, indptr
    return m


def sparse_distances_from_device(idx, dist, deferred: bool = False):
    """`get_sparse_matrix_from_indices_distances(..., keep_self=False)` (_common.py:35-61) for the (n, k) lists of the
    built-in search while they are still torch tensors: the self column is dropped on the device and the two arrays
    arrive on the host in their final layout (no host-side slicing copies)."""
    import torch

    n, k = idx.shape
    if k < 1 or not bool((idx[:, 0] == torch.arange(n, device=idx.device, dtype=idx.dtype)).any()):
        msg = "The first neighbor should be the cell itself."
        raise AssertionError(msg)
    nnz = n * (k - 1)
    itype = torch.int32 if max(nnz, n) < 2**31 else torch.int64
    from .._device import to_host

    indices_d = idx[:, 1:].to(itype).contiguous().reshape(-1)
    data_d = dist[:, 1:].contiguous().reshape(-1)

    def finish(indices, data):
        indptr = np.arange(0, nnz + 1, k - 1, dtype=indices.dtype) if k > 1 else np.zeros(n + 1, dtype=indices.dtype)
        return csr_from_trusted_arrays(data, indices, indptr, (n, n))

    if not deferred:
        return finish(to_host(indices_d), to_host(data_d))
    # deferred: the two arrays cross the link on a side stream while the caller's next kernels (the connectivities) run on the
    # compute stream; the returned callable waits for them and builds the matrix (host-to-host metric)
    from .._device import _PINNED_RESULT_MAX

    nbytes = max(indices_d.numel() * indices_d.element_size(), data_d.numel() * data_d.element_size())
    if not indices_d.is_cuda or nbytes < (8 << 20) or nbytes > _PINNED_RESULT_MAX:
        out = finish(to_host(indices_d), to_host(data_d))
        return lambda: out
    dev = indices_d.device
    main = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(device=dev)
    ready = torch.cuda.Event()
    ready.record(main)
    hosts = []
    with torch.cuda.stream(side):
        side.wait_event(ready)
        for t in (indices_d, data_d):
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            t.record_stream(side)
            hosts.append(h)
    done = torch.cuda.Event()
    done.record(side)

    def wait():
        done.synchronize()
        return finish(hosts[0].numpy(), hosts[1].numpy())

    return wait


def graph_from_device(indptr, indices, data, n_obs: int) -> sparse.csr_matrix:
    """connectivities CSR (device tensors of a `_kernels.*connectivities` / `fuzzy_simplicial_set` call) -> scipy"""
    import torch

    itype = torch.int32 if max(int(indices.numel()), n_obs) < 2**31 else torch.int64
    from .._device import to_host

    m = csr_from_trusted_arrays(to_host(data), to_host(indices.to(itype)), to_host(indptr.to(itype)), (n_obs, n_obs))
    # the C ABI promises sorted, duplicate-free rows (include/scanpy_amd.h): recording it saves scipy (and tl.leiden)
    # an O(nnz) check per use
    m.has_canonical_format = True
    return m


def get_indices_distances_from_dense_matrix(d: np.ndarray, n_neighbors: int):
    """_common.py:64-71."""
    sample_range = np.arange(d.shape[0])[:, None]
    indices = np.argpartition(d, n_neighbors - 1, axis=1)[:, :n_neighbors]
    indices = indices[sample_range, np.argsort(d[sample_range, indices])]
    return indices, d[sample_range, indices]


def _ind_dist_shortcut(d: sparse.csr_matrix):
    """_common.py:126-143."""
    nnzs = np.diff(d.indptr)
    if len(nnzs) == 0 or not (nnzs == nnzs[0]).all():
        warnings.warn("Sparse matrix has no constant number of neighbors per row. "
                      "Cannot efficiently get indices and distances.", RuntimeWarning, stacklevel=3)
        return None
    n_obs, n_neighbors = d.shape[0], int(nnzs[0])
    return d.indices.reshape(n_obs, n_neighbors), d.data.reshape(n_obs, n_neighbors)


def _ind_dist_slow(d: sparse.csr_matrix, n_neighbors: int):
    """_common.py:101-123."""
    indices = np.zeros((d.shape[0], n_neighbors), dtype=int)
    distances = np.zeros((d.shape[0], n_neighbors), dtype=d.dtype)
    m1 = n_neighbors - 1
    for i in range(indices.shape[0]):
        row = d[i]
        cols, vals = row.indices, row.data  # 'true' and 'spurious' zeros alike
        indices[i, 0] = i
        distances[i, 0] = 0
        if len(cols) > m1:
            order = np.argsort(vals)[:m1]
            indices[i, 1:] = cols[order]
            distances[i, 1:] = vals[order]
        else:
            indices[i, 1 : 1 + len(cols)] = cols
            distances[i, 1 : 1 + len(cols)] = vals
    return indices, distances


def get_indices_distances_from_sparse_matrix(d, n_neighbors: int):
    """_common.py:74-98: first column = the cell itself, at most n_neighbors columns."""
    d = sparse.csr_matrix(d)
    shortcut = _ind_dist_shortcut(d)
    indices, distances = shortcut if shortcut is not None else _ind_dist_slow(d, n_neighbors)
    if not has_self_column(indices):  # RAPIDS-style rows lack the self column
        indices = np.hstack([np.arange(indices.shape[0])[:, None], indices])
        distances = np.hstack([np.zeros(distances.shape[0])[:, None], distances])
    if indices.shape[1] > n_neighbors:
        indices, distances = indices[:, :n_neighbors], distances[:, :n_neighbors]
    return indices, distances
