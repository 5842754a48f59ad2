"""CPU stand-in for `scanpy_amd.preprocessing._pca_solver.GpuBackend` (TEST INFRASTRUCTURE).

Lets the host-side solver / sharding logic run under pytest on a machine without a GPU (and under
world_size-2 gloo).  It mimics the kernel contracts: float32 SpMM output, float64-accumulated
transposed product, float64 column sums.  Never imported by the product.
"""
from __future__ import annotations

import numpy as np
import torch
from scipy import sparse


class CpuStubBackend:
    device = torch.device("cpu")

    def upload(self, x_csr):
        x = sparse.csr_matrix(x_csr).astype(np.float32)
        x.sort_indices()
        return (x, None, None, x.shape[0], x.shape[1])

    def transpose(self, a):
        xt = a[0].T.tocsr()
        xt.sort_indices()
        return (xt, None, None, xt.shape[0], xt.shape[1])

    def row_stats(self, a):
        x = a[0].astype(np.float64)
        s = np.asarray(x.sum(axis=1)).ravel()
        q = np.asarray(x.multiply(x).sum(axis=1)).ravel()
        return torch.from_numpy(s), torch.from_numpy(q)

    def spmm(self, a, b, shift):
        y = a[0].astype(np.float64) @ b.numpy().astype(np.float64)
        if shift is not None:
            y = y - shift.numpy().astype(np.float64)[None, :]
        return torch.from_numpy(np.ascontiguousarray(y.astype(np.float32)))

    def spmm_f64acc(self, a, b):
        return torch.from_numpy(np.ascontiguousarray(a[0].astype(np.float64) @ b.numpy().astype(np.float64)))

    def colsum(self, y):
        return torch.from_numpy(y.numpy().astype(np.float64).sum(axis=0))
