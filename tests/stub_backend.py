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

    def absmax(self, a) -> float:
        return float(np.abs(a[0].data).max()) if a[0].nnz else 0.0

    def gram(self, a, scale_bits: int):
        """Mimics scamd_csr_gram_f32: per-product rounding to 2^-scale_bits, int64 accumulation."""
        x = a[0].astype(np.float64).tocsr()
        g = x.shape[1]
        sc = 2.0 ** scale_bits
        gram = np.zeros((g, g), dtype=np.int64)
        for r0 in range(0, x.shape[0], 2048):
            blk = x[r0:r0 + 2048].toarray()
            # exact products, rounded once each, then summed as integers
            for row in blk:
                nz = np.flatnonzero(row)
                v = row[nz]
                gram[np.ix_(nz, nz)] += np.rint(np.outer(v, v) * sc).astype(np.int64)
        colsum = np.asarray(np.rint(x.multiply(sc).toarray()).sum(axis=0)).ravel().astype(np.int64)
        return torch.from_numpy(gram), torch.from_numpy(colsum)
