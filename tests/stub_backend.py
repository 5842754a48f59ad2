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
        if hasattr(x_csr, "to_scipy"):  # CsrRowsView of the chunked PCA
            x_csr = x_csr.to_scipy()
        x = sparse.csr_matrix(x_csr).astype(np.float32)
        x.sort_indices()
        return (x, None, None, x.shape[0], x.shape[1])

    def apply_ops(self, a, ops):
        """same float32 arithmetic as CpuStubPPBackend.row_divide_ / log1p_ below"""
        x = a[0]
        rows = np.repeat(np.arange(x.shape[0]), np.diff(x.indptr))
        for kind, arg in ops:
            if kind == "row_divide":
                f = np.asarray(arg, dtype=np.float32)
                f = np.where(f == 0, np.float32(1), f)
                x.data = (x.data / f[rows]).astype(np.float32)
            else:
                v = np.log1p(x.data)
                x.data = (v * np.float32(1.0 / np.log(arg))).astype(np.float32) if arg is not None else v

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


class CpuStubPPBackend:
    """CPU stand-in for `scanpy_amd.preprocessing._csr_device.GpuPPBackend` (TEST INFRASTRUCTURE): mimics the kernel
    contracts of csrc/preprocess.hip in numpy so that the host logic of normalize_total / log1p /
    highly_variable_genes / scale (binning, cut-offs, write-back, errors) runs under pytest without a GPU."""

    def upload(self, x, *, want_csr_rows=True):
        from scanpy_amd.preprocessing._csr_device import DeviceMatrix, _check_dtype

        if sparse.issparse(x):
            _check_dtype(x.dtype)
            if x.format == "csc" and want_csr_rows or x.format not in ("csr", "csc"):
                x = x.tocsr()
            x = x.copy()
            x.sum_duplicates()
            m = DeviceMatrix(x.format, x.shape, x.indptr.astype(np.int64), x.indices.astype(np.int32),
                             x.data.astype(np.float32), x)
        else:
            x = np.asarray(x)
            _check_dtype(x.dtype)
            n, g = x.shape
            m = DeviceMatrix("dense", x.shape, np.arange(n + 1, dtype=np.int64) * g,
                             np.tile(np.arange(g, dtype=np.int32), n), x.astype(np.float32).ravel(), x)
        m.rows = np.repeat(np.arange(len(m.indptr) - 1), np.diff(m.indptr))
        return m

    def download(self, m):
        if m.kind == "dense":
            return m.data.reshape(m.shape).copy()
        return type(m.host)((m.data.copy(), m.host.indices.copy(), m.host.indptr.copy()), shape=m.shape)

    def row_sums(self, m, col_skip=None):
        keep = np.ones(m.data.size, bool) if col_skip is None else np.asarray(col_skip)[m.indices] == 0
        n = len(m.indptr) - 1
        return np.bincount(m.rows[keep], weights=m.data[keep].astype(np.float64), minlength=n).astype(np.float32)

    def row_count_positive(self, m):
        n = len(m.indptr) - 1
        return np.bincount(m.rows[m.data > 0], minlength=n).astype(np.int64)

    def count_high(self, m, row_total, max_fraction):
        hi = m.data > np.float32(max_fraction) * row_total.astype(np.float32)[m.rows]
        return np.bincount(m.indices[hi], minlength=m.shape[1]).astype(np.int32)

    def row_divide_(self, m, factor):
        f = np.asarray(factor, dtype=np.float32)
        f = np.where(f == 0, np.float32(1), f)
        m.data = (m.data / f[m.rows]).astype(np.float32)

    def log1p_(self, m, base=None):
        v = np.log1p(m.data)
        m.data = (v * np.float32(1.0 / np.log(base))).astype(np.float32) if base is not None else v

    def col_stats(self, m, *, row_mask=None, expm1_scale=None, count_positive=False):
        sel = np.ones(m.data.size, bool) if row_mask is None else np.asarray(row_mask, bool)[m.rows]
        v = m.data[sel]
        if expm1_scale is not None:
            v = np.expm1(v * np.float32(expm1_scale)).astype(np.float32)
        cols = m.indices[sel]
        g = m.shape[1]
        v64 = v.astype(np.float64)
        return (np.bincount(cols, weights=v64, minlength=g), np.bincount(cols, weights=v64 * v64, minlength=g),
                np.bincount(cols[v > 0], minlength=g).astype(np.int64) if count_positive else None)

    def nonnegative_integers(self, m):
        return not np.signbit(m.data).any() and not np.any((m.data % 1) != 0)

    def clip_col_sums(self, m, clip_val, *, row_mask=None):
        sel = np.ones(m.data.size, bool) if row_mask is None else np.asarray(row_mask, bool)[m.rows]
        v = np.minimum(m.data[sel].astype(np.float64), np.asarray(clip_val, dtype=np.float64)[m.indices[sel]])
        g = m.shape[1]
        return (np.bincount(m.indices[sel], weights=v * v, minlength=g),
                np.bincount(m.indices[sel], weights=v, minlength=g))

    def scale_csr_(self, m, std, *, max_value=None, row_mask=None):
        sel = np.ones(m.data.size, bool) if row_mask is None else np.asarray(row_mask, bool)[m.rows]
        v = m.data[sel].astype(np.float64) / np.asarray(std)[m.indices[sel]]
        if max_value is not None:
            v = np.minimum(v, max_value)
        m.data[sel] = v.astype(np.float32)

    def scale_dense(self, m, mean, std, *, max_value=None, row_mask=None, out_f64=True):
        dt = np.float64 if out_f64 else np.float32
        n, g = m.shape
        dense = np.zeros((n, g), dtype=np.float64)
        dense[m.rows, m.indices] = m.data
        on = np.ones(n, bool) if row_mask is None else np.asarray(row_mask, bool)
        z = (dense[on] - mean).astype(dt).astype(np.float64) / std
        if max_value is not None:
            z = np.clip(z, -max_value, max_value)
        out = dense.astype(dt)
        out[on] = z.astype(dt)
        return out
