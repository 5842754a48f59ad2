"""On-disk CSR matrices streamed by row ranges (out-of-core CSR -> host -> HBM).

The reference's lazy path wraps the `X/{data,indices,indptr}` group of an `.h5ad` / `.zarr` in a dask array of CSR row
blocks (`docs/tutorials/experimental/dask.ipynb:843-879`, `anndata.experimental.read_elem_lazy`).  Here the same group
is a `BackedCsr`: `indptr` is resident (8 bytes per cell), `rows(i0, i1)` decodes exactly the inner chunks of `data` /
`indices` that hold those rows into two contiguous host buffers, and `row_chunks(step)` hands `pp.pca(chunked=True)`
lazy row chunks that a reader thread materialises one ahead of the device (`_pca_solver._ChunkedRows.handles`).
"""
from __future__ import annotations

import threading

import numpy as np
from scipy import sparse


class HostCsrRows:
    """Rows of a CSR matrix as three host arrays (what `GpuBackend.upload` reads) -- no scipy validation pass."""

    def __init__(self, indptr, indices, data, shape, ops=()):
        self.indptr, self.indices, self.data, self.shape = indptr, indices, data, (int(shape[0]), int(shape[1]))
        self.ops = tuple(ops)  # pending device transforms of these rows (see `BackedCsr.with_op`)

    @property
    def nbytes(self) -> int:
        return self.data.nbytes + self.indices.nbytes + 8 * (self.shape[0] + 1)

    def to_scipy(self):
        return sparse.csr_matrix((self.data, self.indices, self.indptr), shape=self.shape)


def _rows_sorted(indptr: np.ndarray, indices: np.ndarray) -> bool:
    if indices.size < 2:
        return True
    up = indices[1:] > indices[:-1]
    starts = indptr[1:-1]  # positions where a new row begins: no order constraint across the boundary
    starts = starts[(starts > 0) & (starts < indices.size)]
    up[starts - 1] = True
    return bool(up.all())


class BackedCsr:
    """A `csr_matrix` group of an AnnData zarr store (`encoding-type: csr_matrix`, arrays data / indices / indptr,
    attribute `shape`), read by row ranges.  `cols` = optional boolean / index column selection applied on load."""

    is_backed = True
    format = "csr"
    ndim = 2

    def __init__(self, group, *, cols: np.ndarray | None = None, _indptr: np.ndarray | None = None, _ops=()):
        self._ops = tuple(_ops)
        enc = group.attrs.get("encoding-type")
        enc = enc.decode() if isinstance(enc, bytes) else enc
        if enc is None and "h5sparse_format" in group.attrs:  # files written by anndata < 0.7
            fmt = group.attrs["h5sparse_format"]
            enc = f"{fmt.decode() if isinstance(fmt, bytes) else fmt}_matrix"
        if enc != "csr_matrix":
            raise ValueError(f"{group.path!r} holds a {enc!r}, not a csr_matrix: only CSR (rows = cells) can be "
                             "streamed by row ranges")
        self.group = group
        self._data, self._indices = group["data"], group["indices"]
        n_rows, n_cols = (int(s) for s in (group.attrs["shape"] if "shape" in group.attrs
                                            else group.attrs["h5sparse_shape"]))
        self._n_cols_disk = n_cols
        self.indptr = np.asarray(group["indptr"].read(), dtype=np.int64) if _indptr is None else _indptr
        if self.indptr.shape != (n_rows + 1,) or self.indptr[0] != 0 or self.indptr[-1] != self._data.shape[0]:
            raise ValueError(f"{group.path!r}: indptr does not describe {n_rows} rows of {self._data.shape[0]} values")
        self.dtype = self._data.dtype
        self._cols = None
        if cols is not None:
            cols = np.asarray(cols)
            self._cols = np.flatnonzero(cols) if cols.dtype == bool else cols.astype(np.int64)
            n_cols = int(self._cols.size)
        self.shape = (n_rows, n_cols)

    @property
    def nnz(self) -> int:
        """stored values on disk (before any column selection)"""
        return int(self.indptr[-1])

    def with_op(self, *op) -> "BackedCsr":
        """the same matrix with one more PENDING transform -- ('row_divide', float32 factor per row) or ('log1p', base)
        -- that is applied on the device to every row chunk after its upload (`pp.normalize_total` / `pp.log1p` on a
        backed matrix): the values on disk are never rewritten and no arithmetic happens on the host."""
        return BackedCsr(self.group, cols=self._cols, _indptr=self.indptr, _ops=self._ops + (tuple(op),))

    def _ops_for(self, i0: int, i1: int) -> tuple:
        return tuple((k, v[i0:i1]) if k == "row_divide" else (k, v) for k, v in self._ops)

    def rows(self, i0: int, i1: int, *, out: tuple[np.ndarray, np.ndarray] | None = None) -> HostCsrRows:
        """`out` = (indices buffer, data buffer) of the on-disk dtypes and at least nnz(rows) elements each: the chunks
        are decoded into their heads instead of fresh arrays (first-touch page faults of fresh memory cost more than
        the decompression itself, so a streaming reader recycles two such pairs)."""
        n = self.shape[0]
        if not 0 <= i0 <= i1 <= n:
            raise IndexError(f"rows [{i0}, {i1}) outside a matrix of {n} rows")
        p0, p1 = int(self.indptr[i0]), int(self.indptr[i1])
        indptr = self.indptr[i0:i1 + 1] - p0
        indices = self._indices.read(p0, p1, out=None if out is None else out[0][:p1 - p0])
        data = self._data.read(p0, p1, out=None if out is None else out[1][:p1 - p0])
        if self._cols is None and _rows_sorted(indptr, indices):
            return HostCsrRows(indptr, indices, data, (i1 - i0, self.shape[1]), self._ops_for(i0, i1))
        m = sparse.csr_matrix((data, indices, indptr), shape=(i1 - i0, self._n_cols_disk))
        if self._cols is not None:
            m = m[:, self._cols]
        if not m.has_sorted_indices:
            m.sort_indices()
        return HostCsrRows(m.indptr.astype(np.int64), m.indices, m.data, m.shape, self._ops_for(i0, i1))

    def absmax(self) -> float | None:
        """max |value| over the stored values, from the `data` array alone (40 % of the bytes of a full pass) -- or
        None under a column selection, where the excluded columns would count."""
        if self._cols is not None or self._ops:  # (pending transforms change the values: the device pass answers)
            return None
        from ._zarr3 import decode_pool

        arr, nnz = self._data, self.nnz
        step = max(1, arr.inner[0])
        local = threading.local()

        def piece(p0: int) -> float:
            p1 = min(p0 + step, nnz)
            buf = getattr(local, "buf", None)
            if buf is None:
                buf = local.buf = np.empty(step, dtype=arr.dtype)
            v = arr.read(p0, p1, out=buf[:p1 - p0], parallel=False)
            return float(max(v.max(), -v.min())) if v.size else 0.0

        return max(decode_pool().map(piece, range(0, nnz, step)), default=0.0)

    def row_chunks(self, step: int, start: int = 0, stop: int | None = None) -> list["LazyRows"]:
        """lazy chunks of `step` rows covering rows [start, stop) -- one rank's row block in a sharded run"""
        stop = self.shape[0] if stop is None else stop
        if not 0 <= start <= stop <= self.shape[0] or step < 1:
            raise IndexError(f"rows [{start}, {stop}) in steps of {step} outside a matrix of {self.shape[0]} rows")
        return [LazyRows(self, i, min(i + step, stop)) for i in range(start, stop, step)]

    def to_memory(self):
        rows = self.rows(0, self.shape[0])
        if not rows.ops:
            return rows.to_scipy()
        from .preprocessing import _csr_device  # pending transforms run on the device, like everywhere else

        be = _csr_device.default_backend()
        m = be.upload(rows.to_scipy())
        apply_ops_pp(be, m, rows.ops)
        return be.download(m)

    def __getitem__(self, index):
        """`x[i0:i1]` -> scipy CSR of those rows; `x[:, mask]` -> a BackedCsr with the column selection pending."""
        if isinstance(index, tuple) and len(index) == 2:
            rows, cols = index
            if not (isinstance(rows, slice) and rows == slice(None)):
                raise IndexError("a backed CSR matrix is subset by columns (`x[:, mask]`) or by a row range (`x[i0:i1]`)")
            cols = np.asarray(cols)
            if self._cols is not None:
                cols = self._cols[np.flatnonzero(cols) if cols.dtype == bool else cols]
            return BackedCsr(self.group, cols=cols, _indptr=self.indptr, _ops=self._ops)
        if isinstance(index, slice) and index == slice(None):
            return self
        if isinstance(index, slice) and index.step in (None, 1):
            i0, i1, _ = index.indices(self.shape[0])
            if self._ops:
                raise IndexError("row slices of a backed matrix with pending transforms are not offered: use to_memory()")
            return self.rows(i0, max(i0, i1)).to_scipy()
        raise IndexError("a backed CSR matrix is subset by columns (`x[:, mask]`) or by a row range (`x[i0:i1]`)")

    def copy(self) -> "BackedCsr":
        return self  # read-only: nothing to protect from writes

    def __repr__(self) -> str:
        pending = f", pending {[k for k, _ in self._ops]}" if self._ops else ""
        return (f"<BackedCsr {self.shape[0]} x {self.shape[1]} {self.dtype} with {self.nnz} stored values at "
                f"{self.group.path!r}{pending}>")


class LazyRows:
    """Rows [i0, i1) of a BackedCsr, not read yet: `load()` -> HostCsrRows."""

    def __init__(self, x: BackedCsr, i0: int, i1: int):
        self.x, self.i0, self.i1 = x, i0, i1
        self.shape = (i1 - i0, x.shape[1])
        nnz = int(x.indptr[i1] - x.indptr[i0])
        self.nbytes = nnz * 8 + 8 * (i1 - i0 + 1)  # as uploaded: float32 values + int32 columns + int64 offsets

        self.nnz = nnz

    def buffers(self, nnz: int) -> tuple[np.ndarray, np.ndarray]:
        """a recyclable (indices, data) buffer pair for `load(out=...)`"""
        return np.empty(nnz, dtype=self.x._indices.dtype), np.empty(nnz, dtype=self.x._data.dtype)

    def load(self, out: tuple[np.ndarray, np.ndarray] | None = None) -> HostCsrRows:
        return self.x.rows(self.i0, self.i1, out=out)


def apply_ops_pp(be, m, ops) -> None:
    """pending row transforms on a `_csr_device.DeviceMatrix` through the normalisation kernels"""
    for kind, arg in ops:
        if kind == "row_divide":
            be.row_divide_(m, arg)
        elif kind == "log1p":
            be.log1p_(m, arg)
        else:
            raise ValueError(f"unknown pending transform {kind!r}")


def is_backed(x) -> bool:
    return bool(getattr(x, "is_backed", False))
