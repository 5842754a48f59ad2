"""Host <-> device plumbing of the upstream normalisation chain: a host matrix (CSR / CSC / dense, float32 or integer)
becomes a CSR float32 on the device; the passes themselves are the HIP kernels of csrc/preprocess.hip
(`_kernels.pp_*`).  No arithmetic on the matrix happens on the host."""
from __future__ import annotations

import numpy as np
from scipy import sparse


class DeviceMatrix:
    """CSR float32 view of a host matrix on the device.

    kind 'csr' / 'csc': `indptr/indices` describe the compressed axis of the host format.  'csc' only occurs for the
    element-wise pass (log1p keeps the storage format); every row/column pass converts CSC -> CSR on upload, as the
    reference does (`x.tocsr()`, _normalization.py:266-267);
    kind 'dense': full pattern (indptr[i] = i * g, indices = column ids), `data` is the row-major matrix."""

    def __init__(self, kind, shape, indptr, indices, data, host):
        self.kind, self.shape, self.indptr, self.indices, self.data, self.host = kind, shape, indptr, indices, data, host

    @property
    def n_major(self) -> int:
        return int(self.indptr.numel() - 1)


def _check_dtype(dt) -> None:
    if np.issubdtype(dt, np.integer) or np.issubdtype(dt, np.bool_) or dt == np.float32:
        return
    msg = (f"the MI355X normalisation path computes in float32 (what scanpy's readers produce); got dtype {dt}. "
           "Cast with `.astype(np.float32)` first.")
    raise NotImplementedError(msg)


class GpuPPBackend:
    """The only product backend: everything runs through libscanpy_amd.so (raises without a GPU)."""

    def __init__(self):
        from .. import _kernels
        from .._device import require_gpu

        self.K = _kernels
        self.device = require_gpu()

    # ---- transfer ------------------------------------------------------------------------------------------
    def upload(self, x, *, want_csr_rows: bool = True) -> DeviceMatrix:
        import torch

        dev = self.device
        if sparse.issparse(x):
            _check_dtype(x.dtype)
            if x.format == "csc" and want_csr_rows:
                x = x.tocsr()
            elif x.format not in ("csr", "csc"):
                x = x.tocsr()
            if not x.has_canonical_format:
                x = x.copy()
                x.sum_duplicates()
            indptr = torch.from_numpy(x.indptr.astype(np.int64)).to(dev)
            indices = torch.from_numpy(x.indices.astype(np.int32)).to(dev)
            data = torch.from_numpy(np.ascontiguousarray(x.data, dtype=np.float32)).to(dev)
            return DeviceMatrix(x.format, x.shape, indptr, indices, data, x)
        x = np.asarray(x)
        if x.ndim != 2:
            raise ValueError("expected a 2-D matrix")
        _check_dtype(x.dtype)
        n, g = x.shape
        data = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev).reshape(-1)
        indptr = torch.arange(n + 1, dtype=torch.int64, device=dev) * g
        indices = torch.arange(g, dtype=torch.int32, device=dev).repeat(n)
        return DeviceMatrix("dense", x.shape, indptr, indices, data, x)

    def download(self, m: DeviceMatrix):
        """Host matrix of the input's kind with the device values (float32)."""
        data = m.data.cpu().numpy()
        if m.kind == "dense":
            return data.reshape(m.shape)
        cls = type(m.host)
        return cls((data, m.host.indices.copy(), m.host.indptr.copy()), shape=m.shape)

    # ---- passes (device in, small host vectors out) ---------------------------------------------------------
    def row_sums(self, m: DeviceMatrix, col_skip=None) -> np.ndarray:
        import torch

        skip = None if col_skip is None else torch.from_numpy(np.ascontiguousarray(col_skip, dtype=np.int32)).to(self.device)
        return self.K.pp_row_sums(m.indptr, m.indices, m.data, m.n_major, skip).cpu().numpy()

    def row_count_positive(self, m: DeviceMatrix) -> np.ndarray:
        return self.K.pp_row_count_positive(m.indptr, m.data, m.n_major).cpu().numpy().astype(np.int64)

    def count_high(self, m: DeviceMatrix, row_total: np.ndarray, max_fraction: float) -> np.ndarray:
        import torch

        rt = torch.from_numpy(np.ascontiguousarray(row_total, dtype=np.float32)).to(self.device)
        return self.K.pp_count_high(m.indptr, m.indices, m.data, m.n_major, m.shape[1], rt, max_fraction).cpu().numpy()

    def row_divide_(self, m: DeviceMatrix, factor: np.ndarray) -> None:
        import torch

        f = torch.from_numpy(np.ascontiguousarray(factor, dtype=np.float32)).to(self.device)
        self.K.pp_row_divide_(m.indptr, m.data, m.n_major, f)

    def log1p_(self, m: DeviceMatrix, base=None) -> None:
        self.K.pp_log1p_(m.data, base)

    def col_stats(self, m: DeviceMatrix, *, row_mask=None, expm1_scale=None, count_positive: bool = False):
        """-> (sum, sumsq, n_positive or None) per gene over the masked rows (numpy float64 / int64)."""
        import torch

        mask = None if row_mask is None else torch.from_numpy(np.ascontiguousarray(row_mask, dtype=np.uint8)).to(self.device)
        s, sq, npos = self.K.pp_col_stats(m.indptr, m.indices, m.data, m.n_major, m.shape[1], row_mask=mask,
                                          expm1_scale=expm1_scale, count_positive=count_positive)
        return s.cpu().numpy(), sq.cpu().numpy(), None if npos is None else npos.cpu().numpy()

    def nonnegative_integers(self, m: DeviceMatrix) -> bool:
        """`check_nonnegative_integers` (src/scanpy/_utils/__init__.py:761-773) on the stored values, on the device"""
        import torch

        d = m.data
        return not bool(torch.signbit(d).any()) and not bool((d != torch.floor(d)).any())

    def clip_col_sums(self, m: DeviceMatrix, clip_val: np.ndarray, *, row_mask=None):
        """`clip_square_sum` (_highly_variable_genes.py:75-115): per gene, sum and sum of squares of
        min(x, clip_val[gene]) over the stored values of the masked rows -> (sum of squares, sum), float64.
        One sweep of `scamd_pp_col_stats_clip_f32` (the column-statistics kernel with the clip applied before the sums)."""
        import torch

        dev = m.data.device
        clip = torch.from_numpy(np.ascontiguousarray(clip_val, dtype=np.float64)).to(dev)
        mask = None if row_mask is None else torch.from_numpy(np.ascontiguousarray(row_mask, dtype=np.uint8)).to(dev)
        s, sq = self.K.pp_col_stats_clip(m.indptr, m.indices, m.data, m.n_major, m.shape[1], clip, row_mask=mask)
        return sq.cpu().numpy(), s.cpu().numpy()

    def scale_csr_(self, m: DeviceMatrix, std: np.ndarray, *, max_value=None, row_mask=None) -> None:
        import torch

        mask = None if row_mask is None else torch.from_numpy(np.ascontiguousarray(row_mask, dtype=np.uint8)).to(self.device)
        sd = torch.from_numpy(np.ascontiguousarray(std, dtype=np.float64)).to(self.device)
        self.K.pp_scale_csr_(m.indptr, m.indices, m.data, m.n_major, sd, max_value=max_value, row_mask=mask)

    def scale_dense(self, m: DeviceMatrix, mean: np.ndarray, std: np.ndarray, *, max_value=None, row_mask=None,
                    out_f64: bool = True) -> np.ndarray:
        import torch

        mask = None if row_mask is None else torch.from_numpy(np.ascontiguousarray(row_mask, dtype=np.uint8)).to(self.device)
        mu = torch.from_numpy(np.ascontiguousarray(mean, dtype=np.float64)).to(self.device)
        sd = torch.from_numpy(np.ascontiguousarray(std, dtype=np.float64)).to(self.device)
        out = self.K.pp_scale_dense(m.indptr, m.indices, m.data, m.n_major, m.shape[1], mu, sd, max_value=max_value,
                                    row_mask=mask, out_dtype=torch.float64 if out_f64 else torch.float32)
        return out.cpu().numpy()


def default_backend():
    return GpuPPBackend()


def in_memory(x):
    """The upstream chain works on a matrix in memory (most of its passes rewrite it); an on-disk matrix
    (`read_h5ad` / `read_zarr(..., backed='r')`) streams through `pp.pca` and `pp.highly_variable_genes`;
    `pp.normalize_total` / `pp.log1p` turn into pending transforms of it."""
    if getattr(x, "is_backed", False):
        raise NotImplementedError(
            "this function needs the matrix in memory: load it with `adata.X = adata.X.to_memory()` (or read without "
            "backed='r'); on a backed matrix only normalize_total, log1p, highly_variable_genes and pca are offered")
    return x


def mean_var_from_sums(s: np.ndarray, sq: np.ndarray, n: int, *, correction: int = 1):
    """fast_array_utils.stats.mean_var semantics: mean, (E[x^2] - E[x]^2) * n / (n - correction)."""
    mean = s / n
    var = sq / n - mean * mean
    if correction and n > correction:
        var = var * (n / (n - correction))
    return mean, var
